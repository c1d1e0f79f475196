"""Shared geometry / comparison helpers for the tests (pure numpy)."""
import numpy as np


class Curve:
    """Closed curve r(theta) = 1 + a cos(f theta) (our reading of pybie2d's
    star(n, a, f), reference examples/interior_poisson.py:41), scaled by `scale`
    and shifted by `center`; equispaced in theta, outward unit normals,
    weights = speed * dtheta."""

    def __init__(self, n, a=0.2, f=5, scale=1.0, center=(0.0, 0.0)):
        t = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
        r = 1.0 + a * np.cos(f * t)
        rp = -a * f * np.sin(f * t)
        self.x = scale * r * np.cos(t) + center[0]
        self.y = scale * r * np.sin(t) + center[1]
        xp = scale * (rp * np.cos(t) - r * np.sin(t))
        yp = scale * (rp * np.sin(t) + r * np.cos(t))
        self.speed = np.hypot(xp, yp)
        self.normal_x = yp / self.speed
        self.normal_y = -xp / self.speed
        self.tangent_x = xp / self.speed
        self.tangent_y = yp / self.speed
        self.dt = 2 * np.pi / n
        self.weights = self.speed * self.dt
        self.N = n
        self.t = t
        self.scale = scale
        self.a = a
        self.f = f
        self.center = center

    def radius_at(self, theta):
        return self.scale * (1.0 + self.a * np.cos(self.f * theta))

    def get_stacked_boundary(self):
        return np.vstack([self.x, self.y])


class Points:
    def __init__(self, x, y):
        self.x = np.ascontiguousarray(x, dtype=float).ravel()
        self.y = np.ascontiguousarray(y, dtype=float).ravel()
        self.N = self.x.shape[0]

    def get_stacked_boundary(self):
        return np.vstack([self.x, self.y])


def grid_targets(curve, ngrid, lim=1.5, clearance=5.0, inside_only=False):
    """Regular grid on [-lim, lim)^2 minus the points within clearance*h of the
    curve (the solver never evaluates on-surface)."""
    v = np.linspace(-lim, lim, ngrid, endpoint=False)
    h = v[1] - v[0]
    X, Y = np.meshgrid(v, v, indexing="ij")
    x, y = X.ravel(), Y.ravel()
    xc, yc = x - curve.center[0], y - curve.center[1]
    theta = np.arctan2(yc, xc)
    rr = np.hypot(xc, yc)
    rb = curve.radius_at(theta)
    keep = np.abs(rr - rb) > clearance * h * 1.5
    if inside_only:
        keep &= rr < rb
    return Points(x[keep], y[keep]), h


def rel_err(a, b):
    """max |a-b| / max |b|"""
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
