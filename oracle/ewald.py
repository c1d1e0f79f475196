"""CPU restatement of the Ewald-type grid evaluator — TEST INFRASTRUCTURE ONLY
(never imported by ipde_amd).

Follows ipde/grid_evaluators/scalar_grid_evaluator.py: `excisor` (:22-44) with a
Kaiser-Bessel step in place of the Slepian one, the local/spread pass
`ewald_local_freespace` (:189-229) / `ewald_local_periodic` (:131-178), the
truncated-kernel operator T/TH (:277-293) and the assembly (:299-307); kernel
definitions from laplace_grid_evaluator.py:8-33 and
modified_helmholtz_grid_evaluator.py:8-17 (`gf`, `fs`, `ifs`, `trunc_sgf`; pinned by
tests/golden/grid_evaluator_kernels.npz, generated from the reference's functions).
Instead of differentiating the screened kernel numerically (:92-121) the spread density
is the closed form  rho = 2 chi' G' + (chi'' + chi'/r) G.
"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import i0e, i1e, j0, j1, k0, k1


class KaiserBesselStep(object):
    """bump(x) = I0(beta sqrt(1-x^2))/I0(beta) on [-1,1]; step = normalised integral."""

    def __init__(self, beta, deg=None):
        self.beta = float(beta)
        deg = int(2 * beta + 40) if deg is None else deg
        xc = np.cos(np.pi * (np.arange(deg + 1) + 0.5) / (deg + 1))
        c = C.chebfit(xc, self.bump(xc), deg)
        c[1::2] = 0.0
        s = C.chebint(c, lbnd=-1.0)
        self.norm = float(C.chebval(1.0, s))
        self.step_c = s / self.norm

    def bump(self, x):
        s = np.sqrt(np.maximum(1.0 - x * x, 0.0))
        return i0e(self.beta * s) * np.exp(self.beta * (s - 1.0))

    def dbump(self, x):
        s = np.sqrt(np.maximum(1.0 - x * x, 1e-300))
        return -self.beta * x * i1e(self.beta * s) * np.exp(self.beta * (s - 1.0)) / s

    def step(self, x):
        return C.chebval(x, self.step_c)

    def chi(self, r, R):
        """chi(r) = step(1 - 2r/R) (1 at r = 0, 0 at r = R) and its first two r-derivatives"""
        x = 1.0 - 2.0 * r / R
        return self.step(x), -2.0 / R * self.bump(x) / self.norm, 4.0 / R ** 2 * self.dbump(x) / self.norm


# -- kernels -----------------------------------------------------------------
def laplace_gf(r):
    return -np.log(r) / (2 * np.pi)


def laplace_fs(kx, ky):
    return -(kx * kx + ky * ky)


def laplace_ifs(kx, ky):
    lap = laplace_fs(kx, ky)
    lap[0, 0] = 1.0
    out = 1.0 / lap
    out[0, 0] = 0.0
    return out


def laplace_trunc_sgf(k, L):
    k = np.asarray(k, dtype=float)
    out = np.empty(k.shape)
    z = k == 0
    kn = k[~z]
    out[z] = -L ** 2 * np.log(L) + L ** 2 * (1 + 2 * np.log(L)) / 4
    out[~z] = (1.0 - j0(L * kn)) / kn ** 2 - L * np.log(L) * j1(L * kn) / kn
    return out


def modhelm_gf(r, helmholtz_k):
    return k0(helmholtz_k * r) / (2 * np.pi)


def modhelm_fs(kx, ky, helmholtz_k):
    return -helmholtz_k ** 2 - kx ** 2 - ky ** 2


def modhelm_ifs(kx, ky, helmholtz_k):
    return 1.0 / modhelm_fs(kx, ky, helmholtz_k)


def modhelm_trunc_sgf(k, L, helmholtz_k):
    kk0, kk1 = k0(L * helmholtz_k), k1(L * helmholtz_k)
    return (1.0 + L * k * j1(L * k) * kk0 - L * helmholtz_k * j0(L * k) * kk1) / (k ** 2 + helmholtz_k ** 2)


def radial_parts(mol, r, R, helmholtz_k=None):
    """(chi G)(r) and rho(r) = 2 chi' G' + (chi'' + chi'/r) G"""
    chi, c1, c2 = mol.chi(r, R)
    if helmholtz_k is None:
        G, Gp = -np.log(r) / (2 * np.pi), -1.0 / (2 * np.pi * r)
    else:
        G, Gp = k0(helmholtz_k * r) / (2 * np.pi), -helmholtz_k * k1(helmholtz_k * r) / (2 * np.pi)
    # (rho is zero to rounding near r = 0; chi' is ~1e-16 rather than 0 there, so the 1/r
    # factors are cut off below 1e-6 R)
    return chi * G, np.where(r > 1e-6 * R, 2 * c1 * Gp + (c2 + c1 / r) * G, 0.0)


def spread(mol, sx, sy, q, x0, y0, h, sw, nbx, nby, offx, offy, periodic, helmholtz_k=None):
    R = sw * h
    ul, op = np.zeros((nbx, nby)), np.zeros((nbx, nby))
    for j in range(len(sx)):
        cx, cy = int(np.floor((sx[j] - x0) / h)), int(np.floor((sy[j] - y0) / h))
        I = np.arange(cx - sw - 1, cx + sw + 2)
        J = np.arange(cy - sw - 1, cy + sw + 2)
        X, Y = np.meshgrid(x0 + I * h, y0 + J * h, indexing='ij')
        d = np.hypot(X - sx[j], Y - sy[j])
        m = (d <= R) & (d > 0)
        loc, rho = radial_parts(mol, np.where(m, d, 0.5 * R), R, helmholtz_k)
        gi, gj = I + offx, J + offy
        if periodic:
            gi, gj = gi % nbx, gj % nby
        np.add.at(ul, np.ix_(gi, gj), np.where(m, loc, 0.0) * q[j])
        np.add.at(op, np.ix_(gi, gj), np.where(m, rho, 0.0) * q[j])
    return ul, op


def truncated_operator(n_big, h, L, helmholtz_k=None):
    """TH with the kernel's origin at index (0, 0): spectrum (n_big, n_big) such that
    ifft2(fft2(f) TH) is the free-space convolution for f supported in half the box
    (reference :283-293 builds the same on the 2x finer spectral grid and crops)."""
    N = 2 * n_big
    kv = np.fft.fftfreq(N, h / (2 * np.pi))
    kk = np.hypot(*np.meshgrid(kv, kv, indexing='ij'))
    ts = laplace_trunc_sgf(kk, L) if helmholtz_k is None else modhelm_trunc_sgf(kk, L, helmholtz_k)
    T = np.fft.ifft2(ts).real / (h * h)           # kernel samples, origin at (0,0), period N
    idx = np.r_[0:n_big // 2, N - n_big // 2:N]    # the n_big entries nearest the origin
    return np.fft.fft2(T[np.ix_(idx, idx)]) * (h * h)


def freespace_eval(sx, sy, q, xv, yv, sw, beta=None, helmholtz_k=None):
    """sum_j q_j G(|x - s_j|) on the (n, n) grid through the split."""
    n = len(xv)
    h = xv[1] - xv[0]
    mol = KaiserBesselStep(1.6 * sw if beta is None else beta)
    E = n + 2 * sw
    big = 2 * E
    off = sw
    ul, op = spread(mol, sx, sy, q, xv[0], yv[0], h, sw, big, big, off, off, False, helmholtz_k)
    drange = xv[-1] - xv[0] + h
    TH = truncated_operator(big, h, 2.5 * drange, helmholtz_k)
    far = np.fft.ifft2(np.fft.fft2(op) * TH).real
    return (ul + far)[off:off + n, off:off + n]


def periodic_eval(sx, sy, q, xv, yv, sw, beta=None, helmholtz_k=None):
    """the periodic-image sum (Laplace: zero-mean part, needs sum q = 0)"""
    nx, ny = len(xv), len(yv)
    h = xv[1] - xv[0]
    mol = KaiserBesselStep(1.6 * sw if beta is None else beta)
    ul, op = spread(mol, sx, sy, q, xv[0], yv[0], h, sw, nx, ny, 0, 0, True, helmholtz_k)
    kx, ky = np.meshgrid(np.fft.fftfreq(nx, h / (2 * np.pi)), np.fft.fftfreq(ny, h / (2 * np.pi)),
                         indexing='ij')
    isym = -laplace_ifs(kx, ky) if helmholtz_k is None else -modhelm_ifs(kx, ky, helmholtz_k)
    return ul + np.fft.ifft2(np.fft.fft2(op) * isym).real


# -- Stokes (stokeslet with pressure) through Laplace potentials ---------------
def stokes_freespace_eval(sx, sy, fx, fy, xv, yv, sw, beta=None):
    """(u, v, p) of sum_s stokeslet(x - y_s) f_s on the grid through the Laplace split.
    With G = -log r / (2 pi) and G[q] = sum_s G(x - y_s) q_s:
        u_i = G[f_i]/2 - x_i d_j G[f_j]/2 + d_j G[y_i f_j]/2,      p = -d_j G[f_j]
    (r_i r_j / r^2 f_j = (x_i - y_i) d_j(log r) f_j; the classical reduction of the Stokes
    FMM to Laplace FMMs).  Near part: chi G and its radial derivative directly, with
    (x_i - y_i) formed exactly; far part: six spread densities rho q, one padded
    convolution each, the derivative as i k in Fourier space; coordinates are centred on
    the grid to keep the far-field combination x_i B - C_i well conditioned."""
    n = len(xv)
    h = xv[1] - xv[0]
    R = sw * h
    mol = KaiserBesselStep(1.6 * sw if beta is None else beta)
    E = n + 2 * sw
    big = 2 * E
    off = sw
    cx, cy = 0.5 * (xv[0] + xv[-1]), 0.5 * (yv[0] + yv[-1])
    loc = np.zeros((3, big, big))
    op = np.zeros((6, big, big))
    for j in range(len(sx)):
        ix, iy = int(np.floor((sx[j] - xv[0]) / h)), int(np.floor((sy[j] - yv[0]) / h))
        I = np.arange(ix - sw - 1, ix + sw + 2)
        J = np.arange(iy - sw - 1, iy + sw + 2)
        X, Y = np.meshgrid(xv[0] + I * h, yv[0] + J * h, indexing='ij')
        rx, ry = X - sx[j], Y - sy[j]
        d = np.hypot(rx, ry)
        m = (d <= R) & (d > 0)
        dd = np.where(m, d, 0.5 * R)
        chi, c1, c2 = mol.chi(dd, R)
        G, Gp = -np.log(dd) / (2 * np.pi), -1.0 / (2 * np.pi * dd)
        lg, dlg = chi * G, c1 * G + chi * Gp
        rho = np.where(dd > 1e-6 * R, 2 * c1 * Gp + (c2 + c1 / dd) * G, 0.0)
        rf = (rx * fx[j] + ry * fy[j]) / dd
        sl = np.ix_(I + off, J + off)
        loc[0][sl] += np.where(m, 0.5 * (lg * fx[j] - dlg * rx * rf), 0.0)
        loc[1][sl] += np.where(m, 0.5 * (lg * fy[j] - dlg * ry * rf), 0.0)
        loc[2][sl] += np.where(m, -dlg * rf, 0.0)
        yx, yy = sx[j] - cx, sy[j] - cy
        for k, q in enumerate((fx[j], fy[j], yx * fx[j], yx * fy[j], yy * fx[j], yy * fy[j])):
            op[k][sl] += np.where(m, rho * q, 0.0)
    drange = xv[-1] - xv[0] + h
    TH = truncated_operator(big, h, 2.5 * drange, None)
    kv = np.fft.fftfreq(big, h / (2 * np.pi))
    if big % 2 == 0:
        kv[big // 2] = 0.0
    ikx, iky = 1j * kv[:, None], 1j * kv[None, :]
    F = np.fft.fft2(op, axes=(1, 2))
    B = np.fft.ifft2(TH * (ikx * F[0] + iky * F[1])).real
    ux = np.fft.ifft2(TH * 0.5 * (F[0] + ikx * F[2] + iky * F[3])).real
    uy = np.fft.ifft2(TH * 0.5 * (F[1] + ikx * F[4] + iky * F[5])).real
    xb = xv[0] + h * (np.arange(big) - off) - cx
    yb = yv[0] + h * (np.arange(big) - off) - cy
    u = loc[0] + ux - 0.5 * xb[:, None] * B
    v = loc[1] + uy - 0.5 * yb[None, :] * B
    p = loc[2] - B
    s = slice(off, off + n)
    return u[s, s], v[s, s], p[s, s]
