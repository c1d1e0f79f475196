"""The layer-potential kernels nothing in the reference tree can compute (Laplace / modified-
Helmholtz double layers, Stokes velocities: pybie2d / pyfmmlib2d arithmetic, absent) tied to the
kernels that ARE pinned by reference-computed fixtures (tests/golden/layer_kernels.npz: Laplace
SLP, modified-Helmholtz SLP, Stokes pressures) through derivative relations — none of these checks
evaluates the unpinned kernel by its own closed formula on the checking side:

  Laplace / MH DLP   D[tau](x)  = sum_j tau_j w_j  n_j . grad_s G(x - s_j)          (source-side FD of the SLP)
  stokeslet          u_i        = ( G[f_i] - x_i d_j G[f_j] + d_j G[y_i f_j] ) / 2  (G = pinned Laplace SLP)
  stresslet          u_i        = p^S[g n_i] - d_k U_i[g n_k] - d_i U_k[g n_k]      (-sigma.n of the stokeslet)
  momentum, mass     -grad p + lap u = 0,  div u = 0                                (FD of the outputs)

Derivatives are 8th-order central differences of the PINNED functions, so the bar is the
differences' own accuracy (1e-9 of max|u|), which still fixes every sign, factor and convention.
(CPU; the HIP kernels get the same relations in tests/test_layer_gpu.py.)"""
import numpy as np
import pytest

from oracle import layer_potentials as olp

# 8th-order central first derivative: offsets +-1..4
_C1 = {1: 4.0 / 5.0, 2: -1.0 / 5.0, 3: 4.0 / 105.0, 4: -1.0 / 280.0}
# 8th-order central second derivative
_C2_0 = -205.0 / 72.0
_C2 = {1: 8.0 / 5.0, 2: -1.0 / 5.0, 3: 8.0 / 315.0, 4: -1.0 / 560.0}


def _setup(ns=60, nt=40, seed=0):
    rng = np.random.default_rng(seed)
    t = 2 * np.pi * np.arange(ns) / ns
    r = 1.0 + 0.2 * np.cos(3 * t)
    sx, sy = r * np.cos(t), r * np.sin(t)
    dx = -0.6 * np.sin(3 * t) * np.cos(t) - r * np.sin(t)
    dy = -0.6 * np.sin(3 * t) * np.sin(t) + r * np.cos(t)
    sp = np.hypot(dx, dy)
    nx, ny = dy / sp, -dx / sp
    w = sp * 2 * np.pi / ns
    # targets well away from the curve (inside, |x| <= 0.35, and outside, 2.2 <= |x| <= 3)
    a = rng.uniform(0, 2 * np.pi, nt)
    rad = np.where(np.arange(nt) % 2 == 0, rng.uniform(0.05, 0.35, nt), rng.uniform(2.2, 3.0, nt))
    tx, ty = rad * np.cos(a), rad * np.sin(a)
    return dict(sx=sx, sy=sy, nx=nx, ny=ny, w=w, tx=tx, ty=ty, rng=rng)


def d_target(fn, tx, ty, axis, eps):
    """8th-order central difference of fn(tx, ty) along x (axis 0) or y (axis 1)."""
    out = 0.0
    for m, c in _C1.items():
        ex, ey = (m * eps, 0.0) if axis == 0 else (0.0, m * eps)
        out = out + c * (fn(tx + ex, ty + ey) - fn(tx - ex, ty - ey))
    return out / eps


def lap_target(fn, tx, ty, eps):
    out = 2 * _C2_0 * fn(tx, ty)
    for m, c in _C2.items():
        out = out + c * (fn(tx + m * eps, ty) + fn(tx - m * eps, ty) + fn(tx, ty + m * eps) + fn(tx, ty - m * eps))
    return out / eps ** 2


def dlp_from_slp(slp, S, tau, eps):
    """sum_j tau_j w_j n_j . grad_s G: the SLP with the sources moved along their normals."""
    out = 0.0
    for m, c in _C1.items():
        out = out + c * (slp(S["sx"] + m * eps * S["nx"], S["sy"] + m * eps * S["ny"], tau)
                         - slp(S["sx"] - m * eps * S["nx"], S["sy"] - m * eps * S["ny"], tau))
    return out / eps


def test_laplace_dlp_is_normal_derivative_of_pinned_slp():
    S = _setup()
    tau = S["rng"].standard_normal(S["sx"].size)
    slp = lambda sx, sy, q: olp.laplace_layer_apply(sx, sy, S["tx"], S["ty"], charge=q, weights=S["w"])
    ref = dlp_from_slp(slp, S, tau, 0.01)
    got = olp.laplace_layer_apply(S["sx"], S["sy"], S["tx"], S["ty"], dipstr=tau, weights=S["w"],
                                  nx=S["nx"], ny=S["ny"])
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


@pytest.mark.parametrize("k", [0.7, 10.0])
def test_modhelm_dlp_is_normal_derivative_of_pinned_slp(k):
    S = _setup()
    tau = S["rng"].standard_normal(S["sx"].size)
    slp = lambda sx, sy, q: olp.modified_helmholtz_layer_apply(sx, sy, S["tx"], S["ty"], k, charge=q,
                                                               weights=S["w"])
    ref = dlp_from_slp(slp, S, tau, 0.004 if k > 5 else 0.01)
    got = olp.modified_helmholtz_layer_apply(S["sx"], S["sy"], S["tx"], S["ty"], k, dipstr=tau,
                                             weights=S["w"], nx=S["nx"], ny=S["ny"])
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


def stokeslet_from_laplace(S, f, tx, ty, eps=0.01, lap_slp=None):
    """u_i = ( G[f_i] - x_i d_j G[f_j] + d_j G[y_i f_j] ) / 2 with G the Laplace single layer."""
    if lap_slp is None:
        lap_slp = lambda q, x, y: olp.laplace_layer_apply(S["sx"], S["sy"], x, y, charge=q, weights=S["w"])
    ys = (S["sx"], S["sy"])
    xs = (tx, ty)
    u = []
    for i in range(2):
        ui = lap_slp(f[i], tx, ty)
        for j in range(2):
            ui = ui - xs[i] * d_target(lambda x, y: lap_slp(f[j], x, y), tx, ty, j, eps)
            ui = ui + d_target(lambda x, y: lap_slp(ys[i] * f[j], x, y), tx, ty, j, eps)
        u.append(0.5 * ui)
    return u


def test_stokeslet_velocity_from_pinned_laplace_slp():
    S = _setup()
    f = S["rng"].standard_normal((2, S["sx"].size))
    u_ref, v_ref = stokeslet_from_laplace(S, f, S["tx"], S["ty"])
    u, v, _ = olp.stokes_layer_apply(S["sx"], S["sy"], S["tx"], S["ty"], force=f, weights=S["w"])
    scale = max(np.max(np.abs(u_ref)), np.max(np.abs(v_ref)))
    assert max(np.max(np.abs(u - u_ref)), np.max(np.abs(v - v_ref))) < 1e-9 * scale


def stresslet_from_stokeslet(S, g, tx, ty, stokeslet, eps=0.01):
    """u_i = p^S[g n_i] - sum_k d_k U_i[g n_k] - sum_k d_i U_k[g n_k]; stokeslet(force, x, y) -> (u, v, p)."""
    n = (S["nx"], S["ny"])
    out = []
    for i in range(2):
        ui = stokeslet(g * n[i][None, :], tx, ty)[2]
        for k in range(2):
            fk = g * n[k][None, :]
            ui = ui - d_target(lambda x, y: stokeslet(fk, x, y)[i], tx, ty, k, eps)
            ui = ui - d_target(lambda x, y: stokeslet(fk, x, y)[k], tx, ty, i, eps)
        out.append(ui)
    return out


def test_stresslet_velocity_is_traction_of_stokeslet():
    S = _setup()
    g = S["rng"].standard_normal((2, S["sx"].size))
    sto = lambda f, x, y: olp.stokes_layer_apply(S["sx"], S["sy"], x, y, force=f, weights=S["w"])
    u_ref, v_ref = stresslet_from_stokeslet(S, g, S["tx"], S["ty"], sto)
    u, v, _ = olp.stokes_layer_apply(S["sx"], S["sy"], S["tx"], S["ty"], dipstr=g, weights=S["w"],
                                     nx=S["nx"], ny=S["ny"])
    scale = max(np.max(np.abs(u_ref)), np.max(np.abs(v_ref)))
    assert max(np.max(np.abs(u - u_ref)), np.max(np.abs(v - v_ref))) < 1e-9 * scale


@pytest.mark.parametrize("layer", ["slp", "dlp"])
def test_stokes_fields_satisfy_momentum_and_mass(layer):
    """-grad p + lap u = 0 and div u = 0 away from the curve: ties the (unpinned) velocities to
    the (pinned) pressures."""
    S = _setup()
    d = S["rng"].standard_normal((2, S["sx"].size))
    kw = dict(force=d) if layer == "slp" else dict(dipstr=d, nx=S["nx"], ny=S["ny"])
    fn = lambda x, y: olp.stokes_layer_apply(S["sx"], S["sy"], x, y, weights=S["w"], **kw)
    eps = 0.02
    tx, ty = S["tx"], S["ty"]
    px = d_target(lambda x, y: fn(x, y)[2], tx, ty, 0, eps)
    py = d_target(lambda x, y: fn(x, y)[2], tx, ty, 1, eps)
    lu = lap_target(lambda x, y: fn(x, y)[0], tx, ty, eps)
    lv = lap_target(lambda x, y: fn(x, y)[1], tx, ty, eps)
    div = d_target(lambda x, y: fn(x, y)[0], tx, ty, 0, eps) + d_target(lambda x, y: fn(x, y)[1], tx, ty, 1, eps)
    scale = max(np.max(np.abs(px)), np.max(np.abs(py)))
    assert scale > 0
    assert max(np.max(np.abs(lu - px)), np.max(np.abs(lv - py))) < 1e-7 * scale
    assert np.max(np.abs(div)) < 1e-8 * scale
