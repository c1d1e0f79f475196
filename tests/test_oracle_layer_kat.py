"""Pin the oracle's layer-potential conventions with analytic known answers (CPU).

The reference holds no tests and no golden vectors for this arithmetic (it lives
in un-vendored pybie2d / pyfmmlib2d), so the oracle is checked against closed-form
potential-theory identities and the numpy and C restatements against each other.
"""
import numpy as np
import pytest
from scipy.special import i0, i1, k0, k1

import oracle
from oracle import layer_potentials as olp
from util import Curve, Points, rel_err


@pytest.fixture(scope="module")
def star():
    return Curve(600, a=0.2, f=5)


@pytest.fixture(scope="module")
def circle():
    return Curve(400, a=0.0)


def _pts_inside():
    rng = np.random.default_rng(1)
    r = 0.5 * np.sqrt(rng.uniform(0, 1, 40))
    t = rng.uniform(0, 2 * np.pi, 40)
    return r * np.cos(t), r * np.sin(t)


def _pts_outside():
    rng = np.random.default_rng(2)
    r = rng.uniform(1.6, 3.0, 40)
    t = rng.uniform(0, 2 * np.pi, 40)
    return r * np.cos(t), r * np.sin(t)


def test_laplace_dlp_gauss_identity(star):
    one = np.ones(star.N)
    xi, yi = _pts_inside()
    xo, yo = _pts_outside()
    ui = olp.laplace_layer_apply(star.x, star.y, xi, yi, dipstr=one, weights=star.weights,
                                 nx=star.normal_x, ny=star.normal_y)
    uo = olp.laplace_layer_apply(star.x, star.y, xo, yo, dipstr=one, weights=star.weights,
                                 nx=star.normal_x, ny=star.normal_y)
    assert np.max(np.abs(ui + 1.0)) < 1e-12   # -1 inside (jump D - I/2, interior_poisson.py:19)
    assert np.max(np.abs(uo)) < 1e-12         # 0 outside


def test_laplace_slp_circle_eigenfunction(circle):
    m = 3
    sigma = np.cos(m * circle.t)
    xi, yi = _pts_inside()
    u = olp.laplace_layer_apply(circle.x, circle.y, xi, yi, charge=sigma, weights=circle.weights)
    r, th = np.hypot(xi, yi), np.arctan2(yi, xi)
    exact = r ** m * np.cos(m * th) / (2 * m)
    assert np.max(np.abs(u - exact)) < 1e-13


def test_laplace_slp_constant_density_outside(circle):
    # S[1](x) = -log|x| for |x| > 1 on the unit circle: sign of -(1/2pi) log r
    xo, yo = _pts_outside()
    u = olp.laplace_layer_apply(circle.x, circle.y, xo, yo, charge=np.ones(circle.N),
                                weights=circle.weights)
    assert np.max(np.abs(u + np.log(np.hypot(xo, yo)))) < 1e-13


def test_modhelm_circle_graf(circle):
    k = 3.7
    xi, yi = _pts_inside()
    r = np.hypot(xi, yi)
    one = np.ones(circle.N)
    us = olp.modified_helmholtz_layer_apply(circle.x, circle.y, xi, yi, k, charge=one,
                                            weights=circle.weights)
    assert rel_err(us, i0(k * r) * k0(k)) < 1e-13          # R I0(kr) K0(kR), R = 1
    ud = olp.modified_helmholtz_layer_apply(circle.x, circle.y, xi, yi, k, dipstr=one,
                                            weights=circle.weights, nx=circle.normal_x,
                                            ny=circle.normal_y)
    assert rel_err(ud, -k * i0(k * r) * k1(k)) < 1e-13     # R d/dR [I0(kr) K0(kR)]
    xo, yo = _pts_outside()
    ro = np.hypot(xo, yo)
    uo = olp.modified_helmholtz_layer_apply(circle.x, circle.y, xo, yo, k, dipstr=one,
                                            weights=circle.weights, nx=circle.normal_x,
                                            ny=circle.normal_y)
    assert rel_err(uo, k * i1(k) * k0(k * ro)) < 1e-13     # R d/dR [I0(kR) K0(kr)]


def test_stokes_identities(star):
    xi, yi = _pts_inside()
    xo, yo = _pts_outside()
    n = np.vstack([star.normal_x, star.normal_y])
    # Stokeslet with f = n: zero velocity everywhere, pressure -1 inside / 0 outside
    u, v, p = olp.stokes_layer_apply(star.x, star.y, xi, yi, force=n, weights=star.weights)
    assert max(np.max(np.abs(u)), np.max(np.abs(v))) < 1e-12
    assert np.max(np.abs(p + 1.0)) < 1e-12
    u, v, p = olp.stokes_layer_apply(star.x, star.y, xo, yo, force=n, weights=star.weights)
    assert max(np.max(np.abs(u)), np.max(np.abs(v)), np.max(np.abs(p))) < 1e-12
    # stresslet with constant density g: u = -g inside, 0 outside, zero pressure
    g = np.vstack([0.7 * np.ones(star.N), -1.3 * np.ones(star.N)])
    u, v, p = olp.stokes_layer_apply(star.x, star.y, xi, yi, dipstr=g, weights=star.weights,
                                     nx=star.normal_x, ny=star.normal_y)
    assert np.max(np.abs(u + 0.7)) < 1e-12 and np.max(np.abs(v - 1.3)) < 1e-12
    assert np.max(np.abs(p)) < 1e-11
    u, v, p = olp.stokes_layer_apply(star.x, star.y, xo, yo, dipstr=g, weights=star.weights,
                                     nx=star.normal_x, ny=star.normal_y)
    assert max(np.max(np.abs(u)), np.max(np.abs(v)), np.max(np.abs(p))) < 1e-12


def test_stokes_pressure_formulas_match_reference_restatement(star):
    """The reference restates the Stokes pressure kernels in-tree
    (ipde/solvers/internals/stokes_save.py:69-81 `eval_p1`); evaluate that formula
    directly at one point and compare."""
    rng = np.random.default_rng(5)
    slp = rng.standard_normal((2, star.N))
    dlp = rng.standard_normal((2, star.N))
    px, py = 0.13, -0.21
    dx, dy = px - star.x, py - star.y
    r2 = dx * dx + dy * dy
    ir2 = 1 / r2
    sir2 = ir2 * 0.5 / np.pi * star.weights
    p_ref = np.sum(dx * sir2 * slp[0]) + np.sum(dy * sir2 * slp[1])
    rdotnir4 = (dx * star.normal_x + dy * star.normal_y) * ir2 * ir2
    wx = (-star.normal_x * ir2 + 2 * rdotnir4 * dx) / np.pi * star.weights
    wy = (-star.normal_y * ir2 + 2 * rdotnir4 * dy) / np.pi * star.weights
    p_ref += np.sum(wx * dlp[0]) + np.sum(wy * dlp[1])
    _, _, p = olp.stokes_layer_apply(star.x, star.y, [px], [py], force=slp, dipstr=dlp,
                                     weights=star.weights, nx=star.normal_x, ny=star.normal_y)
    assert abs(p[0] - p_ref) < 1e-12 * max(1.0, abs(p_ref))


def test_c_oracle_matches_numpy_oracle(star):
    rng = np.random.default_rng(3)
    sig = rng.standard_normal(star.N)
    tau = rng.standard_normal(star.N)
    tx = rng.uniform(-1.5, 1.5, 3000)
    ty = rng.uniform(-1.5, 1.5, 3000)
    a = olp.laplace_layer_apply(star.x, star.y, tx, ty, charge=sig, dipstr=tau,
                                weights=star.weights, nx=star.normal_x, ny=star.normal_y)
    b = oracle.c_laplace_apply(star.x, star.y, tx, ty, w_sigma=sig * star.weights,
                               nx=star.normal_x, ny=star.normal_y, w_tau=tau * star.weights)
    assert rel_err(b, a) < 1e-13
    f = rng.standard_normal((2, star.N))
    g = rng.standard_normal((2, star.N))
    ua, va, pa = olp.stokes_layer_apply(star.x, star.y, tx, ty, force=f, dipstr=g,
                                        weights=star.weights, nx=star.normal_x, ny=star.normal_y)
    w = star.weights
    ub, vb, pb = oracle.c_stokes_apply(star.x, star.y, tx, ty, wfx=f[0] * w, wfy=f[1] * w,
                                       nx=star.normal_x, ny=star.normal_y, wdx=g[0] * w,
                                       wdy=g[1] * w)
    assert rel_err(ub, ua) < 1e-13 and rel_err(vb, va) < 1e-13 and rel_err(pb, pa) < 1e-13


def test_self_evaluation_skips_coincident(circle):
    sig = np.cos(circle.t)
    u = olp.laplace_layer_apply(circle.x, circle.y, circle.x, circle.y, charge=sig,
                                weights=circle.weights, skip_coincident=True)
    assert np.all(np.isfinite(u))
    b = oracle.c_laplace_apply(circle.x, circle.y, circle.x, circle.y,
                               w_sigma=sig * circle.weights, skip_coincident=True)
    assert rel_err(b, u) < 1e-13
