"""The Ewald-type grid evaluator on CPU: the oracle's kernel functions against the
reference's (golden vectors generated from ipde/grid_evaluators/*.py), the oracle's
split against the dense oracle sum, and the host-side tables of the product module."""
import os

import numpy as np
import pytest

from oracle import ewald as ew
from oracle import layer_potentials as olp

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_kernel_functions_match_reference_golden():
    g = np.load(os.path.join(G, "grid_evaluator_kernels.npz"))
    r, kx, ky = g["r"], g["kx"], g["ky"]
    kk = np.hypot(kx, ky)
    assert np.allclose(ew.laplace_gf(r), g["laplace_gf"], rtol=1e-15, atol=0)
    assert np.array_equal(ew.laplace_fs(kx, ky), g["laplace_fs"])
    assert np.allclose(ew.laplace_ifs(kx, ky), g["laplace_ifs"], rtol=1e-15, atol=0)
    for i in range(3):
        L = float(g["L_%d" % i])
        ref = g["laplace_tsgf_%d" % i]
        assert np.allclose(ew.laplace_trunc_sgf(kk, L), ref, rtol=1e-14, atol=1e-15 * np.abs(ref).max())
        sc = g["laplace_tsgf_scalar_%d" % i]
        assert np.allclose(ew.laplace_trunc_sgf(np.array([0.0, 2.5]), L), sc, rtol=1e-14)
    for j in range(2):
        hk = float(g["hk_%d" % j])
        assert np.allclose(ew.modhelm_gf(r, hk), g["modhelm_gf_%d" % j], rtol=1e-15, atol=0)
        assert np.array_equal(ew.modhelm_fs(kx, ky, hk), g["modhelm_fs_%d" % j])
        for i in range(3):
            L = float(g["L_%d" % i])
            ref = g["modhelm_tsgf_%d_%d" % (j, i)]
            assert np.allclose(ew.modhelm_trunc_sgf(kk, L, hk), ref, rtol=1e-14,
                               atol=1e-15 * np.abs(ref).max())


def _sources(ns, seed=0):
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, 2 * np.pi, ns)
    rad = 0.9 * (1 + 0.2 * np.cos(5 * th))
    return rad * np.cos(th), rad * np.sin(th), rng.standard_normal(ns)


@pytest.mark.parametrize("hk", [None, 5.0])
def test_oracle_split_equals_dense_sum(hk):
    n = 64
    xv = -1.5 + 3.0 / n * np.arange(n)
    sx, sy, q = _sources(30)
    X, Y = np.meshgrid(xv, xv, indexing='ij')
    u = ew.freespace_eval(sx, sy, q, xv, xv, 24, helmholtz_k=hk)
    if hk is None:
        ref = olp.laplace_layer_apply(sx, sy, X.ravel(), Y.ravel(), charge=q)
    else:
        ref = olp.modified_helmholtz_layer_apply(sx, sy, X.ravel(), Y.ravel(), hk, charge=q)
    assert np.abs(u.ravel() - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())


def test_oracle_periodic_equals_image_sum():
    n, hk = 64, 6.0
    xv = -1.5 + 3.0 / n * np.arange(n)
    sx, sy, q = _sources(12, seed=3)
    X, Y = np.meshgrid(xv, xv, indexing='ij')
    u = ew.periodic_eval(sx, sy, q, xv, xv, 24, helmholtz_k=hk)
    ref = 0.0
    for a in range(-3, 4):
        for b in range(-3, 4):
            ref = ref + olp.modified_helmholtz_layer_apply(sx + 3.0 * a, sy + 3.0 * b, X.ravel(),
                                                           Y.ravel(), hk, charge=q)
    assert np.abs(u.ravel() - ref).max() < 1e-13


def test_product_tables_and_fft_size():
    from ipde_amd.grid_evaluators.ewald import KaiserBesselStep, fast_fft_size, NI, DEG
    m, mo = KaiserBesselStep(38.4), ew.KaiserBesselStep(38.4)
    R = 0.07
    t = m.tables(R)
    assert t.shape == (3, NI, DEG + 1)
    rng = np.random.default_rng(0)
    r = rng.uniform(0, R, 4000)
    x = 1 - 2 * r / R
    fi = (x + 1) / 2 * NI
    i = np.minimum(fi.astype(int), NI - 1)
    tt = 2 * (fi - i) - 1
    ex = mo.chi(r, R)
    for f in range(3):
        got = np.array([np.polyval(t[f, a][::-1], b) for a, b in zip(i, tt)])
        assert np.abs(got - ex[f]).max() < 2e-14 * np.abs(ex[f]).max()
    for n in (4192, 8288, 100, 2):
        mm = fast_fft_size(n)
        assert mm >= n and mm % 2 == 0
        for p in (2, 3, 5, 7):
            while mm % p == 0:
                mm //= p
        assert mm == 1


def test_device_side_bessel_j0_j1_and_truncated_symbols():
    """grid_evaluators/ewald.py: J0 / J1 by degree-10 Chebyshev pieces fitted from scipy's values (the
    evaluator's set-up evaluates them at 1.8e7 .. 7e7 wavenumbers: on the device since round 3; here the
    torch form, the HIP kernel's checker in tests/test_ewald_gpu.py) against
    scipy itself (whose own error grows to ~8e-15 at x ~ 2e4), and the truncated spectral Green's
    functions built on them against the reference's formulas (laplace_grid_evaluator.py:21-33,
    modified_helmholtz_grid_evaluator.py:14-17) evaluated with scipy."""
    import torch
    from scipy.special import j0, j1, k0, k1
    from ipde_amd.grid_evaluators.ewald import bessel_j01, _trunc_sgf_quadrant_torch as _trunc_sgf_quadrant
    rng = np.random.default_rng(0)
    for lo, hi, tol in ((0.0, 100.0, 5e-15), (100.0, 3.0e4, 3e-14)):
        x = np.concatenate([rng.uniform(lo, hi, 20000), [lo, hi - 1e-9]])
        J0, J1 = bessel_j01(torch.as_tensor(x))
        assert np.abs(J0.numpy() - j0(x)).max() < tol and np.abs(J1.numpy() - j1(x)).max() < tol
    kq = np.abs(np.fft.fftfreq(400, 0.01 / (2 * np.pi))[:201])
    L = 3.7
    kk = np.hypot(kq[:, None], kq[None, :])
    ks = np.where(kk == 0, 1.0, kk)
    ref = (1.0 - j0(L * kk)) / ks ** 2 - L * np.log(L) * j1(L * kk) / ks
    ref[kk == 0] = -L ** 2 * np.log(L) + L ** 2 * (1 + 2 * np.log(L)) / 4
    got = _trunc_sgf_quadrant(kq, kq[:150], L, None, 'cpu').numpy()
    assert got.shape == (201, 150) and np.abs(got - ref[:, :150]).max() < 1e-14 * np.abs(ref).max()
    kap = 10.0
    ref = (1.0 + L * kk * j1(L * kk) * k0(L * kap) - L * kap * j0(L * kk) * k1(L * kap)) / (kk ** 2 + kap ** 2)
    got = _trunc_sgf_quadrant(kq, kq, L, kap, 'cpu').numpy()
    assert np.abs(got - ref).max() < 1e-14 * np.abs(ref).max()
