"""Pin the layer-potential oracle (and, on the GPU box, the HIP kernels through the C ABI)
to numbers computed by the reference's own code: tests/golden/layer_kernels.npz, made by
tests/golden/make_golden.py::golden_layer_kernels from
  ipde/solvers/multi_boundary/poisson.py:10-17         Laplace_Eval
  ipde/grid_evaluators/laplace_grid_evaluator.py:8-12  gf
  ipde/grid_evaluators/modified_helmholtz_grid_evaluator.py:8-9  gf
  ipde/solvers/internals/stokes_save.py:29-81          PSLP / PDLP pressure rows, eval_p1
What this does NOT pin (pybie2d / pyfmmlib2d arithmetic, absent): the Laplace and
modified-Helmholtz double layers and the Stokes velocities; those stay on the analytic
identities of tests/test_oracle_layer_kat.py."""
import os

import numpy as np
import pytest

import oracle
from oracle import layer_potentials as olp

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "layer_kernels.npz"))
TOL = 1e-12
TOL_STOKES = 1e-10


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def test_fixture_is_self_consistent():
    """the reference's two statements of the Laplace kernel and of the stokeslet pressure"""
    assert rel(G["laplace_slp_eval"], G["laplace_slp_gf"]) < 1e-14
    assert rel(G["stokes_p_slp_row"], G["stokes_p_slp_eval_p1"]) < 1e-14
    assert rel(G["stokes_p_slp_row"] + G["stokes_p_dlp_row"], G["stokes_p_both_eval_p1"]) < 1e-13


def test_numpy_oracle_against_reference_numbers():
    g = G
    u = olp.laplace_layer_apply(g["sx"], g["sy"], g["tx"], g["ty"], charge=g["sigma"], weights=g["w"])
    assert rel(u, g["laplace_slp_eval"]) < 1e-14 and rel(u, g["laplace_slp_gf"]) < 1e-14
    for j, k in enumerate(g["modhelm_k"]):
        u = olp.modified_helmholtz_layer_apply(g["sx"], g["sy"], g["tx"], g["ty"], float(k),
                                               charge=g["sigma"], weights=g["w"])
        assert rel(u, g["modhelm_slp_gf_%d" % j]) < 1e-14, k
    _, _, p = olp.stokes_layer_apply(g["sx"], g["sy"], g["tx"], g["ty"], force=g["force"],
                                     weights=g["w"])
    assert rel(p, g["stokes_p_slp_row"]) < 1e-13
    _, _, p = olp.stokes_layer_apply(g["sx"], g["sy"], g["tx"], g["ty"], dipstr=g["dipstr"],
                                     weights=g["w"], nx=g["nx"], ny=g["ny"])
    assert rel(p, g["stokes_p_dlp_row"]) < 1e-13
    _, _, p = olp.stokes_layer_apply(g["sx"], g["sy"], g["tx"], g["ty"], force=g["force"],
                                     dipstr=g["dipstr"], weights=g["w"], nx=g["nx"], ny=g["ny"])
    assert rel(p, g["stokes_p_both_eval_p1"]) < 1e-13


def test_c_oracle_against_reference_numbers():
    g = G
    u = oracle.c_laplace_apply(g["sx"], g["sy"], g["tx"], g["ty"], w_sigma=g["sigma"] * g["w"])
    assert rel(u, g["laplace_slp_eval"]) < 1e-14
    f, d, w = g["force"], g["dipstr"], g["w"]
    _, _, p = oracle.c_stokes_apply(g["sx"], g["sy"], g["tx"], g["ty"], wfx=f[0] * w, wfy=f[1] * w)
    assert rel(p, g["stokes_p_slp_row"]) < 1e-13
    _, _, p = oracle.c_stokes_apply(g["sx"], g["sy"], g["tx"], g["ty"], nx=g["nx"], ny=g["ny"],
                                    wdx=d[0] * w, wdy=d[1] * w)
    assert rel(p, g["stokes_p_dlp_row"]) < 1e-13


# ---- the HIP kernels against the same numbers (GPU box) --------------------------------

@pytest.fixture(scope="module")
def lp():
    from ipde_amd import layer_potentials
    return layer_potentials


@pytest.mark.gpu
def test_hip_laplace_patch_kernel_against_reference_numbers():
    """ipde_laplace_apply_patches on the fixture's 160 (scattered) targets: patch p holds
    xs = tx[4p : 4p+4], ys = ty[4p : 4p+4], so its DIAGONAL points are targets 4p .. 4p+3 and the
    twelve others are computed and not stored (pout = -1) — the patch kernel's arithmetic against
    the numbers of the reference's own Laplace_Eval / gf."""
    import torch
    from ipde_amd import target_plan
    from ipde_amd.device import to_device
    g = G
    n = g["tx"].shape[0]
    assert n % 4 == 0
    npatch = n // 4
    pxy = np.concatenate([g["tx"].reshape(npatch, 4).T, g["ty"].reshape(npatch, 4).T], axis=0)
    pout = np.full((16, npatch), -1, dtype=np.int32)
    for a in range(4):
        pout[5 * a] = 4 * np.arange(npatch) + a
    dev = to_device(g["tx"]).device
    plan = target_plan.TargetPlan(n, torch.as_tensor(np.ascontiguousarray(pxy), device=dev),
                                  torch.as_tensor(pout, device=dev), torch.empty(0, dtype=torch.int64, device=dev),
                                  torch.empty(0, dtype=torch.float64, device=dev),
                                  torch.empty(0, dtype=torch.float64, device=dev))
    u = target_plan.laplace_apply(plan, g["sx"], g["sy"], w_sigma=g["sigma"] * g["w"]).cpu().numpy()
    assert rel(u, g["laplace_slp_eval"]) < TOL and rel(u, g["laplace_slp_gf"]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
def test_hip_laplace_slp_against_reference_numbers(lp, generic):
    g = G
    u = lp.laplace_apply(g["sx"], g["sy"], g["tx"], g["ty"], w_sigma=g["sigma"] * g["w"],
                         generic_math=generic)
    assert rel(u, g["laplace_slp_eval"]) < TOL and rel(u, g["laplace_slp_gf"]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 9])
def test_hip_laplace_slp_kernel_variants_against_reference_numbers(lp, ctx, variant):
    """strided table kernel (1) and the default row-run kernel (9), on the fixture's list
    tiled to a size that takes the large-list path"""
    g = G
    rep = 64
    tx, ty = np.tile(g["tx"], rep), np.tile(g["ty"], rep)
    old = ctx.get_option("laplace_variant")
    ctx.set_option("laplace_variant", variant)
    try:
        u = lp.laplace_apply(g["sx"], g["sy"], tx, ty, w_sigma=g["sigma"] * g["w"])
    finally:
        ctx.set_option("laplace_variant", old)
    assert rel(u, np.tile(g["laplace_slp_eval"], rep)) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
def test_hip_modhelm_slp_against_reference_numbers(lp, generic):
    g = G
    for j, k in enumerate(g["modhelm_k"]):
        u = lp.modified_helmholtz_apply(g["sx"], g["sy"], g["tx"], g["ty"], float(k),
                                        w_sigma=g["sigma"] * g["w"], generic_math=generic)
        assert rel(u, g["modhelm_slp_gf_%d" % j]) < TOL, k


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
def test_hip_stokes_pressure_against_reference_numbers(lp, ctx, generic):
    g = G
    f, d, w = g["force"], g["dipstr"], g["w"]
    for variant in (0, 1):
        old = ctx.get_option("stokes_variant")
        ctx.set_option("stokes_variant", variant)
        try:
            _, _, p = lp.stokes_apply(g["sx"], g["sy"], g["tx"], g["ty"], wfx=f[0] * w, wfy=f[1] * w,
                                      generic_math=generic)
            assert rel(p, g["stokes_p_slp_row"]) < TOL_STOKES
            _, _, p = lp.stokes_apply(g["sx"], g["sy"], g["tx"], g["ty"], nx=g["nx"], ny=g["ny"],
                                      wdx=d[0] * w, wdy=d[1] * w, generic_math=generic)
            assert rel(p, g["stokes_p_dlp_row"]) < TOL_STOKES
            _, _, p = lp.stokes_apply(g["sx"], g["sy"], g["tx"], g["ty"], wfx=f[0] * w, wfy=f[1] * w,
                                      nx=g["nx"], ny=g["ny"], wdx=d[0] * w, wdy=d[1] * w,
                                      generic_math=generic)
            assert rel(p, g["stokes_p_both_eval_p1"]) < TOL_STOKES
        finally:
            ctx.set_option("stokes_variant", old)
