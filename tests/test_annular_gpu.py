"""Parity of the device annular solvers against the reference goldens (operator
level: exact to rounding) and the numpy oracle (solve level: solver tolerance)."""
import os

import numpy as np
import pytest

from oracle import annular as oa
from util import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gs():
    return np.load(os.path.join(G, "annular_scalar.npz"))


@pytest.fixture(scope="module")
def scalar_geo(gs):
    from ipde_amd.annular.annular_full import ApproximateAnnularGeometry, RealAnnularGeometry
    n, M, width, radius = gs["params"]
    aag = ApproximateAnnularGeometry(int(n), int(M), width, radius)
    return aag, RealAnnularGeometry(gs["speed"], gs["curvature"], aag)


@pytest.mark.parametrize("tag", ["mh", "po"])
def test_scalar_apply_and_preconditioner_golden(gs, scalar_geo, tag):
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde_amd.annular.poisson import AnnularPoissonSolver
    aag, rag = scalar_geo
    S = AnnularModifiedHelmholtzSolver(aag, float(gs["mh_k"][0])) if tag == "mh" \
        else AnnularPoissonSolver(aag)
    S._set_geometry(rag)
    uh = gs[tag + "_uh"]
    assert rel_err(S._apply(uh), gs[tag + "_apply"]) < 1e-12
    assert rel_err(S._optim_preconditioner(uh), gs[tag + "_prec"]) < 1e-12


@pytest.mark.parametrize("tag", ["mh", "po"])
def test_scalar_solve_golden(gs, scalar_geo, tag):
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde_amd.annular.poisson import AnnularPoissonSolver
    aag, rag = scalar_geo
    S = AnnularModifiedHelmholtzSolver(aag, float(gs["mh_k"][0])) if tag == "mh" \
        else AnnularPoissonSolver(aag)
    u = S.solve(rag, gs[tag + "_force"], gs[tag + "_ig"], gs[tag + "_og"], tol=1e-13,
                maxiter=200, restart=60)
    assert rel_err(u, gs[tag + "_sol_ref"]) < 1e-10
    assert abs(S.iterations_last_call - int(gs[tag + "_iters"][0])) <= 2
    assert S.residual_last_call <= 1e-13
    # restarts exercised: same answer with a tiny Krylov space
    u2 = S.solve(rag, gs[tag + "_force"], gs[tag + "_ig"], gs[tag + "_og"], tol=1e-12,
                 maxiter=200, restart=4)
    assert rel_err(u2, gs[tag + "_sol_ref"]) < 1e-9


def test_scalar_solver_circle_manufactured_large():
    """n = 4096 tangential points, M = 20 (the BASELINE boundary size): concentric
    annulus, u = r^3 cos(3 theta) is harmonic; spectral accuracy expected."""
    from ipde_amd.annular.annular_full import ApproximateAnnularGeometry, RealAnnularGeometry
    from ipde_amd.annular.poisson import AnnularPoissonSolver
    n, M, width, R = 4096, 20, 0.1, 1.0
    aag = ApproximateAnnularGeometry(n, M, width, R)
    rag = RealAnnularGeometry(np.full(n, R), np.full(n, 1.0 / R), aag)
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    rr = (R + aag.rv0)[:, None]
    sol = rr ** 3 * np.cos(3 * t)[None, :] + np.exp(rr * np.cos(t)[None, :]) * np.cos(rr * np.sin(t)[None, :])
    S = AnnularPoissonSolver(aag)
    ig = aag.CO.ibc_dirichlet[0] @ sol
    og = aag.CO.obc_dirichlet[0] @ sol
    u = S.solve(rag, np.zeros((M, n)), ig, og, tol=1e-13, maxiter=100, restart=50)
    assert rel_err(u, sol) < 1e-11
    assert S.iterations_last_call <= 5   # the preconditioner is exact on a true circle


@pytest.fixture(scope="module")
def gst():
    return np.load(os.path.join(G, "annular_stokes.npz"))


@pytest.fixture(scope="module")
def stokes_geo(gst):
    from ipde_amd.annular.annular import ApproximateAnnularGeometry, RealAnnularGeometry
    n, M, width, radius = gst["params"]
    aag = ApproximateAnnularGeometry(int(n), int(M), width, radius)
    return aag, RealAnnularGeometry(gst["speed"], gst["curvature"], aag)


def test_stokes_apply_and_preconditioner_golden(gst, stokes_geo):
    from ipde_amd.annular.stokes import AnnularStokesSolver
    aag, rag = stokes_geo
    S = AnnularStokesSolver(aag, 1.0)
    S._set_geometry(rag)
    for kind in ("random", "herm"):
        assert rel_err(S._apply_optim_real(gst["vec_" + kind]), gst["apply_" + kind]) < 1e-12, kind
    # the (3M-1)^2 blocks are ill-conditioned (|K^-1| ~ 1e3): two LAPACK inversions of
    # the same block (ours batched, the reference's one by one) differ by cond*eps
    assert rel_err(S._preconditioner(gst["vec_random"]), gst["prec_random"]) < 1e-10


def test_stokes_solve_golden(gst, stokes_geo):
    from ipde_amd.annular.stokes import AnnularStokesSolver
    aag, rag = stokes_geo
    S = AnnularStokesSolver(aag, 1.0)
    ur, ut, p = S.solve(rag, gst["fr"], gst["ft"], gst["irg"], gst["itg"], gst["org"], gst["otg"],
                        tol=1e-12, maxiter=300, restart=100)
    assert rel_err(ur, gst["sol_ur"]) < 1e-8 and rel_err(ut, gst["sol_ut"]) < 1e-8
    assert rel_err(p, gst["sol_p"]) < 1e-8


def test_stokes_solver_matches_oracle_mid_size():
    from ipde_amd.annular.annular import ApproximateAnnularGeometry, RealAnnularGeometry
    from ipde_amd.annular.stokes import AnnularStokesSolver
    n, M, width, R = 256, 12, 0.15, 1.0
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    a, f = 0.1, 4
    r = 1 + a * np.cos(f * t)
    rp = -a * f * np.sin(f * t)
    rpp = -a * f * f * np.cos(f * t)
    speed = np.sqrt(r * r + rp * rp)
    curv = (r * r + 2 * rp * rp - r * rpp) / speed ** 3
    aag = ApproximateAnnularGeometry(n, M, width, R)
    rag = RealAnnularGeometry(speed, curv, aag)
    oaag = oa.AAG(n, M, width, R, full=False)
    orag = oa.RAG(speed, curv, oaag)
    rng = np.random.default_rng(0)
    S = AnnularStokesSolver(aag, 1.0)
    O = oa.StokesSolver(oaag, 1.0)
    S._set_geometry(rag)
    v = rng.standard_normal(S.NB) + 1j * rng.standard_normal(S.NB)
    grouped = S._apply_optim_real(v)
    assert rel_err(grouped, O.apply(v, orag)) < 1e-12
    # the operator with one launch per term (what the grouped launches replace) and with grouped
    # launches but separate real-to-complex copies and closing launches (annular_grouped = 1; the
    # default, 2, merges those): same arithmetic in the same order, bit for bit
    assert S.ctx.get_option("annular_grouped") == 2
    for variant in (0, 1):
        S.ctx.set_option("annular_grouped", variant)
        try:
            other = S._apply_optim_real(v)
        finally:
            S.ctx.set_option("annular_grouped", 2)
        assert np.array_equal(np.asarray(grouped), np.asarray(other)), variant
    assert rel_err(S._preconditioner(v), O.precondition(v)) < 1e-11
    T = t[None, :]
    rv = aag.rv0[:, None]
    fr = np.cos(2 * T) * (1 + rv)
    ft = np.sin(3 * T) * (1 - 0.5 * rv)
    z = np.zeros(n)
    ur, ut, p = S.solve(rag, fr, ft, z, z, z, z, tol=1e-12, maxiter=300, restart=100)
    our, out, op = O.solve(orag, fr, ft, z, z, z, z, tol=1e-12, maxiter=300, restart=100)
    assert rel_err(ur, our) < 1e-8 and rel_err(ut, out) < 1e-8 and rel_err(p, op) < 1e-8


def test_gmres_argument_checks(gs, scalar_geo):
    """restart beyond the pinned Hessenberg buffer and tol <= 0 are refused (status, no
    overrun); a restart larger than maxiter is clamped to it"""
    from ipde_amd._lib import IpdeHipError
    from ipde_amd.annular.poisson import AnnularPoissonSolver
    aag, rag = scalar_geo
    S = AnnularPoissonSolver(aag)
    args = (rag, gs["po_force"], gs["po_ig"], gs["po_og"])
    with pytest.raises(IpdeHipError, match="invalid argument"):
        S.solve(*args, tol=1e-12, maxiter=5000, restart=3000)
    with pytest.raises(IpdeHipError, match="invalid argument"):
        S.solve(*args, tol=0.0, maxiter=50, restart=20)
    u = S.solve(*args, tol=1e-13, maxiter=200, restart=10 ** 6)    # clamped to maxiter = 200
    assert rel_err(u, gs["po_sol_ref"]) < 1e-10


def test_plan_creation_from_two_threads_on_private_contexts():
    """Two host threads, each with a library context of its own, create (never-seen-before)
    batched 1-D and 2-D rocFFT plans at the same time and run them: plan creation is
    serialised inside the library (g_rocfft_plan_mutex), results are those of numpy."""
    import threading
    import torch
    from ipde_amd.device import private_context
    from ipde_amd.spectral import GridPlan, fft1
    errs = []

    def work(seed, n1, n2):
        try:
            ctx = private_context()
            rng = np.random.default_rng(seed)
            a = rng.standard_normal((7, n1)) + 1j * rng.standard_normal((7, n1))
            got = fft1(a, -1, ctx=ctx)
            assert rel_err(got, np.fft.fft(a, axis=1)) < 1e-13
            f = rng.standard_normal((n2, n2 + 2))
            plan = GridPlan(n2, n2 + 2, 0.1, 0.1, ctx)
            assert rel_err(plan.fft2(f), np.fft.fft2(f)) < 1e-13
            plan.close()
        except Exception as e:       # pragma: no cover
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(1, 1234, 118)),
          threading.Thread(target=work, args=(2, 1238, 122))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_context_releases_its_solver_handles_first(gs, scalar_geo):
    """a private context that goes away before the solvers created on it frees their library
    handles itself (no leak, no use of a dead context afterwards)"""
    import gc
    from ipde_amd.device import private_context
    from ipde_amd.annular.poisson import AnnularPoissonSolver
    aag, rag = scalar_geo
    ctx = private_context()
    S = AnnularPoissonSolver(aag, ctx=ctx)
    u = S.solve(rag, gs["po_force"], gs["po_ig"], gs["po_og"], tol=1e-13, maxiter=200, restart=60)
    assert rel_err(u, gs["po_sol_ref"]) < 1e-10
    assert S in ctx._children and S.handle
    ctx.close()
    assert S.handle is None and ctx.handle is None
    del S, ctx
    gc.collect()


def test_gmres_graph_replay_is_bitwise_the_eager_solve(gs, scalar_geo, gst, stokes_geo):
    """option "gmres_graphs" (csrc/annular.hip): the inner GMRES iterations captured into hipGraphs
    at their first use and replayed afterwards — capture solve, replay solve and the default
    launch-by-launch solve give the same bits, with restarts (restart < iterations) too"""
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde_amd.annular.stokes import AnnularStokesSolver
    aag, rag = scalar_geo
    S = AnnularModifiedHelmholtzSolver(aag, float(gs["mh_k"][0]))
    args = (rag, gs["mh_force"], gs["mh_ig"], gs["mh_og"])
    for kw in (dict(tol=1e-13, maxiter=200, restart=100), dict(tol=1e-13, maxiter=200, restart=4)):
        eager = np.array(S.solve(*args, **kw))
        it = S.iterations_last_call
        S.ctx.set_option("gmres_graphs", 1)
        try:
            first = np.array(S.solve(*args, **kw))            # captures (and runs) the graphs
            again = np.array(S.solve(*args, **kw))            # replays them
        finally:
            S.ctx.set_option("gmres_graphs", 0)
        assert S.iterations_last_call == it
        assert np.array_equal(first, again) and np.array_equal(first, eager)
    aag, rag = stokes_geo
    V = AnnularStokesSolver(aag, 1.0)
    sargs = (rag, gst["fr"], gst["ft"], gst["irg"], gst["itg"], gst["org"], gst["otg"])
    kw = dict(tol=1e-12, maxiter=300, restart=100)
    c = [np.array(x) for x in V.solve(*sargs, **kw)]
    V.ctx.set_option("gmres_graphs", 1)
    try:
        a = [np.array(x) for x in V.solve(*sargs, **kw)]
        b = [np.array(x) for x in V.solve(*sargs, **kw)]
    finally:
        V.ctx.set_option("gmres_graphs", 0)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)


def test_gmres_lookahead_is_bitwise_the_wait_per_iteration_solve(gs, scalar_geo, gst, stokes_geo):
    """option "gmres_lookahead" (default on; csrc/annular.hip): inner iteration j + 1 is enqueued
    before the host has read the Hessenberg column of iteration j (two pinned areas, an event
    each; no look-ahead when the residual history predicts the last iteration).  Same bits and
    iteration counts as enqueue-wait-enqueue: full cycles, restarts shorter than the solve, a
    maxiter that stops it early (status NOCONV either way), tolerances that end on different
    iterations."""
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde_amd.annular.stokes import AnnularStokesSolver
    aag, rag = scalar_geo
    S = AnnularModifiedHelmholtzSolver(aag, float(gs["mh_k"][0]))
    assert S.ctx.get_option("gmres_lookahead") == 1
    args = (rag, gs["mh_force"], gs["mh_ig"], gs["mh_og"])

    def both(solver, call):
        on = call()
        it_on = solver.iterations_last_call
        solver.ctx.set_option("gmres_lookahead", 0)
        try:
            off = call()
        finally:
            solver.ctx.set_option("gmres_lookahead", 1)
        assert solver.iterations_last_call == it_on
        return on, off

    for kw in (dict(tol=1e-13, maxiter=200, restart=100), dict(tol=1e-13, maxiter=200, restart=4),
               dict(tol=1e-13, maxiter=200, restart=3), dict(tol=1e-6, maxiter=200, restart=100),
               dict(tol=1e-3, maxiter=200, restart=100), dict(tol=1e-9, maxiter=200, restart=2)):
        on, off = both(S, lambda: np.array(S.solve(*args, **kw)))
        assert np.array_equal(on, off)
    # maxiter reached (the solvers return the unconverged iterate): same iterate after the same 3 iterations
    on, off = both(S, lambda: np.array(S.solve(*args, tol=1e-13, maxiter=3, restart=100)))
    assert np.array_equal(on, off) and S.iterations_last_call == 3
    aag, rag = stokes_geo
    V = AnnularStokesSolver(aag, 1.0)
    sargs = (rag, gst["fr"], gst["ft"], gst["irg"], gst["itg"], gst["org"], gst["otg"])
    for kw in (dict(tol=1e-12, maxiter=300, restart=100), dict(tol=1e-12, maxiter=300, restart=5)):
        on, off = both(V, lambda: [np.array(x) for x in V.solve(*sargs, **kw)])
        for x, y in zip(on, off):
            assert np.array_equal(x, y)


def test_gmres_normalisation_inside_the_preconditioner_kernel_is_bitwise(gs, scalar_geo, gst, stokes_geo):
    """option "gmres_fused_scale" (default on): v_j = w / ||w|| is formed inside the preconditioner's
    kernel of iteration j instead of by a launch of its own at the end of iteration j - 1 — same
    bits, with and without the look-ahead, across restarts"""
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde_amd.annular.stokes import AnnularStokesSolver
    aag, rag = scalar_geo
    S = AnnularModifiedHelmholtzSolver(aag, float(gs["mh_k"][0]))
    assert S.ctx.get_option("gmres_fused_scale") == 1
    args = (rag, gs["mh_force"], gs["mh_ig"], gs["mh_og"])
    aag2, rag2 = stokes_geo
    V = AnnularStokesSolver(aag2, 1.0)
    sargs = (rag2, gst["fr"], gst["ft"], gst["irg"], gst["itg"], gst["org"], gst["otg"])
    for solver, call, kws in (
            (S, lambda kw: [np.array(S.solve(*args, **kw))],
             (dict(tol=1e-13, maxiter=200, restart=100), dict(tol=1e-13, maxiter=200, restart=3),
              dict(tol=1e-5, maxiter=200, restart=100), dict(tol=1e-13, maxiter=4, restart=100))),
            (V, lambda kw: [np.array(x) for x in V.solve(*sargs, **kw)],
             (dict(tol=1e-12, maxiter=300, restart=100), dict(tol=1e-12, maxiter=300, restart=5)))):
        for kw in kws:
            ref = call(kw)
            it = solver.iterations_last_call
            for look in (1, 0):
                solver.ctx.set_option("gmres_fused_scale", 0)
                solver.ctx.set_option("gmres_lookahead", look)
                try:
                    got = call(kw)
                finally:
                    solver.ctx.set_option("gmres_fused_scale", 1)
                    solver.ctx.set_option("gmres_lookahead", 1)
                assert solver.iterations_last_call == it
                for x, y in zip(ref, got):
                    assert np.array_equal(x, y)


def test_helper_jump_kernels_match_the_numpy_statements():
    """ipde_scalar_interface_jumps / ipde_stokes_rotate / ipde_stokes_interface_jumps (csrc/annular.hip)
    against the numpy statements of the helpers they replace (reference internals/scalar.py:76-90,
    internals/vector.py:65-144), on random data"""
    import ctypes
    import torch
    from ipde_amd import _lib
    from ipde_amd.device import get_context, ptr
    ctx = get_context()
    rng = np.random.default_rng(21)
    M, N = 14, 600
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    new = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda")
    # (device copies are held in names: a temporary passed as ptr(up(x)) would be freed, and its
    # memory handed to the next one, before the kernel runs)
    # scalar
    ur, est, nrm, bd = rng.standard_normal((M, N)), rng.standard_normal(M), rng.standard_normal((2, N)), \
        rng.standard_normal((3, N))
    d_ur, d_est, d_nrm, d_bd = up(ur), up(est), up(nrm), up(bd)
    slp, dlp = new(N), new(N)
    for sign in (1.0, -1.0):
        ctx.check(ctx.lib.ipde_scalar_interface_jumps(ctx.handle, M, N, ptr(d_ur), ptr(d_est), ptr(d_nrm),
                                                      ptr(d_bd), sign, ptr(slp), ptr(dlp)))
        ref = sign * (est @ ur - (bd[1] * nrm[0] + bd[2] * nrm[1]))
        assert np.abs(slp.cpu().numpy() - ref).max() < 1e-13 * np.abs(ref).max()
        assert np.array_equal(dlp.cpu().numpy(), sign * bd[0])
    # Stokes: rotation there and back, host and device inputs
    th = rng.uniform(0, 2 * np.pi, N)
    geom = np.stack([np.cos(th), np.sin(th), -np.sin(th), np.cos(th), np.cos(th + 0.1), np.sin(th + 0.1)])
    d_geom = up(geom)
    fu, fv = rng.standard_normal((M, N)), rng.standard_normal((M, N))
    fr, ft, bu, bv = new(M, N), new(M, N), new(M, N), new(M, N)
    ctx.check(ctx.lib.ipde_stokes_rotate(ctx.handle, _lib.IPDE_HOST, M, N, ptr(fu), ptr(fv), ptr(d_geom), 1,
                                         ptr(fr), ptr(ft)))
    assert np.allclose(fr.cpu().numpy(), fu * geom[0] + fv * geom[1], rtol=0, atol=1e-15)
    assert np.allclose(ft.cpu().numpy(), fu * geom[2] + fv * geom[3], rtol=0, atol=1e-15)
    ctx.check(ctx.lib.ipde_stokes_rotate(ctx.handle, _lib.IPDE_DEVICE, M, N, ptr(fr), ptr(ft), ptr(d_geom), 0,
                                         ptr(bu), ptr(bv)))
    assert np.abs(bu.cpu().numpy() - fu).max() < 1e-14 and np.abs(bv.cpu().numpy() - fv).max() < 1e-14
    # Stokes: tractions and jumps
    rr, tr, pr = (rng.standard_normal((M, N)) for _ in range(3))
    rs = 1.0 + 0.3 * rng.uniform(size=(M, N))
    irs = 1.0 / rs
    D00, rk, bdata = rng.standard_normal((M, M)), np.fft.fftfreq(N, 1.0 / N) * 1.3, rng.standard_normal((5, N))
    tder = lambda f: np.fft.ifft(np.fft.fft(f) * 1j * rk).real
    Urr, Urt, Utr = D00 @ rr, tder(rr) * irs, rs * (D00 @ (tr * irs))
    Tr, Tt = 2 * est @ Urr - est @ pr, est @ Utr + est @ Urt
    rtx, rty = Tr * geom[0] + Tt * geom[2], Tr * geom[1] + Tt * geom[3]
    gx, gy = bdata[2] * geom[4] + bdata[3] * geom[5], bdata[3] * geom[4] + bdata[4] * geom[5]
    o_ur, o_vr, taus, taud = new(M, N), new(M, N), new(2 * N), new(2 * N)
    dev = [up(a) for a in (rr, tr, pr, geom, rs, irs, D00, est, rk, bdata)]
    ctx.check(ctx.lib.ipde_stokes_interface_jumps(ctx.handle, M, N, *[ptr(a) for a in dev], -1.0, ptr(o_ur),
                                                  ptr(o_vr), ptr(taus), ptr(taud)))
    ref_s = -np.concatenate([rtx - gx, rty - gy])
    assert np.abs(taus.cpu().numpy() - ref_s).max() < 1e-12 * np.abs(ref_s).max()
    assert np.array_equal(taud.cpu().numpy(), -np.concatenate([bdata[0], bdata[1]]))
    assert np.allclose(o_ur.cpu().numpy(), rr * geom[0] + tr * geom[2], rtol=0, atol=1e-15)
    assert np.allclose(o_vr.cpu().numpy(), rr * geom[1] + tr * geom[3], rtol=0, atol=1e-15)
    # argument checks
    assert ctx.lib.ipde_scalar_interface_jumps(ctx.handle, 0, N, ptr(slp), ptr(slp), ptr(slp), ptr(slp), 1.0,
                                               ptr(slp), ptr(dlp)) == 1
    assert ctx.lib.ipde_scalar_interface_jumps(ctx.handle, M, N, ptr(slp), ptr(slp), ptr(slp), ptr(slp), 0.5,
                                               ptr(slp), ptr(dlp)) == 1
    assert ctx.lib.ipde_stokes_rotate(ctx.handle, 7, M, N, ptr(fr), ptr(ft), ptr(fr), 1, ptr(fr), ptr(ft)) == 1
    assert ctx.lib.ipde_stokes_rotate(ctx.handle, _lib.IPDE_DEVICE, M, N, None, ptr(ft), ptr(fr), 1, ptr(fr),
                                      ptr(ft)) == 1
    ctx.sync()


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192])
def test_fused_transform_pairs_equal_the_rocfft_stages(n):
    """option "annular_fused_fft" (default on): for power-of-two n <= 8192 (8192 — BASELINE configs[3]'s boundary
    — as two 4096-point halves and a radix-2 level) a stage of the scalar
    operator — inverse FFT, metric field, forward FFT — is one kernel on the fft_core.h transforms
    instead of two rocFFT calls around a pointwise kernel: the same operator to rounding, the same
    solution, on a non-circular annulus"""
    from ipde_amd.annular.annular_full import ApproximateAnnularGeometry, RealAnnularGeometry
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    M, width, R = 16, 0.12, 1.0
    aag = ApproximateAnnularGeometry(n, M, width, R)
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    speed = R * (1.0 + 0.2 * np.cos(3 * t))
    curv = (1.0 / R) * (1.0 + 0.3 * np.sin(2 * t))
    rag = RealAnnularGeometry(speed, curv, aag)
    S = AnnularModifiedHelmholtzSolver(aag, 2.0)
    S._set_geometry(rag)
    rng = np.random.default_rng(n)
    uh = rng.standard_normal(M * n) + 1j * rng.standard_normal(M * n)
    assert S.ctx.get_option("annular_fused_fft") == 1
    a1 = np.asarray(S._apply(uh))
    f = rng.standard_normal((M, n))
    ig, og = rng.standard_normal(n), rng.standard_normal(n)
    u1 = np.array(S.solve(rag, f, ig, og, tol=1e-12, maxiter=200, restart=60))
    it1 = S.iterations_last_call
    S.ctx.set_option("annular_fused_fft", 0)
    try:
        a0 = np.asarray(S._apply(uh))
        u0 = np.array(S.solve(rag, f, ig, og, tol=1e-12, maxiter=200, restart=60))
        it0 = S.iterations_last_call
    finally:
        S.ctx.set_option("annular_fused_fft", 1)
    assert np.abs(a1 - a0).max() < 1e-13 * np.abs(a0).max()
    assert abs(it1 - it0) <= 1
    assert np.abs(u1 - u0).max() < 1e-9 * np.abs(u0).max()


@pytest.mark.parametrize("n,M", [(512, 8), (1024, 12), (2048, 16), (4096, 20), (4096, 31)])
def test_device_side_gmres_cycle_against_the_launch_per_stage_cycle(n, M):
    """option "gmres_persistent" (csrc/annular_gmres_persist.h; off by default: measured 8-14 % slower
    than the launch-per-stage cycle, profiles/r03_gmres_persistent_ab.txt): the first GMRES cycle of
    the scalar annular solve in ONE launch, Arnoldi / Givens bookkeeping on the device.  Same operator
    and preconditioner bits; the inner products are summed in another order, so: same iteration counts
    (to one), solutions equal to 1e-12 of max|u|, residuals under the tolerance — for a cycle that
    converges, one that runs out (restart shorter than the solve: the launch-per-stage cycles take
    over from its estimate), a maxiter that stops the solve early, and a zero right-hand side."""
    from ipde_amd.annular.annular_full import ApproximateAnnularGeometry, RealAnnularGeometry
    from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde_amd.annular.poisson import AnnularPoissonSolver
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    rad = 1.0 + 0.2 * np.cos(5 * t)
    drad, d2rad = -np.sin(5 * t), -5.0 * np.cos(5 * t)
    speed = np.sqrt(rad ** 2 + drad ** 2)
    curv = (rad ** 2 + 2 * drad ** 2 - rad * d2rad) / speed ** 3
    width = M * (2 * np.pi / n) * speed.min()
    aag = ApproximateAnnularGeometry(n, M, width, 1.0)
    rag = RealAnnularGeometry(speed, curv, aag)
    rng = np.random.default_rng(n + M)
    f = rng.standard_normal((M, n))
    ig, og = rng.standard_normal(n), rng.standard_normal(n)
    for S in (AnnularPoissonSolver(aag), AnnularModifiedHelmholtzSolver(aag, 7.0)):
        found = S.ctx.get_option("gmres_persistent")

        def both(**kw):
            S.ctx.set_option("gmres_persistent", 1)
            try:
                on = np.array(S.solve(rag, f, ig, og, **kw))
                it_on, r_on = S.iterations_last_call, S.residual_last_call
                S.ctx.set_option("gmres_persistent", 0)
                off = np.array(S.solve(rag, f, ig, og, **kw))
            finally:
                S.ctx.set_option("gmres_persistent", found)
            return on, off, it_on, S.iterations_last_call, r_on

        on, off, it_on, it_off, r_on = both(tol=1e-12, maxiter=100, restart=30)
        assert it_on >= 4 and abs(it_on - it_off) <= 1 and r_on <= 1e-12
        assert rel_err(on, off) < 1e-11
        full = it_on
        # the first cycle runs out (a restart a little over half the solve): launch-per-stage cycles take over
        on, off, it_on, it_off, r_on = both(tol=1e-12, maxiter=200, restart=full // 2 + 2)
        assert it_on > full // 2 + 2 and abs(it_on - it_off) <= 2 and r_on <= 1e-12 and rel_err(on, off) < 1e-10
        on, off, it_on, it_off, _ = both(tol=1e-13, maxiter=3, restart=30)         # stopped early
        assert it_on == it_off == 3 and rel_err(on, off) < 1e-7    # (an unconverged iterate of an ill-conditioned system: the last bits of h are amplified, 1.6e-9 at n = 4096)
        on, off, it_on, it_off, _ = both(tol=1e-6, maxiter=100, restart=30)
        assert abs(it_on - it_off) <= 1 and rel_err(on, off) < 1e-5
        S.ctx.set_option("gmres_persistent", 1)
        try:
            z = np.array(S.solve(rag, np.zeros((M, n)), np.zeros(n), np.zeros(n), tol=1e-12, maxiter=50, restart=20))
            assert S.iterations_last_call == 0 and not np.any(z)
            # repeated solves are bitwise reproducible (partials are summed in a fixed order)
            a = np.array(S.solve(rag, f, ig, og, tol=1e-12, maxiter=100, restart=30))
            b = np.array(S.solve(rag, f, ig, og, tol=1e-12, maxiter=100, restart=30))
            assert np.array_equal(a, b)
        finally:
            S.ctx.set_option("gmres_persistent", found)
