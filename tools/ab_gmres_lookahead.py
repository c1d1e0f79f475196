#!/usr/bin/env python3
"""A/B of the annular GMRES with and without the one-iteration look-ahead (option
"gmres_lookahead") and the normalisation folded into the preconditioner kernel ("gmres_fused_scale"): single-body annular Poisson (n = 4096) and Stokes (n = 4096 and 3200) solves,
median wall time of 30 solves each, the two settings alternating."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import Curve
from ipde_amd.annular.annular_full import ApproximateAnnularGeometry as AAGf, RealAnnularGeometry
from ipde_amd.annular.annular import ApproximateAnnularGeometry as AAGd
from ipde_amd.annular.poisson import AnnularPoissonSolver
from ipde_amd.annular.stokes import AnnularStokesSolver


def geometry(nb, M, full):
    c = Curve(nb, a=0.2, f=5)
    tt = c.t
    r, rp, rpp = 1 + 0.2 * np.cos(5 * tt), -1.0 * np.sin(5 * tt), -5.0 * np.cos(5 * tt)
    curv = (r * r + 2 * rp * rp - r * rpp) / c.speed ** 3
    aag = (AAGf if full else AAGd)(nb, M, M * c.dt * c.speed.min(), 1.0)
    return c, aag, RealAnnularGeometry(c.speed, curv, aag)


def ab(name, solver, call):
    res = {}
    call()
    settings = ((1, 1), (1, 0), (0, 1), (0, 0))      # (gmres_lookahead, gmres_fused_scale)
    for rep in range(30):
        for look, fused in settings:
            solver.ctx.set_option("gmres_lookahead", look)
            solver.ctx.set_option("gmres_fused_scale", fused)
            t0 = time.perf_counter(); call(); res.setdefault((look, fused), []).append((time.perf_counter() - t0) * 1e3)
    solver.ctx.set_option("gmres_lookahead", 1)
    solver.ctx.set_option("gmres_fused_scale", 1)
    print(f"{name}: {solver.iterations_last_call} iterations; " + ", ".join(
        f"look-ahead {l} fused scale {f}: {np.median(res[(l, f)]):.3f} ms" for l, f in settings), flush=True)


c, aag, rag = geometry(4096, 20, True)
S = AnnularPoissonSolver(aag)
fr = np.cos(3 * c.t)[None, :] * (1 + aag.rv0[:, None])
ab("annular Poisson n=4096 M=20", S, lambda: S.solve(rag, fr, 0.0, 0.0, tol=1e-12, maxiter=100, restart=50))
for nb, M in ((4096, 20), (3200, 14)):
    c, aag, rag = geometry(nb, M, False)
    V = AnnularStokesSolver(aag, 1.0)
    fr = np.cos(3 * c.t)[None, :] * (1 + aag.rv0[:, None])
    ft = np.sin(2 * c.t)[None, :] * (1 - aag.rv0[:, None])
    z = np.zeros(nb)
    ab(f"annular Stokes n={nb} M={M}", V, lambda: V.solve(rag, fr, ft, z, z, z, z, tol=1e-10, maxiter=200, restart=100))
