#!/usr/bin/env python3
"""A/B of the scalar annular GMRES with the device-side first cycle (option "gmres_persistent") and the
launch-per-stage cycles: annular Poisson / modified Helmholtz solves at n = 1024 .. 4096, median wall
time of 30 solves each, the two settings alternating; then the warm 2048^2 interior Poisson solve."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
from util import Curve
from ipde_amd.annular.annular_full import ApproximateAnnularGeometry as AAGf, RealAnnularGeometry
from ipde_amd.annular.poisson import AnnularPoissonSolver
from ipde_amd.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver


def geometry(nb, M):
    c = Curve(nb, a=0.2, f=5)
    tt = c.t
    r, rp, rpp = 1 + 0.2 * np.cos(5 * tt), -1.0 * np.sin(5 * tt), -5.0 * np.cos(5 * tt)
    curv = (r * r + 2 * rp * rp - r * rpp) / c.speed ** 3
    aag = AAGf(nb, M, M * c.dt * c.speed.min(), 1.0)
    return c, aag, RealAnnularGeometry(c.speed, curv, aag)


def ab(name, ctxs, call, its, reps=30):
    res = {0: [], 1: []}
    call()
    for rep in range(reps):
        for on in (1, 0):
            for c in ctxs:
                c.set_option("gmres_persistent", on)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); call(); torch.cuda.synchronize()
            res[on].append((time.perf_counter() - t0) * 1e3)
    for c in ctxs:
        c.set_option("gmres_persistent", 0)
    print(f"{name}: {its()} iterations; one launch per cycle {np.median(res[1]):.3f} ms, "
          f"launch per stage {np.median(res[0]):.3f} ms", flush=True)


for nb, M in ((1024, 12), (2048, 16), (4096, 20)):
    c, aag, rag = geometry(nb, M)
    fr = np.cos(3 * c.t)[None, :] * (1 + aag.rv0[:, None])
    for S, nm in ((AnnularPoissonSolver(aag), "Poisson"), (AnnularModifiedHelmholtzSolver(aag, 10.0), "mod. Helmholtz k=10")):
        ab(f"annular {nm} n={nb} M={M}", [S.ctx], lambda: S.solve(rag, fr, 0.0, 0.0, tol=1e-12, maxiter=100, restart=20),
           lambda: S.iterations_last_call)

import interior_poisson
from ipde_amd.embedded_function import EmbeddedFunction
err, scale, solver, ue, T = interior_poisson.run(nb=4096, M=20, Ns=[2048, 2048], solver_tol=1e-12)
f = EmbeddedFunction(solver.ebdyc)
f.define_via_function(lambda x, y: (2.0 * np.cos(x) + 3.0 * np.cos(x) * np.sin(x) - np.cos(x) ** 3) * np.exp(np.sin(x)) * np.sin(y))
ctxs = list({id(h.annular_solver.ctx): h.annular_solver.ctx for h in solver.helpers}.values())
ab("interior Poisson 2048^2, 4096 nodes, warm solve", ctxs, lambda: solver(f, tol=1e-12, maxiter=100, restart=20),
   lambda: solver.iteration_counts, reps=20)
