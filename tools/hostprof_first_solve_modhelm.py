#!/usr/bin/env python3
"""cProfile of the FIRST inhomogeneous solve of BASELINE configs[3] (modified Helmholtz k = 10, 4096^2 grid,
8192 nodes) after set-up: the one-time costs that are not in a warm solve."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch
import interior_modified_helmholtz as imh

real_call = imh.ModifiedHelmholtzSolver.__call__
state = {"n": 0}


def profiled(self, *a, **k):
    state["n"] += 1
    if state["n"] != 1:
        return real_call(self, *a, **k)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    out = real_call(self, *a, **k)
    torch.cuda.synchronize()
    pr.disable()
    print("first solve %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    return out


imh.ModifiedHelmholtzSolver.__call__ = profiled
err, scale, solver, ue, T = imh.run(nb=8192, M=20, helmholtz_k=10.0, Ns=[4096, 4096])
print({k: v for k, v in T.items()})
