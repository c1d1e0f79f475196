#!/usr/bin/env python3
"""cProfile of the FIRST inhomogeneous solve of BASELINE configs[4] (3-body Stokes, 4096^2 grid) after set-up."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch
import multi_stokes as ms

real_call = ms.StokesSolver.__call__
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
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
    return out


ms.StokesSolver.__call__ = profiled
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ms.run(nb, 14, ng=ng)
