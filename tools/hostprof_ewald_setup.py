#!/usr/bin/env python3
"""cProfile of the Ewald-split grid evaluator's construction at 2048^2 (second construction in the
process: one-time library loads excluded)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ipde_amd.grid_evaluators.laplace_grid_evaluator import LaplaceGridBackend, LaplaceFreespaceGridEvaluator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
xv = np.linspace(-1.5, 1.5, n, endpoint=False)
h = xv[1] - xv[0]
LaplaceFreespaceGridEvaluator(LaplaceGridBackend(h, 24, method='ewald'), xv[:512], xv[:512], allow_rectangular=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
ev = LaplaceFreespaceGridEvaluator(LaplaceGridBackend(h, 24, method='ewald'), xv, xv, allow_rectangular=True)
torch.cuda.synchronize()
pr.disable()
print("construction %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
