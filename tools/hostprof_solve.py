#!/usr/bin/env python3
"""cProfile of warm 2048^2 Poisson solves: where the HOST spends the solve (top cumulative)."""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np
import torch
import interior_poisson
from ipde_amd.embedded_function import EmbeddedFunction
err, scale, solver, ue, T = interior_poisson.run(nb=4096, M=20, Ns=[2048, 2048], solver_tol=1e-12)
f = EmbeddedFunction(solver.ebdyc)
f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
for _ in range(3):
    solver(f, tol=1e-12, maxiter=100, restart=20)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    solver(f, tol=1e-12, maxiter=100, restart=20)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
