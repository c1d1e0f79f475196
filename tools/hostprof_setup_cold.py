#!/usr/bin/env python3
"""cProfile of the set-up + first solve of the 2048^2 Poisson example as the FIRST work of a
process (what bench.py's full_poisson_solve block sees on a fresh box)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch
import interior_poisson
from ipde_amd.device import get_context
get_context()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
err, scale, solver, ue, T = interior_poisson.run(nb=4096, M=20, Ns=[2048, 2048], solver_tol=1e-12,
                                                 grid_backend=os.environ.get('IPDE_PROFILE_GRID_BACKEND') or None)
torch.cuda.synchronize()
pr.disable()
print(T)
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
