#!/usr/bin/env python3
"""cProfile of examples/multi_stokes.py at BASELINE configs[4] size (n_b = 2400, 4096^2 grid), second run
in the process (one-time library loads excluded): where the 3-body Stokes set-up goes."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch
import multi_stokes as ms
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ms.run(nb=400, M=12, simple=True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
ue, ve, pe, scale, T = ms.run(nb=nb, M=14, ng=ng)
torch.cuda.synchronize()
pr.disable()
print(ue / scale, ve / scale, T)
pstats.Stats(pr).sort_stats("cumulative").print_stats(70)
