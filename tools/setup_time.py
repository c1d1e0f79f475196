#!/usr/bin/env python3
"""Set-up of the 3-body Stokes solver at configs[4] (n_b = 2400, 4096^2): `setup_s` as the example brackets it and
the time to a drained GPU, second and third construction in the process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch
import multi_stokes as ms
ms.run(nb=400, M=12, simple=True)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    ue, ve, pe, scale, T = ms.run(nb=2400, M=14, ng=4096)
    torch.cuda.synchronize()
    print("setup_s %.3f  first solve %.3f  whole run %.3f  (errors %.1e %.1e)"
          % (T['setup_s'], T['inhomogeneous_solve_s'], time.perf_counter() - t0, ue, ve), flush=True)
