#!/usr/bin/env python3
"""A/B timing of the 2-D FFT grid operators (select the library with IPDE_HIP_LIBRARY)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ipde_amd.spectral import get_plan
res = {}
for n in (2048, 4096):
    f = torch.randn(n, n, dtype=torch.float64, device="cuda"); f -= f.mean()
    plan = get_plan(n, n, 3.0 / n, 3.0 / n)
    for name, fn in (("poisson", lambda: plan.poisson_solve(f)), ("dx", lambda: plan.dx(f))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize()
        res["%s_%d_us" % (name, n)] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
print(os.environ.get("IPDE_HIP_LIBRARY", "default"), res)
