#!/usr/bin/env python3
"""Interface interpolation (values + gradient at points along a star curve) in a loop, for
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_interp.py [n npts band]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ipde_amd.device import get_context  # noqa: E402
from ipde_amd.spectral import GridPlan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
band = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = get_context()
ctx.set_option("interp_band", band)
h = 3.0 / n
x = torch.arange(n, dtype=torch.float64, device="cuda") * h
X, Y = torch.meshgrid(x, x, indexing="ij")
f = torch.exp(torch.sin(2 * np.pi * X / 3.0)) * torch.cos(4 * np.pi * Y / 3.0)
f -= f.mean()
del X, Y
th = 2 * np.pi * np.arange(npts) / npts
r = 1.0 + 0.2 * np.cos(5 * th)
px = torch.as_tensor((1.5 + r * np.cos(th)) * 2 * np.pi / 3.0, device="cuda")
py = torch.as_tensor((1.5 + r * np.sin(th)) * 2 * np.pi / 3.0, device="cuda")
plan = GridPlan(n, n, h, h)
plan.keep_spectrum(True)
plan.poisson_solve(f)
for _ in range(3):
    plan.interp_gradient(px, py)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    plan.interp_gradient(px, py)
torch.cuda.synchronize()
print("interp %d^2 x %d points, band %d: %.3f ms" % (n, npts, band, (time.perf_counter() - t0) / 20 * 1e3))
