#!/usr/bin/env python3
"""A/B timing of the modified-Helmholtz table kernel (2048^2 x 4096, k = 10): prints kernel ms
(hipEvents, median of 7) for SLP and DLP.  Select the library with IPDE_HIP_LIBRARY."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import Curve, grid_targets
from ipde_amd.device import get_context
from ipde_amd import layer_potentials as lp
ctx = get_context()
c = Curve(4096, a=0.2, f=5)
trg, h = grid_targets(c, 2048)
dt = lp.DeviceTargets(trg)
rng = np.random.default_rng(0)
sig = rng.standard_normal(c.N)
ctx.enable_timing(True)
res = {}
for name, kw in (("slp", dict(charge=sig)), ("dlp", dict(dipstr=sig))):
    ts = []
    for _ in range(9):
        lp.Modified_Helmholtz_Layer_Apply(c, dt, k=10.0, **kw)
        torch.cuda.synchronize()
        ts.append(ctx.last_kernel_ms())
    res[name] = float(np.median(ts[2:]))
ts = []
for _ in range(7):
    lp.Laplace_Layer_Apply(c, dt, charge=sig); torch.cuda.synchronize(); ts.append(ctx.last_kernel_ms())
res["laplace_slp"] = float(np.median(ts[2:]))
print(os.environ.get("IPDE_HIP_LIBRARY", "default"), res)
