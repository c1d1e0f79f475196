#!/usr/bin/env python3
"""What one rank of a strong-scaling run of bench.py does, on one GPU: the pair-by-pair Laplace SLP
kernel on the first 1/N of the BASELINE configs[1] target list (N = 1, 2, 4, 8), timed by the
library's event pairs — the projected parallel efficiency of the kernel alone (no collective)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from util import Curve, grid_targets
from ipde_amd import layer_potentials as lp, target_plan
from ipde_amd.device import get_context
from ipde_amd.sharding import target_slice

c = Curve(4096, a=0.2, f=5)
trg, h = grid_targets(c, 2048)
ctx = get_context()
dev = ctx.torch_device()
sx, sy, w = (torch.as_tensor(a, device=dev) for a in (c.x, c.y, c.weights))
sig = torch.as_tensor(np.random.default_rng(0).standard_normal(c.N), device=dev)
base = None
for N in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, N // 2, N - 1}):
        sl = target_slice(trg.N, rank, N)
        dt = lp.DeviceTargets(trg.x[sl], trg.y[sl], ctx=ctx, plan=True)
        plan = dt.plan()
        out = torch.empty(dt.N, dtype=torch.float64, device=dev)
        ctx.enable_timing(True)
        for _ in range(12):
            if plan is not None:
                target_plan.laplace_apply(plan, sx, sy, w_sigma=sig * w, ctx=ctx, out=out)
            else:
                lp.laplace_apply(sx, sy, dt.x, dt.y, w_sigma=sig * w, ctx=ctx, out=out)
        torch.cuda.synchronize()
        ms = float(np.mean(ctx.kernel_ms_history()[-8:]))
        ctx.enable_timing(False)
        worst = max(worst, ms)
    base = base or worst
    print("N = %d: slowest rank's kernel %.3f ms (plan: %s)  -> kernel-only efficiency %.1f %%"
          % (N, worst, "patches" if plan is not None else "list", 100.0 * base / (N * worst)))
