#!/usr/bin/env python3
"""Does the ORDER of a grid list matter to the table kernels?  The same 2048^2 x 4096 sums on the list in
C order and on the list enumerated tile by tile (4 x 4 tiles, tiles in 8 x 8 blocks: the 16 lanes of an LDS
pass then hold one compact tile, i.e. few distinct table entries).  Kernel ms by hipEvents, median of 7."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import Curve, grid_targets
from ipde_amd.device import get_context
from ipde_amd import layer_potentials as lp, target_plan
ctx = get_context()
c = Curve(4096, a=0.2, f=5)
trg, h = grid_targets(c, 2048)
dt = lp.DeviceTargets(trg)
plan = target_plan.build(dt.x, dt.y)
perm = plan.pout.t().reshape(-1).long()
perm = torch.cat([perm[perm >= 0], plan.rest])
assert perm.numel() == dt.N
dtp = lp.DeviceTargets(dt.x[perm].contiguous(), dt.y[perm].contiguous())
rng = np.random.default_rng(0)
sig = rng.standard_normal(c.N)
f2 = rng.standard_normal((2, c.N))
ctx.enable_timing(True)


def kern(fn):
    ts = []
    for _ in range(9):
        fn(); torch.cuda.synchronize(); ts.append(ctx.last_kernel_ms())
    return float(np.median(ts[2:]))


for name, fn in (("modhelm slp", lambda t: lp.Modified_Helmholtz_Layer_Apply(c, t, k=10.0, charge=sig)),
                 ("modhelm dlp", lambda t: lp.Modified_Helmholtz_Layer_Apply(c, t, k=10.0, dipstr=sig)),
                 ("laplace slp (list kernel)", lambda t: lp.Laplace_Layer_Apply(c, t, charge=sig)),
                 ("stokes slp", lambda t: lp.Stokes_Layer_Apply(c, t, forces=f2))):
    a = fn(dt); b = fn(dtp)
    a0 = a[0] if isinstance(a, tuple) else a
    b0 = b[0] if isinstance(b, tuple) else b
    err = float((b0 - a0[perm]).abs().max() / a0.abs().max())
    print(f"{name}: C order {kern(lambda: fn(dt)):.3f} ms, tile order {kern(lambda: fn(dtp)):.3f} ms, max rel diff {err:.1e}", flush=True)
