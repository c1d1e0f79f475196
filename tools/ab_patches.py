#!/usr/bin/env python3
"""A/B of the Laplace list kernel against the 4 x 4 patch kernel on the bench's target list
(2048^2 grid minus the band, 4096 sources): kernel ms (hipEvents inside the library, median),
wall ms of the whole planned apply, max difference; for several hand-out orders of the tiles."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import Curve, grid_targets
from ipde_amd.device import get_context
from ipde_amd import layer_potentials as lp, target_plan

ctx = get_context()
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
c = Curve(2 * ng, a=0.2, f=5)
trg, h = grid_targets(c, ng)
dt = lp.DeviceTargets(trg)
rng = np.random.default_rng(0)
sig = rng.standard_normal(c.N)
src = lp._source_side(c, dt)
w = lp._weighted(sig, src.weights)
ctx.enable_timing(True)


def wall(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[2:]))


def kern(fn, reps=7):
    ts = []
    for _ in range(reps):
        fn(); torch.cuda.synchronize(); ts.append(ctx.last_kernel_ms())
    return float(np.median(ts[2:]))


for mode, kw in (("slp", dict(w_sigma=w)), ("dlp", dict(nx=src.normal_x, ny=src.normal_y, w_tau=w)),
                 ("both", dict(w_sigma=w, nx=src.normal_x, ny=src.normal_y, w_tau=w))):
    ref = lp.laplace_apply(src.x, src.y, dt.x, dt.y, **kw)
    f = lambda: lp.laplace_apply(src.x, src.y, dt.x, dt.y, **kw)
    print(f"{mode} list kernel: {kern(f):.3f} ms kernel, {wall(f):.3f} ms wall, N = {dt.N}", flush=True)
    for block in ((1, 1 << 20), (8, 8), (4, 4), (2, 32), (4, 16), (16, 4)) if mode == "slp" else ((8, 8),):
        t0 = time.perf_counter()
        plan = target_plan.build(dt.x, dt.y, block=block)
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        only = torch.empty(plan.n, dtype=torch.float64, device=dt.x.device)
        g = lambda: ctx.check(ctx.lib.ipde_laplace_apply_patches(
            ctx.handle, int(src.x.shape[0]), lp.ptr(src.x), lp.ptr(src.y), lp.ptr(kw.get("w_sigma")),
            lp.ptr(kw.get("nx")), lp.ptr(kw.get("ny")), lp.ptr(kw.get("w_tau")), plan.np, lp.ptr(plan.pxy),
            lp.ptr(plan.pout), lp.ptr(only)))
        km = kern(g)
        p = lambda: target_plan.laplace_apply(plan, src.x, src.y, **kw)
        out = p()
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"{mode} patches block {block}: np = {plan.np}, rest = {plan.nrest}, plan built in {tb * 1e3:.1f} ms; "
              f"patch kernel {km:.3f} ms, planned apply wall {wall(p):.3f} ms, max rel diff {err:.2e}", flush=True)

# sustained: blocks of 20 back-to-back applies, the two kernels alternating (clocks under load)
plan = target_plan.build(dt.x, dt.y)
out = torch.empty(dt.N, dtype=torch.float64, device=dt.x.device)
fl = lambda: lp.laplace_apply(src.x, src.y, dt.x, dt.y, w_sigma=w, out=out)
fp = lambda: target_plan.laplace_apply(plan, src.x, src.y, w_sigma=w, out=out)
for rnd in range(4):
    for name, fn in (("list", fl), ("patches", fp)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(f"sustained round {rnd} {name}: {(time.perf_counter() - t0) * 50:.3f} ms per apply", flush=True)
