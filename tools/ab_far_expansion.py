#!/usr/bin/env python3
"""Laplace sums onto the 2048^2 band list (BASELINE configs[1]): the direct patch kernel against the
far-field form (ipde_laplace_apply_patches_far) — difference and kernel times by event pairs."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from util import Curve, grid_targets
from ipde_amd import layer_potentials as lp, target_plan
from ipde_amd.device import get_context

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = Curve(nb, a=0.2, f=5)
trg, h = grid_targets(c, ng)
ctx = get_context()
plain = lp.DeviceTargets(trg, plan=True)
far = lp.DeviceTargets(trg, plan=True, far=True)
p0, p1 = plain.plan(), far.plan()
print("targets %d  patches %d  padded patches %d (+%.1f %%)" % (trg.N, p0.np, p1.np, 100.0 * (p1.np / p0.np - 1)))
rng = np.random.default_rng(0)
s1, s2 = rng.standard_normal(c.N), rng.standard_normal(c.N)
for name, kw in (("slp", dict(charge=s1)), ("dlp", dict(dipstr=s2)), ("both", dict(charge=s1, dipstr=s2))):
    a = lp.Laplace_Layer_Apply(c, plain, **kw)
    b = lp.Laplace_Layer_Apply(c, far, **kw)
    torch.cuda.synchronize()
    d = float(torch.max(torch.abs(a - b)))
    print("%-4s max|direct| %.3e  max|far - direct| %.3e  (rel %.2e)" % (name, float(a.abs().max()), d, d / float(a.abs().max())))
    for tname, t in (("direct", plain), ("far", far)):
        for _ in range(3):
            lp.Laplace_Layer_Apply(c, t, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            lp.Laplace_Layer_Apply(c, t, **kw)
        torch.cuda.synchronize()
        print("   %-6s %.3f ms per apply (wall, %d applies)" % (tname, (time.perf_counter() - t0) / n * 1e3, n))

def timed(call, n=10):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# Stokes sums with pressure: stokeslet, stresslet, both (round 4: the double layer has its far-field form too)
plain_list = lp.DeviceTargets(trg)
f, g = rng.standard_normal((2, c.N)), rng.standard_normal((2, c.N))
for name, kw in (("stokeslet", dict(forces=f)), ("stresslet", dict(dipstr=g)), ("both", dict(forces=f, dipstr=g))):
    a = lp.Stokes_Layer_Apply(c, plain_list, **kw)
    b = lp.Stokes_Layer_Apply(c, far, **kw)
    torch.cuda.synchronize()
    print("stokes %-9s " % name + "  ".join("%s: max %.2e diff %.2e" % (q, float(x.abs().max()), float((x - y).abs().max()))
                                          for q, x, y in zip("uvp", a, b)))
    print("   stokes %-9s direct %.3f ms   far %.3f ms per apply (wall)"
          % (name, timed(lambda: lp.Stokes_Layer_Apply(c, plain_list, **kw)), timed(lambda: lp.Stokes_Layer_Apply(c, far, **kw))))

# modified Helmholtz: single layer, double layer, both
for k in (10.0, 100.0):
    s_, d_ = rng.standard_normal(c.N), rng.standard_normal(c.N)
    for name, kw in (("slp", dict(charge=s_)), ("dlp", dict(dipstr=d_)), ("both", dict(charge=s_, dipstr=d_))):
        a = torch.as_tensor(lp.Modified_Helmholtz_Layer_Apply(c, plain_list, k=k, **kw))
        b = torch.as_tensor(lp.Modified_Helmholtz_Layer_Apply(c, far, k=k, **kw))
        torch.cuda.synchronize()
        print("modhelm k = %g %-4s max|direct| %.3e  max|far - direct| %.3e" % (k, name, float(a.abs().max()), float((a - b).abs().max())))
        print("   modhelm %-4s direct %.3f ms   far %.3f ms per apply (wall)"
              % (name, timed(lambda: lp.Modified_Helmholtz_Layer_Apply(c, plain_list, k=k, **kw)),
                 timed(lambda: lp.Modified_Helmholtz_Layer_Apply(c, far, k=k, **kw))))
