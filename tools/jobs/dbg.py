import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from util import Curve, grid_targets
from ipde_amd import layer_potentials as lp, target_plan
c = Curve(512, a=0.2, f=5)
trg, h = grid_targets(c, 640, clearance=2.0)
rng = np.random.default_rng(1152)
fx, fy = rng.standard_normal(c.N) * c.weights, rng.standard_normal(c.N) * c.weights
dev = lp.get_context().torch_device()
plan = target_plan.build_host(trg.x, trg.y, device=dev, pad_blocks=True)
u, v, p = (a.cpu().numpy() for a in target_plan.stokes_apply(plan, c.x, c.y, fx, fy))
lu, lv, lpp = lp.stokes_apply(c.x, c.y, trg.x, trg.y, wfx=fx, wfy=fy)
bad = np.nonzero(np.abs(u - lu) > 1e-10)[0]
print("np", plan.np, "blocks", plan.np // 64, "parents", -(-plan.np // 1024), "bad targets", bad.size)
pout = plan.pout.cpu().numpy()
owner = np.full(trg.N, -1)
for r in range(16):
    m = pout[r] >= 0
    owner[pout[r][m]] = np.nonzero(m)[0]
pb = owner[bad]
print("bad patches", np.unique(pb)[:20], "blocks", np.unique(pb // 64), "parents", np.unique(pb // 1024))
print("err sample", (u - lu)[bad][:5], lu[bad][:5])
