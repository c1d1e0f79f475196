#!/usr/bin/env python3
"""Clocks and socket power while the Laplace kernels run back to back (rocm-smi sampled from a
thread): is the dense sum's issue rate bounded by the power limit?  Prints, per kernel, the
median sclk and power over a ~4 s loop, and the idle values."""
import json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import Curve, grid_targets
from ipde_amd import layer_potentials as lp, target_plan

c = Curve(4096, a=0.2, f=5)
trg, h = grid_targets(c, 2048)
dt = lp.DeviceTargets(trg)
src = lp._source_side(c, dt)
w = lp._weighted(np.random.default_rng(0).standard_normal(c.N), src.weights)
plan = target_plan.build(dt.x, dt.y)
out = torch.empty(dt.N, dtype=torch.float64, device=dt.x.device)


def sample():
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp", "--json"],
                           capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        return next(iter(d.values()))
    except Exception as e:
        return {"error": repr(e)}


print("idle:", json.dumps(sample())[:600], flush=True)
for name, fn in (("list", lambda: lp.laplace_apply(src.x, src.y, dt.x, dt.y, w_sigma=w, out=out)),
                 ("patches", lambda: target_plan.laplace_apply(plan, src.x, src.y, w_sigma=w, out=out)),
                 ("list", lambda: lp.laplace_apply(src.x, src.y, dt.x, dt.y, w_sigma=w, out=out))):
    samples, stop = [], threading.Event()

    def watcher():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.2)
    th = threading.Thread(target=watcher); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    dtm = (time.perf_counter() - t0) / n * 1e3
    stop.set(); th.join()
    print(f"{name}: {dtm:.3f} ms per apply; samples:", flush=True)
    for s in samples[2:8]:
        print("   ", json.dumps({k: v for k, v in s.items() if any(t in k.lower() for t in ("power", "sclk", "temp", "error"))})[:400], flush=True)
