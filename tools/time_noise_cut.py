#!/usr/bin/env python3
"""Cost of the QFS density noise cut (ipde_density_noise_cut through Stokes_QFS._lowpass) per call."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ipde_amd.qfs import Stokes_QFS

me = types.SimpleNamespace(NOISE_CUT=True, RISE=30.0, FLOOR=1e-5)
for n in (800, 2390, 3200, 9560):
    mu = torch.as_tensor(np.random.default_rng(n).standard_normal(2 * n), device="cuda")
    for _ in range(5):
        Stokes_QFS._lowpass(me, mu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        Stokes_QFS._lowpass(me, mu)
    torch.cuda.synchronize()
    print("N = %5d: %.1f us per call" % (n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
