#!/usr/bin/env python3
"""Own LU factorisation timed at a few sizes (and checked against the schedule selected by the environment)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ipde_amd import qfs

for n in [int(a) for a in sys.argv[1:]] or [4096, 9600, 19200]:
    rng = np.random.default_rng(n)
    A = torch.as_tensor(rng.standard_normal((n, n)), device="cuda")
    f = qfs._own_lu(A)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f = qfs._own_lu(A)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    h = float(f.LU.double().sum()), int(f.perm.long().sum()), float((f.LU * f.LU).sum())
    print("n = %5d: %.1f ms  (%.1f TFLOP/s)  checksum %r" % (n, t * 1e3, 2 * n ** 3 / 3 / t / 1e12, h), flush=True)
