#!/usr/bin/env python3
"""Blocked substitution (ipde_dense_lu_solve_batch) per call: one system and two in lock-step, n = 4096 / 8192 / 19200."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ipde_amd import qfs

for n in (4096, 8192, 19200):
    rng = np.random.default_rng(n)
    A = torch.as_tensor(rng.standard_normal((n, n)) + 0.05 * n * np.eye(n), device="cuda")
    f = qfs._own_lu(A)
    b = torch.as_tensor(rng.standard_normal(n), device="cuda")
    for nsys in (1, 2):
        facts, bs = [f] * nsys, [b] * nsys
        for _ in range(3):
            qfs._DeviceLU._subst_batch(facts, bs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            qfs._DeviceLU._subst_batch(facts, bs)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
        print("n = %5d, %d system(s): %.3f ms per substitution (two passes), %.2f us per block pair and pass, factors at %.2f TB/s"
              % (n, nsys, t * 1e3, t * 1e6 / 2 / ((n + 127) // 128), nsys * n * n * 8 / t / 1e12), flush=True)
    del A, f
