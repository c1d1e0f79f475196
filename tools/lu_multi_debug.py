#!/usr/bin/env python3
"""Where do the own factorisation's pivots leave LAPACK's?  (IPDE_LU_FORCE_MULTI=1: the multi-CU panel at small n)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.linalg, torch
from ipde_amd import qfs
from ipde_amd.device import get_context

for n in [int(a) for a in sys.argv[1:]] or [700, 1500, 2050]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    f = qfs._own_lu(torch.as_tensor(A, device="cuda"))
    get_context().sync()
    lu, piv = scipy.linalg.lu_factor(A)
    p = np.arange(n)
    for i, q in enumerate(piv):
        if q != i:
            p[i], p[q] = p[q], p[i]
    mine = f.perm.cpu().numpy()
    if os.environ.get("IPDE_LU_DEBUG_ROCSOLVER"):
        _, rp = torch.linalg.lu_factor(torch.as_tensor(A, device="cuda"))
        rp = rp.cpu().numpy() - 1
        pr = np.arange(n)
        for i, q in enumerate(rp):
            if q != i:
                pr[i], pr[q] = pr[q], pr[i]
        b2 = np.nonzero(pr != p)[0]
        print(n, "rocSOLVER against host LAPACK: first pivot mismatch at", (int(b2[0]) if b2.size else None), "of", b2.size,
              flush=True)
    bad = np.nonzero(mine != p)[0]
    T = f.LU
    full = T.permute(0, 3, 1, 2).reshape(T.shape[0] * 64, -1)[:n, :n].cpu().numpy()
    print(n, "first pivot mismatch at", (int(bad[0]) if bad.size else None), "of", bad.size,
          " max|LU - lapack| = %.2e" % np.abs(full - lu).max(),
          " first bad column of the factors:", (int(np.nonzero(np.abs(full - lu).max(axis=0) > 1e-8)[0][:1].tolist()[0])
                                               if (np.abs(full - lu).max(axis=0) > 1e-8).any() else None))
