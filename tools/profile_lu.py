#!/usr/bin/env python3
"""A few own LU factorisations (csrc/lu_factor.hip) of a random n x n matrix — for
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_lu.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ipde_amd import qfs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = torch.as_tensor(np.random.default_rng(0).standard_normal((n, n)), device="cuda")
qfs._own_lu(A)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    qfs._own_lu(A)
torch.cuda.synchronize()
print("n = %d: %.2f ms per factorisation (tiling included)" % (n, (time.perf_counter() - t0) / 5 * 1e3))
