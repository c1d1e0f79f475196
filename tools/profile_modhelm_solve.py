#!/usr/bin/env python3
"""Warm modified Helmholtz solves at BASELINE configs[3] (k = 10, 4096^2 grid, 8192 nodes) in a loop — for
    rocprofv3 --kernel-trace -- python3 tools/profile_modhelm_solve.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import interior_modified_helmholtz as imh  # noqa: E402
from ipde_amd.embedded_function import EmbeddedFunction  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
err, scale, solver, ue, T = imh.run(nb=nb, M=20, helmholtz_k=10.0, Ns=[ng, ng],
                                    grid_backend=os.environ.get('IPDE_PROFILE_GRID_BACKEND') or None)
print('error %.3e' % (err / scale))
f = EmbeddedFunction(solver.ebdyc)
f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
if os.environ.get("IPDE_PROFILE_RESIDENT") == "1":
    from ipde_amd import hostio
    f = hostio.DeviceFunction.from_host(f)
solver(f, tol=1e-12, maxiter=100, restart=20)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get('IPDE_PROFILE_SOLVES', '10'))
for _ in range(n):
    solver(f, tol=1e-12, maxiter=100, restart=20)
torch.cuda.synchronize()
print("warm solve %.2f ms" % ((time.perf_counter() - t0) / n * 1e3))
