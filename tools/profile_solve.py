#!/usr/bin/env python3
"""Warm full Poisson solves (2048^2 grid, 4096 nodes) in a loop — the target of
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_solve -- python3 tools/profile_solve.py
(the cold solve is included once; 10 warm solves dominate the kernel statistics)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import interior_poisson  # noqa: E402
from ipde_amd.embedded_function import EmbeddedFunction  # noqa: E402

T = {}
err, scale, solver, ue, T = interior_poisson.run(nb=4096, M=20, Ns=[2048, 2048], solver_tol=1e-12, timings=T,
                                                 grid_backend=os.environ.get('IPDE_PROFILE_GRID_BACKEND') or None)
print('error %.3e  timings %s' % (err / scale if scale else err, {k: round(v, 4) for k, v in T.items() if isinstance(v, float)}))
if os.environ.get("IPDE_AB_NULL_STREAM") == "1":   # A/B: the library's work on the legacy default stream, where torch's is
    import ctypes
    from ipde_amd import device as _dv
    for _c in list(_dv._contexts.values()):
        _c.sync()
        _c.check(_c.lib.ipde_ctx_set_stream(_c.handle, ctypes.c_void_p(1)))
f = EmbeddedFunction(solver.ebdyc)
f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
if os.environ.get("IPDE_PROFILE_RESIDENT") == "1":     # right-hand side and answer stay in HBM
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
