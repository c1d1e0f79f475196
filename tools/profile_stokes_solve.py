#!/usr/bin/env python3
"""Warm 3-body Stokes solves (examples/multi_stokes.py set-up, nb = 800) in a loop — for
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_stokes_solve.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch  # noqa: E402
import multi_stokes as ms  # noqa: E402

state = {}
orig = ms.StokesSolver.__call__


def wrapped(self, fu, fv, **kw):
    out = orig(self, fu, fv, **kw)
    if 'done' not in state:
        state['done'] = True
        if os.environ.get("IPDE_PROFILE_RESIDENT") == "1":     # forcings and answers stay in HBM
            from ipde_amd import hostio
            fu, fv = hostio.DeviceFunction.from_host(fu), hostio.DeviceFunction.from_host(fv)
        orig(self, fu, fv, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = int(os.environ.get('IPDE_PROFILE_SOLVES', '10'))
        for _ in range(n):
            orig(self, fu, fv, **kw)
        torch.cuda.synchronize()
        print("warm stokes solve %.2f ms" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
        if os.environ.get("IPDE_PROFILE_STOP_AFTER_WARM"):
            # end the process here (normally, so that a profiler flushes): the trace then ends
            # with the ten warm solves, which is what tools/analyze_trace.py assumes
            raise SystemExit(0)
    return out


ms.StokesSolver.__call__ = wrapped
if os.environ.get("IPDE_VECTOR_DEVICE_FLOW") is not None:      # A/B of the helper flow
    from ipde_amd.solvers.multi_boundary.vector import VectorSolver
    VectorSolver.DEVICE_FLOW = os.environ["IPDE_VECTOR_DEVICE_FLOW"] != "0"
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 800
ng = int(sys.argv[2]) if len(sys.argv) > 2 else None       # 4096: BASELINE configs[4] (with nb = 2400)
ms.run(nb, 14, **({} if ng is None else {'ng': ng}))
