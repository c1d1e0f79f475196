#!/usr/bin/env python3
"""Set-up of the 3-body Stokes solver at BASELINE configs[4] (n_b = 2400, 4096^2 grid), second construction in the
process, for   rocprofv3 --kernel-trace --stats -- python3 tools/profile_stokes_setup.py   (which kernels the GPU
side of the set-up is made of: matrix assembly, factorisations, inverse blocks)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import torch  # noqa: E402
import multi_stokes as ms  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


class Stop(Exception):
    pass


def stop(self, *a, **k):
    torch.cuda.synchronize()
    raise Stop()


ms.run(nb=400, M=12, simple=True)          # one-time loads
torch.cuda.synchronize()
orig = ms.StokesSolver.__call__
ms.StokesSolver.__call__ = stop            # the run ends where the first solve would start
t0 = time.perf_counter()
try:
    ms.run(nb=nb, M=14, ng=ng)
except Stop:
    pass
print("set-up to first solve, GPU drained: %.3f s" % (time.perf_counter() - t0))
