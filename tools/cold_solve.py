#!/usr/bin/env python3
"""Cold-process timing of the full 2048^2 interior Poisson solve (BASELINE configs[2]):
process start -> imports -> set-up -> first solve -> homogeneous correction, then the warm
solve time.  One JSON line on stdout.
    python tools/cold_solve.py [--nb 4096 --M 20 --n 2048]"""
import time
T0 = time.perf_counter()
import argparse  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
ap = argparse.ArgumentParser()
ap.add_argument('--nb', type=int, default=4096)
ap.add_argument('--M', type=int, default=20)
ap.add_argument('--n', type=int, default=2048)
a = ap.parse_args()
import numpy as np  # noqa: E402
import torch  # noqa: E402
import interior_poisson  # noqa: E402
from ipde_amd.embedded_function import EmbeddedFunction  # noqa: E402
t_import = time.perf_counter() - T0
T = {}
err, scale, solver, ue, T = interior_poisson.run(nb=a.nb, M=a.M, Ns=[a.n, a.n], solver_tol=1e-12, timings=T)
t_end = time.perf_counter() - T0
f = EmbeddedFunction(solver.ebdyc)
f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
solver(f, tol=1e-12, maxiter=100, restart=20)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    solver(f, tol=1e-12, maxiter=100, restart=20)
torch.cuda.synchronize()
out = {k: v for k, v in T.items() if isinstance(v, (int, float, str, list, tuple))}      # (not the resident objects)
out.update(import_s=t_import, process_start_to_solution_s=t_end, rel_err=err / scale,
           warm_inhomogeneous_solve_ms=(time.perf_counter() - t0) / 10 * 1e3,
           rocfft_kernel_cache=os.environ.get('ROCFFT_RTC_CACHE_PATH'))
print(json.dumps(out))
