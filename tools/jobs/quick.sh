# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 - <<'PY' 2>&1 | grep -v Warn | tee $O/config3_setup.txt
import sys, time
sys.path.insert(0, 'examples')
import torch, interior_modified_helmholtz as imh
import numpy as np
from ipde_amd.embedded_function import EmbeddedFunction
for gb in (None, None, 'hip'):
    t0 = time.perf_counter()
    err, scale, solver, ue, T = imh.run(nb=8192, M=20, helmholtz_k=10.0, Ns=[4096, 4096], grid_backend=gb)
    torch.cuda.synchronize()
    f = EmbeddedFunction(solver.ebdyc); f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
    solver(f, tol=1e-12, maxiter=100, restart=20); torch.cuda.synchronize()
    t1 = time.perf_counter(); solver(f, tol=1e-12, maxiter=100, restart=20); torch.cuda.synchronize(); warm = time.perf_counter() - t1
    print(gb, 'split' if solver.split_grid_evaluation else 'dense', 'total %.3f' % (time.perf_counter() - t0), {k: round(v, 3) for k, v in T.items() if isinstance(v, float)}, 'err %.2e' % (err / scale), 'warm %.1f ms' % (1e3 * warm), flush=True)
    del solver, ue, f
    import gc; gc.collect(); torch.cuda.empty_cache()
PY
