# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_ewald_gpu.py tests/test_ewald_cpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not rank and not rehearsal" 2>&1 | tail -4
timeout -k 10 600 python3 - <<'PY' 2>&1 | grep -v Warn | tee $O/ewald_setup.txt
import sys, time
sys.path.insert(0, 'examples')
import torch, interior_poisson
for gb in (None, 'ewald', 'ewald', None):
    t0 = time.perf_counter()
    err, scale, solver, ue, T = interior_poisson.run(nb=4096, M=20, Ns=[2048, 2048], solver_tol=1e-12, grid_backend=gb)
    torch.cuda.synchronize()
    print(gb, 'total %.3f' % (time.perf_counter() - t0), {k: round(v, 3) for k, v in T.items() if isinstance(v, float)}, 'err %.2e' % (err / scale), flush=True)
    del solver, ue
PY
