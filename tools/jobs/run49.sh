set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python3 -m pytest tests/test_lu_factor_gpu.py tests/test_dense_gpu.py tests/test_solver_gpu.py tests/test_configs_gpu.py tests/test_compat.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -8
for v in 1 1; do
IPDE_OWN_LU=$v python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fft 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); f=b['full_poisson_solve']; print('own_lu=$v', {k:f[k] for k in ('setup_s','first_inhomogeneous_solve_s','homogeneous_correction_s','end_to_end_s','warm_inhomogeneous_solve_ms')})"
done
IPDE_OWN_LU=1 timeout -k 10 300 python3 tools/hostprof_setup.py 2>&1 | grep "setup_s" | cut -c1-120
