# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "column_far or helmholtz or config3" 2>&1 | tail -4
export IPDE_PROFILE_SOLVES=20
for i in 1 2; do timeout -k 10 400 python3 tools/profile_modhelm_solve.py 2>&1 | grep "warm" | sed 's/^/configs[3] /'; done
