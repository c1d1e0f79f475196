# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py tests/test_solver_gpu.py -m gpu -x -q -k "column_far or poisson" 2>&1 | tail -4
export IPDE_PROFILE_SOLVES=60
for i in 1 2; do timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve\|error" | cut -c1-40; done
IPDE_PROFILE_RESIDENT=1 timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve" | sed 's/^/resident /'
