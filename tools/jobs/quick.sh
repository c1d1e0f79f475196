# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 400 python3 tools/hostprof_first_solve.py 2>&1 | grep "first solve"
export IPDE_PROFILE_SOLVES=60
for i in 1 2; do timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve"; done
timeout -k 10 1000 python3 -m pytest tests/test_solver_gpu.py tests/test_annular_gpu.py -m gpu -x -q 2>&1 | tail -3
