set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_dense_gpu.py tests/test_solver_gpu.py tests/test_lu_factor_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -5
for i in 1 2; do timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1; done
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -1
