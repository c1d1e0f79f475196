set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_annular_gpu.py tests/test_solver_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -5
for i in 1 2 3; do timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1; done
