set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_solver_gpu.py -m gpu -x -q -k "modified_helmholtz_solver_far" 2>&1 | grep -B5 -A25 "def test_modified\|Error" | tail -60
