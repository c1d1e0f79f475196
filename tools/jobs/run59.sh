set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -6
for i in 1 2; do timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1; done
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -1
