set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3 4; do timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1; done
