set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/jobs/dbg.py 2>&1 | tail -8
