set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 0; do timeout -k 10 300 python3 tools/ab_modhelm.py 2>&1 | tail -1; done
