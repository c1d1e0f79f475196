set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/ab_gmres_lookahead.py 2>&1 | grep -v amdgpu.ids | tail
