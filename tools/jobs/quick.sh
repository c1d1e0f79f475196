# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 python3 tools/ab_gmres_cgs2.py 2>&1 | grep -v Warn | tail -3
