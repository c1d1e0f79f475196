# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 300 python3 tools/ab_far_expansion.py 2>&1 | tee $O/ab_far.txt
