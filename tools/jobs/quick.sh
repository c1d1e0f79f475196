# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
timeout -k 10 300 python3 tools/slice_scaling.py 2>&1 | tee $O/slice_scaling.txt
