# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/gputest_two_level.txt
