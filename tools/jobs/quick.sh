# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gputest_last.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
