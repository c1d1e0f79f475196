# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 tools/hostprof_setup.py > $O/hostprof_setup.txt 2>&1
head -90 $O/hostprof_setup.txt | cut -c1-170
