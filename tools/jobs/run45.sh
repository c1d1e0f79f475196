set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -15 > gpurun_out/r02/gputest_full2.log
cat gpurun_out/r02/gputest_full2.log
timeout -k 10 300 python3 tools/hostprof_setup.py 2>&1 | grep "setup_s" | cut -c1-300
