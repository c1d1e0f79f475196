set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02/gputest_full.log
cat gpurun_out/r02/gputest_full.log
