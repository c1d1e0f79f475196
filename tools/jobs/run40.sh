set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02/gputest_devflow.log
cat gpurun_out/r02/gputest_devflow.log
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -1
