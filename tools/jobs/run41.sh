set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 300 python3 -m pytest tests/test_lu_factor_gpu.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r02/gputest_lufactor.log
cat gpurun_out/r02/gputest_lufactor.log
