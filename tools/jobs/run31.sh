set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 300 python3 -m pytest tests/test_dense_gpu.py -m gpu -x -q -s -k "lu_solve or substitution or qfs" 2>&1 | tail -15 > gpurun_out/r02/gputest_lu.log
cat gpurun_out/r02/gputest_lu.log
