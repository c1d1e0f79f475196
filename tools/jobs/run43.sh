set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 300 python3 -m pytest tests/test_lu_factor_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^  File\|^Extension" | tail -14
rm -rf gpurun_out/r02/lu_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/lu_trace -- python3 tools/profile_lu.py 4096 > gpurun_out/r02/lu_trace.log 2>&1
tail -1 gpurun_out/r02/lu_trace.log
f=$(find gpurun_out/r02/lu_trace -name "*kernel_stats.csv" | head -1)
head -6 $f | cut -c1-150
