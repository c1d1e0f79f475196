set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputest_a.log 2>&1 || (tail -40 gpurun_out/r02/gputest_a.log; exit 1)
tail -3 gpurun_out/r02/gputest_a.log
rocprofv3 -L > gpurun_out/r02/counters_avail.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/fft_trace -- python3 tools/profile_fft.py 2048 20 > gpurun_out/r02/fft_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/fft_fetch -- python3 tools/profile_fft.py 2048 4 > gpurun_out/r02/fft_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/fft_write -- python3 tools/profile_fft.py 2048 4 > gpurun_out/r02/fft_write.log 2>&1
echo done
