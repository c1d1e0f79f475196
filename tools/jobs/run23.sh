set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_spectral_gpu.py -m gpu -x -q > gpurun_out/r02/gputest_fft.log 2>&1 || (tail -40 gpurun_out/r02/gputest_fft.log; exit 1)
tail -2 gpurun_out/r02/gputest_fft.log
python tools/ab_fft.py
echo done
