set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 800 python -m pytest tests/test_spectral_gpu.py tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -x -q -s -k "interp or stokes or Stokes" > gpurun_out/r02/gputest_interp.log 2>&1 || (tail -40 gpurun_out/r02/gputest_interp.log; exit 1)
grep -E "grid \(|passed|failed" gpurun_out/r02/gputest_interp.log
echo done
