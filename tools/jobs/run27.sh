set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 800 python -m pytest tests/test_spectral_gpu.py -m gpu -x -q -s -k "interp" > gpurun_out/r02/gputest_interp.log 2>&1 || (tail -40 gpurun_out/r02/gputest_interp.log; exit 1)
grep -E "grid \(|passed|failed" gpurun_out/r02/gputest_interp.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r02/gputest_k.log 2>&1 || (tail -40 gpurun_out/r02/gputest_k.log; exit 1)
tail -3 gpurun_out/r02/gputest_k.log
export IPDE_PROFILE_STOP_AFTER_WARM=1
python tools/profile_stokes_solve.py 2>/dev/null | grep warm
python examples/interior_poisson.py --nb 800 --M 20 2>/dev/null | tail -2
echo done
