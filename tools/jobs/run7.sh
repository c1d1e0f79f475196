set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_spectral_gpu.py -m gpu -x -q -s -k "interp or pipeline" > gpurun_out/r02/gputest_interp.log 2>&1 || (tail -50 gpurun_out/r02/gputest_interp.log; exit 1)
grep -E "grid \(|passed|failed" gpurun_out/r02/gputest_interp.log
timeout -k 10 900 python -m pytest tests/test_solver_gpu.py tests/test_configs_gpu.py tests/test_compat.py -m gpu -x -q > gpurun_out/r02/gputest_solv.log 2>&1 || (tail -50 gpurun_out/r02/gputest_solv.log; exit 1)
tail -3 gpurun_out/r02/gputest_solv.log
python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02/bench_d.json 2> gpurun_out/r02/bench_d.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/bench_d.json')); print(json.dumps(b['full_poisson_solve']))"
echo done
