set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python bench.py > gpurun_out/r02/bench_cold_patches.json 2> gpurun_out/r02/bench_cold.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/bench_cold_patches.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['kernel_ms']); f=b['full_poisson_solve']; print({k:f[k] for k in ('setup_s','first_inhomogeneous_solve_s','homogeneous_correction_s','end_to_end_s','warm_inhomogeneous_solve_ms')})"
for v in 0 1 0 1; do IPDE_PATCH_TARGETS=$v timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1; done
timeout -k 10 1000 python3 -m pytest tests/test_layer_gpu.py tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -8
