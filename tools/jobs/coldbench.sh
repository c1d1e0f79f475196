# what the driver sees: bench.py as the FIRST process on a fresh box, then once more (warm file cache)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for i in 1 2; do
python bench.py > gpurun_out/r02/bench_cold$i.json 2> gpurun_out/r02/bench_cold.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/bench_cold$i.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac']); f=b['full_poisson_solve']; print({k:f[k] for k in ('setup_s','first_inhomogeneous_solve_s','homogeneous_correction_s','end_to_end_s','warm_inhomogeneous_solve_ms')})"
done
