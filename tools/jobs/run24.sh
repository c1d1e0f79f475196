set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python tools/bench_kernels.py > gpurun_out/r02/kernels_final.json 2> gpurun_out/r02/kernels_final.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/kernels_final.json'))
for k,v in b.items():
    if isinstance(v, dict) and 'kernel_ms' in v: print(k, '%.3f ms %.3e pairs/s' % (v['kernel_ms'], v['pairs_per_s']))
    elif isinstance(v, dict): print(k, json.dumps(v)[:200])
"
python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/bench_final.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], json.dumps(b['fft']['poisson_grid_solve'])); print(json.dumps(b['full_poisson_solve'])[:700])"
rm -rf gpurun_out/r02/solve_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/solve_trace -- python3 tools/profile_solve.py > gpurun_out/r02/solve_trace.log 2>&1
ms=$(grep "warm solve" gpurun_out/r02/solve_trace.log | awk '{print $3}')
python3 tools/analyze_trace.py gpurun_out/r02/solve_trace $ms 10 > gpurun_out/r02/solve_budget.json
python3 -c "
import json; b=json.load(open('gpurun_out/r02/solve_budget.json')); print({k:v for k,v in b.items() if k!='kernels_ms_per_solve'})"
echo done
