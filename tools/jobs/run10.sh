set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_layer_gpu.py tests/test_layer_golden.py tests/test_ewald_gpu.py -m gpu -x -q -k "modhelm or helm or empty or golden" > gpurun_out/r02/gputest_mh.log 2>&1 || (tail -50 gpurun_out/r02/gputest_mh.log; exit 1)
tail -3 gpurun_out/r02/gputest_mh.log
python tools/bench_kernels.py > gpurun_out/r02/kernels_a.json 2> gpurun_out/r02/kernels_a.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/kernels_a.json'))
for k,v in b.items():
    if isinstance(v, dict) and 'pairs_per_s' in v: print(k, '%.3f ms %.3e pairs/s' % (v['kernel_ms'], v['pairs_per_s']))
"
echo done
