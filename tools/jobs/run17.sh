set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -s -k "bench" > gpurun_out/r02/gputest_bench.log 2>&1 || (tail -40 gpurun_out/r02/gputest_bench.log; exit 1)
tail -4 gpurun_out/r02/gputest_bench.log
python bench.py > gpurun_out/r02/bench_e.json 2> gpurun_out/r02/bench_e.err
cut -c1-600 gpurun_out/r02/bench_e.json
echo done
