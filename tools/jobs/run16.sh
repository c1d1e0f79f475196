set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write gpurun_out/r02/bench_trace
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > gpurun_out/r02/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > gpurun_out/r02/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/bench_trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-solve > gpurun_out/r02/bench_trace.json 2> gpurun_out/r02/bench_trace.err
tail -1 gpurun_out/r02/bench_trace.json | cut -c1-300
echo done
