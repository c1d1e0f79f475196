# round-3 final evidence on one MI355X: the driver's literal test command, smoke, bench, the bench under rocprofv3
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/final
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1 || { tail -60 $O/gputest.txt; exit 1; }
tail -3 $O/gputest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve > $O/bench_under_rocprof.json 2> $O/bench_trace.err
python3 -c "
import json; r=json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'], {k:v for k,v in r['roofline'].items() if k in ('kernel_ms','kernel_ms_isolated','frac')}, r['fft']['poisson_grid_solve']['ms'], r['full_poisson_solve']['warm_inhomogeneous_solve_ms'], r['full_poisson_solve']['setup_s'], r['cpu_baseline']['value'])"
