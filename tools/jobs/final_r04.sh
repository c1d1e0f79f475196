# round-4 final evidence on one MI355X: the driver's literal test command, smoke, bench, the bench under rocprofv3
# (kernel stats) and the two HBM-traffic counter passes of the bench kernel (counters in runs of their own).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04/final
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1 || { tail -60 $O/gputest.txt; exit 1; }
tail -3 $O/gputest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve > $O/bench_under_rocprof.json 2> $O/bench_trace.err
cp $(ls -t $(find $O/bench_trace -name "*kernel_stats.csv") | head -1) $O/bench_kernel_stats.csv
rm -rf $O/bench_trace
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > $O/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > $O/bench_write.log 2>&1
python3 tools/collect_traffic.py $O/bench_fetch $O/bench_write laplace_patch_kernel r04 > $O/traffic.txt 2>&1 || true
cp profiles/traffic_r04.json $O/ 2>/dev/null || true
cp profiles/traffic_latest.json $O/traffic_latest.json 2>/dev/null || true
rm -rf $O/bench_fetch $O/bench_write
# the bare N > 1 command on the one GPU (two ranks share it, gloo): the self-launch path end to end
python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft --rehearse-shared-gpu > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err || { tail -20 $O/bench_gpus2_rehearsal.err; exit 1; }
python3 -c "
import json; r=json.load(open('$O/bench.json')); f=r['full_poisson_solve']
print(r['value'], r['ms_per_step'], {k:v for k,v in r['roofline'].items() if k in ('kernel_ms','kernel_ms_isolated','frac')}, r['fft']['poisson_grid_solve']['ms'])
print({k: f[k] for k in ('setup_s','warm_inhomogeneous_solve_ms','warm_inhomogeneous_solve_resident_ms','warm_homogeneous_apply_ms','warm_homogeneous_apply_resident_ms','warm_end_to_end_solve_ms','warm_end_to_end_solve_resident_ms')})
print(json.dumps(r['baseline_configs'])[:1500])
print(json.load(open('$O/bench_gpus2_rehearsal.json'))['n_gpus'])"
