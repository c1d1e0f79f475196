# the whole GPU test-suite (the driver's literal command), smoke, bench (one MI355X)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1 || { tail -60 $O/gputest.txt; exit 1; }
tail -4 $O/gputest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --steps 20 --warmup 5 > $O/bench_b.json 2> $O/bench_b.err || { tail -20 $O/bench_b.err; exit 1; }
python3 -c "
import json; r=json.load(open('$O/bench_b.json')); print(r['value'], r['ms_per_step'], {k:v for k,v in r['roofline'].items() if k in ('kernel_ms','kernel_ms_isolated','frac','traffic')}, r['fft']['poisson_grid_solve'], r['fft']['traffic'], r['expansion_form'], {k: v for k, v in r['full_poisson_solve'].items() if 'warm' in k or 'grid' in k or k == 'setup_s'})"
