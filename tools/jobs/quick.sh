# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
SECONDS=0; python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -30 $O/bench_default.err; exit 1; }
echo "bench wall ${SECONDS} s"
python3 -c "
import json; r=json.load(open('$O/bench_default.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac']); print(json.dumps(r['baseline_configs'], indent=1)); print({k: v for k, v in r['full_poisson_solve'].items() if 'warm' in k and 'note' not in k})"
