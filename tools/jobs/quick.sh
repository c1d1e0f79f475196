# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
SECONDS=0; python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
echo "bench wall ${SECONDS} s"
python3 -c "
import json; r=json.load(open('$O/bench.json')); print(json.dumps(r['baseline_configs']['configs[4]'])[:600])"
