# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
for i in 1 2; do
timeout -k 10 500 python3 tools/run_sharded_solve.py --problem modhelm --nb 8192 --M 20 --k 10 --ng 4096 > $O/c3.json 2>> $O/c3.err
python3 -c "
import json; d=json.load(open('$O/c3.json')); print('config3', 'setup %.2f s' % d['timings']['setup_s'], 'first %.3f s' % d['timings']['inhomogeneous_solve_s'], 'warm %.1f ms' % (1e3*d['warm_inhomogeneous_solve_s']), 'err %.1e' % d['error'])"
done
timeout -k 10 900 python3 -m pytest tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "helmholtz or config3" 2>&1 | tail -3
