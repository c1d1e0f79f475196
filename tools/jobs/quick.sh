set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline --no-fft > gpurun_out/b1.json 2> gpurun_out/b1.err
python3 -c "
import json; d=json.load(open('gpurun_out/b1.json')); f=d['full_poisson_solve']; print(d['value'], {k:v for k,v in f.items() if not isinstance(v,(dict,list,str))})"
timeout -k 10 300 python3 tools/cold_solve.py 2>/dev/null | tail -1 | cut -c1-400
timeout -k 10 600 python3 -m pytest tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -3
