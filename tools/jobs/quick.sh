set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests/test_layer_gpu.py tests/test_layer_golden.py tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -q -k "modhelm or helmholtz or config3 or extreme or empty" 2>&1 | grep -v "^  File\|^Extension" | tail -8
for v in 0 0; do timeout -k 10 300 python3 tools/ab_modhelm.py 2>&1 | tail -1; done
