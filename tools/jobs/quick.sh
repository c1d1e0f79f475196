# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py tests/test_solver_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "stokes" 2>&1 | tail -4
timeout -k 10 300 python3 tools/ab_far_expansion.py 2>&1 | grep "stokes" | tee $O/ab_far.txt
