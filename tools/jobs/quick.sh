# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py -m gpu -x -q -k "ragged_rectangular" 2>&1 | tail -15
