# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_layer_gpu.py -m gpu -x -q -k "stokes_far_expansion_table_miss" 2>&1 | tail -12
