set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py tests/test_layer_golden.py tests/test_ewald_gpu.py -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | tail -5
