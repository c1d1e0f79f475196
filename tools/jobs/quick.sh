set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "two_rank" 2>&1 | grep -v "^  File\|^Extension" | tail -6
