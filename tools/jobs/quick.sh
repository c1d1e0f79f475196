# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_configs_gpu.py tests/test_solver_gpu.py -m gpu -x -q -k "rank or sharded or two_ranks" > $O/quick_tests.txt 2>&1 || { tail -60 $O/quick_tests.txt; exit 1; }
tail -5 $O/quick_tests.txt
