# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_annular_gpu.py -m gpu -x -q > $O/quick_tests.txt 2>&1 || { tail -60 $O/quick_tests.txt; exit 1; }
tail -5 $O/quick_tests.txt
