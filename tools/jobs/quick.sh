# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 tools/hostprof_stokes_setup.py > $O/hostprof_stokes_setup.txt 2>&1
head -45 $O/hostprof_stokes_setup.txt | cut -c1-165
timeout -k 10 600 python3 -m pytest tests/test_solver_gpu.py tests/test_dense_gpu.py -m gpu -x -q -k "stokes or qfs or Stokes" 2>&1 | tail -3
