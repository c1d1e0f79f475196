# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_annular_gpu.py tests/test_solver_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -1
timeout -k 10 300 python3 tools/ab_gmres_lookahead.py 2>&1 | grep -v Warn | tail -3
