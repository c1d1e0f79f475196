# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_annular_gpu.py tests/test_solver_gpu.py -m gpu -x -q -k "stokes or Stokes" 2>&1 | tail -3
export IPDE_PROFILE_SOLVES=30
for i in 1 2 3; do
IPDE_HIP_OPTIONS=annular_grouped=1 timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed 's/^/grouped=1 /' | tee -a $O/ab_stokes_merged.txt
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed 's/^/grouped=2 /' | tee -a $O/ab_stokes_merged.txt
done
