# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
timeout -k 10 300 python3 tools/ab_far_expansion.py 2>&1 | tee $O/ab_far_2048_b.txt
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py -m gpu -x -q -k "far_expansion" 2>&1 | tail -5
export IPDE_PROFILE_SOLVES=20
for i in 1 2; do
IPDE_FAR_EXPANSION=0 timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm\|err" | sed 's/^/direct /' | tee -a $O/ab_stokes_solve.txt
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm\|err" | sed 's/^/far    /' | tee -a $O/ab_stokes_solve.txt
done
