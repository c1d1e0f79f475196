# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py tests/test_solver_gpu.py -m gpu -x -q -k "far_expansion or patches or resident" 2>&1 | tail -5
export IPDE_PROFILE_SOLVES=60
for i in 1 2; do
IPDE_FAR_EXPANSION=0 timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve\|error" | cut -c1-60 | sed 's/^/direct          /' | tee -a $O/ab_solve.txt
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve\|error" | cut -c1-60 | sed 's/^/far             /' | tee -a $O/ab_solve.txt
IPDE_PROFILE_RESIDENT=1 timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve" | sed 's/^/far resident    /' | tee -a $O/ab_solve.txt
done
