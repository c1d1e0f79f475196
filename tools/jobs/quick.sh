# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_solver_gpu.py -m gpu -x -q -k "resident" 2>&1 | tail -3
export IPDE_PROFILE_SOLVES=20
for i in 1 2; do
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed 's/^/3-body 1370^2 host containers /' | tee -a $O/ab_stokes_resident.txt
IPDE_PROFILE_RESIDENT=1 timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed 's/^/3-body 1370^2 resident        /' | tee -a $O/ab_stokes_resident.txt
done
export IPDE_PROFILE_SOLVES=8
timeout -k 10 400 python3 tools/profile_stokes_solve.py 2400 4096 2>&1 | grep "warm" | sed 's/^/configs[4] host containers /' | tee -a $O/ab_stokes_resident.txt
IPDE_PROFILE_RESIDENT=1 timeout -k 10 400 python3 tools/profile_stokes_solve.py 2400 4096 2>&1 | grep "warm" | sed 's/^/configs[4] resident        /' | tee -a $O/ab_stokes_resident.txt
