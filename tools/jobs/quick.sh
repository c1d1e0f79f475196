# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
export IPDE_PROFILE_SOLVES=30
for q in 4 8 16 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed "s/^/hwq=$q /" | tee -a $O/ab_hwqueues.txt
done
export IPDE_PROFILE_SOLVES=60
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm" | sed "s/^/poisson hwq=$q /" | tee -a $O/ab_hwqueues.txt
done
