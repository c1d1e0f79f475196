# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/legacy_stream
mkdir -p $O
export IPDE_PROFILE_SOLVES=60
for i in 1 2 3; do
IPDE_CTX_OWN_STREAM=1 timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve" | sed 's/^/own    /' | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | grep "warm solve" | sed 's/^/legacy /' | tee -a $O/ab.txt
done
export IPDE_PROFILE_SOLVES=20
for i in 1 2; do
IPDE_CTX_OWN_STREAM=1 timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed 's/^/own    /' | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | grep "warm" | sed 's/^/legacy /' | tee -a $O/ab.txt
done
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/gputest.txt
