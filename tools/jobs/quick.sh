# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
IPDE_PG_PROFILE=1 timeout -k 10 600 python3 tools/ab_gmres_persistent.py 2>&1 | grep -v Warning | tee $O/ab_gmres_persistent.txt
