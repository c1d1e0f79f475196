set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0 1 0; do
echo "IPDE_OWN_LU=$v"
IPDE_OWN_LU=$v timeout -k 10 300 python3 tools/hostprof_setup.py 2>&1 | grep "setup_s" | cut -c1-120
done
