set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for g in 0 1 0 1; do
echo "gmres_graphs=$g"
IPDE_HIP_OPTIONS="gmres_graphs=$g" timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1
IPDE_HIP_OPTIONS="gmres_graphs=$g" timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -1
done
