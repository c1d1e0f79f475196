# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1000 python3 tools/diag_stokes.py 2386,14,4096 2388,14,4096 2390,14,4096 2390,14,4100 2394,14,4096 2396,14,4096 2390,14,4160 2> /dev/null | cut -c1-140 | tee $O/stokes_nb_neighbours.log
