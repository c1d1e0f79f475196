# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
timeout -k 10 400 python3 tools/hostprof_first_solve_stokes.py > $O/first_solve_stokes.txt 2>&1
grep -n "first solve" -A40 $O/first_solve_stokes.txt | cut -c1-150 | head -48
