set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/cold_solve.py 2>/dev/null | tail -1 | cut -c1-330
timeout -k 10 300 python3 tools/cold_solve.py 2>/dev/null | tail -1 | cut -c1-330
timeout -k 10 300 python3 tools/cold_solve.py 2>/dev/null | tail -1 | cut -c1-330
timeout -k 10 300 python3 tools/hostprof_first_solve.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/first_solve_prof.txt
head -20 gpurun_out/first_solve_prof.txt | tail -16
timeout -k 10 300 python3 examples/multi_stokes.py --nb 800 --M 14 --warm 2>&1 | tail -3 | cut -c1-300
