set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -2
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -2
rm -rf gpurun_out/r02/solve_trace2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/solve_trace2 -- python3 tools/profile_solve.py > gpurun_out/r02/solve_trace2.log 2>&1
tail -2 gpurun_out/r02/solve_trace2.log
