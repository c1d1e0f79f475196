set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/solve_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/solve_trace -- python3 tools/profile_solve.py > gpurun_out/r02/solve_trace.log 2>&1
grep "warm solve" gpurun_out/r02/solve_trace.log
ms=$(grep "warm solve" gpurun_out/r02/solve_trace.log | awk '{print $3}')
python3 tools/analyze_trace.py gpurun_out/r02/solve_trace $ms 10 > gpurun_out/r02/solve_budget.json
cat gpurun_out/r02/solve_budget.json
python3 tools/profile_solve.py
echo done
