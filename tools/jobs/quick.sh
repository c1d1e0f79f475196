# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
rm -rf $O/solve_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/solve_trace -- python3 tools/profile_solve.py > $O/solve_trace.log 2>&1
ms=$(grep "warm solve" $O/solve_trace.log | awk '{print $3}')
python3 tools/analyze_trace.py $O/solve_trace $ms 10 > $O/poisson_solve_budget_far.json
find $O/solve_trace -name "*.csv" -size +20M -delete
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r03/far/poisson_solve_budget_far.json"))
print(d["warm_solve_ms"], d["gpu_busy_ms_per_solve"], d["launches_per_solve"])
for k,v in list(d["kernels_ms_per_solve"].items())[:16]: print("%8.3f %5.1f  %s"%(v["ms"],v["launches"],k[:80]))
PY
