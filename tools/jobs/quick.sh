# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
rm -rf $O/solve_trace
IPDE_PROFILE_RESIDENT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/solve_trace -- python3 tools/profile_solve.py > $O/solve_trace.log 2>&1
ms=$(grep "warm solve" $O/solve_trace.log | awk '{print $3}')
python3 tools/analyze_trace.py $O/solve_trace $ms 10 > $O/poisson_resident_budget.json
find $O/solve_trace -name "*.csv" -size +20M -delete
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r03/scratch/poisson_resident_budget.json"))
print(d["warm_solve_ms"], d["gpu_busy_ms_per_solve"], d["launches_per_solve"])
for k,v in list(d["kernels_ms_per_solve"].items())[:14]: print("%8.3f %6.1f  %s"%(v["ms"],v["launches"],k[:90]))
for k,v in list(d["idle_ms_per_solve_by_neighbours"].items())[:14]: print("idle %8.3f %5.1f  %s"%(v["ms"],v["count"],k))
PY
