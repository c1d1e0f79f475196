# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/scratch; mkdir -p $O
rm -rf $O/c4_trace
IPDE_PROFILE_STOP_AFTER_WARM=1 IPDE_PROFILE_SOLVES=5 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/c4_trace -- python3 tools/profile_stokes_solve.py 2400 4096 > $O/c4_trace.log 2>&1
ms=$(grep "warm stokes" $O/c4_trace.log | awk '{print $4}')
python3 tools/analyze_trace.py $O/c4_trace $ms 5 > $O/config4_budget.json
find $O/c4_trace -name "*.csv" -size +20M -delete
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r03/scratch/config4_budget.json"))
print(d["warm_solve_ms"], d["gpu_busy_ms_per_solve"], d["launches_per_solve"])
for k,v in list(d["kernels_ms_per_solve"].items())[:24]:
    if "stokes" in k or "copy" in k: print("%8.3f %6.1f  %s"%(v["ms"],v["launches"],k[:90]))
PY
