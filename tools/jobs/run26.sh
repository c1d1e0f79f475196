set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/stokes_trace
export IPDE_PROFILE_STOP_AFTER_WARM=1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/stokes_trace -- python3 tools/profile_stokes_solve.py > gpurun_out/r02/stokes_trace.log 2>&1
grep -i "warm stokes" gpurun_out/r02/stokes_trace.log
ms=$(grep -i "warm" gpurun_out/r02/stokes_trace.log | tail -1 | grep -o "[0-9.]* ms" | awk '{print $1}')
python3 tools/analyze_trace.py gpurun_out/r02/stokes_trace $ms 10 > gpurun_out/r02/stokes_budget.json
python3 - <<'PY'
import json
b=json.load(open("gpurun_out/r02/stokes_budget.json"))
print({k:v for k,v in b.items() if k!="kernels_ms_per_solve"})
for k,v in list(b["kernels_ms_per_solve"].items())[:22]:
    print("%-92s %.3f ms  x%.0f"%(k,v["ms"],v["launches"]))
PY
echo done
