# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_layer_gpu.py tests/test_solver_gpu.py -m gpu -x -q -k "modhelm_far or modified_helmholtz_solver_far" 2>&1 | tail -4
rm -rf $O/k_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_trace -- python3 tools/ab_far_expansion.py > $O/k_trace.log 2>&1
grep -h "modhelm.*far" $(find $O/k_trace -name "*kernel_stats.csv") | awk -F'",' '{print substr($1,1,72), $2}' | cut -c1-150
grep "modhelm" $O/k_trace.log | head -9
find $O/k_trace -name "*.csv" -size +5M -delete
export IPDE_PROFILE_SOLVES=20
timeout -k 10 400 python3 tools/profile_modhelm_solve.py 2>&1 | grep "warm" | sed 's/^/configs[3] /'
