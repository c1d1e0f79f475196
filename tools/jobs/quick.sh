# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/far
mkdir -p $O
rm -rf $O/k_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_trace -- python3 tools/ab_far_expansion.py > $O/k_trace.log 2>&1
grep -h "_far_" $(find $O/k_trace -name "*kernel_stats.csv") | cut -c1-75,180-330
find $O/k_trace -name "*.csv" -size +5M -delete
