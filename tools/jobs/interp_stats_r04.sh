# kernel trace of the band interpolation at 2048^2 x 4096 and 4096^2 x 8192 points
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for n in 2048 4096; do
  timeout -k 10 200 python3 tools/profile_interp.py $n $((2*n)) 1 2>/dev/null | tail -1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/interp_trace_$n -- python3 tools/profile_interp.py $n $((2*n)) 1 > $O/interp_trace_$n.log 2>&1
  cp $(ls -t $(find $O/interp_trace_$n -name "*kernel_stats.csv") | head -1) $O/interp_kernel_stats_b_$n.csv
  rm -rf $O/interp_trace_$n
  python3 - $O/interp_kernel_stats_b_$n.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print('   %-58s %4s calls  avg %8.1f us' % (r['Name'][:58], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
