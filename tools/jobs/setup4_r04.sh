# configs[4] set-up (3-body Stokes, n_b = 2400, 4096^2) with the library's own factorisation at 19 200 rows: wall
# time of the second construction in a process, then the same under a kernel trace (which kernels the GPU side is).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 300 python3 tools/profile_stokes_setup.py > $O/setup4_own.log 2>&1
tail -2 $O/setup4_own.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/setup4_trace -- python3 tools/profile_stokes_setup.py > $O/setup4_trace.log 2>&1
tail -2 $O/setup4_trace.log
cp $(ls -t $(find $O/setup4_trace -name "*kernel_stats.csv") | head -1) $O/setup4_own_kernel_stats.csv
rm -rf $O/setup4_trace
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r04/setup4_own_kernel_stats.csv')))[:14]:
    print('%-70s %6s calls  total %8.1f ms  avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
