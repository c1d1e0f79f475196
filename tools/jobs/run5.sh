set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_spectral_gpu.py -m gpu -x -q > gpurun_out/r02/gputest_fft.log 2>&1 || (tail -60 gpurun_out/r02/gputest_fft.log; exit 1)
tail -3 gpurun_out/r02/gputest_fft.log
rm -rf gpurun_out/r02/fft2_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/fft2_trace -- python3 tools/profile_fft.py ${1:-2048} 20 > gpurun_out/r02/fft2_trace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r02/fft2_trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "anonymous" in r["Name"]:
        print("%-100s calls=%5s avg_us=%9.1f min=%.1f"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
echo done
