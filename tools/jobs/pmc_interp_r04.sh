# round-4 counter passes on the band interpolation's kernels (2048^2 x 4096 and 4096^2 x 8192 points): SQ counters in
# runs of their own.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04/pmc
mkdir -p $O
for n in 2048 4096; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/interp_sq_a_$n -- python3 tools/profile_interp.py $n $((2*n)) 1 > $O/interp_sq_a_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/interp_sq_b_$n -- python3 tools/profile_interp.py $n $((2*n)) 1 > $O/interp_sq_b_$n.log 2>&1
  python3 tools/summarize_pmc.py $O/interp_sq_a_$n $O/interp_sq_a_$n.json band_ > /dev/null
  python3 tools/summarize_pmc.py $O/interp_sq_b_$n $O/interp_sq_b_$n.json band_ > /dev/null
  rm -rf $O/interp_sq_a_$n $O/interp_sq_b_$n
  tail -2 $O/interp_sq_b_$n.log
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/pmc/interp_sq_*.json')):
    d = json.load(open(f))
    print(f)
    for k, v in d.items():
        print('  ', k[:70], {a: round(b) for a, b in v.items()})
PY
