# round-3 counter passes for the far-field kernels (Laplace and stokeslet forms): SQ counters and HBM
# traffic, counters in runs of their own; kernel-trace stats of the same script beside them
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/pmc_far
mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq_a -- python3 tools/ab_far_expansion.py > $O/sq_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq_b -- python3 tools/ab_far_expansion.py > $O/sq_b.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/ab_far_expansion.py > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 tools/ab_far_expansion.py > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/ab_far_expansion.py > $O/trace.log 2>&1
for n in sq_a sq_b fetch write; do python3 tools/summarize_pmc.py $O/$n $O/$n.json _far_ > /dev/null; done
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*.csv" -size +5M -delete
grep "_far_" $O/kernel_stats.csv | cut -c1-60,200-400
