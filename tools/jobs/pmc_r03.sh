# round-3 counter passes: HBM traffic and SQ counters of the 2-D FFT kernels, SQ counters of the
# Stokes kernels, HBM traffic of the bench kernel (one MI355X; counters in their own runs)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/pmc
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fft_fetch -- python3 tools/profile_fft.py 2048 6 > $O/fft_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/fft_write -- python3 tools/profile_fft.py 2048 6 > $O/fft_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/fft_sq_a -- python3 tools/profile_fft.py 2048 6 > $O/fft_sq_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/fft_sq_b -- python3 tools/profile_fft.py 2048 6 > $O/fft_sq_b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fft_trace -- python3 tools/profile_fft.py 2048 20 > $O/fft_trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/stokes_sq_a -- python3 tools/profile_dense.py 2 stokes,stokes_dlp > $O/stokes_sq_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/stokes_sq_b -- python3 tools/profile_dense.py 2 stokes,stokes_dlp > $O/stokes_sq_b.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > $O/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > $O/bench_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve > $O/bench_under_rocprof.json 2> $O/bench_trace.err
python3 tools/summarize_pmc.py $O/fft_sq_a $O/fft_sq_a.json row_r2c row_c2r col_kernel > /dev/null
python3 tools/summarize_pmc.py $O/fft_sq_b $O/fft_sq_b.json row_r2c row_c2r col_kernel > /dev/null
python3 tools/summarize_pmc.py $O/stokes_sq_a $O/stokes_sq_a.json stokes_ > /dev/null
python3 tools/summarize_pmc.py $O/stokes_sq_b $O/stokes_sq_b.json stokes_ > /dev/null
find $O -name "*kernel_stats.csv" | head
ls $O
