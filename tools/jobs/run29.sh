set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/pmc_fft
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r02/pmc_fft -- python3 tools/profile_fft.py 2048 4 > gpurun_out/r02/pmc_fft.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/r02/pmc_fft gpurun_out/r02/pmc_fft.json row_r2c row_c2r col_kernel
echo done
