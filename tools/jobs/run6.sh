set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02/gputest_d.log 2>&1 || true
tail -14 gpurun_out/r02/gputest_d.log
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/r02/pmc_sq_a -- python3 tools/profile_dense.py 2 laplace,modhelm > gpurun_out/r02/pmc_sq_a.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/r02/pmc_sq_a gpurun_out/r02/pmc_sq_a.json rowrun table > /dev/null
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/r02/pmc_sq_b -- python3 tools/profile_dense.py 2 laplace,modhelm > gpurun_out/r02/pmc_sq_b.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/r02/pmc_sq_b gpurun_out/r02/pmc_sq_b.json rowrun table
python bench.py > gpurun_out/r02/bench_c.json 2> gpurun_out/r02/bench_c.err
python3 -c "
import json; b=json.load(open('gpurun_out/r02/bench_c.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac']); print(json.dumps(b['fft'])); print(json.dumps(b['full_poisson_solve']))"
echo done
