# Regenerates the measurements the docs and profiles/ quote (one MI355X):  bash tools/jobs/evidence.sh
# Everything lands under gpurun_out/r02/final/; tools/jobs/evidence_collect.sh copies the
# summaries into profiles/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02/final
rm -rf $O
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
# (the bench line first, as the driver takes it: on a box that has not just run 95 s of tests)
python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 1000 python3 -m pytest tests -m gpu -q 2>&1 | grep -v '^  File\|^Extension' | tail -4 > $O/gputest.txt
python tools/bench_kernels.py > $O/kernels.json 2> $O/kernels.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-solve > $O/bench_under_rocprof.json 2> $O/bench_trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-solve --no-fft > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solve_trace -- python3 tools/profile_solve.py > $O/solve_trace.log 2>&1
ms=$(grep "warm solve" $O/solve_trace.log | awk '{print $3}')
python3 tools/analyze_trace.py $O/solve_trace $ms 10 > $O/poisson_solve_budget.json
python3 tools/profile_solve.py 2>/dev/null | tail -1 > $O/poisson_warm_unprofiled.txt
IPDE_PROFILE_STOP_AFTER_WARM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stokes_trace -- python3 tools/profile_stokes_solve.py > $O/stokes_trace.log 2>&1
ms=$(grep "warm stokes" $O/stokes_trace.log | awk '{print $4}')
python3 tools/analyze_trace.py $O/stokes_trace $ms 10 > $O/stokes_solve_budget.json
python3 tools/profile_stokes_solve.py 2>/dev/null | grep "warm stokes" > $O/stokes_warm_unprofiled.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fft_trace -- python3 tools/profile_fft.py 2048 20 > $O/fft_trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lu_trace -- python3 tools/profile_lu.py 4096 > $O/lu_trace.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPDE_LU_STAMPS tools/lu_persist_probe.hip -o /tmp/lu_probe 2>/dev/null
timeout -k 10 60 /tmp/lu_probe 4096 1 > $O/lu_subst_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPDE_LU_STAMPS tools/lu_panel_probe.hip -o /tmp/lu_panel_probe 2>/dev/null
timeout -k 10 60 /tmp/lu_panel_probe 4096 0 > $O/lu_panel_probe.txt
timeout -k 10 60 /tmp/lu_panel_probe 4096 60 >> $O/lu_panel_probe.txt
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq_a -- python3 tools/profile_dense.py 2 patches,laplace > $O/pmc_sq_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq_b -- python3 tools/profile_dense.py 2 patches,laplace > $O/pmc_sq_b.log 2>&1
python3 tools/summarize_pmc.py $O/pmc_sq_a $O/pmc_sq_laplace_a.json laplace_ > /dev/null
python3 tools/summarize_pmc.py $O/pmc_sq_b $O/pmc_sq_laplace_b.json laplace_ > /dev/null
timeout -k 10 120 python3 tools/power_probe.py 2>&1 | grep -v amdgpu.ids > $O/power_probe.txt
timeout -k 10 300 python3 tools/ab_patches.py 2>&1 | grep -v amdgpu.ids > $O/patches_ab.txt
for i in 1 2 3; do timeout -k 10 200 python3 tools/cold_solve.py 2>/dev/null | tail -1 >> $O/cold_process.jsonl; done
find $O -name "*_kernel_trace.csv" -size +20M -delete
cat $O/smoke.txt $O/gputest.txt $O/poisson_warm_unprofiled.txt $O/stokes_warm_unprofiled.txt
python3 -c "
import json; b=json.load(open('$O/bench.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac']); print(json.dumps(b['fft'])[:400]); print(json.dumps(b['full_poisson_solve'])[:600])"
echo evidence done
