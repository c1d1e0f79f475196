# Local step after tools/jobs/evidence.sh: copy the summaries into profiles/ (run from the repo root).
set -e
# (gpurun MERGES a call's files into gpurun_out/: an earlier run's traces stay beside the new ones, so take the newest)
O=gpurun_out/r02/final
P=profiles
cp $O/bench.json $P/r02_bench_n1_final.json
cp $O/bench_under_rocprof.json $P/r02_bench_n1_under_rocprof.json
cp $O/kernels.json $P/r02_kernels.json
cp $(ls -t $(find $O/bench_trace -name "*kernel_stats.csv") | head -1) $P/r02_bench_kernel_stats.csv
cp $(ls -t $(find $O/solve_trace -name "*kernel_stats.csv") | head -1) $P/r02_poisson_2048_solve_kernel_stats.csv
cp $(ls -t $(find $O/stokes_trace -name "*kernel_stats.csv") | head -1) $P/r02_stokes_3body_solve_kernel_stats.csv
cp $(ls -t $(find $O/fft_trace -name "*kernel_stats.csv") | head -1) $P/r02_fft2d_kernel_stats.csv
cp $(ls -t $(find $O/lu_trace -name "*kernel_stats.csv") | head -1) $P/r02_lu_factor_kernel_stats.csv
cp $O/poisson_solve_budget.json $P/r02_poisson_2048_solve_budget.json
cp $O/stokes_solve_budget.json $P/r02_stokes_3body_solve_budget.json
cp $O/lu_subst_probe.txt $P/r02_lu_subst_probe.txt
cp $O/lu_panel_probe.txt $P/r02_lu_panel_probe.txt
cat $O/poisson_warm_unprofiled.txt $O/stokes_warm_unprofiled.txt > $P/r02_warm_solves_unprofiled.txt
python3 tools/collect_traffic.py $O/pmc_fetch $O/pmc_write laplace_patch_kernel r02
cp $(ls -t $(find $O/pmc_fetch -name "*counter_collection.csv") | head -1) $P/r02_pmc_fetch_laplace.csv
cp $(ls -t $(find $O/pmc_write -name "*counter_collection.csv") | head -1) $P/r02_pmc_write_laplace.csv
cp $O/pmc_sq_laplace_a.json $P/r02_pmc_sq_laplace_patch_a.json
cp $O/pmc_sq_laplace_b.json $P/r02_pmc_sq_laplace_patch_b.json
cp $O/power_probe.txt $P/r02_power_clock_probe.txt
cp $O/patches_ab.txt $P/r02_patches_ab.txt
cp $O/cold_process.jsonl $P/r02_poisson_2048_cold_process.jsonl
echo collected
