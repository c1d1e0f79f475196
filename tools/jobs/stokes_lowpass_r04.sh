# round 4: the noise cut of the Stokes QFS densities (qfs.Stokes_QFS.NOISE_CUT) — the n_b neighbours of configs[4] with and
# without it, the Stokes tests, the 3-body example at its own size.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for lp in 1 0; do
  echo "IPDE_STOKES_QFS_NOISE_CUT=$lp" >> $O/stokes_nb_density_lowpass.log
  IPDE_STOKES_QFS_NOISE_CUT=$lp timeout -k 10 900 python3 tools/diag_stokes.py 2386,14,4096 2388,14,4096 2390,14,4096 2392,14,4096 2394,14,4096 2396,14,4096 2400,14,4096 2390,14,- > $O/diag_lp_$lp.log 2>&1
  python3 - $O/diag_lp_$lp.log >> $O/stokes_nb_density_lowpass.log <<'PY'
import sys, ast
for l in open(sys.argv[1]):
    if l.startswith("{'nb'"):
        d = ast.literal_eval(l)
        print({k: d[k] for k in ('nb', 'grid', 'grid_err', 'grid_err_in_annuli', 'grid_err_outside_annuli', 'radial_err', 'max_sigma_g', 'gmres_iterations')})
PY
done
cat $O/stokes_nb_density_lowpass.log | cut -c1-330
timeout -k 10 900 python3 -m pytest tests/test_solver_gpu.py tests/test_configs_gpu.py tests/test_dense_gpu.py -q -k "stokes or Stokes or qfs or noise_cut" -s > $O/t_stokes_lp.log 2>&1
tail -5 $O/t_stokes_lp.log
timeout -k 10 300 python3 examples/multi_stokes.py > $O/multi_stokes_lp.log 2>&1
tail -4 $O/multi_stokes_lp.log
