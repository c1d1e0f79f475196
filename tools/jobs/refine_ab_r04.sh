# Is the refinement step of the Stokes QFS solves still needed once the noise cut is in?  Errors and warm solves with
# IPDE_STOKES_QFS_REFINE_STEPS = 1 (default) and 0: the example's own size, configs[4] (n_b = 2390 and 2400), and
# config-5 scale (n_b = 3100), where round 1 measured a 30x loss without it.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for st in 1 0; do
  echo "IPDE_STOKES_QFS_REFINE_STEPS=$st" >> $O/stokes_refine_ab.txt
  IPDE_STOKES_QFS_REFINE_STEPS=$st timeout -k 10 900 python3 tools/diag_stokes.py 800,14,- 1600,14,- 2390,14,- 2400,14,4096 3100,14,- > $O/diag_refine_$st.log 2>&1
  python3 - $O/diag_refine_$st.log >> $O/stokes_refine_ab.txt <<'PY'
import sys, ast
for l in open(sys.argv[1]):
    if l.startswith("{'nb'"):
        d = ast.literal_eval(l)
        print({k: d[k] for k in ('nb', 'grid', 'grid_err', 'radial_err', 'max_sigma_g', 'gmres_iterations')})
PY
  export IPDE_PROFILE_STOP_AFTER_WARM=1
  IPDE_STOKES_QFS_REFINE_STEPS=$st timeout -k 10 200 python3 tools/profile_stokes_solve.py 2>/dev/null | grep "warm stokes" | sed 's/^/example (nb 800): /' >> $O/stokes_refine_ab.txt
  IPDE_STOKES_QFS_REFINE_STEPS=$st timeout -k 10 300 python3 tools/profile_stokes_solve.py 2390 2>/dev/null | grep "warm stokes" | sed 's/^/configs[4] (nb 2390): /' >> $O/stokes_refine_ab.txt
  unset IPDE_PROFILE_STOP_AFTER_WARM
done
cut -c1-420 $O/stokes_refine_ab.txt
