# round-4 extras: unprofiled warm solves (3-body example with and without the noise cut, legacy / own stream A/B,
# configs[4] at n_b = 2390), the far-field forms against the pair-by-pair kernels for every layer, LU timings, and the
# solve budgets under the kernel trace (tools/jobs/budget_r04.sh).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04/extras
mkdir -p $O
export IPDE_PROFILE_STOP_AFTER_WARM=1
{
  echo "3-body example (nb = 800, 1370^2), unprofiled:"
  timeout -k 10 200 python3 tools/profile_stokes_solve.py 2>/dev/null | grep "warm stokes"
  echo "  ... resident containers:"; IPDE_PROFILE_RESIDENT=1 timeout -k 10 200 python3 tools/profile_stokes_solve.py 2>/dev/null | grep "warm stokes"
  echo "  ... without the noise cut (IPDE_STOKES_QFS_NOISE_CUT=0):"; IPDE_STOKES_QFS_NOISE_CUT=0 timeout -k 10 200 python3 tools/profile_stokes_solve.py 2>/dev/null | grep "warm stokes"
  echo "  ... process-wide context on a stream of its own (IPDE_CTX_OWN_STREAM=1):"; IPDE_CTX_OWN_STREAM=1 timeout -k 10 200 python3 tools/profile_stokes_solve.py 2>/dev/null | grep "warm stokes"
  echo "configs[4] (nb = 2390, 4096^2), unprofiled:"
  timeout -k 10 300 python3 tools/profile_stokes_solve.py 2390 2>/dev/null | grep "warm stokes"
  echo "  ... resident containers:"; IPDE_PROFILE_RESIDENT=1 timeout -k 10 300 python3 tools/profile_stokes_solve.py 2390 2>/dev/null | grep "warm stokes"
} > $O/warm_solves_unprofiled.txt 2>&1
cat $O/warm_solves_unprofiled.txt
unset IPDE_PROFILE_STOP_AFTER_WARM
timeout -k 10 300 python3 tools/ab_far_expansion.py > $O/far_double_layer_ab.txt 2>$O/far_ab.err
tail -32 $O/far_double_layer_ab.txt
timeout -k 10 300 python3 tools/lu_time.py 4096 9600 19200 > $O/lu_time.txt 2>/dev/null
cat $O/lu_time.txt
bash tools/jobs/budget_r04.sh > $O/budget.log 2>&1
tail -3 $O/budget.log
