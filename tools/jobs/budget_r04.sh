# round-4 solve budgets: kernel traces of warm solves at BASELINE configs[2], [3], [4] and the example's own 3-body
# size -> per-solve kernel budgets (tools/analyze_trace.py), the inputs of DESIGN §4's replicated / sharded / owner
# table.  Traces stay on the box; the budgets (JSON) come back under gpurun_out/r04/budget/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04/budget
mkdir -p $O
trace() {    # name, marker of the warm line, awk field, command...
  name=$1; marker=$2; field=$3; shift 3
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_trace -- "$@" > $O/${name}_trace.log 2>&1
  ms=$(grep "$marker" $O/${name}_trace.log | tail -1 | awk "{print \$$field}")
  python3 tools/analyze_trace.py $O/${name}_trace $ms ${IPDE_PROFILE_SOLVES:-10} > $O/${name}_solve_budget.json
  rm -rf $O/${name}_trace
  echo "$name: warm $ms ms (under the trace)"
}
export IPDE_PROFILE_RESIDENT=1
trace poisson_2048_resident "warm solve" 3 python3 tools/profile_solve.py
trace config3_resident "warm solve" 3 python3 tools/profile_modhelm_solve.py
unset IPDE_PROFILE_RESIDENT
trace poisson_2048 "warm solve" 3 python3 tools/profile_solve.py
trace config3 "warm solve" 3 python3 tools/profile_modhelm_solve.py
export IPDE_PROFILE_STOP_AFTER_WARM=1
trace config4 "warm stokes" 4 python3 tools/profile_stokes_solve.py 2400 4096
trace stokes_3body "warm stokes" 4 python3 tools/profile_stokes_solve.py
unset IPDE_PROFILE_STOP_AFTER_WARM
for f in $O/*_solve_budget.json; do echo $f; head -c 600 $f; echo; done
