# round-3 solve budgets: kernel traces of ten warm solves (Poisson 2048^2, 3-body Stokes), GPU busy share
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/budget
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solve_trace -- python3 tools/profile_solve.py > $O/solve_trace.log 2>&1
ms=$(grep "warm solve" $O/solve_trace.log | awk '{print $3}')
python3 tools/analyze_trace.py $O/solve_trace $ms 10 > $O/poisson_solve_budget.json
IPDE_PROFILE_STOP_AFTER_WARM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stokes_trace -- python3 tools/profile_stokes_solve.py > $O/stokes_trace.log 2>&1
ms=$(grep "warm stokes" $O/stokes_trace.log | awk '{print $4}')
python3 tools/analyze_trace.py $O/stokes_trace $ms 10 > $O/stokes_solve_budget.json
timeout -k 10 300 python3 tools/profile_solve.py > $O/warm_unprofiled.txt 2>&1
timeout -k 10 300 python3 tools/profile_stokes_solve.py >> $O/warm_unprofiled.txt 2>&1
grep -i "warm" $O/warm_unprofiled.txt
head -c 1500 $O/poisson_solve_budget.json
