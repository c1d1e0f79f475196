set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0 1; do
IPDE_OWN_LU=$v python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fft 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); f=b['full_poisson_solve']; print('own_lu=$v', {k:f[k] for k in ('setup_s','first_inhomogeneous_solve_s','homogeneous_correction_s','end_to_end_s','warm_inhomogeneous_solve_ms')})"
done
