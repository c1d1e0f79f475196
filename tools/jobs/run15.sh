set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02/gputest_f.log 2>&1 || true
tail -5 gpurun_out/r02/gputest_f.log
python - > gpurun_out/r02/stokes_warm.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "examples")
import multi_stokes
from ipde_amd.solvers.multi_boundary.vector import VectorSolver
for fast in (True, False):
    VectorSolver.USE_FAST_INTERP = fast
    for kw in (dict(nb=800, M=14), dict(nb=1500, M=14, ng=2048), dict(nb=2400, M=14, ng=4096)):
        ue, ve, pe, scale, T = multi_stokes.run(warm=True, **kw)
        print("fast_interp", fast, kw, "err %.2e %.2e p %.2e" % (ue, ve, pe), "grid", T["grid"],
              "first %.3f s warm %.4f s" % (T["inhomogeneous_solve_s"], T["warm_inhomogeneous_solve_s"]), flush=True)
PY
cat gpurun_out/r02/stokes_warm.log | grep fast_interp
echo done
