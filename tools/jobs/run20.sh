set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r02/gputest_i.log 2>&1 || (tail -40 gpurun_out/r02/gputest_i.log; exit 1)
tail -3 gpurun_out/r02/gputest_i.log
python - <<'PY'
import time, torch
from ipde_amd.spectral import get_plan
from ipde_amd.device import get_context
ctx = get_context()
n = 2048
f = torch.randn(n, n, dtype=torch.float64, device="cuda"); f -= f.mean()
g = torch.randn(n, n, dtype=torch.float64, device="cuda"); g -= g.mean()
plan = get_plan(n, n, 3.0 / n, 3.0 / n)
for opt in (1, 0):
    ctx.set_option("fft2d", opt)
    for _ in range(5): plan.stokes_solve(f, g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): plan.stokes_solve(f, g)
    torch.cuda.synchronize()
    print("stokes grid solve 2048^2, fft2d =", opt, "%.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
ctx.set_option("fft2d", 1)
PY
echo done
