#!/bin/bash
# Rebuild ipde_amd/kernel_cache/rocfft_gfx950.db on an MI355X: run the workloads whose FFT
# lengths should start without run-time compilation, with rocFFT's cache pointed at a fresh
# file, then copy that file into the package.
#   gpurun -- 'bash tools/build_kernel_cache.sh'  ->  gpurun_out/rocfft_gfx950.db ;
#   gzip -9 -c gpurun_out/rocfft_gfx950.db > ipde_amd/kernel_cache/rocfft_gfx950.db.gz
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export ROCFFT_RTC_CACHE_PATH="$PWD/gpurun_out/rocfft_gfx950.db"
rm -f "$ROCFFT_RTC_CACHE_PATH"
# start from the shipped cache (KEEP=1) or from scratch
if [ "${KEEP:-1}" = "1" ] && [ -f ipde_amd/kernel_cache/rocfft_gfx950.db.gz ]; then
    gzip -dc ipde_amd/kernel_cache/rocfft_gfx950.db.gz > "$ROCFFT_RTC_CACHE_PATH"
fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x
timeout -k 10 200 python bench.py --steps 3 --warmup 1 > /dev/null
timeout -k 10 200 python examples/multi_stokes.py > /dev/null
timeout -k 10 200 python examples/multi_modified_helmholtz.py > /dev/null
timeout -k 10 200 python examples/interior_modified_helmholtz.py > /dev/null
# the Ewald-split grid evaluator's padded transforms at the BASELINE grid sizes
timeout -k 10 200 python tools/run_sharded_solve.py --problem poisson --nb 4096 --M 20 --ng 2048 --grid-backend ewald > /dev/null
timeout -k 10 300 python tools/run_sharded_solve.py --problem modhelm --nb 8192 --M 20 --ng 4096 --k 10 --grid-backend ewald > /dev/null
timeout -k 10 300 python examples/multi_stokes.py --nb 800 --M 14 > /dev/null
timeout -k 10 300 python examples/multi_stokes.py --nb 3100 --M 14 > /dev/null    # config-5 scale (5312^2 grid)
# plans of the larger BASELINE configurations (4096^2 grid, 8192-node boundaries) and of
# the common power-of-two boundary sizes
timeout -k 10 300 python - <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from ipde_amd.device import get_context
from ipde_amd.spectral import get_plan
ctx = get_context(0)
for n in (512, 1024, 4096):
    get_plan(n, n, 1.0 / n, 1.0 / n, ctx)
for n in (512, 1024, 2048, 4096, 8192, 16384):
    for batch in (16, 20, 24):
        for b in (batch, batch - 1, batch - 2):
            ctx.lib.ipde_fft1_prepare(ctx.handle, b, n)
    ctx.lib.ipde_fft1_prepare(ctx.handle, 1, n)          # (the noise cut of a Stokes QFS density: one row)
    x = torch.ones((4, n), dtype=torch.complex128, device='cuda')
    torch.fft.fft(x, dim=1); torch.fft.ifft(x, dim=1)
    torch.fft.rfft(x.real, dim=1)
torch.cuda.synchronize()
PY
ls -la "$ROCFFT_RTC_CACHE_PATH"
