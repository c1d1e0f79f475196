#!/usr/bin/env python3
"""Profiling driver: the spectral grid path at 2048^2 (BASELINE configs[2] grid), device
resident, nothing else in the process.  Run directly after `rocprofv3 ... --`.

    python3 tools/profile_fft.py [n] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import torch
    from ipde_amd.spectral import get_plan
    torch.manual_seed(0)
    f = torch.randn(n, n, dtype=torch.float64, device="cuda")
    f -= f.mean()
    h = 3.0 / n
    plan = get_plan(n, n, h, h)
    for _ in range(reps):
        plan.poisson_solve(f)
    torch.cuda.synchronize()
    for _ in range(reps):
        plan.dx(f)
    torch.cuda.synchronize()
    for _ in range(reps):
        plan.dy(f)
    torch.cuda.synchronize()
    for _ in range(max(1, reps // 4)):
        plan.stokes_solve(f, f)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
