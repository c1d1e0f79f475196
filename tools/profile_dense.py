#!/usr/bin/env python3
"""Profiling driver: the dense layer-potential kernels at BASELINE configs[1]
(2048^2 grid x 4096 nodes), device resident.  Run directly after `rocprofv3 ... --`.

    python3 tools/profile_dense.py [reps] [families: laplace,patches,modhelm,stokes,stokes_dlp]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    fam = (sys.argv[2] if len(sys.argv) > 2 else "laplace,modhelm,stokes").split(",")
    import torch
    from util import Curve, grid_targets
    from ipde_amd import layer_potentials as lp
    c = Curve(4096, a=0.2, f=5)
    trg, h = grid_targets(c, 2048)
    dt = lp.DeviceTargets(trg)
    rng = np.random.default_rng(0)
    sig, tau = rng.standard_normal(c.N), rng.standard_normal(c.N)
    f2 = rng.standard_normal((2, c.N))
    dtp = lp.DeviceTargets(trg, plan=True) if "patches" in fam else None
    for _ in range(reps):
        if "patches" in fam:         # the 4 x 4 patch kernel (laplace_patch_kernel)
            lp.Laplace_Layer_Apply(c, dtp, charge=sig)
            lp.Laplace_Layer_Apply(c, dtp, dipstr=tau)
        if "laplace" in fam:
            lp.Laplace_Layer_Apply(c, dt, charge=sig)
            lp.Laplace_Layer_Apply(c, dt, dipstr=tau)
        if "modhelm" in fam:
            lp.Modified_Helmholtz_Layer_Apply(c, dt, k=10.0, charge=sig)
            lp.Modified_Helmholtz_Layer_Apply(c, dt, k=10.0, dipstr=tau)
        if "stokes" in fam:
            lp.Stokes_Layer_Apply(c, dt, forces=f2)
        if "stokes_dlp" in fam:
            lp.Stokes_Layer_Apply(c, dt, dipstr=f2)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
