#!/usr/bin/env python3
"""A/B of the examples' homogeneous correction stage at BASELINE configs[2] (interior Poisson, 2048^2 grid,
4096 nodes): refinement step of the second-kind boundary solve on / off, far-field forms on / off; errors
against the manufactured solution and warm times of the stage (host containers and resident)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    import torch
    import interior_poisson
    from ipde_amd import hostio
    from ipde_amd.embedded_function import EmbeddedFunction
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ng = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    for refine in ("1", "0"):
        for far in (True, False):
            os.environ["IPDE_CORRECTION_REFINE"] = refine
            err, scale, solver, ue, T = interior_poisson.run(nb=nb, M=20, Ns=[ng, ng], solver_tol=1e-12,
                                                             correction_far=far)
            corr = T["correction"]
            f = EmbeddedFunction(solver.ebdyc)
            f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
            out = {"refine": int(refine), "far": far, "rel_err": err / scale}
            for resident in (False, True):
                g = hostio.DeviceFunction.from_host(f) if resident else f
                u = corr(solver(g, tol=1e-12, maxiter=100, restart=20))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    u = corr(u)
                torch.cuda.synchronize()
                out["apply_resident_ms" if resident else "apply_host_ms"] = 1e2 * (time.perf_counter() - t0)
            print(json.dumps(out), flush=True)
            del solver, ue, corr, T, u, g
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
