#!/usr/bin/env python3
"""BASELINE configs[2]: full interior Poisson solve on a 2048^2 grid, 4096-node star
boundary, M = 20 — timed as bracketed in the reference's examples/poisson_for_paper.py:60-92
(setup / inhomogeneous solve / homogeneous form / homogeneous apply)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import interior_poisson  # noqa: E402

if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ng = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    T = {}
    t0 = time.perf_counter()
    err, scale, solver, ue, T = interior_poisson.run(nb=nb, M=20, Ns=[ng, ng], timings=T, solver_tol=1e-12)
    T['total_s'] = time.perf_counter() - t0
    T['max_err'] = err
    T['rel_err'] = err / scale
    # second solve on the warmed-up solver (plans, tables, resident targets)
    import numpy as np
    from ipde_amd.embedded_function import EmbeddedFunction
    f = EmbeddedFunction(solver.ebdyc)
    f.define_via_function(lambda x, y: (2.0 * np.cos(x) + 3.0 * np.cos(x) * np.sin(x) - np.cos(x) ** 3)
                          * np.exp(np.sin(x)) * np.sin(y))
    t0 = time.perf_counter()
    solver(f, tol=1e-12, maxiter=100, restart=20)
    T['inhomogeneous_solve_warm_s'] = time.perf_counter() - t0
    print(json.dumps({k: v for k, v in T.items() if isinstance(v, (int, float, str, list, tuple))}))
