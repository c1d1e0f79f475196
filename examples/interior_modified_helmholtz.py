#!/usr/bin/env python3
"""Interior modified-Helmholtz problem (k^2 - Lap) u = f on a star domain — the flow of
the reference's examples/interior_modified_helmholtz.py / multi_modified_helmholtz_*.py
(manufactured solution exp(sin kx) sin ky, inhomogeneous solve, double-layer boundary
correction through QFS) on the MI355X stack.

    python examples/interior_modified_helmholtz.py [--nb 800] [--M 20] [--k 10]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ipde_amd.ebdy_collection import EmbeddedBoundaryCollection  # noqa: E402
from ipde_amd.embedded_boundary import EmbeddedBoundary  # noqa: E402
from ipde_amd.embedded_function import EmbeddedFunction, BoundaryFunction  # noqa: E402
from ipde_amd.heavisides import SlepianMollifier  # noqa: E402
from ipde_amd.layer_potentials import Modified_Helmholtz_Layer_Apply, DeviceTargets, ShardedTargets  # noqa: E402
from ipde_amd.pybie2d_compat import star, Global_Smooth_Boundary as GSB  # noqa: E402
from homogeneous_correction import HomogeneousCorrection  # noqa: E402  (examples/homogeneous_correction.py)
from ipde_amd.solvers.multi_boundary.modified_helmholtz import ModifiedHelmholtzSolver  # noqa: E402


def run(nb=800, M=20, helmholtz_k=10.0, solver_tol=1e-14, Ns=None, verbose=False, timings=None,
        grid_backend=None, sharded_result=False, resident=False, correction_far=True):
    T = {} if timings is None else timings
    t0 = time.perf_counter()
    MOL = SlepianMollifier(1.5 * M)
    bdy = GSB(c=star(nb, a=0.2, f=5))
    bh = bdy.dt * bdy.speed.min()
    ebdy = EmbeddedBoundary(bdy, True, M, bh, pad_zone=0, heaviside=MOL.step, qfs_tolerance=1e-14)
    ebdyc = EmbeddedBoundaryCollection([ebdy, ])
    grid = ebdyc.generate_grid(bh, Ns=Ns)
    # set-up bracket as in the reference's examples/poisson_for_paper.py:60-64: geometry, grid, solver
    solver = ModifiedHelmholtzSolver(ebdyc, k=helmholtz_k, grid_backend=grid_backend)
    from ipde_amd.sharding import is_distributed
    if not sharded_result and not is_distributed():
        # grid_and_radial_pts resident for the correction stage: its patch plan is cut by a background thread
        ebdyc.resident_grid_and_radial_pts(far=correction_far)
    T['setup_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    kk = 2 * np.pi / 3
    solution_func = lambda x, y: np.exp(np.sin(kk * x)) * np.sin(kk * y)
    force_func = lambda x, y: helmholtz_k ** 2 * solution_func(x, y) \
        - kk ** 2 * np.exp(np.sin(kk * x)) * np.sin(kk * y) * (np.cos(kk * x) ** 2 - np.sin(kk * x) - 1.0)
    f = EmbeddedFunction(ebdyc)
    f.define_via_function(force_func)
    ua = EmbeddedFunction(ebdyc)
    ua.define_via_function(solution_func)
    bc = BoundaryFunction(ebdyc)
    bc.define_via_function(solution_func)
    T['problem_definition_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    # sharded_result (torch.distributed): the answer stays sharded through the solve and the correction
    # below, `ue.owned` marks the entries complete on this rank (ipde_amd/solvers/multi_boundary/scalar.py)
    if resident:        # right-hand side and answer stay in HBM (hostio.DeviceFunction)
        from ipde_amd.hostio import DeviceFunction
        T['f'] = f = DeviceFunction.from_host(f)
    ue = solver(f, tol=solver_tol, verbose=verbose, maxiter=100, restart=20, sharded_result=sharded_result)
    T['inhomogeneous_solve_s'] = time.perf_counter() - t0
    # homogeneous correction with a double layer on the boundary (interior: D - I/2)
    t0 = time.perf_counter()
    import torch
    from ipde_amd import dense_forms as df      # the dense matrices are assembled on the GPU
    dev = torch.device('cuda', torch.cuda.current_device())
    eye = lambda n: torch.eye(n, dtype=torch.float64, device=dev)
    K = lambda src, _: df.modhelm_singular_form(src, dev, helmholtz_k, ifdipole=True) - 0.5 * eye(src.N)
    Naive_SLP = lambda src, trg: df.modhelm_form(src, trg, dev, helmholtz_k, ifcharge=True)
    correction = HomogeneousCorrection(
        solver, bdy, ebdy, bc, K, Naive_SLP,
        lambda src, trg, ch: Modified_Helmholtz_Layer_Apply(src, trg, k=helmholtz_k, charge=ch),
        owned=getattr(ue, 'owned', None), far=correction_far)
    T['homogeneous_form_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ue = correction(ue)
    T['homogeneous_apply_s'] = time.perf_counter() - t0
    T['correction'] = correction
    if resident:
        ue = ue.to_host()
    err = np.abs(np.asarray(ue) - np.asarray(ua))
    if getattr(ue, 'owned', None) is not None:      # a sharded answer: this rank's entries, then the max over the ranks
        from ipde_amd.sharding import global_max
        err = np.array([global_max(err[ue.owned].max())])
    T['dof'] = int(ebdyc.dof)
    T['grid'] = list(grid.shape)
    T['gmres_iterations'] = solver.iteration_counts
    return float(err.max()), float(np.abs(np.asarray(ua)).max()), solver, ue, T


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=800)
    ap.add_argument("--M", type=int, default=20)
    ap.add_argument("--k", type=float, default=10.0)
    a = ap.parse_args()
    T = {}
    err, scale, *_ = run(a.nb, a.M, a.k, verbose=True, timings=T)
    print('Error: {:0.2e}'.format(err), ' (|u|max %.3f)' % scale)
    print(T)
