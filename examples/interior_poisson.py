#!/usr/bin/env python3
"""Interior Poisson problem on a 5-arm star — the same steps as the reference's
examples/interior_poisson.py (construct boundary :41, EmbeddedBoundary :47, grid :49,
bump :51, manufactured solution :59-61, PoissonSolver :80-81, homogeneous correction
:84-92, max error :95-96) on the MI355X stack.

    python examples/interior_poisson.py [--nb 800] [--M 20] [--hard]
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
from ipde_amd.layer_potentials import Laplace_Layer_Apply, DeviceTargets, ShardedTargets  # noqa: E402
from ipde_amd.pybie2d_compat import star, Global_Smooth_Boundary as GSB  # noqa: E402
from homogeneous_correction import HomogeneousCorrection  # noqa: E402  (examples/homogeneous_correction.py)
from ipde_amd.solvers.multi_boundary.poisson import PoissonSolver  # noqa: E402



def _forms():
    """the example's dense matrices (reference :16-20), assembled on the GPU"""
    import torch
    from ipde_amd import dense_forms as df
    dev = torch.device('cuda', torch.cuda.current_device())
    eye = lambda n: torch.eye(n, dtype=torch.float64, device=dev)
    singular_dlp = lambda src, _: df.laplace_singular_form(src, dev, ifdipole=True) - 0.5 * eye(src.N)
    naive_slp = lambda src, trg: df.laplace_form(src, trg, dev, ifcharge=True)
    return singular_dlp, naive_slp


def run(nb=800, M=20, problem='easy', solver_type='spectral', solver_tol=1e-14, grid_upsample=1,
        Ns=None, verbose=False, timings=None, grid_backend=None, h=None, sharded_result=False,
        resident=False, correction_far=True):
    T = {} if timings is None else timings
    t0 = time.perf_counter()
    MOL = SlepianMollifier(1.5 * M)
    bdy = GSB(c=star(nb, a=0.2, f=5))
    bh = bdy.dt * bdy.speed.min() if h is None else h      # h: grid spacing override (BASELINE configs[0])
    ebdy = EmbeddedBoundary(bdy, True, M, bh / grid_upsample, pad_zone=0, heaviside=MOL.step,
                            qfs_tolerance=1e-14, coordinate_tolerance=1e-14)
    ebdyc = EmbeddedBoundaryCollection([ebdy, ])
    grid = ebdyc.generate_grid(bh / grid_upsample, Ns=Ns)
    ebdyc.ready_bump(MOL.bump, (grid.x_bounds[1] - ebdy.radial_width, grid.y_bounds[1] - ebdy.radial_width),
                     ebdyc[0].radial_width)
    # the reference's set-up bracket (examples/poisson_for_paper.py:60-64) ends with the solver's
    # construction: geometry, grid registration, solver.  The manufactured problem is defined after it.
    solver = PoissonSolver(ebdyc, solver_type=solver_type, grid_backend=grid_backend)
    from ipde_amd.sharding import is_distributed
    if not sharded_result and not is_distributed():
        # grid_and_radial_pts resident for the correction stage: its patch plan is cut by a background thread
        ebdyc.resident_grid_and_radial_pts(far=correction_far)
    T['setup_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if problem == 'easy':
        solution_func = lambda x, y: -np.cos(x) * np.exp(np.sin(x)) * np.sin(y)
        force_func = lambda x, y: (2.0 * np.cos(x) + 3.0 * np.cos(x) * np.sin(x) - np.cos(x) ** 3) \
            * np.exp(np.sin(x)) * np.sin(y)
    else:
        k = 10 * np.pi / 3
        solution_func = lambda x, y: np.exp(np.sin(k * x)) * np.sin(k * y)
        force_func = lambda x, y: k ** 2 * np.exp(np.sin(k * x)) * np.sin(k * y) \
            * (np.cos(k * x) ** 2 - np.sin(k * x) - 1.0)
    f = EmbeddedFunction(ebdyc)
    f.define_via_function(force_func)
    ua = EmbeddedFunction(ebdyc)
    ua.define_via_function(solution_func)
    bc = BoundaryFunction(ebdyc)
    bc.define_via_function(solution_func)
    T['problem_definition_s'] = time.perf_counter() - t0     # f, u_exact, boundary data on 2.2 M points (numpy)

    t0 = time.perf_counter()
    # sharded_result (torch.distributed): the answer stays sharded through the solve and the correction
    # below, `ue.owned` marks the entries complete on this rank (ipde_amd/solvers/multi_boundary/scalar.py)
    if resident:        # right-hand side and answer stay in HBM (hostio.DeviceFunction)
        from ipde_amd.hostio import DeviceFunction
        T['f'] = f = DeviceFunction.from_host(f)
    ue = solver(f, tol=solver_tol, verbose=verbose, maxiter=100, restart=20, sharded_result=sharded_result)
    T['inhomogeneous_solve_s'] = time.perf_counter() - t0

    # homogeneous correction: double-layer density on the boundary, evaluated through QFS
    t0 = time.perf_counter()
    Singular_DLP, Naive_SLP = _forms()
    correction = HomogeneousCorrection(solver, bdy, ebdy, bc, Singular_DLP, Naive_SLP,
                                       lambda src, trg, ch: Laplace_Layer_Apply(src, trg, charge=ch),
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
    ap.add_argument("--hard", action="store_true")
    a = ap.parse_args()
    T = {}
    err, scale, *_ = run(a.nb, a.M, 'hard' if a.hard else 'easy', verbose=True, timings=T)
    print('Error: {:0.2e}'.format(err), ' (|u|max %.3f)' % scale)
    print(T)
