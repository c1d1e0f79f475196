#!/usr/bin/env python3
"""Modified Helmholtz on a multiply connected domain: one outer (interior-type)
boundary and two holes (exterior-type) — the flow of the reference's
examples/multi_modified_helmholtz_update_to_sparse.py (:30-150): manufactured
solution, ModifiedHelmholtzSolver over the collection, block boundary-integral
correction (double layer on the outer curve, combined single+double layer on the
holes; :100-120), QFS sources per boundary, one dense evaluation onto all grid and
radial points.

    python examples/multi_modified_helmholtz.py [--nb 400] [--M 16] [--k 2.0]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ipde_amd.ebdy_collection import EmbeddedBoundaryCollection  # noqa: E402
from ipde_amd.embedded_boundary import EmbeddedBoundary  # noqa: E402
from ipde_amd.embedded_function import EmbeddedFunction  # noqa: E402
from ipde_amd.heavisides import SlepianMollifier  # noqa: E402
from ipde_amd.layer_potentials import Modified_Helmholtz_Layer_Apply, DeviceTargets, ShardedTargets  # noqa: E402
from ipde_amd.pybie2d_compat import star, squish, Grid, Global_Smooth_Boundary as GSB  # noqa: E402
from ipde_amd.qfs import QFS_Evaluator, DenseSolver  # noqa: E402
from ipde_amd.solvers.multi_boundary.modified_helmholtz import ModifiedHelmholtzSolver  # noqa: E402


def run(nb=400, M=16, helmholtz_k=2.0, verbose=False, return_solution=False):
    T = {}
    t0 = time.perf_counter()
    MOL = SlepianMollifier(1.5 * M)
    bdy1 = GSB(c=star(4 * nb, a=0.05, r=3, f=7))
    bdy2 = GSB(c=squish(nb, x=-1.2, y=-0.7, r=0.8, b=0.6, rot=-np.pi / 4))
    bdy3 = GSB(c=star(nb, x=1, y=0.5, r=0.9, a=0.2, f=3))
    bdys = [bdy1, bdy2, bdy3]
    bh = min(b.dt * b.speed.min() for b in bdys)
    ng = 2 * int(0.5 * 7 // bh)
    grid = Grid([-3.5, 3.5], ng, [-3.5, 3.5], ng, x_endpoints=[True, False], y_endpoints=[True, False])
    ebdys = [EmbeddedBoundary(b, b is bdy1, M, bh, pad_zone=0, heaviside=MOL.step, qfs_tolerance=1e-14)
             for b in bdys]
    ebdyc = EmbeddedBoundaryCollection(ebdys)
    ebdyc.register_grid(grid)
    # set-up bracket as in the reference's examples/poisson_for_paper.py:60-64: geometry, grid, solver
    solver = ModifiedHelmholtzSolver(ebdyc, k=helmholtz_k)
    from ipde_amd.sharding import is_distributed
    if not is_distributed():
        # grid_and_radial_pts resident for the correction stage: its patch plan is cut by a background thread
        ebdyc.resident_grid_and_radial_pts()
    T['setup_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    kk = 2 * np.pi / 7
    solution_func = lambda x, y: np.exp(np.sin(kk * x)) * np.sin(kk * y)
    force_func = lambda x, y: helmholtz_k ** 2 * solution_func(x, y) \
        - kk ** 2 * np.exp(np.sin(kk * x)) * np.sin(kk * y) * (np.cos(kk * x) ** 2 - np.sin(kk * x) - 1.0)
    f = EmbeddedFunction(ebdyc)
    f.define_via_function(force_func)
    ua = EmbeddedFunction(ebdyc)
    ua.define_via_function(solution_func)
    bcs2v = solution_func(ebdyc.all_bvx, ebdyc.all_bvy)
    T['problem_definition_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ue = solver(f, tol=1e-14, verbose=verbose, maxiter=100, restart=20)
    T['inhomogeneous_solve_s'] = time.perf_counter() - t0

    # block boundary-integral system (reference :100-120)
    t0 = time.perf_counter()
    k = helmholtz_k
    import torch
    from ipde_amd import dense_forms as df      # the dense matrices are assembled on the GPU
    dev = torch.device('cuda', torch.cuda.current_device())
    d_only = lambda src, trg: df.modhelm_form(src, trg, dev, k, ifdipole=True)
    c_and_d = lambda src, trg: df.modhelm_form(src, trg, dev, k, ifcharge=True, ifdipole=True)
    d_singular = lambda src: df.modhelm_singular_form(src, dev, k, ifdipole=True)
    cd_singular = lambda src: df.modhelm_singular_form(src, dev, k, ifcharge=True, ifdipole=True)
    half_eye = lambda src: torch.eye(src.N, dtype=torch.float64, device=dev) * 0.5
    Ns = [b.N for b in bdys]
    off = np.concatenate([[0], np.cumsum(Ns)])
    MAT = torch.zeros((int(off[-1]), int(off[-1])), dtype=torch.float64, device=dev)
    for i, bi in enumerate(bdys):          # target boundary
        for j, bj in enumerate(bdys):      # source boundary
            blk = MAT[off[i]:off[i + 1], off[j]:off[j + 1]]
            if i == j:
                blk[:] = d_singular(bi) - half_eye(bi) if i == 0 else cd_singular(bi) + half_eye(bi)
            else:
                blk[:] = d_only(bj, bi) if j == 0 else c_and_d(bj, bi)
    bvs = np.concatenate(solver.get_boundary_values(ue.get_radial_value_list()).bdy_value_list)
    tau = DenseSolver(MAT).solve(bcs2v - bvs)
    taul = ebdyc.v2l(tau)
    Naive_SLP = lambda src, trg: df.modhelm_form(src, trg, dev, k, ifcharge=True)
    sigmal = []
    for ebdy, t in zip(ebdys, taul):
        if ebdy.interior:
            K = lambda src, _: d_singular(src) - half_eye(src)
        else:
            K = lambda src, _: cd_singular(src) + half_eye(src)
        qfs = QFS_Evaluator(ebdy.bdy_qfs, ebdy.interior, [K, ], Naive_SLP, on_surface=True, form_b2c=False)
        sigmal.append(qfs([t, ]))
    sigmav = np.concatenate(sigmal)
    targets = ShardedTargets(ebdyc.grid_and_radial_pts) if is_distributed() else ebdyc.resident_grid_and_radial_pts()
    out = Modified_Helmholtz_Layer_Apply(ebdyc.bdy_inward_sources, targets, k=k, charge=sigmav).cpu().numpy()
    gslp, rslpl = ebdyc.divide_grid_and_radial(out)
    for i in range(len(ebdys)):
        ue[i] += rslpl[i].reshape(ebdys[i].radial_shape)
    ue['grid'] += gslp
    T['homogeneous_s'] = time.perf_counter() - t0
    err = np.abs(np.asarray(ue) - np.asarray(ua))
    T['dof'] = int(ebdyc.dof)
    T['grid'] = list(grid.shape)
    T['gmres_iterations'] = solver.iteration_counts
    if return_solution:
        return float(err.max()), float(np.abs(np.asarray(ua)).max()), T, np.array(ue)
    return float(err.max()), float(np.abs(np.asarray(ua)).max()), T


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=400)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--k", type=float, default=2.0)
    a = ap.parse_args()
    err, scale, T = run(a.nb, a.M, a.k, verbose=True)
    print('Error: {:0.2e}'.format(err), ' (|u|max %.3f)' % scale)
    print(T)
