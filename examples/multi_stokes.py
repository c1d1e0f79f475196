#!/usr/bin/env python3
"""Stokes flow on a multiply connected domain (BASELINE config 5): one outer boundary
and two holes — the flow of the reference's examples/multi_stokes.py (:29-222):
arclength-parameterised curves (:40-43), manufactured stream-function solution
(:64-82), StokesSolver over the collection (:95-96), block boundary-integral
correction (double layer + pressure fix on the outer curve, combined single + double
layer on the holes; :121-159), Stokes QFS sources per boundary (:164-177), one dense
stokeslet evaluation onto all grid and radial points (:179-181).

    python examples/multi_stokes.py [--nb 800] [--M 14]

The outer 11-arm star is strongly curved: its annulus (M nodes wide at the boundary's node
spacing) only stays a valid coordinate patch for n_b >~ 600 at M = 14 (curvature x width
< 0.5; EmbeddedBoundary warns otherwise) — the reference's own n_b = 500, M = 10 resolves the
flow to 1e-6.
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
from ipde_amd.layer_potentials import Stokes_Layer_Apply, DeviceTargets, ShardedTargets  # noqa: E402
from ipde_amd.pybie2d_compat import (star, squish, Grid, Global_Smooth_Boundary as GSB,  # noqa: E402
                                     BoundaryCollection, arc_length_parameterize)
from ipde_amd.qfs import Stokes_QFS, DenseSolver  # noqa: E402
from ipde_amd.solvers.multi_boundary.stokes import StokesSolver  # noqa: E402


def v2f(x):
    return x.reshape(2, x.size // 2)


def run(nb=800, M=14, a=4.0, b=3.0, verbose=False, solver_type='spectral', holes=True,
        return_fields=False, simple=False, warm=False, grid_backend=None, ng=None, tol=1e-12):
    T = {}
    t0 = time.perf_counter()
    MOL = SlepianMollifier(1.5 * M)
    bdy1 = GSB(c=star(4 * nb, a=0.1, r=3, f=11))
    bdy2 = GSB(c=squish(nb, x=-1.2, y=-0.7, r=0.8, b=0.6, rot=-np.pi / 4))
    bdy3 = GSB(c=star(nb, x=1, y=0.5, r=0.9, a=0.3, f=3))
    bdys = [bdy1, bdy2, bdy3] if holes else [bdy1]
    L = 3.5
    if simple:      # one gently curved boundary in a small box (quick accuracy check)
        bdys, L = [GSB(c=star(4 * nb, a=0.2, r=1, f=5))], 1.5
    bdys = [GSB(*arc_length_parameterize(bd.x, bd.y)) for bd in bdys]
    bdy1 = bdys[0]
    bh = min(bd.dt * bd.speed.min() for bd in bdys)
    if ng is None:      # grid spacing matched to the boundary spacing (reference :48-52)
        ng = 2 * int(0.5 * 2 * L // bh)
    grid = Grid([-L, L], ng, [-L, L], ng, x_endpoints=[True, False], y_endpoints=[True, False])
    ebdys = [EmbeddedBoundary(bd, bd is bdy1, M, bh, pad_zone=0, heaviside=MOL.step, qfs_tolerance=1e-14)
             for bd in bdys]
    ebdyc = EmbeddedBoundaryCollection(ebdys)
    ebdyc.register_grid(grid)
    rw = ebdys[0].radial_width
    ebdyc.ready_bump(MOL.bump, (L - rw, L - rw), rw)

    # manufactured solution from a stream function (reference :64-82)
    p_a, p_b = 3.0, 1.0
    sin, cos = np.sin, np.cos
    esin = lambda x: np.exp(sin(x))
    # set-up bracket as in the reference's examples/poisson_for_paper.py:60-64: geometry, grid, solver
    solver = StokesSolver(ebdyc, solver_type=solver_type, grid_backend=grid_backend)
    from ipde_amd.sharding import is_distributed
    if not is_distributed():
        # grid_and_radial_pts resident for the correction stage: its patch plan is cut by a background thread
        ebdyc.resident_grid_and_radial_pts()
    T['setup_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    psix = lambda x, y: esin(a * x) * cos(b * y)
    psiy = lambda x, y: esin(a * x) * sin(b * y)
    u_function = lambda x, y: psix(x, y)
    v_function = lambda x, y: -a / b * cos(a * x) * psiy(x, y)
    p_function = lambda x, y: cos(p_a * x) + esin(p_b * y)
    fu_function = lambda x, y: (a ** 2 * (sin(a * x) - cos(a * x) ** 2) + b ** 2) * psix(x, y) - p_a * sin(p_a * x)
    fv_function = lambda x, y: -a * b * cos(a * x) * psiy(x, y) * (1 + (a / b) ** 2 * sin(a * x) * (3 + sin(a * x))) \
        + p_b * cos(p_b * y) * esin(p_b * y)
    fu = EmbeddedFunction(ebdyc, function=fu_function)
    fv = EmbeddedFunction(ebdyc, function=fv_function)
    ua = EmbeddedFunction(ebdyc, function=u_function)
    va = EmbeddedFunction(ebdyc, function=v_function)
    pa = EmbeddedFunction(ebdyc, function=p_function)
    all_b = BoundaryCollection()
    for ebdy in ebdyc:
        all_b.add(ebdy.bdy, 'i' if ebdy.interior else 'e')
    all_b.amass_information()
    bdy_u = u_function(all_b.x, all_b.y)
    bdy_v = v_function(all_b.x, all_b.y)

    T['problem_definition_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    uc, vc, pc = solver(fu, fv, tol=tol, verbose=verbose, maxiter=200, restart=100)
    T['inhomogeneous_solve_s'] = time.perf_counter() - t0
    if warm:
        t0 = time.perf_counter()
        uc, vc, pc = solver(fu, fv, tol=tol, maxiter=200, restart=100)
        T['warm_inhomogeneous_solve_s'] = time.perf_counter() - t0

    # block boundary-integral system for the homogeneous correction (reference :104-159)
    t0 = time.perf_counter()
    # (assembled on the GPU: the numpy forms of the same names in pybie2d_compat are the
    # host reference; at 18 600 boundary nodes the matrix is 37 200^2)
    import torch
    from ipde_amd import dense_forms as df
    dev = torch.device("cuda", torch.cuda.current_device())
    d_only = lambda src, trg: df.stokes_form(src, trg, dev, ifdipole=True)
    c_and_d = lambda src, trg: df.stokes_form(src, trg, dev, ifforce=True, ifdipole=True)
    d_singular = lambda src: df.stokes_singular_form(src, dev, ifdipole=True)
    cd_singular = lambda src: df.stokes_singular_form(src, dev, ifforce=True, ifdipole=True)
    fix = lambda src, trg: df.stokes_pressure_fix(src, trg, dev)
    half_eye = lambda src: torch.eye(2 * src.N, dtype=torch.float64, device=dev) * 0.5
    Ns = [bd.N for bd in bdys]
    off = 2 * np.concatenate([[0], np.cumsum(Ns)])
    MAT = torch.zeros((int(off[-1]), int(off[-1])), dtype=torch.float64, device=dev)
    for i, bi in enumerate(bdys):          # target boundary
        for j, bj in enumerate(bdys):      # source boundary
            if i == j:
                blk = (d_singular(bi) - half_eye(bi) + fix(bi, bi)) if i == 0 \
                    else (cd_singular(bi) + half_eye(bi))
            elif j == 0:
                blk = d_only(bj, bi) + fix(bj, bi)
            else:
                blk = c_and_d(bj, bi)
            MAT[off[i]:off[i + 1], off[j]:off[j + 1]] = blk
            del blk
    bu = solver.get_boundary_values(uc.get_radial_value_list())
    bv = solver.get_boundary_values(vc.get_radial_value_list())
    bu_adj = ebdyc.v2l(bdy_u - bu)
    bv_adj = ebdyc.v2l(bdy_v - bv)
    bc_adj = np.concatenate([np.concatenate([p, q]) for p, q in zip(bu_adj, bv_adj)])
    tau = DenseSolver(MAT).solve(bc_adj)
    taul = ebdyc.v2l2(tau)
    sigmal, sources = [], BoundaryCollection()
    for ebdy, t in zip(ebdys, taul):
        qfs = Stokes_QFS(ebdy.bdy, ebdy.interior, not ebdy.interior, True, qfs_boundary=ebdy.bdy_qfs)
        sigmal.append(qfs([t, t] if qfs.interior else [t, ]))
        sources.add(qfs.source, 'i' if qfs.interior else 'e')
    sources.amass_information()
    sigmav = np.column_stack([v2f(s) for s in sigmal])
    # the one big sum of the correction (reference :179): onto grid_and_radial_pts resident in HBM in the form its
    # sums are fastest in — grid points in padded 4 x 4 patch blocks, radial grids as columns, far sources block
    # by block in local expansions (ipde_stokes_apply_patches_far / _columns_far); target-sharded under
    # torch.distributed
    from ipde_amd.sharding import is_distributed
    targets = ShardedTargets(ebdyc.grid_and_radial_pts) if is_distributed() else ebdyc.resident_grid_and_radial_pts()
    t1 = time.perf_counter()
    out = Stokes_Layer_Apply(sources, targets, forces=sigmav)
    torch.cuda.synchronize()
    T['homogeneous_sum_s'] = time.perf_counter() - t1
    for f, o in zip((uc, vc, pc), out):
        f += o.cpu().numpy()
    T['homogeneous_s'] = time.perf_counter() - t0

    u_err = np.abs(np.asarray(uc) - np.asarray(ua)).max()
    v_err = np.abs(np.asarray(vc) - np.asarray(va)).max()
    # the pressure is defined up to a constant: compare after removing the domain means
    area = EmbeddedFunction(ebdyc, function=lambda x, y: np.ones_like(x)).integrate()
    pd = pc - pa
    pd -= pd.integrate() / area
    p_err = np.abs(np.asarray(pd)).max()
    T['dof'] = int(ebdyc.dof)
    T['grid'] = list(grid.shape)
    T['gmres_iterations'] = solver.iteration_counts
    scale = max(np.abs(np.asarray(ua)).max(), np.abs(np.asarray(va)).max())
    if return_fields:
        return ebdyc, (uc, vc, pc), (ua, va, pa), pd
    return float(u_err), float(v_err), float(p_err), float(scale), T


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=800)
    ap.add_argument("--M", type=int, default=14)
    ap.add_argument("--a", type=float, default=4.0)
    ap.add_argument("--b", type=float, default=3.0)
    ap.add_argument("--single", action="store_true", help="outer boundary only")
    ap.add_argument("--warm", action="store_true", help="time a second (warm) inhomogeneous solve")
    ap.add_argument("--simple", action="store_true", help="one 5-arm star of radius 1 in [-1.5, 1.5]^2")
    a_ = ap.parse_args()
    ue, ve, pe, scale, T = run(a_.nb, a_.M, a_.a, a_.b, verbose=True, holes=not a_.single, simple=a_.simple, warm=a_.warm,
                               grid_backend=None)
    print('Error, u {:0.2e}'.format(ue))
    print('Error, v {:0.2e}'.format(ve))
    print('Error, p {:0.2e} (mean removed)'.format(pe), ' (|u|max %.3f)' % scale)
    print(T)
