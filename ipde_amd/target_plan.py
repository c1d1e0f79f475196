"""Target lists as 4 x 4 tensor patches (C-ABI: ipde_laplace_apply_patches, include/ipde_hip.h).

The multi-boundary solvers evaluate the boundary densities onto `grid_pnai`: the points of a
regular grid, in C order, minus a band around every curve (reference
ipde/solvers/multi_boundary/scalar.py:63-71, ipde/ebdy_collection.py:426-429).  Such a list is,
up to the band's edges, a union of full 4 x 4 tiles of the grid, and over a tile the squared
distance to a source separates, d^2 = dx^2[a] + dy^2[b]: the patch kernel forms the eight squares
once per source and pays one more instruction per pair.  This module finds the tiles — once per
target set, the lists live as long as the solver; tiles the band cut into are patches with
unstored points, and whatever fits no tile (repeated points, lists that are no grid) stays with
the list kernel.

Nothing here assumes a uniform grid: the lattice is the set of distinct x and distinct y VALUES of
the list (exact comparisons), so any tensor-product set qualifies and anything else simply yields
no patches.  torch does the sorting on whichever device holds the list.
"""
import torch

from .device import get_context, ptr

# lanes of a wave take consecutive patches: (8, 8) makes a wave an 8 x 8 block of tiles, so the
# sixteen lanes of one LDS pass look up neighbouring intervals of the log table
DEFAULT_BLOCK = (8, 8)
# below this many full tiles the list kernel alone is used (its lanes hold 4 targets, a patch
# lane 16: a small list would leave most of the CUs without a workgroup)
MIN_PATCHES = 1 << 14
# tiles with missing points are taken as patches while the points of all tiles are at least
# this fraction of 16 per tile (a grid_pnai list: 0.995)
PARTIAL_MIN_FILL = 0.9
# a coordinate value is a grid line if at least this many points of the list have it
LINE_MIN_POINTS = 4


class TargetPlan:
    """pxy (8, np) float64: rows 0-3 the xs, rows 4-7 the ys of patch p; pout (16, np) int32: the
    list positions of its targets, row 4 a + b for (xs[a], ys[b]), -1 where the list has no such
    point; rest (nrest,) int64: the list positions no patch holds, with their coordinates
    rest_x, rest_y."""

    def __init__(self, n, pxy, pout, rest, rest_x, rest_y, padded_blocks=False):
        self.n, self.pxy, self.pout = int(n), pxy, pout
        self.rest, self.rest_x, self.rest_y = rest, rest_x, rest_y
        self.np = int(pxy.shape[1])
        self.nrest = int(rest.shape[0])
        # 64 consecutive patches are one 8 x 8 block of tiles (build_host(pad_blocks=True)): what the
        # far-field form of the sum needs (ipde_laplace_apply_patches_far)
        self.padded_blocks = bool(padded_blocks)


def build(x, y, block=DEFAULT_BLOCK, min_patches=0):
    """x, y: 1-D float64 tensors of equal length (any device)."""
    x, y = x.reshape(-1), y.reshape(-1)
    n = int(x.shape[0])
    dev = x.device
    empty = lambda: TargetPlan(n, torch.empty((8, 0), dtype=torch.float64, device=dev),
                               torch.empty((16, 0), dtype=torch.int32, device=dev),
                               torch.arange(n, device=dev), x.contiguous(), y.contiguous())
    if n < 16 or n >= 2 ** 31:
        return empty()
    # the lattice: coordinate values that several points share (a grid line).  Points off the lattice
    # — grid_pnai ends with the interface nodes, reference ebdy_collection.py:426-429 — stay in the list
    def lines(v):
        u, inv, cnt = torch.unique(v, return_inverse=True, return_counts=True)
        keep = (cnt >= LINE_MIN_POINTS) & torch.isfinite(u)
        slot = torch.cumsum(keep, 0) - 1
        return u[keep], torch.where(keep[inv], slot[inv], torch.full_like(inv, -1))
    ux, ix = lines(x)
    uy, iy = lines(y)
    nx, ny = int(ux.shape[0]), int(uy.shape[0])
    on = (ix >= 0) & (iy >= 0)
    if nx < 4 or ny < 4 or nx * ny > 4 * n + 4096:
        return empty()                   # scattered points: the lattice of their values is mostly holes
    nxp, nyp = -(-nx // 4) * 4, -(-ny // 4) * 4
    me = torch.arange(n, device=dev, dtype=torch.int32)
    pos = torch.full((nxp, nyp), -1, dtype=torch.int32, device=dev)
    ixc, iyc = ix.clamp(min=0), iy.clamp(min=0)
    pos[ix[on], iy[on]] = me[on]
    owner = on & (pos[ixc, iyc] == me)   # a repeated point keeps one lattice slot; the others stay in the list
    count = (pos.view(nxp // 4, 4, nyp // 4, 4) >= 0).sum(dim=(1, 3))
    full = count > 0                     # tiles the band cut into run as patches too (missing points: pout = -1)...
    if float(count.sum()) < PARTIAL_MIN_FILL * 16.0 * float(full.sum()):
        full = count == 16               # ...unless that would compute many points nobody asked for
    I, J = torch.nonzero(full, as_tuple=True)
    if int(I.shape[0]) < max(1, min_patches):
        return empty()
    bi, bj = block
    njb = -(-(nyp // 4) // bj)
    key = ((I // bi) * njb + J // bj) * (bi * bj) + (I % bi) * bj + (J % bj)
    order = torch.argsort(key)
    I, J = I[order], J[order]
    four = torch.arange(4, device=dev)
    ia = 4 * I[None, :] + four[:, None]                  # (4, np) lattice rows of the patches
    jb = 4 * J[None, :] + four[:, None]
    uxp = torch.cat([ux, ux[-1:].expand(nxp - nx)])
    uyp = torch.cat([uy, uy[-1:].expand(nyp - ny)])
    pxy = torch.cat([uxp[ia], uyp[jb]], dim=0).contiguous()
    pout = pos[ia[:, None, :], jb[None, :, :]].reshape(16, -1).contiguous()
    in_patch = full[ixc // 4, iyc // 4] & owner
    rest = torch.nonzero(~in_patch, as_tuple=True)[0]
    return TargetPlan(n, pxy, pout, rest, x[rest].contiguous(), y[rest].contiguous())


def build_host(x, y, device=None, block=DEFAULT_BLOCK, min_patches=0, pad_blocks=False):
    """The same cut by the library's host routine (ipde_target_plan_build, csrc/target_plan.hip):
    x, y numpy arrays; the plan's arrays are uploaded to `device` (None: they stay on the host).
    What the solvers use — no GPU library is touched, so it runs beside a cold set-up for free
    (torch's first sort loads ~0.6 s of code objects).  pad_blocks: every block of tiles is filled up
    to a whole wave of patches (copies that store nothing), block (8, 8) only: the plan then also
    serves `laplace_apply(..., far=True)`."""
    import ctypes
    import numpy as np
    from . import _lib
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    n = int(x.shape[0])
    assert y.shape[0] == n
    handle = ctypes.c_void_p()
    if pad_blocks and tuple(block) != (8, 8):
        raise ValueError("pad_blocks needs block = (8, 8): one wave of patches per block")
    _lib.check(lib.ipde_target_plan_build_blocks(n, ptr(x), ptr(y), int(block[0]), int(min(block[1], 2 ** 30)),
                                                 float(PARTIAL_MIN_FILL), int(min_patches), int(LINE_MIN_POINTS),
                                                 int(bool(pad_blocks)), ctypes.byref(handle)))
    try:
        np_, nrest = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.ipde_target_plan_sizes(handle, ctypes.byref(np_), ctypes.byref(nrest)))
        pxy = np.empty((8, np_.value), dtype=np.float64)
        pout = np.empty((16, np_.value), dtype=np.int32)
        rest = np.empty(nrest.value, dtype=np.int64)
        _lib.check(lib.ipde_target_plan_export(handle, ptr(pxy), ptr(pout), ptr(rest)))
    finally:
        lib.ipde_target_plan_destroy(handle)
    up = (lambda a: torch.from_numpy(a)) if device is None else (lambda a: torch.as_tensor(a, device=device))
    return TargetPlan(n, up(pxy), up(pout), up(rest), up(np.ascontiguousarray(x[rest])),
                      up(np.ascontiguousarray(y[rest])), padded_blocks=bool(pad_blocks) and np_.value > 0)


def laplace_apply(plan, sx, sy, w_sigma=None, nx=None, ny=None, w_tau=None, ctx=None, out=None, far=False):
    """ipde_laplace_apply over a planned list (device tensors; densities weight-multiplied as in
    layer_potentials.laplace_apply): the patches through ipde_laplace_apply_patches, the
    remainder through ipde_laplace_apply.  far=True (plans with padded blocks): the patches through
    ipde_laplace_apply_patches_far — every block's far sources in a local expansion."""
    from . import _lib
    from .layer_potentials import _match, laplace_apply as list_apply
    ctx = ctx or get_context()
    sx, sy, w_sigma, nx, ny, w_tau = (_match(a, _lib.IPDE_DEVICE, ctx) for a in (sx, sy, w_sigma, nx, ny, w_tau))
    if plan.np == 0:
        return list_apply(sx, sy, plan.rest_x, plan.rest_y, w_sigma=w_sigma, nx=nx, ny=ny, w_tau=w_tau, ctx=ctx,
                          out=out)
    if out is None:
        out = torch.empty(plan.n, dtype=torch.float64, device=plan.pxy.device)
    assert out.is_contiguous() and out.dtype == torch.float64 and out.numel() == plan.n
    if plan.nrest:
        part = list_apply(sx, sy, plan.rest_x, plan.rest_y, w_sigma=w_sigma, nx=nx, ny=ny, w_tau=w_tau, ctx=ctx)
        out[plan.rest] = part
    if far and not plan.padded_blocks:
        raise ValueError("far=True needs a plan built with pad_blocks=True")
    fn = ctx.lib.ipde_laplace_apply_patches_far if far else ctx.lib.ipde_laplace_apply_patches
    ctx.check(fn(ctx.handle, int(sx.shape[0]), ptr(sx), ptr(sy), ptr(w_sigma), ptr(nx), ptr(ny), ptr(w_tau),
                 plan.np, ptr(plan.pxy), ptr(plan.pout), ptr(out)))
    return out


def stokes_apply(plan, sx, sy, wfx=None, wfy=None, nx=None, ny=None, wdx=None, wdy=None, pressure=True, ctx=None):
    """Stokeslet and / or stresslet sums (u, v, p) over a planned list with padded blocks: the patches through
    ipde_stokes_apply_patches_far (every block's far sources in local expansions), the remainder
    through ipde_stokes_apply.  Device tensors; wfx, wfy / wdx, wdy weight-multiplied."""
    from . import _lib
    from .layer_potentials import _match, stokes_apply as list_apply
    ctx = ctx or get_context()
    if not plan.padded_blocks:
        raise ValueError("the far-field form needs a plan built with pad_blocks=True")
    sx, sy, wfx, wfy, nx, ny, wdx, wdy = (_match(a, _lib.IPDE_DEVICE, ctx) for a in (sx, sy, wfx, wfy, nx, ny, wdx, wdy))
    dev = plan.pxy.device
    u = torch.empty(plan.n, dtype=torch.float64, device=dev)
    v = torch.empty(plan.n, dtype=torch.float64, device=dev)
    p = torch.empty(plan.n, dtype=torch.float64, device=dev) if pressure else None
    if plan.nrest:
        ru, rv, rp = list_apply(sx, sy, plan.rest_x, plan.rest_y, wfx=wfx, wfy=wfy, nx=nx, ny=ny, wdx=wdx, wdy=wdy,
                                pressure=pressure, ctx=ctx)
        u[plan.rest] = ru
        v[plan.rest] = rv
        if pressure:
            p[plan.rest] = rp
    ctx.check(ctx.lib.ipde_stokes_apply_patches_far(ctx.handle, int(sx.shape[0]), ptr(sx), ptr(sy), ptr(wfx),
                                                    ptr(wfy), ptr(nx), ptr(ny), ptr(wdx), ptr(wdy), plan.np,
                                                    ptr(plan.pxy), ptr(plan.pout), ptr(u), ptr(v), ptr(p)))
    return u, v, p


def modhelm_apply(plan, k, sx, sy, w_sigma=None, nx=None, ny=None, w_tau=None, ctx=None, out=None):
    """Modified Helmholtz single- and / or double-layer sums over a planned list with padded blocks: the
    patches through ipde_modhelm_apply_patches_far (far sources block by block in local expansions), the
    remainder through ipde_modhelm_apply.  Device tensors; w_sigma, w_tau weight-multiplied."""
    from . import _lib
    from .layer_potentials import _match, modified_helmholtz_apply as list_apply
    ctx = ctx or get_context()
    if not plan.padded_blocks:
        raise ValueError("the far-field form needs a plan built with pad_blocks=True")
    sx, sy, w_sigma, nx, ny, w_tau = (_match(a, _lib.IPDE_DEVICE, ctx) for a in (sx, sy, w_sigma, nx, ny, w_tau))
    if out is None:
        out = torch.empty(plan.n, dtype=torch.float64, device=plan.pxy.device)
    assert out.is_contiguous() and out.dtype == torch.float64 and out.numel() == plan.n
    if plan.nrest:
        out[plan.rest] = list_apply(sx, sy, plan.rest_x, plan.rest_y, float(k), w_sigma=w_sigma, nx=nx, ny=ny,
                                    w_tau=w_tau, ctx=ctx)
    ctx.check(ctx.lib.ipde_modhelm_apply_patches_far(ctx.handle, float(k), int(sx.shape[0]), ptr(sx), ptr(sy),
                                                     ptr(w_sigma), ptr(nx), ptr(ny), ptr(w_tau), plan.np,
                                                     ptr(plan.pxy), ptr(plan.pout), ptr(out)))
    return out
