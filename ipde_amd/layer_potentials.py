"""Layer-potential applies on the MI355X.

High-level functions keep the call shape of the pybie2d functions the reference
binds (`Laplace_Layer_Apply`, `Modified_Helmholtz_Layer_Apply`; reference
ipde/solvers/internals/poisson.py:12,35, modified_helmholtz.py:3,37) and of the
Stokes `Layer_Apply(src, trg, f) -> (u, v, p)` closure
(ipde/solvers/internals/stokes.py:25-35): `source` is any object with
.x, .y, .weights (and .normal_x, .normal_y for double layers), `target` any object
with .x, .y; densities are NOT yet multiplied by the weights.

Targets may carry device-resident coordinates (see `DeviceTargets`): then the
result is a torch CUDA tensor and nothing crosses PCIe except the (tiny) source
arrays.
"""
import numpy as np
import torch

from . import _lib
from .device import get_context, location_of, as_f64, ptr, empty_like_loc, to_device

__all__ = [
    "laplace_apply", "modified_helmholtz_apply", "stokes_apply",
    "Laplace_Layer_Apply", "Modified_Helmholtz_Layer_Apply", "Stokes_Layer_Apply",
    "DeviceTargets", "ShardedTargets", "CompositeTargets", "make_laplace_layer_apply", "make_modified_helmholtz_layer_apply",
    "make_stokes_layer_apply",
]


_plan_threads = []


def _join_plan_threads():
    # a process must not end while a daemon thread is inside torch / HIP (see device._drain_at_exit)
    for t in _plan_threads:
        t.join(30.0)


import atexit
atexit.register(_join_plan_threads)


# radial grids with fewer lines than this keep the list kernel (the column form's blocks are 64 lines)
COLUMNS_FAR_MIN = 1024


class DeviceTargets:
    """A target point set kept resident in HBM (the solver evaluates onto the same
    `grid_pnai` / `radial_targ` / `grid_and_radial_pts` sets in every solve;
    reference ipde/ebdy_collection.py:426-429,488-491)."""

    def __init__(self, x, y=None, ctx=None, plan=False, far=False, columns=None):
        """plan=True: the list is also cut into 4 x 4 tensor patches for the Laplace patch kernel
        (ipde_amd/target_plan.py) — in a background thread; the first Laplace apply onto the set
        joins it.  For the big grid lists (grid_pnai); lists under 2^18 points keep the list kernel.
        far=True (with plan): blocks of 8 x 8 patches padded to whole waves, and Laplace sums onto the
        set take every block's far sources through a local expansion (ipde_laplace_apply_patches_far)."""
        if y is None:  # a PointSet-like object
            x, y = x.x, x.y
        self.ctx = ctx or get_context()
        # (host copies kept for the plan builder only while it runs)
        self._host_xy = None
        if plan and not isinstance(x, torch.Tensor) and not isinstance(y, torch.Tensor):
            self._host_xy = (np.ascontiguousarray(x, dtype=np.float64).ravel(),
                             np.ascontiguousarray(y, dtype=np.float64).ravel())
        self.x = to_device(np.asarray(x, dtype=np.float64).ravel(), self.ctx) \
            if not isinstance(x, torch.Tensor) else x.to(torch.float64).contiguous().view(-1)
        self.y = to_device(np.asarray(y, dtype=np.float64).ravel(), self.ctx) \
            if not isinstance(y, torch.Tensor) else y.to(torch.float64).contiguous().view(-1)
        self.N = int(self.x.shape[0])
        self._plan, self._plan_thread, self._plan_error = None, None, None
        # columns = (M, N): the list is an (M, N) array raveled in C order whose columns are radial lines of
        # an annulus (EmbeddedBoundary.radial_x / radial_y): sums that have a column far-field form use it
        self.columns = None
        if columns is not None:
            M, N = int(columns[0]), int(columns[1])
            if M * N != self.N or M < 2:
                raise ValueError("columns = (M, N) must describe the whole list")
            self.columns = (M, N)
        self.far = bool(far) and bool(plan)
        if plan:
            self.request_plan()

    def request_plan(self):
        from . import target_plan
        if self._plan_thread is not None or not self.x.is_cuda or self.N < 16 * target_plan.MIN_PATCHES:
            return
        import threading
        dev = self.x.device

        def work():
            try:
                torch.cuda.set_device(dev)
                side = torch.cuda.Stream(dev)       # not in the way of the set-up's own launches
                with torch.cuda.stream(side):
                    if self._host_xy is not None:   # the library's host routine, then four uploads
                        plan = target_plan.build_host(*self._host_xy, device=dev,
                                                      min_patches=target_plan.MIN_PATCHES, pad_blocks=self.far)
                    else:                           # a list that only exists on the device: torch sorts
                        plan = target_plan.build(self.x, self.y, min_patches=target_plan.MIN_PATCHES)
                side.synchronize()
                self._plan = plan if plan.np else None
            except Exception as e:                  # the list kernel needs no plan
                self._plan_error = e
            finally:
                self._host_xy = None
        self._plan_thread = threading.Thread(target=work, name="ipde-target-plan", daemon=True)
        self._plan_thread.start()
        _plan_threads.append(self._plan_thread)

    def plan(self):
        """the patch plan if one was requested (waits for the thread that builds it), else None"""
        if self._plan_thread is None:
            return None
        self._plan_thread.join()
        return self._plan


class ShardedTargets:
    """A resident target set split over the ranks of a torch.distributed job (one process
    per GPU): this rank keeps its contiguous slice in HBM, the high-level *_Layer_Apply
    functions evaluate the slice and all-gather the slices, so the caller sees the full
    result on every rank exactly as with `DeviceTargets`.  In a single process it IS a
    `DeviceTargets`.  For the big one-off sums of the example scripts (homogeneous
    correction onto grid_and_radial_pts, reference examples/interior_poisson.py:84-92)."""

    def __init__(self, x, y=None, ctx=None, owned=None):
        """owned: boolean mask over the points — evaluate only there and leave the result sharded:
        full-length tensors with zeros at the other points, no collective (the form a solver's
        `sharded_result` answer takes: its `owned` mask partitions the points over the ranks)."""
        from . import sharding
        if y is None:
            x, y = x.x, x.y
        x, y = np.asarray(x, dtype=np.float64).ravel(), np.asarray(y, dtype=np.float64).ravel()
        self.N = int(x.shape[0])
        _, rank, world = sharding._dist_state()
        self.world = world
        self._owned_idx = None
        if owned is not None:
            idx = np.nonzero(np.asarray(owned, dtype=bool).ravel())[0]
            self.local = DeviceTargets(x[idx], y[idx], ctx=ctx)
            self._owned_idx = to_device(idx.astype(np.int64), self.local.ctx)
        else:
            sl = sharding.target_slice(self.N, rank, world)
            self.local = DeviceTargets(x[sl], y[sl], ctx=ctx)
        self._gathers = {}

    def evaluate(self, apply_local):
        """apply_local(DeviceTargets) -> tensor or tuple of tensors on the local slice"""
        out = apply_local(self.local)
        if self._owned_idx is not None:
            parts = out if isinstance(out, tuple) else (out,)
            full = []
            for p in parts:
                z = torch.zeros(self.N, dtype=p.dtype, device=p.device)
                z[self._owned_idx] = p
                full.append(z)
            return tuple(full) if isinstance(out, tuple) else full[0]
        if self.world == 1:
            return out
        from . import sharding
        parts = out if isinstance(out, tuple) else (out,)
        g = self._gathers.get(len(parts))
        if g is None:
            g = self._gathers[len(parts)] = sharding.ResultGather(self.N, len(parts))
        full = g(parts)
        return full if isinstance(out, tuple) else full[0]


class CompositeTargets:
    """Resident target sets end to end, each in the form its sums are fastest in — what
    `ebdyc.grid_and_radial_pts` is (reference ipde/ebdy_collection.py:426-429): the physical grid points
    (a lattice: 4 x 4 patches in padded blocks, far sources block by block in local expansions) followed by
    every boundary's (M, N) radial grid (blocks of 64 radial lines, the column far-field forms).  The
    high-level *_Layer_Apply functions evaluate part by part into ONE device tensor in the list's order,
    which is the storage order of an EmbeddedFunction / hostio.DeviceFunction: the example scripts'
    homogeneous correction (reference examples/interior_poisson.py:84-92) adds it onto the solution where it
    lies.  Built once per geometry (`EmbeddedBoundaryCollection.resident_grid_and_radial_pts`)."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.N = int(sum(p.N for p in self.parts))
        self.ctx = self.parts[0].ctx

    def evaluate(self, apply_part):
        """apply_part(DeviceTargets) -> tensor or tuple of tensors; -> the same over the whole list"""
        outs = [apply_part(p) for p in self.parts]
        if isinstance(outs[0], tuple):
            return tuple(None if c[0] is None else (torch.cat([torch.as_tensor(a) for a in c]) if len(c) > 1
                                                    else torch.as_tensor(c[0])) for c in zip(*outs))
        return torch.cat([torch.as_tensor(a) for a in outs]) if len(outs) > 1 else torch.as_tensor(outs[0])

    def wait(self):
        """join the background threads that cut the parts' patch plans"""
        for p in self.parts:
            p.plan()
        return self


def _match(a, loc, ctx):
    """Bring a (small) source-side array to the location of the targets."""
    if a is None:
        return None
    if loc == _lib.IPDE_DEVICE:
        if isinstance(a, torch.Tensor):
            return a.to(device=ctx.torch_device(), dtype=torch.float64).contiguous()
        return to_device(np.asarray(a, dtype=np.float64), ctx)
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return as_f64(a, loc)


def _flags(skip_coincident, generic_math):
    return (_lib.FLAG_SKIP_COINCIDENT if skip_coincident else 0) | \
           (_lib.FLAG_GENERIC_MATH if generic_math else 0)


def laplace_apply(sx, sy, tx, ty, w_sigma=None, nx=None, ny=None, w_tau=None,
                  skip_coincident=False, generic_math=False, ctx=None, out=None):
    """ipde_laplace_apply: densities already weight-multiplied (C-ABI convention)."""
    ctx = ctx or get_context()
    loc = location_of(tx, ty)
    tx, ty = as_f64(tx, loc), as_f64(ty, loc)
    sx, sy, w_sigma, nx, ny, w_tau = (_match(a, loc, ctx) for a in (sx, sy, w_sigma, nx, ny, w_tau))
    nt, ns = int(tx.shape[0]), int(sx.shape[0])
    if out is None:
        out = empty_like_loc((nt,), loc, ctx)
    ctx.check(ctx.lib.ipde_laplace_apply(ctx.handle, loc, ns, ptr(sx), ptr(sy), ptr(w_sigma),
                                         ptr(nx), ptr(ny), ptr(w_tau), nt, ptr(tx), ptr(ty),
                                         ptr(out), _flags(skip_coincident, generic_math)))
    return out


def modified_helmholtz_apply(sx, sy, tx, ty, k, w_sigma=None, nx=None, ny=None, w_tau=None,
                             skip_coincident=False, generic_math=False, ctx=None, out=None):
    ctx = ctx or get_context()
    loc = location_of(tx, ty)
    tx, ty = as_f64(tx, loc), as_f64(ty, loc)
    sx, sy, w_sigma, nx, ny, w_tau = (_match(a, loc, ctx) for a in (sx, sy, w_sigma, nx, ny, w_tau))
    nt, ns = int(tx.shape[0]), int(sx.shape[0])
    if out is None:
        out = empty_like_loc((nt,), loc, ctx)
    ctx.check(ctx.lib.ipde_modhelm_apply(ctx.handle, loc, float(k), ns, ptr(sx), ptr(sy),
                                         ptr(w_sigma), ptr(nx), ptr(ny), ptr(w_tau), nt, ptr(tx),
                                         ptr(ty), ptr(out), _flags(skip_coincident, generic_math)))
    return out


def stokes_apply(sx, sy, tx, ty, wfx=None, wfy=None, nx=None, ny=None, wdx=None, wdy=None,
                 pressure=True, skip_coincident=False, generic_math=False, ctx=None):
    ctx = ctx or get_context()
    loc = location_of(tx, ty)
    tx, ty = as_f64(tx, loc), as_f64(ty, loc)
    sx, sy, wfx, wfy, nx, ny, wdx, wdy = (_match(a, loc, ctx)
                                          for a in (sx, sy, wfx, wfy, nx, ny, wdx, wdy))
    nt, ns = int(tx.shape[0]), int(sx.shape[0])
    u = empty_like_loc((nt,), loc, ctx)
    v = empty_like_loc((nt,), loc, ctx)
    p = empty_like_loc((nt,), loc, ctx) if pressure else None
    ctx.check(ctx.lib.ipde_stokes_apply(ctx.handle, loc, ns, ptr(sx), ptr(sy), ptr(wfx), ptr(wfy),
                                        ptr(nx), ptr(ny), ptr(wdx), ptr(wdy), nt, ptr(tx), ptr(ty),
                                        ptr(u), ptr(v), ptr(p),
                                        _flags(skip_coincident, generic_math)))
    return u, v, p


# ---------------------------------------------------------------------------
def _xy(obj):
    return obj.x, obj.y


class _DeviceSource(object):
    """x, y, weights (and normals) of a source curve resident in HBM.  The solvers apply the
    same few source curves in every solve (the QFS source curves of the interfaces, their
    collection): without this every apply re-sent four to six small arrays, ~30 us each."""
    __slots__ = ("key", "x", "y", "weights", "normal_x", "normal_y")


def _device_source(source, ctx):
    """cached device copies of a source's geometry; rebuilt if the object's arrays were replaced"""
    cache = source.__dict__.setdefault("_ipde_dev_source", {}) if hasattr(source, "__dict__") else {}
    key = (id(source.x), id(source.y), id(source.weights))
    d = cache.get(ctx.device)
    if d is None or d.key != key:
        d = _DeviceSource()
        d.key = key
        d.x = to_device(np.ascontiguousarray(source.x, dtype=np.float64).ravel(), ctx)
        d.y = to_device(np.ascontiguousarray(source.y, dtype=np.float64).ravel(), ctx)
        d.weights = to_device(np.ascontiguousarray(source.weights, dtype=np.float64).ravel(), ctx)
        nx = getattr(source, "normal_x", None)
        d.normal_x = None if nx is None else to_device(np.ascontiguousarray(nx, dtype=np.float64).ravel(), ctx)
        ny = getattr(source, "normal_y", None)
        d.normal_y = None if ny is None else to_device(np.ascontiguousarray(ny, dtype=np.float64).ravel(), ctx)
        cache[ctx.device] = d
    return d


def _source_side(source, trg):
    """the source's geometry where the targets are: the cached device copy for device-resident
    targets, the object itself (host arrays) otherwise"""
    if isinstance(trg.x, torch.Tensor) and trg.x.is_cuda:
        return _device_source(source, getattr(trg, "ctx", None) or get_context())
    return source


def _weighted(density, weights, rows=None):
    """density * weights; a device density stays on the device (weights: the cached device copy).
    rows: reshape to (rows, -1) first (the (2, N) Stokes densities)."""
    if density is None:
        return None
    if isinstance(density, torch.Tensor):
        d = density.to(torch.float64)
        w = weights if isinstance(weights, torch.Tensor) else torch.as_tensor(
            np.asarray(weights, dtype=np.float64), device=d.device)
        return (d.reshape(rows, -1) if rows else d.reshape(-1)) * w
    if isinstance(weights, torch.Tensor):      # host density onto device-resident targets
        d = torch.as_tensor(np.ascontiguousarray(density, dtype=np.float64), device=weights.device)
        return (d.reshape(rows, -1) if rows else d.reshape(-1)) * weights
    d = np.asarray(density, dtype=np.float64)
    return (d.reshape(rows, -1) if rows else d) * np.asarray(weights, dtype=np.float64)


def Laplace_Layer_Apply(source, target=None, charge=None, dipstr=None, backend=None, **kwargs):
    """out = SLP[charge] + DLP[dipstr] evaluated at target (pybie2d call shape;
    reference ipde/solvers/internals/poisson.py:35, examples/interior_poisson.py:89).
    target None: source onto itself, the coincident pairs skipped."""
    if isinstance(target, (ShardedTargets, CompositeTargets)):
        return target.evaluate(lambda t: Laplace_Layer_Apply(source, t, charge=charge, dipstr=dipstr))
    self_eval = target is None
    trg = source if self_eval else target
    tx, ty = _xy(trg)
    if charge is None and dipstr is None:
        raise ValueError("need a charge and/or a dipstr density")
    src = _source_side(source, trg)
    plan = target.plan() if isinstance(target, DeviceTargets) else None
    if plan is not None:
        from . import target_plan
        return target_plan.laplace_apply(plan, src.x, src.y,
                                         w_sigma=_weighted(charge, src.weights),
                                         nx=None if dipstr is None else src.normal_x,
                                         ny=None if dipstr is None else src.normal_y,
                                         w_tau=_weighted(dipstr, src.weights), ctx=target.ctx,
                                         far=target.far and plan.padded_blocks)
    if not self_eval and isinstance(target, DeviceTargets) and target.columns is not None \
            and target.columns[1] >= COLUMNS_FAR_MIN:
        # a radial grid: blocks of 64 radial lines, far sources in the blocks' local expansions
        ctx = target.ctx
        M, N = target.columns
        sxd, syd, wd, nxd, nyd, td = (_match(a, _lib.IPDE_DEVICE, ctx) for a in (
            src.x, src.y, _weighted(charge, src.weights), None if dipstr is None else src.normal_x,
            None if dipstr is None else src.normal_y, _weighted(dipstr, src.weights)))
        out = torch.empty(target.N, dtype=torch.float64, device=target.x.device)
        ctx.check(ctx.lib.ipde_laplace_apply_columns_far(ctx.handle, int(sxd.shape[0]), ptr(sxd), ptr(syd), ptr(wd),
                                                         ptr(nxd), ptr(nyd), ptr(td), M, N, ptr(target.x),
                                                         ptr(target.y), ptr(out)))
        return out
    return laplace_apply(src.x, src.y, tx, ty,
                         w_sigma=_weighted(charge, src.weights),
                         nx=None if dipstr is None else src.normal_x,
                         ny=None if dipstr is None else src.normal_y,
                         w_tau=_weighted(dipstr, src.weights),
                         skip_coincident=self_eval)


def Modified_Helmholtz_Layer_Apply(source, target=None, k=1.0, charge=None, dipstr=None,
                                   backend=None, **kwargs):
    """(reference ipde/solvers/internals/modified_helmholtz.py:37)"""
    if isinstance(target, (ShardedTargets, CompositeTargets)):
        return target.evaluate(lambda t: Modified_Helmholtz_Layer_Apply(source, t, k=k, charge=charge,
                                                                        dipstr=dipstr))
    self_eval = target is None
    trg = source if self_eval else target
    tx, ty = _xy(trg)
    if charge is None and dipstr is None:
        raise ValueError("need a charge and/or a dipstr density")
    src = _source_side(source, trg)
    if not self_eval and isinstance(target, DeviceTargets) and target.far:
        plan = target.plan()
        if plan is not None and plan.padded_blocks:      # far sources block by block in local expansions
            from . import target_plan
            return target_plan.modhelm_apply(plan, k, src.x, src.y, w_sigma=_weighted(charge, src.weights),
                                             nx=None if dipstr is None else src.normal_x,
                                             ny=None if dipstr is None else src.normal_y,
                                             w_tau=_weighted(dipstr, src.weights), ctx=target.ctx)
    if not self_eval and isinstance(target, DeviceTargets) and target.columns is not None \
            and target.columns[1] >= COLUMNS_FAR_MIN:
        # a radial grid: blocks of 64 radial lines, far sources in the blocks' local expansions
        ctx = target.ctx
        M, N = target.columns
        sxd, syd, wd, nxd, nyd, td = (_match(a, _lib.IPDE_DEVICE, ctx) for a in (
            src.x, src.y, _weighted(charge, src.weights), None if dipstr is None else src.normal_x,
            None if dipstr is None else src.normal_y, _weighted(dipstr, src.weights)))
        out = torch.empty(target.N, dtype=torch.float64, device=target.x.device)
        ctx.check(ctx.lib.ipde_modhelm_apply_columns_far(ctx.handle, float(k), int(sxd.shape[0]), ptr(sxd), ptr(syd),
                                                         ptr(wd), ptr(nxd), ptr(nyd), ptr(td), M, N, ptr(target.x),
                                                         ptr(target.y), ptr(out)))
        return out
    return modified_helmholtz_apply(src.x, src.y, tx, ty, k,
                                    w_sigma=_weighted(charge, src.weights),
                                    nx=None if dipstr is None else src.normal_x,
                                    ny=None if dipstr is None else src.normal_y,
                                    w_tau=_weighted(dipstr, src.weights),
                                    skip_coincident=self_eval)


def Stokes_Layer_Apply(source, target=None, forces=None, dipstr=None, pressure=True, **kwargs):
    """Returns (u, v, p) like the reference's Stokes Layer_Apply closure
    (ipde/solvers/internals/stokes.py:25-35).  forces / dipstr have shape (2, N)."""
    if isinstance(target, (ShardedTargets, CompositeTargets)):
        return target.evaluate(lambda t: Stokes_Layer_Apply(source, t, forces=forces, dipstr=dipstr,
                                                            pressure=pressure))
    self_eval = target is None
    trg = source if self_eval else target
    tx, ty = _xy(trg)
    if forces is None and dipstr is None:
        raise ValueError("need a forces and/or a dipstr density")
    src = _source_side(source, trg)
    f = _weighted(forces, src.weights, rows=2)
    g = _weighted(dipstr, src.weights, rows=2)
    if not self_eval and isinstance(target, DeviceTargets) and target.far:
        plan = target.plan()
        if plan is not None and plan.padded_blocks:      # far sources block by block in local expansions
            from . import target_plan
            return target_plan.stokes_apply(plan, src.x, src.y, wfx=None if f is None else f[0],
                                            wfy=None if f is None else f[1],
                                            nx=None if g is None else src.normal_x,
                                            ny=None if g is None else src.normal_y,
                                            wdx=None if g is None else g[0], wdy=None if g is None else g[1],
                                            pressure=pressure, ctx=target.ctx)
    if not self_eval and isinstance(target, DeviceTargets) and target.columns is not None \
            and target.columns[1] >= COLUMNS_FAR_MIN:
        # a radial grid: blocks of 64 radial lines, far sources in the blocks' local expansions
        ctx = target.ctx
        M, N = target.columns
        sxd, syd, fxd, fyd, nxd, nyd, gxd, gyd = (_match(a, _lib.IPDE_DEVICE, ctx) for a in (
            src.x, src.y, None if f is None else f[0], None if f is None else f[1],
            None if g is None else src.normal_x, None if g is None else src.normal_y,
            None if g is None else g[0], None if g is None else g[1]))
        dev = target.x.device
        u = torch.empty(target.N, dtype=torch.float64, device=dev)
        v = torch.empty(target.N, dtype=torch.float64, device=dev)
        p = torch.empty(target.N, dtype=torch.float64, device=dev) if pressure else None
        ctx.check(ctx.lib.ipde_stokes_apply_columns_far(ctx.handle, int(sxd.shape[0]), ptr(sxd), ptr(syd), ptr(fxd),
                                                        ptr(fyd), ptr(nxd), ptr(nyd), ptr(gxd), ptr(gyd), M, N,
                                                        ptr(target.x), ptr(target.y), ptr(u), ptr(v), ptr(p)))
        return u, v, p
    return stokes_apply(src.x, src.y, tx, ty,
                        wfx=None if f is None else f[0], wfy=None if f is None else f[1],
                        nx=None if g is None else src.normal_x,
                        ny=None if g is None else src.normal_y,
                        wdx=None if g is None else g[0], wdy=None if g is None else g[1],
                        pressure=pressure, skip_coincident=self_eval)


# -- the `Layer_Apply(src, trg, ch)` closures the solver helpers store ---------
def make_laplace_layer_apply():
    """Drop-in for PoissonHelper._define_layer_apply (internals/poisson.py:27-36)."""
    def func(src, trg, ch):
        return Laplace_Layer_Apply(src, trg, charge=ch)
    return func


def make_modified_helmholtz_layer_apply(k):
    """Drop-in for ModifiedHelmholtzHelper._define_layer_apply
    (internals/modified_helmholtz.py:28-37)."""
    def func(src, trg, ch):
        return Modified_Helmholtz_Layer_Apply(src, trg, charge=ch, k=k)
    return func


def make_stokes_layer_apply():
    """Drop-in for StokesHelper._define_layer_apply (internals/stokes.py:25-35):
    f is (2, N) unweighted; returns (u, v, p)."""
    def func(src, trg, f):
        return Stokes_Layer_Apply(src, trg, forces=f)
    return func
