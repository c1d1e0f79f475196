"""Multi-GPU evaluation of layer potentials: one process per GPU.

The target list shards (targets are independent units): rank r owns the
contiguous slice `target_slice(nt, r, world)`; every rank holds all sources.  Exchanges:

  * boundary densities (a few tens of KB): all-gathered before an apply when each rank
    holds only "its" part (`allgather_density`, bench.py), or — inside the solvers, where the
    annular solve of boundary i runs on rank i mod world — summed into a zero-filled buffer
    (`exchange_owned`: x + 0 + ... + 0 is exact, so an all-reduce IS the gather for
    disjoint owners);
  * results: every rank's slice into one preallocated buffer with `all_gather_into_tensor`
    (`ResultGather`; equal-size padded segments, nothing allocated per call).

RCCL over xGMI with the nccl backend; gloo in the CPU tests and in the one-GPU rehearsals
(device tensors then travel through host memory).

The compute function is injected so that the host logic can be exercised on CPU with
the gloo backend (tests/test_sharding.py uses the oracle as the compute function;
production passes `ipde_amd.layer_potentials.laplace_apply` & friends).
"""
import numpy as np

# logical payload of the collectives issued through this module (bytes this rank receives),
# so that a solve's exchange volume can be printed (tools/run_sharded_solve.py)
STATS = {"collectives": 0, "bytes": 0}


def reset_stats():
    STATS["collectives"] = 0
    STATS["bytes"] = 0


def _count(nbytes):
    STATS["collectives"] += 1
    STATS["bytes"] += int(nbytes)


# sums smaller than this many source-target pairs are evaluated by every rank itself (a
# collective costs ~50 us; 2.5e8 pairs is ~0.15 ms of kernel time)
MIN_PAIRS_TO_SHARD = 2.5e8


def target_slice(nt, rank, world):
    """Contiguous, balanced slice of range(nt) owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(nt), int(world))
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def shard_sizes(n, world):
    return [target_slice(n, r, world).stop - target_slice(n, r, world).start for r in range(world)]


def _dist_state(dist=None, group=None):
    """(dist module, rank, world) — (dist, 0, 1) when no process group is up"""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dist, 0, 1
    return dist, dist.get_rank(group), dist.get_world_size(group)


def is_distributed(dist=None, group=None):
    return _dist_state(dist, group)[2] > 1


class ResultGather:
    """All-gather of ragged contiguous shards of `ncomp` equally long 1-D results into
    full-length results on every rank.  Buffers are allocated once per (device, dtype):
    send (ncomp, m), receive (world, ncomp, m), m = ceil(nt / world); one
    `all_gather_into_tensor` per call whatever `ncomp` is."""

    def __init__(self, nt, ncomp=1, dist=None, group=None):
        self.dist, self.rank, self.world = _dist_state(dist, group)
        self.group = group
        self.nt, self.ncomp = int(nt), int(ncomp)
        self.sizes = shard_sizes(self.nt, self.world)
        self.m = max(self.sizes) if self.sizes else 0
        self._bufs = {}
        self._keep = {}

    def _buffers(self, device, dtype):
        import torch
        key = (str(device), dtype)
        b = self._bufs.get(key)
        if b is None:
            send = torch.zeros((self.ncomp, self.m), dtype=dtype, device=device)
            recv = torch.empty((self.world, self.ncomp, self.m), dtype=dtype, device=device)
            b = self._bufs[key] = (send, recv)
        return b

    def _keep_index(self, device):
        """positions of the real entries in a flattened (world, m) segment table"""
        import torch
        k = self._keep.get(str(device))
        if k is None:
            idx = np.concatenate([r * self.m + np.arange(s) for r, s in enumerate(self.sizes)]) \
                if self.nt else np.zeros(0, dtype=np.int64)
            k = self._keep[str(device)] = torch.as_tensor(idx, dtype=torch.int64, device=device)
        return k

    def __call__(self, parts):
        """parts: tuple of `ncomp` 1-D torch tensors (this rank's shard of each result)"""
        import torch
        if self.world == 1:
            return tuple(parts)
        t0 = parts[0]
        via_host = t0.is_cuda and self.dist.get_backend(self.group) == "gloo"
        dev = torch.device("cpu") if via_host else t0.device
        send, recv = self._buffers(dev, t0.dtype)
        n = self.sizes[self.rank]
        for c, p in enumerate(parts):
            send[c, :n].copy_(p)
        self.dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        _count(recv.numel() * recv.element_size())
        even = self.nt == self.m * self.world
        out = []
        for c in range(self.ncomp):
            seg = recv[:, c, :].reshape(-1)
            full = seg if even else seg.index_select(0, self._keep_index(dev))
            if full.untyped_storage().data_ptr() == recv.untyped_storage().data_ptr():
                full = full.clone()        # a view: the receive buffer is reused by the next call
            out.append(full.to(t0.device) if via_host else full)
        return tuple(out)


def allgather_density(local, n_total, dist=None, group=None):
    """All-gather a 1-D (or (c, n_local)) density shard into the full density.

    Shards may be ragged (n_total not divisible by the world size): they are padded to
    the largest shard for the collective and trimmed afterwards."""
    import torch
    dist, rank, world = _dist_state(dist, group)
    if world == 1:
        return local
    t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(local))
    lead = t.shape[:-1]
    ncomp = int(np.prod(lead)) if len(lead) else 1
    g = ResultGather(n_total, ncomp, dist, group)
    parts = t.reshape(ncomp, -1)
    full = torch.stack(list(g(tuple(parts[c] for c in range(ncomp))))).reshape(tuple(lead) + (int(n_total),))
    if isinstance(local, torch.Tensor):
        return full
    return full.cpu().numpy()


class ShardedLayerApply:
    """`Layer_Apply(src, trg, ch)` over N ranks, each rank holding a shard of the density.

    apply_fn(src, tx_local, ty_local, density_full) -> local result (array or tuple of
    arrays).  `trg` is sharded once at construction (the solver evaluates onto fixed
    target sets)."""

    def __init__(self, apply_fn, trg_x, trg_y, rank=None, world=None, dist=None, group=None,
                 wrap_targets=None):
        self.dist, r, w = _dist_state(dist, group)
        self.group = group
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.nt = int(len(trg_x))
        self.slice = target_slice(self.nt, self.rank, self.world)
        tx, ty = trg_x[self.slice], trg_y[self.slice]
        # wrap_targets: e.g. DeviceTargets, to keep the shard resident in HBM
        self.targets = wrap_targets(tx, ty) if wrap_targets else (tx, ty)
        self.apply_fn = apply_fn
        self._gathers = {}

    def __call__(self, src, density_local, n_density_total, gather_result=False):
        dens = allgather_density(density_local, n_density_total, self.dist, self.group)
        t = self.targets
        local = self.apply_fn(src, t, dens) if not isinstance(t, tuple) \
            else self.apply_fn(src, t[0], t[1], dens)
        if not gather_result or self.world == 1:
            return local
        return self.gather(local)

    def gather(self, local):
        """Assemble the full-length result on every rank (when the FFT owner needs the
        whole grid).  Tuples (Stokes u, v, p) travel in one collective."""
        import torch
        parts = local if isinstance(local, tuple) else (local,)
        as_t = tuple(p if isinstance(p, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(p))
                     for p in parts)
        g = self._gathers.get(len(parts))
        if g is None:
            g = self._gathers[len(parts)] = ResultGather(self.nt, len(parts), self.dist, self.group)
        full = g(as_t)
        full = tuple(f if isinstance(p, torch.Tensor) else f.cpu().numpy() for f, p in zip(full, parts))
        return full if isinstance(local, tuple) else full[0]


def gather_tail(parts, sl, nt, n_tail, dist=None, group=None):
    """The last `n_tail` entries of full-length results of which this rank holds the slice `sl`
    (`parts`: tuple of 1-D tensors, the local slice of each component) — on every rank.

    One all-reduce(SUM) of a zero-filled (ncomp, n_tail) buffer: every position is written by the
    one rank whose slice holds it, and x + 0 + ... + 0 == x exactly.  This is all a solve has to
    exchange of a sum onto grid_pnai when its grid part stays sharded (the tail of that list is
    the interface nodes, reference ebdy_collection.py:488-491): n_tail * 8 bytes per component
    instead of the whole list."""
    import torch
    dist, rank, world = _dist_state(dist, group)
    t0 = parts[0]
    lo = nt - n_tail                       # first list position of the tail
    a, b = max(sl.start, lo), max(sl.stop, lo)
    if world == 1:
        return tuple(p[lo - sl.start:] for p in parts)
    via_host = t0.is_cuda and dist.get_backend(group) == "gloo"
    dev = torch.device("cpu") if via_host else t0.device
    buf = torch.zeros((len(parts), n_tail), dtype=t0.dtype, device=dev)
    if b > a:
        for c, p in enumerate(parts):
            buf[c, a - lo:b - lo].copy_(p[a - sl.start:b - sl.start])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    _count(buf.numel() * buf.element_size())
    if via_host:
        buf = buf.to(t0.device)
    return tuple(buf[c] for c in range(len(parts)))


def gather_owned(values, owned, dist=None, group=None):
    """Replicate a vector of which every rank holds some entries complete (`owned`, a boolean mask;
    the masks of the ranks partition the index range): all-reduce(SUM) of the vector with the
    entries of other ranks zeroed — exact, one collective.  numpy in, numpy out."""
    import torch
    dist, rank, world = _dist_state(dist, group)
    v = np.where(owned, np.asarray(values, dtype=np.float64), 0.0)
    if world == 1:
        return v
    t = torch.from_numpy(np.ascontiguousarray(v))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    _count(t.numel() * 8)
    return t.numpy()


def global_max(x, dist=None, group=None):
    """max over the ranks of a host scalar (error norms of sharded results)"""
    import torch
    dist, rank, world = _dist_state(dist, group)
    if world == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class ShardedOut:
    """Result of a sharded evaluation that stays sharded: `local` (tensor or tuple of tensors) are
    the entries `slice` of the full-length result, `tail` the last `n_tail` entries of it, complete
    on every rank."""
    __slots__ = ("local", "slice", "tail", "nt")

    def __init__(self, local, sl, tail, nt):
        self.local, self.slice, self.tail, self.nt = local, sl, tail, nt


def make_sharded_evaluator(layer_apply, targets, wrap_targets, dist=None, group=None,
                           min_pairs=0.0):
    """`evaluate(sources, density)` onto a FIXED target set, every rank holding the full
    (replicated) density.

    Single process: one dense sum onto the resident target set.  Under torch.distributed
    (one process per GPU) each rank evaluates its contiguous slice of the targets and the
    slices are all-gathered into preallocated buffers (`ResultGather`).  Used for the
    solvers' Grid_Evaluator onto grid_pnai (reference multi_boundary/scalar.py:63-71), for
    the example-level correction sum onto grid_and_radial_pts (reference
    examples/interior_poisson.py:84-92) and for correct()'s radial sums (reference
    internals/scalar.py:95-116).  `layer_apply(src, trg, density)` returns a device tensor
    or a tuple of them (Stokes u, v, p).  Sums with fewer than `min_pairs` source-target
    pairs are evaluated by every rank in full (no collective)."""
    dist, rank, world = _dist_state(dist, group)
    nt = int(len(targets.x))
    state = {}

    def resident():
        if "all" not in state:
            state["all"] = wrap_targets(targets.x, targets.y)
        return state["all"]

    def _sharded_single(sources, density, n_tail):
        out = layer_apply(sources, resident(), density)
        parts = out if isinstance(out, tuple) else (out,)
        tail = tuple(p[nt - n_tail:] for p in parts)
        return ShardedOut(out, slice(0, nt), tail if isinstance(out, tuple) else tail[0], nt)

    if world == 1:
        single = lambda sources, density: layer_apply(sources, resident(), density)
        single.prepare = lambda n_sources=None: resident()
        single.sharded = _sharded_single
        return single
    sl = target_slice(nt, rank, world)

    def local():
        if "local" not in state:
            state["local"] = wrap_targets(targets.x[sl], targets.y[sl])
        return state["local"]

    def evaluator(sources, density):
        if nt * float(sources.N) < min_pairs:
            return layer_apply(sources, resident(), density)
        out = layer_apply(sources, local(), density)
        parts = out if isinstance(out, tuple) else (out,)
        g = state.get(len(parts))
        if g is None:
            g = state[len(parts)] = ResultGather(nt, len(parts), dist, group)
        full = g(parts)
        return full if isinstance(out, tuple) else full[0]
    def sharded(sources, density, n_tail):
        """the same sum, the result left sharded: this rank's slice and the gathered tail"""
        if nt * float(sources.N) < min_pairs:
            return _sharded_single(sources, density, n_tail)
        out = layer_apply(sources, local(), density)
        parts = out if isinstance(out, tuple) else (out,)
        tail = gather_tail(parts, sl, nt, n_tail, dist, group)
        return ShardedOut(out, sl, tail if isinstance(out, tuple) else tail[0], nt)
    evaluator.sharded = sharded
    # make the target set resident ahead of the first evaluation (the solvers' set-up)
    evaluator.prepare = lambda n_sources=None: (resident() if n_sources is not None and nt * float(n_sources) < min_pairs
                                                else local())
    return evaluator


def make_pnai_evaluator(layer_apply, sources, targets, wrap_targets, dist=None, group=None):
    """The solvers' `Grid_Evaluator(density)` onto grid_pnai: `make_sharded_evaluator`
    bound to the solver's grid sources (BASELINE configs 4 and 5)."""
    ev = make_sharded_evaluator(layer_apply, targets, wrap_targets, dist, group)
    on_pnai = lambda density: ev(sources, density)
    on_pnai.prepare = lambda: ev.prepare(sources.N)
    # the sum with its grid part left sharded: n_tail = number of interface nodes at the list's end
    on_pnai.sharded = lambda density, n_tail: ev.sharded(sources, density, n_tail)
    return on_pnai


# -- per-boundary work distributed over ranks ------------------------------------------------
def owner_of(i, world):
    """rank that runs the annular solve / QFS solves of boundary i"""
    return i % world


def exchange_owned(values, shapes, dist=None, group=None, device=None, extra=None, as_tensors=None):
    """Every rank contributes the arrays of the boundaries it owns (`values[i]`, shape
    `shapes[i]`, ignored where `owner_of(i) != rank`) and receives those of all boundaries.

    One all-reduce(SUM) of a zero-filled flat buffer: each position is written by exactly
    one rank, and x + 0 + ... + 0 == x exactly, so this is a gather with no per-boundary
    collectives and no ragged bookkeeping.  `extra`: an optional small vector with one entry
    per boundary (zero for the boundaries of other ranks) that rides along.

    Owned values that are torch tensors (the solvers' device-resident flow) are packed into a
    flat buffer ON THEIR DEVICE and come back as device tensors: with the nccl backend nothing
    passes through host memory; gloo (CPU tests, one-GPU rehearsals) moves the flat buffer through
    the host.  numpy values take the host path and come back as numpy arrays.  `device`: where the
    collective of the numpy path runs (a CUDA device for nccl).  `as_tensors`: the kind of the result,
    stated by the caller — True: device tensors on `device` WHATEVER this rank owns (a rank that owns no
    boundary, world > number of boundaries, passes only None and must still come back with what the
    other ranks get); False: numpy; None (legacy): inferred from the owned values.  Returns the list of
    arrays (and the summed `extra`, as numpy, when given)."""
    import torch
    dist, rank, world = _dist_state(dist, group)
    if as_tensors is None:
        as_tensors = any(isinstance(v, torch.Tensor) for v in values if v is not None)
    if world == 1:
        if as_tensors:
            vals = [torch.as_tensor(v, device=device if not isinstance(v, torch.Tensor) else None).reshape(s)
                    for v, s in zip(values, shapes)]
        else:
            vals = [np.asarray(v, dtype=np.float64).reshape(s) for v, s in zip(values, shapes)]
        return vals if extra is None else (vals, np.asarray(extra, dtype=np.float64))
    sizes = [int(np.prod(s)) for s in shapes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_extra = 0 if extra is None else len(extra)
    if as_tensors:
        tdev = next((v.device for v in values if isinstance(v, torch.Tensor)), None)
        if tdev is None:        # nothing owned here: the caller's device
            tdev = torch.device(device) if device is not None else torch.device("cpu")
        flat = torch.zeros(int(off[-1]) + n_extra, dtype=torch.float64, device=tdev)
        for i, (v, s) in enumerate(zip(values, shapes)):
            if owner_of(i, world) == rank:
                flat[off[i]:off[i + 1]].copy_(torch.as_tensor(v, device=tdev).reshape(-1))
        if n_extra:
            flat[off[-1]:].copy_(torch.as_tensor(np.asarray(extra, dtype=np.float64)))
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        _count(flat.numel() * 8)
        vals = [flat[off[i]:off[i + 1]].reshape(s) for i, s in enumerate(shapes)]
        return vals if extra is None else (vals, flat[off[-1]:].cpu().numpy())
    flat = np.zeros(int(off[-1]) + n_extra)
    for i, (v, s) in enumerate(zip(values, shapes)):
        if owner_of(i, world) == rank:
            flat[off[i]:off[i + 1]] = np.asarray(v, dtype=np.float64).reshape(-1)
    if n_extra:
        flat[off[-1]:] = np.asarray(extra, dtype=np.float64)
    on_dev = dist.get_backend(group) != "gloo" and device is not None
    t = torch.as_tensor(flat, device=device) if on_dev else torch.from_numpy(flat)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    _count(flat.size * 8)
    flat = t.cpu().numpy() if on_dev else flat
    vals = [flat[off[i]:off[i + 1]].reshape(s).copy() for i, s in enumerate(shapes)]
    return vals if extra is None else (vals, flat[off[-1]:].copy())
