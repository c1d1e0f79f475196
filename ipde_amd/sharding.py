"""Multi-GPU evaluation of layer potentials: one process per GPU.

The target list shards (targets are independent units): rank r owns the
contiguous slice `target_slice(nt, r, world)`; every rank holds all sources.  The one
exchange is the boundary density: each rank owns the densities of "its" boundary
nodes (the annular solves that produce them run one boundary per GPU) and they are
all-gathered (RCCL over xGMI; gloo in the CPU tests) before every apply — a few tens
of KB, latency bound.  Results stay sharded unless `gather_result` is asked for.

The compute function is injected so that the host logic can be exercised on CPU with
the gloo backend (tests/test_sharding.py uses the oracle as the compute function;
production passes `ipde_amd.layer_potentials.laplace_apply` & friends).
"""
import numpy as np


def target_slice(nt, rank, world):
    """Contiguous, balanced slice of range(nt) owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(nt), int(world))
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def shard_sizes(n, world):
    return [target_slice(n, r, world).stop - target_slice(n, r, world).start for r in range(world)]


def allgather_density(local, n_total, dist=None, group=None):
    """All-gather a 1-D (or (c, n_local)) density shard into the full density.

    Shards may be ragged (n_total not divisible by the world size): they are padded to
    the largest shard for the collective and trimmed afterwards."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(local))
    lead = tuple(t.shape[:-1])
    m = max(sizes)
    pad = torch.zeros(lead + (m,), dtype=t.dtype, device=t.device)
    pad[..., :t.shape[-1]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    full = torch.cat([o[..., :s] for o, s in zip(out, sizes)], dim=-1)
    if isinstance(local, torch.Tensor):
        return full
    return full.cpu().numpy()


class ShardedLayerApply:
    """`Layer_Apply(src, trg, ch)` over N ranks.

    apply_fn(src, tx_local, ty_local, density_full) -> local result (array or tuple of
    arrays).  `trg` is sharded once at construction (the solver evaluates onto fixed
    target sets)."""

    def __init__(self, apply_fn, trg_x, trg_y, rank=None, world=None, dist=None, group=None,
                 wrap_targets=None):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank(group) if inited else 0) if rank is None else rank
        self.world = (dist.get_world_size(group) if inited else 1) if world is None else world
        self.nt = int(len(trg_x))
        self.slice = target_slice(self.nt, self.rank, self.world)
        tx, ty = trg_x[self.slice], trg_y[self.slice]
        # wrap_targets: e.g. DeviceTargets, to keep the shard resident in HBM
        self.targets = wrap_targets(tx, ty) if wrap_targets else (tx, ty)
        self.apply_fn = apply_fn

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
        whole grid).  Tuples (Stokes u, v, p) are gathered component-wise."""
        if isinstance(local, tuple):
            return tuple(self.gather(c) for c in local)
        import torch
        t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(local))
        sizes = shard_sizes(self.nt, self.world)
        m = max(sizes)
        pad = torch.zeros(m, dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad, group=self.group)
        full = torch.cat([o[:s] for o, s in zip(out, sizes)])
        return full if isinstance(local, torch.Tensor) else full.cpu().numpy()


def _all_gather_padded(t, sizes, dist, group=None):
    """all_gather of ragged 1-D shards of a torch tensor (device tensors over RCCL; with
    the gloo backend, which the CPU / one-GPU rehearsals use, through host memory)."""
    import torch
    m = max(sizes)
    backend = dist.get_backend(group)
    via_host = t.is_cuda and backend == "gloo"
    src = t.cpu() if via_host else t
    pad = torch.zeros(m, dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    out = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(out, pad, group=group)
    full = torch.cat([o[:s] for o, s in zip(out, sizes)])
    return full.to(t.device) if via_host else full


def make_pnai_evaluator(layer_apply, sources, targets, wrap_targets, dist=None, group=None):
    """The solvers' `Grid_Evaluator(density)` onto a fixed target set (grid_pnai).

    Single process: one dense sum onto the resident target set.  Under
    torch.distributed (one process per GPU, every rank holding the replicated solver
    state and hence the full density) each rank evaluates its contiguous slice of the
    targets and the slices are all-gathered — the only collective of a solve
    (BASELINE configs 4 and 5).  `layer_apply(src, trg, density)` returns a device
    tensor or a tuple of them (Stokes u, v, p)."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        resident = wrap_targets(targets.x, targets.y)
        return lambda density: layer_apply(sources, resident, density)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nt = int(len(targets.x))
    sl = target_slice(nt, rank, world)
    local = wrap_targets(targets.x[sl], targets.y[sl])
    sizes = shard_sizes(nt, world)

    def evaluator(density):
        out = layer_apply(sources, local, density)
        if isinstance(out, tuple):
            return tuple(_all_gather_padded(o, sizes, dist, group) for o in out)
        return _all_gather_padded(out, sizes, dist, group)
    return evaluator
