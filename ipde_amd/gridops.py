"""Grid <-> list moves of the multi-boundary solvers as library calls (csrc/geometry.hip:
ipde_grid_scatter / ipde_grid_add_at / ipde_grid_gather) — what the reference does with numpy
masks (ipde/embedded_function.py:105-113,135-138, ipde/solvers/multi_boundary/scalar.py:72-117).
torch provides the device memory only."""
import torch

from .device import get_context, ptr


def _check(idx, *vals):
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    for v in vals:
        assert v.dtype == torch.float64 and v.is_contiguous()


def scatter(idx, src, ngrid, scale=None, ctx=None):
    """zero grid of `ngrid` doubles with out[idx] = src (* scale[idx]); returns the flat grid"""
    ctx = ctx or get_context()
    out = torch.empty(ngrid, dtype=torch.float64, device=src.device)
    scale = None if scale is None else scale.reshape(-1)
    _check(idx, src, *([] if scale is None else [scale]))
    ctx.check(ctx.lib.ipde_grid_scatter(ctx.handle, ngrid, idx.shape[0], ptr(idx), ptr(src), ptr(scale), ptr(out)))
    return out


def add_at(idx, src, out, ctx=None):
    """out[idx] += src, in place (idx without repetitions)"""
    ctx = ctx or get_context()
    _check(idx, src, out)
    ctx.check(ctx.lib.ipde_grid_add_at(ctx.handle, idx.shape[0], ptr(idx), ptr(src), ptr(out)))
    return out


def gather(idx, grid, ctx=None, out=None):
    """grid[idx] as a new tensor (or into `out`, a contiguous fp64 tensor of idx's length)"""
    ctx = ctx or get_context()
    if out is None:
        out = torch.empty(idx.shape[0], dtype=torch.float64, device=grid.device)
    elif not (out.is_contiguous() and out.dtype == torch.float64 and out.numel() == idx.shape[0]
              and out.device == grid.device):
        raise ValueError("gridops.gather: `out` must be a contiguous fp64 tensor of len(idx) on the grid's device")
    _check(idx, grid)
    ctx.check(ctx.lib.ipde_grid_gather(ctx.handle, idx.shape[0], ptr(idx), ptr(grid), ptr(out)))
    return out


def rows(tensors):
    """The 1-D device tensors as the rows of one (len, n) tensor — by device-to-device copies (no torch
    kernel: `torch.stack` loads its code object at first use, 9 ms of a first solve, and costs a launch
    each time)."""
    first = tensors[0]
    out = torch.empty((len(tensors), first.shape[0]), dtype=first.dtype, device=first.device)
    for i, t in enumerate(tensors):
        out[i].copy_(t.reshape(-1))
    return out


def concat(tensors):
    """1-D device tensors end to end (one of them: itself, no copy)."""
    if len(tensors) == 1:
        return tensors[0]
    out = torch.empty(sum(int(t.shape[0]) for t in tensors), dtype=tensors[0].dtype, device=tensors[0].device)
    a = 0
    for t in tensors:
        out[a:a + t.shape[0]].copy_(t)
        a += int(t.shape[0])
    return out
