"""Host <-> HBM movement of the grid-sized halves of EmbeddedFunctions (reference data
layout: ipde/embedded_function.py:16-113).  The solvers take and return numpy-backed
containers, as the reference does; what is tuned here is how their physical grid values
(17.6 MB at 2048^2) cross PCIe:

  upload    the caller's array is pageable.  Worker threads stage it into a pinned buffer in
            chunks (numpy's contiguous copy releases the GIL) and the DMA of chunk k is queued
            as soon as chunk k is staged, so staging and DMA overlap;
  results   are built directly over pinned memory from torch's caching host allocator: the
            device->host copy lands in the array the caller receives — no second host copy.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .embedded_function import EmbeddedFunction

STAGE_THREADS = 4
_MIN_CHUNK = 1 << 18      # doubles; below this a chunk's thread hand-off costs more than its copy
_pool = None


def _stage_pool():
    global _pool
    if _pool is None:
        _pool = ThreadPoolExecutor(STAGE_THREADS, thread_name_prefix="ipde-stage")
    return _pool


def upload(dst, src, pin):
    """dst (device, 1-D fp64, n) <- src (numpy, n) through the pinned tensor `pin` (n).
    Asynchronous on the current stream: `pin` must not be rewritten before the stream has
    passed this point (the solvers synchronise before they return)."""
    src = np.asarray(src, dtype=np.float64).reshape(-1)
    n = src.shape[0]
    assert dst.numel() == n and pin.numel() == n
    host = pin.numpy()
    nchunk = max(1, min(2 * STAGE_THREADS, n // _MIN_CHUNK))
    if nchunk == 1:
        host[:] = src
        dst.copy_(pin, non_blocking=True)
        return
    edges = [n * k // nchunk for k in range(nchunk + 1)]
    stage = lambda k: np.copyto(host[edges[k]:edges[k + 1]], src[edges[k]:edges[k + 1]])
    futures = [_stage_pool().submit(stage, k) for k in range(nchunk)]
    for k, fut in enumerate(futures):
        fut.result()
        dst[edges[k]:edges[k + 1]].copy_(pin[edges[k]:edges[k + 1]], non_blocking=True)


def pinned_function(ebdyc):
    """(EmbeddedFunction whose storage is one pinned host block, the block as a tensor).
    The numpy view keeps the tensor alive; when the caller drops the function the block
    goes back to torch's caching host allocator and the next solve reuses it."""
    gn = ebdyc.grid_phys.N
    rn = int(sum(int(np.prod(ebdy.radial_shape)) for ebdy in ebdyc))
    block = torch.empty(gn + rn, dtype=torch.float64, pin_memory=True)
    return EmbeddedFunction(ebdyc, array=block.numpy()), block
