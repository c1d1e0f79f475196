"""Host <-> HBM movement of the grid-sized halves of EmbeddedFunctions (reference data
layout: ipde/embedded_function.py:16-113).  The solvers take and return numpy-backed
containers, as the reference does; what is tuned here is how their physical grid values
(17.6 MB at 2048^2) cross PCIe:

  upload    the caller's array is pageable.  Worker threads stage it into a pinned buffer in
            chunks (numpy's contiguous copy releases the GIL) and the DMA of chunk k is queued
            as soon as chunk k is staged, so staging and DMA overlap;
  results   are built directly over pinned memory from torch's caching host allocator: the
            device->host copy lands in the array the caller receives — no second host copy.

`DeviceFunction` is the same container resident in HBM: a caller that produces the right-hand side
on the device, or feeds one solve's answer into the next (a time-stepper), hands the scalar solvers
one and gets one back — nothing crosses PCIe (2048^2 Poisson: ~0.8 ms of a 7 ms solve).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .embedded_function import EmbeddedFunction

import os as _os

STAGE_THREADS = int(_os.environ.get("IPDE_STAGE_THREADS", "4"))
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


def add_from_device(dst, src, pin=None):
    """dst (numpy, 1-D fp64, n) += src (device tensor, n): the device->host copy goes chunk by chunk into a
    pinned buffer and worker threads add each chunk onto `dst` as soon as it has landed (numpy's add
    releases the GIL), so the DMA and the host's additions overlap — the host-container form of the
    examples' homogeneous correction (17.6 MB at 2048^2: one blocking copy + one numpy add took ~2 ms).
    Complete on return."""
    dst = np.asarray(dst).reshape(-1)
    n = dst.shape[0]
    assert src.numel() == n and dst.dtype == np.float64 and dst.flags.c_contiguous
    src = src.reshape(-1)
    if pin is None or pin.numel() < n:
        pin = torch.empty(n, dtype=torch.float64, pin_memory=True)
    host = pin.numpy()
    nchunk = max(1, min(4 * STAGE_THREADS, n // _MIN_CHUNK))
    edges = [n * k // nchunk for k in range(nchunk + 1)]
    events = []
    for k in range(nchunk):
        pin[edges[k]:edges[k + 1]].copy_(src[edges[k]:edges[k + 1]], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        events.append(ev)

    def add(k):
        events[k].synchronize()
        a, b = edges[k], edges[k + 1]
        np.add(dst[a:b], host[a:b], out=dst[a:b])
    if nchunk == 1:
        add(0)
        return pin
    for fut in [_stage_pool().submit(add, k) for k in range(nchunk)]:
        fut.result()
    return pin


def pinned_function(ebdyc):
    """(EmbeddedFunction whose storage is one pinned host block, the block as a tensor).
    The numpy view keeps the tensor alive; when the caller drops the function the block
    goes back to torch's caching host allocator and the next solve reuses it."""
    gn = ebdyc.grid_phys.N
    rn = int(sum(int(np.prod(ebdy.radial_shape)) for ebdy in ebdyc))
    block = torch.empty(gn + rn, dtype=torch.float64, pin_memory=True)
    return EmbeddedFunction(ebdyc, array=block.numpy()), block


class DeviceFunction(object):
    """The values of an EmbeddedFunction in HBM: `data`, one fp64 device tensor in the
    EmbeddedFunction's own storage order — the physical grid values, then every boundary's radial
    block (ipde/embedded_function.py:16-113).  `ScalarSolver.__call__` takes either kind and answers
    in the kind it was given."""

    def __init__(self, ebdyc, data=None, device=None):
        self.ebdyc = ebdyc
        self.n_grid = int(ebdyc.grid_phys.N)
        self.radial_shapes = [tuple(int(s) for s in ebdy.radial_shape) for ebdy in ebdyc]
        self.radial_slices, start = [], self.n_grid
        for sh in self.radial_shapes:
            self.radial_slices.append(slice(start, start + sh[0] * sh[1]))
            start += sh[0] * sh[1]
        self.n = start
        if data is None:
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device())
            data = torch.empty(self.n, dtype=torch.float64, device=device)
        if not (isinstance(data, torch.Tensor) and data.is_cuda and data.dtype == torch.float64
                and data.dim() == 1 and data.numel() == self.n and data.is_contiguous()):
            raise ValueError("DeviceFunction: data must be a contiguous fp64 device tensor of %d values" % self.n)
        self.data = data

    @classmethod
    def from_host(cls, f, device=None):
        """EmbeddedFunction -> DeviceFunction (one host->device copy)."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        ebdyc = f._ebdyc_test()
        return cls(ebdyc, torch.as_tensor(np.ascontiguousarray(np.asarray(f), dtype=np.float64).reshape(-1),
                                          device=device))

    def to_host(self):
        """-> EmbeddedFunction over pinned memory (one device->host copy, complete on return)."""
        ue, block = pinned_function(self.ebdyc)
        block.copy_(self.data, non_blocking=False)
        return ue

    @property
    def grid_values(self):
        return self.data[:self.n_grid]

    def get_radial_value_list(self):
        return [self.data[sl].view(sh) for sl, sh in zip(self.radial_slices, self.radial_shapes)]
