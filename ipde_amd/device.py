"""Context handling and array marshalling for the C ABI.

Arrays may be numpy (host; the library stages them through its own device
workspace) or torch CUDA tensors (device; zero-copy).  torch is used for device
memory only — every computation on this path is a kernel of libipde_hip.so.
"""
import atexit
import ctypes
import os
import threading
import time
import weakref

import numpy as np
import torch

from . import _lib

_contexts = {}
_lock = threading.Lock()


class Context:
    """One per GPU per process (wraps ipde_ctx)."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.IpdeHipError("no GPU visible: ipde_amd has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        h = ctypes.c_void_p()
        _lib.check(self.lib.ipde_ctx_create(self.device, ctypes.byref(h)))
        self.handle = h
        self._plans = {}
        # IPDE_HIP_OPTIONS="name=value,...": tuning knobs of ipde_ctx_set_option for every context
        # of the process (A/B measurements with unmodified scripts)
        for item in filter(None, os.environ.get("IPDE_HIP_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            self.set_option(name.strip(), int(value))
        # objects holding library handles created on this context (annular solvers, Ewald
        # cores): the context releases them before it goes, whatever order the garbage
        # collector finds them in
        self._children = weakref.WeakSet()

    _private = False

    def adopt(self, owner):
        """register an object with a `_release()` method that frees its library handle"""
        self._children.add(owner)

    def _release_children(self):
        for c in list(self._children):
            try:
                c._release()
            except Exception:
                pass

    def __del__(self):
        # private contexts (private_context()) go with their owner — the owner holds the only
        # reference, so its own handles are destroyed first; the process-wide ones live on
        try:
            import sys
            if self._private and self.handle and not sys.is_finalizing():
                self._release_children()
                h, self.handle = self.handle, None
                self.lib.ipde_ctx_destroy(h)
        except Exception:
            pass

    # -- misc ---------------------------------------------------------------
    def check(self, status, allow=()):
        return _lib.check(status, self.handle, allow)

    def sync(self):
        self.check(self.lib.ipde_ctx_sync(self.handle))

    def use_torch_stream(self, stream=None):
        """Queue all subsequent work on a torch stream (default: torch's current)."""
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        self.check(self.lib.ipde_ctx_set_stream(self.handle, ctypes.c_void_p(s.cuda_stream)))

    def use_legacy_default_stream(self):
        """Queue all subsequent work on the legacy default stream — the one torch's default stream
        is.  The process-wide context does (see get_context): a solve interleaves the library's
        kernels with a dozen small torch operations, and with a stream of its own every change of
        stream was a cross-queue dependency, ~17 us of idle GPU each (2048^2 Poisson: 7.3 -> 6.7 ms)."""
        self.check(self.lib.ipde_ctx_use_legacy_stream(self.handle))

    def use_background_stream(self):
        """The context's own stream becomes a non-blocking one of the device's lowest priority; returns it as a
        torch stream (for the events that order it against the others)."""
        self.check(self.lib.ipde_ctx_use_background_stream(self.handle))
        return torch.cuda.ExternalStream(int(self.lib.ipde_ctx_get_stream(self.handle)), device=self.device)

    def set_option(self, name, value):
        self.check(self.lib.ipde_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_int()
        self.check(self.lib.ipde_ctx_get_option(self.handle, name.encode(), ctypes.byref(v)))
        return v.value

    def enable_timing(self, on=True):
        self.check(self.lib.ipde_ctx_enable_timing(self.handle, int(bool(on))))

    def last_kernel_ms(self):
        ms = ctypes.c_double()
        self.check(self.lib.ipde_ctx_last_kernel_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    def kernel_ms_history(self, cap=256):
        """Durations (ms) of the dominant kernel of every apply since enable_timing(True), oldest
        first (at most the last 256): recorded without a host sync, resolved here."""
        buf = (ctypes.c_double * cap)()
        n = ctypes.c_int()
        self.check(self.lib.ipde_ctx_kernel_ms_history(self.handle, buf, cap, ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]

    def torch_device(self):
        return torch.device("cuda", self.device)

    def close(self):
        if self.handle:
            self._release_children()
            for p in list(self._plans.values()):
                p.close()
            self._plans.clear()
            self.lib.ipde_ctx_destroy(self.handle)
            self.handle = None


# every tuning knob of a context (ipde_ctx_set_option; csrc/ctx.hip option_slot)
OPTION_NAMES = ("laplace_variant", "stokes_variant", "modhelm_variant", "dense_pairs", "dense_persistent",
                "gmres_graphs", "gmres_persistent", "gmres_lookahead", "gmres_fused_scale", "annular_fused_fft",
                "annular_grouped", "fft2d", "interp_shifted", "interp_band", "timing_split")


def snapshot_options():
    """{context: {option: value}} of the shared contexts (the knobs are per-context state a caller
    may have changed): what tests/conftest.py restores after every GPU test, whatever the test did."""
    return {ctx: {n: ctx.get_option(n) for n in OPTION_NAMES} for ctx in list(_contexts.values()) if ctx.handle}


def restore_options(snapshot):
    for ctx, opts in snapshot.items():
        if ctx.handle:
            for n, v in opts.items():
                if ctx.get_option(n) != v:
                    ctx.set_option(n, v)


def get_context(device=None):
    """Process-wide context of a device (created on first use)."""
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.IpdeHipError("no GPU visible: ipde_amd has no CPU fallback")
        device = torch.cuda.current_device()
    created = False
    with _lock:
        ctx = _contexts.get(device)
        if ctx is None or not ctx.handle:
            ctx = Context(device)
            if os.environ.get("IPDE_CTX_OWN_STREAM", "0") != "1":
                ctx.use_legacy_default_stream()
            _contexts[device] = ctx
            created = True
    if created and device == torch.cuda.current_device():
        prewarm()            # the host BLAS's thread pool comes up beside whatever the caller does next
    return ctx


def private_context(device=None):
    """A context of its own (own stream, own work buffers and FFT plans) — what an object needs
    to be driven from a separate host thread concurrently with the others (the annular
    solvers of a multiply connected domain, solvers/multi_boundary/vector.py)."""
    ctx = Context(device)
    ctx._private = True
    return ctx


_warm = {"thread": None, "queue": None, "keys": set()}
PREWARM_TIMEOUT_S = 300.0


def _warm_worker():
    q = _warm["queue"]
    while True:
        fn = q.get()
        try:
            fn()
        except Exception:       # best effort; real errors surface in the real calls
            pass
        finally:
            q.task_done()


PREPARE_INTERP = True


def prewarm(grid_shape=None, h=None, wait=False, fft1=()):
    """Pay the one-time library costs of a first solve in ONE background thread while the
    host does geometry set-up: the host BLAS's thread pool (numpy's first LAPACK call, ~0.1 s
    on a many-core box), the rocFFT kernels of the batched 1-D transforms `fft1` =
    ((batch, n), ...) (annular solver, radial interpolation) and, if the grid is given, of
    the 2-D grid solve (~1.6 s of run-time compilation at 2048^2 without the shipped kernel
    cache).  Jobs run strictly one after the other, and every
    FFT entry point of the package joins the thread first (`prewarm_wait`): rocFFT's plan
    creation / run-time compilation is never entered from two threads at once (doing so
    crashed a two-rank rehearsal with SIGSEGV).  No-op without a GPU.
    (rocBLAS / rocSOLVER are no longer loaded here: since the QFS systems are factored,
    substituted and multiplied by the library's own kernels the scalar solvers never call
    them, and every FFT entry point used to wait for their ~0.5 s cold load.)"""
    import queue
    import threading
    if not torch.cuda.is_available():
        return
    dev = torch.cuda.current_device()

    def libs():
        e = np.eye(8)
        np.linalg.inv(e + e @ e)
        # torch loads the code object of every kernel type at its first launch (~10 ms apiece in a
        # fresh process): a miniature of the QFS matrix assembly (dense_forms) touches the types
        # the set-up uses, here instead of on the main thread
        try:
            from . import dense_forms as df
            from .pybie2d_compat import Global_Smooth_Boundary, star
            torch.cuda.set_device(dev)
            d = torch.device("cuda", dev)
            b = Global_Smooth_Boundary(c=star(32, a=0.2, f=5))
            s2 = Global_Smooth_Boundary(c=1.1 * star(32, a=0.2, f=5))
            m = df.laplace_form(s2, b, d, ifcharge=True, ifdipole=True) \
                + df.laplace_singular_form(b, d, ifcharge=True, ifdipole=True)
            m = m + torch.eye(32, dtype=torch.float64, device=d)
            pad = torch.eye(128, dtype=torch.float64, device=d)
            pad[:32, :32] = m
            pad.view(2, 64, 2, 64).permute(0, 2, 3, 1).contiguous().sum().item()
        except Exception:
            pass

    def plans():
        torch.cuda.set_device(dev)
        from .spectral import get_plan
        p = get_plan(grid_shape[0], grid_shape[1], h[0], h[1], get_context(dev))
        # the interface interpolation's state as well where the plan has that path (the spectral
        # solvers' default; ~60 ms and 0.8 GB at 2048^2): a first solve would build it otherwise
        if PREPARE_INTERP and p.keep_spectrum(False):
            p.prepare_interp()

    def fft1_plans():
        ctx = get_context(dev)
        for batch, n in fft1:
            ctx.lib.ipde_fft1_prepare(ctx.handle, int(batch), int(n))
    jobs = [("libs", libs)]
    if fft1:
        jobs.append((("fft1", tuple(fft1)), fft1_plans))
    if grid_shape is not None:
        jobs.append((("plan", tuple(grid_shape)), plans))
    _enqueue(jobs)
    if wait:
        prewarm_wait()


def _enqueue(jobs):
    import queue
    import threading
    with _lock:
        if _warm["queue"] is None:
            _warm["queue"] = queue.Queue()
            _warm["thread"] = threading.Thread(target=_warm_worker, name="ipde-prewarm", daemon=True)
            _warm["thread"].start()
        for key, fn in jobs:
            if key not in _warm["keys"]:
                _warm["keys"].add(key)
                _warm["queue"].put(fn)


def prewarm_submit(key, fn):
    """One more job for the warm-up thread (run once per key, after the jobs already queued; every
    FFT entry point joins the thread first).  The solvers use it for one-time device state a first
    solve would otherwise build: the interface interpolation's fine-grid plan (~60 ms at 2048^2)."""
    if not torch.cuda.is_available():
        return
    dev = torch.cuda.current_device()

    def job():
        torch.cuda.set_device(dev)
        fn()
    _enqueue([(key, job)])


def _drain_at_exit():
    """A process that ends while the warm-up thread is inside torch / HIP aborts during interpreter
    shutdown (daemon threads are not joined): let the queue drain first — its jobs are fractions of
    a second."""
    q = _warm["queue"]
    if q is None:
        return
    deadline = time.monotonic() + 30.0
    with q.all_tasks_done:
        while q.unfinished_tasks and time.monotonic() < deadline:
            q.all_tasks_done.wait(0.05)
        stuck = q.unfinished_tasks
    if stuck:
        # carrying on would run the interpreter's teardown under a thread that is still inside
        # torch / HIP: the abort this hook exists to prevent.  End the process here, visibly.
        import sys
        sys.stderr.write("ipde_amd: %d warm-up job(s) still running 30 s after exit was requested; "
                         "ending the process without interpreter teardown (exit status 70)\n" % stuck)
        sys.stderr.flush()
        try:
            sys.stdout.flush()
        except Exception:
            pass
        os._exit(70)


atexit.register(_drain_at_exit)


def prewarm_wait():
    """Block until the warm-up thread is idle (called by every FFT entry point)."""
    import threading
    q = _warm["queue"]
    if q is None or threading.current_thread() is _warm["thread"]:
        return
    # queue.join() without the possibility of waiting forever: the jobs are seconds of
    # library loading / kernel compilation.  Carrying on while the thread is still inside
    # rocFFT would enter plan creation from two threads (the condition this join exists to
    # exclude), so a timeout is an error.
    deadline = time.monotonic() + PREWARM_TIMEOUT_S
    with q.all_tasks_done:
        while q.unfinished_tasks:
            remaining = deadline - time.monotonic()
            if remaining <= 0.0:
                raise _lib.IpdeHipError("the library warm-up thread did not finish within %.0f s"
                                        % PREWARM_TIMEOUT_S)
            q.all_tasks_done.wait(remaining)


# ---------------------------------------------------------------------------
def is_device_array(a):
    return isinstance(a, torch.Tensor) and a.is_cuda


def location_of(*arrays):
    """IPDE_DEVICE if the (non-None) arrays are torch CUDA tensors, IPDE_HOST if they
    are host arrays; mixing is an error."""
    dev = [is_device_array(a) for a in arrays if a is not None]
    if dev and all(dev):
        return _lib.IPDE_DEVICE
    if any(dev):
        raise ValueError("mixing host arrays and device tensors in one call")
    return _lib.IPDE_HOST


def as_f64(a, loc):
    """Contiguous fp64 view/copy of `a` in the representation matching `loc`."""
    if a is None:
        return None
    if loc == _lib.IPDE_DEVICE:
        if a.dtype != torch.float64 and a.dtype != torch.complex128:
            a = a.to(torch.float64)
        return a.contiguous()
    a = np.asarray(a)
    if a.dtype != np.float64 and a.dtype != np.complex128:
        a = a.astype(np.float64)
    return np.ascontiguousarray(a)


def ptr(a):
    """Raw address for the C ABI (None -> NULL)."""
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    return ctypes.c_void_p(a.ctypes.data)


def empty_like_loc(shape, loc, ctx, dtype="f8"):
    if loc == _lib.IPDE_DEVICE:
        td = torch.float64 if dtype == "f8" else torch.complex128
        return torch.empty(shape, dtype=td, device=ctx.torch_device())
    return np.empty(shape, dtype=np.float64 if dtype == "f8" else np.complex128)


def to_device(a, ctx=None):
    """numpy -> torch CUDA tensor (fp64 / complex128), device memory only."""
    ctx = ctx or get_context()
    if is_device_array(a):
        return a
    return torch.as_tensor(np.ascontiguousarray(a), device=ctx.torch_device())
