"""Quadrature by fundamental solutions (QFS) for the interface/boundary potentials —
own restatement of what the reference obtains from the absent `qfs` package
(`Laplace_QFS`, `Modified_Helmholtz_QFS`, `QFS_Evaluator`; reference
ipde/solvers/internals/poisson.py:22-26, internals/modified_helmholtz.py:22-27,
examples/interior_poisson.py:87-88).

Given a smooth closed curve G with N nodes and a side of evaluation, the layer
potential  S_G[sigma] + D_G[tau]  restricted to that side is reproduced by a plain
single layer on a SOURCE CURVE shifted to the other side by delta(t) = alpha h(t)
(h = local node spacing).  The source strengths mu solve the collocation system
        S_{G <- src} mu = (S_G sigma + (D_G -/+ I/2) tau)|_G
whose right-hand side is the one-sided limit computed with spectrally accurate
on-surface quadrature (Kress' log split; ipde_amd.pybie2d_compat.*_Singular_Form).
Evaluating the point sources at any target at distance >= 0 from G on the evaluation
side is then a smooth sum whose trapezoid error is ~exp(-2 pi alpha) — this is what
lets the dense GPU kernel (`Layer_Apply`) serve near and far targets alike.  The
collocation matrix is ill-conditioned (e^{pi alpha}) but the solve is backward stable
and the data are resolved, so the potential keeps ~1e-14 accuracy (measured on the
5-arm star: alpha = 5 reproduces S+D to 4e-15 down to one node spacing from G).
The matrices are built (ipde_amd.dense_forms) and factored (rocSOLVER) on the GPU; the
per-call solve is the library's blocked substitution (csrc/dense.hip).  Host LAPACK
without a GPU (CPU tests).
"""
import os

import numpy as np
import scipy.linalg

from .pybie2d_compat import (Global_Smooth_Boundary, Laplace_Layer_Form,
                             Laplace_Layer_Singular_Form, Modified_Helmholtz_Layer_Form,
                             Modified_Helmholtz_Layer_Singular_Form, Stokes_Layer_Form,
                             Stokes_Layer_Singular_Form, Stokes_Pressure_Fix)


class QFS_Boundary(object):
    """The two source curves of a boundary (reference embedded_boundary.py:534-551
    reads `.interior_source_bdy` / `.exterior_source_bdy`): sources for evaluating
    INTO the interior sit outside the curve and vice versa."""

    def __init__(self, bdy, eps=1e-12, forced_source_upsampling_factor=None, FF=0.0, modes=2,
                 alpha=None):
        self.bdy = bdy
        self.eps = eps
        if alpha is None:
            alpha = max(4.0, -np.log(eps) / (2 * np.pi) + 1.0) + FF
        self.alpha = alpha
        fs = 1 if forced_source_upsampling_factor is None else int(forced_source_upsampling_factor)
        self.upsample = fs
        fine = bdy if fs == 1 else bdy.generate_resampled_boundary(fs * bdy.N)
        shift = alpha * fine.speed * fine.dt
        self.interior_source_bdy = Global_Smooth_Boundary(c=fine.c + shift * fine.normal_c)
        self.exterior_source_bdy = Global_Smooth_Boundary(c=fine.c - shift * fine.normal_c)


class _QFS(object):
    """qfs([sigma, tau]) -> mu on qfs.source;  qfs.u2s(u) -> mu reproducing boundary
    values u.  Subclasses provide the off-surface and on-surface forms."""

    DEVICE_SOLVE = True
    REFINE_STEPS = 0
    REFINE_DD = False      # refinement residuals in double-double (csrc/dense.hip: ipde_dense_residual)

    def __init__(self, bdy, interior, slp, dlp, qfs_boundary=None, eps=1e-12):
        self.bdy = bdy
        self.interior = interior
        q = qfs_boundary if qfs_boundary is not None else QFS_Boundary(bdy, eps=eps)
        self.source = q.interior_source_bdy if interior else q.exterior_source_bdy
        self.slp, self.dlp = slp, dlp
        self._dev = _device() if self.DEVICE_SOLVE else None
        on_dev = self._dev is not None and hasattr(self, '_s2b_dev')
        # (N, Ns) collocation matrix and the one-sided on-surface forms; built on the GPU
        # when the kernel has torch builders (ipde_amd.dense_forms), else on the host
        A = self._s2b_dev(self.source, bdy, self._dev) if on_dev else self._s2b(self.source, bdy)
        self._nrow = A.shape[0]
        jump = -0.5 if interior else 0.5
        S = self._cached_singular(bdy, True, False, on_dev) if slp else None
        D = self._cached_singular(bdy, False, True, on_dev) if dlp else None
        if self._dev is not None and A.shape[0] == A.shape[1]:
            # factor with rocSOLVER, substitute with the library's blocked kernels (the
            # per-solve cost of three 4096^2 host LU back-substitutions was the largest
            # single item of a warm 2048^2 Poisson solve)
            import torch
            as_dev = lambda M: (M if isinstance(M, torch.Tensor) else torch.as_tensor(M, device=self._dev)).contiguous()
            self._A = as_dev(A)
            self._fact = _lu_async(self._A)
            # the on-surface forms are the curve's, shared by the two QFS objects of an interface (no copy of D with
            # the jump folded into its diagonal: 2.95 GB per side at 19 200 rows); the one-sided limit is
            # S sigma + D tau + jump tau, and its first two terms are computed once per interface (call_many)
            self._S = None if S is None else as_dev(S)
            self._D = None if D is None else as_dev(D)
            self._jump = jump
        else:
            self._dev = None
            to_np = lambda M: M.cpu().numpy() if hasattr(M, 'cpu') else M
            A, S, D = to_np(A), (None if S is None else to_np(S)), (None if D is None else to_np(D))
            if D is not None:
                D = D + jump * np.eye(D.shape[0])
            self._S, self._D = S, D
            if A.shape[0] == A.shape[1]:
                self._lu = scipy.linalg.lu_factor(A)
                self._solve_host = lambda u: scipy.linalg.lu_solve(self._lu, u)
            else:
                U, sv, Vt = np.linalg.svd(A, full_matrices=False)
                keep = sv > sv[0] * 1e-14
                self._pinv = (Vt[keep].T / sv[keep]) @ U[:, keep].T
                self._solve_host = lambda u: self._pinv @ u

    def _cached_singular(self, bdy, c, d, on_dev=False):
        """the on-surface forms depend on the curve only: the two QFS objects of an
        interface (grid side / radial side) share them"""
        cache = bdy.__dict__.setdefault('_singular_forms', {})
        key = (type(self).__name__, getattr(self, 'k', None), c, d, on_dev)
        if key not in cache:
            cache[key] = self._singular_dev(bdy, c, d, self._dev) if on_dev else self._singular(bdy, c, d)
        return cache[key]

    def _solve(self, u):
        if self._dev is None:
            return self._solve_host(np.asarray(u, dtype=float))
        import torch
        ud = u if isinstance(u, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(u, dtype=float),
                                                                  device=self._dev)
        x = self._fact.solve(self._A, ud, steps=self.REFINE_STEPS, dd=self.REFINE_DD)
        return x if isinstance(u, torch.Tensor) else x.cpu().numpy()

    def principal_value(self, densities):
        """S[sigma] + D[tau] on the curve itself (device path): what the limits from either side share"""
        import torch
        densities = [_on_device(d, self._dev) for d in densities]
        u = None
        i = 0
        if self.slp:
            u = _gemv(self._S, densities[i])
            i += 1
        if self.dlp:
            u = _gemv(self._D, densities[i], u)
        return u if u is not None else torch.zeros(self._nrow, dtype=torch.float64, device=self._dev)

    def boundary_limit(self, densities, pv=None):
        """one-sided limit on the curve of S[sigma] + D[tau]; pv: principal_value(densities) when the caller has
        it already (the other side's QFS object of the same interface asked for the same product)"""
        densities = list(densities)
        if self._dev is not None:
            if pv is None:
                pv = self.principal_value(densities)
            if not self.dlp:
                return pv
            return pv + self._jump * _on_device(densities[1 if self.slp else 0], self._dev)
        else:
            u = np.zeros(self._nrow)
        i = 0
        if self.slp:
            u = u + self._S @ densities[i]
            i += 1
        if self.dlp:
            u = u + self._D @ densities[i]
        return u

    def __call__(self, densities):
        densities = self._prepare(densities)
        x = self._solve(self.boundary_limit(densities))
        return self._post(x.cpu().numpy() if self._dev is not None else x, densities)

    # hooks of the subclasses around the linear solve (Stokes: density bookkeeping, pressure)
    def _prepare(self, densities):
        return list(densities)

    def _post(self, mu, densities):
        return mu

    def _filter(self, mu):      # (what u2s applies to its result: Stokes_QFS low-passes the density)
        return mu

    def u2s(self, u):
        return self._filter(self._solve(np.asarray(u, dtype=float)))


def _gemv(A, x, y=None):
    """A x (or y += A x) on the device through the library's own kernel (csrc/dense.hip:
    ipde_dense_gemv) — A: contiguous (m, n) fp64 device tensor"""
    import torch
    from .device import get_context, ptr
    ctx = get_context(A.device.index)
    A = A if A.is_contiguous() else A.contiguous()      # (the QFS objects keep theirs contiguous)
    x = x.contiguous()
    acc = y is not None
    if y is None:
        y = torch.empty(A.shape[0], dtype=torch.float64, device=A.device)
    ctx.check(ctx.lib.ipde_dense_gemv(ctx.handle, A.shape[0], A.shape[1], ptr(A), ptr(x), ptr(y), int(acc)))
    return y


def _residual(A, x, b, dd=False):
    """b - A x on the device; dd: products and sum carried in double-double (csrc/dense.hip:
    ipde_dense_residual) — the residual of a refinement step where the plain one's rounding,
    eps |A| |x|, is as large as the residual itself (Stokes QFS: condition ~1e15, densities 1e3..1e4)"""
    if not dd:
        return b - _gemv(A, x)
    import torch
    from .device import get_context, ptr
    ctx = get_context(A.device.index)
    A = A if A.is_contiguous() else A.contiguous()
    x, b = x.contiguous(), b.contiguous()
    r = torch.empty_like(b)
    ctx.check(ctx.lib.ipde_dense_residual(ctx.handle, A.shape[0], A.shape[1], ptr(A), ptr(x), ptr(b), ptr(r)))
    return r


def _on_device(d, dev):
    """density as a flat fp64 device tensor (device tensors pass through: the solvers keep
    their per-boundary vectors in HBM between the stages of a solve)"""
    import torch
    if isinstance(d, torch.Tensor):
        return d.to(device=dev, dtype=torch.float64).reshape(-1)
    return torch.as_tensor(np.ascontiguousarray(d, dtype=float), device=dev).reshape(-1)


def _wants_device(densities):
    import torch
    return any(isinstance(d, torch.Tensor) for d in densities)


def call_many(requests):
    """[q(densities) for q, densities in requests] — the QFS solves of one stage of a solver:
    the grid-side and the annulus-side system of every interface (reference
    internals/scalar.py:87-88, internals/vector.py:133-134).  Systems factored on the GPU are
    substituted (and refined) in lock-step through batched launch sequences, grouped by their
    number of refinement steps; the rest are solved one by one."""
    out = [None] * len(requests)
    groups = {}
    for i, (q, d) in enumerate(requests):
        if q._dev is not None and hasattr(q, '_fact'):
            groups.setdefault((q.REFINE_STEPS, q.REFINE_DD), []).append(i)
        else:
            out[i] = q(d)
    for (steps, dd), idx in groups.items():
        qs = [requests[i][0] for i in idx]
        ds = [q._prepare(requests[i][1]) for q, i in zip(qs, idx)]
        # the two sides of an interface ask for the limits of the same layers of the same densities: S sigma + D tau
        # once (the matrices twice fewer through HBM: 2 x 2.95 GB at 19 200 rows)
        pvs, us = {}, []
        for q, d in zip(qs, ds):
            key = (id(q._S), id(q._D), q.slp, q.dlp) + tuple(id(x) for x in d)
            if key not in pvs:
                pvs[key] = q.principal_value(d)
            us.append(q.boundary_limit(d, pvs[key]))
        xs = _DeviceLU.solve_batch([q._fact for q in qs], [q._A for q in qs], us, steps=steps, dd=dd)
        for i, q, d, x in zip(idx, qs, ds, xs):
            # device densities in -> device density out (no host round trip)
            out[i] = q._post(x if _wants_device(d) else x.cpu().numpy(), d)
    return out


def call_pair(qa, qb, densities):
    """(qa(densities), qb(densities)) in one batched substitution (see call_many)"""
    return tuple(call_many([(qa, densities), (qb, densities)]))


def u2s_many(requests):
    """[q.u2s(u) for q, u in requests], batched like call_many"""
    import torch
    out = [None] * len(requests)
    groups = {}
    for i, (q, u) in enumerate(requests):
        if q._dev is not None and hasattr(q, '_fact'):
            groups.setdefault((q.REFINE_STEPS, q.REFINE_DD), []).append(i)
        else:
            out[i] = q.u2s(u)
    for (steps, dd), idx in groups.items():
        qs = [requests[i][0] for i in idx]
        us = [_on_device(requests[i][1], q._dev) for q, i in zip(qs, idx)]
        xs = _DeviceLU.solve_batch([q._fact for q in qs], [q._A for q in qs], us, steps=steps, dd=dd)
        for q, i, x in zip(qs, idx, xs):
            x = q._filter(x)
            out[i] = x if isinstance(requests[i][1], torch.Tensor) else x.cpu().numpy()
    return out


OWN_FACTORISATION = os.environ.get("IPDE_OWN_LU", "1") != "0"   # False: rocSOLVER getrf (torch.linalg.lu_factor) for every size
OWN_FACTORISATION_MAX_ROWS = 32768      # (beyond 8192 padded rows: the panel across several CUs, csrc/lu_factor.hip)


def _factor(A):
    """(LU, piv) of rocSOLVER — the factorisation of matrices beyond OWN_FACTORISATION_MAX_ROWS
    padded rows (0.27 s at n = 16 384: ~37 000 column-level launches per 4096 columns, host-launch
    bound; run on pool threads, see _lu_async)"""
    import torch
    return torch.linalg.lu_factor(A)


def _tiled(M):
    """(n, n) device matrix -> the tiled storage of csrc/dense.hip / lu_factor.hip: 64x64 tiles
    contiguous, column-major inside a tile, identity padding to a multiple of 128 rows"""
    import torch
    n = int(M.shape[0])
    nb = 2 * ((n + 127) // 128)
    pad = torch.eye(nb * 64, dtype=torch.float64, device=M.device)
    pad[:n, :n] = M
    return pad.view(nb, 64, nb, 64).permute(0, 2, 3, 1).contiguous()


def _own_lu(A):
    """_DeviceLU of A through the library's own blocked factorisation (csrc/lu_factor.hip:
    ipde_dense_lu_factor), enqueued on the context's stream — no host synchronisation, no pool
    thread: ~260 launches, ~20 ms of GPU time at n = 4096."""
    import torch
    from .device import get_context, ptr
    ctx = get_context(A.device.index)
    T = _tiled(A)
    perm = torch.empty(T.shape[0] * 64, dtype=torch.int32, device=A.device)
    ctx.check(ctx.lib.ipde_dense_lu_factor(ctx.handle, T.shape[0] * 64, ptr(T), ptr(perm)))
    return _DeviceLU.from_tiled(T, perm, int(A.shape[0]), ctx)


# (factorisation streams of the device's lowest priority: measured, nothing — configs[4] set-up 1.32 against 1.29 s,
#  round 4 — the dispatcher does not let the foreground's short kernels overtake; off)
BACKGROUND_PRIORITY = os.environ.get("IPDE_LU_BACKGROUND_PRIORITY", "0") != "0"
_own_slots = {}        # device index -> [(private context, torch stream), ...], used round robin
_own_next = [0]
OWN_FACTORISATION_STREAMS = 4


class _OwnAsyncLU(object):
    """The own factorisation on a NON-blocking side stream (a private library context bound to a
    torch stream): most of a factorisation is its single-workgroup panel kernel, so the two QFS
    systems of every interface and the example's integral equation factor side by side, under the
    host's geometry work and the default stream's kernels (on the context's own blocking stream
    every default-stream operation of the set-up waited for them: +0.08 s).  Every use before
    the factorisation has finished makes the consumer's stream wait for its event."""

    def __init__(self, A):
        import torch
        from .device import get_context, private_context, ptr
        dev = A.device.index
        slots = _own_slots.setdefault(dev, [])
        k = _own_next[0] % OWN_FACTORISATION_STREAMS
        _own_next[0] += 1
        while len(slots) <= k:
            pctx = private_context(dev)
            if BACKGROUND_PRIORITY:
                side = pctx.use_background_stream()      # non-blocking, lowest priority: see the class comment
            else:
                side = torch.cuda.Stream(device=A.device)
                pctx.use_torch_stream(side)
            slots.append((pctx, side))
        pctx, side = slots[k]
        T = _tiled(A)
        perm = torch.empty(T.shape[0] * 64, dtype=torch.int32, device=A.device)
        ready = torch.cuda.Event()
        ready.record()                       # T is complete on the caller's stream from here on
        side.wait_event(ready)
        T.record_stream(side)
        perm.record_stream(side)
        pctx.check(pctx.lib.ipde_dense_lu_factor(pctx.handle, T.shape[0] * 64, ptr(T), ptr(perm)))
        self._done = torch.cuda.Event()
        self._done.record(side)
        self._obj = _DeviceLU.from_tiled(T, perm, int(A.shape[0]), get_context(dev))
        self._complete = False

    def _get(self):
        # every use before the factorisation has FINISHED orders the consumer behind it: the
        # current torch stream (which may be a non-blocking one inside `with torch.cuda.stream`)
        # and the legacy default stream, which the library's own blocking streams run behind.
        # (A one-time flag set by the first consumer would leave later consumers on other
        # streams unordered.)  Once the event has completed nothing is enqueued any more.
        if not self._complete:
            if self._done.query():
                self._complete = True
            else:
                import torch
                cur = torch.cuda.current_stream()
                cur.wait_event(self._done)
                dflt = torch.cuda.default_stream()
                if dflt != cur:
                    dflt.wait_event(self._done)
        return self._obj

    def __getattr__(self, name):
        return getattr(self._get(), name)


_lu_pool = None
ASYNC_FACTORISATION = True     # False: factor where the matrix is built (one after the other)


def _lu_async(A):
    """_DeviceLU of A, factored in the background: rocSOLVER's getrf is ~37 000 column-level
    launches at n = 4096 (65 ms, host-launch bound) and a set-up holds several independent
    matrices — the two QFS systems of every interface, the example's boundary integral
    equation — so each factorisation runs on a pool thread with a stream of its own while the
    host goes on assembling the next matrix; the first use joins it."""
    if OWN_FACTORISATION and 128 * ((int(A.shape[0]) + 127) // 128) <= OWN_FACTORISATION_MAX_ROWS:
        return _OwnAsyncLU(A) if ASYNC_FACTORISATION else _own_lu(A)
    if not ASYNC_FACTORISATION:
        return _DeviceLU(*_factor(A))
    return _AsyncLU(A)


class _AsyncLU(object):
    def __init__(self, A):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        global _lu_pool
        if _lu_pool is None:
            _lu_pool = ThreadPoolExecutor(4, thread_name_prefix="ipde-lu")
        ready = torch.cuda.Event()
        ready.record()                 # A is complete on the caller's stream from here on
        dev = A.device

        def work():
            torch.cuda.set_device(dev)
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                s.wait_event(ready)
                obj = _DeviceLU(*_factor(A))
                done = torch.cuda.Event()
                done.record(s)
            return obj, done
        self._obj = None
        self._fut = _lu_pool.submit(work)

    def _get(self):
        obj = self._obj
        if obj is None:
            import torch
            obj, done = self._fut.result()      # (a Future hands its result to any number of callers)
            torch.cuda.current_stream().wait_event(done)
            self._obj = obj
        return obj

    def __getattr__(self, name):       # ctx, n, LU, perm, solve, _subst: the realised object's
        return getattr(self._get(), name)


class _DeviceLU(object):
    """Substitution with rocSOLVER's factors through the library's own blocked kernels
    (csrc/dense.hip): plain substitution is backward stable where the library TRSM is
    not (residual 4e-14 vs 1.3e-9 on a cond-1e12 collocation matrix) and an order of
    magnitude faster for one right-hand side; one refinement step on top."""

    def __init__(self, LU, piv=None, perm=None):
        import torch
        from .device import get_context
        self.ctx = get_context(LU.device.index)
        self.n = n = int(LU.shape[0])
        # an even number of 64-row blocks (two per substitution step); 64x64 tiles stored
        # contiguously, column-major inside a tile, identity padding (layout of ipde_dense_lu_solve)
        self.LU = _tiled(LU)
        if perm is None:
            p = np.arange(self.n)
            for i, q in enumerate(piv.cpu().numpy() - 1):  # LAPACK ipiv -> permutation vector
                if q != i:
                    p[i], p[q] = p[q], p[i]
            perm = torch.as_tensor(p, device=LU.device)
        self.perm = perm.to(torch.int32).contiguous()

    @classmethod
    def from_tiled(cls, T, perm, n, ctx):
        """factors already in the tiled storage (ipde_dense_lu_factor); perm: int32 device tensor
        over the padded rows — rows >= n are the identity padding and pivot on themselves"""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.n = n
        self.LU = T
        self.perm = perm[:n].contiguous()
        return self

    def _subst(self, b):
        import torch
        from .device import ptr
        b = b.contiguous()
        x = torch.empty_like(b)
        self.ctx.check(self.ctx.lib.ipde_dense_lu_solve(self.ctx.handle, self.n, ptr(self.LU),
                                                        ptr(self.perm), ptr(b), ptr(x)))
        return x

    def solve(self, A, b, steps=0, dd=False):
        """substitution (+ `steps` of iterative refinement; the plain substitution already
        has LAPACK's residual, so the scalar QFS solves use none; dd: residuals in double-double)"""
        x = self._subst(b)
        for _ in range(steps):
            x = x + self._subst(_residual(A, x, b, dd))
        return x

    @staticmethod
    def _subst_batch(facts, bs):
        """the substitutions of several systems in lock-step (ipde_dense_lu_solve_batch):
        latency bound, so a batch costs what its largest member does.  Up to 8 systems per
        call; longer lists go in groups."""
        import ctypes
        import torch
        ctx = facts[0].ctx
        bs = [b.contiguous() for b in bs]
        xs = [torch.empty_like(b) for b in bs]
        for a in range(0, len(facts), 8):
            fa, ba, xa = facts[a:a + 8], bs[a:a + 8], xs[a:a + 8]
            k = len(fa)
            arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
            ns = (ctypes.c_int64 * k)(*[f.n for f in fa])
            ctx.check(ctx.lib.ipde_dense_lu_solve_batch(ctx.handle, k, ns, arr([f.LU for f in fa]),
                                                        arr([f.perm for f in fa]), arr(ba), arr(xa)))
        return xs

    @staticmethod
    def solve_batch(facts, As, bs, steps=0, dd=False):
        """solve() for several systems together; facts: _DeviceLU objects"""
        xs = _DeviceLU._subst_batch(facts, bs)
        for _ in range(steps):
            ds = _DeviceLU._subst_batch(facts, [_residual(A, x, b, dd) for A, b, x in zip(As, bs, xs)])
            xs = [x + d for x, d in zip(xs, ds)]
        return xs


class DenseSolver(object):
    """LU of a well-conditioned dense system (the second-kind boundary integral
    equations of the example scripts) — on the GPU through torch with one refinement
    step, host LAPACK without a GPU.  solve(b) -> numpy."""

    def __init__(self, A, refine=1):
        self._dev = _device()
        self.refine = int(refine)      # refinement steps on top of the substitution
        if self._dev is not None:
            import torch
            self._A = A.to(self._dev) if isinstance(A, torch.Tensor) \
                else torch.as_tensor(np.ascontiguousarray(A), device=self._dev)
            self._fact = _lu_async(self._A)
        else:
            self._lu = scipy.linalg.lu_factor(A)

    def solve(self, b):
        if self._dev is None:
            return scipy.linalg.lu_solve(self._lu, np.asarray(b, dtype=float))
        import torch
        if isinstance(b, torch.Tensor):        # device in, device out (the examples' resident correction)
            return self._fact.solve(self._A, b.to(device=self._dev, dtype=torch.float64).reshape(-1),
                                    steps=self.refine)
        bd = torch.as_tensor(np.ascontiguousarray(b, dtype=float), device=self._dev)
        return self._fact.solve(self._A, bd, steps=self.refine).cpu().numpy()


def _device():
    """torch device for the QFS dense algebra, or None on a machine without a GPU (the
    host path keeps the CPU geometry tests running; it is set-up math, not the hot path)."""
    try:
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    except Exception:
        return None


class Laplace_QFS(_QFS):
    def __init__(self, bdy, interior, slp=True, dlp=True, qfs_boundary=None, eps=1e-12):
        super().__init__(bdy, interior, slp, dlp, qfs_boundary, eps)

    def _s2b(self, src, trg):
        return Laplace_Layer_Form(src, trg, ifcharge=True)

    def _singular(self, bdy, c, d):
        return Laplace_Layer_Singular_Form(bdy, ifcharge=c, ifdipole=d)

    def _s2b_dev(self, src, trg, dev):
        from . import dense_forms as df
        return df.laplace_form(src, trg, dev, ifcharge=True)

    def _singular_dev(self, bdy, c, d, dev):
        from . import dense_forms as df
        return df.laplace_singular_form(bdy, dev, ifcharge=c, ifdipole=d)


class Modified_Helmholtz_QFS(_QFS):
    """(reference internals/modified_helmholtz.py:22-27 call shape)"""

    def __init__(self, bdy, interior, slp=True, dlp=True, k=1.0, source_upsample_factor=1.0,
                 closer_source=True, qfs_boundary=None, eps=1e-12):
        self.k = k
        if qfs_boundary is None and source_upsample_factor and source_upsample_factor > 1:
            qfs_boundary = QFS_Boundary(bdy, eps=eps,
                                        forced_source_upsampling_factor=int(np.ceil(source_upsample_factor)))
        super().__init__(bdy, interior, slp, dlp, qfs_boundary, eps)

    def _s2b(self, src, trg):
        return Modified_Helmholtz_Layer_Form(src, trg, k=self.k, ifcharge=True)

    def _singular(self, bdy, c, d):
        return Modified_Helmholtz_Layer_Singular_Form(bdy, k=self.k, ifcharge=c, ifdipole=d)

    def _s2b_dev(self, src, trg, dev):
        from . import dense_forms as df
        return df.modhelm_form(src, trg, dev, self.k, ifcharge=True)

    def _singular_dev(self, bdy, c, d, dev):
        from . import dense_forms as df
        return df.modhelm_singular_form(bdy, dev, self.k, ifcharge=c, ifdipole=d)


class Stokes_QFS(_QFS):
    """Vector QFS (reference ipde/solvers/internals/stokes.py:21-24,
    examples/multi_stokes.py:166-171): densities and results are stacked [x; y] vectors of
    length 2N.  The stokeslet collocation matrix from a closed source curve has a
    one-dimensional null space on both sides — the source normal n_s produces no flow
    (only a constant pressure), and every single-layer field has zero flux through the
    curve — so the rank-one term  n_trg (x) n_src w_src / L  is added: for flux-free data
    (all the solver produces) it leaves the velocity untouched and selects the density
    with  int mu.n = 0, which also pins the pressure constant."""

    MAX_ALPHA = float(os.environ.get("IPDE_STOKES_QFS_MAX_ALPHA", "5.4"))
    # factor with rocSOLVER, substitute with csrc/dense.hip (with the library TRSM the
    # potentials lost four digits at condition 1e15 — 1e-8 vs 3e-13 — which is why this
    # class first ran on host LAPACK; plain substitution has LAPACK's residual)
    DEVICE_SOLVE = True
    REFINE_STEPS = int(os.environ.get("IPDE_STOKES_QFS_REFINE_STEPS", "1"))
    REFINE_DD = os.environ.get("IPDE_STOKES_QFS_REFINE_DD", "0") != "0"

    def __init__(self, bdy, interior, slp=True, dlp=True, qfs_boundary=None, eps=1e-12):
        # Per Fourier mode k the stokeslet block from a curve at distance d is
        # e^{-|k|d}/(4|k|) (I + |k|d N) with N nilpotent: its small singular value carries an
        # extra 1/(2|k|d) compared with the Laplace single layer.  At the scalar solvers'
        # alpha = 6.1 (eps 1e-14) the 2N x 2N matrix is numerically singular (smallest
        # singular values 1e-16, measured at N = 2000); alpha <= 5.4 keeps the condition
        # number where the Laplace one is (~1e13) at a quadrature error e^{-2 pi alpha} = 2e-15.
        if qfs_boundary is None or qfs_boundary.alpha > self.MAX_ALPHA:
            up = None if qfs_boundary is None else qfs_boundary.upsample
            qfs_boundary = QFS_Boundary(bdy, eps=eps, forced_source_upsampling_factor=up,
                                        alpha=min(self.MAX_ALPHA,
                                                  max(4.0, -np.log(eps) / (2 * np.pi) + 1.0)))
        super().__init__(bdy, interior, slp, dlp, qfs_boundary, eps)
        if interior:
            self._pressure_calibration()
        if self._dev is not None and self.NOISE_CUT:
            from .device import prewarm
            prewarm(fft1=((1, int(self.source.N)),))      # the noise cut's transform plan, under the set-up

    def _s2b(self, src, trg):
        return Stokes_Layer_Form(src, trg, ifforce=True) + Stokes_Pressure_Fix(src, trg)

    def _singular(self, bdy, c, d):
        return Stokes_Layer_Singular_Form(bdy, ifforce=c, ifdipole=d)

    def _s2b_dev(self, src, trg, dev):
        from . import dense_forms as df
        return df.stokes_form(src, trg, dev, ifforce=True) + df.stokes_pressure_fix(src, trg, dev)

    def _singular_dev(self, bdy, c, d, dev):
        from . import dense_forms as df
        return df.stokes_singular_form(bdy, dev, ifforce=c, ifdipole=d)

    def _pressure_calibration(self):
        """Inside the curve the density n_src adds a constant pressure and no velocity, so
        the collocation leaves the pressure level open.  It is pinned at one point deep
        inside (where the plain trapezoid rule on the curve itself is spectrally accurate):
        row vectors giving the pressure there of the boundary layers and of the sources."""
        b, s = self.bdy, self.source
        x0, y0 = _deep_interior_point(b)
        self._p_point = (x0, y0)

        def rows(c, force, dipole):
            dx, dy = x0 - c.x, y0 - c.y
            ir2 = 1.0 / (dx * dx + dy * dy)
            out = []
            if force:
                out.append(np.concatenate([dx, dy]) * np.tile(ir2 * c.weights, 2) * (0.5 / np.pi))
            if dipole:
                dn = dx * c.normal_x + dy * c.normal_y
                q = 2.0 * dn * ir2 * ir2
                out.append(np.concatenate([(-c.normal_x * ir2 + q * dx), (-c.normal_y * ir2 + q * dy)])
                           * np.tile(c.weights, 2) / np.pi)
            return out
        self._p_rows = rows(b, self.slp, self.dlp)
        self._p_src = rows(s, True, False)[0]
        self._n_src = np.concatenate([s.normal_x, s.normal_y])
        self._p_null = float(self._p_src @ self._n_src)      # pressure of the null density (-1)

    def _prepare(self, densities):
        densities = list(densities)
        want = int(self.slp) + int(self.dlp)
        # the reference's interior double-layer call passes [tau, tau] (one entry feeds its
        # pressure-fix block, which vanishes for flux-free tau): keep the last `want`
        densities = densities[max(0, len(densities) - want):]
        # ... and its combined-layer call on a hole passes one tau for both layers (:171)
        return densities + [densities[-1]] * (want - len(densities))

    # Noise cut of the source density.  The collocation's singular values fall like e^{-|k| alpha h} / |k|, so
    # past the point where the boundary data have decayed to their noise floor the solve returns AMPLIFIED noise:
    # measured on the 3-body example's outer boundary at configs[4] scale (N = 9560) the density's spectrum falls
    # to 6e-9 of its mean by k ~ 700, then RISES again to a peak of 0.6 ... 5 at k = 3857 = 0.81 Nyquist — max|density|
    # 57 or 90 depending on the boundary size — and that noise costs the solution up to a digit and a half next to
    # the curve (n_b = 2390: 5.2e-10 with it, 2.6e-11 without; 2388: 7.3e-11 / 7.1e-12; 2386 ... 2400 all
    # 5e-12 ... 2.6e-11 without: profiles/r04_stokes_nb_density_lowpass.log).  The cut is placed by the spectrum
    # itself (ipde_density_noise_cut, csrc/spectral.hip; `_noise_cut_host` is the same rule in numpy): band maxima,
    # running minimum, and where the whole tail up to 0.9 Nyquist lies RISE times above a minimum that is itself
    # below FLOOR of the largest band, every mode above the minimum's band goes.  A density whose spectrum is still
    # decaying at the Nyquist frequency (the example at its own size, N = 3200) is left alone: a fixed cut at
    # 0.75 Nyquist cost THAT case a factor four (1.1e-10 -> 4.4e-10), measured.  NOISE_CUT = 0 switches it off.
    NOISE_CUT = os.environ.get("IPDE_STOKES_QFS_NOISE_CUT", "1") != "0"
    RISE, FLOOR = 30.0, 1e-5

    @staticmethod
    def _noise_cut_host(mu, rise=30.0, floor_rel=1e-5, max_keep=1.0):
        """(filtered density, last mode kept) — the rule of ipde_density_noise_cut in numpy"""
        mu = np.asarray(mu, dtype=float)
        n = mu.shape[0] // 2
        H = n // 2
        z = np.fft.fft(mu[:n] + 1j * mu[n:])
        k = np.arange(H + 1)
        a = np.maximum(np.abs(z[k]), np.abs(z[(n - k) % n]))
        w = max(4, (H + 1 + 511) // 512)
        nbands = (H + w) // w
        B = np.array([a[j * w:min((j + 1) * w, H + 1)].max() for j in range(nbands)])
        top = min(nbands - 1, int(0.9 * H) // w)
        S = np.minimum.accumulate(B[:top + 1][::-1])[::-1]
        kc, m, jmin = H, B[0], 0
        for j in range(1, top + 1):
            if m < floor_rel * B.max() and S[j] > rise * m:
                kc = (jmin + 1) * w - 1
                break
            if B[j] < m:
                m, jmin = B[j], j
        kc = min(kc, H if max_keep >= 1.0 else int(max_keep * H))
        kk = np.minimum(np.arange(n), n - np.arange(n))       # |k| as integers
        z[kk > kc] = 0.0
        z = np.fft.ifft(z)
        return np.concatenate([z.real, z.imag]), kc

    def _lowpass(self, mu):
        if not self.NOISE_CUT or int(mu.shape[0]) < 64:      # (a curve of fewer than 32 nodes has no spectrum to speak of)
            return mu
        if type(mu).__module__.startswith('torch'):
            import torch
            from .device import get_context, ptr
            ctx = get_context(mu.device.index)
            mu = mu.contiguous()
            out = torch.empty_like(mu)
            ctx.check(ctx.lib.ipde_density_noise_cut(ctx.handle, int(mu.shape[0]) // 2, ptr(mu), ptr(out),
                                                     float(self.RISE), float(self.FLOOR), 1.0, None))
            return out
        return self._noise_cut_host(mu, self.RISE, self.FLOOR)[0]

    def _post(self, mu, densities):
        return self._lowpass(self._calibrate(mu, densities))

    def _filter(self, mu):
        return self._lowpass(mu)

    def _calibrate(self, mu, densities):
        if self.interior:
            if type(mu).__module__.startswith('torch'):
                # device densities (the solvers' device-resident flow): the same calibration with
                # the row vectors resident in HBM, no host value in between
                import torch
                c = getattr(self, '_p_dev', None)
                if c is None:
                    up = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=float), device=mu.device)
                    c = self._p_dev = ([up(r) for r in self._p_rows], up(self._p_src), up(self._n_src))
                rows, p_src, n_src = c
                # (products and sums, not torch.dot: that would load rocBLAS for three dot products)
                p_true = sum((r * _on_device(d, mu.device)).sum() for r, d in zip(rows, densities))
                return mu + (p_true - (p_src * mu).sum()) / self._p_null * n_src
            p_true = sum(r @ np.asarray(d, dtype=float) for r, d in zip(self._p_rows, densities))
            mu = mu + (p_true - self._p_src @ mu) / self._p_null * self._n_src
        return mu


def _deep_interior_point(b, ngrid=48):
    """A point inside the closed curve far from it (largest distance to the nodes among a
    coarse lattice of candidates)."""
    from .near import points_inside_curve
    xs = np.linspace(b.x.min(), b.x.max(), ngrid + 2)[1:-1]
    ys = np.linspace(b.y.min(), b.y.max(), ngrid + 2)[1:-1]
    X, Y = np.meshgrid(xs, ys, indexing='ij')
    X, Y = X.ravel(), Y.ravel()
    # (the polygon of the nodes themselves: the point wanted is far from the curve, where the nodes'
    # polygon and the 8x upsampled one agree; distances to every fourth node rank the candidates)
    inside = points_inside_curve(b, X, Y, upsample=1)
    X, Y = X[inside], Y[inside]
    st = max(1, b.N // 2400)
    d2 = ((X[:, None] - b.x[None, ::st]) ** 2 + (Y[:, None] - b.y[None, ::st]) ** 2).min(axis=1)
    i = int(np.argmax(d2))
    return float(X[i]), float(Y[i])


class QFS_Evaluator(object):
    """The call shape of qfs.two_d_qfs.QFS_Evaluator used by the examples
    (examples/interior_poisson.py:87): boundary-to-check functions give the one-sided
    boundary values (`on_surface=True`), `s2c_func(src, trg)` the source-to-boundary
    matrix.  qfs([tau, ...]) -> source density."""

    def __init__(self, qfs_boundary, interior, b2c_funcs, s2c_func, on_surface=True, form_b2c=False,
                 vector=False):
        if not on_surface:
            raise NotImplementedError("only the on-surface (boundary-matching) variant is built")
        self.bdy = qfs_boundary.bdy
        self.interior = interior
        self.source = qfs_boundary.interior_source_bdy if interior else qfs_boundary.exterior_source_bdy
        self.b2c_mats = [f(self.bdy, self.bdy) for f in b2c_funcs]
        A = s2c_func(self.source, self.bdy)
        self._dev = _device()
        if self._dev is not None:
            import torch
            self._A = torch.as_tensor(A, device=self._dev)
            self._fact = _lu_async(self._A)
            self.b2c_mats = [torch.as_tensor(B, device=self._dev).contiguous() for B in self.b2c_mats]
        else:
            self._lu = scipy.linalg.lu_factor(A)

    def __call__(self, densities):
        if self._dev is not None:
            import torch
            on_dev = _wants_device(densities)
            u = None
            for B, d in zip(self.b2c_mats, densities):
                u = _gemv(B if B.is_contiguous() else B.contiguous(), _on_device(d, self._dev), u)
            x = self._fact.solve(self._A, u)
            return x if on_dev else x.cpu().numpy()
        u = sum(B @ np.asarray(d) for B, d in zip(self.b2c_mats, densities))
        return scipy.linalg.lu_solve(self._lu, u)
