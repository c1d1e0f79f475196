"""Periodic spectral grid operators on the device.

`GridPlan` wraps an ipde_fft_plan (rocFFT D2Z/Z2D plans + fused symbol kernels)
for an (nx, ny) C-ordered grid with spacings (hx, hy).  Arrays may be numpy
(staged) or torch CUDA tensors (zero-copy); the result matches the input kind.
"""
import ctypes

import numpy as np

from . import _lib
from .device import get_context, location_of, as_f64, ptr, empty_like_loc, prewarm_wait


class GridPlan:
    def __init__(self, nx, ny, hx, hy, ctx=None):
        self.ctx = ctx or get_context()
        self.nx, self.ny, self.hx, self.hy = int(nx), int(ny), float(hx), float(hy)
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.ipde_fft_plan2d_create(self.ctx.handle, self.nx, self.ny,
                                                           self.hx, self.hy, ctypes.byref(h)))
        self.handle = h
        self.shape = (self.nx, self.ny)

    def close(self):
        if self.handle:
            self.ctx.lib.ipde_fft_plan2d_destroy(self.handle)
            self.handle = None

    # -- helpers --------------------------------------------------------------
    def _real_in(self, f):
        loc = location_of(f)
        f = as_f64(f, loc)
        if tuple(f.shape) != self.shape:
            raise ValueError("grid shape %s does not match the plan %s" % (tuple(f.shape), self.shape))
        return loc, f

    # -- spectrum kept on the device for interpolation to points -----------------
    def keep_spectrum(self, on=True):
        """Ask the scalar grid solves on this plan (called without want_uhat) to keep
        fft2(f) * symbol on the device for `interp_gradient`.  Returns False if this grid
        size has no such path (then use want_uhat and ipde_amd.interp).  keep_spectrum(False)
        switches it off and still reports whether the path exists (`interp_fields` needs it)."""
        ok = ctypes.c_int()
        self.ctx.check(self.ctx.lib.ipde_fft_plan2d_keep_spectrum(self.handle, int(bool(on)),
                                                                  ctypes.byref(ok)))
        return bool(ok.value)

    def prepare_interp(self):
        """Build the interpolation state now (fine-grid plan, window factors, work grids) instead
        of inside the first interp_gradient / interp_fields (the solvers run it on the warm-up
        thread: device.prewarm_submit)."""
        self.ctx.check(self.ctx.lib.ipde_grid_interp_prepare(self.handle))

    def interp_gradient(self, x, y):
        """(3, P): the last kept grid solution, its x and its y derivative at the points
        (x, y) given in box units [0, 2 pi) (reference multi_boundary/scalar.py:80-88)"""
        loc = location_of(x, y)
        x, y = as_f64(x, loc), as_f64(y, loc)
        n = int(x.shape[0])
        out = empty_like_loc((3, n), loc, self.ctx)
        self.ctx.check(self.ctx.lib.ipde_grid_interp(self.handle, loc, n, ptr(x), ptr(y), ptr(out)))
        return out

    def interp_fields(self, fields, outputs, x, y):
        """Linear combinations of real grid fields and their first derivatives at the points
        (x, y) (box units [0, 2 pi)).  fields: list of up to three (nx, ny) arrays; outputs: list
        of lists of (coef, field index, der) with der 0: value, 1: d/dx, 2: d/dy.  Returns
        (len(outputs), P).  E.g. the Stokes stress xx: [(2.0, 0, 1), (-1.0, 2, 0)]."""
        loc = location_of(x, y, *fields)
        x, y = as_f64(x, loc), as_f64(y, loc)
        fields = [self._real_in(f)[1] for f in fields]
        n = int(x.shape[0])
        ptrs = (ctypes.c_void_p * len(fields))(*[ptr(f) for f in fields])
        start, src, der, coef = [0], [], [], []
        for terms in outputs:
            for c, k, d in terms:
                coef.append(float(c))
                src.append(int(k))
                der.append(int(d))
            start.append(len(src))
        arr = lambda v, t: (t * len(v))(*v)
        out = empty_like_loc((len(outputs), n), loc, self.ctx)
        self.ctx.check(self.ctx.lib.ipde_grid_interp_fields(
            self.handle, loc, len(fields), ptrs, len(outputs), arr(start, ctypes.c_int),
            arr(src, ctypes.c_int), arr(der, ctypes.c_int), arr(coef, ctypes.c_double), n, ptr(x), ptr(y),
            ptr(out)))
        return out

    # -- grid solves ----------------------------------------------------------
    def poisson_solve(self, f, want_uhat=False):
        """u = ifft2(fft2(f) * ilap).real  (multi_boundary/poisson.py:30-38; f demeaned
        by the caller).  Returns (uhat, u) like the reference when want_uhat."""
        loc, f = self._real_in(f)
        u = empty_like_loc(self.shape, loc, self.ctx)
        uh = empty_like_loc(self.shape, loc, self.ctx, "c16") if want_uhat else None
        self.ctx.check(self.ctx.lib.ipde_poisson_grid_solve(self.handle, loc, ptr(f), ptr(u), ptr(uh)))
        return (uh, u) if want_uhat else u

    def modhelm_solve(self, f, k, want_uhat=False):
        """(multi_boundary/modified_helmholtz.py:40-46)"""
        loc, f = self._real_in(f)
        u = empty_like_loc(self.shape, loc, self.ctx)
        uh = empty_like_loc(self.shape, loc, self.ctx, "c16") if want_uhat else None
        self.ctx.check(self.ctx.lib.ipde_modhelm_grid_solve(self.handle, loc, float(k), ptr(f),
                                                            ptr(u), ptr(uh)))
        return (uh, u) if want_uhat else u

    def stokes_solve(self, fu, fv):
        """(uc, vc, pc) of multi_boundary/stokes.py:34-45 (fu, fv demeaned by the caller)"""
        loc = location_of(fu, fv)
        fu, fv = as_f64(fu, loc), as_f64(fv, loc)
        u, v, p = (empty_like_loc(self.shape, loc, self.ctx) for _ in range(3))
        self.ctx.check(self.ctx.lib.ipde_stokes_grid_solve(self.handle, loc, ptr(fu), ptr(fv),
                                                           ptr(u), ptr(v), ptr(p)))
        return u, v, p

    # -- derivatives ----------------------------------------------------------
    def dx(self, f):
        return self._deriv(f, 0)

    def dy(self, f):
        return self._deriv(f, 1)

    def _deriv(self, f, axis):
        loc, f = self._real_in(f)
        out = empty_like_loc(self.shape, loc, self.ctx)
        self.ctx.check(self.ctx.lib.ipde_fourier_deriv(self.handle, loc, ptr(f), axis, ptr(out)))
        return out

    def fourier_multiply(self, f, sym):
        """ifft2(fft2(f) * sym).real for an arbitrary complex symbol array."""
        loc, f = self._real_in(f)
        if loc == _lib.IPDE_HOST:
            sym = np.ascontiguousarray(np.broadcast_to(np.asarray(sym, dtype=np.complex128), self.shape))
        else:
            import torch
            if not isinstance(sym, torch.Tensor):
                sym = torch.as_tensor(np.ascontiguousarray(
                    np.broadcast_to(np.asarray(sym, dtype=np.complex128), self.shape)), device=f.device)
            sym = sym.to(torch.complex128).expand(self.shape).contiguous()
        out = empty_like_loc(self.shape, loc, self.ctx)
        self.ctx.check(self.ctx.lib.ipde_fourier_multiply(self.handle, loc, ptr(f), ptr(sym), ptr(out)))
        return out

    # -- plain transforms (ipde.utilities.fft2 / ifft2) ------------------------
    def fft2(self, a):
        loc = location_of(a)
        if (loc == _lib.IPDE_HOST and np.isrealobj(a)) or (loc == _lib.IPDE_DEVICE and not a.is_complex()):
            a = as_f64(a, loc)
            out = empty_like_loc(self.shape, loc, self.ctx, "c16")
            self.ctx.check(self.ctx.lib.ipde_fft2_r2c_full(self.handle, loc, ptr(a), ptr(out)))
            return out
        return self._c2c(a, -1)

    def ifft2(self, a):
        return self._c2c(a, +1)

    def _c2c(self, a, direction):
        loc = location_of(a)
        if loc == _lib.IPDE_HOST:
            a = np.ascontiguousarray(a, dtype=np.complex128)
        else:
            import torch
            a = a.to(torch.complex128).contiguous()
        out = empty_like_loc(self.shape, loc, self.ctx, "c16")
        self.ctx.check(self.ctx.lib.ipde_fft2_c2c(self.handle, loc, direction, ptr(a), ptr(out)))
        return out


_plan_lock = __import__('threading').RLock()


def get_plan(nx, ny, hx, hy, ctx=None):
    """Cached GridPlan (rocFFT plan creation is expensive; shapes repeat)."""
    ctx = ctx or get_context()
    key = (int(nx), int(ny), float(hx), float(hy))
    # join the warm-up thread BEFORE taking the lock: its plan job takes the same lock, and
    # the plan creation below would otherwise wait for it while holding the lock
    prewarm_wait()
    with _plan_lock:
        p = ctx._plans.get(key)
        if p is None or not p.handle:
            p = GridPlan(nx, ny, hx, hy, ctx)
            ctx._plans[key] = p
    return p


def fd4(f, h, axis, periodic_fix=False, ctx=None):
    """4th-order centred difference along axis 0 (x) or 1 (y)."""
    ctx = ctx or get_context()
    loc = location_of(f)
    f = as_f64(f, loc)
    nx, ny = int(f.shape[0]), int(f.shape[1])
    out = empty_like_loc((nx, ny), loc, ctx)
    ctx.check(ctx.lib.ipde_fd4(ctx.handle, loc, nx, ny, float(h), int(axis), int(bool(periodic_fix)),
                               ptr(f), ptr(out)))
    return out


def fft1(a, direction, ctx=None):
    """Batched 1-D complex FFT along the last axis of a 2-D (batch, n) array."""
    ctx = ctx or get_context()
    loc = location_of(a)
    if loc == _lib.IPDE_HOST:
        a = np.ascontiguousarray(a, dtype=np.complex128)
    else:
        import torch
        a = a.to(torch.complex128).contiguous()
    shape = tuple(a.shape)
    batch = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    n = int(shape[-1])
    out = empty_like_loc(shape, loc, ctx, "c16")
    ctx.check(ctx.lib.ipde_fft1_c2c(ctx.handle, loc, batch, n, int(direction), ptr(a), ptr(out)))
    return out
