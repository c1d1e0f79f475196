"""AnnularStokesSolver on the MI355X — mirrors ipde/annular/stokes.py:73-541.

Solves  -mu L u + grad p = f,  div u = 0  in (r, t) coordinates on the annulus with
Dirichlet data on both rims.  Unknown ordering [ur (M,ns); ut (M,ns); p (M-1,ns)]
with ns = n-1 Fourier modes (Nyquist dropped), as in the reference.
"""
import ctypes

import numpy as np

from .. import _lib
from ..device import get_context, location_of, as_f64, ptr, empty_like_loc


def _host(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def stokes_inverse_blocks(AAG, mu):
    """Per-mode (3M-1)^2 blocks of the circular-annulus Stokes operator and their
    inverses (reference `_construct`, :107-153), vectorised over the modes.
    Pure numpy (set-up).  Returns (ns, 3M-1, 3M-1) complex."""
    CO, M, ns = AAG.CO, AAG.M, AAG.ns
    ap0, ap1 = AAG.approx_psi0, AAG.approx_psi1
    ai1, ai2 = AAG.approx_inv_psi1, AAG.approx_inv_psi2
    iks = 1j * AAG.ks
    radial = ai2[:, None] * (CO.D12 @ (ap1[:, None] * CO.D01))
    tangential = CO.R12 @ (ai1[:, None] * CO.R01)
    A = (-radial + (ai2 ** 2)[:, None] * CO.R02)[None] + \
        (AAG.ks ** 2)[:, None, None] * tangential[None]                 # -LL + ipsi2^2 R02
    Bm = iks[:, None, None] * ((2 * ai2 ** 2)[:, None] * CO.R02)[None]   # 2 ipsi2^2 R02 ik
    K = np.zeros((ns, 3 * M - 1, 3 * M - 1), dtype=complex)
    K[:, 0:M - 2, 0:M] = mu * A
    K[:, 0:M - 2, M:2 * M] = mu * Bm
    K[:, 0:M - 2, 2 * M:] = CO.D12
    K[:, M - 2, 0:M] = CO.ibc_dirichlet[0]
    K[:, M - 1, 0:M] = CO.obc_dirichlet[0]
    K[:, M:2 * M - 2, 0:M] = -mu * Bm
    K[:, M:2 * M - 2, M:2 * M] = mu * A
    K[:, M:2 * M - 2, 2 * M:] = iks[:, None, None] * (ai2[:, None] * CO.R12)[None]
    K[:, 2 * M - 2, M:2 * M] = CO.ibc_dirichlet[0]
    K[:, 2 * M - 1, M:2 * M] = CO.obc_dirichlet[0]
    K[:, 2 * M:, 0:M] = ai1[:, None] * (CO.D01 * ap0[None, :])
    K[:, 2 * M:, M:2 * M] = iks[:, None, None] * (ai1[:, None] * CO.R01)[None]
    K[0, 2 * M:, 2 * M:] += CO.VI1[0]   # pressure nullspace fix at mode 0 (:149-150)
    return _batched_inv(K)


def _batched_inv(K):
    """np.linalg.inv of a stack of small blocks, the stack cut over a few host threads (LAPACK
    releases the GIL; one thread took 0.2 s for the 9599 blocks of a 9600-node boundary)"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(16, os.cpu_count() or 1))
    if nthreads == 1 or K.shape[0] < 256:
        return np.linalg.inv(K)
    out = np.empty_like(K)
    bounds = np.linspace(0, K.shape[0], nthreads + 1).astype(int)

    def work(i):
        out[bounds[i]:bounds[i + 1]] = np.linalg.inv(K[bounds[i]:bounds[i + 1]])
    with ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(work, range(nthreads)))
    return out


class AnnularStokesSolver(object):
    def __init__(self, AAG, mu, ctx=None):
        if AAG.ns != AAG.n - 1:
            raise Exception("the Stokes annular solver needs the Nyquist-dropping geometry "
                            "(annular.ApproximateAnnularGeometry, ns == n-1)")
        self.ctx = ctx or get_context()
        self.AAG = AAG
        self.mu = mu
        self.M, self.ns, self.n = AAG.M, AAG.ns, AAG.n
        self.NU = self.M * self.ns
        self.NP = (self.M - 1) * self.ns
        self.NB = 2 * self.NU + self.NP
        self.u_small_shape = (self.M, self.ns)
        self.u_shape = (self.M, self.n)
        self.p_small_shape = (self.M - 1, self.ns)
        self.p_shape = (self.M - 1, self.n)
        self.handle = None
        self._rag_id = None
        self.iterations_last_call = None
        self.residual_last_call = None
        self._construct()

    def _construct(self):
        CO, mu = self.AAG.CO, self.mu
        self.Stacked_KINVS = stokes_inverse_blocks(self.AAG, mu)
        mats = [_host(m) for m in (CO.R01, CO.R12, CO.R02, CO.D01, CO.D12, CO.ibc_dirichlet[0],
                                   CO.obc_dirichlet[0], CO.VI1[0])]
        kinv = np.ascontiguousarray(self.Stacked_KINVS, dtype=np.complex128)
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.ipde_annular_stokes_create(
            self.ctx.handle, self.M, self.n, float(mu), *[ptr(m) for m in mats], ptr(kinv),
            ctypes.byref(h)))
        self.handle = h
        self.ctx.adopt(self)

    def _release(self):
        """free the library handle (also called by the owning context before it goes)"""
        h, self.handle = self.handle, None
        if h and self.ctx.handle:
            self.ctx.lib.ipde_annular_stokes_destroy(h)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _set_geometry(self, RAG):
        if self._rag_id == id(RAG):
            return
        f = [_host(a) for a in (RAG.psi0, RAG.psi1, RAG.inv_psi1, RAG.inv_psi2, RAG.DR_psi2,
                                RAG.ipsi_DR_ipsi_DT_psi2, RAG.ipsi_DT_ipsi_DR_psi2)]
        self.ctx.check(self.ctx.lib.ipde_annular_stokes_set_geometry(
            self.handle, _lib.IPDE_HOST, *[ptr(a) for a in f]))
        self._rag_id = id(RAG)
        self.RAG = RAG

    def _vec(self, fn, v):
        loc = location_of(v)
        if loc == _lib.IPDE_HOST:
            v = np.ascontiguousarray(v, dtype=np.complex128).ravel()
        out = empty_like_loc((self.NB,), loc, self.ctx, "c16")
        self.ctx.check(fn(self.handle, loc, ptr(v), ptr(out)))
        return out

    def _apply_optim_real(self, uuh):
        """Operator application (reference :321-385); needs a geometry (call solve or
        _set_geometry first)."""
        return self._vec(self.ctx.lib.ipde_annular_stokes_apply, uuh)

    _apply = _apply_optim_real

    def _preconditioner(self, ffh):
        """(reference :200-210)"""
        return self._vec(self.ctx.lib.ipde_annular_stokes_precondition, ffh)

    def solve(self, RAG, fr, ft, irg, itg, org, otg, verbose=False, tol=1e-12, maxiter=300,
              restart=100, **kwargs):
        """Returns (ur, ut, p) on the (M, n) radial grid (reference :519-541)."""
        self._set_geometry(RAG)
        loc = location_of(fr, ft)
        fr, ft = as_f64(fr, loc), as_f64(ft, loc)
        if loc == _lib.IPDE_HOST:
            irg, itg, org, otg = (_host(np.broadcast_to(a, (self.n,))) for a in (irg, itg, org, otg))
        ur, ut, p = (empty_like_loc(self.u_shape, loc, self.ctx) for _ in range(3))
        P10 = _host(self.AAG.CO.P10)
        iters = ctypes.c_int()
        resid = ctypes.c_double()
        st = self.ctx.lib.ipde_annular_stokes_solve(
            self.handle, loc, ptr(fr), ptr(ft), ptr(irg), ptr(itg), ptr(org), ptr(otg), ptr(P10),
            float(tol), int(maxiter), int(restart), ptr(ur), ptr(ut), ptr(p), ctypes.byref(iters),
            ctypes.byref(resid))
        self.ctx.check(st, allow=(_lib.IPDE_ERR_NOCONV,))
        self.iterations_last_call = iters.value
        self.residual_last_call = resid.value
        if verbose:
            print('GMRES took:', iters.value, 'iterations; relative residual %.2e%s' %
                  (resid.value, '' if st == 0 else ' (maxiter reached)'))
        return ur, ut, p
