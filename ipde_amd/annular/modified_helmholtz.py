"""AnnularModifiedHelmholtzSolver on the MI355X — mirrors
ipde/annular/modified_helmholtz.py:90-203 (same constructor, `solve`, attributes).

Set-up (Chebyshev matrices, per-mode preconditioner blocks) is host numpy like the
reference's `_construct` (:123-154), vectorised over the Fourier modes.  The solve —
forward FFT of the right-hand side, right-preconditioned GMRES with the spectral
operator apply and the block preconditioner, inverse FFT — runs on the device
(ipde_annular_scalar_* in the C ABI).
"""
import ctypes

import numpy as np

from .. import _lib
from ..device import get_context, location_of, as_f64, ptr, empty_like_loc


def _host(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def scalar_inverse_blocks(AAG, k, ibc, obc):
    """inv(K_i) for every Fourier mode i (reference `_construct`, :123-154):
    K_i = [k^2 R02 - LL_i ; ibc ; obc],  LL_i = radial part - ks_i^2 * tangential part.
    Pure numpy (set-up), vectorised over the modes.  Returns (ns, M, M) real."""
    CO, M = AAG.CO, AAG.M
    radial = AAG.approx_inv_psi2[:, None] * (CO.D12 @ (AAG.approx_psi1[:, None] * CO.D01))
    tangential = CO.R12 @ (AAG.approx_inv_psi1[:, None] * CO.R01)
    K = np.empty((AAG.ns, M, M))
    K[:, :M - 2, :] = (k ** 2 * CO.R02 - radial)[None] + \
        (AAG.ks ** 2)[:, None, None] * tangential[None]
    K[:, M - 2, :] = ibc
    K[:, M - 1, :] = obc
    return np.linalg.inv(K)


class AnnularModifiedHelmholtzSolver(object):
    """Spectrally accurate solver of (k^2 - L) u = f on the annulus AAG with Robin
    data  ia*u(ri) + ib*u_r(ri) = ig,  oa*u(ro) + ob*u_r(ro) = og."""

    def __init__(self, AAG, k, ia=1.0, ib=0.0, oa=1.0, ob=0.0, ctx=None):
        if AAG.ns != AAG.n:
            raise Exception("the scalar annular solver needs the Nyquist-keeping geometry "
                            "(annular_full.ApproximateAnnularGeometry, ns == n)")
        self.ctx = ctx or get_context()
        self.AAG = AAG
        self.ia, self.ib, self.oa, self.ob = ia, ib, oa, ob
        self.k = k
        self.M, self.ns, self.n = AAG.M, AAG.ns, AAG.n
        self.NB = self.M * self.ns
        self.small_shape = (self.M, self.ns)
        self.shape = (self.M, self.n)
        self.handle = None
        self._rag_id = None
        self.iterations_last_call = None
        self.residual_last_call = None
        self._construct()

    # -- set-up ---------------------------------------------------------------
    def _bc_rows(self):
        CO = self.AAG.CO
        return (self.ia * CO.ibc_dirichlet[0] + self.ib * CO.ibc_neumann[0],
                self.oa * CO.obc_dirichlet[0] + self.ob * CO.obc_neumann[0])

    def _construct(self):
        ibc, obc = self._bc_rows()
        self.Stacked_KINVS = scalar_inverse_blocks(self.AAG, self.k, ibc, obc)
        self._bc_at_construct = (self.ia, self.ib, self.oa, self.ob)
        self._create_handle()

    def _create_handle(self):
        if self.handle:
            self.ctx.lib.ipde_annular_scalar_destroy(self.handle)
            self.handle = None
        CO = self.AAG.CO
        ibc, obc = self._bc_rows()
        mats = [_host(m) for m in (CO.R01, CO.R12, CO.R02, CO.D01, CO.D12, ibc, obc,
                                   self.Stacked_KINVS)]
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.ipde_annular_scalar_create(
            self.ctx.handle, self.M, self.n, float(self.k), *[ptr(m) for m in mats],
            ctypes.byref(h)))
        self.handle = h
        self.ctx.adopt(self)
        self._rag_id = None

    def _release(self):
        """free the library handle (also called by the owning context before it goes)"""
        h, self.handle = self.handle, None
        if h and self.ctx.handle:
            self.ctx.lib.ipde_annular_scalar_destroy(h)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _set_geometry(self, RAG):
        if self._rag_id == id(RAG):
            return
        f = [_host(a) for a in (RAG.psi1, RAG.inv_psi1, RAG.inv_psi2)]
        self.ctx.check(self.ctx.lib.ipde_annular_scalar_set_geometry(
            self.handle, _lib.IPDE_HOST, *[ptr(a) for a in f]))
        self._rag_id = id(RAG)
        self.RAG = RAG

    # -- operator-level entry points (parity tests; GMRES calls them on the device) --
    def _vec(self, fn, v):
        loc = location_of(v)
        if loc == _lib.IPDE_HOST:
            v = np.ascontiguousarray(v, dtype=np.complex128).ravel()
        out = empty_like_loc((self.NB,), loc, self.ctx, "c16")
        self.ctx.check(fn(self.handle, loc, ptr(v), ptr(out)))
        return out

    def _apply(self, uh):
        """Operator application in Fourier space (reference :172-186)."""
        return self._vec(self.ctx.lib.ipde_annular_scalar_apply, uh)

    def _optim_preconditioner(self, fh):
        """Per-mode block preconditioner (reference :157-159, :68-88)."""
        return self._vec(self.ctx.lib.ipde_annular_scalar_precondition, fh)

    _preconditioner = _optim_preconditioner

    # -- solve ------------------------------------------------------------------
    def solve(self, RAG, f, ig, og, ia=None, ib=None, oa=None, ob=None, verbose=False,
              tol=1e-12, maxiter=200, restart=50, _negate_f=False, **kwargs):
        """Returns u on the (M, n) radial grid (reference :187-203)."""
        new_bc = tuple(new if new is not None else old for new, old in
                       zip((ia, ib, oa, ob), (self.ia, self.ib, self.oa, self.ob)))
        if new_bc != (self.ia, self.ib, self.oa, self.ob):
            # boundary rows of the operator change; the preconditioner keeps the rows it
            # was built with ("may not work so well", reference docstring :101-102)
            self.ia, self.ib, self.oa, self.ob = new_bc
            self._create_handle()
        self._set_geometry(RAG)
        loc = location_of(f)
        f = as_f64(f, loc)
        if loc == _lib.IPDE_HOST:
            ig = _host(np.broadcast_to(ig, (self.n,)))
            og = _host(np.broadcast_to(og, (self.n,)))
        out = empty_like_loc(self.shape, loc, self.ctx)
        iters = ctypes.c_int()
        resid = ctypes.c_double()
        st = self.ctx.lib.ipde_annular_scalar_solve(
            self.handle, loc, ptr(f), ptr(ig), ptr(og), int(bool(_negate_f)), float(tol),
            int(maxiter), int(restart), ptr(out), ctypes.byref(iters), ctypes.byref(resid))
        self.ctx.check(st, allow=(_lib.IPDE_ERR_NOCONV,))
        self.iterations_last_call = iters.value
        self.residual_last_call = resid.value
        if verbose:
            print('GMRES took:', iters.value, 'iterations; relative residual %.2e%s' %
                  (resid.value, '' if st == 0 else ' (maxiter reached)'))
        return out
