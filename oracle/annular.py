"""TEST INFRASTRUCTURE ONLY — numpy restatement of ipde's annular (Chebyshev x
Fourier) operators and solvers.  Pinned by tests/golden/annular_scalar.npz and
annular_stokes.npz (outputs of the reference's own classes, see
tests/golden/make_golden.py): Chebyshev matrices, metric fields, per-mode inverse
blocks, operator applies and preconditioner applies are exact goldens; solves are
compared at solver tolerance.

Follows:
  ChebyshevOperators            ipde/annular/annular.py:7-50
  ApproximateAnnularGeometry    ipde/annular/annular.py:52-85 (ns = n-1) and
                                ipde/annular/annular_full.py:60-85 (ns = n)
  RealAnnularGeometry           ipde/annular/annular.py:87-108 (the overwritten
                                "these are what work" metric terms, :107-108)
  scalar solver                 ipde/annular/modified_helmholtz.py:25-38,68-88,123-203
  Poisson                       ipde/annular/poisson.py:3-21
  Stokes                        ipde/annular/stokes.py:107-153,200-210,321-385,519-541
  Nyquist handling              ipde/utilities.py:78-101
"""
import numpy as np
from numpy.polynomial import chebyshev as C


def chebyshev_nodes(lb, ub, order):
    xc, _ = C.chebgauss(order)
    xc = xc[::-1]
    rat = (ub - lb) / 2.0
    return xc, (xc + 1.0) * rat + lb, rat


class ChebOps:
    def __init__(self, M, rat):
        self.M = M
        xc0, _ = C.chebgauss(M)
        xc1, _ = C.chebgauss(M - 1)
        xc2, _ = C.chebgauss(M - 2)
        V0, V1, V2 = C.chebvander(xc0, M - 1), C.chebvander(xc1, M - 2), C.chebvander(xc2, M - 3)
        VI0, VI1 = np.linalg.inv(V0), np.linalg.inv(V1)
        self.VI1 = VI1
        DC01 = C.chebder(np.eye(M)) / rat
        DC12 = C.chebder(np.eye(M - 1)) / rat
        DC00 = np.vstack([DC01, np.zeros(M)])
        self.D00 = V0 @ DC00 @ VI0
        self.D01 = V1 @ DC01 @ VI0
        self.D12 = V2 @ DC12 @ VI1
        self.ibc_dirichlet = C.chebvander(1, M - 1) @ VI0
        self.obc_dirichlet = C.chebvander(-1, M - 1) @ VI0
        self.ibc_neumann = self.ibc_dirichlet @ self.D00
        self.obc_neumann = self.obc_dirichlet @ self.D00
        t = np.zeros((M - 1, M))
        np.fill_diagonal(t, 1.0)
        self.R01 = V1 @ t @ VI0
        t = np.zeros((M - 2, M - 1))
        np.fill_diagonal(t, 1.0)
        self.R12 = V2 @ t @ VI1
        self.R02 = self.R12 @ self.R01
        t = np.zeros((M, M - 1))
        np.fill_diagonal(t, 1.0)
        self.P10 = V0 @ t @ VI1


class AAG:
    """full=True -> annular_full (ns = n), else annular (ns = n-1, Nyquist dropped)."""

    def __init__(self, n, M, width, approx_r, full=True):
        self.n, self.M, self.width, self.radius = n, M, width, approx_r
        self.n2 = n // 2
        self.k = np.fft.fftfreq(n, 1.0 / n)
        if full:
            self.ns, self.ks = n, self.k
        else:
            self.ns = n - 1
            self.ks = np.concatenate([self.k[:self.n2], self.k[self.n2 + 1:]])
        self.iks = 1j * self.ks
        _, self.rv0, rat0 = chebyshev_nodes(-width, 0.0, M)
        _, self.rv1, _ = chebyshev_nodes(-width, 0.0, M - 1)
        _, self.rv2, _ = chebyshev_nodes(-width, 0.0, M - 2)
        self.ratio = -rat0
        self.approx_psi0 = approx_r + self.rv0
        self.approx_psi1 = approx_r + self.rv1
        self.approx_psi2 = approx_r + self.rv2
        self.CO = ChebOps(M, self.ratio)


class RAG:
    def __init__(self, speed, curvature, aag):
        n = curvature.shape[0]
        k = np.fft.fftfreq(n, 1.0 / n)
        dt_curv = np.fft.ifft(np.fft.fft(curvature) * 1j * k).real
        self.psi0 = speed * (1 + aag.rv0[:, None] * curvature)
        self.psi1 = speed * (1 + aag.rv1[:, None] * curvature)
        self.psi2 = speed * (1 + aag.rv2[:, None] * curvature)
        self.inv_psi0, self.inv_psi1, self.inv_psi2 = 1 / self.psi0, 1 / self.psi1, 1 / self.psi2
        self.DR_psi2 = speed * curvature * np.ones((aag.rv2.shape[0], 1))
        idenom2 = 1.0 / (speed * (1 + aag.rv2[:, None] * curvature) ** 3)
        self.ipsi_DR_ipsi_DT_psi2 = dt_curv * idenom2
        self.ipsi_DT_ipsi_DR_psi2 = dt_curv * idenom2


# ---- Nyquist-dropping transforms ------------------------------------------
def mfft(f):
    n = f.shape[1]
    fh = np.fft.fft(f)
    return np.concatenate([fh[:, :n // 2], fh[:, n // 2 + 1:]], axis=1)


def mifft(fh):
    ns = fh.shape[1]
    n = ns + 1
    t = np.zeros((fh.shape[0], n), dtype=complex)
    t[:, :n // 2] = fh[:, :n // 2]
    t[:, n // 2 + 1:] = fh[:, n // 2:]
    return np.fft.ifft(t)


def gmres_right(apply, prec, b, tol, maxiter, restart):
    """Right-preconditioned restarted GMRES (MGS).  Returns x, residual history."""
    n = b.shape[0]
    x = np.zeros(n, dtype=complex)
    bn = np.linalg.norm(b)
    hist = []
    if bn == 0:
        return x, hist
    it = 0
    while it < maxiter:
        r = b - apply(x) if it else b.copy()
        beta = np.linalg.norm(r)
        if beta <= tol * bn:
            break
        V, Z = [r / beta], []
        H = np.zeros((restart + 1, restart), dtype=complex)
        g = np.zeros(restart + 1, dtype=complex)
        g[0] = beta
        for k in range(restart):
            z = prec(V[k])
            Z.append(z)
            w = apply(z)
            for i in range(k + 1):
                H[i, k] = np.vdot(V[i], w)
                w = w - H[i, k] * V[i]
            H[k + 1, k] = np.linalg.norm(w)
            V.append(w / H[k + 1, k])
            y, *_ = np.linalg.lstsq(H[:k + 2, :k + 1], g[:k + 2], rcond=None)
            rr = np.linalg.norm(H[:k + 2, :k + 1] @ y - g[:k + 2])
            hist.append(rr / bn)
            it += 1
            if rr <= tol * bn or it >= maxiter:
                break
        x = x + sum(yi * zi for yi, zi in zip(y, Z))
        if hist[-1] <= tol:
            break
    return x, hist


class ScalarSolver:
    """(k^2 - Lap) u = f on the annulus; k = 0 and f -> -f gives Poisson."""

    def __init__(self, aag, k, ia=1.0, ib=0.0, oa=1.0, ob=0.0):
        self.aag, self.k = aag, k
        self.M, self.n, self.ns = aag.M, aag.n, aag.ns
        CO = aag.CO
        self.ibc = ia * CO.ibc_dirichlet + ib * CO.ibc_neumann
        self.obc = oa * CO.obc_dirichlet + ob * CO.obc_neumann
        M = self.M
        kinv = []
        for i in range(self.ns):
            LL = (1.0 / aag.approx_psi2)[:, None] * (CO.D12 @ (aag.approx_psi1[:, None] * CO.D01)) \
                - aag.ks[i] ** 2 * (CO.R12 @ ((1.0 / aag.approx_psi1)[:, None] * CO.R01))
            K = np.empty((M, M))
            K[:M - 2] = k ** 2 * CO.R02 - LL
            K[M - 2] = self.ibc
            K[M - 1] = self.obc
            kinv.append(np.linalg.inv(K))
        self.kinv = np.stack(kinv)

    def _fm(self, fh, m):
        return np.fft.fft(m * np.fft.ifft(fh))

    def apply(self, uh, rag):
        CO, aag = self.aag.CO, self.aag
        uh = uh.reshape(self.M, self.ns)
        uh_t = CO.R01 @ (uh * aag.iks)
        uh_tt = CO.R12 @ (self._fm(uh_t, rag.inv_psi1) * aag.iks)
        uh_rr = CO.D12 @ self._fm(CO.D01 @ uh, rag.psi1)
        luh = self._fm(uh_rr + uh_tt, rag.inv_psi2)
        fuh = self.k ** 2 * (CO.R02 @ uh) - luh
        return np.concatenate([fuh.ravel(), (self.ibc @ uh).ravel(), (self.obc @ uh).ravel()])

    def precondition(self, fh):
        x = fh.reshape(self.M, self.ns)
        out = np.einsum("ijk,ki->ji", self.kinv, x)
        return out.ravel()

    def solve(self, rag, f, ig, og, tol=1e-12, maxiter=200, restart=50, negate_f=False):
        CO = self.aag.CO
        ff = np.concatenate([(CO.R02 @ (-f if negate_f else f)).ravel(), ig, og])
        ffh = np.fft.fft(ff.reshape(self.M, self.n)).ravel()
        x, hist = gmres_right(lambda v: self.apply(v, rag), self.precondition, ffh, tol, maxiter,
                              restart)
        self.iterations_last_call = len(hist)
        return np.fft.ifft(x.reshape(self.M, self.ns)).real


class StokesSolver:
    def __init__(self, aag, mu=1.0):
        assert aag.ns == aag.n - 1, "Stokes uses the Nyquist-dropping geometry"
        self.aag, self.mu = aag, mu
        self.M, self.n, self.ns = aag.M, aag.n, aag.ns
        M, ns, CO = self.M, self.ns, aag.CO
        self.NU, self.NP = M * ns, (M - 1) * ns
        self.NB = 2 * self.NU + self.NP
        ap0, ap1 = aag.approx_psi0, aag.approx_psi1
        ai1, ai2 = 1.0 / aag.approx_psi1, 1.0 / aag.approx_psi2
        kinv = []
        for i in range(ns):
            ks = aag.ks[i]
            K = np.zeros((3 * M - 1, 3 * M - 1), dtype=complex)
            LL = ai2[:, None] * (CO.D12 @ (ap1[:, None] * CO.D01)) \
                - ks ** 2 * (CO.R12 @ (ai1[:, None] * CO.R01))
            A = -LL + (ai2 ** 2)[:, None] * CO.R02
            Bm = (2 * ai2 ** 2)[:, None] * (CO.R02 * (1j * ks))
            K[0:M - 2, 0:M] = mu * A
            K[0:M - 2, M:2 * M] = mu * Bm
            K[0:M - 2, 2 * M:] = CO.D12
            K[M - 2, 0:M] = CO.ibc_dirichlet
            K[M - 1, 0:M] = CO.obc_dirichlet
            K[M:2 * M - 2, 0:M] = -mu * Bm
            K[M:2 * M - 2, M:2 * M] = mu * A
            K[M:2 * M - 2, 2 * M:] = ai2[:, None] * (CO.R12 * (1j * ks))
            K[2 * M - 2, M:2 * M] = CO.ibc_dirichlet
            K[2 * M - 1, M:2 * M] = CO.obc_dirichlet
            K[2 * M:, 0:M] = ai1[:, None] * (CO.D01 * ap0[None, :])
            K[2 * M:, M:2 * M] = ai1[:, None] * (CO.R01 * (1j * ks))
            if i == 0:
                K[2 * M:, 2 * M:] += CO.VI1[0]
            kinv.append(np.linalg.inv(K))
        self.kinv = np.stack(kinv)

    def _split(self, v):
        M, ns = self.M, self.ns
        return (v[:self.NU].reshape(M, ns), v[self.NU:2 * self.NU].reshape(M, ns),
                v[2 * self.NU:].reshape(M - 1, ns))

    def apply(self, uuh, rag):
        CO, aag, mu = self.aag.CO, self.aag, self.mu
        iks = 1j * aag.ks
        urh, uth, ph = self._split(uuh)
        combo1 = 2 * rag.DR_psi2 * rag.inv_psi2 ** 2
        combo2 = rag.DR_psi2 ** 2 * rag.inv_psi2 ** 2
        ibr, ibt = CO.ibc_dirichlet @ urh, CO.ibc_dirichlet @ uth
        obr, obt = CO.obc_dirichlet @ urh, CO.obc_dirichlet @ uth
        r = lambda fh: mifft(fh).real
        ur, ut, p = r(urh), r(uth), r(ph)
        dur, dut = r(iks * urh), r(iks * uth)
        W1r, W1t = CO.R02 @ ur, CO.R02 @ ut
        t1 = (CO.R02 @ dut) * combo1
        t2 = W1r * combo2
        t3 = W1t * rag.ipsi_DR_ipsi_DT_psi2
        t4 = CO.D12 @ p
        ur_t = CO.R01 @ dur
        ur_tt = CO.R12 @ r(mfft(ur_t * rag.inv_psi1) * iks)
        ur_rr = CO.D12 @ ((CO.D01 @ ur) * rag.psi1)
        lap_ur = (ur_rr + ur_tt) * rag.inv_psi2
        frh = mfft(mu * (-lap_ur + t1 + t2 + t3) + t4)
        t1 = (CO.R02 @ dur) * combo1
        t2 = W1t * combo2
        t3 = W1r * rag.ipsi_DT_ipsi_DR_psi2
        t4 = (CO.R12 @ r(ph * iks)) * rag.inv_psi2
        W2 = CO.R01 @ dut
        ut_tt = CO.R12 @ r(mfft(W2 * rag.inv_psi1) * iks)
        ut_rr = CO.D12 @ ((CO.D01 @ ut) * rag.psi1)
        lap_ut = (ut_rr + ut_tt) * rag.inv_psi2
        fth = mfft(mu * (-lap_ut - t1 + t2 - t3) + t4)
        fph = mfft((CO.D01 @ (ur * rag.psi0) + W2) * rag.inv_psi1)
        fph[:, 0] += (CO.VI1 @ ph)[0, 0]
        return np.concatenate([frh.ravel(), ibr.ravel(), obr.ravel(), fth.ravel(), ibt.ravel(),
                               obt.ravel(), fph.ravel()])

    def precondition(self, ffh):
        frh, fth, fph = self._split(ffh)
        ff = np.vstack([frh, fth, fph])          # (3M-1, ns)
        out = np.einsum("ijk,ki->ji", self.kinv, ff)
        return out.ravel()

    def solve(self, rag, fr, ft, irg, itg, org, otg, tol=1e-12, maxiter=300, restart=100):
        CO, M, n = self.aag.CO, self.M, self.n
        ffr = np.concatenate([(CO.R02 @ fr).ravel(), irg, org]).reshape(M, n)
        fft_ = np.concatenate([(CO.R02 @ ft).ravel(), itg, otg]).reshape(M, n)
        b = np.concatenate([mfft(ffr).ravel(), mfft(fft_).ravel(), np.zeros(self.NP, dtype=complex)])
        x, hist = gmres_right(lambda v: self.apply(v, rag), self.precondition, b, tol, maxiter,
                              restart)
        self.iterations_last_call = len(hist)
        urh, uth, ph = self._split(x)
        return mifft(urh).real, mifft(uth).real, CO.P10 @ mifft(ph).real
