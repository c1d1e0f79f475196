"""Set-up objects of the annular solvers (host numpy; the reference builds these
once per boundary as well: SURVEY §8 a9).

ChebyshevOperators           ipde/annular/annular.py:7-50
ApproximateAnnularGeometry   ipde/annular/annular.py:52-85 / annular_full.py:60-85
RealAnnularGeometry          ipde/annular/annular.py:87-108
"""
import numpy as np
from numpy.polynomial import chebyshev as cheb

from ..utilities import get_chebyshev_nodes


def _embed(rows, cols):
    t = np.zeros([rows, cols], dtype=float)
    np.fill_diagonal(t, 1.0)
    return t


class ChebyshevOperators(object):
    """Chebyshev-Gauss collocation operators on the nested grids M, M-1, M-2.

    M: number of radial modes; rat: ratio of the annulus width to [-1, 1]."""

    def __init__(self, M, rat):
        self.M = M
        nodes = [cheb.chebgauss(M - i)[0] for i in range(3)]
        self.V0, self.V1, self.V2 = (cheb.chebvander(x, M - 1 - i) for i, x in enumerate(nodes))
        self.VI0, self.VI1, self.VI2 = (np.linalg.inv(V) for V in (self.V0, self.V1, self.V2))
        DC01 = cheb.chebder(np.eye(M)) / rat
        DC12 = cheb.chebder(np.eye(M - 1)) / rat
        DC00 = np.vstack([DC01, np.zeros(M)])
        self.D00 = self.V0 @ DC00 @ self.VI0
        self.D01 = self.V1 @ DC01 @ self.VI0
        self.D12 = self.V2 @ DC12 @ self.VI1
        self.ibc_dirichlet = cheb.chebvander(1, M - 1) @ self.VI0
        self.obc_dirichlet = cheb.chebvander(-1, M - 1) @ self.VI0
        self.ibc_neumann = self.ibc_dirichlet @ self.D00
        self.obc_neumann = self.obc_dirichlet @ self.D00
        self.R01 = self.V1 @ _embed(M - 1, M) @ self.VI0
        self.R12 = self.V2 @ _embed(M - 2, M - 1) @ self.VI1
        self.R02 = self.R12 @ self.R01
        self.P10 = self.V0 @ _embed(M, M - 1) @ self.VI1


NESTED = 3      # the radial collocation grids: M, M - 1 and M - 2 Chebyshev-Gauss nodes


class ApproximateAnnularGeometry(object):
    """n tangential points, M radial Chebyshev modes, annulus of `width` around a
    circle of radius approx_r.  keep_nyquist=True is the reference's annular_full
    (ns = n), False its annular (ns = n-1).

    The per-grid attributes the solvers read (`rv0..2`, `approx_psi0..2`, `approx_inv_psi0..2`: radial node
    offsets, radii of the approximating circles and their reciprocals on the grids of M, M - 1, M - 2
    nodes) are filled by one pass over the nested grids."""

    def __init__(self, n, M, width, approx_r, keep_nyquist=False):
        self.n, self.M = n, M
        self.radius, self.width = approx_r, width
        self.radial_h, self.tangent_h = width / M, 2 * np.pi / n
        self.n2 = n // 2
        self.k = np.fft.fftfreq(n, 1.0 / n)
        drop = [] if keep_nyquist else [self.n2]           # tangential modes the solver keeps
        self.ks = np.delete(self.k, drop)
        self.ns = self.ks.shape[0]
        self.iks = 1j * self.ks
        for g in range(NESTED):
            _, rv, rat = get_chebyshev_nodes(-width, 0.0, M - g)
            if g == 0:
                self.ratio = -rat
            setattr(self, 'rv%d' % g, rv)
            setattr(self, 'approx_psi%d' % g, approx_r + rv)
            setattr(self, 'approx_inv_psi%d' % g, 1.0 / (approx_r + rv))
        self.CO = ChebyshevOperators(M, self.ratio)


class RealAnnularGeometry(object):
    """Metric fields of the true annulus on the nested radial grids: psi_g = speed (1 + r_g curvature) and its
    reciprocal for g = 0, 1, 2 (`psi0..2`, `inv_psi0..2`), and on the finest-but-two grid the derivative
    terms of the Laplacian's cross part."""

    def __init__(self, speed, curvature, AAG):
        n = curvature.shape[0]
        k = np.fft.fftfreq(n, 1.0 / n)
        dt_curvature = np.fft.ifft(np.fft.fft(curvature) * 1j * k).real
        for g in range(NESTED):
            stretch = 1 + getattr(AAG, 'rv%d' % g)[:, None] * curvature
            setattr(self, 'psi%d' % g, speed * stretch)
            setattr(self, 'inv_psi%d' % g, 1.0 / (speed * stretch))
        rows = AAG.rv2.shape[0]
        self.DR_psi2 = np.tile(speed * curvature, (rows, 1)) if np.ndim(speed * curvature) else \
            np.full((rows, 1), float(speed * curvature))
        # the reference computes two candidate forms of the cross terms and keeps ONE expression for both
        # (annular.py:103-108, "these are what work"): dt_curvature / (speed stretch_2^3)
        cross = dt_curvature / (speed * stretch ** 3)
        self.ipsi_DR_ipsi_DT_psi2 = cross
        self.ipsi_DT_ipsi_DR_psi2 = cross.copy()
