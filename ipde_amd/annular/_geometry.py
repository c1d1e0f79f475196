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


class ApproximateAnnularGeometry(object):
    """n tangential points, M radial Chebyshev modes, annulus of `width` around a
    circle of radius approx_r.  keep_nyquist=True is the reference's annular_full
    (ns = n), False its annular (ns = n-1)."""

    def __init__(self, n, M, width, approx_r, keep_nyquist=False):
        self.n = n
        self.M = M
        self.radius = approx_r
        self.width = width
        self.radial_h = self.width / self.M
        self.tangent_h = 2 * np.pi / n
        self.n2 = int(self.n / 2)
        self.k = np.fft.fftfreq(self.n, 1.0 / self.n)
        if keep_nyquist:
            self.ns = self.n
            self.ks = self.k
        else:
            self.ns = self.n - 1
            self.ks = np.concatenate([self.k[:self.n2], self.k[self.n2 + 1:]])
        self.iks = 1j * self.ks
        _, self.rv0, rat0 = get_chebyshev_nodes(-self.width, 0.0, self.M - 0)
        _, self.rv1, rat1 = get_chebyshev_nodes(-self.width, 0.0, self.M - 1)
        _, self.rv2, rat2 = get_chebyshev_nodes(-self.width, 0.0, self.M - 2)
        self.ratio = -rat0
        self.approx_psi0 = self.radius + self.rv0
        self.approx_psi1 = self.radius + self.rv1
        self.approx_psi2 = self.radius + self.rv2
        self.approx_inv_psi0 = 1.0 / self.approx_psi0
        self.approx_inv_psi1 = 1.0 / self.approx_psi1
        self.approx_inv_psi2 = 1.0 / self.approx_psi2
        self.CO = ChebyshevOperators(M, self.ratio)


class RealAnnularGeometry(object):
    """Metric fields psi_k = speed (1 + r_k curvature) of the true annulus."""

    def __init__(self, speed, curvature, AAG):
        n = curvature.shape[0]
        k = np.fft.fftfreq(n, 1.0 / n)
        dt_curvature = np.fft.ifft(np.fft.fft(curvature) * 1j * k).real
        rv0, rv1, rv2 = AAG.rv0, AAG.rv1, AAG.rv2
        self.psi0 = speed * (1 + rv0[:, None] * curvature)
        self.psi1 = speed * (1 + rv1[:, None] * curvature)
        self.psi2 = speed * (1 + rv2[:, None] * curvature)
        self.inv_psi0 = 1.0 / self.psi0
        self.inv_psi1 = 1.0 / self.psi1
        self.inv_psi2 = 1.0 / self.psi2
        self.DR_psi2 = speed * curvature * np.ones(rv2[:, None].shape)
        idenom2 = 1.0 / (speed * (1 + rv2[:, None] * curvature) ** 3)
        # the reference computes two candidate forms and keeps these (annular.py:107-108)
        self.ipsi_DR_ipsi_DT_psi2 = dt_curvature * idenom2
        self.ipsi_DT_ipsi_DR_psi2 = dt_curvature * idenom2
