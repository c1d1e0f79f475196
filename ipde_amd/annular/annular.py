"""ipde/annular/annular.py: the Nyquist-dropping geometry (ns = n-1), used by the
vector (Stokes) helpers (reference ipde/solvers/internals/vector.py:2)."""
from ._geometry import ChebyshevOperators, RealAnnularGeometry
from ._geometry import ApproximateAnnularGeometry as _AAG


class ApproximateAnnularGeometry(_AAG):
    def __init__(self, n, M, width, approx_r):
        super().__init__(n, M, width, approx_r, keep_nyquist=False)


__all__ = ["ChebyshevOperators", "ApproximateAnnularGeometry", "RealAnnularGeometry"]
