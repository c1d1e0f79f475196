"""ipde/annular/annular_full.py: keeps the Nyquist mode (ns = n), used by the scalar
helpers (reference ipde/solvers/internals/scalar.py:2-3)."""
from ._geometry import ChebyshevOperators, RealAnnularGeometry
from ._geometry import ApproximateAnnularGeometry as _AAG


class ApproximateAnnularGeometry(_AAG):
    def __init__(self, n, M, width, approx_r):
        super().__init__(n, M, width, approx_r, keep_nyquist=True)


__all__ = ["ChebyshevOperators", "ApproximateAnnularGeometry", "RealAnnularGeometry"]
