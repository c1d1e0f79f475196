"""AnnularPoissonSolver — mirrors ipde/annular/poisson.py:3-21 (k = 0, f -> -f)."""
from .modified_helmholtz import AnnularModifiedHelmholtzSolver


class AnnularPoissonSolver(AnnularModifiedHelmholtzSolver):
    """Solves L u = f on the annulus with the Robin data of the parent class."""

    def __init__(self, AAG, ia=1.0, ib=0.0, oa=1.0, ob=0.0, ctx=None):
        super().__init__(AAG, 0.0, ia=ia, ib=ib, oa=oa, ob=ob, ctx=ctx)

    def solve(self, RAG, f, ig, og, ia=None, ib=None, oa=None, ob=None, verbose=False, **kwargs):
        return super().solve(RAG, f, ig=ig, og=og, ia=ia, ib=ib, oa=oa, ob=ob, verbose=verbose,
                             _negate_f=True, **kwargs)
