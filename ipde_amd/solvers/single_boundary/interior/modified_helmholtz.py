"""Single-boundary interior modified-Helmholtz solver, old call form (reference
ipde/solvers/single_boundary/interior/modified_helmholtz.py:15-102)."""
from ._adapter import SingleBoundaryAdapter


class ModifiedHelmholtzSolver(SingleBoundaryAdapter):
    def __init__(self, ebdy, k, solver_type='spectral', AMHS=None):
        self.k = k
        self._AMHS = AMHS
        super().__init__(ebdy, solver_type)

    def _make_solver(self, ebdyc, solver_type):
        from ...multi_boundary.modified_helmholtz import ModifiedHelmholtzSolver as Multi
        return Multi(ebdyc, k=self.k, solver_type=solver_type)
