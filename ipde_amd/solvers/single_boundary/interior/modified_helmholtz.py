"""Single-boundary interior modified-Helmholtz solver, old call form (reference
ipde/solvers/single_boundary/interior/modified_helmholtz.py:15-102)."""
from ._adapter import SingleBoundaryAdapter, annular_solver_holder


class ModifiedHelmholtzSolver(SingleBoundaryAdapter):
    def __init__(self, ebdy, k, solver_type='spectral', AMHS=None):
        self.k = k
        self._holder = annular_solver_holder(AMHS, ebdy, 'AMHS')
        if self._holder is not None and getattr(AMHS, 'k', k) != k:
            raise ValueError('AMHS was built for k = %r, the solver is asked for k = %r' % (AMHS.k, k))
        super().__init__(ebdy, solver_type)

    def _make_solver(self, ebdyc, solver_type):
        from ...multi_boundary.modified_helmholtz import ModifiedHelmholtzSolver as Multi
        return Multi(ebdyc, k=self.k, solver_type=solver_type,
                     helpers=None if self._holder is None else [self._holder])
