"""Single-boundary interior Poisson solver, old call form (reference
ipde/solvers/single_boundary/interior/poisson.py:20-63; examples/poisson_for_paper.py:63 calls it as
`PoissonSolver(ebdy, MOL.bump, bump_loc=(...), solver_type=...)`): `bump` is the bump FUNCTION on
[0, 1], `bump_loc` its centre; the compatibility condition's bump is readied on the one-boundary
collection with the annulus width as its radius (reference :59-61)."""
from ._adapter import SingleBoundaryAdapter, annular_solver_holder


class PoissonSolver(SingleBoundaryAdapter):
    def __init__(self, ebdy, bump, bump_loc=None, solver_type='spectral', APS=None):
        if bump is not None and not callable(bump):
            # (a mollifier object was accepted here before: take its bump function)
            bump = getattr(bump, 'bump', None)
            if not callable(bump):
                raise TypeError('bump must be a function on [0, 1] (e.g. SlepianMollifier(...).bump)')
        self._bump, self._bump_loc = bump, bump_loc
        self._holder = annular_solver_holder(APS, ebdy, 'APS')
        super().__init__(ebdy, solver_type)

    def _make_solver(self, ebdyc, solver_type):
        from ...multi_boundary.poisson import PoissonSolver as Multi
        if not ebdyc.bumpy_readied and self._bump is not None:
            rw = self.ebdy.radial_width
            loc = self._bump_loc
            if loc is None:
                g = self.ebdy.grid
                loc = (g.x_bounds[1] - rw, g.y_bounds[1] - rw)
            ebdyc.ready_bump(self._bump, loc, rw)
        return Multi(ebdyc, solver_type=solver_type,
                     AS_list=None if self._holder is None else [self._holder])
