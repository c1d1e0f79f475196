"""Single-boundary interior Poisson solver, old call form (reference
ipde/solvers/single_boundary/interior/poisson.py:14-104).  The collection needs its bump
(`ebdy.solo_collection().ready_bump(...)`) for the compatibility condition."""
from ._adapter import SingleBoundaryAdapter


class PoissonSolver(SingleBoundaryAdapter):
    def __init__(self, ebdy, MOL=None, bump_loc=None, solver_type='spectral', APS=None):
        self._MOL, self._bump_loc = MOL, bump_loc
        super().__init__(ebdy, solver_type)

    def _make_solver(self, ebdyc, solver_type):
        from ...multi_boundary.poisson import PoissonSolver as Multi
        if not ebdyc.bumpy_readied and self._MOL is not None:
            rw = self.ebdy.radial_width
            loc = self._bump_loc
            if loc is None:
                g = self.ebdy.grid
                loc = (g.x_bounds[1] - rw, g.y_bounds[1] - rw)
            ebdyc.ready_bump(self._MOL.bump, loc, rw)
        return Multi(ebdyc, solver_type=solver_type)
