"""The OLD single-boundary solver call form of the reference
(ipde/solvers/single_boundary/interior/{poisson,modified_helmholtz}.py:15-102, used by
examples/interior_modified_helmholtz.py:41-44,62-63):

    ebdy = EmbeddedBoundary(bdy, True, M, h, pad_zone, MOL.step)
    ebdy.register_grid(grid)
    solver = ModifiedHelmholtzSolver(ebdy, k, solver_type='spectral')
    ue, uer = solver(f, fr, tol=1e-12)        # full-grid array, (M, N) radial array
    bv = solver.get_bv(uer);  solver.radp, solver.gridp, solver.gridpa

as an adapter over the multi-boundary solvers (one-boundary collection): same numbers, one
code path, the GPU kernels underneath."""
import numpy as np

from ....ebdy_collection import EmbeddedBoundaryCollection
from ....embedded_function import EmbeddedFunction
from ....pybie2d_compat import PointSet


class _AnnularSolverHolder(object):
    """what the multi-boundary helpers take as `helper`: an object carrying .annular_solver"""
    def __init__(self, annular_solver):
        self.annular_solver = annular_solver


def annular_solver_holder(AS, ebdy, name):
    """The reference's optional pre-built annular solver argument (APS / AMHS; reference
    single_boundary/interior/poisson.py:30-34): reused when it fits the boundary, refused otherwise."""
    if AS is None:
        return None
    AAG = getattr(AS, 'AAG', None)
    if AAG is None or not callable(getattr(AS, 'solve', None)):
        raise TypeError('%s must be an annular solver of this package (has .AAG and .solve)' % name)
    if (AAG.n, AAG.M) != (ebdy.bdy.N, ebdy.M):
        raise ValueError('%s was built for n = %d, M = %d; the boundary has n = %d, M = %d'
                         % (name, AAG.n, AAG.M, ebdy.bdy.N, ebdy.M))
    return _AnnularSolverHolder(AS)


class SingleBoundaryAdapter(object):
    def __init__(self, ebdy, solver_type='spectral'):
        if getattr(ebdy, 'grid', None) is None:
            raise Exception('register a grid first: ebdy.register_grid(grid)')
        self.ebdy = ebdy
        self.interior = ebdy.interior
        self.solver_type = self.type = solver_type
        self.ebdyc = ebdy.solo_collection()
        self.solver = self._make_solver(self.ebdyc, solver_type)
        self.helper = self.solver.helpers[0]
        self.annular_solver = self.helper.annular_solver
        self.RAG = self.helper.RAG
        grid, c = ebdy.grid, self.ebdyc
        self.radp = PointSet(ebdy.radial_x.ravel(), ebdy.radial_y.ravel())
        self.gridp = PointSet(grid.xg[c.phys_not_in_annulus], grid.yg[c.phys_not_in_annulus])
        self.gridpa = PointSet(grid.xg[c.phys], grid.yg[c.phys])

    def _make_solver(self, ebdyc, solver_type):
        raise NotImplementedError

    def get_bv(self, ur):
        return self.helper.get_boundary_values(ur)

    def get_bn(self, ur):
        return self.helper.get_boundary_normal_derivatives(ur)

    def __call__(self, f, fr, **kwargs):
        """f: (Nx, Ny) forcing on the grid (values outside the physical domain ignored),
        fr: (M, N) forcing on the radial grid -> (u on the grid, zero outside; u radial)"""
        c = self.ebdyc
        ef = EmbeddedFunction(c)
        ef.load_data(np.asarray(f, dtype=float)[c.phys], [np.asarray(fr, dtype=float)])
        ue = self.solver(ef, **kwargs)
        self.iterations_last_call = self.solver.iteration_counts[0]
        uc = np.zeros(self.ebdy.grid.shape)
        uc[c.phys] = ue['grid']
        return uc, np.array(ue[0])
