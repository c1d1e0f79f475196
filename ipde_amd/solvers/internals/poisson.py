"""PoissonHelper — mirrors ipde/solvers/internals/poisson.py:14-36.  The
`Layer_Apply` closure (:27-36) is the GPU dense sum."""
from .scalar import ScalarHelper
from ...annular.poisson import AnnularPoissonSolver
from ...layer_potentials import make_laplace_layer_apply
from ...qfs import Laplace_QFS


class PoissonHelper(ScalarHelper):
    """Inhomogeneous Poisson solver on a general domain (per-boundary part)."""

    def __init__(self, ebdy, annular_solver=None, grid_backend='hip', private_ctx=False):
        super().__init__(ebdy, annular_solver, grid_backend, private_ctx)

    def _define_annular_solver(self):
        self.annular_solver = AnnularPoissonSolver(self.AAG, ctx=self._annular_ctx())

    def _get_qfs(self):
        q = self.ebdy.interface_qfs
        self.interface_qfs_g = Laplace_QFS(self.ebdy.interface, self.interior, True, True,
                                           qfs_boundary=q)
        self.interface_qfs_r = Laplace_QFS(self.ebdy.interface, not self.interior, True, True,
                                           qfs_boundary=q)

    def _define_layer_apply(self):
        self.Layer_Apply = make_laplace_layer_apply()
