"""StokesHelper — mirrors ipde/solvers/internals/stokes.py:10-35.  `Layer_Apply`
(:25-35, an SFMM call with velocity + stress in the reference) is the GPU stokeslet sum
returning (u, v, p)."""
from .vector import VectorHelper
from ...annular.stokes import AnnularStokesSolver
from ...layer_potentials import make_stokes_layer_apply
from ...qfs import Stokes_QFS


class StokesHelper(VectorHelper):
    """Inhomogeneous Stokes solver on a general domain (per-boundary part)."""

    def __init__(self, ebdy, annular_solver=None):
        super().__init__(ebdy, annular_solver)

    def _define_annular_solver(self):
        self.annular_solver = AnnularStokesSolver(self.AAG, mu=1.0)

    def _get_qfs(self):
        q = self.ebdy.interface_qfs
        self.interface_qfs_g = Stokes_QFS(self.ebdy.interface, self.interior, True, True,
                                          qfs_boundary=q)
        self.interface_qfs_r = Stokes_QFS(self.ebdy.interface, not self.interior, True, True,
                                          qfs_boundary=q)

    def _define_layer_apply(self):
        self.Layer_Apply = make_stokes_layer_apply()
