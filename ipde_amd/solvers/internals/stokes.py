"""StokesHelper — mirrors ipde/solvers/internals/stokes.py:10-35.  `Layer_Apply`
(:25-35, an SFMM call with velocity + stress in the reference) is the GPU stokeslet sum
returning (u, v, p)."""
from .vector import VectorHelper
from ...annular.stokes import AnnularStokesSolver
from ...layer_potentials import make_stokes_layer_apply
from ...qfs import Stokes_QFS


class StokesHelper(VectorHelper):
    """Inhomogeneous Stokes solver on a general domain (per-boundary part)."""

    def __init__(self, ebdy, annular_solver=None, private_ctx=False):
        # private_ctx: give the annular solver a library context of its own (stream, work
        # buffers, FFT plans), so that the solver of a multiply connected domain can run the
        # annular solves of its boundaries concurrently from separate host threads
        self._private_ctx = private_ctx
        super().__init__(ebdy, annular_solver)

    def _define_annular_solver(self):
        ctx = None
        if self._private_ctx:
            from ...device import private_context
            ctx = private_context()
        self.annular_solver = AnnularStokesSolver(self.AAG, mu=1.0, ctx=ctx)

    def _get_qfs(self):
        q = self.ebdy.interface_qfs
        self.interface_qfs_g = Stokes_QFS(self.ebdy.interface, self.interior, True, True,
                                          qfs_boundary=q)
        self.interface_qfs_r = Stokes_QFS(self.ebdy.interface, not self.interior, True, True,
                                          qfs_boundary=q)

    def _define_layer_apply(self):
        self.Layer_Apply = make_stokes_layer_apply()
