"""ModifiedHelmholtzHelper — mirrors ipde/solvers/internals/modified_helmholtz.py:13-37."""
from .scalar import ScalarHelper
from ...annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
from ...layer_potentials import make_modified_helmholtz_layer_apply
from ...qfs import Modified_Helmholtz_QFS, QFS_Boundary


class ModifiedHelmholtzHelper(ScalarHelper):
    """Inhomogeneous modified-Helmholtz solver on a general domain (per boundary)."""

    def __init__(self, ebdy, annular_solver=None, k=1.0, source_upsample_factor=1.0,
                 grid_backend='hip', private_ctx=False):
        self.k = k
        self.source_upsample_factor = source_upsample_factor
        super().__init__(ebdy, annular_solver, grid_backend, private_ctx)

    def _define_annular_solver(self):
        self.annular_solver = AnnularModifiedHelmholtzSolver(self.AAG, k=self.k, ctx=self._annular_ctx())

    def _get_qfs(self):
        q = self.ebdy.interface_qfs
        if self.source_upsample_factor and self.source_upsample_factor > 1:
            q = QFS_Boundary(self.ebdy.interface, eps=self.ebdy.qfs_tolerance,
                             forced_source_upsampling_factor=int(self.source_upsample_factor))
        self.interface_qfs_g = Modified_Helmholtz_QFS(self.ebdy.interface, self.interior, True, True,
                                                      self.k, qfs_boundary=q)
        self.interface_qfs_r = Modified_Helmholtz_QFS(self.ebdy.interface, not self.interior, True,
                                                      True, self.k, qfs_boundary=q)

    def _define_layer_apply(self):
        self.Layer_Apply = make_modified_helmholtz_layer_apply(self.k)
