"""Grid evaluator for the stokeslet sum with pressure — the reference has none (its
Stokes solver evaluates through an FMM, ipde/solvers/internals/stokes.py:25-35); this one
follows the pattern of its scalar evaluators (grid_evaluators/scalar_grid_evaluator.py):
a backend holding the split parameters and an evaluator bound to a grid."""
import numpy as np

from .scalar_grid_evaluator import ScalarGridBackend


class StokesGridBackend(ScalarGridBackend):
    def __init__(self, h, spread_width=24):
        super().__init__(h, spread_width, {}, method='ewald')


class StokesFreespaceGridEvaluator(object):
    """__call__(src (2, N), forces (2, N) already weight-multiplied) -> (u, v, p), each
    (nx, ny), device tensors (device_result=True) or numpy."""

    def __init__(self, backend, xv, yv):
        from .ewald import StokesFreespaceEwald
        self.backend = backend
        self.xv = np.asarray(xv, dtype=float)
        self.yv = np.asarray(yv, dtype=float)
        backend.check_either(self.xv, self.yv)
        self._ewald = StokesFreespaceEwald(backend.core, self.xv, self.yv)

    def __call__(self, src, forces, device_result=False):
        src = np.asarray(src, dtype=float)
        f = np.asarray(forces, dtype=float).reshape(2, -1)
        out = self._ewald(src[0], src[1], f[0], f[1])
        return out if device_result else tuple(o.cpu().numpy() for o in out)
