"""ipde/grid_evaluators/modified_helmholtz_grid_evaluator.py API (reference :19-30)."""

from .scalar_grid_evaluator import (ScalarGridBackend, ScalarFreespaceGridEvaluator,
                                    ScalarPeriodicGridEvaluator)
from ..layer_potentials import modified_helmholtz_apply


class ModifiedHelmholtzGridBackend(ScalarGridBackend):
    def __init__(self, h, spread_width, helmholtz_k, funcgen_tol=1e-10, inline_core=True,
                 method='dense'):
        super().__init__(h, spread_width, {'helmholtz_k': helmholtz_k}, funcgen_tol, inline_core,
                         method)


class ModifiedHelmholtzFreespaceGridEvaluator(ScalarFreespaceGridEvaluator):
    def __init__(self, backend, xv, yv, allow_rectangular=False):
        self.k = backend.kernel_kwargs['helmholtz_k']
        super().__init__(backend, xv, yv, allow_rectangular)

    def _apply(self, sx, sy, ch):
        return modified_helmholtz_apply(sx, sy, self.targets.x, self.targets.y, self.k, w_sigma=ch)


class ModifiedHelmholtzPeriodicGridEvaluator(ScalarPeriodicGridEvaluator):
    pass
