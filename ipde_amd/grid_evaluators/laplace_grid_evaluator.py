"""ipde/grid_evaluators/laplace_grid_evaluator.py API (reference :35-45), exact sum."""
import numpy as np

from .scalar_grid_evaluator import (ScalarGridBackend, ScalarFreespaceGridEvaluator,
                                    ScalarPeriodicGridEvaluator)
from ..layer_potentials import laplace_apply

scale = -1.0 / (2 * np.pi)


def gf(r):
    """Green's function the evaluator sums (reference :8-12)."""
    return scale * np.log(r)


class LaplaceGridBackend(ScalarGridBackend):
    def __init__(self, h, spread_width, funcgen_tol=1e-10, inline_core=True, method='dense'):
        super().__init__(h, spread_width, {}, funcgen_tol, inline_core, method)


class LaplaceFreespaceGridEvaluator(ScalarFreespaceGridEvaluator):
    def __init__(self, backend, xv, yv, allow_rectangular=False):
        super().__init__(backend, xv, yv, allow_rectangular)

    PATCH_TARGETS = True     # the full grid is all full tiles: ipde_laplace_apply_patches

    def _apply(self, sx, sy, ch):
        plan = self.targets.plan()
        if plan is not None:
            from .. import target_plan
            return target_plan.laplace_apply(plan, sx, sy, w_sigma=ch)
        return laplace_apply(sx, sy, self.targets.x, self.targets.y, w_sigma=ch)


class LaplacePeriodicGridEvaluator(ScalarPeriodicGridEvaluator):
    pass
