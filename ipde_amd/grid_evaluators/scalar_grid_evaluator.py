"""Grid evaluators with the API of ipde/grid_evaluators/scalar_grid_evaluator.py
(reference :50-307) backed by the EXACT dense sum on the MI355X.

The reference's backend is an Ewald-type split (compact screened-kernel spread +
FFT far field) that its own docstring limits to ~10 digits (:60-67), which cannot
meet the 1e-12 parity bar of this path; on the GPU the direct sum over a 2048^2
grid takes a few milliseconds, so the same classes evaluate it exactly.  The
constructor arguments that only tune the Ewald split (spread_width, funcgen_tol,
inline_core) are accepted and recorded, and the input checks raise the same
`Exception`s (:232-244).
"""
import numpy as np

from ..layer_potentials import DeviceTargets


class ScalarGridBackend(object):
    def __init__(self, h, spread_width, kernel_kwargs=None, funcgen_tol=1e-10, inline_core=True):
        self.h = h
        self.spread_width = spread_width
        self.kernel_kwargs = {} if kernel_kwargs is None else kernel_kwargs
        self.funcgen_tol = funcgen_tol
        self.inline_core = inline_core

    def initialize_periodic(self):
        pass

    def initialize_freespace(self):
        pass

    def check_periodic(self, xv, yv):
        self.check_either(xv, yv)

    def check_freespace(self, xv, yv):
        self.check_either(xv, yv)
        if xv.size != yv.size:
            raise Exception('Square grid required for freespace evaluator')

    def check_either(self, xv, yv):
        xh = xv[1] - xv[0]
        if np.abs(xh - self.h) > 1e-15:
            raise Exception('h of input xv vector not same as backend')
        yh = yv[1] - yv[0]
        if np.abs(yh - self.h) > 1e-15:
            raise Exception('h of input yv vector not same as backend')


class ScalarFreespaceGridEvaluator(object):
    """__call__(src (2,N), ch) -> (n, n) grid of sum_j G(|x - s_j|) ch_j, `ch` already
    weight-multiplied (reference :299-307; callers pass ch*weights,
    multi_boundary/poisson.py:44-47)."""

    def __init__(self, backend, xv, yv):
        self.backend = backend
        self.xv = np.asarray(xv, dtype=float)
        self.yv = np.asarray(yv, dtype=float)
        self.backend.check_freespace(self.xv, self.yv)
        self.backend.initialize_freespace()
        self.h = self.backend.h
        self.n = self.xv.size
        xg, yg = np.meshgrid(self.xv, self.yv, indexing='ij')
        self.targets = DeviceTargets(xg.ravel(), yg.ravel())   # resident across calls

    def _apply(self, sx, sy, ch):
        raise NotImplementedError

    def __call__(self, src, ch, device_result=False):
        src = np.asarray(src, dtype=float)
        out = self._apply(src[0], src[1], np.asarray(ch, dtype=float))
        out = out.view(self.n, self.n)
        return out if device_result else out.cpu().numpy()


class ScalarPeriodicGridEvaluator(object):
    def __init__(self, backend, xv, yv):
        raise NotImplementedError(
            "the periodic-image evaluator (reference scalar_grid_evaluator.py:246-264) is not "
            "built; the solvers on this path only use the free-space evaluator "
            "(multi_boundary/poisson.py:41-43)")
