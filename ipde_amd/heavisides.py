"""`ipde.heavisides.SlepianMollifier` (module missing from the reference tree but
imported by every example, e.g. examples/interior_poisson.py:9; its live equivalent
is ipde/slepian/chebeval_bump_step.py:23-44).

Construction follows ipde/slepian/function_generator_bump_step.py:38-50: the bump is
the discrete prolate spheroidal (Slepian) window `dpss(N, 0.25 r)` on [-1, 1], the
step its normalised running integral.  Both are represented as Chebyshev series so
that they are smooth functions evaluated consistently everywhere (the grid/radial
partition of unity needs consistency, not a particular window).  Host numpy: evaluated
once at set-up on O(annulus) points.
"""
import numpy as np
from numpy.polynomial import chebyshev as C


def _dpss(n, nw):
    """scipy.signal.windows.dpss(n, nw).  `import scipy.signal` pulls in scipy.stats,
    .optimize, .interpolate, ... (0.3 s, an eighth of a cold 2048^2 solve); the window
    functions live in one file that only needs scipy.linalg / special / fft, so that file is
    loaded on its own when it is where this scipy keeps it."""
    try:
        import importlib.util
        import os
        import scipy
        path = os.path.join(os.path.dirname(scipy.__file__), "signal", "windows", "_windows.py")
        spec = importlib.util.spec_from_file_location("_ipde_scipy_windows", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.dpss(n, nw)
    except Exception:
        import scipy.signal
        return scipy.signal.windows.dpss(n, nw)

_cache = {}


def _build(r, nsamp=1000, deg=200):
    key = (float(r), nsamp, deg)
    if key in _cache:
        return _cache[key]
    # sample the dpss window at Chebyshev-Lobatto-like points by interpolating the
    # (very smooth) discrete sequence with a high-order fit
    x = np.linspace(-1.0, 1.0, nsamp)
    w = _dpss(nsamp, 0.25 * float(r))
    w = 0.5 * (w + w[::-1])  # exact evenness
    # quintic spline through the samples (as the reference's construction), then
    # Chebyshev interpolation at deg+1 Chebyshev points (square, well conditioned)
    from scipy.interpolate import InterpolatedUnivariateSpline
    spl = InterpolatedUnivariateSpline(x, w, k=5)
    xc = np.cos(np.pi * (np.arange(deg + 1) + 0.5) / (deg + 1))
    bump_c = C.chebfit(xc, spl(xc), deg)
    bump_c[1::2] = 0.0       # even function
    step_c = C.chebint(bump_c, lbnd=-1.0)
    total = C.chebval(1.0, step_c)
    step_c = step_c / total   # the bump keeps the window's own normalisation (peak ~ 1)
    _cache[key] = (bump_c, step_c)
    return bump_c, step_c


class SlepianMollifier(object):
    """step(x): 0 for x <= -1, 1 for x >= 1, smooth in between; bump = d(step)/dx."""

    def __init__(self, r):
        self.r = r
        self.bump_c, self.step_c = _build(r)

    def bump(self, x, check_bounds=True):
        x = np.asarray(x, dtype=float)
        out = C.chebval(np.clip(x, -1.0, 1.0), self.bump_c)
        if check_bounds:
            out = np.where((x > -1.0) & (x < 1.0), out, 0.0)
        return out

    def step(self, x, check_bounds=True):
        x = np.asarray(x, dtype=float)
        out = C.chebval(np.clip(x, -1.0, 1.0), self.step_c)
        if check_bounds:
            out = np.where(x <= -1.0, 0.0, np.where(x >= 1.0, 1.0, out))
        return out
