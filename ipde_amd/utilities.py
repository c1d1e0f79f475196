"""Drop-in for the parts of ipde/utilities.py the hot path uses (reference :5-124).

fft / ifft / fft2 / ifft2 are backed by rocFFT through the C ABI (the reference
aliases mkl_fft or numpy here, :5-12); the Nyquist-dropping variants are built on
them exactly as the reference builds them on its aliases (:78-101).  Small host
helpers (chebyshev nodes, affine map, concat) are set-up code and stay numpy.
"""
import numpy as np

from .spectral import fft1, get_plan


def fft(a):
    """np.fft.fft along the last axis (1-D or 2-D input)."""
    a = np.asarray(a)
    if a.ndim == 1:
        return fft1(a[None, :], -1)[0]
    return fft1(a, -1)


def ifft(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return fft1(a[None, :], +1)[0]
    return fft1(a, +1)


def fft2(a):
    nx, ny = a.shape
    return get_plan(nx, ny, 1.0, 1.0).fft2(a)


def ifft2(a):
    nx, ny = a.shape
    return get_plan(nx, ny, 1.0, 1.0).ifft2(a)


def concat(*args):
    """All arguments (scalars, lists, arrays) flattened into one vector (ipde/utilities.py:14-15)."""
    return np.hstack([np.ravel(np.asarray(arg)) for arg in args])


def affine_transformation(xin, min_in, max_in, min_out, max_out, return_ratio=False,
                          use_numexpr=False):
    rat = (max_out - min_out) / (max_in - min_in)
    xout = (xin - min_in) * rat + min_out
    return (xout, rat) if return_ratio else xout


def get_chebyshev_nodes(lb, ub, order):
    """Chebyshev-Gauss nodes in ascending order, on [-1, 1] and mapped to [lb, ub]
    (ipde/utilities.py:36-49): returns (unscaled, scaled, d scaled / d unscaled)."""
    ascending = np.polynomial.chebyshev.chebgauss(order)[0][::-1]
    scaled, rat = affine_transformation(ascending, -1, 1, lb, ub, return_ratio=True)
    return ascending, scaled, rat


# (ndim of M1, ndim of M2) -> product with 1-D operands read as diagonal matrices
_FAST_DOT = {
    (1, 1): lambda a, b: a * b,
    (2, 1): lambda a, b: a * b,            # columns scaled: A diag(b)
    (1, 2): lambda a, b: a[:, None] * b,   # rows scaled: diag(a) B
    (2, 2): lambda a, b: a.dot(b),
}


def fast_dot(M1, M2):
    """Matrix product in which a 1-D operand stands for a diagonal matrix (ipde/utilities.py:51-66)."""
    rule = _FAST_DOT.get((M1.ndim, M2.ndim))
    if rule is None:
        raise Exception('fast_dot requires shapes to be 1 or 2')
    return rule(M1, M2)


# Nyquist-dropping transforms (ipde/utilities.py:78-101)
def mfft(f):
    N = f.shape[1]
    N2 = int(N / 2)
    fh = fft(f)
    return np.concatenate([fh[:, :N2], fh[:, N2 + 1:]], axis=1)


def mifft(fh):
    M, NS = fh.shape
    N = NS + 1
    N2 = int(N / 2)
    temp = np.zeros((M, N), dtype=complex)
    temp[:, :N2] = fh[:, :N2]
    temp[:, N2 + 1:] = fh[:, N2:]
    return ifft(temp)


def mifftr(fh):
    return mifft(fh).real


def fourier_multiply(fh, m):
    return mfft(m * mifft(fh))


def ffourier_multiply(fh, m):
    return fft(m * ifft(fh))


# Transforms that zero entry N//2 + 1 instead of dropping the Nyquist column (ipde/utilities.py
# :103-121; imported by ipde/annular/stokes.py).  Like the reference's, the ones that take a
# spectrum zero that entry of their INPUT in place.
def _blank(fh):
    fh[:, fh.shape[1] // 2 + 1] = 0.0
    return fh


def pfft(f):
    return _blank(fft(f))


def pifft(fh):
    return ifft(_blank(fh))


def pifftr(fh):
    return pifft(fh).real


def pfourier_multiply(fh, m):
    return pfft(m * pifft(fh))


def fast_LU_solve(LU, b):
    """zgetrs without scipy's argument checks (ipde/utilities.py:68-76; complex factors)"""
    from scipy.linalg.lapack import zgetrs
    return zgetrs(LU[0], LU[1], b)[0]


class SimpleFourierFilter(object):
    """Fourier filter of a periodic vector (ipde/utilities.py:126-162): `modes` are the wavenumbers
    in np.fft order; 'fraction' keeps |k| <= fraction max|k|, 'rule 36' multiplies by
    exp(-p (|k| / max|k|)^p), p = `power` (36).  Tiny 1-D boundary data, host side."""

    def __init__(self, modes, filter_type, **kwargs):
        self.modes = modes
        self.n = modes.shape[0]
        self.filter_type = filter_type
        top = np.abs(modes).max()
        rel = np.abs(modes) / top
        if filter_type == 'fraction':
            self.filter = np.where(np.abs(modes) > top * kwargs['fraction'], 0.0, 1.0)
        elif filter_type == 'rule 36':
            power = kwargs.get('power', 36)
            self.filter = np.exp(-power * rel ** power)
        else:
            raise Exception('Filter type not defined.')

    def __call__(self, fin, input_type='space', output_type='space'):
        from_space, to_space = input_type == 'space', output_type == 'space'
        spectrum = (np.fft.fft(fin) if from_space else fin) * self.filter
        if not to_space:
            return spectrum
        out = np.fft.ifft(spectrum)
        # real samples in, real samples out
        return out.real if (from_space and fin.dtype == float) else out
