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
    return np.concatenate([np.array(arg).ravel() for arg in args])


def affine_transformation(xin, min_in, max_in, min_out, max_out, return_ratio=False,
                          use_numexpr=False):
    rat = (max_out - min_out) / (max_in - min_in)
    xout = (xin - min_in) * rat + min_out
    return (xout, rat) if return_ratio else xout


def get_chebyshev_nodes(lb, ub, order):
    """Chebyshev-Gauss nodes, lowest first, scaled to [lb, ub]
    (ipde/utilities.py:36-49): returns (unscaled, scaled, ratio)."""
    xc, _ = np.polynomial.chebyshev.chebgauss(order)
    x, rat = affine_transformation(xc[::-1], -1, 1, lb, ub, return_ratio=True)
    return xc[::-1], x, rat


def fast_dot(M1, M2):
    """1-D operands stand for diagonal matrices (ipde/utilities.py:51-66)."""
    if len(M1.shape) in [1, 2] and len(M2.shape) == 1:
        return M1 * M2
    elif len(M1.shape) == 1 and len(M2.shape) == 2:
        return M1[:, None] * M2
    elif len(M1.shape) == 2 and len(M2.shape) == 2:
        return M1.dot(M2)
    raise Exception('fast_dot requires shapes to be 1 or 2')


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
# :103-121; imported by ipde/annular/stokes.py).  Like the reference's, they zero that entry of
# their INPUT in place.
def pfourier_multiply(fh, m):
    pos = int(fh.shape[1] // 2 + 1)
    fh[:, pos] = 0.0
    oh = fft(m * ifft(fh))
    oh[:, pos] = 0.0
    return oh


def pfft(f):
    pos = int(f.shape[1] // 2 + 1)
    fh = fft(f)
    fh[:, pos] = 0.0
    return fh


def pifft(fh):
    pos = int(fh.shape[1] // 2 + 1)
    fh[:, pos] = 0.0
    return ifft(fh)


def pifftr(fh):
    return pifft(fh).real


def fast_LU_solve(LU, b):
    """zgetrs without scipy's argument checks (ipde/utilities.py:68-76; complex factors)"""
    from scipy.linalg.lapack import zgetrs
    return zgetrs(LU[0], LU[1], b)[0]


class SimpleFourierFilter(object):
    """Fourier filter of a periodic vector (ipde/utilities.py:126-162); tiny 1-D
    boundary data, host side."""

    def __init__(self, modes, filter_type, **kwargs):
        self.n = modes.shape[0]
        self.modes = modes
        self.filter_type = filter_type
        max_k = np.abs(self.modes).max()
        if filter_type == 'fraction':
            self.filter = np.ones(self.n, dtype=float)
            self.filter[np.abs(self.modes) > max_k * kwargs['fraction']] = 0.0
        elif filter_type == 'rule 36':
            p = kwargs.get('power', 36)
            self.filter = np.exp(-p * (np.abs(self.modes) / max_k) ** p)
        else:
            raise Exception('Filter type not defined.')

    def __call__(self, fin, input_type='space', output_type='space'):
        input_is_real = fin.dtype == float and input_type == 'space'
        if input_type == 'space':
            fin = np.fft.fft(fin)
        fout = fin * self.filter
        if output_type == 'space':
            fout = np.fft.ifft(fout)
            if input_is_real:
                fout = fout.real
        return fout
