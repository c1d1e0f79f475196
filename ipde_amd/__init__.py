"""ipde_amd — MI355X (gfx950) implementation of dbstein/ipde's hot path.

Layer-potential sums, periodic FFT grid solves / derivatives and the annular
Chebyshev x Fourier solves run as hand-written HIP kernels (+ rocFFT) in
libipde_hip.so, called through a C ABI (include/ipde_hip.h) via ctypes.  This
package is the host-side mirror of the reference's Python interface for that path.
There is no CPU fallback: without the library and a gfx950 device the compute
entry points raise.
"""
__version__ = "0.1.0"


def library_path():
    from ._lib import LIB_PATH
    return LIB_PATH
