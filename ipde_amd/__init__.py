"""ipde_amd — MI355X (gfx950) implementation of dbstein/ipde's hot path.

Layer-potential sums, periodic FFT grid solves / derivatives and the annular
Chebyshev x Fourier solves run as hand-written HIP kernels (+ rocFFT) in
libipde_hip.so, called through a C ABI (include/ipde_hip.h) via ctypes.  This
package is the host-side mirror of the reference's Python interface for that path.
There is no CPU fallback: without the library and a gfx950 device the compute
entry points raise.
"""
__version__ = "0.1.0"

import os as _os
import shutil as _shutil


def _install_fft_kernel_cache():
    """rocFFT compiles its kernels at run time on first use of each transform length
    (~4 s for the plans of one 2048^2 solve on a fresh machine: 0.9 s batched 1-D, 1.5 s
    2-D, 1.7 s for torch's own copy of the library) and keeps them in a small sqlite
    cache.  The package ships that cache for gfx950 with the lengths of its tests,
    examples and benchmarks (kernel_cache/rocfft_gfx950.db.gz, built by
    tools/build_kernel_cache.sh on an MI355X) and points rocFFT at a per-user copy, so new
    lengths are added there and the shipped file stays untouched.  Entries of another
    rocFFT version or architecture simply miss and are compiled as before.  A
    ROCFFT_RTC_CACHE_PATH set by the user wins."""
    if "ROCFFT_RTC_CACHE_PATH" in _os.environ or not _os.path.exists("/dev/kfd"):
        return                              # user's choice / no AMD GPU on this machine
    shipped = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "kernel_cache", "rocfft_gfx950.db.gz")
    try:
        base = _os.environ.get("XDG_CACHE_HOME") or _os.path.join(_os.path.expanduser("~"), ".cache")
        dst_dir = _os.path.join(base, "ipde_amd")
        _os.makedirs(dst_dir, exist_ok=True)
        dst = _os.path.join(dst_dir, "rocfft_gfx950.db")
        if _os.path.exists(shipped) and (not _os.path.exists(dst)
                                         or _os.path.getmtime(dst) < _os.path.getmtime(shipped)):
            import gzip
            tmp = dst + ".%d.tmp" % _os.getpid()
            with gzip.open(shipped, "rb") as fi, open(tmp, "wb") as fo:
                _shutil.copyfileobj(fi, fo)
            _os.replace(tmp, dst)          # atomic: ranks starting together never see half a file
        _os.environ["ROCFFT_RTC_CACHE_PATH"] = dst
    except OSError:
        pass                                # no writable cache directory: rocFFT's default behaviour


_install_fft_kernel_cache()


def library_path():
    from ._lib import LIB_PATH
    return LIB_PATH
