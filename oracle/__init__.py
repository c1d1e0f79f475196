"""TEST INFRASTRUCTURE ONLY — the CPU oracle for ipde_amd.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package.  The product (ipde_amd/) never does and fails loudly
when its HIP library is missing instead of falling back to this code.

Contents
  layer_potentials.py   numpy restatement of the six layer-potential kernels
  spectral.py           numpy restatement of the FFT grid solves / derivatives
  annular.py            numpy restatement of the annular operators and solvers
  layer_oracle.c        C/OpenMP restatement of the Laplace/Stokes sums (fast checker,
                        cpu_baseline "port")
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c_oracle(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "layer_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def c_oracle():
    """ctypes handle of liboracle.so (built on demand with gcc)."""
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c_oracle())
        dp = ctypes.POINTER(ctypes.c_double)
        i64 = ctypes.c_int64
        lib.oracle_num_threads.restype = ctypes.c_int
        lib.oracle_laplace_apply.argtypes = [i64, dp, dp, dp, dp, dp, dp, i64, dp, dp, dp,
                                             ctypes.c_int]
        lib.oracle_laplace_apply.restype = None
        lib.oracle_stokes_apply.argtypes = [i64, dp, dp, dp, dp, dp, dp, dp, dp, i64, dp, dp, dp,
                                            dp, dp, ctypes.c_int]
        lib.oracle_stokes_apply.restype = None
        _LIB = lib
    return _LIB


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def c_laplace_apply(sx, sy, tx, ty, w_sigma=None, nx=None, ny=None, w_tau=None,
                    skip_coincident=False):
    """C oracle; densities already weight-multiplied (same as the C ABI)."""
    sx, sy, tx, ty, w_sigma, nx, ny, w_tau = map(_c, (sx, sy, tx, ty, w_sigma, nx, ny, w_tau))
    out = np.empty(tx.shape[0])
    c_oracle().oracle_laplace_apply(sx.shape[0], _p(sx), _p(sy), _p(w_sigma), _p(nx), _p(ny),
                                    _p(w_tau), tx.shape[0], _p(tx), _p(ty), _p(out),
                                    int(skip_coincident))
    return out


def c_stokes_apply(sx, sy, tx, ty, wfx=None, wfy=None, nx=None, ny=None, wdx=None, wdy=None,
                   skip_coincident=False):
    sx, sy, tx, ty, wfx, wfy, nx, ny, wdx, wdy = map(
        _c, (sx, sy, tx, ty, wfx, wfy, nx, ny, wdx, wdy))
    nt = tx.shape[0]
    u, v, p = np.empty(nt), np.empty(nt), np.empty(nt)
    c_oracle().oracle_stokes_apply(sx.shape[0], _p(sx), _p(sy), _p(wfx), _p(wfy), _p(nx), _p(ny),
                                   _p(wdx), _p(wdy), nt, _p(tx), _p(ty), _p(u), _p(v), _p(p),
                                   int(skip_coincident))
    return u, v, p
