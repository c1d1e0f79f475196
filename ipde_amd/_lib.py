"""ctypes binding of libipde_hip.so (the C ABI declared in include/ipde_hip.h).

The product has NO CPU fallback: if the library is missing or no gfx950 device is
usable, importing/using this module raises.  torch is imported first on purpose:
its bundled libamdhip64 / librocfft (sonames libamdhip64.so.7 / librocfft.so.0)
must be the single HIP runtime of the process so that torch device pointers and
our kernels live in the same runtime.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede loading libipde_hip.so, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (IPDE_HIP_LIBRARY: an alternative build of the same ABI, for A/B measurements)
LIB_PATH = os.environ.get("IPDE_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libipde_hip.so")

IPDE_HOST = 0
IPDE_DEVICE = 1

IPDE_OK = 0
IPDE_ERR_INVALID = 1
IPDE_ERR_NOCONV = 6
FLAG_NONE = 0
FLAG_SKIP_COINCIDENT = 1
FLAG_GENERIC_MATH = 2

_STATUS = {
    1: "invalid argument",
    2: "HIP runtime error",
    3: "rocFFT error",
    4: "no usable gfx950 device",
    5: "device allocation failed",
    6: "GMRES did not converge",
}


class IpdeHipError(RuntimeError):
    pass


_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

# name -> (restype, argtypes); pointer-to-double arguments are declared c_void_p so
# that raw device addresses (ints) and numpy buffers can both be passed.
SIGNATURES = {
    "ipde_version": (ctypes.c_char_p, []),
    "ipde_ctx_create": (_int, [_int, _c_void_pp]),
    "ipde_ctx_destroy": (_int, [_vp]),
    "ipde_ctx_sync": (_int, [_vp]),
    "ipde_ctx_set_stream": (_int, [_vp, _vp]),
    "ipde_ctx_get_stream": (_vp, [_vp]),
    "ipde_ctx_use_legacy_stream": (_int, [_vp]),
    "ipde_ctx_use_background_stream": (_int, [_vp]),
    "ipde_last_error": (ctypes.c_char_p, [_vp]),
    "ipde_ctx_set_option": (_int, [_vp, ctypes.c_char_p, _int]),
    "ipde_ctx_get_option": (_int, [_vp, ctypes.c_char_p, ctypes.POINTER(_int)]),
    "ipde_ctx_enable_timing": (_int, [_vp, _int]),
    "ipde_ctx_last_kernel_ms": (_int, [_vp, _c_double_p]),
    "ipde_ctx_kernel_ms_history": (_int, [_vp, _c_double_p, _int, ctypes.POINTER(_int)]),
    "ipde_multi_create": (_int, [_int, ctypes.POINTER(_int), _int, _c_void_pp]),
    "ipde_multi_destroy": (_int, [_vp]),
    "ipde_multi_ndev": (_int, [_vp, ctypes.POINTER(_int)]),
    "ipde_multi_ctx": (_int, [_vp, _int, _c_void_pp]),
    "ipde_multi_has_comm": (_int, [_vp, ctypes.POINTER(_int)]),
    "ipde_multi_last_error": (ctypes.c_char_p, [_vp]),
    "ipde_multi_set_targets": (_int, [_vp, _i64, _vp, _vp]),
    "ipde_multi_target_slice": (_int, [_vp, _int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "ipde_multi_laplace_apply": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "ipde_multi_modhelm_apply": (_int, [_vp, _dbl, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "ipde_multi_stokes_apply": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "ipde_laplace_apply": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                  _vp, _int]),
    "ipde_laplace_apply_patches": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ipde_laplace_apply_patches_far": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ipde_modhelm_apply_patches_far": (_int, [_vp, _dbl, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ipde_laplace_apply_columns_far": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _vp]),
    "ipde_modhelm_apply_columns_far": (_int, [_vp, _dbl, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _vp]),
    "ipde_stokes_apply_columns_far": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _vp,
                                             _vp, _vp]),
    "ipde_stokes_apply_patches_far": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                             _vp]),
    "ipde_target_plan_build": (_int, [_i64, _vp, _vp, _int, _int, _dbl, _i64, _int, _c_void_pp]),
    "ipde_target_plan_build_blocks": (_int, [_i64, _vp, _vp, _int, _int, _dbl, _i64, _int, _int, _c_void_pp]),
    "ipde_target_plan_sizes": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "ipde_target_plan_export": (_int, [_vp, _vp, _vp, _vp]),
    "ipde_target_plan_destroy": (_int, [_vp]),
    "ipde_modhelm_apply": (_int, [_vp, _int, _dbl, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                                  _vp, _vp, _int]),
    "ipde_stokes_apply": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                 _vp, _vp, _vp, _vp, _vp, _int]),
    "ipde_fft_plan2d_create": (_int, [_vp, _i64, _i64, _dbl, _dbl, _c_void_pp]),
    "ipde_fft_plan2d_destroy": (_int, [_vp]),
    "ipde_fft2_c2c": (_int, [_vp, _int, _int, _vp, _vp]),
    "ipde_fft2_r2c_full": (_int, [_vp, _int, _vp, _vp]),
    "ipde_poisson_grid_solve": (_int, [_vp, _int, _vp, _vp, _vp]),
    "ipde_modhelm_grid_solve": (_int, [_vp, _int, _dbl, _vp, _vp, _vp]),
    "ipde_stokes_grid_solve": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "ipde_fourier_deriv": (_int, [_vp, _int, _vp, _int, _vp]),
    "ipde_fourier_multiply": (_int, [_vp, _int, _vp, _vp, _vp]),
    "ipde_fft_plan2d_keep_spectrum": (_int, [_vp, _int, ctypes.POINTER(_int)]),
    "ipde_grid_interp_prepare": (_int, [_vp]),
    "ipde_grid_interp": (_int, [_vp, _int, _i64, _vp, _vp, _vp]),
    "ipde_grid_interp_fields": (_int, [_vp, _int, _int, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ipde_dense_lu_solve": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "ipde_dense_lu_solve_batch": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "ipde_chebfourier_gather": (_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ipde_dense_lu_factor": (_int, [_vp, _i64, _vp, _vp]),
    "ipde_scalar_interface_jumps": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, ctypes.c_double, _vp, _vp]),
    "ipde_stokes_rotate": (_int, [_vp, _int, _int, _int, _vp, _vp, _vp, _int, _vp, _vp]),
    "ipde_stokes_interface_jumps": (_int, [_vp, _int, _int] + [_vp] * 10 + [ctypes.c_double] + [_vp] * 4),
    "ipde_grid_scatter": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "ipde_grid_add_at": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "ipde_grid_gather": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "ipde_dense_gemv": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _int]),
    "ipde_dense_residual": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "ipde_density_noise_cut": (_int, [_vp, _i64, _vp, _vp, _dbl, _dbl, _dbl, _vp]),
    "ipde_radial_to_grid": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ipde_curve_local_coordinates": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _dbl, _dbl, _int, _vp, _vp]),
    "ipde_grid_inside_scan": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "ipde_trunc_sgf_quadrant": (_int, [_vp, _i64, _i64, _vp, _vp, _dbl, _int, _dbl, _dbl, _dbl, _vp, _i64, _int,
                                       _dbl, _vp]),
    "ipde_ewald_create": (_int, [_vp, _int, _dbl, _dbl, _int, _vp, _int, _int, _c_void_pp]),
    "ipde_ewald_destroy": (_int, [_vp]),
    "ipde_ewald_spread": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _dbl, _dbl, _i64, _i64, _i64, _i64,
                                 _int, _vp, _vp]),
    "ipde_ewald_spread_stokes": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl,
                                        _i64, _i64, _i64, _i64, _vp, _vp]),
    "ipde_fd4": (_int, [_vp, _int, _i64, _i64, _dbl, _int, _int, _vp, _vp]),
    "ipde_fft1_prepare": (_int, [_vp, _i64, _i64]),
    "ipde_fft1_c2c": (_int, [_vp, _int, _i64, _i64, _int, _vp, _vp]),
    "ipde_annular_scalar_create": (_int, [_vp, _int, _int, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _c_void_pp]),
    "ipde_annular_scalar_destroy": (_int, [_vp]),
    "ipde_annular_scalar_set_geometry": (_int, [_vp, _int, _vp, _vp, _vp]),
    "ipde_annular_scalar_apply": (_int, [_vp, _int, _vp, _vp]),
    "ipde_annular_scalar_precondition": (_int, [_vp, _int, _vp, _vp]),
    "ipde_annular_scalar_solve": (_int, [_vp, _int, _vp, _vp, _vp, _int, _dbl, _int, _int, _vp,
                                         ctypes.POINTER(_int), _c_double_p]),
    "ipde_annular_stokes_create": (_int, [_vp, _int, _int, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _c_void_pp]),
    "ipde_annular_stokes_destroy": (_int, [_vp]),
    "ipde_annular_stokes_set_geometry": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ipde_annular_stokes_apply": (_int, [_vp, _int, _vp, _vp]),
    "ipde_annular_stokes_precondition": (_int, [_vp, _int, _vp, _vp]),
    "ipde_annular_stokes_solve": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _int,
                                         _int, _vp, _vp, _vp, ctypes.POINTER(_int), _c_double_p]),
}

_lib = None


def load():
    """Load libipde_hip.so and attach signatures.  Raises IpdeHipError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IpdeHipError(
            "libipde_hip.so not found at %s — build it with `python -m ipde_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = _Library(lib)
    return _lib


def _uses_fft(name):
    """entry points that may create or run rocFFT plans"""
    return name.startswith(("ipde_fft", "ipde_fourier", "ipde_annular", "ipde_grid_interp", "ipde_radial_to_grid",
                            "ipde_stokes_interface")) \
        or name.endswith("_grid_solve")


class _Library(object):
    """The loaded library.  Entry points that can reach rocFFT first join the warm-up thread
    (ipde_amd.device.prewarm_wait): rocFFT plan creation / run-time compilation must not be
    entered from two threads at once."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if _uses_fft(name):
            def guarded(*args, _fn=fn):
                from .device import prewarm_wait
                prewarm_wait()
                return _fn(*args)
            fn = guarded
        self.__dict__[name] = fn
        return fn


def check(status, ctx_handle=None, allow=()):
    if status == IPDE_OK or status in allow:
        return status
    msg = _STATUS.get(status, "status %d" % status)
    if ctx_handle is not None and _lib is not None:
        detail = _lib.ipde_last_error(ctx_handle)
        if detail:
            msg += ": " + detail.decode(errors="replace")
    raise IpdeHipError(msg)
