"""One host process driving several GPUs through the library's multi-device context
(`ipde_multi_*`, csrc/multi.hip; SURVEY §8(b) "ipde_ctx_create(ndev, dev_ids, &ctx) owns streams,
rocFFT plans, RCCL comm", §8(e)): a fixed target set split over the devices, the sources broadcast
with RCCL over xGMI inside the library, the slices summed side by side.  Host numpy arrays in and
out — the form a C caller of libipde_hip.so uses; the torch.distributed path (one process per GPU,
ipde_amd/sharding.py) is the other way to the same partition."""
import ctypes

import numpy as np

from . import _lib

FORCE_COMM = 1    # IPDE_MULTI_FORCE_COMM


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class MultiDevice(object):
    def __init__(self, dev_ids=(0,), force_comm=False):
        self.lib = _lib.load()
        ids = (ctypes.c_int * len(dev_ids))(*[int(d) for d in dev_ids])
        h = ctypes.c_void_p()
        st = self.lib.ipde_multi_create(len(dev_ids), ids, FORCE_COMM if force_comm else 0, ctypes.byref(h))
        if st != 0:
            raise _lib.IpdeHipError("ipde_multi_create(%r) failed: %s" % (list(dev_ids), _lib._STATUS.get(st, st)))
        self.handle = h
        self.ndev = len(dev_ids)
        self.nt = 0

    def _check(self, st):
        if st != 0:
            raise _lib.IpdeHipError("%s: %s" % (_lib._STATUS.get(st, st),
                                                self.lib.ipde_multi_last_error(self.handle).decode()))

    @property
    def has_comm(self):
        v = ctypes.c_int()
        self._check(self.lib.ipde_multi_has_comm(self.handle, ctypes.byref(v)))
        return bool(v.value)

    def set_targets(self, x, y=None):
        if y is None:
            x, y = x.x, x.y
        x, y = _f64(x).ravel(), _f64(y).ravel()
        self._check(self.lib.ipde_multi_set_targets(self.handle, x.shape[0], _p(x), _p(y)))
        self.nt = int(x.shape[0])

    def target_slice(self, i):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.ipde_multi_target_slice(self.handle, i, ctypes.byref(a), ctypes.byref(b)))
        return slice(a.value, b.value)

    def laplace_apply(self, sx, sy, w_sigma=None, nx=None, ny=None, w_tau=None, flags=0):
        sx, sy, w_sigma, nx, ny, w_tau = (_f64(a) for a in (sx, sy, w_sigma, nx, ny, w_tau))
        out = np.empty(self.nt)
        self._check(self.lib.ipde_multi_laplace_apply(self.handle, sx.shape[0], _p(sx), _p(sy), _p(w_sigma),
                                                      _p(nx), _p(ny), _p(w_tau), _p(out), flags))
        return out

    def modified_helmholtz_apply(self, sx, sy, k, w_sigma=None, nx=None, ny=None, w_tau=None, flags=0):
        sx, sy, w_sigma, nx, ny, w_tau = (_f64(a) for a in (sx, sy, w_sigma, nx, ny, w_tau))
        out = np.empty(self.nt)
        self._check(self.lib.ipde_multi_modhelm_apply(self.handle, float(k), sx.shape[0], _p(sx), _p(sy),
                                                      _p(w_sigma), _p(nx), _p(ny), _p(w_tau), _p(out), flags))
        return out

    def stokes_apply(self, sx, sy, wfx=None, wfy=None, nx=None, ny=None, wdx=None, wdy=None, pressure=True,
                     flags=0):
        sx, sy, wfx, wfy, nx, ny, wdx, wdy = (_f64(a) for a in (sx, sy, wfx, wfy, nx, ny, wdx, wdy))
        u, v = np.empty(self.nt), np.empty(self.nt)
        p = np.empty(self.nt) if pressure else None
        self._check(self.lib.ipde_multi_stokes_apply(self.handle, sx.shape[0], _p(sx), _p(sy), _p(wfx), _p(wfy),
                                                     _p(nx), _p(ny), _p(wdx), _p(wdy), _p(u), _p(v), _p(p), flags))
        return (u, v, p) if pressure else (u, v)

    def close(self):
        if self.handle:
            self.lib.ipde_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
