"""The library's multi-device context (ipde_multi_*, csrc/multi.hip) on the one
GPU of the box: one device without a communicator, and the same with the RCCL communicator forced
(sources through ncclBroadcast) — results bitwise those of the plain single-context applies, the
target partition the one of ipde_amd/sharding.py.  More than one device: the driver's 8-GPU node."""
import numpy as np
import pytest

from util import Curve, grid_targets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("force_comm", [False, True])
def test_multi_device_context_applies_equal_single_context_ones(force_comm):
    from ipde_amd import layer_potentials as lp
    from ipde_amd.multi import MultiDevice
    from ipde_amd.sharding import target_slice
    c = Curve(300, a=0.2, f=5)
    trg, _ = grid_targets(c, 80)
    rng = np.random.default_rng(5)
    sig, tau = rng.standard_normal(c.N), rng.standard_normal(c.N)
    f, g = rng.standard_normal((2, c.N)), rng.standard_normal((2, c.N))
    w = c.weights
    md = MultiDevice([0], force_comm=force_comm)
    assert md.has_comm == force_comm and md.ndev == 1
    md.set_targets(trg.x, trg.y)
    sl = md.target_slice(0)
    assert (sl.start, sl.stop) == (0, trg.N) == (target_slice(trg.N, 0, 1).start, target_slice(trg.N, 0, 1).stop)
    for rep in range(2):       # (second pass: buffers reused)
        got = md.laplace_apply(c.x, c.y, w_sigma=sig * w, nx=c.normal_x, ny=c.normal_y, w_tau=tau * w)
        ref = lp.laplace_apply(c.x, c.y, trg.x, trg.y, w_sigma=sig * w, nx=c.normal_x, ny=c.normal_y, w_tau=tau * w)
        assert np.array_equal(got, ref)
        got = md.laplace_apply(c.x, c.y, w_sigma=sig * w)
        assert np.array_equal(got, lp.laplace_apply(c.x, c.y, trg.x, trg.y, w_sigma=sig * w))
        got = md.modified_helmholtz_apply(c.x, c.y, 7.0, w_sigma=sig * w, nx=c.normal_x, ny=c.normal_y, w_tau=tau * w)
        ref = lp.modified_helmholtz_apply(c.x, c.y, trg.x, trg.y, 7.0, w_sigma=sig * w, nx=c.normal_x,
                                          ny=c.normal_y, w_tau=tau * w)
        assert np.array_equal(got, ref)
        u, v, p = md.stokes_apply(c.x, c.y, wfx=f[0] * w, wfy=f[1] * w, nx=c.normal_x, ny=c.normal_y,
                                  wdx=g[0] * w, wdy=g[1] * w)
        ur, vr, pr = lp.stokes_apply(c.x, c.y, trg.x, trg.y, wfx=f[0] * w, wfy=f[1] * w, nx=c.normal_x,
                                     ny=c.normal_y, wdx=g[0] * w, wdy=g[1] * w)
        assert np.array_equal(u, ur) and np.array_equal(v, vr) and np.array_equal(p, pr)
        u2, v2 = md.stokes_apply(c.x, c.y, wfx=f[0] * w, wfy=f[1] * w, pressure=False)
        ur, vr = lp.stokes_apply(c.x, c.y, trg.x, trg.y, wfx=f[0] * w, wfy=f[1] * w, pressure=False)[:2]
        assert np.array_equal(u2, ur) and np.array_equal(v2, vr)
    # a larger source set than the first one: the source buffers grow
    c2 = Curve(700, a=0.2, f=5)
    s2 = rng.standard_normal(c2.N)
    assert np.array_equal(md.laplace_apply(c2.x, c2.y, w_sigma=s2 * c2.weights),
                          lp.laplace_apply(c2.x, c2.y, trg.x, trg.y, w_sigma=s2 * c2.weights))
    # the entry points walk the devices with hipSetDevice and put the caller's device back (csrc/multi.hip
    # DeviceGuard; with one device this can only show that nothing moved)
    import torch
    assert torch.cuda.current_device() == 0
    md.close()
    assert torch.cuda.current_device() == 0


def test_multi_device_context_argument_errors():
    from ipde_amd import _lib
    from ipde_amd.multi import MultiDevice
    with pytest.raises(_lib.IpdeHipError):
        MultiDevice([0, 0])            # one context per physical device
    md = MultiDevice([0])
    with pytest.raises(_lib.IpdeHipError):
        md.laplace_apply(np.zeros(4), np.zeros(4), w_sigma=np.ones(4))      # no targets yet
    md.set_targets(np.zeros(0), np.zeros(0))
    assert md.laplace_apply(np.zeros(4), np.arange(4.0), w_sigma=np.ones(4)).shape == (0,)
    md.close()
