"""ipde_amd/target_plan.py: the split of a target list into 4 x 4 tensor patches and a remainder
(host logic, torch on the CPU here; the kernel it feeds is tested in tests/test_layer_gpu.py)."""
import numpy as np
import pytest
import torch

from ipde_amd import target_plan


def _check_partition(x, y, plan):
    n = x.shape[0]
    pout = plan.pout.numpy()
    pxy = plan.pxy.numpy()
    seen = np.zeros(n, dtype=int)
    np.add.at(seen, pout[pout >= 0], 1)
    np.add.at(seen, plan.rest.numpy(), 1)
    assert np.array_equal(seen, np.ones(n, dtype=int))          # every target exactly once
    for a in range(4):
        for b in range(4):
            idx = pout[4 * a + b]
            m = idx >= 0
            assert np.array_equal(x[idx[m]], pxy[a][m]) and np.array_equal(y[idx[m]], pxy[4 + b][m])
    assert np.array_equal(plan.rest_x.numpy(), x[plan.rest.numpy()])
    assert np.array_equal(plan.rest_y.numpy(), y[plan.rest.numpy()])


def _band_list(ngrid=96, lim=1.5, clearance=4.0):
    v = np.linspace(-lim, lim, ngrid, endpoint=False)
    X, Y = np.meshgrid(v, v, indexing="ij")
    x, y = X.ravel(), Y.ravel()
    r = np.hypot(x, y)
    rb = 1.0 + 0.2 * np.cos(5 * np.arctan2(y, x))
    keep = np.abs(r - rb) > clearance * (v[1] - v[0])
    return x[keep], y[keep]


@pytest.mark.parametrize("block", [(1, 1 << 20), (8, 8), (4, 4)])
def test_grid_with_a_band_removed_splits_into_tiles_and_remainder(block):
    x, y = _band_list(160, clearance=2.5)
    plan = target_plan.build(torch.from_numpy(x), torch.from_numpy(y), block=block)
    _check_partition(x, y, plan)
    assert plan.nrest == 0 and x.shape[0] / 16 < plan.np < 1.25 * x.shape[0] / 16
    assert (plan.pout.numpy() < 0).any()                # tiles the band cut into
    # the same patches whatever the hand-out order
    ref = target_plan.build(torch.from_numpy(x), torch.from_numpy(y), block=(1, 1 << 20))
    key = lambda p: np.sort(p.pout.numpy().max(axis=0))
    assert np.array_equal(key(plan), key(ref))
    # a list too ragged for that keeps its full tiles and a remainder
    old, target_plan.PARTIAL_MIN_FILL = target_plan.PARTIAL_MIN_FILL, 0.999
    try:
        plan = target_plan.build(torch.from_numpy(x), torch.from_numpy(y), block=block)
    finally:
        target_plan.PARTIAL_MIN_FILL = old
    _check_partition(x, y, plan)
    assert plan.np > 0.6 * x.shape[0] / 16 and 0 < plan.nrest < 0.4 * x.shape[0]
    assert (plan.pout.numpy() >= 0).all()


def test_full_grid_has_no_remainder_and_odd_sizes_keep_their_edges():
    v = np.linspace(0.0, 1.0, 32)
    X, Y = np.meshgrid(v, v * 2.0, indexing="ij")
    plan = target_plan.build(torch.from_numpy(X.ravel()), torch.from_numpy(Y.ravel()))
    assert plan.np == 64 and plan.nrest == 0
    _check_partition(X.ravel(), Y.ravel(), plan)
    w = np.linspace(0.0, 1.0, 30)
    X, Y = np.meshgrid(v[:27], w, indexing="ij")
    plan = target_plan.build(torch.from_numpy(X.ravel()), torch.from_numpy(Y.ravel()))
    assert plan.np == 7 * 8 and plan.nrest == 0         # 27 x 30: edge tiles with unstored points
    _check_partition(X.ravel(), Y.ravel(), plan)


def test_lists_that_are_no_grid_and_repeated_points():
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(500), rng.standard_normal(500)
    plan = target_plan.build(torch.from_numpy(x), torch.from_numpy(y))
    assert plan.np == 0 and plan.nrest == 500
    _check_partition(x, y, plan)
    # a grid list in which some points occur twice: one copy sits in a tile, the other in the remainder
    gx, gy = _band_list(48)
    x, y = np.concatenate([gx, gx[100:140]]), np.concatenate([gy, gy[100:140]])
    plan = target_plan.build(torch.from_numpy(x), torch.from_numpy(y))
    _check_partition(x, y, plan)
    assert plan.np > 0
    # too few tiles for the patch kernel to be worth a launch: everything stays in the list
    plan = target_plan.build(torch.from_numpy(gx), torch.from_numpy(gy), min_patches=10 ** 6)
    assert plan.np == 0 and plan.nrest == gx.shape[0]
    # an unordered grid list (any order of a tensor-product set qualifies)
    perm = rng.permutation(gx.shape[0])
    plan = target_plan.build(torch.from_numpy(gx[perm]), torch.from_numpy(gy[perm]))
    _check_partition(gx[perm], gy[perm], plan)
    assert plan.np > 0


def test_grid_list_followed_by_off_lattice_points():
    """grid_pnai = the grid points outside the annuli FOLLOWED BY the interface nodes (reference
    ipde/ebdy_collection.py:426-429): the curve's nodes are on no grid line and stay in the list"""
    gx, gy = _band_list(160, clearance=2.5)
    t = np.linspace(0, 2 * np.pi, 300, endpoint=False)
    cx, cy = 0.8 * np.cos(t) + 1e-3, 0.8 * np.sin(t) - 2e-3
    cx[:3], cy[:3] = gx[[5, 900, 4000]], cy[:3]              # on a grid line in x only
    x, y = np.concatenate([gx, cx]), np.concatenate([gy, cy])
    plan = target_plan.build(torch.from_numpy(x), torch.from_numpy(y))
    _check_partition(x, y, plan)
    assert plan.nrest == 300 and np.array_equal(np.sort(plan.rest.numpy()), gx.shape[0] + np.arange(300))
    ref = target_plan.build(torch.from_numpy(gx), torch.from_numpy(gy))
    assert plan.np == ref.np


def _lists():
    rng = np.random.default_rng(7)
    gx, gy = _band_list(160, clearance=2.5)
    t = np.linspace(0, 2 * np.pi, 300, endpoint=False)
    yield "band", gx, gy
    yield "band + curve nodes", np.concatenate([gx, 0.8 * np.cos(t) + 1e-3]), np.concatenate([gy, 0.8 * np.sin(t)])
    perm = rng.permutation(gx.shape[0])
    yield "unordered", gx[perm], gy[perm]
    v = np.linspace(0.0, 1.0, 27)
    X, Y = np.meshgrid(v, np.linspace(-2.0, 0.0, 30), indexing="ij")
    yield "27 x 30 full", X.ravel(), Y.ravel()
    yield "scattered", rng.standard_normal(500), rng.standard_normal(500)
    yield "coarse ragged", *_band_list(40, clearance=1.0)
    bad = gx.copy()
    bad[[3, 77]] = [np.nan, np.inf]
    yield "non-finite entries", bad, gy
    yield "signed zeros", np.where(gx == 0.0, -0.0, gx), gy
    yield "tiny", gx[:9], gy[:9]


def test_host_builder_of_the_library_equals_the_torch_builder():
    """ipde_target_plan_build (csrc/target_plan.hip, plain C++ behind the C ABI: the builder the
    solvers use) against the torch statement of the same algorithm: identical patches in identical
    order, identical remainder — lists without repeated points (for those the choice of the copy
    that sits on the lattice is free: partition checked instead)."""
    for block in ((8, 8), (1, 1 << 20), (4, 4)):
        for name, x, y in _lists():
            a = target_plan.build_host(x, y, block=block)
            b = target_plan.build(torch.from_numpy(x), torch.from_numpy(y), block=block)
            assert (a.np, a.nrest) == (b.np, b.nrest), name
            assert np.array_equal(a.pxy.numpy(), b.pxy.numpy(), equal_nan=True), name
            assert np.array_equal(a.pout.numpy(), b.pout.numpy()), name
            assert np.array_equal(a.rest.numpy(), b.rest.numpy()), name
            assert np.array_equal(a.rest_x.numpy(), b.rest_x.numpy(), equal_nan=True), name
            if np.isfinite(x).all():
                _check_partition(x, y, a)
    gx, gy = _band_list(48)
    x, y = np.concatenate([gx, gx[100:140]]), np.concatenate([gy, gy[100:140]])
    a = target_plan.build_host(x, y)
    _check_partition(x, y, a)
    assert a.np > 0 and a.nrest >= 40
    assert target_plan.build_host(gx, gy, min_patches=10 ** 6).np == 0
    old, target_plan.PARTIAL_MIN_FILL = target_plan.PARTIAL_MIN_FILL, 2.0
    try:
        a = target_plan.build_host(gx, gy)
        b = target_plan.build(torch.from_numpy(gx), torch.from_numpy(gy))
    finally:
        target_plan.PARTIAL_MIN_FILL = old
    assert a.np == b.np and a.nrest == b.nrest > 0 and np.array_equal(a.pout.numpy(), b.pout.numpy())


def test_host_builder_argument_checks():
    import ctypes
    from ipde_amd import _lib
    lib = _lib.load()
    x = np.zeros(32)
    h = ctypes.c_void_p()
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert lib.ipde_target_plan_build(32, P(x), None, 8, 8, 0.9, 0, 4, ctypes.byref(h)) == _lib.IPDE_ERR_INVALID
    assert lib.ipde_target_plan_build(-1, P(x), P(x), 8, 8, 0.9, 0, 4, ctypes.byref(h)) == _lib.IPDE_ERR_INVALID
    assert lib.ipde_target_plan_build(32, P(x), P(x), 0, 8, 0.9, 0, 4, ctypes.byref(h)) == _lib.IPDE_ERR_INVALID
    assert lib.ipde_target_plan_build(32, P(x), P(x), 8, 8, 0.9, 0, 4, None) == _lib.IPDE_ERR_INVALID
    assert lib.ipde_target_plan_build(0, None, None, 8, 8, 0.9, 0, 4, ctypes.byref(h)) == _lib.IPDE_OK
    n1, n2 = ctypes.c_int64(-1), ctypes.c_int64(-1)
    assert lib.ipde_target_plan_sizes(h, ctypes.byref(n1), ctypes.byref(n2)) == _lib.IPDE_OK and (n1.value, n2.value) == (0, 0)
    assert lib.ipde_target_plan_export(h, None, None, None) == _lib.IPDE_OK
    assert lib.ipde_target_plan_sizes(None, ctypes.byref(n1), ctypes.byref(n2)) == _lib.IPDE_ERR_INVALID
    assert lib.ipde_target_plan_destroy(h) == _lib.IPDE_OK
    assert lib.ipde_target_plan_destroy(None) == _lib.IPDE_OK


def test_padded_blocks_are_whole_waves_of_one_block_of_tiles():
    """build_host(pad_blocks=True) (ipde_target_plan_build_blocks): the same real patches in the same
    order as the unpadded plan, every 8 x 8 block of tiles filled up to 64 patches with copies of its
    first one that store nothing — so that patches [64 k, 64 k + 64) lie in one 32 x 32 window of the
    lattice (what the far-field kernels' per-wave expansions assume)."""
    x, y = _band_list(200, clearance=2.5)
    x = np.concatenate([x, [0.123456, -0.7]])         # two off-lattice points: the remainder
    y = np.concatenate([y, [0.654321, 0.31]])
    plain = target_plan.build_host(x, y)
    pad = target_plan.build_host(x, y, pad_blocks=True)
    _check_partition(x, y, pad)
    assert pad.padded_blocks and not plain.padded_blocks
    assert pad.np % 64 == 0 and pad.np > plain.np and pad.nrest == plain.nrest == 2
    pout, pxy = pad.pout.numpy(), pad.pxy.numpy()
    h = 3.0 / 200
    real = (pout >= 0).any(axis=0)
    assert int(real.sum()) == plain.np
    # the same patches (the blocks of a padded plan come in Z order, so as sets: every patch by its first target)
    key = lambda po: np.argsort(po.max(axis=0), kind="stable")
    ka, kb = key(pout[:, real]), key(plain.pout.numpy())
    assert np.array_equal(pout[:, real][:, ka], plain.pout.numpy()[:, kb])
    assert np.array_equal(pxy[:, real][:, ka], plain.pxy.numpy()[:, kb])
    # Z order: sixteen consecutive blocks lie in one 128 x 128 window of the lattice wherever all sixteen
    # blocks of that window hold patches (true for most of the interior of this list)
    par = pxy.reshape(8, -1, 1024) if pad.np % 1024 == 0 else pxy[:, :pad.np // 1024 * 1024].reshape(8, -1, 1024)
    ext = np.maximum(par[:4].max(axis=(0, 2)) - par[:4].min(axis=(0, 2)), par[4:].max(axis=(0, 2)) - par[4:].min(axis=(0, 2)))
    assert (ext < 127.5 * h).mean() > 0.3
    blocks = pxy.reshape(8, -1, 64)
    assert (blocks[:4].max(axis=(0, 2)) - blocks[:4].min(axis=(0, 2))).max() < 31.5 * h
    assert (blocks[4:].max(axis=(0, 2)) - blocks[4:].min(axis=(0, 2))).max() < 31.5 * h
    # a dummy is a copy of its block's first patch
    first = np.repeat(blocks[:, :, :1], 64, axis=2).reshape(8, -1)
    assert np.array_equal(pxy[:, ~real], first[:, ~real])
    with pytest.raises(ValueError):
        target_plan.build_host(x, y, block=(4, 4), pad_blocks=True)
    # fewer patches than asked for, or no grid: no plan, padded or not
    assert target_plan.build_host(x, y, min_patches=10 ** 6, pad_blocks=True).np == 0
    assert not target_plan.build_host(np.random.default_rng(0).uniform(size=500),
                                      np.random.default_rng(1).uniform(size=500), pad_blocks=True).padded_blocks
