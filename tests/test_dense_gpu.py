"""ipde_dense_lu_solve (csrc/dense.hip): blocked substitution with LU factors, against
host LAPACK — sizes that are not multiples of the block, and a QFS collocation matrix
(cond ~1e12) where backward stability is the point."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 128, 129, 300, 1000])
def test_lu_solve_matches_lapack(n):
    import torch
    from ipde_amd.qfs import _DeviceLU
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) + n * np.eye(n) * 0.1
    b = rng.standard_normal(n)
    Ad = torch.as_tensor(A, device="cuda")
    f = _DeviceLU(*torch.linalg.lu_factor(Ad))
    x = f._subst(torch.as_tensor(b, device="cuda")).cpu().numpy()
    ref = scipy.linalg.solve(A, b)
    assert np.abs(x - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    assert np.abs(A @ x - b).max() < 1e-12 * n * max(1.0, np.abs(x).max())
    # the default is one launch per triangular pass with in-launch hand-off of the solutions;
    # a launch per block pair, and a launch per 64-row block: the same arithmetic, bit for bit
    assert f.ctx.get_option("dense_persistent") == 1
    try:
        f.ctx.set_option("dense_persistent", 0)
        x2 = f._subst(torch.as_tensor(b, device="cuda")).cpu().numpy()
        f.ctx.set_option("dense_pairs", 0)
        x1 = f._subst(torch.as_tensor(b, device="cuda")).cpu().numpy()
    finally:
        f.ctx.set_option("dense_pairs", 1)
        f.ctx.set_option("dense_persistent", 1)
    assert np.array_equal(x, x2) and np.array_equal(x, x1)


def test_persistent_substitution_large_batches_under_load_bitwise():
    """the one-launch-per-pass substitution (in-launch hand-off through sentinel slots) on the
    sizes of the solvers — 4096 (Poisson QFS), 6400 (Stokes, 2 x 3200) — in batches of mixed
    sizes, repeated while another stream keeps the chip busy (hand-offs must not depend on
    placement or timing): bit for bit the launch-per-step results, every time"""
    import time
    import torch
    from ipde_amd.qfs import _DeviceLU
    rng = np.random.default_rng(17)
    sizes = [4096, 6400, 1600, 4096, 130, 6400]
    As = [torch.as_tensor(rng.standard_normal((n, n)) + 0.1 * n * np.eye(n), device="cuda") for n in sizes]
    bs = [torch.as_tensor(rng.standard_normal(n), device="cuda") for n in sizes]
    fs = [_DeviceLU(*torch.linalg.lu_factor(A)) for A in As]
    ctx = fs[0].ctx
    ctx.set_option("dense_persistent", 0)
    try:
        ref = _DeviceLU._subst_batch(fs, bs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _DeviceLU._subst_batch(fs, bs)
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - t0) / 5
    finally:
        ctx.set_option("dense_persistent", 1)
    got = _DeviceLU._subst_batch(fs, bs)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _DeviceLU._subst_batch(fs, bs)
    torch.cuda.synchronize()
    t_pers = (time.perf_counter() - t0) / 5
    print("batch of 6 (n up to 6400): %.3f ms per launch-per-step solve, %.3f ms persistent" % (t_step * 1e3, t_pers * 1e3))
    # under load: a second stream streams through HBM and occupies CUs meanwhile
    side = torch.cuda.Stream()
    big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4):
                big.mul_(1.0000001)
        got = _DeviceLU._subst_batch(fs, bs)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    torch.cuda.synchronize()
    # the residual is LAPACK's
    for A, b, x in zip(As, bs, got):
        assert float((A @ x - b).abs().max()) < 1e-11 * A.shape[0]


def test_lu_solve_is_backward_stable_on_qfs_matrix():
    import torch
    from ipde_amd.pybie2d_compat import star, Global_Smooth_Boundary as GSB, Laplace_Layer_Form
    from ipde_amd.qfs import QFS_Boundary, _DeviceLU
    N = 1500
    b = GSB(c=star(N, a=0.2, f=5))
    qb = QFS_Boundary(b, eps=1e-14)
    A = Laplace_Layer_Form(qb.interior_source_bdy, b, ifcharge=True)
    u = np.exp(np.cos(b.t)) + 0.3 * np.sin(3 * b.t)
    Ad = torch.as_tensor(A, device="cuda")
    f = _DeviceLU(*torch.linalg.lu_factor(Ad))
    ud = torch.as_tensor(u, device="cuda")
    x0 = f._subst(ud)
    r0 = float((Ad @ x0 - ud).abs().max())
    x1 = f.solve(Ad, ud)
    r1 = float((Ad @ x1 - ud).abs().max())
    xs = scipy.linalg.solve(A, u)
    rs = np.abs(A @ xs - u).max()
    print(r0, r1, rs)
    assert r0 < 50 * max(rs, 1e-15) and r1 < 1e-13


@pytest.mark.parametrize("M,N", [(12, 64), (20, 257), (16, 1000)])
def test_chebyshev_fourier_eval_is_exact_for_full_band_data(M, N):
    """radial -> scattered points (reference embedded_boundary.py:419-434): every Fourier
    mode up to Nyquist and every Chebyshev degree present; compared with the direct sums"""
    from ipde_amd.interp import chebyshev_fourier_eval
    rng = np.random.default_rng(M * N)
    xc = np.polynomial.chebyshev.chebgauss(M)[0][::-1]
    tt = 2 * np.pi * np.arange(N) / N
    kmax = N // 2
    ck = rng.standard_normal((M, kmax + 1))
    sk = rng.standard_normal((M, kmax + 1))
    if N % 2 == 0:
        sk[:, kmax] = 0.0          # sin(N/2 t) vanishes on the grid
    k = np.arange(kmax + 1)

    def rows(t):                    # (M, len(t)) values of the M coefficient rows
        return ck @ np.cos(np.outer(k, t)) + sk @ np.sin(np.outer(k, t))
    T = np.polynomial.chebyshev.chebvander(xc, M - 1)         # (M nodes, M degrees)
    fr = T @ rows(tt)                                          # values on the radial grid
    P = 3000
    xi, t = rng.uniform(-1, 1, P), rng.uniform(0, 2 * np.pi, P)
    t[:5] = tt[:5]                                             # exact grid hits
    got = chebyshev_fourier_eval(fr, xi, t).cpu().numpy()
    ref = np.einsum('pm,mp->p', np.polynomial.chebyshev.chebvander(xi, M - 1), rows(t))
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("on_device", [False, True])
def test_radial_to_grid_three_fields_scattered_into_grids(on_device):
    """ipde_radial_to_grid with several fields sharing their targets and the write through an
    index list (the end of a Stokes solve, reference multi_boundary/stokes.py:104-110): equals
    the one-field evaluations, touches nothing outside idx, host and device inputs agree"""
    import torch
    from ipde_amd.interp import radial_to_grid, chebyshev_fourier_eval
    rng = np.random.default_rng(11)
    M, N, P, G = 14, 800, 5000, 20000
    frs = [rng.standard_normal((M, N)) for _ in range(3)]
    xi = torch.as_tensor(rng.uniform(-1, 1, P), device="cuda")
    t = torch.as_tensor(rng.uniform(-7, 7, P), device="cuda")     # any real t: periodic
    idx = torch.as_tensor(rng.permutation(G)[:P].astype(np.int64), device="cuda")
    outs = [torch.full((G,), 7.0, dtype=torch.float64, device="cuda") for _ in range(3)]
    inp = [torch.as_tensor(f, device="cuda") for f in frs] if on_device else frs
    res = radial_to_grid(inp, xi, t, idx=idx, outs=outs)
    assert res is outs
    mask = torch.ones(G, dtype=torch.bool, device="cuda")
    mask[idx] = False
    for f, o in zip(frs, outs):
        one = chebyshev_fourier_eval(f, xi, t)
        assert torch.equal(o[idx], one)
        assert bool((o[mask] == 7.0).all())
    # against the direct sums
    k = np.fft.fftfreq(N, 1.0 / N)
    xc = np.polynomial.chebyshev.chebgauss(M)[0][::-1]
    c = np.linalg.solve(np.polynomial.chebyshev.chebvander(xc, M - 1), frs[0])
    ch = np.fft.fft(c, axis=1) / N
    if N % 2 == 0:
        ch[:, N // 2] = ch[:, N // 2].real      # cos(N/2 t) convention of the real interpolant
    tt = t.cpu().numpy()[:200]
    rows = (ch[:, :, None] * np.exp(1j * k[None, :, None] * tt[None, None, :])).sum(axis=1).real
    ref = np.einsum('pm,mp->p', np.polynomial.chebyshev.chebvander(xi.cpu().numpy()[:200], M - 1), rows)
    got = outs[0][idx].cpu().numpy()[:200]
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()


def test_radial_to_grid_argument_checks():
    import ctypes
    import torch
    from ipde_amd import _lib
    from ipde_amd.device import get_context, ptr
    ctx = get_context()
    z = torch.zeros(64, dtype=torch.float64, device="cuda")
    w = np.ones(16)
    op = (ctypes.c_void_p * 1)(z.data_ptr())
    fr = np.zeros((4, 32))
    call = lambda nf, M, N: ctx.lib.ipde_radial_to_grid(ctx.handle, _lib.IPDE_HOST, nf, M, N, ptr(fr), ptr(w), 8,
                                                         ptr(z), ptr(z), None, op)
    assert call(1, 4, 32) == _lib.IPDE_OK
    assert call(0, 4, 32) == 1
    assert call(9, 4, 32) == 1
    assert call(1, 0, 32) == 1
    assert call(1, 4, 8) == 1          # shorter than the stencil
    assert ctx.lib.ipde_radial_to_grid(ctx.handle, 5, 1, 4, 32, ptr(fr), ptr(w), 8, ptr(z), ptr(z), None,
                                       op) == 1
    assert ctx.lib.ipde_radial_to_grid(None, _lib.IPDE_HOST, 1, 4, 32, ptr(fr), ptr(w), 8, ptr(z), ptr(z), None,
                                       op) == 1
    ctx.sync()


@pytest.mark.parametrize("Nx,Ny", [(16, 16), (24, 17), (15, 32)])
def test_periodic_interp2d_real_part_path(Nx, Ny):
    """half-spectrum evaluation == real part of the full complex sum, also for spectra
    that are not Hermitian (ik * fhat with the Nyquist row kept, as the solver passes)"""
    import torch
    from ipde_amd.interp import periodic_interp2d
    rng = np.random.default_rng(Nx * Ny)
    f = rng.standard_normal((Nx, Ny))
    fh = np.fft.fft2(f)
    kx = np.fft.fftfreq(Nx, 1.0 / Nx)[:, None]
    stack = np.stack([fh, 1j * kx * fh, rng.standard_normal((Nx, Ny)) + 1j * rng.standard_normal((Nx, Ny))])
    x, y = rng.uniform(0, 2 * np.pi, 40), rng.uniform(0, 2 * np.pi, 40)
    full = periodic_interp2d(stack, x, y).cpu().numpy()
    half = periodic_interp2d(stack, x, y, real_part=True).cpu().numpy()
    assert half.shape == full.shape and half.dtype == np.float64
    assert np.abs(half - full.real).max() < 1e-12 * np.abs(full).max()
    kxv, kyv = np.fft.fftfreq(Nx, 1.0 / Nx), np.fft.fftfreq(Ny, 1.0 / Ny)
    ref = np.array([(stack[0] * np.exp(1j * (kxv[:, None] * a + kyv[None, :] * b))).sum() / (Nx * Ny)
                    for a, b in zip(x, y)])
    assert np.abs(full[0] - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("Nx,Ny", [(16, 16), (24, 17), (15, 32)])
def test_periodic_interp2d_gradient_equals_three_field_stack(Nx, Ny):
    """value + gradient through two GEMM fields == the three separately interpolated
    fields [F, ikx F, iky F] (multipliers in physical units, Nyquist entries kept)"""
    from ipde_amd.interp import periodic_interp2d, periodic_interp2d_gradient
    rng = np.random.default_rng(7 * Nx + Ny)
    fh = np.fft.fft2(rng.standard_normal((Nx, Ny)))
    ikx = 1j * np.fft.fftfreq(Nx, 0.37 / (2 * np.pi))[:, None]
    iky = 1j * np.fft.fftfreq(Ny, 0.21 / (2 * np.pi))
    x, y = rng.uniform(0, 2 * np.pi, 33), rng.uniform(0, 2 * np.pi, 33)
    ref = periodic_interp2d(np.stack([fh, ikx * fh, iky * fh]), x, y).cpu().numpy().real
    got = periodic_interp2d_gradient(fh, x, y, ikx, iky).cpu().numpy()
    assert got.shape == (3, 33)
    for a, b in zip(got, ref):
        assert np.abs(a - b).max() < 1e-12 * max(np.abs(b).max(), 1.0)


def test_device_form_builders_match_host_forms():
    """ipde_amd.dense_forms (torch, GPU) against the numpy forms of pybie2d_compat"""
    import torch
    from ipde_amd import dense_forms as df
    from ipde_amd import pybie2d_compat as pc
    dev = torch.device("cuda")
    b = pc.Global_Smooth_Boundary(c=pc.star(300, a=0.2, f=5))
    s = pc.Global_Smooth_Boundary(c=b.c * 1.1)

    def same(a, ref):
        assert np.abs(a.cpu().numpy() - ref).max() < 1e-13 * np.abs(ref).max()
    same(df.laplace_form(s, b, dev, ifcharge=True, ifdipole=True), pc.Laplace_Layer_Form(s, b, ifcharge=True, ifdipole=True))
    same(df.laplace_singular_form(b, dev, ifcharge=True), pc.Laplace_Layer_Singular_Form(b, ifcharge=True))
    same(df.laplace_singular_form(b, dev, ifdipole=True), pc.Laplace_Layer_Singular_Form(b, ifdipole=True))
    same(df.stokes_form(s, b, dev, ifforce=True, ifdipole=True), pc.Stokes_Layer_Form(s, b, ifforce=True, ifdipole=True))
    same(df.stokes_singular_form(b, dev, ifforce=True), pc.Stokes_Layer_Singular_Form(b, ifforce=True))
    same(df.stokes_singular_form(b, dev, ifdipole=True), pc.Stokes_Layer_Singular_Form(b, ifdipole=True))
    same(df.stokes_pressure_fix(s, b, dev), pc.Stokes_Pressure_Fix(s, b))
    for k in (1.0, 5.0):            # k = 5: the cut-off band of the Kress split (k r in 2..6) is active
        same(df.modhelm_form(s, b, dev, k, ifcharge=True, ifdipole=True),
             pc.Modified_Helmholtz_Layer_Form(s, b, k=k, ifcharge=True, ifdipole=True))
        same(df.modhelm_singular_form(b, dev, k, ifcharge=True),
             pc.Modified_Helmholtz_Layer_Singular_Form(b, k=k, ifcharge=True))
        same(df.modhelm_singular_form(b, dev, k, ifdipole=True),
             pc.Modified_Helmholtz_Layer_Singular_Form(b, k=k, ifdipole=True))


def test_local_coordinates_kernel_matches_host_iteration():
    """csrc/geometry.hip (closest-point Newton, one thread per point) against the numpy
    iteration of ipde_amd/near.py on a star: same (r, t) to rounding, and p = X(t) + r n(t)."""
    from ipde_amd import near
    from ipde_amd.pybie2d_compat import Global_Smooth_Boundary, star
    bdy = Global_Smooth_Boundary(c=star(400, a=0.2, f=5))
    rng = np.random.default_rng(3)
    width = 12 * bdy.max_h
    j = rng.integers(0, bdy.N, 20000)
    rr = rng.uniform(-1.2 * width, 1.2 * width, j.shape[0])
    px = bdy.x[j] + rr * bdy.normal_x[j] + rng.uniform(-1, 1, j.shape[0]) * bdy.max_h
    py = bdy.y[j] + rr * bdy.normal_y[j] + rng.uniform(-1, 1, j.shape[0]) * bdy.max_h
    # a few points exactly on nodes of the upsampled curve and on the curve itself
    px[:50], py[:50] = bdy.x[:50], bdy.y[:50]
    r_h, t_h, f_h = near.local_coordinates(bdy, px, py, width, on_device=False)
    r_d, t_d, f_d = near.local_coordinates(bdy, px, py, width, on_device=True)
    assert f_h.all() and (f_h == f_d).all()
    dt = np.abs(np.angle(np.exp(1j * (t_h - t_d))))
    assert np.abs(r_h - r_d).max() < 1e-13 and dt.max() < 1e-12
    ev = near.CurveEvaluator(bdy)
    X, Xp, _ = ev(t_d)
    n = -1j * Xp / np.abs(Xp)
    assert np.abs(X + r_d * n - (px + 1j * py)).max() < 1e-12


def test_batched_substitution_equals_separate_solves():
    """ipde_dense_lu_solve_batch: systems in lock-step == each one alone, bit for bit, with
    and without the refinement step"""
    import torch
    from ipde_amd.qfs import _DeviceLU
    rng = np.random.default_rng(5)
    # systems of different sizes: the batch runs the steps of its largest member, the
    # smaller ones drop out of the forward pass early and join the backward pass late
    sizes = [700, 64, 1300, 129, 700, 1, 300, 128, 257, 500]      # > 8: goes in two groups
    As = [torch.as_tensor(rng.standard_normal((n, n)) + 0.1 * n * np.eye(n), device="cuda") for n in sizes]
    bs = [torch.as_tensor(rng.standard_normal(n), device="cuda") for n in sizes]
    fs = [_DeviceLU(*torch.linalg.lu_factor(A)) for A in As]
    for steps in (0, 1):
        single = [f.solve(A, b, steps=steps) for f, A, b in zip(fs, As, bs)]
        batch = _DeviceLU.solve_batch(fs, As, bs, steps=steps)
        for a, b in zip(single, batch):
            assert torch.equal(a, b)


@pytest.mark.parametrize("m,n", [(1, 1), (5, 3), (64, 64), (257, 1023), (1000, 4096), (4096, 4096)])
def test_gemv_matches_numpy(m, n):
    """ipde_dense_gemv (y = A x, y += A x): the QFS boundary limits and refinement residuals"""
    import torch
    from ipde_amd.qfs import _gemv
    rng = np.random.default_rng(m * 7 + n)
    A, x, y0 = rng.standard_normal((m, n)), rng.standard_normal(n), rng.standard_normal(m)
    Ad, xd = torch.as_tensor(A, device="cuda"), torch.as_tensor(x, device="cuda")
    ref = A @ x
    tol = 1e-14 * n * max(1.0, np.abs(A).max() * np.abs(x).max())
    assert np.abs(_gemv(Ad, xd).cpu().numpy() - ref).max() <= tol
    yd = torch.as_tensor(y0, device="cuda")
    out = _gemv(Ad, xd, yd)
    assert out is yd and np.abs(yd.cpu().numpy() - (y0 + ref)).max() <= tol
    # an x that is not 16-byte aligned takes the scalar-load path: same sums up to their order
    xo = torch.empty(n + 1, dtype=torch.float64, device="cuda")[1:]
    xo.copy_(xd)
    assert np.abs(_gemv(Ad, xo).cpu().numpy() - ref).max() <= tol


def test_qfs_call_pair_equals_two_calls():
    from ipde_amd import qfs
    from ipde_amd.pybie2d_compat import Global_Smooth_Boundary, star
    bdy = Global_Smooth_Boundary(c=star(300, a=0.2, f=5))
    rng = np.random.default_rng(9)
    for cls, dens in ((qfs.Laplace_QFS, [rng.standard_normal(300), rng.standard_normal(300)]),
                      (qfs.Stokes_QFS, [rng.standard_normal(600), rng.standard_normal(600)])):
        qa, qb = cls(bdy, True, True, True), cls(bdy, False, True, True)
        a1, b1 = qa(dens), qb(dens)
        a2, b2 = qfs.call_pair(qa, qb, dens)
        assert np.array_equal(a1, a2) and np.array_equal(b1, b2)


def test_setup_kernels_reject_bad_arguments_and_accept_empty_sets():
    """ipde_curve_local_coordinates / ipde_chebfourier_gather: status codes, no work for
    zero points"""
    import torch
    from ipde_amd._lib import IpdeHipError
    from ipde_amd.device import get_context, ptr
    ctx = get_context(0)
    tab = torch.zeros(3 * 64, dtype=torch.complex128, device="cuda")
    w = np.ones(16)
    z = torch.zeros(0, dtype=torch.float64, device="cuda")
    one = torch.zeros(1, dtype=torch.float64, device="cuda")
    # empty point sets are fine
    ctx.check(ctx.lib.ipde_curve_local_coordinates(ctx.handle, 64, ptr(tab), ptr(w), 0, ptr(z), ptr(z), ptr(z),
                                                   0.1, 1e-14, 30, ptr(z), ptr(z)))
    cf = torch.zeros(4 * 64, dtype=torch.float64, device="cuda")
    ctx.check(ctx.lib.ipde_chebfourier_gather(ctx.handle, 4, 64, ptr(cf), ptr(w), 0, ptr(z), ptr(z), ptr(z)))
    # too short a table, non-positive width, missing weights
    for args in ((ctx.handle, 8, ptr(tab), ptr(w), 1, ptr(one), ptr(one), ptr(one), 0.1, 1e-14, 30, ptr(one), ptr(one)),
                 (ctx.handle, 64, ptr(tab), ptr(w), 1, ptr(one), ptr(one), ptr(one), 0.0, 1e-14, 30, ptr(one), ptr(one)),
                 (ctx.handle, 64, ptr(tab), None, 1, ptr(one), ptr(one), ptr(one), 0.1, 1e-14, 30, ptr(one), ptr(one))):
        with pytest.raises(IpdeHipError):
            ctx.check(ctx.lib.ipde_curve_local_coordinates(*args))
    with pytest.raises(IpdeHipError):
        ctx.check(ctx.lib.ipde_chebfourier_gather(ctx.handle, 0, 64, ptr(cf), ptr(w), 1, ptr(one), ptr(one), ptr(one)))
    with pytest.raises(IpdeHipError):
        ctx.check(ctx.lib.ipde_chebfourier_gather(ctx.handle, 4, 64, ptr(cf), ptr(w), 1, None, ptr(one), ptr(one)))


@pytest.mark.parametrize("n", [1200, 1875, 9560])
def test_density_noise_cut_kernel_equals_the_numpy_rule(n):
    """ipde_density_noise_cut (csrc/spectral.hip; Stokes_QFS's filter of its source densities) against the same rule
    in numpy (qfs.Stokes_QFS._noise_cut_host): the cut lands on the same mode and the filtered densities agree; a
    density without a turnaround comes back unchanged; in place; argument checks."""
    import torch
    from test_geometry_cpu import _density_with_noise_turnaround
    from ipde_amd.qfs import Stokes_QFS
    from ipde_amd.device import get_context, ptr
    ctx = get_context()
    rng = np.random.default_rng(n)
    H = n // 2
    t = 2 * np.pi * np.arange(n) / n
    k = np.arange(1, H)
    decaying = ((10.0 ** (-7.0 * k / H))[:, None] * np.cos(k[:, None] * t[None, :])).sum(axis=0)
    cases = [_density_with_noise_turnaround(n, rng, n // 8, 0.5), _density_with_noise_turnaround(n, rng, n // 5, 20.0),
             np.concatenate([decaying, -decaying]), rng.standard_normal(2 * n)]
    for i, mu in enumerate(cases):
        want, kc = Stokes_QFS._noise_cut_host(mu, 30.0, 1e-5)
        d = torch.as_tensor(mu, device="cuda")
        out = torch.empty_like(d)
        kcut = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctx.check(ctx.lib.ipde_density_noise_cut(ctx.handle, n, ptr(d), ptr(out), 30.0, 1e-5, 1.0, ptr(kcut)))
        assert int(kcut.item()) == kc
        assert (kc < H) == (i < 2)
        assert np.abs(out.cpu().numpy() - want).max() < 1e-12 * np.abs(mu).max()
        ctx.check(ctx.lib.ipde_density_noise_cut(ctx.handle, n, ptr(d), ptr(d), 30.0, 1e-5, 1.0, None))     # in place
        assert torch.equal(d, out)
    # a cap below the turnaround wins; the class hook goes through the kernel for device densities
    d = torch.as_tensor(cases[0], device="cuda")
    ctx.check(ctx.lib.ipde_density_noise_cut(ctx.handle, n, ptr(d), ptr(out), 30.0, 1e-5, 0.05, ptr(kcut)))
    assert int(kcut.item()) == int(0.05 * H)
    import types
    hook = Stokes_QFS._lowpass(types.SimpleNamespace(NOISE_CUT=True, RISE=30.0, FLOOR=1e-5,
                                                      _noise_cut_host=Stokes_QFS._noise_cut_host), d)
    assert np.abs(hook.cpu().numpy() - Stokes_QFS._noise_cut_host(cases[0])[0]).max() < 1e-12 * np.abs(cases[0]).max()
    bad = ctx.lib.ipde_density_noise_cut
    assert bad(ctx.handle, 8, ptr(d), ptr(out), 30.0, 1e-5, 1.0, None) == 1
    assert bad(ctx.handle, n, None, ptr(out), 30.0, 1e-5, 1.0, None) == 1
    assert bad(ctx.handle, n, ptr(d), ptr(out), 1.0, 1e-5, 1.0, None) == 1
