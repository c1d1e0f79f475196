"""Parity of the device spectral operators (rocFFT + symbol kernels, stencils)
against the reference goldens and the numpy oracle.  fp64 tolerance 1e-12 of
max|out| (BASELINE north_star)."""
import os

import numpy as np
import pytest

from oracle import spectral as osp
from util import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-12


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_derivatives_against_reference_golden(tag):
    from ipde_amd import derivatives as d
    from ipde_amd.spectral import GridPlan
    g = np.load(os.path.join(G, "derivatives.npz"))
    f, (hx, hy) = g[tag + "_f"], g[tag + "_h"]
    nx, ny = f.shape
    assert rel_err(d.fd_x_4(f, hx), g[tag + "_fdx"]) < 1e-14
    assert rel_err(d.fd_y_4(f, hy), g[tag + "_fdy"]) < 1e-14
    assert rel_err(d.fd_x_4(f, hx, periodic_fix=True), g[tag + "_fdx_p"]) < 1e-14
    assert rel_err(d.fd_y_4(f, hy, periodic_fix=True), g[tag + "_fdy_p"]) < 1e-14
    kx, ky = osp.wavenumbers(nx, ny, hx, hy)
    assert rel_err(d.fourier(f, 1j * kx), g[tag + "_dx"]) < TOL
    assert rel_err(d.fourier(f, 1j * ky), g[tag + "_dy"]) < TOL
    assert rel_err(d.fourier(f, g[tag + "_sym"]), g[tag + "_gen"]) < TOL
    plan = GridPlan(nx, ny, hx, hy)
    assert rel_err(plan.dx(f), g[tag + "_dx"]) < TOL
    assert rel_err(plan.dy(f), g[tag + "_dy"]) < TOL
    plan.close()


@pytest.mark.parametrize("shape", [(64, 64), (48, 40), (45, 37), (128, 96)])
def test_grid_solves_against_oracle(shape):
    from ipde_amd.spectral import GridPlan
    nx, ny = shape
    hx, hy = 3.0 / nx, 2.5 / ny
    rng = np.random.default_rng(nx * 1000 + ny)
    f = rng.standard_normal(shape)
    f -= f.mean()
    g = rng.standard_normal(shape)
    g -= g.mean()
    plan = GridPlan(nx, ny, hx, hy)
    uh_ref, u_ref = osp.poisson_grid_solve(f, hx, hy)
    uh, u = plan.poisson_solve(f, want_uhat=True)
    assert rel_err(u, u_ref) < TOL and rel_err(uh, uh_ref) < TOL
    assert rel_err(plan.poisson_solve(f), u_ref) < TOL
    uh_ref, u_ref = osp.modhelm_grid_solve(f, 10.0, hx, hy)
    uh, u = plan.modhelm_solve(f, 10.0, want_uhat=True)
    assert rel_err(u, u_ref) < TOL and rel_err(uh, uh_ref) < TOL
    ur, vr, pr = osp.stokes_grid_solve(f, g, hx, hy)
    u, v, p = plan.stokes_solve(f, g)
    assert rel_err(u, ur) < TOL and rel_err(v, vr) < TOL and rel_err(p, pr) < TOL
    plan.close()


def test_fft_aliases_match_numpy():
    from ipde_amd import utilities as ut
    rng = np.random.default_rng(3)
    a = rng.standard_normal((12, 64)) + 1j * rng.standard_normal((12, 64))
    assert rel_err(ut.fft(a), np.fft.fft(a)) < TOL
    assert rel_err(ut.ifft(a), np.fft.ifft(a)) < TOL
    assert rel_err(ut.fft(a[0]), np.fft.fft(a[0])) < TOL
    b = rng.standard_normal((40, 36))
    assert rel_err(ut.fft2(b), np.fft.fft2(b)) < TOL
    c = b + 1j * rng.standard_normal((40, 36))
    assert rel_err(ut.fft2(c), np.fft.fft2(c)) < TOL
    assert rel_err(ut.ifft2(c), np.fft.ifft2(c)) < TOL
    r = rng.standard_normal((6, 32))
    assert rel_err(ut.mifft(ut.mfft(r)).real[:, :], np.fft.ifft(_drop_nyq(np.fft.fft(r))).real) < TOL


def _drop_nyq(fh):
    fh = fh.copy()
    fh[:, fh.shape[1] // 2] = 0.0
    return fh


def test_device_resident_grid_solve_and_roundtrip_full_size():
    """2048^2 (BASELINE size): device-resident in/out; Laplacian(u) == f spectrally
    and derivative of a known trigonometric field."""
    import torch
    from ipde_amd.spectral import GridPlan
    n = 2048
    h = 2 * np.pi / n
    plan = GridPlan(n, n, h, h)
    x = torch.arange(n, dtype=torch.float64, device="cuda") * h
    X, Y = torch.meshgrid(x, x, indexing="ij")
    f = torch.exp(torch.sin(X)) * torch.cos(2 * Y)
    f = f - f.mean()
    u = plan.poisson_solve(f)
    assert isinstance(u, torch.Tensor) and u.is_cuda
    lap = plan.dx(plan.dx(u)) + plan.dy(plan.dy(u))
    # two spectral derivatives amplify rounding by k_max^2 = 1024^2
    assert float(torch.max(torch.abs(lap - f)) / torch.max(torch.abs(f))) < 5e-9
    g = torch.sin(3 * X) * torch.cos(2 * Y)
    ug = plan.poisson_solve(g)
    assert float(torch.max(torch.abs(ug + g / 13.0))) < 1e-13
    dfdx = plan.dx(f)
    exact = torch.cos(X) * torch.exp(torch.sin(X)) * torch.cos(2 * Y)
    assert float(torch.max(torch.abs(dfdx - exact))) < 1e-11
    plan.close()


@pytest.mark.parametrize("shape", [(512, 1024), (1024, 2048), (2048, 1024), (2048, 2048), (4096, 4096),
                                   (512, 8192)])
def test_hand_written_fft_pipeline_against_oracle(ctx, shape):
    """Power-of-two grids take the three-kernel pipeline of csrc/fft2d.hip (rows r2c,
    fused column FFT x symbol x inverse FFT, rows c2r): every scalar operator against the
    numpy oracle (host arrays and device tensors), and against the rocFFT path of the same
    library (option fft2d = 0)."""
    import torch
    from ipde_amd.spectral import GridPlan
    assert ctx.get_option("fft2d") == 1
    nx, ny = shape
    hx, hy = 3.0 / nx, 2.5 / ny
    rng = np.random.default_rng(nx * 7 + ny)
    f = rng.standard_normal(shape)
    f -= f.mean()
    kx, ky = osp.wavenumbers(nx, ny, hx, hy)
    plan = GridPlan(nx, ny, hx, hy)
    refs = {
        "poisson": (lambda a: plan.poisson_solve(a), osp.poisson_grid_solve(f, hx, hy)[1]),
        "modhelm": (lambda a: plan.modhelm_solve(a, 7.5), osp.modhelm_grid_solve(f, 7.5, hx, hy)[1]),
        "dx": (lambda a: plan.dx(a), osp.fourier(f, 1j * kx)),
        "dy": (lambda a: plan.dy(a), osp.fourier(f, 1j * ky)),
    }
    fd = torch.as_tensor(f, device="cuda")
    for name, (fn, ref) in refs.items():
        got_h = fn(f)
        got_d = fn(fd)
        assert isinstance(got_h, np.ndarray) and got_d.is_cuda
        assert rel_err(got_h, ref) < TOL, name
        assert np.array_equal(got_d.cpu().numpy(), got_h), name
        ctx.set_option("fft2d", 0)
        try:
            lib = fn(fd).cpu().numpy()
        finally:
            ctx.set_option("fft2d", 1)
        assert rel_err(got_h, lib) < 1e-13, name
    # Stokes grid solve (two forward, packed symbols, three inverse transforms)
    g2 = rng.standard_normal(shape)
    g2 -= g2.mean()
    ref3 = osp.stokes_grid_solve(f, g2, hx, hy)
    got3 = plan.stokes_solve(f, g2)
    got3d = plan.stokes_solve(fd, torch.as_tensor(g2, device="cuda"))
    ctx.set_option("fft2d", 0)
    try:
        lib3 = plan.stokes_solve(f, g2)
    finally:
        ctx.set_option("fft2d", 1)
    for a, b, c, r in zip(got3, got3d, lib3, ref3):
        assert rel_err(a, r) < TOL and np.array_equal(b.cpu().numpy(), a) and rel_err(a, c) < 1e-13
    # the wanted-spectrum variant keeps returning fft2(f) * symbol on these sizes too
    uh_ref, u_ref = osp.poisson_grid_solve(f, hx, hy)
    uh, u = plan.poisson_solve(f, want_uhat=True)
    assert rel_err(u, u_ref) < TOL and rel_err(uh, uh_ref) < TOL
    plan.close()


@pytest.mark.parametrize("shape,npts,shifted", [((512, 1024), 700, 0), ((2048, 2048), 4096, 0),
                                                ((512, 1024), 700, 1), ((2048, 1024), 900, 1),
                                                ((4096, 4096), 600, 0),
                                                # any other size: rocFFT spectrum, fine grid = next power
                                                # of two >= 2 n (oversampling 2 .. 4), odd sizes too
                                                ((48, 40), 300, 0), ((301, 255), 500, 0),
                                                ((600, 900), 800, 0), ((1370, 1370), 2000, 0)])
@pytest.mark.parametrize("band", [1, 0])
def test_grid_interp_against_dense_fourier_sums(ctx, shape, npts, shifted, band):
    """Values and gradient of the grid solution at scattered points: the oversampled-FFT
    interpolation of csrc/nufft.hip (what the solvers use on power-of-two grids) against the
    exact dense Fourier sums of ipde_amd.interp (the checker), <= 1e-13 of each field's
    maximum; timing of both printed.  shifted = 1 forces the variant that grids of 4096 points
    a side use by themselves (four half-cell-shifted coarse transforms instead of one fine one).
    band = 1 (the default): the oversampled transform along x only, exact sums along y per point;
    band = 0: full oversampled fine grids and the 2-D window gather."""
    if band and shifted:
        pytest.skip("the band form has one variant for the packed spectra")
    import time
    import torch
    from ipde_amd.spectral import GridPlan
    from ipde_amd.interp import periodic_interp2d_gradient
    nx, ny = shape
    hx, hy = 3.0 / nx, 2.6 / ny
    x = torch.arange(nx, dtype=torch.float64, device="cuda") * hx
    y = torch.arange(ny, dtype=torch.float64, device="cuda") * hy
    X, Y = torch.meshgrid(x, y, indexing="ij")
    # a smooth periodic forcing plus a little broadband content down to the Nyquist lines
    g = torch.Generator(device="cuda").manual_seed(5)
    f = torch.exp(torch.sin(2 * np.pi * X / 3.0)) * torch.cos(4 * np.pi * Y / 2.6) \
        + 1e-3 * torch.randn(nx, ny, dtype=torch.float64, device="cuda", generator=g)
    f -= f.mean()
    ctx.set_option("interp_shifted", shifted)
    ctx.set_option("interp_band", band)
    plan = GridPlan(nx, ny, hx, hy)
    assert plan.keep_spectrum(True)
    u = plan.poisson_solve(f)
    rng = np.random.default_rng(8)
    px = torch.as_tensor(rng.uniform(0, 2 * np.pi, npts), device="cuda")
    py = torch.as_tensor(rng.uniform(0, 2 * np.pi, npts), device="cuda")
    px[0], py[0], px[1], py[1] = 0.0, 0.0, 2 * np.pi * (1 - 1e-12), 1e-9     # box corners / wrap
    got = plan.interp_gradient(px, py)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = plan.interp_gradient(px, py)
    torch.cuda.synchronize()
    t_fft = time.perf_counter() - t0
    # the checker: same spectrum by the rocFFT path of the library, dense sums
    plan.keep_spectrum(False)
    uh, u2 = plan.poisson_solve(f, want_uhat=True)
    kx = torch.as_tensor(np.fft.fftfreq(nx, hx / (2 * np.pi)), device="cuda")
    ky = torch.as_tensor(np.fft.fftfreq(ny, hy / (2 * np.pi)), device="cuda")
    ref = periodic_interp2d_gradient(uh, px, py, 1j * kx[:, None], 1j * ky)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = periodic_interp2d_gradient(uh, px, py, 1j * kx[:, None], 1j * ky)
    torch.cuda.synchronize()
    t_gemm = time.perf_counter() - t0
    print("grid %s, %d points: %s %.3f ms, dense sums %.3f ms"
          % (shape, npts, "band form" if band else "oversampled FFT", 1e3 * t_fft, 1e3 * t_gemm))
    for k in range(3):
        err = float((got[k] - ref[k]).abs().max() / ref[k].abs().max())
        assert err < 1e-13, (k, err)
    # host arrays in, host arrays out
    plan.keep_spectrum(True)
    plan.poisson_solve(f)
    got_h = plan.interp_gradient(px.cpu().numpy(), py.cpu().numpy())
    assert isinstance(got_h, np.ndarray) and np.array_equal(got_h, got.cpu().numpy())
    # and the value row is the grid solution at grid points
    ii, jj = 37, ny - 5
    at = plan.interp_gradient(np.array([2 * np.pi * ii / nx]), np.array([2 * np.pi * jj / ny]))
    assert abs(at[0, 0] - float(u[ii, jj])) < 1e-13 * float(u.abs().max())
    plan.close()
    ctx.set_option("interp_shifted", 0)


def test_grid_interp_needs_a_kept_spectrum_and_a_supported_grid():
    from ipde_amd._lib import IpdeHipError
    from ipde_amd.spectral import GridPlan
    plan = GridPlan(2500, 64, 0.1, 0.1)
    assert plan.keep_spectrum(True) is False          # the 2x fine grid would need 8192 rows: dense sums
    with pytest.raises(IpdeHipError, match="no kept spectrum"):
        plan.interp_gradient(np.zeros(3), np.zeros(3))
    plan.close()
    plan = GridPlan(48, 40, 0.1, 0.1)                 # any moderate size has the path
    assert plan.keep_spectrum(True) is True
    plan.close()
    plan = GridPlan(512, 1024, 0.1, 0.1)
    assert plan.keep_spectrum(True) is True
    with pytest.raises(IpdeHipError, match="no kept spectrum"):    # no solve yet
        plan.interp_gradient(np.zeros(3), np.zeros(3))
    plan.close()


@pytest.mark.parametrize("shape,npts,shifted", [((512, 1024), 500, 0), ((1024, 1024), 900, 1),
                                                ((2048, 2048), 3000, 0), ((301, 255), 400, 0),
                                                ((1370, 1370), 1500, 0)])
@pytest.mark.parametrize("band", [1, 0])
def test_grid_interp_fields_against_dense_fourier_sums(ctx, shape, npts, shifted, band):
    """ipde_grid_interp_fields: the five interface fields of the Stokes solver (u, v and the
    stress T = grad u + grad u^T - p I of three real grid fields) against the dense Fourier sums
    of ipde_amd.interp on the same multiplied spectra (reference multi_boundary/vector.py:66-82)."""
    if band and shifted:
        pytest.skip("the band form has one variant for the packed spectra")
    import torch
    from ipde_amd.spectral import GridPlan
    from ipde_amd.interp import periodic_interp2d
    from ipde_amd.solvers.multi_boundary.vector import VectorSolver
    nx, ny = shape
    hx, hy = 3.1 / nx, 2.9 / ny
    x = torch.arange(nx, dtype=torch.float64, device="cuda") * hx
    y = torch.arange(ny, dtype=torch.float64, device="cuda") * hy
    X, Y = torch.meshgrid(x, y, indexing="ij")
    g = torch.Generator(device="cuda").manual_seed(9)
    a, b = 2 * np.pi / 3.1, 2 * np.pi / 2.9
    # (a little broadband content down to the Nyquist lines; the window's error scales with the
    # l1 norm of the spectrum, which white noise inflates by sqrt(nx ny) over smooth fields)
    noise = lambda: 1e-6 * torch.randn(nx, ny, dtype=torch.float64, device="cuda", generator=g)
    u = torch.exp(torch.sin(a * X)) * torch.cos(2 * b * Y) + noise()
    v = torch.sin(3 * a * X + 0.2) * torch.exp(torch.cos(b * Y)) + noise()
    p = torch.cos(a * X) * torch.sin(b * Y + 0.7) + noise()
    rng = np.random.default_rng(3)
    px = torch.as_tensor(rng.uniform(0, 2 * np.pi, npts), device="cuda")
    py = torch.as_tensor(rng.uniform(0, 2 * np.pi, npts), device="cuda")
    ctx.set_option("interp_shifted", shifted)
    ctx.set_option("interp_band", band)
    plan = GridPlan(nx, ny, hx, hy)
    got = plan.interp_fields([u, v, p], VectorSolver._STRESS_FIELDS, px, py)
    ikx = torch.as_tensor(1j * np.fft.fftfreq(nx, hx / (2 * np.pi)), device="cuda")[:, None]
    iky = torch.as_tensor(1j * np.fft.fftfreq(ny, hy / (2 * np.pi)), device="cuda")
    uh, vh, ph = plan.fft2(u), plan.fft2(v), plan.fft2(p)
    stack = torch.stack([uh, vh, 2 * ikx * uh - ph, iky * uh + ikx * vh, 2 * iky * vh - ph])
    ref = periodic_interp2d(stack, px, py, real_part=True)
    for k in range(5):
        err = float((got[k] - ref[k]).abs().max() / ref[k].abs().max())
        # values: 1e-13.  Stress rows: the two sides differentiate spectra from two different
        # forward FFTs (fft2d here, rocFFT in the checker), whose rounding floors differ by
        # ~1e-16 max|F| per mode; i k amplifies that by up to k_max ~ n: a few 1e-16 n
        # (measured 2e-13 at 512 x 1024, 8e-13 at 2048^2; the Stokes tolerance is 1e-10)
        assert err < (1e-13 if k < 2 else 1e-15 * max(nx, ny)), (k, err)
    # host arrays
    got_h = plan.interp_fields([a_.cpu().numpy() for a_ in (u, v, p)], VectorSolver._STRESS_FIELDS,
                               px.cpu().numpy(), py.cpu().numpy())
    assert isinstance(got_h, np.ndarray) and np.abs(got_h - got.cpu().numpy()).max() < 1e-12
    plan.close()
    ctx.set_option("interp_shifted", 0)


def test_grid_interp_fields_argument_checks():
    from ipde_amd._lib import IpdeHipError
    from ipde_amd.spectral import GridPlan
    z = np.zeros(4)
    plan = GridPlan(2500, 64, 0.1, 0.1)                # no fine grid within the FFT kernels' sizes
    with pytest.raises(IpdeHipError, match="no fft2d path"):
        plan.interp_fields([np.zeros((2500, 64))], [[(1.0, 0, 0)]], z, z)
    plan.close()
    plan = GridPlan(512, 1024, 0.1, 0.1)
    f = np.zeros((512, 1024))
    with pytest.raises(IpdeHipError, match="invalid argument"):      # a field index out of range
        plan.interp_fields([f], [[(1.0, 1, 0)]], z, z)
    with pytest.raises(IpdeHipError, match="invalid argument"):      # four terms in one output
        plan.interp_fields([f], [[(1.0, 0, 0)] * 4], z, z)
    with pytest.raises(ValueError):                                  # wrong grid shape
        plan.interp_fields([np.zeros((8, 8))], [[(1.0, 0, 0)]], z, z)
    assert plan.interp_fields([f], [[(1.0, 0, 1)]], np.zeros(0), np.zeros(0)).shape == (1, 0)
    plan.close()


def test_grid_list_moves():
    """ipde_grid_scatter / add_at / gather (csrc/geometry.hip): the mask operations of the
    multi-boundary solvers (reference embedded_function.py:105-113,135-138), exact"""
    import torch
    from ipde_amd import gridops
    from ipde_amd.device import get_context, ptr
    rng = np.random.default_rng(2)
    ngrid, n = 300 * 200, 31000
    idx = torch.as_tensor(np.sort(rng.permutation(ngrid)[:n]).astype(np.int64), device="cuda")
    src = torch.as_tensor(rng.standard_normal(n), device="cuda")
    scale = torch.as_tensor(rng.standard_normal((300, 200)), device="cuda")
    ref = torch.zeros(ngrid, dtype=torch.float64, device="cuda")
    ref[idx] = src
    assert torch.equal(gridops.scatter(idx, src, ngrid), ref)
    assert torch.equal(gridops.scatter(idx, src, ngrid, scale=scale), ref * scale.reshape(-1))
    g = torch.as_tensor(rng.standard_normal(ngrid), device="cuda")
    want = g.clone()
    want[idx] += src
    assert gridops.add_at(idx, src, g) is g and torch.equal(g, want)
    assert torch.equal(gridops.gather(idx, g), g[idx])
    ctx = get_context()
    e = torch.empty(0, dtype=torch.float64, device="cuda")
    ei = torch.empty(0, dtype=torch.int64, device="cuda")
    assert ctx.lib.ipde_grid_gather(ctx.handle, 0, ptr(ei), ptr(g), ptr(e)) == 0          # empty lists
    assert ctx.lib.ipde_grid_add_at(ctx.handle, 0, ptr(ei), ptr(e), ptr(g)) == 0
    assert ctx.lib.ipde_grid_scatter(ctx.handle, 8, 0, ptr(ei), ptr(e), None, ptr(g)) == 0
    assert ctx.lib.ipde_grid_scatter(ctx.handle, 4, 8, ptr(idx), ptr(src), None, ptr(g)) == 1   # more entries than grid
    assert ctx.lib.ipde_grid_gather(ctx.handle, 5, None, ptr(g), ptr(e)) == 1
    assert ctx.lib.ipde_grid_add_at(None, 0, ptr(ei), ptr(e), ptr(g)) == 1
    ctx.sync()


def test_host_io_staged_upload_and_pinned_result():
    """ipde_amd/hostio.py: the forcing's chunked, thread-staged upload and the result container
    built over pinned memory"""
    import torch
    from ipde_amd import hostio
    rng = np.random.default_rng(4)
    for n in (10, (1 << 18) - 3, 3 * (1 << 20) + 17):
        src = rng.standard_normal(n)
        pin = torch.empty(n, dtype=torch.float64, pin_memory=True)
        dst = torch.empty(n, dtype=torch.float64, device="cuda")
        hostio.upload(dst, src, pin)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), src)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
    import interior_poisson   # (an ebdyc to build a container on)
    _, _, solver, ue, _ = interior_poisson.run(nb=400, M=12)
    f, block = hostio.pinned_function(solver.ebdyc)
    assert block.is_pinned() and f.shape == np.asarray(ue).shape
    block.copy_(torch.as_tensor(np.asarray(ue), device="cuda"), non_blocking=False)
    assert np.array_equal(np.asarray(f), np.asarray(ue))
    assert np.array_equal(f[0], ue[0])        # the radial block of boundary 0 through the container


@pytest.mark.parametrize("shape,npts", [((2048, 2048), 4096), ((4096, 4096), 8192), ((1024, 8192), 3000)])
def test_grid_interp_band_form_on_points_along_a_curve(ctx, shape, npts):
    """The band form where it is used: points along a closed curve (the interface nodes of a solve — many points
    share fine rows where the curve runs along y, few where it runs along x), at the sizes of BASELINE configs[2]
    and configs[3], against the full-fine-grid form on the same kept spectrum (1e-13) and, on a sample, the dense
    Fourier sums; results of two calls identical bit for bit; timings of both forms printed."""
    import time
    import torch
    from ipde_amd.spectral import GridPlan
    from ipde_amd.interp import periodic_interp2d_gradient
    nx, ny = shape
    hx, hy = 3.0 / nx, 3.0 / ny
    x = torch.arange(nx, dtype=torch.float64, device="cuda") * hx
    y = torch.arange(ny, dtype=torch.float64, device="cuda") * hy
    X, Y = torch.meshgrid(x, y, indexing="ij")
    f = torch.exp(torch.sin(2 * np.pi * X / 3.0)) * torch.cos(4 * np.pi * Y / 3.0) + torch.sin(2 * np.pi * (3 * X + Y) / 3.0)
    f -= f.mean()
    del X, Y
    th = 2 * np.pi * np.arange(npts) / npts
    r = 1.0 + 0.2 * np.cos(5 * th)
    px = torch.as_tensor((1.5 + r * np.cos(th)) * 2 * np.pi / 3.0, device="cuda")
    py = torch.as_tensor((1.5 + r * np.sin(th)) * 2 * np.pi / 3.0, device="cuda")
    out, ms = {}, {}
    for band in (1, 0):
        ctx.set_option("interp_band", band)
        plan = GridPlan(nx, ny, hx, hy)
        assert plan.keep_spectrum(True)
        plan.poisson_solve(f)
        a = plan.interp_gradient(px, py)
        b = plan.interp_gradient(px, py)
        assert torch.equal(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.interp_gradient(px, py)
        torch.cuda.synchronize()
        ms[band] = 1e3 * (time.perf_counter() - t0) / 5
        out[band] = a
        if band == 0:
            plan.keep_spectrum(False)
            uh, _ = plan.poisson_solve(f, want_uhat=True)
        plan.close()
        del plan
        torch.cuda.empty_cache()
    print("grid %s, %d curve points: band form %.3f ms, full fine grids %.3f ms" % (shape, npts, ms[1], ms[0]))
    for k in range(3):
        assert float((out[1][k] - out[0][k]).abs().max()) < 1e-13 * float(out[0][k].abs().max()), k
    idx = torch.as_tensor(np.random.default_rng(2).choice(npts, 400, replace=False), device="cuda")
    kx = torch.as_tensor(np.fft.fftfreq(nx, hx / (2 * np.pi)), device="cuda")
    ky = torch.as_tensor(np.fft.fftfreq(ny, hy / (2 * np.pi)), device="cuda")
    ref = periodic_interp2d_gradient(uh, px[idx], py[idx], 1j * kx[:, None], 1j * ky)
    for k in range(3):
        assert float((out[1][k][idx] - ref[k]).abs().max()) < 1e-13 * float(ref[k].abs().max()), k
