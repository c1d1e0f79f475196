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
    # the wanted-spectrum variant keeps returning fft2(f) * symbol on these sizes too
    uh_ref, u_ref = osp.poisson_grid_solve(f, hx, hy)
    uh, u = plan.poisson_solve(f, want_uhat=True)
    assert rel_err(u, u_ref) < TOL and rel_err(uh, uh_ref) < TOL
    plan.close()
