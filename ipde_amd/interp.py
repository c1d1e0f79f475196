"""Spectral evaluation at scattered points ("type-2 NUFFT" by dense matrix products).

Own replacement for the finufft / near_finder calls that sit on either side of the
hot path (reference multi_boundary/scalar.py:80-88 `periodic_interp2d`;
embedded_boundary.py:419-434 `interpolate_radial_to_points`).  The number of points
is O(boundary nodes) or O(grid points in the annulus), so the exact evaluation
     f(x_p, y_p) = sum_{kx,ky} F[kx,ky] e^{i (kx x_p + ky y_p)}
is two dense complex128 matrix products — library GEMMs (rocBLAS through
torch.matmul, the MFMA f64 path) — instead of a spreading NUFFT: exact to rounding,
no tolerance parameter.  torch is used for the GEMM and device memory only.
"""
import numpy as np
import torch

from .device import get_context
from .spectral import fft1


def _dev(a, ctx, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(device=ctx.torch_device(), dtype=dtype)
    return torch.as_tensor(np.ascontiguousarray(a), device=ctx.torch_device()).to(dtype)


def _cis(theta):
    """exp(i theta) for a real tensor (torch.exp on a complex tensor goes through a kernel
    compiled at first use, ~0.2 s; cos / sin are prebuilt)"""
    return torch.complex(torch.cos(theta), torch.sin(theta))


def periodic_interp2d_gradient(fh, x, y, ikx, iky, ctx=None, chunk=16384):
    """(3, P) real tensor: the real Fourier series fh (Nx, Ny) and its x and y derivatives
    (multipliers ikx (Nx, 1), iky (Ny,) as in ebdy_collection.ikx/iky) at the points
    (x, y) — what reference multi_boundary/scalar.py:80-88 gets from three type-2 NUFFTs.
    Two fields go through the kx GEMM (F and ikx F); the y derivative is taken on its
    product."""
    ctx = ctx or get_context()
    fh = _dev(fh, ctx, torch.complex128)
    ikx = _dev(ikx, ctx, torch.complex128).reshape(-1, 1)
    iky = _dev(iky, ctx, torch.complex128).reshape(-1)
    out = _interp2d_real(torch.stack([fh, ikx * fh]), _dev(x, ctx, torch.float64),
                         _dev(y, ctx, torch.float64), chunk, iky0=iky)
    return out


def periodic_interp2d(fh, x, y, ctx=None, chunk=16384, real_part=False):
    """Evaluate the Fourier series with fft2-ordered coefficients fh (K, Nx, Ny) (or
    (Nx, Ny)) at the points (x, y) given in [0, 2 pi) units of the periodic box.
    Returns a (K, P) (or (P,)) complex torch tensor on the device:
        out[k, p] = (1/(Nx Ny)) sum fh[k, a, b] exp(i (kx_a x_p + ky_b y_p))
    real_part=True returns only its real part (a real tensor) at half the cost: the
    spectrum is Hermitian-symmetrised, (F(k) + conj F(-k))/2 — which leaves the real part
    of the sum unchanged — and only the ky >= 0 half enters the GEMM."""
    ctx = ctx or get_context()
    fh = _dev(fh, ctx, torch.complex128)
    squeeze = fh.dim() == 2
    if squeeze:
        fh = fh[None]
    K, Nx, Ny = fh.shape
    x = _dev(x, ctx, torch.float64)
    y = _dev(y, ctx, torch.float64)
    if real_part:
        out = _interp2d_real(fh, x, y, chunk)
        return out[0] if squeeze else out
    kx = torch.fft.fftfreq(Nx, 1.0 / Nx, dtype=torch.float64, device=fh.device)
    ky = torch.fft.fftfreq(Ny, 1.0 / Ny, dtype=torch.float64, device=fh.device)
    P = x.shape[0]
    out = torch.empty((K, P), dtype=torch.complex128, device=fh.device)
    f2 = fh.reshape(K * Nx, Ny)
    for a in range(0, P, chunk):
        b = min(P, a + chunk)
        Ey = _cis(ky[:, None] * y[None, a:b])          # (Ny, p)
        A = (f2 @ Ey).reshape(K, Nx, b - a)                       # GEMM
        Ex = _cis(kx[:, None] * x[None, a:b])          # (Nx, p)
        out[:, a:b] = (A * Ex[None]).sum(dim=1)
    out /= float(Nx * Ny)
    return out[0] if squeeze else out


def _interp2d_real(fh, x, y, chunk, iky0=None):
    """Re sum_k F(k) e^{i k.x} with the fft-ordered wavenumbers (the Nyquist index carries
    -N/2), through the ky >= 0 half of the spectrum.  Modes pair up with their negatives,
    Re(F(k) e^{ik.x} + F(-k) e^{-ik.x}) = Re((F(k) + conj F(-k)) e^{ik.x}), except on the
    Nyquist row / column, where index negation does not conjugate the phase: the column
    ky = -Ny/2 enters unsymmetrised with weight 1, the row kx = -Nx/2 is summed separately
    over all ky (a rank-one correction).
    iky0 (Ny,) complex: append one more output row, the field 0 multiplied by iky0 (its y
    derivative) — the multiplier commutes with the kx GEMM, so it costs no GEMM of its own."""
    K, Nx, Ny = fh.shape
    dev = fh.device
    fm = torch.roll(torch.flip(fh, dims=(1, 2)), shifts=(1, 1), dims=(1, 2))   # F(-k)
    nyh = Ny // 2 + 1
    fe = fh[:, :, :nyh] + fm[:, :, :nyh].conj()          # 2 F_eff : weight 2 of 0 < ky < Ny/2
    fe[:, :, 0] *= 0.5
    if Ny % 2 == 0:
        fe[:, :, nyh - 1] = fh[:, :, nyh - 1]
    if Nx % 2 == 0:
        fe[:, Nx // 2, :] = 0.0
    f2 = fe.permute(1, 0, 2).reshape(Nx, K * nyh)                 # (Nx, K nyh)
    kx = torch.fft.fftfreq(Nx, 1.0 / Nx, dtype=torch.float64, device=dev)
    kyf = torch.fft.fftfreq(Ny, 1.0 / Ny, dtype=torch.float64, device=dev)
    ky = kyf[:nyh]
    P = x.shape[0]
    Ko = K + (iky0 is not None)
    out = torch.empty((Ko, P), dtype=torch.float64, device=dev)
    nyq = fh[:, Nx // 2, :]
    if iky0 is not None:
        nyq = torch.cat([nyq, (iky0 * fh[0, Nx // 2, :])[None]], dim=0)
    for a in range(0, P, chunk):
        b = min(P, a + chunk)
        Ex = _cis(x[a:b, None] * kx[None, :])          # (p, Nx)
        A = (Ex @ f2).reshape(b - a, K, nyh)                      # GEMM over kx, half of ky
        Ey = _cis(y[a:b, None] * ky[None, :])          # (p, nyh)
        val = (A * Ey[:, None, :]).sum(dim=2)                     # (p, K)
        if iky0 is not None:
            vdy = (A[:, 0, :] * (Ey * iky0[None, :nyh])).sum(dim=1)
            val = torch.cat([val, vdy[:, None]], dim=1)           # (p, Ko)
        if Nx % 2 == 0:
            Eyf = _cis(y[a:b, None] * kyf[None, :])    # (p, Ny)
            row = Eyf @ nyq.transpose(0, 1)                       # (p, Ko)
            val = val + row * _cis(-0.5 * Nx * x[a:b])[:, None]
        out[:, a:b] = val.real.transpose(0, 1)
    out /= float(Nx * Ny)
    return out


_UP_T = 16     # oversampling of the periodic direction before local interpolation
_NL_T = 16     # Lagrange stencil width
_bary = {}


def _bary_host():
    """barycentric weights of _NL_T equispaced nodes (scaled to max 1), host array"""
    w = _bary.get('host')
    if w is None:
        j = np.arange(_NL_T)
        w = np.array([1.0 / np.prod([float(a - b) for b in j if b != a]) for a in j])
        w = _bary['host'] = np.ascontiguousarray(w / np.abs(w).max())
    return w


def radial_to_grid(frs, xi, t, idx=None, outs=None, ctx=None):
    """Several fields given on (M Chebyshev-Gauss nodes, lowest first) x (N equispaced t) of
    ONE boundary, evaluated at the same scattered (xi in [-1, 1], t): one library call
    (ipde_radial_to_grid, csrc/geometry.hip), no host round trip.

    frs: list of (M, N) real arrays — numpy (staged by the library) or device tensors;
    xi, t: device tensors (P,); idx: optional int64 device tensor (P,) of positions and
    outs: list of 1-D device tensors — then outs[f][idx] = values (the radial -> grid
    write of ipde/solvers/multi_boundary/scalar.py:112-116) and outs is returned;
    otherwise a list of new (P,) tensors."""
    import ctypes
    from . import _lib
    from .device import ptr
    ctx = ctx or get_context()
    dev = ctx.torch_device()
    nf = len(frs)
    on_dev = all(isinstance(f, torch.Tensor) for f in frs)
    if on_dev:
        M, N = frs[0].shape
        stack = (frs[0] if nf == 1 else torch.stack(list(frs))).to(device=dev, dtype=torch.float64).contiguous()
        loc = _lib.IPDE_DEVICE
    else:
        host = [f.cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f, dtype=np.float64) for f in frs]
        M, N = host[0].shape
        stack = np.ascontiguousarray(host[0] if nf == 1 else np.stack(host))
        loc = _lib.IPDE_HOST
    xi = _dev(xi, ctx, torch.float64).contiguous()
    t = _dev(t, ctx, torch.float64).contiguous()
    P = t.shape[0]
    if outs is None:
        assert idx is None
        outs = [torch.empty(P, dtype=torch.float64, device=dev) for _ in range(nf)]
    else:
        assert len(outs) == nf and all(o.is_contiguous() and o.dtype == torch.float64 for o in outs)
    if idx is not None:
        assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.shape[0] == P
    op = (ctypes.c_void_p * nf)(*[o.data_ptr() for o in outs])
    ctx.check(ctx.lib.ipde_radial_to_grid(ctx.handle, loc, nf, M, N, ptr(stack), ptr(_bary_host()), P,
                                          ptr(xi), ptr(t), ptr(idx), op))
    return outs


def chebyshev_fourier_eval(fr, xi, t, ctx=None):
    """Evaluate a function given on (M Chebyshev-Gauss nodes, lowest first) x
    (N equispaced t) at scattered (xi in [-1, 1], t).  fr: (M, N) real.
    Returns a real (P,) torch tensor on the device.

    Along t the M coefficient rows are oversampled 16x (phase-shifted inverse FFTs, one
    batched 1-D transform) and read with 16-point barycentric Lagrange interpolation:
    worst-case error (a mode at the original Nyquist) ~1e-17, cost O(M P 16) instead of
    the O(M N P) dense Fourier sum (13.8 ms -> 1 ms per 2048^2 Poisson solve)."""
    return radial_to_grid([fr], xi, t, ctx=ctx)[0]
