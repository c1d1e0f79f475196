"""Drop-in for ipde/derivatives.py (reference :3-28): same names, argument meaning
and results, computed by HIP kernels / rocFFT."""

from .spectral import fd4, get_plan


def fd_x_4(f, h, periodic_fix=False):
    """4th-order centred x-derivative; rows within 2 of the edge are zero unless
    periodic_fix (ipde/derivatives.py:3-12)."""
    return fd4(f, h, 0, periodic_fix)


def fd_y_4(f, h, periodic_fix=False):
    """(ipde/derivatives.py:14-23)"""
    return fd4(f, h, 1, periodic_fix)


def fourier(f, ik):
    """ifft2(fft2(f) * ik).real (ipde/derivatives.py:25-28).  `ik` is any array that
    broadcasts against f — the solvers pass ebdyc.ikx (Nx,1) or ebdyc.iky (Ny,)."""
    nx, ny = int(f.shape[0]), int(f.shape[1])
    plan = get_plan(nx, ny, 1.0, 1.0)
    return plan.fourier_multiply(f, ik)
