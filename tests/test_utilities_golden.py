"""ipde_amd.utilities against outputs of the reference's own ipde/utilities.py
(tests/golden/utilities.npz, made by tests/golden/make_golden.py::golden_utilities).  The host
helpers on CPU; the FFT-backed functions through the library on the GPU box."""
import os

import numpy as np
import pytest
import scipy.linalg

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "utilities.npz"))


def close(a, b, tol=1e-13):
    return a.shape == b.shape and np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))


def test_host_helpers_against_reference_outputs():
    from ipde_amd import utilities as U
    m, r = G["m"], G["r"]
    assert np.array_equal(U.fast_dot(m, m[0]), G["fast_dot_mv"])
    assert np.array_equal(U.fast_dot(m[:, 0], m), G["fast_dot_vm"])
    assert close(U.fast_dot(m[:, :6], m), G["fast_dot_mm"], 1e-15)
    assert np.array_equal(U.concat(r[0], 3.0, [1.0, 2.0]), G["concat"])
    assert close(U.affine_transformation(r[0], -2.0, 3.0, 0.0, 2 * np.pi), G["affine"], 1e-15)
    xc, x, rat = U.get_chebyshev_nodes(-0.3, 0.1, 12)
    assert close(xc, G["cheb_unscaled"], 1e-15) and close(x, G["cheb_scaled"], 1e-15)
    assert abs(rat - G["cheb_ratio"][0]) < 1e-15
    got = U.fast_LU_solve(scipy.linalg.lu_factor(G["lu_A"]), G["lu_b"])
    assert close(got, G["fast_LU_solve"], 1e-13)


@pytest.mark.gpu
def test_fft_backed_functions_against_reference_outputs():
    from ipde_amd import utilities as U
    a, r, g, m = G["a"], G["r"], G["g"], G["m"]
    assert close(U.fft(a), G["fft"]) and close(U.ifft(a), G["ifft"])
    assert close(U.fft2(g), G["fft2"]) and close(U.ifft2(g), G["ifft2"])
    assert close(U.fft2(g.real.copy()), G["fft2_real"])
    mf = U.mfft(r)
    assert close(mf, G["mfft"])
    assert close(U.mifft(mf), G["mifft"]) and close(U.mifftr(mf), G["mifftr"])
    assert close(U.fourier_multiply(U.mfft(a), m), G["fourier_multiply"])
    assert close(U.ffourier_multiply(a.copy(), m), G["ffourier_multiply"])
    assert close(U.pfourier_multiply(a.copy(), m), G["pfourier_multiply"])
    assert close(U.pfft(r), G["pfft"])
    assert close(U.pifft(a.copy()), G["pifft"]) and close(U.pifftr(a.copy()), G["pifftr"])
    b = a.copy()
    U.pifft(b)
    assert np.all(b[:, b.shape[1] // 2 + 1] == 0)       # like the reference: input zeroed in place


def test_simple_fourier_filter_against_reference_outputs():
    """SimpleFourierFilter (ipde/utilities.py:126-162) against the reference class's own outputs
    (tests/golden/fourier_filter.npz)."""
    from ipde_amd.utilities import SimpleFourierFilter
    F = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fourier_filter.npz"))
    modes, fr, fc = F["modes"], F["fr"], F["fc"]
    f = SimpleFourierFilter(modes, 'fraction', fraction=2.0 / 3.0)
    assert np.array_equal(f.filter, F["fraction_filter"])
    got = f(fr)
    assert got.dtype == F["fraction_real"].dtype and close(got, F["fraction_real"], 1e-15)
    assert close(f(fc), F["fraction_cplx"], 1e-15)
    assert close(f(np.fft.fft(fr), input_type='fourier'), F["fraction_spec_in"], 1e-15)
    assert close(f(fr, output_type='fourier'), F["fraction_spec_out"], 1e-15)
    r = SimpleFourierFilter(modes, 'rule 36')
    assert close(r.filter, F["rule36_filter"], 1e-15) and close(r(fr), F["rule36_real"], 1e-15)
    r8 = SimpleFourierFilter(modes, 'rule 36', power=8)
    assert close(r8.filter, F["rule8_filter"], 1e-15) and close(r8(fc), F["rule8_cplx"], 1e-15)
    with pytest.raises(Exception):
        SimpleFourierFilter(modes, 'no such filter')
