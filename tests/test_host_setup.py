"""Host-side set-up code of the product (pure numpy parts of ipde_amd/annular)
against the reference goldens.  CPU only; no device calls."""
import os

import numpy as np

from ipde_amd.annular.annular import ApproximateAnnularGeometry as AAG_nyq_dropped
from ipde_amd.annular.annular import RealAnnularGeometry
from ipde_amd.annular.annular_full import ApproximateAnnularGeometry as AAG_full
from ipde_amd.annular.modified_helmholtz import scalar_inverse_blocks
from ipde_amd.annular.stokes import stokes_inverse_blocks

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def close(a, b, tol):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= tol * max(1.0, np.max(np.abs(b)))


def test_chebyshev_and_geometry_match_reference():
    g = np.load(os.path.join(G, "annular_scalar.npz"))
    n, M, width, radius = g["params"]
    aag = AAG_full(int(n), int(M), width, radius)
    assert aag.ns == aag.n
    for name in ["D00", "D01", "D12", "R01", "R12", "R02", "P10", "ibc_dirichlet",
                 "obc_dirichlet", "ibc_neumann", "obc_neumann", "VI1"]:
        assert close(getattr(aag.CO, name), g["CO_" + name], 1e-13), name
    assert close(aag.rv0, g["rv0"], 1e-15) and close(aag.rv1, g["rv1"], 1e-15)
    rag = RealAnnularGeometry(g["speed"], g["curvature"], aag)
    for name in ["psi0", "psi1", "psi2", "inv_psi0", "inv_psi1", "inv_psi2", "DR_psi2",
                 "ipsi_DR_ipsi_DT_psi2", "ipsi_DT_ipsi_DR_psi2"]:
        assert close(getattr(rag, name), g["RAG_" + name], 1e-13), name


def test_scalar_blocks_match_reference():
    g = np.load(os.path.join(G, "annular_scalar.npz"))
    n, M, width, radius = g["params"]
    aag = AAG_full(int(n), int(M), width, radius)
    for tag in ("mh", "po"):
        kinv = scalar_inverse_blocks(aag, float(g[tag + "_k"][0]), aag.CO.ibc_dirichlet[0],
                                     aag.CO.obc_dirichlet[0])
        assert close(kinv, g[tag + "_kinv"].real, 1e-11)


def test_stokes_blocks_match_reference():
    g = np.load(os.path.join(G, "annular_stokes.npz"))
    n, M, width, radius = g["params"]
    aag = AAG_nyq_dropped(int(n), int(M), width, radius)
    assert aag.ns == aag.n - 1 and np.array_equal(aag.ks, g["ks"])
    assert close(stokes_inverse_blocks(aag, 1.0), g["kinv"], 1e-10)


def test_embedded_boundary_warns_when_the_annulus_folds():
    import warnings
    from ipde_amd.embedded_boundary import EmbeddedBoundary
    from ipde_amd.heavisides import SlepianMollifier
    from ipde_amd.pybie2d_compat import star, Global_Smooth_Boundary as GSB
    for nb, expect in ((400, True), (800, False)):
        b = GSB(c=star(nb, a=0.2, f=5))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            e = EmbeddedBoundary(b, True, 16, b.dt * b.speed.min(), pad_zone=0,
                                 heaviside=SlepianMollifier(24).step)
        assert (len(w) > 0) == expect
        assert (e.min_radial_jacobian < 0.25) == expect


def test_prewarm_wait_raises_instead_of_carrying_on(monkeypatch):
    """the join of the warm-up thread must not fall through on a timeout (it exists so that
    rocFFT plan creation is never entered from two threads)"""
    import queue
    import pytest
    from ipde_amd import device
    from ipde_amd._lib import IpdeHipError
    q = queue.Queue()
    q.put(lambda: None)                 # a job nobody will ever finish
    monkeypatch.setitem(device._warm, "queue", q)
    monkeypatch.setitem(device._warm, "thread", None)
    monkeypatch.setattr(device, "PREWARM_TIMEOUT_S", 0.2)
    with pytest.raises(IpdeHipError, match="warm-up"):
        device.prewarm_wait()
    q.get()
    q.task_done()
    device.prewarm_wait()               # idle queue: returns at once
