"""Host-side geometry / set-up modules (pure numpy): curve differentiation,
closest-point coordinates, inside classification, Slepian cut-offs, Kress
quadrature, QFS.  CPU only."""
import os

import numpy as np

from ipde_amd.heavisides import SlepianMollifier
from ipde_amd.near import local_coordinates, points_inside_curve, grid_inside_curve
from ipde_amd.pybie2d_compat import (star, Global_Smooth_Boundary as GSB, Grid, PointSet,
                                     Laplace_Layer_Form, Laplace_Layer_Singular_Form,
                                     fourier_resample)
from ipde_amd.qfs import Laplace_QFS, QFS_Boundary
from oracle import layer_potentials as olp

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_global_smooth_boundary_star():
    N = 512
    b = GSB(c=star(N, a=0.2, f=5))
    t = b.t
    r, rp, rpp = 1 + 0.2 * np.cos(5 * t), -np.sin(5 * t), -5 * np.cos(5 * t)
    speed = np.sqrt(r * r + rp * rp)
    curv = (r * r + 2 * rp * rp - r * rpp) / speed ** 3
    assert np.max(np.abs(b.speed - speed)) < 1e-11
    assert np.max(np.abs(b.curvature - curv)) < 1e-9
    # outward normal of a ccw curve, unit length, perimeter / area
    assert np.all(b.normal_x * b.x + b.normal_y * b.y > 0)
    assert np.max(np.abs(np.hypot(b.normal_x, b.normal_y) - 1)) < 1e-14
    assert abs(b.area - np.pi * (1 + 0.02)) < 1e-12     # pi (1 + a^2/2)


def test_fourier_resample_roundtrip():
    t = np.linspace(0, 2 * np.pi, 64, endpoint=False)
    f = np.exp(np.sin(t)) + 0.3 * np.cos(5 * t)
    up = fourier_resample(f, 256)
    tu = np.linspace(0, 2 * np.pi, 256, endpoint=False)
    assert np.max(np.abs(up - (np.exp(np.sin(tu)) + 0.3 * np.cos(5 * tu)))) < 1e-13
    assert np.max(np.abs(fourier_resample(up, 64) - f)) < 1e-13


def test_slepian_mollifier_matches_reference_tables():
    """golden values evaluated from the reference's precomputed Chebyshev tables
    (ipde/slepian/heaviside_coefficients.py via chebeval_bump_step.SlepianMollifier)"""
    g = np.load(os.path.join(G, "slepian_values.npz"))
    for r in (30, 40):
        m = SlepianMollifier(r)
        assert np.max(np.abs(m.step(g["x"]) - g["step_%d" % r])) < 1e-11
        assert np.max(np.abs(m.bump(g["x"]) - g["bump_%d" % r])) < 1e-8
    m = SlepianMollifier(30)
    assert m.step(np.array([-1.5, -1.0, 1.0, 2.0])).tolist() == [0.0, 0.0, 1.0, 1.0]
    x = np.linspace(-0.99, 0.99, 50)
    assert np.max(np.abs(m.step(x) + m.step(-x) - 1.0)) < 1e-13   # odd about 1/2


def test_local_coordinates_and_inside():
    N = 2000
    b = GSB(c=star(N, a=0.2, f=5))
    rng = np.random.default_rng(0)
    width = 20 * b.dt * b.speed.min()
    t = rng.uniform(0, 2 * np.pi, 5000)
    r = rng.uniform(-width, 0.3 * width, 5000)
    rr, rp = 1 + 0.2 * np.cos(5 * t), -np.sin(5 * t)
    X = rr * np.exp(1j * t)
    Xp = (rp + 1j * rr) * np.exp(1j * t)
    p = X + r * (-1j * Xp / np.abs(Xp))
    rf, tf, found = local_coordinates(b, p.real, p.imag, width)
    assert found.all()
    assert np.max(np.abs(rf - r)) < 1e-13
    assert np.max(np.abs(np.angle(np.exp(1j * (tf - t))))) < 1e-12
    assert np.array_equal(points_inside_curve(b, p.real, p.imag, rf, found), r < 0)
    # far points are reported as not found
    _, _, f2 = local_coordinates(b, np.array([0.0, 3.0]), np.array([0.0, 3.0]), width)
    assert not f2.any()


def test_grid_inside_scan_fill_equals_polygon_test():
    b = GSB(c=star(600, a=0.2, f=5))
    width = 16 * b.dt * b.speed.min()
    grid = Grid([-1.6037, 1.5963], 200, [-1.6011, 1.5989], 200, x_endpoints=[True, False],
                y_endpoints=[True, False])   # no node exactly on the curve
    IX, IY = np.meshgrid(np.arange(200), np.arange(200), indexing="ij")
    r, t, found = local_coordinates(b, grid.xg.ravel(), grid.yg.ravel(), width)
    mask = grid_inside_curve(grid.shape, IX.ravel()[found], IY.ravel()[found], r[found])
    assert np.array_equal(mask, points_inside_curve(b, grid.xg, grid.yg))


def test_kress_single_layer_and_dlp_limits():
    """On-surface Nystrom matrices against closed forms on the unit circle:
    S[cos m t] = cos(m t)/(2m),  (D - I/2)[1] = -1 (Gauss)."""
    b = GSB(c=star(128, a=0.0, f=1))
    S = Laplace_Layer_Singular_Form(b, ifcharge=True)
    for m in (1, 3, 7):
        assert np.max(np.abs(S @ np.cos(m * b.t) - np.cos(m * b.t) / (2 * m))) < 1e-13
    bs = GSB(c=star(400, a=0.2, f=5))
    D = Laplace_Layer_Singular_Form(bs, ifdipole=True)
    assert np.max(np.abs((D - 0.5 * np.eye(bs.N)) @ np.ones(bs.N) + 1.0)) < 1e-12


def test_qfs_reproduces_layer_potentials_on_both_sides():
    N = 600
    b = GSB(c=star(N, a=0.2, f=5))
    sig = np.exp(np.cos(b.t)) * np.sin(2 * b.t) + 0.3
    tau = np.cos(3 * b.t) + np.sin(b.t) ** 2
    bf = b.generate_resampled_boundary(12 * N)
    sf, tf = fourier_resample(sig, 12 * N), fourier_resample(tau, 12 * N)
    th = np.random.default_rng(0).uniform(0, 2 * np.pi, 200)
    h = b.dt * b.speed.max()
    rr = 1 + 0.2 * np.cos(5 * th)
    X = rr * np.exp(1j * th)
    Xp = (-np.sin(5 * th) + 1j * rr) * np.exp(1j * th)
    n = -1j * Xp / np.abs(Xp)
    q = QFS_Boundary(b, eps=1e-14)
    for interior, p in ((True, X - 2 * h * n), (False, X + 2 * h * n)):
        ref = olp.laplace_layer_apply(bf.x, bf.y, p.real, p.imag, charge=sf, dipstr=tf,
                                      weights=bf.weights, nx=bf.normal_x, ny=bf.normal_y)
        Q = Laplace_QFS(b, interior, True, True, qfs_boundary=q)
        mu = Q([sig, tau])
        got = olp.laplace_layer_apply(Q.source.x, Q.source.y, p.real, p.imag, charge=mu,
                                      weights=Q.source.weights)
        assert np.max(np.abs(got - ref)) < 1e-12 * np.max(np.abs(ref))
        # u2s: source density reproducing given boundary values (the collocation
        # system is ill-conditioned, so check the values it reproduces, not mu itself)
        A = Laplace_Layer_Form(Q.source, b, ifcharge=True)
        ub = A @ mu
        assert np.max(np.abs(A @ Q.u2s(ub) - ub)) < 1e-11 * np.max(np.abs(ub))


def test_arc_length_parameterize_equalises_speed():
    from ipde_amd.pybie2d_compat import arc_length_parameterize
    b = GSB(c=star(400, a=0.3, f=3))
    x, y = arc_length_parameterize(b.x, b.y)
    b2 = GSB(x=x, y=y)
    assert np.ptp(b2.speed) / b2.speed.mean() < 1e-8
    assert abs(b2.weights.sum() - b.weights.sum()) < 1e-12
    th = np.arctan2(y, x)
    assert np.abs(np.hypot(x, y) - (1 + 0.3 * np.cos(3 * th))).max() < 1e-13


def _exterior_stokeslet(c, x0=1.9, y0=1.4, f0=(1.0, -0.5)):
    """velocity and traction on the curve c of a stokeslet placed outside it"""
    import oracle.layer_potentials as ol
    f0 = np.asarray(f0)
    u, v, p = ol.stokes_layer_apply(np.array([x0]), np.array([y0]), c.x, c.y, force=f0.reshape(2, 1))
    dx, dy = c.x - x0, c.y - y0
    r2 = dx * dx + dy * dy
    q = -(dx * f0[0] + dy * f0[1]) * (dx * c.normal_x + dy * c.normal_y) / (np.pi * r2 * r2)
    return (u, v, p), (q * dx, q * dy)


def test_stokes_singular_forms_solve_interior_dirichlet():
    """D - I/2 (+ n n^T completion) and the Kress-split S reproduce an exterior
    stokeslet's field inside the curve (reference examples/multi_stokes.py:134-138)."""
    import oracle.layer_potentials as ol
    from ipde_amd.pybie2d_compat import (Stokes_Layer_Singular_Form, Stokes_Layer_Form,
                                         Stokes_Pressure_Fix, PointSet)
    b = GSB(c=star(300, a=0.2, f=5))
    (ub, vb, _), _ = _exterior_stokeslet(b)
    rhs = np.concatenate([ub, vb])
    xt, yt = np.array([0.1, -0.3, 0.5]), np.array([0.2, 0.4, -0.1])
    P = type('P', (), dict(x=xt, y=yt, normal_x=0 * xt, normal_y=0 * xt))()
    (ue, ve, _), _ = _exterior_stokeslet(P)
    A = Stokes_Layer_Singular_Form(b, ifdipole=True) - 0.5 * np.eye(2 * b.N) + Stokes_Pressure_Fix(b, b)
    tau = np.linalg.solve(A, rhs)
    u, v, _ = ol.stokes_layer_apply(b.x, b.y, xt, yt, dipstr=tau.reshape(2, -1), weights=b.weights,
                                    nx=b.normal_x, ny=b.normal_y)
    assert max(np.abs(u - ue).max(), np.abs(v - ve).max()) < 1e-13
    S = Stokes_Layer_Singular_Form(b, ifforce=True) + Stokes_Pressure_Fix(b, b)
    sig = np.linalg.solve(S, rhs)
    u, v, _ = ol.stokes_layer_apply(b.x, b.y, xt, yt, force=sig.reshape(2, -1), weights=b.weights)
    assert max(np.abs(u - ue).max(), np.abs(v - ve).max()) < 1e-13
    # dense form == kernel sum
    M = Stokes_Layer_Form(b, PointSet(x=xt, y=yt), ifforce=True, ifdipole=True)
    u, v, _ = ol.stokes_layer_apply(b.x, b.y, xt, yt, force=tau.reshape(2, -1), dipstr=tau.reshape(2, -1),
                                    weights=b.weights, nx=b.normal_x, ny=b.normal_y)
    assert np.abs(M @ tau - np.concatenate([u, v])).max() < 1e-13


def test_stokes_qfs_green_representation():
    """Green's representation u = S[t] - D[u] (inside; 0 outside) of an exterior
    stokeslet through Stokes_QFS: velocity and (calibrated) pressure right up to the curve."""
    import oracle.layer_potentials as ol
    from ipde_amd.qfs import Stokes_QFS
    b = GSB(c=star(400, a=0.2, f=5))
    (ub, vb, _), (tx, ty) = _exterior_stokeslet(b)
    taus, taud = np.concatenate([tx, ty]), -np.concatenate([ub, vb])
    fine = b.generate_resampled_boundary(3200)
    h = b.dt * b.speed.min()
    for interior in (True, False):
        q = Stokes_QFS(b, interior, True, True)
        mu = q([taus, taud])
        c = fine.c + (-1 if interior else 1) * 0.5 * h * fine.normal_c
        xt, yt = c.real[3::41].copy(), c.imag[3::41].copy()
        got = ol.stokes_layer_apply(q.source.x, q.source.y, xt, yt, force=mu.reshape(2, -1),
                                    weights=q.source.weights)
        if interior:
            P = type('P', (), dict(x=xt, y=yt, normal_x=0 * xt, normal_y=0 * xt))()
            (ue, ve, pe), _ = _exterior_stokeslet(P)
        else:
            ue = ve = pe = 0 * xt
        assert max(np.abs(got[0] - ue).max(), np.abs(got[1] - ve).max()) < 1e-11
        assert np.abs(got[2] - pe).max() < 1e-8


def _density_with_noise_turnaround(n, rng, k_floor, top_amp):
    """a smooth two-component density whose spectrum decays to 1e-9 by k_floor and rises again (amplified noise)"""
    t = 2 * np.pi * np.arange(n) / n
    H = n // 2
    k = np.arange(1, H)
    amp = np.where(k <= k_floor, 10.0 ** (-9.0 * k / k_floor),
                   1e-9 * (top_amp / 1e-9) ** ((k - k_floor) / (0.85 * H - k_floor)).clip(max=1.0))
    comp = []
    for _ in range(2):
        ph = rng.uniform(0, 2 * np.pi, k.size)
        comp.append(3.0 + (amp[:, None] * np.cos(k[:, None] * t[None, :] + ph[:, None])).sum(axis=0))
    return np.concatenate(comp)


def test_stokes_qfs_noise_cut_rule():
    """Stokes_QFS._noise_cut_host (the numpy statement of ipde_density_noise_cut): a spectrum that decays to a
    deep minimum and rises again is cut at the minimum, both components alike; a spectrum still decaying at the
    Nyquist frequency, a flat one, and an isolated high mode over an empty spectrum are left alone."""
    from ipde_amd.qfs import Stokes_QFS
    rng = np.random.default_rng(5)
    n = 1200
    H = n // 2
    mu = _density_with_noise_turnaround(n, rng, 150, 0.5)      # (9560: see tests/test_dense_gpu.py)
    got, kc = Stokes_QFS._noise_cut_host(mu)
    w = max(4, (H + 1 + 511) // 512)
    assert 150 - 2 * w <= kc <= 150 + 2 * w
    for comp, gcomp in ((mu[:n], got[:n]), (mu[n:], got[n:])):
        h = np.fft.fft(comp)
        h[np.minimum(np.arange(n), n - np.arange(n)) > kc] = 0.0
        assert np.abs(np.fft.ifft(h).real - gcomp).max() < 1e-13
    assert np.abs(got - mu).max() > 0.1                                   # (the rise was really there)
    t = 2 * np.pi * np.arange(n) / n
    k = np.arange(1, H)
    decaying = np.concatenate([(10.0 ** (-7.0 * k / H))[:, None] * np.cos(k[:, None] * t[None, :])]).sum(axis=0)
    for same in (np.concatenate([decaying, 2 * decaying]),                # under-resolved: still decaying at Nyquist
                 rng.standard_normal(2 * n),                               # flat spectrum
                 np.concatenate([np.cos(400 * t), np.sin(400 * t)])):      # one high mode, nothing below it
        out, kc = Stokes_QFS._noise_cut_host(same)
        assert kc == H and np.abs(out - same).max() < 1e-13 * max(1.0, np.abs(same).max())


def test_kress_single_layer_for_odd_and_even_node_counts():
    """the log-singular quadrature has no Nyquist term for odd N"""
    for N in (300, 301):
        b = GSB(c=star(N, a=0.2, f=5))
        xs, ys, one = np.array([2.0]), np.array([1.5]), np.array([1.0])
        ub = olp.laplace_layer_apply(xs, ys, b.x, b.y, charge=one)
        sig = np.linalg.solve(Laplace_Layer_Singular_Form(b, ifcharge=True), ub)
        xt, yt = np.array([0.1, -0.3]), np.array([0.2, 0.4])
        u = olp.laplace_layer_apply(b.x, b.y, xt, yt, charge=sig, weights=b.weights)
        assert np.abs(u - olp.laplace_layer_apply(xs, ys, xt, yt, charge=one)).max() < 1e-13
