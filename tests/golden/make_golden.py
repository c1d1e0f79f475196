#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference
(/root/reference) in the build container.  The fixtures are data only (seeded
inputs + the reference's outputs); no reference source is copied.

    python tests/golden/make_golden.py

What can be imported:
  * ipde.derivatives                      — numpy only, imports as is;
  * ipde.utilities, ipde.annular.*        — need three packages the image lacks:
      numexpr            (only `import numexpr as ne`; never called on this path)
      numba              (`njit`/`prange` decorators on the batched matvecs)
      personal_utilities.scipy_gmres.right_gmres   (absent upstream module)
    They are satisfied with in-process stand-ins registered in sys.modules below
    (identity decorators; `range` for prange; a small right-preconditioned GMRES).
    The stand-ins only make the modules importable; every number stored here is
    produced by the reference's own code.  Because the GMRES is ours, only
    OPERATOR-level outputs (matrices, apply, preconditioner) are exact goldens;
    the solve outputs are stored with the manufactured solution they must
    reproduce and are compared at solver tolerance.
  * ipde.solvers.multi_boundary.poisson (Laplace_Eval), ipde.solvers.internals.stokes_save
    (PSLP / PDLP pressure rows, eval_p1) — plain numpy inside modules that also import
    pybie2d, pyfmmlib2d, qfs, flexmm, fmm2dpy, near_finder, function_generator: those get
    EMPTY stand-ins (importable, raise when called) from a meta-path finder; see
    golden_layer_kernels().
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_standins():
    ne = types.ModuleType("numexpr")
    ne.evaluate = lambda expr, **kw: (_ for _ in ()).throw(RuntimeError("numexpr stand-in"))
    sys.modules.setdefault("numexpr", ne)

    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f
    nb.njit = njit
    nb.jit = njit
    nb.prange = range
    nb.float64 = float
    nb.int64 = int
    nb.boolean = bool
    sys.modules.setdefault("numba", nb)

    pu = types.ModuleType("personal_utilities")
    sg = types.ModuleType("personal_utilities.scipy_gmres")

    def right_gmres(A, b, M=None, tol=1e-12, maxiter=200, restart=50, verbose=False, **kw):
        """x = M y with A M y = b (restarted GMRES, modified Gram-Schmidt)."""
        n = b.shape[0]
        x = np.zeros(n, dtype=complex)
        bn = np.linalg.norm(b)
        res = []
        if bn == 0:
            return x, 0, res
        it = 0
        while it < maxiter:
            r = b - A.matvec(x) if it else b.copy()
            beta = np.linalg.norm(r)
            if beta <= tol * bn:
                break
            V = [r / beta]
            Z = []
            H = np.zeros((restart + 1, restart), dtype=complex)
            g = np.zeros(restart + 1, dtype=complex)
            g[0] = beta
            k = 0
            for k in range(restart):
                z = M.matvec(V[k]) if M is not None else V[k]
                z = np.array(z, dtype=complex, copy=True)
                Z.append(z)
                w = np.array(A.matvec(z), dtype=complex, copy=True)
                for i in range(k + 1):
                    H[i, k] = np.vdot(V[i], w)
                    w -= H[i, k] * V[i]
                H[k + 1, k] = np.linalg.norm(w)
                V.append(w / H[k + 1, k])
                y, *_ = np.linalg.lstsq(H[:k + 2, :k + 1], g[:k + 2], rcond=None)
                rr = np.linalg.norm(H[:k + 2, :k + 1] @ y - g[:k + 2])
                res.append(rr / bn)
                it += 1
                if rr <= tol * bn or it >= maxiter:
                    break
            x = x + sum(yi * zi for yi, zi in zip(y, Z))
            if res[-1] <= tol:
                break
        return x, 0, res
    sg.right_gmres = right_gmres
    sg.gmres = right_gmres
    pu.scipy_gmres = sg
    sys.modules.setdefault("personal_utilities", pu)
    sys.modules.setdefault("personal_utilities.scipy_gmres", sg)


def golden_derivatives():
    from ipde.derivatives import fd_x_4, fd_y_4, fourier
    rng = np.random.default_rng(101)
    out = {}
    for tag, (nx, ny) in {"even": (48, 40), "odd": (45, 37)}.items():
        f = rng.standard_normal((nx, ny))
        hx, hy = 2 * np.pi / nx * 1.3, 2 * np.pi / ny * 0.7
        kx = np.fft.fftfreq(nx, hx / (2 * np.pi))[:, None]
        ky = np.fft.fftfreq(ny, hy / (2 * np.pi))
        out[tag + "_f"] = f
        out[tag + "_h"] = np.array([hx, hy])
        out[tag + "_fdx"] = fd_x_4(f, hx)
        out[tag + "_fdy"] = fd_y_4(f, hy)
        out[tag + "_fdx_p"] = fd_x_4(f, hx, periodic_fix=True)
        out[tag + "_fdy_p"] = fd_y_4(f, hy, periodic_fix=True)
        out[tag + "_dx"] = fourier(f, 1j * kx)
        out[tag + "_dy"] = fourier(f, 1j * ky)
        sym = rng.standard_normal((nx, ny)) + 1j * rng.standard_normal((nx, ny))
        out[tag + "_sym"] = sym
        out[tag + "_gen"] = fourier(f, sym)
    np.savez_compressed(os.path.join(OUT, "derivatives.npz"), **out)
    print("derivatives.npz", len(out))


def _annulus_fields(n, seed):
    """speed / curvature of a smooth star-like closed curve (analytic)."""
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    a, f = 0.15, 3
    r = 1 + a * np.cos(f * t)
    rp = -a * f * np.sin(f * t)
    rpp = -a * f * f * np.cos(f * t)
    speed = np.sqrt(r * r + rp * rp)
    curvature = (r * r + 2 * rp * rp - r * rpp) / speed ** 3
    return t, speed, curvature


def golden_annular_scalar():
    from ipde.annular.annular_full import ApproximateAnnularGeometry, RealAnnularGeometry
    from ipde.annular.modified_helmholtz import AnnularModifiedHelmholtzSolver
    from ipde.annular.poisson import AnnularPoissonSolver
    rng = np.random.default_rng(202)
    out = {}
    n, M, width, radius = 64, 10, 0.25, 1.0
    AAG = ApproximateAnnularGeometry(n, M, width, radius)
    CO = AAG.CO
    for name in ["D00", "D01", "D12", "R01", "R12", "R02", "P10", "ibc_dirichlet", "obc_dirichlet",
                 "ibc_neumann", "obc_neumann", "VI1"]:
        out["CO_" + name] = getattr(CO, name)
    out["params"] = np.array([n, M, width, radius])
    out["rv0"], out["rv1"], out["rv2"] = AAG.rv0, AAG.rv1, AAG.rv2
    out["ratio"] = np.array([AAG.ratio])
    t, speed, curvature = _annulus_fields(n, 0)
    RAG = RealAnnularGeometry(speed, curvature, AAG)
    out["speed"], out["curvature"] = speed, curvature
    for name in ["psi0", "psi1", "psi2", "inv_psi0", "inv_psi1", "inv_psi2", "DR_psi2",
                 "ipsi_DR_ipsi_DT_psi2", "ipsi_DT_ipsi_DR_psi2"]:
        out["RAG_" + name] = getattr(RAG, name)
    for tag, k in [("mh", 2.5), ("po", 0.0)]:
        S = AnnularModifiedHelmholtzSolver(AAG, k) if tag == "mh" else AnnularPoissonSolver(AAG)
        S.RAG = RAG
        out[tag + "_k"] = np.array([k])
        out[tag + "_kinv"] = S.Stacked_KINVS
        uh = rng.standard_normal(M * n) + 1j * rng.standard_normal(M * n)
        out[tag + "_uh"] = uh
        out[tag + "_apply"] = np.array(S._apply(uh.copy()))
        out[tag + "_prec"] = np.array(S._optim_preconditioner(uh.copy())).copy()
        # manufactured solve on the real geometry: u = exp(sin x) cos(2y) style field
        rv = AAG.rv0
        # boundary-fitted coordinates: point = X(t) + r * n(t), inward r in [-width, 0]
        rr = 1 + 0.15 * np.cos(3 * t)
        bx, by = rr * np.cos(t), rr * np.sin(t)
        rp = -0.45 * np.sin(3 * t)
        xp = rp * np.cos(t) - rr * np.sin(t)
        yp = rp * np.sin(t) + rr * np.cos(t)
        sp = np.hypot(xp, yp)
        nxv, nyv = yp / sp, -xp / sp
        X = bx[None, :] + rv[:, None] * nxv[None, :]
        Y = by[None, :] + rv[:, None] * nyv[None, :]
        kk = 1.3
        sol = np.exp(np.sin(kk * X)) * np.sin(kk * Y)
        lap = kk ** 2 * np.exp(np.sin(kk * X)) * np.sin(kk * Y) * (np.cos(kk * X) ** 2 - np.sin(kk * X) - 1.0)
        force = (k * k * sol - lap) if tag == "mh" else lap
        Xi = bx - width * nxv
        Yi = by - width * nyv
        ig = np.exp(np.sin(kk * Xi)) * np.sin(kk * Yi)
        og = np.exp(np.sin(kk * bx)) * np.sin(kk * by)
        # geometry convention: ibc acts at r = -width... (the solver's own ibc/obc rows decide)
        ig_row = CO.ibc_dirichlet.dot(sol)[0]
        og_row = CO.obc_dirichlet.dot(sol)[0]
        es = S.solve(RAG, force, ig_row, og_row, tol=1e-13, maxiter=200, restart=60)
        out[tag + "_force"] = force
        out[tag + "_ig"] = ig_row
        out[tag + "_og"] = og_row
        out[tag + "_sol_exact"] = sol
        out[tag + "_sol_ref"] = es
        out[tag + "_iters"] = np.array([S.iterations_last_call])
        print(tag, "ref solve err vs manufactured:", np.abs(es - sol).max(), "iters", S.iterations_last_call)
    np.savez_compressed(os.path.join(OUT, "annular_scalar.npz"), **out)
    print("annular_scalar.npz", len(out))


def golden_annular_stokes():
    from ipde.annular.annular import ApproximateAnnularGeometry, RealAnnularGeometry
    from ipde.annular.stokes import AnnularStokesSolver
    rng = np.random.default_rng(303)
    out = {}
    n, M, width, radius = 48, 8, 0.3, 1.0
    AAG = ApproximateAnnularGeometry(n, M, width, radius)
    t, speed, curvature = _annulus_fields(n, 0)
    RAG = RealAnnularGeometry(speed, curvature, AAG)
    S = AnnularStokesSolver(AAG, mu=1.0)
    out["params"] = np.array([n, M, width, radius])
    out["speed"], out["curvature"] = speed, curvature
    out["ks"] = AAG.ks
    for name in ["psi0", "psi1", "inv_psi1", "inv_psi2", "DR_psi2", "ipsi_DR_ipsi_DT_psi2",
                 "ipsi_DT_ipsi_DR_psi2"]:
        out["RAG_" + name] = getattr(RAG, name)
    out["kinv"] = S.Stacked_KINVS
    # set the state solve() would set, then apply to (a) a random complex vector and
    # (b) a Hermitian-consistent one (the transform of real fields), the case that
    # occurs inside GMRES
    S.RAG = RAG
    S.combo1 = 2 * RAG.DR_psi2 * RAG.inv_psi2 ** 2
    S.combo2 = RAG.DR_psi2 ** 2 * RAG.inv_psi2 ** 2
    S.ik = AAG.k * 1j
    S.iks = AAG.ks * 1j
    NB = S.NB
    v = rng.standard_normal(NB) + 1j * rng.standard_normal(NB)
    out["vec_random"] = v
    out["apply_random"] = np.array(S._apply_optim_real(v.copy()))
    out["prec_random"] = np.array(S._preconditioner(v.copy()))
    from ipde.utilities import mfft
    ur = rng.standard_normal((M, n))
    ut = rng.standard_normal((M, n))
    p = rng.standard_normal((M - 1, n))
    vh = np.concatenate([mfft(ur).ravel(), mfft(ut).ravel(), mfft(p).ravel()])
    out["vec_herm"] = vh
    out["apply_herm"] = np.array(S._apply_optim_real(vh.copy()))
    # a full solve with smooth forcing (tolerance-level golden)
    rv = AAG.rv0
    T = t[None, :] + 0 * rv[:, None]
    fr = np.cos(2 * T) * (1 + rv[:, None])
    ft = np.sin(3 * T) * (1 - 0.5 * rv[:, None])
    irg, itg = 0.1 * np.cos(t), 0.2 * np.sin(2 * t)
    org, otg = -0.3 * np.sin(t), 0.1 * np.cos(3 * t)
    # compatibility: net flux through the two boundaries must vanish for div u = 0;
    # use zero-mean normal data (cos/sin are zero-mean)
    urs, uts, ps = S.solve(RAG, fr, ft, irg, itg, org, otg, tol=1e-12, maxiter=300, restart=100)
    out["fr"], out["ft"] = fr, ft
    out["irg"], out["itg"], out["org"], out["otg"] = irg, itg, org, otg
    out["sol_ur"], out["sol_ut"], out["sol_p"] = urs, uts, ps
    out["P10"] = AAG.CO.P10
    print("stokes golden: |ur| %.3e |ut| %.3e |p| %.3e" % (np.abs(urs).max(), np.abs(uts).max(),
                                                       np.abs(ps).max()))
    np.savez_compressed(os.path.join(OUT, "annular_stokes.npz"), **out)
    print("annular_stokes.npz", len(out))


def golden_slepian():
    """Values of the reference's precomputed step/bump tables (values only, not the
    14 847-line coefficient table)."""
    from ipde.slepian.chebeval_bump_step import SlepianMollifier
    x = np.linspace(-1, 1, 41)
    out = {"x": x}
    for r in (10, 30, 40):
        m = SlepianMollifier(r)
        out["step_%d" % r] = np.array([m.step(np.array([xi])) for xi in x]).ravel()
        out["bump_%d" % r] = np.array([m.bump(np.array([xi])) for xi in x]).ravel()
    np.savez(os.path.join(OUT, "slepian_values.npz"), **out)
    print("slepian_values.npz", len(out))


def golden_grid_evaluator_kernels():
    """gf / fs / ifs / trunc_sgf of the two kernel modules (laplace_grid_evaluator.py:8-33,
    modified_helmholtz_grid_evaluator.py:8-17): pure numpy/scipy functions; the modules
    import once `function_generator` (used only inside the backend constructor) has a
    stand-in."""
    fg = types.ModuleType("function_generator")
    fg.FunctionGenerator = type("FunctionGenerator", (), {})
    sys.modules.setdefault("function_generator", fg)
    import ipde.grid_evaluators.laplace_grid_evaluator as L
    import ipde.grid_evaluators.modified_helmholtz_grid_evaluator as M
    rng = np.random.default_rng(11)
    r = np.concatenate([10.0 ** rng.uniform(-6, 1, 200), [1e-3, 1.0, 7.5]])
    kv = np.fft.fftfreq(24, 0.05 / (2 * np.pi))
    kx, ky = np.meshgrid(kv, kv, indexing='ij')
    kk = np.hypot(kx, ky)
    out = {"r": r, "kx": kx, "ky": ky}
    out["laplace_gf"] = np.array([L.gf(x) for x in r])
    out["laplace_fs"] = L.fs(kx, ky)
    out["laplace_ifs"] = L.ifs(kx.copy(), ky.copy())
    for i, Lt in enumerate((1.0, 7.5, 0.3)):
        out["L_%d" % i] = np.array(Lt)
        out["laplace_tsgf_%d" % i] = L.trunc_sgf(kk, Lt)
        out["laplace_tsgf_scalar_%d" % i] = np.array([L.trunc_sgf(0.0, Lt), L.trunc_sgf(2.5, Lt)])
    for j, hk in enumerate((1.0, 10.0)):
        out["hk_%d" % j] = np.array(hk)
        out["modhelm_gf_%d" % j] = M.gf(r, helmholtz_k=hk)
        out["modhelm_fs_%d" % j] = M.fs(kx, ky, helmholtz_k=hk)
        # M.ifs cannot be called: it drops helmholtz_k (`1.0/fs(kx, ky)`, :12-13) -> TypeError
        for i, Lt in enumerate((1.0, 7.5, 0.3)):
            out["modhelm_tsgf_%d_%d" % (j, i)] = M.trunc_sgf(kk, Lt, helmholtz_k=hk)
    np.savez(os.path.join(OUT, "grid_evaluator_kernels.npz"), **out)
    print("grid_evaluator_kernels.npz", len(out))


# ---- layer-potential kernels: what the tree itself can compute -------------------------

_ABSENT = ("pybie2d", "pyfmmlib2d", "qfs", "flexmm", "fmm2dpy", "near_finder",
           "function_generator", "finufft")


class _Stub:
    """attribute of a stand-in package: importable, raises when called"""

    def __init__(self, name):
        self._n = name

    def __getattr__(self, a):
        if a.startswith("__"):
            raise AttributeError(a)
        return _Stub(self._n + "." + a)

    def __call__(self, *a, **k):
        raise RuntimeError("stand-in %s called" % self._n)


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, a):
        if a.startswith("__"):
            raise AttributeError(a)
        return _Stub(self.__name__ + "." + a)


def _install_absent_package_finder():
    """Make `import pybie2d...`, `from qfs.two_d_qfs import X`, ... succeed with empty
    stand-ins (the packages are not in the image and not under the reference tree), so that
    the reference modules which ALSO hold plain-numpy kernel arithmetic can be imported:
    ipde/solvers/multi_boundary/poisson.py (Laplace_Eval, :10-17) and
    ipde/solvers/internals/stokes_save.py (PSLP / PDLP pressure rows, eval_p1, :29-81)."""
    import importlib.abc
    import importlib.machinery

    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in _ABSENT:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            return _StubModule(spec.name)

        def exec_module(self, m):
            pass
    sys.meta_path.append(Finder())


class _Src:
    pass


def _star_source(n, a, f, scale=1.0, cx=0.0, cy=0.0):
    t = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
    r = 1.0 + a * np.cos(f * t)
    rp = -a * f * np.sin(f * t)
    s = _Src()
    s.x = scale * r * np.cos(t) + cx
    s.y = scale * r * np.sin(t) + cy
    xp = scale * (rp * np.cos(t) - r * np.sin(t))
    yp = scale * (rp * np.sin(t) + r * np.cos(t))
    sp = np.hypot(xp, yp)
    s.normal_x, s.normal_y = yp / sp, -xp / sp
    s.weights = sp * 2 * np.pi / n
    s.N = n
    return s


def golden_layer_kernels():
    """Layer-potential numbers computed by the reference's own code:
      * Laplace single layer through `Laplace_Eval` (multi_boundary/poisson.py:10-17) and
        through `gf` (grid_evaluators/laplace_grid_evaluator.py:8-12);
      * modified-Helmholtz single layer through `gf`
        (grid_evaluators/modified_helmholtz_grid_evaluator.py:8-9), k = 0.7, 10, 40;
      * Stokes single- and double-layer PRESSURE through the last rows of `PSLP` / `PDLP`
        and through `eval_p1` (solvers/internals/stokes_save.py:29-81).
    Everything else on the path (Laplace / modified-Helmholtz double layers, Stokes
    velocities) is computed by pybie2d / pyfmmlib2d, absent here: not in this fixture."""
    import ipde.solvers.multi_boundary.poisson as RP
    import ipde.solvers.internals.stokes_save as RS
    import ipde.grid_evaluators.laplace_grid_evaluator as RL
    import ipde.grid_evaluators.modified_helmholtz_grid_evaluator as RM
    assert RP.flexmm_okay
    # velocity blocks of PSLP / PDLP come from pybie2d (absent): zero them, the pressure
    # row below them is the reference's own arithmetic
    RS.Stokes_Layer_Form = lambda src, trg, **kw: np.zeros((2 * trg.N, 2 * src.N))
    rng = np.random.default_rng(404)
    src = _star_source(120, 0.2, 5, scale=0.9, cx=0.1, cy=-0.2)
    nt = 160
    th = rng.uniform(0, 2 * np.pi, nt)
    # inside (0.05..0.7 of the local radius) and outside (1.3..2.5), plus four near-curve points
    fac = np.where(rng.uniform(size=nt) < 0.5, rng.uniform(0.05, 0.7, nt), rng.uniform(1.3, 2.5, nt))
    fac[:4] = [0.97, 1.03, 0.995, 1.005]
    rad = 0.9 * (1.0 + 0.2 * np.cos(5 * th)) * fac
    tx, ty = 0.1 + rad * np.cos(th), -0.2 + rad * np.sin(th)
    sig = rng.standard_normal(src.N)
    f = rng.standard_normal((2, src.N))
    g = rng.standard_normal((2, src.N))
    out = dict(sx=src.x, sy=src.y, nx=src.normal_x, ny=src.normal_y, w=src.weights,
               tx=tx, ty=ty, sigma=sig, force=f, dipstr=g)
    q = sig * src.weights
    out["laplace_slp_eval"] = np.array(
        [np.sum(RP.Laplace_Eval(src.x, src.y, tx[i], ty[i]) * q) for i in range(nt)])
    R = np.hypot(tx[:, None] - src.x[None, :], ty[:, None] - src.y[None, :])
    out["laplace_slp_gf"] = np.array([np.sum(RL.gf(R[i]) * q) for i in range(nt)])
    ks = np.array([0.7, 10.0, 40.0])
    out["modhelm_k"] = ks
    for j, k in enumerate(ks):
        out["modhelm_slp_gf_%d" % j] = np.array(
            [np.sum(RM.gf(R[i], helmholtz_k=k) * q) for i in range(nt)])
    ps, pd, p1s, p1b = np.zeros(nt), np.zeros(nt), np.zeros(nt), np.zeros(nt)
    for i in range(nt):
        trg = _Src()
        trg.x, trg.y, trg.N = tx[i:i + 1], ty[i:i + 1], 1
        ps[i] = RS.PSLP(src, trg)[-1].dot(f.ravel())
        pd[i] = RS.PDLP(src, trg)[-1].dot(g.ravel())
        p1s[i] = RS.eval_p1(src, tx[i], ty[i], f)
        p1b[i] = RS.eval_p1(src, tx[i], ty[i], f, g)
    out["stokes_p_slp_row"], out["stokes_p_dlp_row"] = ps, pd
    out["stokes_p_slp_eval_p1"], out["stokes_p_both_eval_p1"] = p1s, p1b
    np.savez(os.path.join(OUT, "layer_kernels.npz"), **out)
    print("layer_kernels.npz", len(out))


def golden_utilities():
    """ipde/utilities.py (a12): the FFT aliases (numpy's here: mkl_fft is absent), the
    Nyquist-dropping and Nyquist-zeroing transforms, the small host helpers."""
    import scipy.linalg
    import ipde.utilities as U
    rng = np.random.default_rng(505)
    out = {}
    a = rng.standard_normal((6, 32)) + 1j * rng.standard_normal((6, 32))
    r = rng.standard_normal((5, 24))
    g = rng.standard_normal((20, 18)) + 1j * rng.standard_normal((20, 18))
    m = rng.standard_normal((6, 32))
    out["a"], out["r"], out["g"], out["m"] = a, r, g, m
    out["fft"], out["ifft"] = U.fft(a), U.ifft(a)
    out["fft2"], out["ifft2"] = U.fft2(g), U.ifft2(g)
    out["fft2_real"] = U.fft2(g.real.copy())
    out["mfft"] = U.mfft(r)
    out["mifft"], out["mifftr"] = U.mifft(out["mfft"]), U.mifftr(out["mfft"])
    ah = U.mfft(a)                                           # (6, 31)
    out["fourier_multiply"] = U.fourier_multiply(ah.copy(), m)
    out["ffourier_multiply"] = U.ffourier_multiply(a.copy(), m)
    out["pfourier_multiply"] = U.pfourier_multiply(a.copy(), m)
    out["pfft"], out["pifft"], out["pifftr"] = U.pfft(r), U.pifft(a.copy()), U.pifftr(a.copy())
    out["fast_dot_mv"] = U.fast_dot(m, m[0])
    out["fast_dot_vm"] = U.fast_dot(m[:, 0], m)
    out["fast_dot_mm"] = U.fast_dot(m[:, :6], m)
    out["concat"] = U.concat(r[0], 3.0, [1.0, 2.0])
    out["affine"] = U.affine_transformation(r[0], -2.0, 3.0, 0.0, 2 * np.pi)
    xc, x, rat = U.get_chebyshev_nodes(-0.3, 0.1, 12)
    out["cheb_unscaled"], out["cheb_scaled"], out["cheb_ratio"] = xc, x, np.array([rat])
    A = rng.standard_normal((7, 7)) + 1j * rng.standard_normal((7, 7))
    b = rng.standard_normal(7) + 1j * rng.standard_normal(7)
    out["lu_A"], out["lu_b"] = A, b
    out["fast_LU_solve"] = U.fast_LU_solve(scipy.linalg.lu_factor(A), b)
    np.savez(os.path.join(OUT, "utilities.npz"), **out)
    print("utilities.npz", len(out))


def golden_fourier_filter():
    """ipde/utilities.py:126-162 SimpleFourierFilter: both filter types, real / complex samples and
    the spectrum-in / spectrum-out call forms."""
    import ipde.utilities as U
    rng = np.random.default_rng(606)
    out = {}
    n = 48
    modes = np.fft.fftfreq(n, 1.0 / n)
    fr = rng.standard_normal(n)
    fc = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    out["modes"], out["fr"], out["fc"] = modes, fr, fc
    F = U.SimpleFourierFilter(modes, 'fraction', fraction=2.0 / 3.0)
    out["fraction_filter"] = F.filter
    out["fraction_real"], out["fraction_cplx"] = F(fr), F(fc)
    out["fraction_spec_in"] = F(np.fft.fft(fr), input_type='fourier')
    out["fraction_spec_out"] = F(fr, output_type='fourier')
    R = U.SimpleFourierFilter(modes, 'rule 36')
    out["rule36_filter"], out["rule36_real"] = R.filter, R(fr)
    R8 = U.SimpleFourierFilter(modes, 'rule 36', power=8)
    out["rule8_filter"], out["rule8_cplx"] = R8.filter, R8(fc)
    np.savez(os.path.join(OUT, "fourier_filter.npz"), **out)
    print("fourier_filter.npz", len(out))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated where it is")
    _install_standins()
    sys.path.insert(0, REF)
    if sys.argv[1:] == ["fourier_filter"]:      # one fixture alone (the others are unchanged)
        golden_fourier_filter()
        raise SystemExit(0)
    golden_derivatives()
    golden_annular_scalar()
    golden_annular_stokes()
    golden_slepian()
    golden_grid_evaluator_kernels()
    golden_utilities()
    golden_fourier_filter()
    _install_absent_package_finder()
    golden_layer_kernels()
