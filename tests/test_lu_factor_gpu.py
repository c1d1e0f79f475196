"""ipde_dense_lu_factor (csrc/lu_factor.hip): the library's own blocked LU with partial pivoting
on the tiled storage, against host LAPACK — pivots, factors, residuals; sizes around the tile and
panel edges, both panel-kernel instantiations, and a QFS collocation matrix (cond ~1e12)."""
import time

import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


def _untile(T, n):
    nb = T.shape[0]
    return T.permute(0, 3, 1, 2).reshape(nb * 64, nb * 64)[:n, :n].cpu().numpy()


def _lapack_perm(piv):
    p = np.arange(len(piv))
    for i, q in enumerate(piv):
        if q != i:
            p[i], p[q] = p[q], p[i]
    return p


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 127, 128, 129, 300, 1000, 2050])
def test_own_factorisation_matches_lapack(n):
    import torch
    from ipde_amd import qfs
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n))
    b = rng.standard_normal(n)
    f = qfs._own_lu(torch.as_tensor(A, device="cuda"))
    lu, piv = scipy.linalg.lu_factor(A)
    # the same pivots (random data: no ties), hence the same factors up to rounding
    assert np.array_equal(f.perm.cpu().numpy(), _lapack_perm(piv))
    mine = _untile(f.LU, n)
    assert np.abs(mine - lu).max() < 1e-11 * max(1.0, np.abs(lu).max())
    # identity padding untouched: unit diagonal, nothing else
    full = f.LU.permute(0, 3, 1, 2).reshape(f.LU.shape[0] * 64, -1).cpu().numpy()
    assert np.array_equal(full[n:, n:], np.eye(full.shape[0] - n))
    assert not full[n:, :n].any() and not full[:n, n:].any()
    x = f._subst(torch.as_tensor(b, device="cuda")).cpu().numpy()
    ref = scipy.linalg.lu_solve((lu, piv), b)
    assert np.abs(A @ x - b).max() < 20 * max(np.abs(A @ ref - b).max(), 1e-15 * n * np.abs(x).max())


def test_own_factorisation_pivots_like_dgetf2_on_ties_and_zeros():
    """first row of maximal |a| wins a tie; a column that is already zero below the diagonal
    keeps the diagonal; integer data so every operation is exact"""
    import torch
    from ipde_amd import qfs
    n = 70
    A = np.zeros((n, n))
    A[np.arange(n), np.arange(n)] = 2.0
    A[5, 3] = -2.0        # tie with the diagonal of column 3: the diagonal (row 3) comes first
    A[60, 10] = 4.0       # larger below: row 60 must come up
    A[65, 64] = -8.0      # in the second panel
    f = qfs._own_lu(torch.as_tensor(A, device="cuda"))
    lu, piv = scipy.linalg.lu_factor(A)
    assert np.array_equal(f.perm.cpu().numpy(), _lapack_perm(piv))
    assert np.array_equal(_untile(f.LU, n), lu)


def test_own_factorisation_on_qfs_matrix_has_lapack_residual():
    import torch
    from ipde_amd import qfs
    from ipde_amd.pybie2d_compat import Global_Smooth_Boundary, star, Laplace_Layer_Form
    b = Global_Smooth_Boundary(c=star(1000, a=0.2, f=5))
    qb = qfs.QFS_Boundary(b, eps=1e-12)
    A = Laplace_Layer_Form(qb.interior_source_bdy, b, ifcharge=True)
    u = np.exp(np.cos(b.t)) + 0.3 * np.sin(3 * b.t)
    Ad = torch.as_tensor(A, device="cuda")
    x = qfs._own_lu(Ad)._subst(torch.as_tensor(u, device="cuda")).cpu().numpy()
    xs = scipy.linalg.solve(A, u)
    r, rs = np.abs(A @ x - u).max(), np.abs(A @ xs - u).max()
    print("cond %.1e  residual own %.2e  LAPACK %.2e" % (np.linalg.cond(A), r, rs))
    assert r < 50 * max(rs, 1e-15)


@pytest.mark.parametrize("n", [4096, 5000])
def test_own_factorisation_large_and_timing(n):
    """n = 4096: the BASELINE boundary size (panel kernel with four rows per thread); 5000: the
    eight-rows-per-thread instantiation.  Timed against rocSOLVER's getrf on the same matrix."""
    import torch
    from ipde_amd import qfs
    rng = np.random.default_rng(n)
    A = torch.as_tensor(rng.standard_normal((n, n)) + 0.05 * n * np.eye(n), device="cuda")
    b = torch.as_tensor(rng.standard_normal(n), device="cuda")
    f = qfs._own_lu(A)
    x = f._subst(b)
    res = float((A @ x - b).abs().max())
    _, hpiv = scipy.linalg.lu_factor(A.cpu().numpy(), check_finite=False)
    assert np.array_equal(f.perm.cpu().numpy(), _lapack_perm(hpiv))
    lu, piv = torch.linalg.lu_factor(A)
    g = qfs._DeviceLU(lu, piv)
    res_lib = float((A @ g._subst(b) - b).abs().max())
    assert res < 20 * max(res_lib, 1e-13)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        qfs._own_lu(A)
    torch.cuda.synchronize()
    t_own = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        torch.linalg.lu_factor(A)
    torch.cuda.synchronize()
    t_lib = (time.perf_counter() - t0) / 3
    print("n = %d: own factorisation (incl. tiling) %.1f ms, rocSOLVER getrf %.1f ms; residuals %.1e / %.1e"
          % (n, t_own * 1e3, t_lib * 1e3, res, res_lib))


@pytest.mark.parametrize("n", [8300, 9600, 19200])
def test_own_factorisation_beyond_8192_rows_multi_cu_panel(n):
    """Systems beyond 8192 padded rows: the panel dealt out to several workgroups, a row per thread, one exchange
    per column (csrc/lu_factor.hip: lu_panel_multi_kernel); 19 200 = BASELINE configs[4]'s Stokes QFS systems
    (reference ipde/solvers/internals/vector.py:124-125).  LAPACK's pivots — HOST LAPACK's: rocSOLVER's getrf leaves
    them at n = 8300 (its last hundred pivots from row 8196 on differ from dgetrf's, measured; at 9600 the three
    agree) — and the residual of the library's factors; timed against rocSOLVER."""
    import torch
    from ipde_amd import qfs
    from ipde_amd.device import get_context
    rng = np.random.default_rng(n)
    A = torch.as_tensor(rng.standard_normal((n, n)), device="cuda")
    b = torch.as_tensor(rng.standard_normal(n), device="cuda")
    f = qfs._own_lu(A)
    get_context().sync()            # (reports a hand-off time-out of the panel kernel, if there was one)
    x = f._subst(b)
    res = float((A @ x - b).abs().max())
    _, hpiv = scipy.linalg.lu_factor(A.cpu().numpy(), check_finite=False)
    assert np.array_equal(f.perm.cpu().numpy(), _lapack_perm(hpiv))
    lu, piv = torch.linalg.lu_factor(A)
    g = qfs._DeviceLU(lu, piv)
    res_lib = float((A @ g._subst(b) - b).abs().max())
    assert res < 20 * max(res_lib, 1e-12)
    del lu, piv, g
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        qfs._own_lu(A)
    torch.cuda.synchronize()
    t_own = (time.perf_counter() - t0) / 2
    t0 = time.perf_counter()
    for _ in range(2):
        torch.linalg.lu_factor(A)
    torch.cuda.synchronize()
    t_lib = (time.perf_counter() - t0) / 2
    print("n = %d: own factorisation (incl. tiling) %.1f ms, rocSOLVER getrf %.1f ms; residuals %.1e / %.1e"
          % (n, t_own * 1e3, t_lib * 1e3, res, res_lib))


def test_lu_factor_argument_checks():
    import torch
    from ipde_amd.device import get_context, ptr
    ctx = get_context()
    T = torch.eye(128, dtype=torch.float64, device="cuda")
    p = torch.empty(128, dtype=torch.int32, device="cuda")
    call = ctx.lib.ipde_dense_lu_factor
    assert call(ctx.handle, 128, ptr(T), ptr(p)) == 0
    assert call(ctx.handle, 64, ptr(T), ptr(p)) == 1          # not a multiple of 128
    assert call(ctx.handle, 192, ptr(T), ptr(p)) == 1
    assert call(ctx.handle, 32768 + 128, ptr(T), ptr(p)) == 1  # beyond the multi-CU panel's 64 workgroups of 512 rows
    assert call(ctx.handle, 128, None, ptr(p)) == 1
    assert call(ctx.handle, 128, ptr(T), None) == 1
    assert call(None, 128, ptr(T), ptr(p)) == 1
    ctx.sync()


def test_factorisations_on_side_streams_equal_the_in_stream_one():
    """qfs._OwnAsyncLU (private contexts bound to non-blocking torch streams, round robin): five
    matrices factored back to back give the factors and pivots of the in-stream call"""
    import torch
    from ipde_amd import qfs
    rng = np.random.default_rng(8)
    mats = [torch.as_tensor(rng.standard_normal((n, n)), device="cuda") for n in (700, 1300, 200, 1300, 64)]
    side = [qfs._OwnAsyncLU(A) for A in mats]          # all enqueued before any is used
    for A, f in zip(mats, side):
        g = qfs._own_lu(A)
        assert torch.equal(f.perm, g.perm) and torch.equal(f.LU, g.LU)
        b = torch.as_tensor(rng.standard_normal(A.shape[0]), device="cuda")
        assert torch.equal(f._subst(b), g._subst(b))


def test_paired_trailing_updates_give_the_bits_of_panel_by_panel_updates():
    """One trailing update per PAIR of panels (lu_update_pair_kernel) runs the same MFMA chain per element as two
    single-panel updates: factors and pivots bitwise equal.  The schedule is a process-wide switch
    (IPDE_LU_SINGLE_UPDATES), so the two runs are child processes; sizes with an odd / even number of panel pairs and
    one beyond 8192 rows (multi-workgroup panels)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, hashlib, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "from ipde_amd import qfs\n"
            "out = {}\n"
            "for n in (300, 1100, 2050, 8300):\n"
            "    A = torch.as_tensor(np.random.default_rng(n).standard_normal((n, n)), device='cuda')\n"
            "    f = qfs._own_lu(A)\n"
            "    torch.cuda.synchronize()\n"
            "    out[str(n)] = [hashlib.sha256(f.LU.cpu().numpy().tobytes()).hexdigest(),\n"
            "                   hashlib.sha256(f.perm.cpu().numpy().tobytes()).hexdigest()]\n"
            "print(json.dumps(out))\n" % root)
    res = []
    for single in (False, True):
        env = dict(os.environ)
        env.pop("IPDE_LU_SINGLE_UPDATES", None)
        if single:
            env["IPDE_LU_SINGLE_UPDATES"] = "1"
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]))
    assert res[0] == res[1]
