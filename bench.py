#!/usr/bin/env python3
"""Headline benchmark: Laplace SLP boundary->grid evaluation (BASELINE.json configs[1]).

  python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one Laplace single-layer evaluation of a 4096-node star boundary
onto a 2048^2 grid (points within 7.5 h of the curve removed, as the solver never
evaluates on-surface) through the C ABI, inputs resident in HBM: the list, cut once
into 4 x 4 patches as the Poisson solver keeps grid_pnai (ipde_amd/target_plan.py, set-up), goes
through ipde_laplace_apply_patches; --no-patches times ipde_laplace_apply on the plain list.  With N > 1 (launched by torch.distributed.run, one rank per GPU)
the ONE 2048^2 target list is split into N contiguous slices (strong scaling, the
north_star's "2048^2 grid / 4096-node boundary at 1, 2, 4 and 8 GPUs"); every rank owns
1/N of the boundary density, which is all-gathered over RCCL each step — the one real
exchange of the path; results stay sharded (no data-path collective on the targets).
`--scaling weak` gives every rank a whole 2048^2 grid instead.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NGRID = 2048
NBDY = 4096
FLOPS_PER_PAIR_ALGO = 8        # SURVEY §8d counting convention (5 geometry + log + fma)
VALU_INSTR_PER_PAIR = 11.0     # ISA count of the row-run table kernel: 8.5 fp64 (2.5 geometry with
                               # (x-sx)^2 shared by a lane's 4 targets, y, 4 fma log1p + T, accumulate)
                               # + 2.5 int32 (v_bfe, v_lshl_add, one v_min3_u32 per two pairs for the
                               # lower table bound; the upper one is guaranteed by the scaling); every VALU
                               # instruction, fp64 or int32, occupies the SIMD for one quad-cycle here
VALU_INSTR_PER_PAIR_PATCH = 10.4   # ISA count of laplace_patch_kernel: 8.06 fp64 (eight subs and squares per
                                   # source and 16 targets, d2 = one fma, y, 4 fma, accumulate) + 2 int32 +
                                   # 0.34 for the table bound (min of dx^2 + min of dy^2 per patch)
PEAK_VALU_LANE_INSTR = 256 * 4 * 16 * 2.4e9   # CUs x SIMDs x lanes/clk x Hz
PEAK_FP64_VECTOR_TFLOPS = 78.6  # MI355X fp64 vector (= 256 CU*4 SIMD*32 flop/clk*2.4 GHz)
PEAK_HBM_GBS = 8000.0
ALGO_BYTES_PER_TARGET = 24     # read x, y; write u
ALGO_BYTES_PER_PATCH = 8 * 8 + 16 * 4 + 16 * 8   # patch kernel: 4 xs + 4 ys, 16 int32 positions, 16 values


def make_workload():
    from util import Curve, grid_targets
    c = Curve(NBDY, a=0.2, f=5)
    trg, h = grid_targets(c, NGRID, lim=1.5, clearance=5.0)
    rng = np.random.default_rng(0)
    sigma = rng.standard_normal(NBDY)
    return c, trg, sigma


def cpu_baseline(c, trg, sigma, budget_s=15.0):
    """The C/OpenMP restatement (oracle/) on a bounded sample of the same workload."""
    import oracle
    q = sigma * c.weights
    threads = oracle.c_oracle().oracle_num_threads()
    n0 = min(trg.N, 32768)
    t0 = time.perf_counter()
    oracle.c_laplace_apply(c.x, c.y, trg.x[:n0], trg.y[:n0], w_sigma=q)
    t = time.perf_counter() - t0
    rate = c.N * n0 / t
    n1 = int(min(trg.N, max(n0, rate * budget_s / c.N)))
    t0 = time.perf_counter()
    oracle.c_laplace_apply(c.x, c.y, trg.x[:n1], trg.y[:n1], w_sigma=q)
    t = time.perf_counter() - t0
    return {
        "value": c.N * n1 / t, "unit": "pair-interactions/s", "cores": int(threads),
        "kind": "port",
        "sample": "first %d of %d targets x %d sources, C/OpenMP oracle (gcc -O3, libm log), %.1f s"
                  % (n1, trg.N, c.N, t),
    }


def far_form_work(plan, c):
    """Pairs by kind of one far-field apply, counted on the host with the kernels' own rules (csrc/layer_common.h
    FarBlock, csrc/layer_laplace.hip laplace_far_coeff_kernel): blocks = 64 consecutive patches, parents = 16
    consecutive blocks, discs = bounding-box centre and half-diagonal, batches of eight consecutive sources."""
    pxy = plan.pxy.cpu().numpy()                     # (8, np)
    npch = pxy.shape[1]
    ng = (npch + 63) // 64
    pad = ng * 64 - npch
    if pad:
        pxy = np.concatenate([pxy, np.repeat(pxy[:, -1:], pad, axis=1)], axis=1)
    xs, ys = pxy[:4].T.reshape(ng, 256), pxy[4:].T.reshape(ng, 256)

    def discs(x, y):
        cx, cy = 0.5 * (x.min(1) + x.max(1)), 0.5 * (y.min(1) + y.max(1))
        r2 = (0.5 * (x.max(1) - x.min(1))) ** 2 + (0.5 * (y.max(1) - y.min(1))) ** 2
        return cx, cy, r2
    bcx, bcy, br2 = discs(xs, ys)
    ng2 = (ng + 15) // 16
    padg = ng2 * 16 - ng
    xs2 = np.concatenate([xs, np.repeat(xs[-1:], padg, axis=0)]).reshape(ng2, -1)
    ys2 = np.concatenate([ys, np.repeat(ys[-1:], padg, axis=0)]).reshape(ng2, -1)
    pcx, pcy, pr2 = discs(xs2, ys2)
    ns = c.N
    nb = (ns + 7) // 8
    sxp = np.concatenate([c.x, np.repeat(c.x[-1:], nb * 8 - ns)])
    syp = np.concatenate([c.y, np.repeat(c.y[-1:], nb * 8 - ns)])
    # parents: a batch is taken if all eight of its sources are beyond 4 parent radii
    d2p = (sxp[None, :] - pcx[:, None]) ** 2 + (syp[None, :] - pcy[:, None]) ** 2          # (ng2, ns_pad)
    taken = (d2p >= 16.0 * pr2[:, None]).reshape(ng2, nb, 8).all(axis=2)                   # (ng2, nb)
    far_p = int(taken.sum()) * 8
    far_b = near = 0
    for g0 in range(0, ng, 4096):                    # blocks in slabs (memory)
        g1 = min(ng, g0 + 4096)
        d2 = (sxp[None, :] - bcx[g0:g1, None]) ** 2 + (syp[None, :] - bcy[g0:g1, None]) ** 2
        farb = (d2 >= 16.0 * br2[g0:g1, None]).reshape(g1 - g0, nb, 8).all(axis=2)
        mine = ~taken[np.arange(g0, g1) // 16]
        far_b += int((farb & mine).sum()) * 8
        near += int((~farb & mine).sum()) * 8
    return {"blocks": int(ng), "parents": int(ng2), "targets": int(16 * npch),
            "far_pairs_parents": float(far_p), "far_pairs_blocks": float(far_b),
            "near_pairs": float(near) * 1024.0,
            "near_fraction_of_all_pairs": float(near) * 1024.0 / (float(ns) * 16.0 * npch)}


def f_like(solver, EmbeddedFunction):
    f = EmbeddedFunction(solver.ebdyc)
    f.define_via_function(lambda x, y: (2.0 * np.cos(x) + 3.0 * np.cos(x) * np.sin(x) - np.cos(x) ** 3)
                          * np.exp(np.sin(x)) * np.sin(y))
    return f


def full_poisson_solve(nb=4096, ng=2048, M=20):
    """examples/interior_poisson.py at BASELINE configs[2] (the reference's
    examples/poisson_for_paper.py brackets: set-up / inhomogeneous solve / homogeneous
    correction), plus a second, warm inhomogeneous solve on the same solver."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import interior_poisson
    from ipde_amd.embedded_function import EmbeddedFunction
    t0 = time.perf_counter()
    err, scale, solver, ue, T = interior_poisson.run(nb=nb, M=M, Ns=[ng, ng], solver_tol=1e-12)
    total = time.perf_counter() - t0
    from ipde_amd import hostio

    def warm_ms(slv, reps=10):
        """mean of `reps` warm solves: host containers in and out (the reference's call), and the
        same with the right-hand side and the answer resident in HBM (hostio.DeviceFunction)"""
        out = []
        for resident in (False, True):
            g = f_like(slv, EmbeddedFunction)
            if resident:
                g = hostio.DeviceFunction.from_host(g)
            slv(g, tol=1e-12, maxiter=100, restart=20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                slv(g, tol=1e-12, maxiter=100, restart=20)
            torch.cuda.synchronize()
            out.append(1e3 * (time.perf_counter() - t0) / reps)
        return out

    warm, warm_res = warm_ms(solver)
    far_default = bool(solver.FAR_EXPANSION)
    # the reference's whole timed bracket, warm (examples/poisson_for_paper.py:72-92): inhomogeneous solve +
    # homogeneous apply (boundary values, two dense solves, ONE sum onto grid_and_radial_pts, the add) — with
    # the reference's host containers, and with right-hand side, answer and correction resident in HBM
    corr = T.pop("correction")
    T.pop("f", None)

    def warm_brackets(reps=10):
        out = {}
        for resident in (False, True):
            g = f_like(solver, EmbeddedFunction)
            if resident:
                g = hostio.DeviceFunction.from_host(g)
            u = corr(solver(g, tol=1e-12, maxiter=100, restart=20))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                u = corr(solver(g, tol=1e-12, maxiter=100, restart=20))
            torch.cuda.synchronize()
            both = 1e3 * (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                u = corr(u)          # (the stage alone, on the answer it has: same work whatever the values)
            torch.cuda.synchronize()
            out["resident" if resident else "host"] = (both, 1e3 * (time.perf_counter() - t0) / reps)
        # the one sum of the stage alone (device tensors in and out)
        sig = torch.randn(solver.ebdyc.bdy_inward_sources.N, dtype=torch.float64, device="cuda")
        for _ in range(3):
            corr.layer_apply(solver.ebdyc.bdy_inward_sources, corr.targets, sig)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            corr.layer_apply(solver.ebdyc.bdy_inward_sources, corr.targets, sig)
        torch.cuda.synchronize()
        out["sum"] = 1e3 * (time.perf_counter() - t0) / reps
        return out
    wb = warm_brackets()
    n_corr_targets, n_corr_sources = int(corr.targets.N), int(solver.ebdyc.bdy_inward_sources.N)
    del corr
    # every pair of the sum onto grid_pnai directly (the reference's grid_backend='pybie2d' branch)
    del solver, ue
    torch.cuda.empty_cache()
    err_p, scale_p, solver_p, _, _ = interior_poisson.run(nb=nb, M=M, Ns=[ng, ng], solver_tol=1e-12,
                                                          grid_backend='pybie2d')
    warm_p, warm_p_res = warm_ms(solver_p)
    # the same solve with the O(N_s sw^2 + n^2 log n) split grid evaluator instead of the
    # dense sum onto the grid (grid_backend='ewald', 7e-15 from the dense sum)
    del solver_p
    torch.cuda.empty_cache()
    Te = {}
    err_e, scale_e, solver_e, _, Te = interior_poisson.run(nb=nb, M=M, Ns=[ng, ng], solver_tol=1e-12,
                                                           grid_backend='ewald', timings=Te)
    Te.pop("correction", None)
    warm_e, warm_e_res = warm_ms(solver_e)
    return {
        "workload": "interior Poisson, %d^2 grid, %d-node star boundary, M = %d, %d dof" % (ng, nb, M, T["dof"]),
        "max_rel_err_vs_manufactured_solution": err / scale,
        # brackets of the reference's examples/poisson_for_paper.py:60-92: set-up = geometry, grid
        # registration and solver construction; the manufactured f / u / boundary data (numpy on
        # 2.2 M points) are defined after it, as there
        "setup_s": T["setup_s"], "problem_definition_s": T["problem_definition_s"],
        "first_inhomogeneous_solve_s": T["inhomogeneous_solve_s"],
        "homogeneous_correction_s": T["homogeneous_form_s"] + T["homogeneous_apply_s"],
        "end_to_end_s": total, "warm_inhomogeneous_solve_ms": warm,
        "warm_inhomogeneous_solve_resident_ms": warm_res,
        "warm_homogeneous_apply_ms": wb["host"][1], "warm_homogeneous_apply_resident_ms": wb["resident"][1],
        "warm_end_to_end_solve_ms": wb["host"][0], "warm_end_to_end_solve_resident_ms": wb["resident"][0],
        "homogeneous_sum_ms": wb["sum"],
        "homogeneous_sum": "%d sources x %d targets (grid_and_radial_pts resident: grid points through "
                           "ipde_laplace_apply_patches_far, the radial grid through ipde_laplace_apply_columns_far)"
                           % (n_corr_sources, n_corr_targets),
        "warm_bracket_note": "warm_end_to_end_solve = the reference's timed bracket examples/poisson_for_paper.py:72-92 "
                             "(inhomogeneous solve + homogeneous apply), means of 10 on the solver and the correction "
                             "object the first solve built; homogeneous_form (dense DLP matrix, LU, QFS) is set-up of "
                             "the stage and is in homogeneous_correction_s above",
        "grid_sum": "far sources of every 8 x 8 block of patches in local expansions (ipde_laplace_apply_patches_far)"
                    if far_default else "every pair directly",
        "pair_by_pair_grid_sum": {"max_rel_err_vs_manufactured_solution": err_p / scale_p,
                                  "warm_inhomogeneous_solve_ms": warm_p,
                                  "warm_inhomogeneous_solve_resident_ms": warm_p_res},
        "ewald_grid_backend": {"max_rel_err_vs_manufactured_solution": err_e / scale_e,
                               "setup_s": Te["setup_s"],
                               "warm_inhomogeneous_solve_ms": warm_e,
                               "warm_inhomogeneous_solve_resident_ms": warm_e_res},
        "warm_solve_note": "means of 10 solves; *_resident_ms: right-hand side and answer as hostio.DeviceFunction "
                           "(in HBM before and after, no PCIe crossing); the others take and return the "
                           "reference's host containers (17.6 MB each way)",
        "gmres_iterations": T["gmres_iterations"],
        "note": "end_to_end = set-up + first solve + correction in a process that has only run the "
                "dense-sum benchmark before (one-time library loads included; rocFFT kernels come "
                "from the shipped cache, tools/cold_solve.py times the same from process start)",
    }


def baseline_configs():
    """BASELINE configs[3] and configs[4] on ONE GPU (the 8-GPU split of their target sets is the
    driver's to run): errors against the manufactured solutions, set-up and warm solve of the
    examples as shipped (grid sums in their far-field forms)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import interior_modified_helmholtz as imh
    import multi_stokes as ms
    from ipde_amd.embedded_function import EmbeddedFunction
    out = {}
    err, scale, solver, ue, T = imh.run(nb=8192, M=20, helmholtz_k=10.0, Ns=[4096, 4096])
    f = f_like(solver, EmbeddedFunction)
    solver(f, tol=1e-12, maxiter=100, restart=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        solver(f, tol=1e-12, maxiter=100, restart=20)
    torch.cuda.synchronize()
    warm3 = 1e3 * (time.perf_counter() - t0) / 5
    # the reference's whole bracket warm: solve + homogeneous apply, host containers and resident
    from ipde_amd import hostio
    corr = T.pop("correction")
    T.pop("f", None)
    wb = {}
    for resident in (False, True):
        g = hostio.DeviceFunction.from_host(f) if resident else f
        u = corr(solver(g, tol=1e-12, maxiter=100, restart=20))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            u = corr(solver(g, tol=1e-12, maxiter=100, restart=20))
        torch.cuda.synchronize()
        both = 1e3 * (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            u = corr(u)
        torch.cuda.synchronize()
        wb[resident] = (both, 1e3 * (time.perf_counter() - t0) / 5)
    del corr, u, g
    out["configs[3]"] = {"workload": "interior modified Helmholtz k = 10, 4096^2 grid, 8192-node boundary, %d dof, one GPU"
                                     % T["dof"],
                         "max_rel_err_vs_manufactured_solution": err / scale, "setup_s": T["setup_s"],
                         "warm_inhomogeneous_solve_ms": warm3,
                         "warm_homogeneous_apply_ms": wb[False][1], "warm_homogeneous_apply_resident_ms": wb[True][1],
                         "warm_end_to_end_solve_ms": wb[False][0], "warm_end_to_end_solve_resident_ms": wb[True][0],
                         "first_homogeneous_form_s": T["homogeneous_form_s"],
                         "first_homogeneous_apply_s": T["homogeneous_apply_s"],
                         "grid_backend": str(solver.grid_backend), "gmres_iterations": T["gmres_iterations"]}
    del solver, ue, f
    torch.cuda.empty_cache()
    kept = {}
    orig_call = ms.StokesSolver.__call__

    def keeping(self, fu, fv, **kw):          # (the example's own solver and forcings, for the warm solves below)
        kept.setdefault("args", (self, fu, fv, kw))
        return orig_call(self, fu, fv, **kw)
    ms.StokesSolver.__call__ = keeping
    try:
        ue, ve, pe, scale, T = ms.run(nb=2400, M=14, ng=4096, warm=True)
    finally:
        ms.StokesSolver.__call__ = orig_call
    slv, fu, fv, kw = kept["args"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        slv(fu, fv, **kw)
    torch.cuda.synchronize()
    warm4 = 1e3 * (time.perf_counter() - t0) / 5
    del slv, fu, fv, kept
    out["configs[4]"] = {"workload": "multi_stokes, 3 bodies (9600 + 2 x 2400 nodes), 4096^2 grid, %d dof, one GPU"
                                     % T["dof"],
                         "max_err_u_v": max(ue, ve), "max_err_p": pe, "scale": scale, "setup_s": T["setup_s"],
                         "warm_inhomogeneous_solve_ms": warm4,
                         "second_solve_ms": 1e3 * T["warm_inhomogeneous_solve_s"],
                         "homogeneous_correction_s": T["homogeneous_s"], "homogeneous_sum_ms": 1e3 * T["homogeneous_sum_s"],
                         "warm_solve_note": "mean of five solves after the example's own two",
                         "gmres_iterations": T["gmres_iterations"]}
    torch.cuda.empty_cache()
    return out


def fft_block(n=NGRID, reps=40):
    """The spectral half of the path at the BASELINE grid: `_grid_solve`
    (Poisson) and `fourier` (d/dx), device resident.  GB/s on the algorithmic 16 B per grid
    point (read f, write u = 67 MB at 2048^2), GFLOP/s on 2.5 N log2 N per real transform
    (two per call), fraction of the 8 TB/s HBM peak."""
    import torch
    from ipde_amd.spectral import get_plan
    g = torch.Generator(device="cuda").manual_seed(0)
    f = torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g)
    f -= f.mean()
    plan = get_plan(n, n, 3.0 / n, 3.0 / n)
    out = {"grid": [n, n], "algorithmic_bytes": 16 * n * n,
           "flops_convention": "2 x 2.5 N log2 N (one real forward + one real inverse 2-D transform)"}
    N = float(n) * n
    flops = 2 * 2.5 * N * np.log2(N)
    for name, fn in (("poisson_grid_solve", lambda: plan.poisson_solve(f)),
                     ("fourier_dx", lambda: plan.dx(f))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        gbs = 16 * N / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "GB/s": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                     "GFLOP/s": flops / (ms * 1e-3) / 1e9}
    # HBM-side bytes of one Poisson grid solve (its three kernels): rocprofv3 FETCH_SIZE / WRITE_SIZE
    # passes over tools/profile_fft.py (tools/fft_traffic.py), stamped with the commit they were taken at
    tf = os.path.join(ROOT, "profiles", "traffic_fft_latest.json")
    out["traffic"], out["traffic_source"] = None, None
    if n == NGRID and os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            if tj.get("grid") == [n, n]:
                out["traffic"] = tj.get("poisson_solve_hbm_bytes")
                out["traffic_source"] = {k: tj.get(k) for k in ("poisson_solve_kernels", "measured_at_commit",
                                                                 "measured_on", "command")}
        except Exception:
            pass
    return out


def launch_command(n, argv, port=None):
    """The one-node launch the driver itself uses for N > 1 (one rank per GPU over RCCL)."""
    if port is None:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
        [a for a in argv if a != "--print-launch"]


def self_launch(n, argv, dry=False):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): run the ranks as a child process
    (this process never initialises the GPU), pass its stderr through, print the ONE JSON line rank 0
    wrote, return the child's exit status."""
    import subprocess
    cmd = launch_command(n, argv)
    if dry:
        print(json.dumps(cmd), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")                # torchrun would set 1 and say so on stderr
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for line in child.stdout:
        if line.startswith("{"):
            lines.append(line.rstrip("\n"))
        else:                                             # launcher / library chatter: not on fd 1
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, got %d\n" % len(lines))
        rc = 1
    for l in lines[-1:]:
        print(l, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-solve", action="store_true",
                    help="skip the secondary measurement (full interior Poisson solve, 2048^2 grid)")
    ap.add_argument("--variant", type=int, default=None, help="kernel geometry variant (tuning)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): one 2048^2 target list split over the ranks; "
                         "weak: one 2048^2 grid per rank")
    ap.add_argument("--no-fft", action="store_true", help="skip the spectral-path measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[3] / configs[4] solves")
    ap.add_argument("--no-patches", action="store_true",
                    help="the list kernel (laplace_rowrun_kernel) instead of the 4 x 4 patch kernel")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="all ranks on cuda:0, collectives over gloo through host memory: exercises the "
                         "N > 1 code path on a one-GPU box (a rehearsal, not a measurement)")
    ap.add_argument("--print-launch", action="store_true",
                    help="with --gpus N > 1 from a bare shell: print the child command the self-launch would "
                         "run (a JSON list) and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process, before this
        # process has imported torch or touched the GPU (never re-exec a process that has), relay rank 0's
        # single JSON line and leave with the child's status
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], dry=args.print_launch))

    # RCCL writes a version banner to stdout when the communicator is created:
    # keep fd 1 clean for the single JSON line by pointing it at stderr meanwhile
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    rehearse = args.rehearse_shared_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the process group is always created (also for a
    # single rank), so the collective path can be exercised on a one-GPU box
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if use_dist and rehearse:
        dist.init_process_group("gloo")
    elif use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ipde_amd.device import get_context
    from ipde_amd import layer_potentials as lp, target_plan

    ctx = get_context(local_rank)
    if args.variant is not None:
        ctx.set_option("laplace_variant", args.variant)
    c, trg, sigma = make_workload()
    dev = ctx.torch_device()
    from ipde_amd.sharding import target_slice
    strong = args.scaling == "strong"
    sl = target_slice(trg.N, rank, world) if strong else slice(0, trg.N)
    # the list resident in HBM and cut into 4 x 4 patches, as the Poisson solver holds grid_pnai
    # (set-up, not timed: ipde_amd/target_plan.py); --no-patches: the list kernel
    dt = lp.DeviceTargets(trg.x[sl], trg.y[sl], ctx=ctx, plan=not args.no_patches)
    plan = dt.plan()
    sx = torch.as_tensor(c.x, device=dev)
    sy = torch.as_tensor(c.y, device=dev)
    w = torch.as_tensor(c.weights, device=dev)
    sig_full = torch.as_tensor(sigma, device=dev)
    if NBDY % world:
        raise SystemExit("the boundary density (%d nodes) is split evenly: --gpus must divide it" % NBDY)
    shard = NBDY // world
    sig_shard = sig_full[rank * shard:(rank + 1) * shard].contiguous()
    gathered = torch.empty(NBDY, dtype=torch.float64, device=dev)
    out = torch.empty(dt.N, dtype=torch.float64, device=dev)
    if rehearse:     # gloo moves host memory
        sig_shard_h, gathered_h = sig_shard.cpu(), torch.empty(NBDY, dtype=torch.float64)

    def gather_density():
        if rehearse:
            dist.all_gather_into_tensor(gathered_h, sig_shard_h)
            gathered.copy_(gathered_h)
        else:
            dist.all_gather_into_tensor(gathered, sig_shard)

    def step():
        if use_dist:
            gather_density()
            dens = gathered
        else:
            dens = sig_full
        if plan is not None:
            target_plan.laplace_apply(plan, sx, sy, w_sigma=dens * w, ctx=ctx, out=out)
        else:
            lp.laplace_apply(sx, sy, dt.x, dt.y, w_sigma=dens * w, ctx=ctx, out=out)

    for _ in range(args.warmup):
        step()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the library records a hipEvent pair around the dominant kernel of EVERY apply on the stream it
    # launches on (a ring of 256 pairs, no host sync inside the loop): the timed region's own kernel
    # durations are read after the closing fence
    ctx.enable_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # event pairs per step: the patch kernel, plus the list kernel on the plan's remainder when there is one;
    # the ring holds 256 pairs, so a longer run averages its last whole steps
    ppstep = 2 if (plan is not None and plan.nrest > 0) else 1
    hist = ctx.kernel_ms_history()
    nsteps_seen = min(args.steps, len(hist) // ppstep)
    assert nsteps_seen >= 1, "no kernel event pairs recorded in the timed region"
    # (a step's remainder launch, if any, is recorded BEFORE its patch kernel: the dominant kernel is the step's last pair)
    kms_timed = [float(hist[len(hist) - ppstep * (nsteps_seen - i) + ppstep - 1]) for i in range(nsteps_seen)]
    assert len(kms_timed) == min(args.steps, 256 // ppstep), (len(kms_timed), args.steps)
    kernel_ms_avg = float(np.mean(kms_timed))
    # the same launch isolated (a host sync after each): a separate figure, not the one frac uses
    ctx.sync()
    kms = []
    for _ in range(min(5, max(1, args.steps))):
        step()
        kms.append(ctx.last_kernel_ms())
    kernel_ms_isolated = float(np.mean(kms))
    # the list kernel (ipde_laplace_apply on the plain list) beside it: what any target list gets
    list_kernel_ms = None
    if plan is not None:
        lk = []
        for _ in range(4):
            lp.laplace_apply(sx, sy, dt.x, dt.y, w_sigma=sig_full * w, ctx=ctx, out=out)
            ctx.sync()
            lk.append(ctx.last_kernel_ms())
        list_kernel_ms = float(np.mean(lk[1:]))
        # `out` again from the timed route (parity check below; no collective here)
        target_plan.laplace_apply(plan, sx, sy, w_sigma=sig_full * w, ctx=ctx, out=out)
        ctx.sync()
    ctx.enable_timing(False)
    # the same sum with every 8 x 8 block's far sources in a local expansion (what PoissonSolver
    # uses for its sum onto grid_pnai): a separate, labelled figure — `value` and `roofline` above are the
    # pair-by-pair kernel's
    expansion = None
    layers = None
    if plan is not None and world == 1:
        dtf = lp.DeviceTargets(trg.x[sl], trg.y[sl], ctx=ctx, plan=True, far=True)
        planf = dtf.plan()
        outf = torch.empty(dt.N, dtype=torch.float64, device=dev)
        for _ in range(3):
            target_plan.laplace_apply(planf, sx, sy, w_sigma=sig_full * w, ctx=ctx, out=outf, far=True)
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        tf0 = time.perf_counter()
        nf = 20
        for _ in range(nf):
            target_plan.laplace_apply(planf, sx, sy, w_sigma=sig_full * w, ctx=ctx, out=outf, far=True)
        torch.cuda.synchronize()
        msf = 1e3 * (time.perf_counter() - tf0) / nf
        kmsf = float(np.mean(ctx.kernel_ms_history()[-nf:]))       # parents + blocks + patches, by the library's events
        ctx.enable_timing(False)
        # the three stages apart (option timing_split: an event pair each), with the work each does counted on the
        # host from the plan's blocks by the kernels' own rule — a batch of eight sources enters a parent's expansion
        # if all eight lie beyond 4 parent radii, else a block's if all beyond 4 block radii, else it is summed pair
        # by pair for that block — and priced against the fp64 vector peak
        ctx.set_option("timing_split", 1)
        ctx.enable_timing(True)
        for _ in range(nf):
            target_plan.laplace_apply(planf, sx, sy, w_sigma=sig_full * w, ctx=ctx, out=outf, far=True)
        torch.cuda.synchronize()
        hist3 = np.asarray(ctx.kernel_ms_history()[-3 * nf:]).reshape(nf, 3).mean(axis=0)
        ctx.enable_timing(False)
        ctx.set_option("timing_split", 0)
        work = far_form_work(planf, c)
        P1 = 27                      # coefficients k = 0 .. 26
        flop_cfma = 8.0
        stage = {}
        for name, ms_, flops, what in (
                ("parent_coefficients", hist3[0], work["far_pairs_parents"] * P1 * flop_cfma,
                 "far (source, parent) pairs x 27 complex FMAs"),
                ("block_coefficients", hist3[1], work["far_pairs_blocks"] * P1 * flop_cfma,
                 "far (source, block) pairs x 27 complex FMAs"),
                ("patches", hist3[2], work["near_pairs"] * FLOPS_PER_PAIR_ALGO + 2.0 * work["targets"] * 26 * flop_cfma,
                 "near pairs x 8 flop + targets x 2 levels x 26 complex Horner steps")):
            stage[name] = {"kernel_ms": float(ms_), "flops": float(flops), "flops_convention": what,
                           "frac_of_fp64_vector_peak": float(flops / (ms_ * 1e-3) / 1e12 / PEAK_FP64_VECTOR_TFLOPS)}
        expansion = {"what": "ipde_laplace_apply_patches_far: sources beyond 4 block radii of a 32 x 32-point block "
                             "enter 27 complex local-expansion coefficients (truncation 3e-18 of sum|w|), nearer "
                             "batches of eight sources are summed pair by pair",
                     "ms_per_apply": msf, "kernels_ms": kmsf,
                     "effective_pair_interactions_per_s": float(NBDY) * float(dt.N) / (msf * 1e-3),
                     "max_abs_diff_vs_pair_by_pair": float((outf - out).abs().max()),
                     "max_abs_pair_by_pair": float(out.abs().max()),
                     "padded_patches": planf.np,
                     "stages": stage, "work": work,
                     "roofline_note": "fp64 VALU bound like the pair-by-pair kernel; the coefficient passes visit every "
                                      "source once per block and parent (O(N_s N_blocks)), of which only the far ones do "
                                      "the 27-step power loop counted here"}
        # the other layers of configs[1] ("Laplace SLP+DLP grid_evaluator"): double layer and both in one apply,
        # pair by pair (4 x 4 patch kernel) and in the far-field form
        nrm_x = torch.as_tensor(c.normal_x, device=dev)
        nrm_y = torch.as_tensor(c.normal_y, device=dev)
        tau = torch.as_tensor(np.random.default_rng(1).standard_normal(NBDY), device=dev)
        layers = {}
        for name, kw, fl in (("dlp", dict(nx=nrm_x, ny=nrm_y, w_tau=tau * w), 12),
                             ("slp_dlp", dict(w_sigma=sig_full * w, nx=nrm_x, ny=nrm_y, w_tau=tau * w), 15)):
            ctx.enable_timing(True)
            ks = []
            for _ in range(4):
                target_plan.laplace_apply(plan, sx, sy, ctx=ctx, out=out, **kw)
                ctx.sync()
                ks.append(ctx.last_kernel_ms())
            ref = out.clone()
            kf = []
            for _ in range(4):
                target_plan.laplace_apply(planf, sx, sy, ctx=ctx, out=outf, far=True, **kw)
                ctx.sync()
                kf.append(ctx.last_kernel_ms())
            ctx.enable_timing(False)
            kms_ = float(np.mean(ks[1:]))
            pairs = float(NBDY) * float(dt.N - plan.nrest)
            layers[name] = {"kernel": "laplace_patch_kernel<%s>" % name.upper(), "kernel_ms": kms_,
                            "pairs_per_s": pairs / (kms_ * 1e-3), "flops_per_pair_algorithmic": fl,
                            "frac_of_fp64_vector_peak": pairs * fl / (kms_ * 1e-3) / 1e12 / PEAK_FP64_VECTOR_TFLOPS,
                            "far_form_kernels_ms": float(np.mean(kf[1:])),
                            "far_form_max_abs_diff": float((outf - ref).abs().max()),
                            "max_abs": float(ref.abs().max())}
        # `out` again from the timed route (parity check below)
        target_plan.laplace_apply(plan, sx, sy, w_sigma=sig_full * w, ctx=ctx, out=out)
        ctx.sync()
        del dtf, planf, outf

    def allreduce_max(x):
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    if use_dist:
        elapsed = allreduce_max(elapsed)

    # strong: the whole job evaluates the one target list once per step
    pairs_per_step = float(NBDY) * (float(trg.N) if strong else float(dt.N) * world)
    # cost of the step's exchange alone (density all-gather + the fence's barrier), untimed
    collective_ms = None
    if use_dist:
        fence()
        t0 = time.perf_counter()
        for _ in range(50):
            gather_density()
        torch.cuda.synchronize()
        collective_ms = 1e3 * (time.perf_counter() - t0) / 50
    value = pairs_per_step * args.steps / elapsed
    # parity spot check inside the bench run (not timed): on EVERY rank 2048 targets of its own
    # slice against the C oracle; the line reports the worst rank
    import oracle
    idx = np.random.default_rng(1 + rank).choice(dt.N, min(2048, dt.N), replace=False)
    ref = oracle.c_laplace_apply(c.x, c.y, trg.x[sl][idx], trg.y[sl][idx], w_sigma=sigma * c.weights)
    got = out.cpu().numpy()[idx]
    parity = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    if use_dist:
        parity = allreduce_max(parity)
    result = None
    if rank == 0:
        # pairs the timed kernel evaluates FOR targets (the unstored points of cut tiles do not count;
        # a plan's remainder, if any, is a second, small launch of the list kernel)
        kpairs = float(NBDY) * float(dt.N - (plan.nrest if plan is not None else 0)) / (kernel_ms_avg * 1e-3)
        vipp = VALU_INSTR_PER_PAIR_PATCH if plan is not None else VALU_INSTR_PER_PAIR
        # list kernel: x, y in, u out per target; patch kernel: 8 coordinates + 16 positions in, 16 values out per patch
        algo_bytes = float(ALGO_BYTES_PER_PATCH * plan.np if plan is not None else ALGO_BYTES_PER_TARGET * dt.N)
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside
        # this process; the number is the rocprofv3 FETCH_SIZE / WRITE_SIZE measurement of this
        # same command (tools/collect_traffic.py), stamped with the commit it was taken at
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tf) and world == 1:
            try:
                tj = json.load(open(tf))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = {k: tj.get(k) for k in ("kernel", "measured_at_commit", "measured_on", "command")}
            except Exception:
                traffic = None
        result = {
            "metric": "source-target pair-interactions/s (Laplace SLP, fp64)",
            "value": value, "unit": "pair-interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "Laplace SLP grid_evaluator, 2048^2 grid x 4096-node star boundary "
                            "(BASELINE configs[1]); %d targets after removing the grid points within "
                            "7.5 h of the curve (tests/util.py grid_targets, clearance 5 x 1.5 h)" % trg.N,
                "n_sources": NBDY, "n_targets_total": int(trg.N if strong else dt.N * world),
                "n_targets_per_gpu": int(dt.N),
                "parallelism": ("one target list split into %d contiguous slice(s), " % world if strong
                                else "one full grid per rank (%d rank(s)), " % world)
                               + (("density all-gather over gloo (shared-GPU REHEARSAL, not a measurement)"
                                   if rehearse else "density all-gather over RCCL each step") if use_dist
                                  else "single process, no collective"),
                "kernel_variant": args.variant,
                "target_patches": None if plan is None else
                    {"patches": plan.np, "targets_in_patches": int(dt.N - plan.nrest),
                     "unstored_points": int(16 * plan.np - (dt.N - plan.nrest)), "remainder": plan.nrest},
            },
            "collective_ms_per_step": collective_ms,
            "parity_max_rel_err_vs_oracle": parity,
            "roofline": {
                "bound": "fp64 VALU (vector, non-MFMA): the dense sum is compute bound by "
                         "~500x over HBM (SURVEY §8d)",
                "achieved": kpairs * FLOPS_PER_PAIR_ALGO / 1e12,
                "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": kpairs * FLOPS_PER_PAIR_ALGO / 1e12 / PEAK_FP64_VECTOR_TFLOPS,
                "flops_per_pair_algorithmic": FLOPS_PER_PAIR_ALGO,
                "kernel": "laplace_patch_kernel<SLP>" if plan is not None else "laplace_rowrun_kernel<SLP, 4>",
                "frac_of_valu_issue_rate": kpairs * vipp / PEAK_VALU_LANE_INSTR,
                "valu_instr_per_pair": vipp,
                "kernel_ms": kernel_ms_avg,
                "kernel_ms_source": "hipEvent pairs recorded by the library around the kernel on its launch "
                                    "stream in each of the %d timed steps (mean; min %.4f, max %.4f)"
                                    % (len(kms_timed), min(kms_timed), max(kms_timed)),
                "kernel_ms_isolated": kernel_ms_isolated,
                "kernel_pairs_per_s": kpairs,
                "list_kernel": None if list_kernel_ms is None else
                    {"kernel": "laplace_rowrun_kernel<SLP, 4> (ipde_laplace_apply, any target list)",
                     "kernel_ms": list_kernel_ms,
                     "pairs_per_s": float(NBDY) * float(dt.N) / (list_kernel_ms * 1e-3),
                     "valu_instr_per_pair": VALU_INSTR_PER_PAIR},
                "hbm": {
                    "achieved": algo_bytes / (kernel_ms_avg * 1e-3) / 1e9,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": algo_bytes / (kernel_ms_avg * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_launch": algo_bytes,
                },
                "traffic": traffic,
                "traffic_source": traffic_src,
            },
        }
        result["expansion_form"] = expansion
        result["other_layers"] = layers
        if not args.no_fft and world == 1:
            result["fft"] = fft_block()
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(c, trg, sigma)
        else:
            result["cpu_baseline"] = None
        # second half of BASELINE.json's metric ("+ full Poisson solve wall-time, 2048^2
        # grid", configs[2]): measured after, and outside, the timed region above
        if not args.no_full_solve and world == 1:
            result["full_poisson_solve"] = full_poisson_solve()
        if not args.no_configs and not args.no_full_solve and world == 1:
            result["baseline_configs"] = baseline_configs()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
