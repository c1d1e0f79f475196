#!/usr/bin/env python3
"""Where does the 3-body Stokes error sit, and how does it move with n_b and M?"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import multi_stokes


def one(nb, M, ng=None):
    kept = {}
    orig = multi_stokes.StokesSolver.__call__

    def keeping(self, fu, fv, **kw):
        kept["solver"] = self
        return orig(self, fu, fv, **kw)
    multi_stokes.StokesSolver.__call__ = keeping
    # the pressure calibration of the interior-side QFS objects: the pinned point and the multiple of the null density
    # (source normals) every call adds
    from ipde_amd import qfs as _qfs
    post = _qfs.Stokes_QFS._post
    shifts = []

    keep = float(os.environ.get("IPDE_DIAG_LOWPASS", "0"))      # experiment: modes above keep x Nyquist removed

    def post_logged(self, mu, densities):
        out_ = post(self, mu, densities)
        if keep > 0:
            import torch
            dev = hasattr(out_, "cpu")
            x = (out_ if dev else torch.as_tensor(np.asarray(out_))).reshape(2, -1)
            h = torch.fft.rfft(x, dim=1)
            h[:, int(keep * (h.shape[1] - 1)):] = 0
            x = torch.fft.irfft(h, n=x.shape[1], dim=1).reshape(-1)
            out_ = x if dev else x.numpy()
        if self.interior:
            d = out_ - mu
            a = float(abs(d).max()) if not hasattr(d, "cpu") else float(d.abs().max())
            m = float(abs(mu).max()) if not hasattr(mu, "cpu") else float(mu.abs().max())
            shifts.append((int(self.bdy.N), tuple(round(v, 4) for v in self._p_point), round(a, 4), round(m, 4)))
        return out_
    _qfs.Stokes_QFS._post = post_logged
    try:
        ebdyc, (uc, vc, pc), (ua, va, pa), pd = multi_stokes.run(nb=nb, M=M, return_fields=True, ng=ng,
                                                                          tol=float(os.environ.get('IPDE_DIAG_TOL', '1e-12')))
    finally:
        multi_stokes.StokesSolver.__call__ = orig
        _qfs.Stokes_QFS._post = post
    du, dv = uc - ua, vc - va
    out = {"nb": nb, "M": M, "grid": list(ebdyc.grid.shape)}
    out["grid_err"] = [float(np.abs(du['grid']).max()), float(np.abs(dv['grid']).max())]
    # the grid error apart: points inside the annuli (values interpolated from the radial grids) and outside them
    # (grid solve + layer sums)
    ia = ebdyc.in_annulus[ebdyc.phys]
    out["grid_err_in_annuli"] = float(max(np.abs(du['grid'][ia]).max(), np.abs(dv['grid'][ia]).max()))
    out["grid_err_outside_annuli"] = float(max(np.abs(du['grid'][~ia]).max(), np.abs(dv['grid'][~ia]).max()))
    for i, e in enumerate(ebdyc):      # ... and annulus by annulus
        m = np.zeros(ebdyc.grid.shape, dtype=bool)
        m[e.grid_ia_xind, e.grid_ia_yind] = True
        mi = m[ebdyc.phys]
        out["grid_err_annulus_%d" % i] = float(max(np.abs(du['grid'][mi]).max(), np.abs(dv['grid'][mi]).max()))
    out["radial_err"] = [[float(np.abs(du[i]).max()), float(np.abs(dv[i]).max())] for i in range(len(ebdyc))]
    # size of the QFS source densities of the inhomogeneous solve (grid side, annulus side) per boundary
    def amax(t):
        return float(abs(t).max()) if not hasattr(t, "cpu") else float(t.abs().max())
    out["pressure_calibration (N, point, max shift, max mu before)"] = shifts[:12]
    # where the largest grid error outside the annuli sits, and how far from each boundary's nodes
    g = np.where(~ia, np.maximum(np.abs(du['grid']), np.abs(dv['grid'])), 0.0)
    j = int(np.argmax(g))
    gx, gy = ebdyc.grid.xg[ebdyc.phys][j], ebdyc.grid.yg[ebdyc.phys][j]
    out["worst_outside_point"] = [float(gx), float(gy)]
    out["worst_outside_point_distance_to_boundaries"] = [float(np.hypot(e.bdy.x - gx, e.bdy.y - gy).min()) for e in ebdyc]
    out["gmres_iterations"] = list(kept["solver"].iteration_counts)
    out["max_sigma_g"] = [amax(h.sigma_g) for h in kept["solver"].helpers]
    # spectrum of the outer boundary's grid-side density (x components): band maxima of |sigma_hat_k| / N and the peaks
    sg = kept["solver"].helpers[0].sigma_g
    sg = sg.cpu().numpy() if hasattr(sg, "cpu") else np.asarray(sg)
    sx = sg.reshape(2, -1)[0]
    sh = np.abs(np.fft.rfft(sx)) / sx.size
    edges = sorted(set(min(e, sh.size) for e in (0, 10, 100, 500, 1000, 2000, 3000, 4000, sh.size)))
    out["sigma_g0_spectrum_band_max"] = [float(sh[a:b].max()) for a, b in zip(edges[:-1], edges[1:])]
    top = np.argsort(sh)[::-1][:6]
    out["sigma_g0_spectrum_peaks"] = [(int(k), float(sh[k])) for k in top]
    out["max_sigma_r"] = [amax(h.sigma_r) for h in kept["solver"].helpers]
    # radial profile of the error in the worst annulus (rows = radial nodes, boundary first)
    i = int(np.argmax([max(r) for r in out["radial_err"]]))
    out["worst_annulus_profile"] = [float(x) for x in np.abs(du[i]).max(axis=1)]
    print(out, flush=True)


if __name__ == "__main__":
    cases = [(2390, 14, None), (2400, 14, 4096), (2400, 14, None), (2392, 14, None)]
    if len(sys.argv) > 1:
        cases = [tuple(int(v) if v != "-" else None for v in a.split(",")) for a in sys.argv[1:]]
    for nb, M, ng in cases:
        one(nb, M, ng)
