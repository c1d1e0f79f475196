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
    try:
        ebdyc, (uc, vc, pc), (ua, va, pa), pd = multi_stokes.run(nb=nb, M=M, return_fields=True, ng=ng,
                                                                          tol=float(os.environ.get('IPDE_DIAG_TOL', '1e-12')))
    finally:
        multi_stokes.StokesSolver.__call__ = orig
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
    out["gmres_iterations"] = list(kept["solver"].iteration_counts)
    out["max_sigma_g"] = [amax(h.sigma_g) for h in kept["solver"].helpers]
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
