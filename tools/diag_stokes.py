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
    ebdyc, (uc, vc, pc), (ua, va, pa), pd = multi_stokes.run(nb=nb, M=M, return_fields=True, ng=ng)
    du, dv = uc - ua, vc - va
    out = {"nb": nb, "M": M, "grid": list(ebdyc.grid.shape)}
    out["grid_err"] = [float(np.abs(du['grid']).max()), float(np.abs(dv['grid']).max())]
    out["radial_err"] = [[float(np.abs(du[i]).max()), float(np.abs(dv[i]).max())] for i in range(len(ebdyc))]
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
