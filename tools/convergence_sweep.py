#!/usr/bin/env python3
"""Manufactured-solution errors of the three example flows over a range of resolutions (one JSON
line per run): the plateau the reference records for its scripts (poisson_for_paper.py:118-124:
1e-13 .. 4e-13 relative) is the acceptance level of the whole pipeline."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    import interior_poisson
    import interior_modified_helmholtz as imh
    import multi_modified_helmholtz as mmh
    import multi_stokes

    def emit(**kw):
        print(json.dumps(kw), flush=True)
    for nb in (500, 700, 1000, 1500, 2200):
        t0 = time.perf_counter()
        err, scale, solver, ue, T = interior_poisson.run(nb=nb, M=16)
        emit(problem="interior_poisson", nb=nb, M=16, grid=T["grid"], dof=T["dof"], rel_err=err / scale,
             gmres=T["gmres_iterations"], wall_s=time.perf_counter() - t0)
        del solver, ue
    for k in (1.0, 10.0, 40.0):
        for nb in (600, 1000, 1600):
            t0 = time.perf_counter()
            err, scale, solver, ue, T = imh.run(nb=nb, M=16, helmholtz_k=k)
            emit(problem="interior_modified_helmholtz", k=k, nb=nb, M=16, grid=T["grid"], rel_err=err / scale,
                 gmres=T["gmres_iterations"], wall_s=time.perf_counter() - t0)
            del solver, ue
    for nb in (400, 600, 900):
        t0 = time.perf_counter()
        err, scale, T = mmh.run(nb=nb, M=16, helmholtz_k=2.0)
        emit(problem="multi_modified_helmholtz (outer boundary + two holes)", nb=nb, M=16, rel_err=err / scale,
             wall_s=time.perf_counter() - t0)
    for nb in (600, 800, 1100, 1500):
        t0 = time.perf_counter()
        ue, ve, pe, scale, T = multi_stokes.run(nb=nb, M=14)
        emit(problem="multi_stokes (3 bodies)", nb=nb, M=14, grid=T["grid"], dof=T["dof"],
             rel_err_u=ue / scale, rel_err_v=ve / scale, abs_err_p=pe, gmres=T["gmres_iterations"],
             wall_s=time.perf_counter() - t0)


if __name__ == "__main__":
    main()
