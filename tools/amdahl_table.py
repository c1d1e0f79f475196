#!/usr/bin/env python3
"""DESIGN §4's table: the kernels of a warm solve (profiles/r04_*_solve_budget.json, tools/analyze_trace.py) sorted
into what the N-rank path does with them — sharded over targets (the dense sums in their far-field forms), owner-only
(per-boundary work: annular GMRES, QFS substitutions, jumps; every rank runs it when there is one boundary), replicated
(grid solve, interface interpolation, radial -> grid, PCIe) — and the ceiling that leaves at N = 2, 4, 8.

    python tools/amdahl_table.py profiles/r04_poisson_2048_resident_solve_budget.json 1 [...]
(second argument: number of boundaries; with several, the owner-only share of the largest boundary stays serial:
its node fraction is the third argument, default 1/boundaries)."""
import json
import re
import sys

SHARDED = re.compile(r"laplace_patch|laplace_far|_cols_far|modhelm_patch|modhelm_far|stokes_patch|stokes_far|"
                     r"columns_as_patches|laplace_rowrun|modhelm_table|stokes_rowrun|patch_reduce")
OWNER = re.compile(r"fft_pair|mixc|mix_multi|multidot|multiaxpy|prec_|gmres|cscale|lu_subst|gemv_rows|jump|rotate|"
                   r"fft_rtc|r2c_copy|splat|desplat|stokes_finish|bc2|pressure_mean|annular|transpose|real_imag")


def main():
    args = sys.argv[1:]
    rows = []
    while args:
        f = args.pop(0)
        nb = int(args.pop(0)) if args and args[0].isdigit() else 1
        frac = 1.0 / nb
        if args and re.fullmatch(r"0?\.\d+", args[0]):
            frac = float(args.pop(0))
        d = json.load(open(f))
        k = d.get("kernels_ms_per_solve") or d.get("kernel_ms_per_solve")
        tot = {"sharded": 0.0, "owner": 0.0, "replicated": 0.0}
        for name, v in k.items():
            ms = v["ms"] if isinstance(v, dict) else v
            cls = "sharded" if SHARDED.search(name) else "owner" if OWNER.search(name) else "replicated"
            tot[cls] += ms
        busy = sum(tot.values())
        idle = d["warm_solve_ms"] - d["gpu_busy_ms_per_solve"]
        line = {"budget": f.split("/")[-1], "boundaries": nb, "warm_ms_under_trace": d["warm_solve_ms"],
                "gpu_busy_ms": busy, "host_gaps_ms": idle, **{k_: round(v, 3) for k_, v in tot.items()}}
        for N in (2, 4, 8):
            owner_serial = tot["owner"] if nb == 1 else tot["owner"] * max(frac, 1.0 / min(N, nb))
            t = tot["replicated"] + tot["sharded"] / N + owner_serial + idle
            line["N=%d" % N] = round(t, 2)
            line["speedup_N=%d" % N] = round(d["warm_solve_ms"] / t, 2)
        rows.append(line)
    print(json.dumps(rows, indent=1))
    print()
    print("| solve (one GPU, under the kernel trace) | warm ms | sharded | owner-only | replicated | host gaps | "
          "ceiling N=2 | N=4 | N=8 |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s (%d boundar%s) | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f ms (%.2fx) | %.2f (%.2fx) | %.2f (%.2fx) |"
              % (r["budget"].replace("_solve_budget.json", "").replace("r04_", ""), r["boundaries"],
                 "y" if r["boundaries"] == 1 else "ies", r["warm_ms_under_trace"], r["sharded"], r["owner"],
                 r["replicated"], r["host_gaps_ms"], r["N=2"], r["speedup_N=2"], r["N=4"], r["speedup_N=4"],
                 r["N=8"], r["speedup_N=8"]))


if __name__ == "__main__":
    main()
