#!/usr/bin/env python3
"""Per-solve GPU time budget from a rocprofv3 kernel trace of tools/profile_solve.py (or
profile_stokes_solve.py): the trace's tail holds `nsolves` identical warm solves; report GPU
busy time, idle time and the kernels by total duration, per solve.

    python3 tools/analyze_trace.py <dir with *_kernel_trace.csv> <warm solve ms> [nsolves=10]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, solve_ms = sys.argv[1], float(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    # copies the trace saw (--memory-copy-trace), if any: busy intervals too, named by direction
    for fc in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fc)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?")))
    rows.sort()
    t_end = rows[-1][1]
    t0 = t_end - int(n * solve_ms * 1e6)
    win = [r for r in rows if r[0] >= t0]
    busy, last, prev = 0, t0, "(window start)"
    gaps = collections.defaultdict(lambda: [0, 0])
    for s0, e, k in win:          # union of intervals (concurrent streams overlap)
        if s0 - last >= 3000:     # idle stretches of 3 us and more, by the kernels either side
            g = gaps[prev[:48] + "  ->  " + k[:48]]
            g[0] += s0 - last
            g[1] += 1
        s = max(s0, last)
        if e > s:
            busy += e - s
            last = e
            prev = k
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, k in win:
        agg[k[:90]][0] += e - s
        agg[k[:90]][1] += 1
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]
    out = {"warm_solve_ms": solve_ms, "solves_in_window": n, "launches_per_solve": len(win) / n,
           "gpu_busy_ms_per_solve": busy / n / 1e6, "gpu_busy_fraction": busy / (n * solve_ms * 1e6),
           "idle_ms_per_solve_by_neighbours": {k: {"ms": v[0] / n / 1e6, "count": v[1] / n} for k, v in
                                               sorted(gaps.items(), key=lambda kv: -kv[1][0])[:30]},
           "kernels_ms_per_solve": {k: {"ms": v[0] / n / 1e6, "launches": v[1] / n} for k, v in top}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
