#!/usr/bin/env python3
"""The two rocprofv3 PMC passes over tools/profile_fft.py (FETCH_SIZE, WRITE_SIZE; separate runs) ->
HBM-side bytes per launch of every kernel of the 2-D FFT pipeline, and the sum for one Poisson grid
solve (row_r2c + col<POISSON> + row_c2r), which bench.py reports as fft.traffic.

    python3 tools/fft_traffic.py <fetch_dir> <write_dir> <tag> [n]

gfx950 corrections (MI355X_MICROARCH.md, HBM): counter values are KiB; FETCH_SIZE tallies a 128-byte
request of a wide coalesced read (16 B per lane) at 64 bytes and is doubled; WRITE_SIZE is exact for
16-byte-per-lane stores.  The column pass reads 64-byte row segments (four lanes of 16 B per row), a
width the guide calls uncalibrated: both the raw and the doubled figure are kept.  Infinity-Cache hits
are counted by these counters (the three fields of a 2048^2 solve, 100 MB, fit the 256 MiB cache)."""
import csv
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(path, counter):
    out = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("row_r2c" in r["Kernel_Name"] or "row_c2r" in r["Kernel_Name"]
                                                 or "col_kernel" in r["Kernel_Name"]):
                name = r["Kernel_Name"]
                name = name[:name.index(">") + 1].replace("void (anonymous namespace)::", "")
                out.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: 1024.0 * sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main():
    fetch_dir, write_dir, tag = sys.argv[1:4]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
    fe, nfe = collect(fetch_dir, "FETCH_SIZE")
    wr, nwr = collect(write_dir, "WRITE_SIZE")
    field = 8.0 * n * n
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        kernels[k] = {"FETCH_SIZE_bytes_raw": fe.get(k), "FETCH_SIZE_bytes_x2": None if k not in fe else 2 * fe[k],
                      "WRITE_SIZE_bytes": wr.get(k), "launches_sampled": [nfe.get(k, 0), nwr.get(k, 0)],
                      "algorithmic_bytes": 2 * field}
    def pick(sub):
        c = [k for k in kernels if sub in k]
        return c[0] if c else None
    solve = [pick("row_r2c_kernel<%d>" % n), pick("col_kernel<%d, 4, 0, 0>" % n) or pick("col_kernel<%d, 2, 0, 0>" % n),
             pick("row_c2r_kernel<%d>" % n)]
    total = None
    if all(solve):
        total = sum(kernels[k]["FETCH_SIZE_bytes_x2"] + kernels[k]["WRITE_SIZE_bytes"] for k in solve)
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    out = {"grid": [n, n], "kernels": kernels, "poisson_solve_kernels": solve,
           "poisson_solve_hbm_bytes": total, "poisson_solve_algorithmic_bytes": 16.0 * n * n,
           "poisson_solve_stated_bytes_3_passes": 6 * field,
           "measured_at_commit": commit, "measured_on": time.strftime("%Y-%m-%d"),
           "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 tools/profile_fft.py %d 6" % n}
    for name in ("%s_fft_traffic.json" % tag, "traffic_fft_latest.json"):
        json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
