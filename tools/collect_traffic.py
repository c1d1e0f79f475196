#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the
MI355X guide prescribes) into profiles/traffic_latest.json, which bench.py reports
as roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write laplace_table_kernel r01

gfx950 corrections (MI355X_MICROARCH.md §HBM): counter values are KiB; FETCH_SIZE
counts 128-B requests as 64 B for coalesced streaming reads, so it is doubled;
WRITE_SIZE is exact for streaming stores.
"""
import csv
import glob
import json
import os
import subprocess
import sys
import time


def collect(path, kernel, counter):
    vals = []
    for f in glob.glob(os.path.join(path, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


def _commit():
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        return None


def main():
    fetch_dir, write_dir, kernel, tag = sys.argv[1:5]
    fe = collect(fetch_dir, kernel, "FETCH_SIZE")
    wr = collect(write_dir, kernel, "WRITE_SIZE")
    fe_b = 1024.0 * sum(fe) / len(fe)
    wr_b = 1024.0 * sum(wr) / len(wr)
    out = {
        "kernel": kernel,
        "launches_sampled": [len(fe), len(wr)],
        "FETCH_SIZE_bytes_raw": fe_b,
        "WRITE_SIZE_bytes": wr_b,
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": 2.0 * fe_b + wr_b,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)",
        "measured_at_commit": _commit(),
        "measured_on": time.strftime("%Y-%m-%d"),
        "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py --steps 5 "
                   "--warmup 1 --no-cpu-baseline --no-full-solve --no-fft",
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("traffic_%s.json" % tag, "traffic_latest.json"):
        json.dump(out, open(os.path.join(root, "profiles", name), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
