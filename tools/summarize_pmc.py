#!/usr/bin/env python3
"""rocprofv3 --pmc output directory -> JSON of per-kernel average counter values.

    python3 tools/summarize_pmc.py <dir> <out.json> [kernel-substring ...]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1:3]
    keep = sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if keep and not any(k in name for k in keep):
                continue
            agg[name[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
    for k, cs in res.items():
        cs["launches_sampled"] = len(next(iter(agg[k].values())))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
