#!/usr/bin/env python3
"""Mean per-launch values of the counters in a rocprofv3 --pmc run, per kernel (largest grid of each kernel only).
    python3 tools/sq_summary.py <rocprof output dir> [--docs N]"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--docs", type=int, default=0)
a = ap.parse_args()
rows = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"\b(k_\w+(?:<[\w, ]*>)?)", r["Kernel_Name"])
            if m:
                rows[m.group(1)][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
out = {}
for k, cs in rows.items():
    out[k] = {}
    for c, v in cs.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        out[k][c] = sum(vals) / len(vals) / (a.docs or 1)
    out[k]["launches"] = len(vals)
print(json.dumps(out, indent=1, sort_keys=True))
