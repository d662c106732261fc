#!/usr/bin/env python3
"""Fold the rocprofv3 PMC passes into profiles/<name>_pmc_traffic.json.

Usage (on the GPU box, one pass per counter, with --kernel-trace only):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o runc --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o runc --output-format csv -- python3 bench.py ...
    python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json --docs 1000000 ...

FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.  The per-launch figure of
a kernel is the mean over its full-size dispatches (largest grid of that kernel; the warm-up launches have the same size).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def per_kernel(directory, counter):
    rows = defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                m = re.search(r"\b(k_\w+(?:<[\w, ]*>)?)", r["Kernel_Name"])
                if not m:
                    continue
                name = m.group(1)
                rows[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for name, v in rows.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        out[name] = {"launches": len(vals), "mean_KiB": sum(vals) / len(vals)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--terms", type=int, default=10_000)
    ap.add_argument("--exprs", type=int, default=1_000)
    ap.add_argument("--inord", type=float, default=0.0)
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    f = per_kernel(a.fetch_dir, "FETCH_SIZE")
    w = per_kernel(a.write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        fk = f.get(name, {"mean_KiB": 0.0, "launches": 0})
        wk = w.get(name, {"mean_KiB": 0.0, "launches": 0})
        rd = 2.0 * fk["mean_KiB"] * 1024
        wr = wk["mean_KiB"] * 1024
        kernels[name] = {"launches": fk["launches"], "FETCH_SIZE_KiB": fk["mean_KiB"], "WRITE_SIZE_KiB": wk["mean_KiB"],
                         "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes": rd + wr}
    doc = {
        "source": a.source or "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), MI355X",
        "config": {"docs": a.docs, "terms": a.terms, "exprs": a.exprs, "inord": a.inord},
        "note": "KiB per dispatch, mean over the full-size launches of each kernel; read bytes = 2 * FETCH_SIZE * 1024 "
                "(gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE exact.",
        "kernels": kernels,
    }
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in kernels.items():
        print("%-40s launches %3d  read %.3f GB  write %.3f GB" % (k, v["launches"], v["read_bytes_corrected"] / 1e9, v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()
