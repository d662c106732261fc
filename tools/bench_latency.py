#!/usr/bin/env python3
"""Per-call latency of the reference-shaped entry points (one document per call), host memory in, results out:
Finder.ProcessText and GpuEngine.FindSubstrings at a few document sizes.  Not part of the bench contract.

    python tools/bench_latency.py [--terms T] [--exprs E] [--reps N]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine  # noqa: E402
from gofindthem_amd.workload import Workload, make_expressions  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--terms", type=int, default=10000)
ap.add_argument("--exprs", type=int, default=1000)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--batch-docs", type=int, default=0, help="also time ProcessTexts on this many documents from host memory")
args = ap.parse_args()

w = Workload(args.terms)
exprs = make_expressions(w.terms(), args.exprs, cover=True)
f = Finder(GpuEngine(), EmptyRgxEngine(), False)
f.AddExpressions(exprs)
blob, off = w.docs_host(0, 260)
full = blob.tobytes()
one_doc = full[int(off[0]):int(off[1])]
out = {"terms": args.terms, "exprs": args.exprs, "ProcessText_us": {}, "FindSubstrings_us": {}}
for name, text in (("one ~4 KB document", one_doc), ("64 KB", full[:65536]), ("1 MB", full[:1 << 20])):
    f.ProcessText(text)
    f.ProcessText(text)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        f.ProcessText(text)
    out["ProcessText_us"][name] = (time.perf_counter() - t0) / args.reps * 1e6
eng = GpuEngine()
eng.BuildEngine([t.decode() for t in w.terms()], False)
for name, text in (("one ~4 KB document", one_doc), ("1 MB", full[:1 << 20])):
    s = text.decode()
    eng.FindSubstrings(s)
    t0 = time.perf_counter()
    for _ in range(max(args.reps // 4, 5)):
        m = eng.FindSubstrings(s)
    out["FindSubstrings_us"][name] = (time.perf_counter() - t0) / max(args.reps // 4, 5) * 1e6
    out.setdefault("FindSubstrings_matches", {})[name] = len(m)
if args.batch_docs:
    bb, bo = w.docs_host(0, args.batch_docs)
    for _ in range(3):                       # (the first calls size the device buffers and the match pool)
        f.ProcessTexts(blob=bb, doc_off=bo)
    t0 = time.perf_counter()
    for _ in range(5):
        f.ProcessTexts(blob=bb, doc_off=bo)
    dt = (time.perf_counter() - t0) / 5
    out["ProcessTexts_host_memory"] = {"docs": args.batch_docs, "docs_per_s": args.batch_docs / dt, "text_GB_per_s": int(bo[-1]) / dt / 1e9}
    # the same with the result rows going into the caller's own array, batch after batch (what a Go caller's []uint32 is)
    rows = f.ProcessTexts(blob=bb, doc_off=bo)
    t0 = time.perf_counter()
    for _ in range(5):
        f.ProcessTexts(blob=bb, doc_off=bo, out=rows)
    dt = (time.perf_counter() - t0) / 5
    out["ProcessTexts_host_memory_reused_rows"] = {"docs": args.batch_docs, "docs_per_s": args.batch_docs / dt, "text_GB_per_s": int(bo[-1]) / dt / 1e9}
print(json.dumps(out))
