#!/usr/bin/env python3
"""Measurement of the group-finder row: a batch of JSON documents through GroupFinder.ProcessJsons
(host: JSON decode + object walk + tag-rule DSL; GPU: every string leaf scanned and solved in one batch), next to the CPU
oracle doing the finder part of the same work (one ProcessText per leaf, 16 threads).  Not part of the bench.py contract.

    python tools/bench_group.py [--docs N] [--terms T] [--exprs E]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gofindthem_amd import _lib, group  # noqa: E402
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine  # noqa: E402
from gofindthem_amd.workload import Workload, make_expressions  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=50000)
ap.add_argument("--terms", type=int, default=10000)
ap.add_argument("--exprs", type=int, default=1000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cpu-threads", type=int, default=16)
args = ap.parse_args()

w = Workload(args.terms)
exprs = make_expressions(w.terms(), args.exprs, inord_fraction=0.0, cover=True)
tags = ["tag%d" % (i % 50) for i in range(len(exprs))]
f = Finder(GpuEngine(), EmptyRgxEngine(), False)
for e, t in zip(exprs, tags):
    f.AddExpressionWithTag(e, t)
rules = {"rule%d" % i: ['"tag%d" and not "tag%d:Body"' % (i, (i + 7) % 50), '"tag%d:Meta" or "tag%d:Comments"' % ((i + 3) % 50, i)]
         for i in range(50)}
g = group.NewFinderWithRules(f, rules)

# documents: ~4 KB of text each, spread over 8 string leaves in a nested object
text, off = w.docs_host(0, args.docs)
raws = []
for d in range(args.docs):
    t = bytes(text[int(off[d]):int(off[d + 1])]).decode("ascii")
    n = len(t) // 8
    p = [t[i * n:(i + 1) * n] for i in range(8)]
    raws.append(json.dumps({"Id": d, "Title": p[0], "Body": [p[1], p[2], p[3]], "Meta": {"Author": p[4], "Notes": [p[5]]},
                            "Comments": [{"Text": p[6], "Score": 3}, {"Text": p[7], "Score": 5}]}))
L = _lib.load()
eh = f.engine_handle()
res = g.ProcessJsons(raws[:100])        # warm-up (engine build, program upload)
from gofindthem_amd.engine import pack  # noqa: E402
best, kern, parts = 1e9, None, None
for _ in range(args.reps):
    L.gft_profile_enable(eh, 1)
    L.gft_profile_reset(eh)
    t0 = time.perf_counter()
    blob, boff = pack([r.encode() for r in raws])
    t1 = time.perf_counter()
    need = C.c_uint64(0)
    cap = 2 * int(blob.size)
    buf = C.create_string_buffer(cap)
    t2 = time.perf_counter()
    rc = L.gft_group_process_jsons(g._h, blob.ctypes.data, boff.ctypes.data, len(raws), None, 0, None, 0, 0,
                                   C.cast(buf, C.c_void_p), cap, C.byref(need))
    assert rc == 0, L.gft_group_last_error(g._h)
    t3 = time.perf_counter()
    res = json.loads(buf.value.decode())
    t4 = time.perf_counter()
    ms = {}
    for name in (b"scan", b"solve", b"aux"):
        a, n = C.c_double(), C.c_uint64()
        L.gft_profile_read(eh, name, C.byref(a), C.byref(n))
        ms[name.decode()] = a.value
    L.gft_profile_enable(eh, 0)
    if t3 - t2 < best:
        best, kern = t3 - t2, ms
        parts = {"python_pack_s": t1 - t0, "library_call_s": t3 - t2, "python_json_loads_of_result_s": t4 - t3,
                 "result_MB": need.value / 1e6}
leaves, nbytes = g.last_batch()
hits = sum(len(r["rules"]) for r in res)

# CPU: the finder part of the same work on the oracle (leaves as documents), bitmap compared
from oracle.pyoracle import Oracle, pack_strings  # noqa: E402
o = Oracle(sorted(f.GetKeywords()))
o.set_expressions(exprs, False)
n_cpu = min(args.docs, 4000)
leaf_texts = []
for raw in raws[:n_cpu]:
    d = json.loads(raw)
    leaf_texts += [d["Title"]] + d["Body"] + [d["Meta"]["Author"]] + d["Meta"]["Notes"] + [c["Text"] for c in d["Comments"]]
blob, loff = pack_strings([s.encode() for s in leaf_texts])
t0 = time.perf_counter()
bm = o.process(blob, loff, fold=True, n_threads=args.cpu_threads)
cpu_dt = time.perf_counter() - t0
gpu_bm = f.ProcessTexts(leaf_texts)
print(json.dumps({
    "row": "SURVEY 8(f) #2 GroupFinder.ProcessJsons", "docs": args.docs, "leaves": leaves, "leaf_bytes": nbytes,
    "json_bytes": sum(len(r) for r in raws), "rules": len(rules) * 2, "finder_expressions": len(exprs),
    "rule_hits": hits, "library_call_s": best, "docs_per_s": args.docs / best, "leaves_per_s": leaves / best,
    "json_MBps": sum(len(r) for r in raws) / best / 1e6, "python_side": parts,
    "gpu_kernels_ms": kern, "host_share": 1.0 - sum(kern.values()) / 1e3 / best,
    "cpu_finder_only": {"leaves_per_s": len(leaf_texts) / cpu_dt, "threads": args.cpu_threads, "sample_leaves": len(leaf_texts),
                        "bitmap_equal_to_gpu": bool(np.array_equal(bm, gpu_bm))}}))
