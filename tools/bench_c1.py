#!/usr/bin/env python3
"""BASELINE configs[0] -- the reference's own benchmark shape (benchmarks/benchmark_test.go:66,416-426 BMDslSearch): ONE
document of ~100 000 words (~1 MB), ~50 terms, 3 expressions (a plain AND, two INORDs in both orders, an INORD chain
of 45 terms), case-insensitive finder.  Times Finder.ProcessText from host memory (H2D included: the call the
reference's benchmark loop makes) and the CPU restatement on one thread beside it.  Not part of the bench contract.

    python tools/bench_c1.py [--reps N]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine  # noqa: E402
from gofindthem_amd.workload import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--cpu-reps", type=int, default=5)
args = ap.parse_args()

w = Workload(50)
t = [x.decode() for x in w.terms()]
exprs = ['"%s" and "%s"' % (t[0], t[1]),
         'INORD("%s" and "%s") and INORD("%s" and "%s")' % (t[2], t[3], t[3], t[2]),
         "INORD(" + " and ".join('"%s"' % x for x in t[4:49]) + ")"]
blob, off = w.docs_host(0, 250)
text = blob.tobytes()
f = Finder(GpuEngine(), EmptyRgxEngine(), False)
f.AddExpressions(exprs)
got = [r.ExpresionIndex for r in f.ProcessText(text)]          # builds + warms
f.ProcessText(text)
t0 = time.perf_counter()
for _ in range(args.reps):
    f.ProcessText(text)
gpu_ms = (time.perf_counter() - t0) / args.reps * 1e3

import ctypes as C  # noqa: E402

from gofindthem_amd import _lib  # noqa: E402
L, eh = _lib.load(), f.engine_handle()
L.gft_profile_enable(eh, 1)
L.gft_profile_reset(eh)
for _ in range(5):
    f.ProcessText(text)
kern = {}
for name in (b"scan", b"solve", b"aux"):
    ms, n = C.c_double(), C.c_uint64()
    L.gft_profile_read(eh, name, C.byref(ms), C.byref(n))
    kern[name.decode()] = ms.value / 5
L.gft_profile_enable(eh, 0)

cpu_ms, same = None, None
if args.cpu_reps:
    from oracle.pyoracle import Oracle, POS_START              # the checker, timed beside the product (never inside it)
    o = Oracle(w.terms(), POS_START)
    o.set_expressions(exprs, case_sensitive=False)
    one = np.array([0, len(text)], dtype=np.uint64)
    want = o.process(blob, one, fold=True)
    t0 = time.perf_counter()
    for _ in range(args.cpu_reps):
        o.process(blob, one, fold=True)
    cpu_ms = (time.perf_counter() - t0) / args.cpu_reps * 1e3
    same = [i for i in range(len(exprs)) if int(want[0, 0]) >> i & 1] == got
print(json.dumps({"config": "BASELINE configs[0]: 1 document of %d bytes, %d terms, %d expressions" % (len(text), len(t), len(exprs)),
                  "ProcessText_ms": gpu_ms, "MB_per_s": len(text) / gpu_ms / 1e3, "true_expressions": got, "kernels_ms_per_call": kern,
                  "cpu_restatement_ms_1_thread": cpu_ms, "identical_to_cpu": same}))
