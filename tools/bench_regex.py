#!/usr/bin/env python3
"""Measurement of the regex prefilter row: BASELINE configs[4] shape -- a large dictionary plus
r"wA.*wB" regex terms that stay on a host engine (here Python's `re` behind the RegexEngine callback interface, the
stand-in for Go's regexp).  Times Finder.ProcessTexts with the prefilter on and off; results must be identical.

    python tools/bench_regex.py [--docs N] [--terms T] [--regexes R]
"""
import argparse
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=20000)
ap.add_argument("--terms", type=int, default=10000)
ap.add_argument("--exprs", type=int, default=1000)
ap.add_argument("--regexes", type=int, default=16)
ap.add_argument("--mode", default="")
args = ap.parse_args()

if not args.mode:       # one process per mode: the switch is read when the finder is built
    out = {}
    for mode in ("1", "0"):
        env = dict(os.environ, GFT_REGEX_PREFILTER=mode)
        r = subprocess.run([sys.executable, __file__, "--mode", mode, "--docs", str(args.docs), "--terms", str(args.terms),
                            "--exprs", str(args.exprs), "--regexes", str(args.regexes)], env=env, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        out["prefilter_on" if mode == "1" else "prefilter_off"] = json.loads(r.stdout.strip().splitlines()[-1])
    same = out["prefilter_on"].pop("digest") == out["prefilter_off"].pop("digest")
    print(json.dumps({"row": "SURVEY 8(f) #3 regex prefilter", "docs": args.docs, "terms": args.terms, "regexes": args.regexes,
                      "identical_bitmaps": same, **out,
                      "speedup": out["prefilter_off"]["wall_s"] / out["prefilter_on"]["wall_s"]}))
    sys.exit(0 if same else 3)

import hashlib  # noqa: E402

import numpy as np  # noqa: E402

from gofindthem_amd import _lib  # noqa: E402
from gofindthem_amd.finder import Finder, GpuEngine, PyRegexpEngine  # noqa: E402
from gofindthem_amd.workload import Workload, make_expressions  # noqa: E402

w = Workload(args.terms)
terms = w.terms()
rng = np.random.default_rng(11)
rx = []
while len(rx) < args.regexes:          # r"wA.*wB" over dictionary words (benchmark_test.go:280 shape)
    a, b = terms[int(rng.integers(len(terms)))], terms[int(rng.integers(len(terms)))]
    a, b = (a.decode() if isinstance(a, bytes) else a), (b.decode() if isinstance(b, bytes) else b)
    if len(a) >= 5 and len(b) >= 5:
        rx.append("%s.*%s" % (a, b))
exprs = make_expressions(terms, args.exprs, inord_fraction=0.5, regexes=rx, cover=True)
f = Finder(GpuEngine(), PyRegexpEngine(), False)
f.AddExpressions(exprs)
text, off = w.docs_host(0, args.docs)
f.ProcessTexts(blob=text[:int(off[64]) + 1], doc_off=off[:65])          # warm-up: engine builds, program upload
t0 = time.perf_counter()
bm = f.ProcessTexts(blob=text, doc_off=off)
dt = time.perf_counter() - t0
print(json.dumps({"wall_s": dt, "docs_per_s": args.docs / dt, "host_regex_docs": int(_lib.load().gft_finder_last_regex_docs(f._h)),
                  "digest": hashlib.sha256(np.ascontiguousarray(bm).tobytes()).hexdigest()}))
