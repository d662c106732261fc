#!/usr/bin/env python3
"""What does ONE expression beyond the lane-resident INORD limits cost a batch?  
1 M documents resident in HBM, 1 000 ordinary AND/OR/NOT expressions, then the same set plus ONE INORD over an OR of 200
terms AND-ed with an OR of 200 others (400 (slot, threshold) pairs alive: 6 x the 64 a wave's lanes hold).  Round 3 solved
such an expression on the host for every document (CSR of the whole batch, every match over PCIe, one host thread); round 4
keeps its pairs in a per-wave scratch region on the device.  Prints docs/s for both sets and checks the wide expression's
column against the CPU oracle on a sample.
    python tools/bench_overlimit.py [--docs N] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gofindthem_amd import _lib  # noqa: E402
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine  # noqa: E402
from gofindthem_amd.workload import Workload, make_expressions  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--check-docs", type=int, default=20000)
ap.add_argument("--json", default=None)
args = ap.parse_args()

wl = Workload(10_000)
terms = wl.terms()
base = make_expressions(terms, 1000, inord_fraction=0.0, cover=True)
tl = [t.decode() for t in terms]
wide = "inord((%s) and (%s))" % (" or ".join('"%s"' % t for t in tl[:200]), " or ".join('"%s"' % t for t in tl[200:400]))
text, doc_off = wl.docs_device(0, args.docs, device=torch.device("cuda", 0))
L = _lib.load()
out = {"docs": args.docs, "steps": args.steps}
for name, exprs in (("ordinary_1000", base), ("ordinary_1000_plus_one_wide_inord", base + [wide])):
    f = Finder(GpuEngine.__new__(GpuEngine), EmptyRgxEngine(), caseSensitive=False, device=0)
    f.AddExpressions(exprs)
    f.ForceBuild()
    eh = f.engine_handle()
    L.gft_set_stream(eh, torch.cuda.current_stream().cuda_stream)
    words = (len(exprs) + 31) // 32
    bm = torch.zeros((args.docs, words), dtype=torch.int32, device="cuda")
    for _ in range(2):
        f.ProcessDevice(text.data_ptr(), doc_off.data_ptr(), args.docs, bm.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f.ProcessDevice(text.data_ptr(), doc_off.data_ptr(), args.docs, bm.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out[name] = {"ms_per_step": dt * 1e3, "docs_per_s": args.docs / dt, "host_solved_expressions": int(L.gft_n_host_exprs(eh))}
    if len(exprs) > 1000:
        from oracle.pyoracle import Oracle      # (the checker, behind the timed region)
        n = min(args.check_docs, args.docs)
        o = Oracle(sorted(k.encode() for k in f.GetKeywords()))
        o.set_expressions(exprs, False)
        off = doc_off[:n + 1].cpu().numpy().astype(np.uint64)
        want = o.process(text[:int(off[-1])].cpu().numpy(), off, fold=True, n_threads=16)
        got = bm[:n].cpu().numpy().view(np.uint32)
        out[name]["bitmap_equal_to_oracle_on_first_docs"] = [n, bool(np.array_equal(got, want))]
        out[name]["wide_expression_true_in"] = int((want[:, 1000 >> 5] >> (1000 & 31) & 1).sum())
    f.close()
print(json.dumps(out, indent=1))
if args.json:
    with open(args.json, "w") as fh:
        json.dump(out, fh, indent=1)
