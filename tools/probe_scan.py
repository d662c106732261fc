#!/usr/bin/env python3
"""Timing study of the scan kernel alone (not part of the test-suite or the bench contract).
    python tools/probe_scan.py [--docs N] [--terms T]
Prints the scan kernel's HIP-event time for the production kernel and for the GFT_SCAN_DEBUG timing variants.
(The figure is a mean over the launches of a few calls: with a dictionary whose units are resized after the first batch --
100 000 terms -- it includes that batch's short second launch; bench.py's per-step kernel time is the one to quote there.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gofindthem_amd.engine import Engine  # noqa: E402
from gofindthem_amd.workload import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=500000)
ap.add_argument("--terms", type=int, default=10000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--alphabet", default="lower")
ap.add_argument("--modes", default="0,1,2")
ap.add_argument("--unordered", action="store_true", help="time the scan as the process path runs it (any order)")
ap.add_argument("--positions", action="store_true", help="with --unordered: an INORD program, so that the scan also writes positions")
args = ap.parse_args()

wl = Workload(args.terms, alphabet=args.alphabet)
text, off = wl.docs_device(0, args.docs)
nbytes = text.numel()
bm = torch.zeros((args.docs, 1), dtype=torch.int32, device="cuda")
eng = None


def make_engine():
    """GFT_SCAN_DEBUG is read when the engine is created"""
    global eng
    if eng is not None:
        eng.close()
    eng = Engine(0)
    eng.build(sorted({t.decode('utf-8').lower().encode('utf-8') for t in wl.terms()}))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    # (one UNIT program: presence only; inord(t0 and t1): the scan writes positions too)
    eng.set_programs([[1 << 28 | 1 << 27 | 0, 1 << 28 | 1 << 27 | 1, 2 << 28 | 1 << 27, 5 << 28]] if args.positions else [[1 << 28]])


class _M:
    n_matches = -1


def run():
    if args.unordered:
        eng.process_device(text.data_ptr(), off.data_ptr(), args.docs, bm.data_ptr(), fold=True)
        return _M
    return eng.scan_device(text.data_ptr(), off.data_ptr(), args.docs, fold=True)


for mode in args.modes.split(","):
    os.environ["GFT_SCAN_DEBUG"] = mode
    make_engine()
    run()    # warm-up
    eng.profile(True)
    eng.profile_reset()
    for _ in range(args.reps):
        m = run()
    ms, n = eng.profile_read("scan")
    aux, na = eng.profile_read("aux")
    eng.profile(False)
    print("GFT_SCAN_DEBUG=%s  scan kernel %.3f ms/launch  (%.1f GB/s of text)  aux %.3f ms/call  matches=%d"
          % (mode, ms / n, nbytes / (ms / n) / 1e6, aux / max(args.reps, 1), m.n_matches), flush=True)
