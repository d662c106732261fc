#!/usr/bin/env python3
"""Timing study of the solver kernel (GFT_SOLVE_DEBUG variants). Not part of the tests or the bench contract."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from gofindthem_amd import _lib  # noqa: E402
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine  # noqa: E402
from gofindthem_amd.workload import Workload, make_expressions  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=500000)
ap.add_argument("--terms", type=int, default=10000)
ap.add_argument("--exprs", type=int, default=1000)
ap.add_argument("--inord", type=float, default=0.0)
ap.add_argument("--modes", default="0,1,2,4,3,7")
args = ap.parse_args()
wl = Workload(args.terms)
exprs = make_expressions(wl.terms(), args.exprs, inord_fraction=args.inord, cover=True)
L = _lib.load()
text, off = wl.docs_device(0, args.docs)
bm = torch.zeros((args.docs, (args.exprs + 31) // 32), dtype=torch.int32, device="cuda")
for mode in args.modes.split(","):
    os.environ["GFT_SOLVE_DEBUG"] = mode          # (read when the engine is created / programs are set)
    f = Finder(GpuEngine.__new__(GpuEngine), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    eh = f.engine_handle()
    L.gft_set_stream(eh, torch.cuda.current_stream().cuda_stream)
    f.ProcessDevice(text.data_ptr(), off.data_ptr(), args.docs, bm.data_ptr())
    L.gft_profile_enable(eh, 1)
    L.gft_profile_reset(eh)
    for _ in range(3):
        f.ProcessDevice(text.data_ptr(), off.data_ptr(), args.docs, bm.data_ptr())
    out = []
    for name in (b"scan", b"solve", b"aux"):
        ms, n = C.c_double(), C.c_uint64()
        L.gft_profile_read(eh, name, C.byref(ms), C.byref(n))
        out.append("%s %.3f ms x%d" % (name.decode(), ms.value / max(n.value, 1), n.value))
    L.gft_profile_enable(eh, 0)
    print("GFT_SOLVE_DEBUG=%s  %s" % (mode, "  ".join(out)), flush=True)
    f.close()
