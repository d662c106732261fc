"""Size-independent properties at BASELINE scale (the CPU oracle needs seconds per 10 k documents, so full-size runs are
pinned through properties the domain offers instead): ProcessText keeps no state between documents
(finder/finder.go:139-179), hence a batch may be cut anywhere, reordered, or processed twice without changing any
document's result; the oracle then spot-checks a sample of the very same run bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from gofindthem_amd import _lib
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine
from gofindthem_amd.workload import Workload, make_expressions
from oracle.pyoracle import Oracle

pytestmark = pytest.mark.gpu

N_DOCS, N_TERMS, N_EXPRS = 200_000, 10_000, 1_000


@pytest.fixture(scope="module")
def run():
    w = Workload(N_TERMS)
    exprs = make_expressions(w.terms(), N_EXPRS, inord_fraction=0.3, cover=True)
    f = Finder(GpuEngine.__new__(GpuEngine), EmptyRgxEngine(), False, device=0)
    f.AddExpressions(exprs)
    f.ForceBuild()
    L = _lib.load()
    assert L.gft_set_stream(f.engine_handle(), torch.cuda.current_stream().cuda_stream) == 0
    text, off = w.docs_device(0, N_DOCS)
    words = (N_EXPRS + 31) // 32

    def process(t, o, n):
        bm = torch.zeros((n, words), dtype=torch.int32, device="cuda")
        f.ProcessDevice(t.data_ptr(), o.data_ptr(), n, bm.data_ptr())
        return bm
    whole = process(text, off, N_DOCS)
    return dict(w=w, exprs=exprs, f=f, L=L, text=text, off=off, process=process, whole=whole)


def test_idempotent_and_not_trivial(run):
    again = run["process"](run["text"], run["off"], N_DOCS)
    assert torch.equal(again, run["whole"])
    per_doc = (run["whole"] != 0).sum(1)
    assert int(per_doc.min().item()) >= 0 and int(per_doc.max().item()) > 0
    assert 0 < int((run["whole"] != 0).sum().item()) and int((run["whole"] != -1).sum().item()) > 0


def test_batch_can_be_cut_anywhere(run):
    """the same documents processed as three uneven batches (sub-views of the same blob, offsets rebased)"""
    off, text, whole = run["off"], run["text"], run["whole"]
    cuts = [0, 1, 77_777, N_DOCS]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        o = (off[a:b + 1] - off[a]).contiguous()
        t = text[int(off[a].item()):int(off[b].item())]
        pad = torch.cat([t, torch.zeros(64, dtype=torch.uint8, device="cuda")])        # readable slack behind the blob
        parts.append(run["process"](pad, o, b - a))
    assert torch.equal(torch.cat(parts, 0), whole)


def test_document_order_does_not_matter(run):
    """documents in reversed order (doc_off need not walk the blob upwards document by document: rebuild the blob)"""
    off, text, whole = run["off"], run["text"], run["whole"]
    n = 50_000
    lens = (off[1:n + 1] - off[:n])
    rlens = torch.flip(lens, [0])
    roff = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    roff[1:] = torch.cumsum(rlens, 0)
    total = int(roff[-1].item())
    # gather bytes: position j of the reversed blob belongs to reversed doc d = searchsorted(roff, j, right) - 1
    j = torch.arange(total, device="cuda")
    d = torch.searchsorted(roff, j, right=True) - 1
    src = off[:n][n - 1 - d] + (j - roff[d])
    rtext = torch.cat([text[src], torch.zeros(64, dtype=torch.uint8, device="cuda")])
    got = run["process"](rtext, roff, n)
    assert torch.equal(torch.flip(got, [0]), whole[:n])


def test_csr_scan_equals_the_concatenation_of_its_halves(run):
    """FindSubstrings over the batch == over its halves (match counts, terms and positions), and the number of matches is
    the one the process path saw"""
    L, f = run["L"], run["f"]
    eh = f.engine_handle()
    off, text = run["off"], run["text"]
    n = 60_000

    def scan(t, o, k):
        m = _lib.GftMatches()
        assert L.gft_scan_device(eh, t.data_ptr(), o.data_ptr(), k, _lib.GFT_FOLD_ASCII, C.byref(m)) == 0
        nm = int(m.n_matches)
        hip = C.CDLL("libamdhip64.so")
        mo = torch.empty(k + 1, dtype=torch.int64, device="cuda")
        ti = torch.empty(max(nm, 1), dtype=torch.int32, device="cuda")
        po = torch.empty(max(nm, 1), dtype=torch.int32, device="cuda")
        assert hip.hipMemcpy(C.c_void_p(mo.data_ptr()), C.c_void_p(m.match_off), C.c_size_t(8 * (k + 1)), C.c_int(3)) == 0
        if nm:
            assert hip.hipMemcpy(C.c_void_p(ti.data_ptr()), C.c_void_p(m.term_id), C.c_size_t(4 * nm), C.c_int(3)) == 0
            assert hip.hipMemcpy(C.c_void_p(po.data_ptr()), C.c_void_p(m.pos), C.c_size_t(4 * nm), C.c_int(3)) == 0
        return mo, ti[:nm], po[:nm]
    mo, ti, po = scan(text, off[:n + 1].contiguous(), n)
    h = n // 2
    a = scan(text, off[:h + 1].contiguous(), h)
    o2 = (off[h:n + 1] - off[h]).contiguous()
    t2 = torch.cat([text[int(off[h].item()):int(off[n].item())], torch.zeros(64, dtype=torch.uint8, device="cuda")])
    b = scan(t2, o2, n - h)
    assert torch.equal(torch.cat([a[0], b[0][1:] + a[0][-1]]), mo)
    assert torch.equal(torch.cat([a[1], b[1]]), ti) and torch.equal(torch.cat([a[2], b[2]]), po)
    # every per-document list is in emission order: end offsets never decrease (positions are starts: start + len - 1)
    assert int(mo[-1].item()) == ti.numel() > n


def test_oracle_spot_check_of_the_same_run(run):
    """bit-exact against the CPU oracle on documents sampled across the whole batch"""
    w, exprs, whole = run["w"], run["exprs"], run["whole"]
    o = Oracle(w.terms())
    o.set_expressions(exprs, False)
    for first in (0, 99_000, N_DOCS - 300):
        t, off = w.docs_host(first, 300)
        want = o.process(t, off, fold=True, n_threads=4)
        got = whole[first:first + 300].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), first
