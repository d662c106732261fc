"""Several devices behind ONE handle of the C ABI (gft_engine_create_multi): a Go caller keeps
finder.NewFinder(&GpuEngine{...}) and the library fans a batch out -- tables replicated, contiguous document ranges of
near-equal text bytes, one host thread + stream per device, results back in document order.

With >= 2 visible devices the test uses devices [0, 1] (and the RCCL gather of the device-resident entry point);
on a one-GPU box it names device 0 twice: two engines, two host threads and streams on the same card, the gather done by
device-to-device copies -- everything but the RCCL call itself."""
import ctypes as C

import numpy as np
import pytest
import torch

from gofindthem_amd import _lib
from gofindthem_amd.engine import Engine
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine, PyRegexpEngine
from gofindthem_amd.workload import Workload, make_expressions
from helpers import assert_csr_equal
from oracle.pyoracle import Oracle, pack_strings

pytestmark = pytest.mark.gpu


def _devices():
    n = torch.cuda.device_count()
    return [0, 1] if n >= 2 else [0, 0]


def test_split_is_contiguous_and_byte_balanced():
    e = Engine(devices=[0, 0, 0])
    try:
        L = _lib.load()
        assert L.gft_n_devices(e._h) == 3
        off = np.asarray([0, 10, 10, 500, 510, 520, 1000, 1500], np.uint64)
        cut = np.zeros(4, np.uint64)
        assert L.gft_split_docs(e._h, off.ctypes.data, 7, cut.ctypes.data) == 0
        assert cut.tolist() == [0, 3, 6, 7]          # 500 | 500 | 500 bytes
        assert L.gft_split_docs(e._h, off.ctypes.data, 0, cut.ctypes.data) == 0 and cut.tolist() == [0, 0, 0, 0]
    finally:
        e.close()


def test_empty_batch_without_offsets_on_a_multi_device_handle():
    """gft_scan / gft_process accept n_docs == 0 with doc_off == NULL on one device; a multi-device handle used to copy
    doc_off[0 .. 1) per shard."""
    e = Engine(devices=_devices())
    try:
        e.build([b"ab", b"b"])
        e.set_programs([[1 << 28]])
        L = _lib.load()
        m = _lib.GftMatches()
        assert L.gft_scan(e._h, None, None, 0, 0, C.byref(m)) == 0 and m.n_matches == 0
        assert L.gft_process(e._h, None, None, 0, 0, None, None) == 0
    finally:
        e.close()


def test_engine_over_two_devices_equals_one():
    w = Workload(2000)
    terms = w.terms()
    o = Oracle(terms)
    text, off = w.docs_host(0, 501)
    one, two = Engine(0), Engine(devices=_devices())
    try:
        for e in (one, two):
            e.build(terms)
        assert two.terms() == one.terms()
        # FindSubstrings: the shards' CSRs come back as one
        assert_csr_equal(two.scan(text, off, fold=True), o.scan(text, off, fold=True))
        assert_csr_equal(two.scan(text[:0], off[:1]), one.scan(text[:0], off[:1]))
        assert_csr_equal(two.scan(text, off[:2]), one.scan(text, off[:2]))            # one document: the second shard is empty
        exprs = make_expressions(terms, 200, inord_fraction=0.4)
        o.set_expressions(exprs, False)
        from test_gpu_parity import _programs
        progs, _ = _programs(o, one, exprs, False)
        for e in (one, two):
            e.set_programs(progs)
        want = o.process(text, off, fold=True)
        assert np.array_equal(two.process(text, off, fold=True), want)
        assert np.array_equal(two.process(text, off, fold=True), want)
    finally:
        one.close()
        two.close()


def test_finder_over_two_devices_with_regex_terms():
    """the whole Finder on a multi-device handle, regex prefilter (gft_process_again on every shard) included"""
    w = Workload(300)
    terms = w.terms()
    rx = ["en.*nr", "po[a-z]+ud", "q+"]
    exprs = make_expressions(terms, 120, inord_fraction=0.4, regexes=rx)
    text, off = w.docs_host(0, 160)
    got = []
    for kw in (dict(device=0), dict(devices=_devices())):
        f = Finder(GpuEngine(), PyRegexpEngine(), False, **kw)
        f.AddExpressions(exprs)
        got.append(f.ProcessTexts(blob=text, doc_off=off))
        one = f.ProcessText(bytes(text[int(off[5]):int(off[6])]))
        assert [r.ExpresionIndex for r in one] == [i for i in range(len(exprs)) if got[-1][5, i >> 5] >> (i & 31) & 1]
        f.close()
    assert np.array_equal(got[0], got[1]) and got[0].any()


@pytest.mark.parametrize("mode", ["default", "rccl-self"])
def test_device_resident_shards_and_the_gather(mode, monkeypatch):
    """gft_process_device_multi: every device's shard is already in its HBM; the bitmaps are gathered to the first device
    (RCCL when the devices are distinct).  "rccl-self": on a one-GPU box GFT_RCCL_SELF=1 gives the handle over [0, 0, 0] ONE
    communicator of one rank (ncclCommInitAll) and the gather is that rank's grouped ncclSend / ncclRecv to itself -- dlopen,
    every bound entry point, the group and the stream ordering of the N-device gather."""
    devs = _devices()
    if mode == "rccl-self":
        if torch.cuda.device_count() >= 2:
            pytest.skip("distinct devices: the default case already gathers with RCCL")
        monkeypatch.setenv("GFT_RCCL_SELF", "1")
        devs = [0, 0, 0]
    w = Workload(1000)
    terms = w.terms()
    exprs = make_expressions(terms, 96, inord_fraction=0.3, cover=True)
    f = Finder(GpuEngine(), EmptyRgxEngine(), False, devices=devs)
    f.AddExpressions(exprs)
    f.ForceBuild()
    L = _lib.load()
    eh = f.engine_handle()
    want_mode = b"rccl" if mode == "rccl-self" or torch.cuda.device_count() >= 2 else b"copy"
    assert L.gft_gather_mode(eh) == want_mode, L.gft_last_error(eh)
    text, off = w.docs_host(0, 300)
    cut = np.zeros(len(devs) + 1, np.uint64)
    assert L.gft_split_docs(eh, off.ctypes.data, 300, cut.ctypes.data) == 0
    keep, tp, op, nd = [], [], [], []
    for i, d in enumerate(devs):
        a, b = int(cut[i]), int(cut[i + 1])
        t = np.concatenate([text[int(off[a]):int(off[b])], np.zeros(64, np.uint8)])
        o = (off[a:b + 1] - off[a]).astype(np.int64)
        tt, oo = torch.from_numpy(t).to("cuda:%d" % d), torch.from_numpy(o).to("cuda:%d" % d)
        keep += [tt, oo]
        tp.append(tt.data_ptr()); op.append(oo.data_ptr()); nd.append(b - a)
    words = 3
    bm = torch.zeros((300, words), dtype=torch.int32, device="cuda:%d" % devs[0])
    torch.cuda.synchronize()
    # (the finder uploaded its programs when it was built: the engine-level entry point can be used directly)
    f.ProcessTexts(blob=text[:int(off[1])], doc_off=off[:2])
    rc = L.gft_process_device_multi(eh, (C.c_void_p * len(devs))(*tp), (C.c_void_p * len(devs))(*op),
                                    (C.c_uint64 * len(devs))(*nd), _lib.GFT_FOLD_ASCII, bm.data_ptr())
    assert rc == 0, L.gft_last_error(eh)
    o = Oracle(sorted(k.encode() for k in f.GetKeywords()))
    o.set_expressions(exprs, False)
    want = o.process(text, off, fold=True)
    assert np.array_equal(bm.cpu().numpy().astype(np.uint32), want)
    # a second batch through the same communicators (smaller last shard, an empty middle one)
    bm.zero_()
    nd2 = list(nd)
    if len(devs) == 3:
        nd2[1] = 0
    rc = L.gft_process_device_multi(eh, (C.c_void_p * len(devs))(*tp), (C.c_void_p * len(devs))(*op),
                                    (C.c_uint64 * len(devs))(*nd2), _lib.GFT_FOLD_ASCII, bm.data_ptr())
    assert rc == 0, L.gft_last_error(eh)
    got = bm.cpu().numpy().astype(np.uint32)
    rows = [want[int(cut[i]):int(cut[i]) + nd2[i]] for i in range(len(devs))]
    assert np.array_equal(got[:sum(nd2)], np.concatenate(rows))
    f.close()


def test_non_ascii_text_in_the_first_shard_only_is_reported():
    """gft_last_nonascii on a multi-device handle is the OR over ALL shards -- the first device's own flag included (it is
    the handle's own field and used to be cleared before it was read): an upper-case non-ASCII letter in shard 0 only must
    still send the caller to the host's ToLower (finder/finder.go:140-142)."""
    L = _lib.load()
    e = Engine(devices=_devices())
    try:
        e.build([b"ecole", "\u00e9cole".encode()])
        first = ["\u00c9COLE nationale"] + ["plain ascii text"] * 3          # upper-case E-acute: ASCII folding is not ToLower
        rest = ["nothing but ascii here, and plenty of it " * 4] * 4
        blob, off = pack_strings(first + rest)
        cut = np.zeros(3, np.uint64)
        assert L.gft_split_docs(e._h, off.ctypes.data, len(first) + len(rest), cut.ctypes.data) == 0
        assert 1 <= int(cut[1]) < len(first) + len(rest)                 # the non-ASCII document is in shard 0, shard 1 is ASCII
        e.scan(blob, off, fold=True)
        assert L.gft_last_nonascii(e._h) == 1
        ascii_blob, ascii_off = pack_strings(rest + rest)
        e.scan(ascii_blob, ascii_off, fold=True)
        assert L.gft_last_nonascii(e._h) == 0
        # ... and in the last shard only
        blob2, off2 = pack_strings(rest + first[::-1])
        e.scan(blob2, off2, fold=True)
        assert L.gft_last_nonascii(e._h) == 1
        # an empty batch with no offsets at all is fine on a multi-device handle too
        m = L.gft_scan  # (through the wrapper: n_docs = 0)
        mo, ti, po = e.scan(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        assert mo.tolist() == [0] and ti.size == 0
    finally:
        e.close()


def test_empty_batch_after_a_non_ascii_one_forgets_the_verdict():
    """an empty batch scans nothing, so the non-ASCII verdict (and the text range it was taken over) of the
    batch BEFORE must not survive into it -- single handle and multi-device handle (where an empty device-0 shard used
    to carry the stale flag)."""
    L = _lib.load()
    for kw in (dict(device=0), dict(devices=_devices())):
        e = Engine(**kw)
        try:
            e.build([b"ecole", "école".encode()])
            blob, off = pack_strings(["ÉCOLE nationale", "plain ascii"])
            e.scan(blob, off, fold=True)
            assert L.gft_last_nonascii(e._h) == 1
            mo, ti, po = e.scan(np.zeros(0, np.uint8), np.zeros(1, np.uint64), fold=True)
            assert mo.tolist() == [0] and ti.size == 0
            assert L.gft_last_nonascii(e._h) == 0
            # ... and the device-resident entry point with NULL text for an empty batch
            e.set_programs([[1 << 28]])
            e.scan(blob, off, fold=True)
            assert L.gft_last_nonascii(e._h) == 1
            if "device" in kw:
                assert L.gft_process_device(e._h, None, None, 0, _lib.GFT_FOLD_ASCII, None, None) == 0, L.gft_last_error(e._h)
                assert L.gft_last_nonascii(e._h) == 0
        finally:
            e.close()
