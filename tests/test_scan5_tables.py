"""The filter of gft_scan5.hip (one probe per two text bytes over 3-grams of merged byte classes) against the
one-probe-per-byte filter of gft_scan2.hip, on the CPU.

gft_debug_scan5_filter compiles a dictionary into the suffix-window tables, derives the 3-gram table (build_scan5_tables) and
walks one document the way the kernel's lanes do, for either parity of a lane's first byte.  Whatever the exact filter
flags -- and every match ends at a flagged position: tests/test_scan3_tables.py, the GPU parity tests -- the 3-gram filter
must flag too; with no classes merged the two are the same function.
"""
import ctypes as C
import random

import numpy as np
import pytest

from gofindthem_amd import _lib
from gofindthem_amd.workload import Workload
from oracle.pyoracle import Oracle, pack_strings, POS_END


def flags(terms, text, lane_start=0, groups=0, fold=False):
    L = _lib.load()
    tb, to = pack_strings(terms)
    t = np.frombuffer(bytes(text), dtype=np.uint8).copy() if len(text) else np.zeros(1, np.uint8)
    ex, du = np.zeros(max(len(text), 1), np.uint8), np.zeros(max(len(text), 1), np.uint8)
    g = C.c_uint32(0)
    rc = L.gft_debug_scan5_filter(tb.ctypes.data, to.ctypes.data, len(terms), t.ctypes.data, len(text), lane_start,
                                  _lib.GFT_FOLD_ASCII if fold else 0, groups, ex.ctypes.data, du.ctypes.data, C.byref(g))
    assert rc == 0, rc
    return ex[:len(text)], du[:len(text)], g.value


def match_ends(terms, text, fold=False):
    o = Oracle(terms, POS_END)
    blob, off = pack_strings([bytes(text)])
    _, tid, pos = o.scan(blob, off, fold=fold)
    lens = [len(t) for t in o.terms()]
    return {(p, lens[t]) for t, p in zip(tid.tolist(), pos.tolist())}


def test_unmerged_filter_is_the_exact_filter_and_covers_every_match():
    rng = random.Random(5)
    for alpha in (b"ab", b"abc", b"abcdefghijklmnopqrstuvwxyz"):
        terms = sorted({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))) for _ in range(200)})
        text = bytes(rng.choice(alpha + b" ") for _ in range(3000))
        ends = match_ends(terms, text)
        for start in (0, 1):
            ex, du, g = flags(terms, text, lane_start=start)
            assert g == len(set(b"".join(terms))) + 1                      # one group per byte class + "other"
            assert np.array_equal(ex, du)
            # the windows of short terms and of terms anchored at their end are flagged where the term ends; a shifted anchor
            # (scan2_tables.cpp pick_off) is flagged up to four bytes earlier
            for p, length in ends:
                assert ex[max(p - 4, 0):p + 1].any(), (p, length)


@pytest.mark.parametrize("groups", [20, 12, 4, 2])
def test_merged_classes_only_add_flags(groups):
    rng = random.Random(groups)
    alpha = b"abcdefghijklmnopqrstuvwxyz"
    terms = sorted({bytes(rng.choice(alpha) for _ in range(rng.randint(2, 12))) for _ in range(500)})
    text = bytearray(rng.choice(alpha + b"  ") for _ in range(6000))
    for _ in range(60):                                                    # planted terms
        t = rng.choice(terms)
        at = rng.randrange(0, len(text) - len(t))
        text[at:at + len(t)] = t
    for fold in (False, True):
        body = bytes(text).upper() if fold else bytes(text)
        for start in (0, 1, 2, 3):
            ex, du, g = flags(terms, body, lane_start=start, groups=groups, fold=fold)
            assert g == groups
            assert not (ex & ~du & 1).any(), int((ex & ~du & 1).sum())
            assert du.sum() >= ex.sum()


def test_benchmark_dictionary_at_the_planned_group_count():
    """the benchmark's shape: 22 of 27 classes (what fits LDS next to the other tables) must not flag much more than the
    exact filter does (tools/sim: + 4 %)"""
    w = Workload(10000)
    terms = w.terms()
    text, off = w.docs_host(0, 40)
    body = bytes(text[:int(off[40])])
    ex, du, g = flags(terms, body, groups=22, fold=True)
    assert g == 22 and not (ex & ~du & 1).any()
    assert ex.sum() > 0.04 * len(body)
    assert du.sum() <= 1.10 * ex.sum(), (int(ex.sum()), int(du.sum()))
    ex1, du1, _ = flags(terms, body, lane_start=1, groups=22, fold=True)
    assert np.array_equal(ex1, ex) and abs(int(du1.sum()) - int(du.sum())) <= 0.02 * du.sum()


def test_large_alphabet_mixed_word_list():
    """more than 32 byte classes (capitals folded away, digits, punctuation, two-byte UTF-8 letters: 54 classes): no direct
    short-term table, the long-term tables and the 3-gram filter are built all the same -- every flag of the exact filter, and
    with it every match of the oracle, must survive the merge of 54 classes into 22 groups"""
    w = Workload(3000, alphabet="mixed")
    kws = sorted({t.decode("utf-8").lower().encode("utf-8") for t in w.terms()})
    assert len({b for t in kws for b in t}) > 32
    text, off = w.docs_host(0, 30)
    body = bytes(text[:int(off[30])])
    ends = match_ends(kws, body, fold=True)
    for start in (0, 1):
        ex, du, g = flags(kws, body, lane_start=start, groups=22, fold=True)
        assert g == 22 and not (ex & ~du & 1).any()
        assert du.sum() <= 1.25 * ex.sum(), (int(ex.sum()), int(du.sum()))
        for p, length in ends:
            assert ex[max(p - 4, 0):p + 1].any(), (p, length)
    ex, du, g = flags(kws, body, fold=True)            # as little merging as a filter word's 32 bits allow
    assert g == 32 and not (ex & ~du & 1).any()


def test_input_edges():
    ex, du, _ = flags([b"ab"], b"")
    assert ex.size == 0 and du.size == 0
    ex, du, _ = flags([b"ab", b"b"], b"b", lane_start=1)
    assert ex.tolist() == du.tolist() == [1]
    ex, du, _ = flags([b"abcd"], b"xabcd")
    assert ex.tolist() == du.tolist() == [0, 0, 0, 0, 1]
