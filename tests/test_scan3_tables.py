"""The scan kernel's table compiler (csrc/scan3_tables.cpp) against the oracle, on the CPU.

gft_debug_emulate_scan compiles a dictionary into the stride-2 suffix-window tables and walks them over one document the
way the kernel's lanes do (filter bit -> short records / Bloom cell -> bucket slots).  The set of (term, position) pairs
must equal the oracle's MatchAll restatement for both probe parities (lo = 0 / a unit that continues a document).
"""
import ctypes as C
import random

import numpy as np
import pytest

from gofindthem_amd import _lib
from gofindthem_amd.workload import Workload
from oracle.pyoracle import Oracle, pack_strings, POS_END, POS_START


def emulate(terms, text, lo=0, pos_end=False, fold=False):
    L = _lib.load()
    tb, to = pack_strings(terms)
    t = np.frombuffer(bytes(text), dtype=np.uint8).copy() if len(text) else np.zeros(1, np.uint8)
    cap = 16 * (len(text) + 16)
    ot, op = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    need = C.c_uint64(0)
    rc = L.gft_debug_emulate_scan(tb.ctypes.data, to.ctypes.data, len(terms), t.ctypes.data, len(text), lo,
                                  _lib.GFT_POS_END if pos_end else 0, _lib.GFT_FOLD_ASCII if fold else 0,
                                  ot.ctypes.data, op.ctypes.data, cap, C.byref(need))
    assert rc == 0, rc
    n = need.value
    return sorted(zip(ot[:n].tolist(), op[:n].tolist()))


def oracle_pairs(terms, text, lo=0, pos_end=False, fold=False):
    o = Oracle(terms, POS_END if pos_end else POS_START)
    blob, off = pack_strings([bytes(text)])
    _, tid, pos = o.scan(blob, off, fold=fold)
    lens = [len(t) for t in o.terms()]
    out = []
    for t, p in zip(tid.tolist(), pos.tolist()):
        end = p if pos_end else p + lens[t] - 1
        if end >= lo:
            out.append((t, p))
    return sorted(out)


def check(terms, text, **kw):
    for lo in sorted({0, 1, 2, 3, min(5, len(text)), len(text) // 2, max(len(text) - 1, 0)}):
        if lo > len(text):
            continue
        got, want = emulate(terms, text, lo=lo, **kw), oracle_pairs(terms, text, lo=lo, **kw)
        assert got == want, (lo, kw, [x for x in got if x not in want][:5], [x for x in want if x not in got][:5])


def test_ushers():
    check([b"he", b"she", b"his", b"hers"], b"ushers")
    check([b"he", b"she", b"his", b"hers"], b"ushers", pos_end=True)


def test_lengths_one_to_six_every_alignment():
    terms = [b"a", b"ab", b"abc", b"abcd", b"abcde", b"abcdef", b"bcd", b"cd", b"d", b"xbcd", b"bcdx"]
    for pad in range(0, 6):
        text = b"z" * pad + b"abcdefxbcdxabcd" + b"q" * (5 - pad) + b"abc"
        check(terms, text)
        check(terms, text, pos_end=True)


def test_fold():
    check([b"hello", b"lo", b"world"], b"HeLLo WORLD hello", fold=True)
    check([b"Hello"], b"hello Hello", fold=True)      # an upper-case term never matches folded text
    check([b"Hello"], b"hello Hello", fold=False)


@pytest.mark.parametrize("seed", range(6))
def test_random_small_alphabet(seed):
    rng = random.Random(seed)
    alpha = b"abc" if seed % 2 else b"abcdefgh "
    terms = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))) for _ in range(rng.randint(1, 60))})
    text = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 400)))
    check(terms, text, pos_end=bool(seed & 1))


@pytest.mark.parametrize("seed", range(4))
def test_random_large_alphabet_merged_groups(seed):
    # > 26 distinct bytes: byte classes share filter groups, the byte compares keep the result exact
    rng = random.Random(100 + seed)
    alpha = bytes(range(33, 127)) + bytes(range(0xC0, 0xF0))
    hot = alpha[:12]
    terms = list({bytes(rng.choice(hot if rng.random() < 0.7 else alpha) for _ in range(rng.randint(1, 12))) for _ in range(300)})
    text = bytes(rng.choice(hot if rng.random() < 0.8 else alpha) for _ in range(3000))
    # plant terms so that long ones occur
    tl = bytearray(text)
    for _ in range(60):
        t = rng.choice(terms)
        at = rng.randrange(0, len(tl) - len(t))
        tl[at:at + len(t)] = t
    check(terms, bytes(tl), fold=bool(seed & 1))


def test_long_terms_and_shared_windows():
    base = b"abcdefghijklmnopqrstuvwxyz0123456789"
    terms = [base[:n] for n in (5, 8, 19, 20, 21, 24, 25, 30, 36)] + [base[3:], base[7:20], b"xyz0", b"wxyz01"]
    text = b"__" + base + b"--" + base[2:] + base[:22] + b"!"
    check(terms, text)
    check(terms, text, pos_end=True)


def test_workload_documents():
    w = Workload(2000)
    terms = w.terms()
    text, off = w.docs_host(0, 6)
    for d in range(6):
        check(terms, bytes(text[int(off[d]):int(off[d + 1])]), fold=True)


@pytest.mark.parametrize("seed", [3, 5, 7])
def test_input_class_of_the_round_2_abort(seed):
    """tests/test_gpu_parity.py::test_random_dictionaries, the parametrisations behind the one SIGABRT of round 2
    (gpurun_out/t_s3_3.log: inside gft_scan, sixth case onwards): all 256 byte values in the dictionary (seed 3), terms
    of up to 40 / 200 bytes (seeds 5 / 7), documents of up to 70 000 bytes.  The same dictionaries and documents through
    the table compiler and the host emulation of the kernel walk -- this file runs under AddressSanitizer + UBSan in
    tools/asan_host.sh, so an unchecked index or size in the compiler shows here."""
    rng = np.random.default_rng(seed)
    alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256)), b"abcdefghijklmnopqrstuvwxyz "][seed % 5]
    n_terms = [1, 5, 40, 300, 1000, 17, 3000, 64][seed]
    maxlen = [3, 9, 9, 6, 5, 40, 12, 200][seed]
    terms = set()
    for _ in range(n_terms):
        L = int(rng.integers(1, maxlen + 1))
        terms.add(bytes(alpha[i] for i in rng.integers(0, len(alpha), L)))
    terms = sorted(terms)
    for n in (0, 1, 7, 65, 4097, 8449, 70000):
        text = bytes(alpha[i] for i in rng.integers(0, len(alpha), n))
        for lo in sorted({0, min(n, 3), n // 2, min(n, 8192), max(n - 1, 0)}):
            for pos_end in (False, True):
                assert emulate(terms, text, lo=lo, pos_end=pos_end) == oracle_pairs(terms, text, lo=lo, pos_end=pos_end), (n, lo)
    planted = b" ".join(terms[i] for i in rng.integers(0, len(terms), 400))
    assert emulate(terms, planted) == oracle_pairs(terms, planted)
