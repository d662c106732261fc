#!/usr/bin/env python3
"""Differential fuzzing of the scan boundary (and, every fourth iteration, of scan + solve: random expressions with
INORD groups, random solver group widths, hit bitmaps compared) -- the scan boundary (gft_scan: every occurrence of every term, canonical order) against the CPU
oracle, far beyond what the test-suite's fixed seeds cover: random alphabets (2 letters ... all 256 byte values), term
lengths 1 ... 300, texts that mix random bytes with planted terms, document sizes around the work-unit borders, empty
documents, matches that end on the last byte of the blob, ASCII folding on/off, both position modes, and the kernel's
cross-check variants.  Not part of the test-suite or the bench contract; exits non-zero on the first difference.

    python tools/fuzz_scan.py [--iters N] [--seed S]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from helpers import csr_lists, docs  # noqa: E402
from gofindthem_amd.engine import Engine, GftError  # noqa: E402
from oracle.pyoracle import Oracle, POS_END, POS_START  # noqa: E402

ALPHAS = [b"ab", b"abc", b"abcdefgh", b"abcdefghijklmnopqrstuvwxyz ", b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ .,",
          bytes(range(1, 120)), bytes(range(256))]


def _have(kernel):
    """scan2 / scan4 exist only in a GFT_EXTRA_KERNELS=1 build of libgft.so: a product build takes its own choice instead"""
    from gofindthem_amd import _lib
    if kernel in ("scan2", "scan4") and b"extra_kernels=1" not in _lib.load().gft_build_info():
        return "auto"
    return kernel


def run(iters, seed, budget_s, eng=None, progress=False):
    """-> (iterations done, None) or (iterations done, description of the first difference)"""
    own = eng is None
    if own:
        eng = Engine()
    saved = {k: os.environ.get(k) for k in ("GFT_SCAN_ORDERED", "GFT_SCAN_KERNEL", "GFT_SOLVE_GROUP_DOCS", "GFT_SCAN_FPT_GLOBAL", "GFT_SCAN5_BLOOM_KB")}
    t_start = time.time()
    done = 0
    try:
        for it in range(iters):
            if time.time() - t_start > budget_s:
                break
            err = one(eng, seed, it)
            if err:
                return done, err
            done += 1
            if progress and done % 50 == 0:
                print("fuzz_scan: %d iterations, %.0f s" % (done, time.time() - t_start), flush=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if own:
            eng.close()
    return done, None


def one_process(eng, seed, it):
    """scan + solve: random expressions (AND/OR/NOT, parentheses, INORD groups) over a random dictionary -> hit bitmap"""
    from helpers import tree_to_program
    from gofindthem_amd.workload import make_expressions
    from oracle import dsl_ref
    rng = np.random.default_rng(seed * 7919 + it)
    alpha = [b"ab", b"abc", b"abcdefgh", b"abcdefghijklmnopqrstuvwxyz"][int(rng.integers(4))]
    A = np.frombuffer(alpha, dtype=np.uint8)
    n_terms = int(rng.choice([3, 20, 200, 3000]))
    maxlen = int(rng.choice([3, 6, 12, 40]))
    terms = set()
    for _ in range(n_terms):
        terms.add(A[rng.integers(0, len(alpha), int(rng.integers(1, maxlen + 1)))].tobytes())
    tl = sorted(terms)
    pos_mode = POS_END if rng.integers(2) else POS_START
    os.environ["GFT_SCAN_KERNEL"] = _have(["auto", "scan2", "scan3", "scan4", "scan5"][int(rng.integers(5))])     # (auto: the library's own choice)
    os.environ.pop("GFT_SCAN_ORDERED", None)
    g = [None, None, "32", "16", "8", "0"][int(rng.integers(6))]
    if g is None:
        os.environ.pop("GFT_SOLVE_GROUP_DOCS", None)
    else:
        os.environ["GFT_SOLVE_GROUP_DOCS"] = g
    eng.build(tl, pos_end=(pos_mode == POS_END))
    o = Oracle(tl, pos_mode)
    n_exprs = int(rng.choice([1, 31, 33, 200, 2100]))
    exprs = make_expressions(tl, n_exprs, inord_fraction=float(rng.choice([0.0, 0.3, 1.0])), seed=int(rng.integers(1 << 30)),
                             cover=bool(rng.integers(2)) and n_exprs <= len(tl) <= 60 * n_exprs)
    # ... and expressions whose operands are subtrees on both sides of an operator, nested to the right, to the left or
    # balanced: the accumulator stack of the fused programs (two registers, four registers, scratch beyond)
    def lit(t):
        return '"%s"' % t.decode("latin-1")

    def pair():
        a, b = tl[int(rng.integers(len(tl)))], tl[int(rng.integers(len(tl)))]
        return "(%s%s %s %s)" % ("not " if rng.integers(3) == 0 else "", lit(a), "and" if rng.integers(2) else "or", lit(b))

    def nested(depth, side):
        if side == 2:
            return pair() if depth == 0 else "(%s %s %s)" % (nested(depth - 1, 2), "and" if rng.integers(2) else "or", nested(depth - 1, 2))
        e = pair()
        for _ in range(depth):
            op = "and" if rng.integers(2) else "or"
            e = "(%s %s %s)" % ((pair(), op, e) if side == 0 else (e, op, pair()))
            if rng.integers(4) == 0:
                e = "not " + e
        return e
    if all(32 <= c < 127 and c not in b'"\\' for t in tl for c in t):
        extra = [nested(int(rng.choice([1, 2, 3, 6, 20])), int(rng.integers(2))) for _ in range(int(rng.integers(0, 40)))]
        extra += [nested(int(rng.integers(1, 6)), 2) for _ in range(int(rng.integers(0, 6)))]
        for e in extra:
            exprs.insert(int(rng.integers(len(exprs) + 1)), e)
        n_exprs = len(exprs)
    o.set_expressions(exprs, case_sensitive=False)
    progs = [tree_to_program(dsl_ref.parse(e, False)[0], lambda lit: eng.term_id(lit)) for e in exprs]
    try:
        eng.set_programs(progs)
    except GftError:
        return None       # beyond the device solver's documented limits (an INORD group of more than 64 leaves)
    texts = []
    for _ in range(int(rng.choice([1, 2, 63, 64, 65, 130, 400]))):
        n = int(rng.choice([0, 5, 64, 500, 4096, 8193, 20000]))
        parts, size = [], 0
        while size < n:
            w = tl[int(rng.integers(len(tl)))] if rng.integers(2) else A[rng.integers(0, len(alpha), int(rng.integers(1, 9)))].tobytes()
            parts.append(w + (b" " if rng.integers(2) else b""))
            size += len(parts[-1])
        texts.append(b"".join(parts))
    blob, off = docs(texts)
    got = eng.process(blob, off, fold=True)
    want = o.process(blob, off, fold=True)
    os.environ.pop("GFT_SOLVE_GROUP_DOCS", None)
    if np.array_equal(got, want):
        return None
    bad = np.argwhere(got != want)[0]
    return "process: iter %d seed %d doc %d word %d (group docs %s, %d terms, %d expressions, pos_mode %d)" % (
        it, seed, bad[0], bad[1], g, len(tl), n_exprs, pos_mode)


def one(eng, seed, it):
    if it % 4 == 3:
        return one_process(eng, seed, it)
    rng = np.random.default_rng(seed * 100003 + it)
    alpha = ALPHAS[int(rng.integers(len(ALPHAS)))]
    A = np.frombuffer(alpha, dtype=np.uint8)
    n_terms = int(rng.choice([1, 3, 20, 200, 2000, 8000]))
    maxlen = int(rng.choice([3, 5, 8, 12, 30, 300]))
    minlen = int(rng.choice([1, 1, 2, 4, 5]))
    terms = set()
    for _ in range(n_terms):
        L = int(rng.integers(min(minlen, maxlen), maxlen + 1))
        terms.add(A[rng.integers(0, len(alpha), L)].tobytes())
    fold = bool(rng.integers(2)) and not any(65 <= b <= 90 for t in terms for b in t)   # folding needs lower-case terms
    pos_mode = POS_END if rng.integers(2) else POS_START
    variant = ["", "ordered", "dfa", "scan3", "scan2", "scan5"][int(rng.choice([0, 0, 1, 2, 3, 3, 4, 5, 5]))]
    os.environ.pop("GFT_SCAN_ORDERED", None)
    os.environ["GFT_SCAN_KERNEL"] = _have({"dfa": "dfa", "scan3": "scan3", "scan2": "scan2", "ordered": "scan2", "scan4": "scan4", "scan5": "scan5"}.get(variant, "auto"))
    if variant == "ordered" and os.environ["GFT_SCAN_KERNEL"] == "scan2":
        os.environ["GFT_SCAN_ORDERED"] = "1"
    # a third of the builds: the fingerprint table in global memory as for a 100 000-term dictionary, behind Bloom levels of
    # 8 192 bits (crowded) or the default size
    os.environ.pop("GFT_SCAN_FPT_GLOBAL", None)
    os.environ.pop("GFT_SCAN5_BLOOM_KB", None)
    if rng.integers(3) == 0:
        os.environ["GFT_SCAN_FPT_GLOBAL"] = "1"
        if rng.integers(2):
            os.environ["GFT_SCAN5_BLOOM_KB"] = "1"
    tl = sorted(terms)
    eng.build(tl, pos_end=(pos_mode == POS_END))
    o = Oracle(tl, pos_mode)
    texts = []
    n_docs = int(rng.choice([1, 2, 7, 40, 300]))
    for _ in range(n_docs):
        n = int(rng.choice([0, 1, 3, 4, 5, 63, 64, 65, 500, 4095, 4096, 4097, 8191, 8192, 8193, 8200, 16384, 16390, 30000]))
        kind = int(rng.integers(3))
        if kind == 0 or not tl:
            t = A[rng.integers(0, len(alpha), n)].tobytes()
        else:       # planted terms glued together (kind 2: with random separators), cut to n bytes
            parts = []
            size = 0
            while size < n:
                w = tl[int(rng.integers(len(tl)))]
                if kind == 2 and rng.integers(3) == 0:
                    w = w + bytes([alpha[int(rng.integers(len(alpha)))]])
                parts.append(w)
                size += len(w)
            t = b"".join(parts)[:n] if rng.integers(2) else b"".join(parts)      # sometimes end exactly on a term
        if fold and rng.integers(2) and t:
            a = np.frombuffer(t, dtype=np.uint8).copy()
            up = (a >= 97) & (a <= 122) & (rng.integers(0, 3, a.size) == 0)
            a[up] -= 32
            t = a.tobytes()
        texts.append(t)
    blob, off = docs(texts)
    got = csr_lists(*eng.scan(blob, off, fold=fold))
    want = csr_lists(*o.scan(blob, off, fold=fold))
    if got == want:
        return None
    for d, (a, b) in enumerate(zip(got, want)):
        if a != b:
            sa, sb = set(a), set(b)
            return ("iter %d seed %d doc %d (len %d) variant=%r fold=%s pos_mode=%d alphabet=%d terms=%d maxlen=%d; missing %s extra %s "
                    "order-only %s" % (it, seed, d, len(texts[d]), variant, fold, pos_mode, len(alpha), len(tl), maxlen,
                                       sorted(sb - sa)[:10], sorted(sa - sb)[:10], sa == sb))
    return "match_off differs"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=240.0)
    args = ap.parse_args()
    t0 = time.time()
    n, err = run(args.iters, args.seed, args.budget_s, progress=True)
    if err:
        print("MISMATCH", err)
        sys.exit(1)
    print("fuzz_scan: %d iterations identical to the oracle (%.0f s)" % (n, time.time() - t0))
