"""GPU parity: the HIP path (through the C ABI of libgft.so) against the CPU oracle, bit-exact."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libgft.so is loaded: both must share ONE HIP runtime, the one torch brings along)

from helpers import assert_csr_equal, docs, tree_to_program
from oracle import dsl_ref
from oracle.pyoracle import Oracle, POS_END, POS_START

pytestmark = pytest.mark.gpu


def _kernels():
    """the scan kernels of THIS build of libgft.so: the two production kernels and the DFA kernel (an independent
    algorithm); a library built with GFT_EXTRA_KERNELS=1 (the opt-in cross-check job) also has the earlier suffix-window
    kernels scan2 / scan4"""
    from gofindthem_amd import _lib
    try:
        extra = b"extra_kernels=1" in _lib.load().gft_build_info()
    except Exception:                       # (no library: the tests themselves will say so)
        extra = False
    return ["scan5", "scan3", "dfa"] + (["scan4", "scan2", "scan2-ordered"] if extra else [])


KERNELS = _kernels()
needs_extra_kernels = pytest.mark.skipif("scan4" not in KERNELS, reason="libgft.so was built without GFT_EXTRA_KERNELS=1")


@pytest.fixture(params=KERNELS, autouse=True)
def scan_kernel(request, monkeypatch):
    """every test runs against the suffix-window kernel with one filter probe per two bytes (scan5, the default), the stride-2
    suffix-window kernel (scan3, the fallback) and the general two-tier DFA kernel (GFT_SCAN_KERNEL=dfa) -- and, in a
    GFT_EXTRA_KERNELS=1 build, against scan5's streaming form (scan4) and the round-1 suffix-window kernel
    (GFT_SCAN_KERNEL=scan2: balanced path, and its in-kernel ordered path with GFT_SCAN_ORDERED=1); the variable is read
    by gft_build"""
    monkeypatch.setenv("GFT_SCAN_KERNEL", request.param.split("-")[0])
    if request.param.endswith("-ordered"):
        monkeypatch.setenv("GFT_SCAN_ORDERED", "1")
    else:
        monkeypatch.delenv("GFT_SCAN_ORDERED", raising=False)
    return request.param


@pytest.fixture(scope="module")
def eng():
    from gofindthem_amd.engine import Engine
    e = Engine()
    yield e
    e.close()


def both(eng, terms, pos_mode=POS_START):
    eng.build(terms, pos_end=(pos_mode == POS_END))
    o = Oracle(terms, pos_mode)
    assert eng.terms() == o.terms()
    assert eng.n_states == o.n_states
    return o


def test_ushers(eng):
    o = both(eng, ["he", "she", "his", "hers"])
    blob, off = docs(["ushers", "", "she", "xxhishers"])
    got = eng.scan(blob, off)
    assert_csr_equal(got, o.scan(blob, off))
    assert list(zip(got[1][:3].tolist(), got[2][:3].tolist())) == [(3, 1), (0, 2), (1, 2)]
    o = both(eng, ["he", "she", "his", "hers"], POS_END)
    assert_csr_equal(eng.scan(blob, off), o.scan(blob, off))


def test_empty_inputs(eng):
    o = both(eng, ["a"])
    blob, off = docs([])
    mo, ti, po = eng.scan(blob, off)
    assert mo.tolist() == [0] and ti.size == 0
    blob, off = docs(["", "", ""])
    assert_csr_equal(eng.scan(blob, off), o.scan(blob, off))
    both(eng, [])                       # empty dictionary: nothing matches
    blob, off = docs(["abc", "a"])
    mo, ti, po = eng.scan(blob, off)
    assert mo.tolist() == [0, 0, 0]
    o = both(eng, ["", "b"])            # the empty keyword never matches
    assert_csr_equal(eng.scan(blob, off), o.scan(blob, off))


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("pos_mode", [POS_START, POS_END])
def test_random_dictionaries(eng, seed, pos_mode):
    rng = np.random.default_rng(seed)
    alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256)), b"abcdefghijklmnopqrstuvwxyz "][seed % 5]
    n_terms = [1, 5, 40, 300, 1000, 17, 3000, 64][seed]
    maxlen = [3, 9, 9, 6, 5, 40, 12, 200][seed]
    terms = set()
    for _ in range(n_terms):
        L = int(rng.integers(1, maxlen + 1))
        terms.add(bytes(alpha[i] for i in rng.integers(0, len(alpha), L)))
    o = both(eng, sorted(terms), pos_mode)
    lens = [0, 1, 2, 7, 63, 64, 65, 300, 2000, 4095, 4096, 4097, 8191, 8449, 20000, 70000]
    texts = [bytes(alpha[i] for i in rng.integers(0, len(alpha), n)) for n in lens]
    # plant whole terms so long keywords occur too
    tl = sorted(terms)
    planted = b" ".join(tl[i] for i in rng.integers(0, len(tl), 400))
    texts.append(planted)
    blob, off = docs(texts)
    assert_csr_equal(eng.scan(blob, off), o.scan(blob, off))


def test_ascii_fold(eng):
    o = both(eng, ["lorem", "ipsum dolor", "x"])
    blob, off = docs(["LoReM IPSUM DOLOR sit amet X", "\xc3\x89LOREM\xff"])
    assert_csr_equal(eng.scan(blob, off, fold=True), o.scan(blob, off, fold=True))
    assert eng.scan(blob, off, fold=False)[1].size == 0


@pytest.mark.parametrize("n_terms,n_docs", [(50, 300), (1000, 2000), (10000, 1500)])
def test_workload_positions(eng, n_terms, n_docs):
    """BASELINE config 2 shape (1 k terms, ~4 KB docs, positions only) at oracle-friendly size."""
    from gofindthem_amd.workload import Workload
    w = Workload(n_terms)
    o = both(eng, w.terms())
    text, off = w.docs_host(0, n_docs)
    got = eng.scan(text, off)
    assert_csr_equal(got, o.scan(text, off))
    assert got[1].size > n_docs          # planted terms guarantee hits


def _programs(o, eng, exprs, cs, regexes=()):
    trees = [dsl_ref.parse(e, cs)[0] for e in exprs]
    extra = {}

    def slot_of(lit):
        t = eng.term_id(lit)
        if t >= 0:
            return t
        return eng.n_terms + extra.setdefault(lit, len(extra))
    progs = [tree_to_program(t, slot_of) for t in trees]
    return progs, extra


@pytest.mark.parametrize("inord", [0.0, 0.5])
def test_workload_process(eng, inord):
    """BASELINE configs 3/4 shape: expressions over the synthetic corpus, hit bitmap bit-exact."""
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(1000)
    terms = w.terms()
    exprs = make_expressions(terms, 200, inord_fraction=inord)
    o = both(eng, terms)
    o.set_expressions(exprs, case_sensitive=False)
    progs, extra = _programs(o, eng, exprs, False)
    assert not extra
    eng.set_programs(progs)
    text, off = w.docs_host(0, 400)
    got = eng.process(text, off, fold=True)
    want = o.process(text, off, fold=True)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    bits = int(np.unpackbits(want.view(np.uint8)).sum())
    assert 0 < bits < 400 * 200


def test_solver_fixtures_on_gpu(eng):
    """dsl/expression_test.go INORD tables: texts synthesised so that the engine yields the listed positions."""
    cases = [
        ('inord("a" and "b" and "c")', "acabXaXcb", True),
        ('inord("a" and ("b" or "c"))', "cbac", True),
        ('inord("a" and "b" and "c")', "bacb", False),
        ('inord("a" and "b" and "c")', "bcab", False),
        ('inord(("b" or "c") and ("a" or "b"))', "bcab", True),
        ('inord("b" and "c") and inord("a" and "b")', "bcab", True),
        ('not "a" or "q"', "", True),
        ('"a" and not ("b" or "c")', "a", True),
    ]
    o = both(eng, ["a", "b", "c", "q"])
    exprs = [c[0] for c in cases]
    o.set_expressions(exprs, True)
    progs, _ = _programs(o, eng, exprs, True)
    eng.set_programs(progs)
    blob, off = docs([c[1] for c in cases])
    got = eng.process(blob, off)
    want = o.process(blob, off)
    assert np.array_equal(got, want)
    for d, c in enumerate(cases):
        assert bool(got[d, 0] >> d & 1) is c[2], c


def test_not_over_a_one_leaf_inord_group(eng):
    """`not (inord("a"))`: a group of one leaf needs no position algebra (no INORD word is emitted), but the NOT on top of
    it survives the push-down to the leaves -- in a program set WITHOUT any other INORD group the solver must still take
    the variant that knows NOT words"""
    o = both(eng, ["a", "b"])
    exprs = ['not (inord("a"))', '"b" and not (inord("a"))', 'not (inord("a")) or "b"', '"a"']
    o.set_expressions(exprs, True)
    progs, _ = _programs(o, eng, exprs, True)
    eng.set_programs(progs)
    blob, off = docs(["a", "b", "ab", "", "xx"])
    got = eng.process(blob, off)
    assert np.array_equal(got, o.process(blob, off))
    assert [int(x) & 0xF for x in got[:, 0]] == [0b1000, 0b0111, 0b1100, 0b0101, 0b0101]


def _nested(rng, letters, depth):
    """an expression whose accumulator stack gets `depth` deep in the fused form: both operands of the operator are
    subtrees, the right one nests on"""
    def leafpair():
        a, b = rng.choice(letters, 2)
        return '(%s"%s" %s "%s")' % ("not " if rng.integers(3) == 0 else "", a, "and" if rng.integers(2) else "or", b)
    e = leafpair()
    for _ in range(depth):
        e = "(%s %s %s)" % (leafpair(), "and" if rng.integers(2) else "or", e)
        if rng.integers(4) == 0:
            e = "not " + e
    return e


@pytest.mark.parametrize("group_docs", [None, "16", "0"])
def test_nested_expressions_register_and_scratch_stacks(eng, monkeypatch, group_docs):
    """programs that nest 0..2 deep run on the two-register interpreter, 3..4 on the four-register one, deeper ones spill
    to scratch inside it (gft_solve.hip run_program); mixed in one program set, with and without INORD groups"""
    if group_docs:
        monkeypatch.setenv("GFT_SOLVE_GROUP_DOCS", group_docs)
    rng = np.random.default_rng(77)
    letters = list("abcdefgh")
    o = both(eng, letters)
    for with_inord in (False, True):
        exprs = [_nested(rng, letters, d) for d in (0, 1, 2, 3, 4, 5, 6, 9, 17, 40) for _ in range(12)]
        exprs += ['"a"', 'not "b"', '"a" and "b" or "c"']
        if with_inord:
            exprs += ['inord("a" and "b") and %s' % _nested(rng, letters, 5), 'not (inord("c" and ("d" or "a")))']
        order = rng.permutation(len(exprs))
        exprs = [exprs[i] for i in order]
        o.set_expressions(exprs, True)
        progs, _ = _programs(o, eng, exprs, True)
        eng.set_programs(progs)
        texts = ["".join(rng.choice(letters + ["x", "y"], int(rng.integers(0, 7)))) for _ in range(300)]
        blob, off = docs(texts)
        want = o.process(blob, off)
        assert np.array_equal(eng.process(blob, off), want)
        bits = int(np.unpackbits(want.view(np.uint8)).sum())
        assert 0 < bits < 300 * len(exprs)


def test_many_short_terms_overflow_records(eng):
    """more than 254 distinct short-term records: LDS ids run out and the overflow table in global memory takes over
    (scan2_tables.cpp short3_big); dense matches also shrink the work units (adaptive unit size)"""
    import itertools
    alpha = "abcdefghijklmnopqrst"
    terms = ["".join(p) for p in itertools.product(alpha, repeat=2)]                      # 400 two-letter terms
    terms += ["".join(p) for p in itertools.product(alpha[:9], repeat=3)]                   # 729 three-letter terms
    terms += ["abcde", "tsrqponm", "aaaa", "bcbcbcbcbcbcbcbcbcbcbcbcbcbcbcbc"]
    o = both(eng, sorted(set(t.encode() for t in terms)))
    rng = np.random.default_rng(5)
    texts = [bytes(ord(alpha[i]) for i in rng.integers(0, len(alpha), n)) for n in (0, 1, 2, 3, 5, 64, 700, 5000, 9000, 30000)]
    texts.append(b"bc" * 40 + b"abcde" + b"x" + b"tsrqponm")
    blob, off = docs(texts)
    assert_csr_equal(eng.scan(blob, off), o.scan(blob, off))
    # the process path (balanced, unordered kernel) twice: the second call runs with the unit size the first one learned
    exprs = ['"ab" and "abc"', 'inord("aaa" and "bcb")', 'not "tsrqponm"', '"abcde" or "aaaa"', 'inord("qr" and "st" and "ab")']
    o.set_expressions(exprs, True)
    progs, extra = _programs(o, eng, exprs, True)
    assert not extra
    eng.set_programs(progs)
    want = o.process(blob, off)
    assert np.array_equal(eng.process(blob, off), want)
    assert np.array_equal(eng.process(blob, off), want)


def test_deep_buckets_and_long_terms(eng):
    """hundreds of terms ending with the same four bytes (one bucket, far more entries than the deferred list holds) and
    terms longer than a slot's 24 inline bytes (term_blob compare)"""
    rng = np.random.default_rng(11)
    alpha = b"abcdefgh"
    terms = set()
    for _ in range(700):
        L = int(rng.integers(1, 40))
        terms.add(bytes(alpha[i] for i in rng.integers(0, 8, L)) + b"wxyz")
    terms |= {b"wxyz", b"awxyz", b"h" * 60 + b"wxyz", b"abcdefgh" * 6}
    tl = sorted(terms)
    o = both(eng, tl)
    planted = b"".join(tl[i] + b"-" for i in rng.integers(0, len(tl), 600))
    noise = bytes(alpha[i] for i in rng.integers(0, 8, 20000)).replace(b"aaa", b"wxyz")
    blob, off = docs([planted, noise, b"wxyz", b"xyz", b"h" * 59 + b"wxyz", b"h" * 61 + b"wxyz" + b"abcdefgh" * 7])
    assert_csr_equal(eng.scan(blob, off), o.scan(blob, off))
    exprs = ['"wxyz"', '"awxyz" and not "%s"' % ("h" * 60 + "wxyz"), 'inord("%s" and "wxyz")' % ("abcdefgh" * 6)]
    exprs += ['"%s"' % t.decode() for t in tl[:40]]
    o.set_expressions(exprs, True)
    progs, extra = _programs(o, eng, exprs, True)
    eng.set_programs(progs)
    assert np.array_equal(eng.process(blob, off), o.process(blob, off))


def test_table_export_import_roundtrip(eng, scan_kernel):
    """compiled tables as a blob -- a second engine that imports it behaves like the one that built
    them (same term ids, same matches, same expression results); damaged or foreign blobs are refused"""
    from gofindthem_amd.engine import Engine, GftError
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(2000)
    terms = w.terms()
    o = both(eng, terms)
    blob = eng.export_tables()
    assert len(blob) > 100_000
    e2 = Engine()
    try:
        e2.import_tables(blob)
        assert e2.terms() == eng.terms() and e2.n_states == eng.n_states
        text, off = w.docs_host(0, 300)
        assert_csr_equal(e2.scan(text, off, fold=True), o.scan(text, off, fold=True))
        exprs = make_expressions(terms, 120, inord_fraction=0.4)
        o.set_expressions(exprs, False)
        progs, extra = _programs(o, e2, exprs, False)
        e2.set_programs(progs)
        assert np.array_equal(e2.process(text, off, fold=True), o.process(text, off, fold=True))
        for bad in (blob[:-1], blob[:1000], b"GFTT" + blob[4:200], blob[:500] + bytes([blob[500] ^ 1]) + blob[501:], b""):
            with pytest.raises(GftError):
                e2.import_tables(bad)
        # a blob that is internally consistent as far as the checksum goes, but whose tables point outside themselves
        # (stale or crafted): every index-bearing table is validated before anything is uploaded
        import struct

        def resealed(b):
            h = 1469598103934665603
            for c in b[:-8]:
                h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            return b[:-8] + struct.pack("<Q", h)
        n_bad = 0
        for frac in (0.2, 0.35, 0.5, 0.65, 0.8, 0.9, 0.97):
            at = int(len(blob) * frac) & ~3
            forged = resealed(blob[:at] + b"\xff\xff\xff\x7f" + blob[at + 4:])
            try:
                e2.import_tables(forged)
            except GftError:
                n_bad += 1
                continue
            # accepted: then the word was not an index (text bytes, filter bits ...) and the scan must still be in bounds
            e2.scan(text, off, fold=True)
        assert n_bad >= 2
        # relations BETWEEN the tables: the suffix-window set's class count and class map must be the automaton's
        # (build_scan5_tables indexes its counters by the automaton's classes), a bucket key must be four classes, and a key
        # must sit in its own pair of the bucket table
        def kp_offset(b):
            at = 20
            n_terms = struct.unpack_from("<Q", b, at)[0]; at += 8
            for _ in range(n_terms):
                at += 8 + struct.unpack_from("<Q", b, at)[0]
            at += 4 + 256 + 4 + 4                          # n_classes, byte_class, n_states, max_term_len
            for width in (4, 4, 4, 4, 4, 4, 4, 1):         # delta, out_term, out_link, term_len, depth, fail, child_begin, in_class
                at += 8 + width * struct.unpack_from("<Q", b, at)[0]
            return at + 4                                  # behind Scan2Tables::supported
        at = kp_offset(blob)
        kp = struct.unpack_from("<I", blob, at)[0]
        assert 2 <= kp <= 64 and struct.unpack_from("<I", blob, at - 4)[0] == 1, "blob layout changed: adapt kp_offset"
        for forged_kp in (kp - 1, kp + 1, 2):
            with pytest.raises(GftError, match="inconsistent"):
                e2.import_tables(resealed(blob[:at] + struct.pack("<I", forged_kp) + blob[at + 4:]))
        e2.import_tables(blob)
        # a refused blob leaves the engine as it was
        with pytest.raises(GftError):
            e2.import_tables(blob[:-1])
        assert_csr_equal(e2.scan(text, off, fold=True), o.scan(text, off, fold=True))
    finally:
        e2.close()


def test_large_dictionary_spill_path(eng):
    """30 000 terms: more long terms than the LDS fingerprint table holds, so it moves to global memory
    (kScan2FptLdsItems; BASELINE configs[4] "LDS-tile spill path"); dense matches shrink the work units"""
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(30000)
    terms = w.terms()
    o = both(eng, terms)
    text, off = w.docs_host(0, 400)
    assert_csr_equal(eng.scan(text, off, fold=True), o.scan(text, off, fold=True))
    exprs = make_expressions(terms, 150, inord_fraction=0.4)
    o.set_expressions(exprs, False)
    progs, extra = _programs(o, eng, exprs, False)
    eng.set_programs(progs)
    want = o.process(text, off, fold=True)
    assert np.array_equal(eng.process(text, off, fold=True), want)
    assert np.array_equal(eng.process(text, off, fold=True), want)      # second call: adapted unit size


@pytest.mark.parametrize("group_docs", ["32", "16", "8", "0"])
def test_solver_group_widths(eng, monkeypatch, group_docs):
    """the solver's narrower document groups (picked when a large dictionary's presence matrix would not fit LDS at 64
    documents per group) and, with "0" (no width allowed), the HBM-resident presence matrix; ragged last group"""
    from gofindthem_amd.workload import Workload, make_expressions
    monkeypatch.setenv("GFT_SOLVE_GROUP_DOCS", group_docs)
    w = Workload(1000)
    terms = w.terms()
    exprs = make_expressions(terms, 200, inord_fraction=0.4)
    o = both(eng, terms)
    o.set_expressions(exprs, case_sensitive=False)
    progs, _ = _programs(o, eng, exprs, False)
    eng.set_programs(progs)
    text, off = w.docs_host(0, 403)
    want = o.process(text, off, fold=True)
    assert np.array_equal(eng.process(text, off, fold=True), want)
    assert np.array_equal(eng.process(text, off, fold=True), want)      # presence matrix wiped between groups and calls


@pytest.mark.parametrize("group_docs", [None, "16"])
def test_many_expressions_programs_in_global_memory(eng, monkeypatch, group_docs):
    """12 000 expressions: the fused programs exceed LDS, so the interpreter reads the per-block transposed copy from
    global memory; several output tiles per group"""
    from gofindthem_amd.workload import Workload, make_expressions
    if group_docs:
        monkeypatch.setenv("GFT_SOLVE_GROUP_DOCS", group_docs)
    w = Workload(2000)
    terms = w.terms()
    exprs = make_expressions(terms, 12000, inord_fraction=0.2)
    o = both(eng, terms)
    o.set_expressions(exprs, case_sensitive=False)
    progs, _ = _programs(o, eng, exprs, False)
    assert sum(len(p) for p in progs) * 4 > 160 * 1024
    eng.set_programs(progs)
    text, off = w.docs_host(0, 150)
    want = o.process(text, off, fold=True)
    assert np.array_equal(eng.process(text, off, fold=True), want)


@pytest.mark.parametrize("pos_mode", [POS_START, POS_END])
def test_reference_benchmark_shape_single_large_document(eng, pos_mode):
    """BASELINE configs[0]: ~50 terms, 3 expressions and ONE document of ~100 000 words (~1 MB), the
    shape benchmarks/benchmark_test.go:66,273,287,449-460 measures; the document spans hundreds of work units, so
    matches across unit borders, the per-document merge of the units and INORD over 32-bit positions are all exercised.
    Second batch: the same text cut into empty, large and small documents."""
    from gofindthem_amd.workload import Workload
    w = Workload(50)
    terms = w.terms()
    text, off = w.docs_host(0, 250)
    assert 900_000 < int(off[-1]) < 1_300_000
    o = both(eng, terms, pos_mode)
    t = [x.decode() for x in terms]
    exprs = ['"%s" and "%s"' % (t[0], t[1]),
             'INORD("%s" and "%s") and INORD("%s" and "%s")' % (t[2], t[3], t[3], t[2]),
             "INORD(" + " and ".join('"%s"' % x for x in t[4:49]) + ")"]
    o.set_expressions(exprs, case_sensitive=False)
    progs, extra = _programs(o, eng, exprs, False)
    assert not extra
    eng.set_programs(progs)
    one = np.array([0, off[-1]], dtype=np.uint64)
    cut = np.array([0, 0, off[100], off[100], off[101], off[-1], off[-1]], dtype=np.uint64)
    for offsets in (one, cut):
        want = o.scan(text, offsets, fold=True)
        assert_csr_equal(eng.scan(text, offsets, fold=True), want)
        assert np.array_equal(eng.process(text, offsets, fold=True), o.process(text, offsets, fold=True))
    assert int(o.process(text, one, fold=True)[0, 0]) & 3 == 3          # the plain AND and both orders: all true on 1 MB


def test_descending_offsets_are_refused_on_the_device_unit_path(eng):
    """more documents than the host-side unit table handles (kHostUnitDocs): the unit kernel flags offsets that descend
    (the same flag refuses documents of 4 GiB and more, whose positions would not fit 32 bits)"""
    from gofindthem_amd.engine import GftError
    both(eng, [b"ab", b"abc"])
    blob, off = docs(["xxabcxx"] * 3000)
    eng.scan(blob, off)
    bad = off.copy()
    bad[1500] = bad[1501] + 3
    with pytest.raises(GftError):
        eng.scan(blob, bad)
    assert_csr_equal(eng.scan(blob, off), Oracle([b"ab", b"abc"], POS_START).scan(blob, off))


def test_unique_terms_mode_is_the_cloudflare_engine_output(eng):
    """GFT_SCAN_UNIQUE = CloudflareEngine.FindSubstrings (finder/substringEngine.go:77-86): Matcher.Match reports every
    dictionary term that occurs in the text once, Position 0.  Expected from the oracle's full match list: first
    occurrences in emission order."""
    from gofindthem_amd.workload import Workload
    o = both(eng, ["he", "she", "his", "hers"])
    blob, off = docs(["ushers", "", "hishishershe", "xx"])
    mo, ti, po = eng.scan(blob, off, unique=True)
    assert mo.tolist() == [0, 3, 3, 7, 7] and not po.any()
    assert ti.tolist() == [3, 0, 1, 2, 3, 0, 1]          # she he hers | his she he hers (ids: he 0, hers 1, his 2, she 3)
    w = Workload(3000)
    o = both(eng, w.terms())
    text, toff = w.docs_host(0, 700)
    mo, ti, po = eng.scan(text, toff, fold=True, unique=True)
    wo, wt, _ = o.scan(text, toff, fold=True)
    assert not po.any()
    for d in range(700):
        full = wt[int(wo[d]):int(wo[d + 1])].tolist()
        first = list(dict.fromkeys(full))
        assert ti[int(mo[d]):int(mo[d + 1])].tolist() == first, d
    # device-resident variant
    import torch
    t = torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()
    oo = torch.from_numpy(toff.astype(np.int64)).cuda()
    m = eng.scan_device(t.data_ptr(), oo.data_ptr(), 700, fold=True, unique=True)
    assert int(m.n_matches) == int(mo[-1])


@needs_extra_kernels
def test_streaming_kernel_chunks_of_eight_units_and_regions_that_overflow(monkeypatch):
    """gft_scan4 on small batches takes one unit per chunk; GFT_SCAN4_CHUNK=8 forces the production shape (eight units per
    chunk: documents that share a stream, slices of long documents, empty documents in between).  A fresh engine sizes a unit's
    region of the match pool for 0.06 matches per byte: a dictionary that matches at every position overflows ALL eight
    regions of every chunk -- each unit is then walked again on its own (the list of such units holds one chunk's worth)."""
    from gofindthem_amd.engine import Engine
    monkeypatch.setenv("GFT_SCAN_KERNEL", "scan4")
    monkeypatch.setenv("GFT_SCAN4_CHUNK", "8")
    rng = np.random.default_rng(5)
    e = Engine()
    try:
        terms = [b"a", b"b", b"ab", b"ba", b"aab", b"abab", b"bbbb", b"abba", b"aaaaa", b"babab"]
        e.build(terms)
        o = Oracle(terms)
        lens = [5000, 0, 3, 4100, 70000, 1, 0, 0, 4096, 4097, 8176, 8177, 20000, 2, 300] * 3
        texts = [bytes(b"ab"[i] for i in rng.integers(0, 2, n)) for n in lens]
        blob, off = docs(texts)
        for _ in range(2):                                   # (second call: the regions follow the density the first one saw)
            assert_csr_equal(e.scan(blob, off), o.scan(blob, off))
        from gofindthem_amd.workload import Workload
        w = Workload(3000)
        o = Oracle(w.terms())
        e.build(w.terms())
        text, toff = w.docs_host(0, 300)
        assert_csr_equal(e.scan(text, toff, fold=True), o.scan(text, toff, fold=True))
    finally:
        e.close()


def test_two_positions_per_probe_kernel_second_walks_and_merged_groups(monkeypatch):
    """gft_scan5: one filter probe per two bytes over 3-grams of merged byte classes.  Documents of every length around the
    lane and piece borders (a lane owns a multiple of four bytes, a probe pair never leaves its dword), slices of long
    documents, empty documents; a dictionary that matches at every position outgrows the 256-entry fifo in every unit --
    each is then walked a second time straight into a pool region of the counted size; GFT_SCAN5_GROUPS=4 merges the byte
    classes far beyond what LDS asks for (the filter then flags nearly everything: exactness must come from the stages
    behind it).  Both position conventions, positions packed into the fifo entry next to the term id."""
    from gofindthem_amd.engine import Engine
    from gofindthem_amd import _lib
    monkeypatch.setenv("GFT_SCAN_KERNEL", "scan5")
    rng = np.random.default_rng(11)
    e = Engine()
    try:
        for groups in (None, "4"):
            if groups:
                monkeypatch.setenv("GFT_SCAN5_GROUPS", groups)
            terms = [b"a", b"b", b"ab", b"ba", b"aab", b"abab", b"bbbb", b"abba", b"aaaaa", b"babab", b"ab" * 20 + b"b"]
            lens = [5000, 0, 3, 4100, 70000, 1, 0, 0, 1024, 1025, 5120, 5121, 20000, 2, 300, 31, 33, 1023, 2048, 4096] * 2
            texts = [bytes(b"ab"[i] for i in rng.integers(0, 2, n)) for n in lens]
            texts[0] = b"abab" + texts[0]
            blob, off = docs(texts)
            for pos_mode in (POS_START, POS_END):
                e.build(terms, pos_end=(pos_mode == POS_END))
                assert _lib.load().gft_scan_kernel(e._h).decode() == "scan5"
                o = Oracle(terms, pos_mode)
                for _ in range(2):                               # (second call: unit sizes follow the density the first one saw)
                    assert_csr_equal(e.scan(blob, off), o.scan(blob, off))
            from gofindthem_amd.workload import Workload
            w = Workload(3000)
            o = Oracle(w.terms())
            e.build(w.terms())
            text, toff = w.docs_host(0, 300)
            assert_csr_equal(e.scan(text, toff, fold=True), o.scan(text, toff, fold=True))
            assert_csr_equal(e.scan(text, toff, fold=False), o.scan(text, toff, fold=False))
    finally:
        e.close()


def test_rune_offsets_are_the_anknown_engine_positions(eng):
    """GFT_POS_RUNES = AnknownEngine.FindSubstrings (finder/substringEngine.go:44-53): MultiPatternSearch([]rune(text)) reports
    Position over runes.  Expected: the oracle's byte offsets mapped through Go's string -> []rune decoding
    (oracle/runes_ref.py: multi-byte runes count once, every invalid byte is one U+FFFD)."""
    from oracle.runes_ref import rune_index_table
    terms = ["lo", "w\u00f6rld", "\u20acuro", "x", "\U0001d11e", "\u00e9", "rld"]
    o = both(eng, sorted(t.encode("utf-8") for t in terms))
    texts = ["h\u00e9llo w\u00f6rld \u20acuro \U0001d11e x lo", "", "x",
             b"\x80x\xc3 lo \xe2\x82 x \xed\xa0\x80lo \xf5x \xc0\x80 x\xf0\x9f lo".decode("latin-1"),     # invalid sequences of every kind
             ("\u00e9" * 40 + "x") * 30, "\U0001d11e" * 100 + "lo"]
    raw = [t.encode("latin-1") if i == 3 else t.encode("utf-8") for i, t in enumerate(texts)]
    blob = np.frombuffer(b"".join(raw), dtype=np.uint8)
    off = np.zeros(len(raw) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in raw])
    wo, wt, wp = o.scan(blob, off)
    want = wp.copy()
    for d, r in enumerate(raw):
        tab = rune_index_table(r)
        for m in range(int(wo[d]), int(wo[d + 1])):
            want[m] = tab[int(wp[m])]
    mo, ti, po = eng.scan(blob, off, runes=True)
    assert np.array_equal(mo, wo) and np.array_equal(ti, wt)
    assert np.array_equal(po, want), (po[po != want][:8], want[po != want][:8])
    assert (want != wp).any()                                        # (the mapping did something)
    # a longer document: more than one counted block per document, blocks that end inside a rune
    from gofindthem_amd.workload import Workload
    w = Workload(300, alphabet="mixed")
    kws = sorted({t.decode("utf-8").lower().encode("utf-8") for t in w.terms()})
    o = both(eng, kws)
    text, toff = w.docs_host(0, 40)
    wo, wt, wp = o.scan(text, toff)
    want = wp.copy()
    for d in range(40):
        tab = rune_index_table(bytes(text[int(toff[d]):int(toff[d + 1])]))
        for m in range(int(wo[d]), int(wo[d + 1])):
            want[m] = tab[int(wp[m])]
    mo, ti, po = eng.scan(text, toff, runes=True)
    assert np.array_equal(po, want) and np.array_equal(ti, wt)


def test_mixed_alphabet_workload(eng, scan_kernel):
    """a real word list's shape (benchmarks/benchmark_test.go:72-83): capitals, digits, punctuation, two-byte UTF-8
    letters -- more than 48 byte classes after folding, 2- and 3-byte terms included.  The round-1 kernel refuses such a
    dictionary; the stride-2 kernel merges byte classes into filter groups, the two-positions-per-probe kernel does the same
    for its filter and takes the short terms from the stride-2 kernel's group-indexed tables: both must stay bit-exact."""
    from gofindthem_amd import _lib
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(3000, alphabet="mixed")
    kws = sorted({t.decode("utf-8").lower().encode("utf-8") for t in w.terms()})      # the finder's keyword set
    assert len({b for t in kws for b in t}) >= 48 and min(len(t) for t in kws) <= 3
    o = both(eng, kws)
    L = _lib.load()
    assert L.gft_scan_kernel(eng._h).decode() == {"scan5": "scan5", "scan4": "dfa", "scan3": "scan3", "scan2": "dfa", "scan2-ordered": "dfa", "dfa": "dfa"}[scan_kernel]
    text, off = w.docs_host(0, 400)
    assert_csr_equal(eng.scan(text, off, fold=True), o.scan(text, off, fold=True))
    assert_csr_equal(eng.scan(text, off, fold=False), o.scan(text, off, fold=False))
    exprs = make_expressions(kws, 150, inord_fraction=0.4)
    o.set_expressions(exprs, case_sensitive=False)
    progs, _ = _programs(o, eng, exprs, False)
    eng.set_programs(progs)
    assert np.array_equal(eng.process(text, off, fold=True), o.process(text, off, fold=True))
    # this corpus leaves ASCII, but only with lower-case Latin-1 letters: ASCII folding is still strings.ToLower
    assert L.gft_last_nonascii(eng._h) == 0
    bad = np.frombuffer("Ünsafe: an upper-case letter beyond ASCII".encode("utf-8"), dtype=np.uint8)
    eng.process(bad, np.asarray([0, bad.size], np.uint64), fold=True)
    assert L.gft_last_nonascii(eng._h) == 1


def test_large_alphabet_on_a_fresh_engine_and_through_import(scan_kernel, monkeypatch):
    """the two-positions-per-probe kernel on a dictionary over more than 32 byte classes, on engines that never held another
    dictionary (the launch geometry must not lean on what an earlier build left behind) and on one that IMPORTS the tables
    (the long-term tables travel; the importer rebuilds what the blob marks as unsupported for the round-1 kernel);
    GFT_SCAN5_LARGE=0 keeps such a dictionary on the stride-2 kernel"""
    if scan_kernel != "scan5":
        pytest.skip("one kernel's business")
    from gofindthem_amd import _lib
    from gofindthem_amd.engine import Engine
    from gofindthem_amd.workload import Workload
    L = _lib.load()
    w = Workload(3000, alphabet="mixed")
    kws = sorted({t.decode("utf-8").lower().encode("utf-8") for t in w.terms()})
    o = Oracle(kws, POS_START)
    text, off = w.docs_host(0, 2000)
    want = o.scan(text, off, fold=True)
    e1, e2, e3 = Engine(), Engine(), Engine()
    try:
        e1.build(kws)
        assert L.gft_scan_kernel(e1._h).decode() == "scan5"
        assert_csr_equal(e1.scan(text, off, fold=True), want)
        e2.import_tables(e1.export_tables())
        assert L.gft_scan_kernel(e2._h).decode() == "scan5" and e2.terms() == e1.terms()
        assert_csr_equal(e2.scan(text, off, fold=True), want)
        monkeypatch.setenv("GFT_SCAN5_LARGE", "0")
        e3.build(kws)
        assert L.gft_scan_kernel(e3._h).decode() == "scan3"
        assert_csr_equal(e3.scan(text, off, fold=True), want)
    finally:
        e1.close(), e2.close(), e3.close()


def test_bloom_level_in_front_of_a_global_fingerprint_table(scan_kernel, monkeypatch):
    """a dictionary whose fingerprint table does not fit LDS (here forced: GFT_SCAN_FPT_GLOBAL) gets a Bloom level in LDS in
    front of it (gft_kernels.hpp scan5_bloom_g / _x): positions whose bits are clear skip the table's three L2 gathers.
    Terms whose anchor window is their first four bytes are keyed by the window alone, all others by (window, byte in
    front); a tiny Bloom (GFT_SCAN5_BLOOM_KB=1: 8 192 bits, nearly full at 3 000 terms) and none at all must give the same
    matches as the oracle, folded and exact"""
    if scan_kernel != "scan5":
        pytest.skip("one kernel's business")
    from gofindthem_amd.engine import Engine
    from gofindthem_amd.workload import Workload
    monkeypatch.setenv("GFT_SCAN_FPT_GLOBAL", "1")
    w = Workload(3000)
    terms = w.terms()
    text, off = w.docs_host(0, 600)
    o = Oracle(terms, POS_START)
    want_f, want_x = o.scan(text, off, fold=True), o.scan(text, off, fold=False)
    rng = np.random.default_rng(5)
    alpha = b"abcdefgh"
    rterms = sorted({bytes(alpha[i] for i in rng.integers(0, len(alpha), int(rng.integers(1, 9)))) for _ in range(800)})
    rtexts = [bytes(alpha[i] for i in rng.integers(0, len(alpha), n)) for n in (0, 3, 4, 5, 64, 1000, 9000, 30000)]
    rblob, roff = docs(rtexts)
    ro = Oracle(rterms, POS_END)
    for kb in ("1", "32", "0"):
        monkeypatch.setenv("GFT_SCAN5_BLOOM_KB", kb)
        e = Engine()
        try:
            e.build(terms)
            assert_csr_equal(e.scan(text, off, fold=True), want_f)
            assert_csr_equal(e.scan(text, off, fold=False), want_x)
            e.build(rterms, pos_end=True)
            assert_csr_equal(e.scan(rblob, roff), ro.scan(rblob, roff))
        finally:
            e.close()


def _fold_safe_doc(b: bytes) -> bool:
    """the rule of gft_foldsafe_dev.hpp / k_fold_safe for ONE document: ASCII, C2 80..BF, C3 9F..BF and C3 97 only"""
    i = 0
    while i < len(b):
        c = b[i]
        if c < 0x80:
            i += 1
        elif c in (0xC2, 0xC3) and i + 1 < len(b) and 0x80 <= b[i + 1] <= 0xBF and (c == 0xC2 or b[i + 1] >= 0x9F or b[i + 1] == 0x97):
            i += 2
        else:
            return False
    return True


def test_fold_safety_is_decided_per_document(eng, scan_kernel):
    """gft_last_nonascii after a folded scan = some document is not "ASCII + lower-case Latin-1" (then ASCII folding is not
    strings.ToLower, finder.go:140-142).  The suffix-window kernels decide it while they scan (the pieces of text that
    hold high bytes, 64 per trip), the others in a second pass over the blob: documents built around the rule's edges --
    a lead byte as a document's last byte, a continuation byte as its first, pairs that straddle lanes, pieces, units and
    documents, upper-case Latin-1, three-byte sequences, lone high bytes -- against the rule applied document by document;
    every safe batch must also still be a safe batch when it is not the first one."""
    from gofindthem_amd import _lib
    L = _lib.load()
    both(eng, [b"abc", b"b", b"caf\xc3\xa9", b"stra\xc3\x9fe"])
    rng = np.random.default_rng(23)
    safe_units = [b"a", b"b c", b"\xc3\xa9", b"\xc2\xa0", b"\xc3\x9f", b"\xc3\x97", b"\xc3\xbf", b"xyz ", b"\xc2\x80", b"\xc2\xbf"]
    bad_units = [b"\xc3\x89", b"\xc3\x80", b"\xc3\x9e", b"\xe2\x84\xaa", b"\xc3", b"\xa9", b"\xc4\xb0", b"\xff", b"\xc3a", b"\xc2\xc2\xa0", b"\xc1\x81"]

    def doc(n_units, bad_at=None):
        parts = [safe_units[int(k)] for k in rng.integers(0, len(safe_units), n_units)]
        if bad_at is not None:
            parts.insert(bad_at, bad_units[int(rng.integers(len(bad_units)))])
        return b"".join(parts)

    cases = []
    for trial in range(40):
        n_docs = int(rng.integers(1, 40))
        lens = rng.choice([0, 1, 3, 17, 300, 1500, 5000], n_docs)
        batch = [doc(int(n)) for n in lens]
        kind = trial % 4
        if kind == 1:                                   # one unsafe unit somewhere inside one document
            d = int(rng.integers(n_docs))
            batch[d] = doc(int(lens[d]) + 1, bad_at=int(rng.integers(int(lens[d]) + 1)))
        elif kind == 2:                                 # a pair cut by a document border: lead at the end, continuation at the start
            d = int(rng.integers(n_docs))
            batch[d] = batch[d] + b"\xc3"
            if d + 1 < n_docs:
                batch[d + 1] = b"\xa9" + batch[d + 1]
        elif kind == 3:                                 # safe text that ends and starts with whole two-byte letters
            batch = [b"\xc3\xa9" + x + b"\xc2\xa0" for x in batch]
        cases.append(batch)
    seen = set()
    for batch in cases:
        blob, off = docs(batch)
        want = 0 if all(_fold_safe_doc(x) for x in batch) else 1
        eng.scan(blob, off, fold=True)
        assert L.gft_last_nonascii(eng._h) == want, (scan_kernel, [x[-4:] for x in batch][:6])
        seen.add(want)
        eng.scan(blob, off, fold=False)                 # (case-sensitive scans never ask)
        assert L.gft_last_nonascii(eng._h) == 0
    assert seen == {0, 1}


_C5_REF = {}      # the oracle's side of configs[4], computed by the first parametrisation and shared by the others


def test_config5_100k_terms_with_regex_leaves(scan_kernel):
    """BASELINE configs[4]: 100 000 terms (the large automaton: second-level filter and tables spill from LDS to L2, ~1 200
    matches per document, the solver's presence matrix at 8 documents per group) + r"..." regex terms through the host
    RegexpEngine (finder/regexEngine.go:36-47), CSR and bitmap against the oracle."""
    from gofindthem_amd.engine import Engine
    from gofindthem_amd.finder import Finder, GpuEngine, PyRegexpEngine
    from gofindthem_amd.workload import Workload, make_expressions
    R = _C5_REF
    if "terms" not in R:
        w = Workload(100_000)
        R["terms"] = w.terms()
        assert len(R["terms"]) == 100_000
        R["text"], R["off"] = w.docs_host(0, 260)
        o = Oracle(R["terms"])
        R["n_states"] = o.n_states
        R["csr"] = o.scan(R["text"], R["off"], fold=True)
        # the finder: > 90 k keywords in 3 000 expressions (INORD included), 16 regexes of the benchmark's r"wA.*wB" shape
        rx = ["%s.*%s" % (R["terms"][7 * i + 1].decode(), R["terms"][11 * i + 5].decode()) for i in range(16)]
        R["exprs"] = make_expressions(R["terms"], 3000, inord_fraction=0.4, regexes=rx, cover=True)
    terms, text, off, want, exprs = R["terms"], R["text"], R["off"], R["csr"], R["exprs"]
    e = Engine()
    try:
        e.build(terms)
        assert e.n_states == R["n_states"]
        got = e.scan(text, off, fold=True)
        assert_csr_equal(got, want)
        assert want[1].size > 1000 * 260                  # > 1 000 matches per document
        mo, ti, _ = e.scan(text, off, fold=True, unique=True)
        assert ti[:int(mo[1])].tolist() == list(dict.fromkeys(want[1][:int(want[0][1])].tolist()))
    finally:
        e.close()
    f = Finder(GpuEngine(), PyRegexpEngine(), False)
    try:
        f.AddExpressions(exprs)
        assert len(f.GetKeywords()) > 90_000 and len(f.GetRegexes()) == 16      # (one leaf in eight is a regex)
        bm = f.ProcessTexts(blob=text, doc_off=off)
        kws, rgx = sorted(f.GetKeywords()), sorted(f.GetRegexes())
        if "bitmap" not in R:
            o2 = Oracle(kws)
            o2.set_expressions(exprs, False)
            reng = PyRegexpEngine()
            reng.BuildEngine(rgx, False)
            offs, lits, poss = [0], [], []
            for d in range(260):
                t = bytes(text[int(off[d]):int(off[d + 1])])
                for m in reng.FindRegexes(t):
                    lits.append(o2.literals.index(m.Term))
                    poss.append(m.Position)
                offs.append(len(lits))
            extra = (np.asarray(offs, np.uint64), np.asarray(lits or [0], np.int32), np.asarray(poss or [0], np.int64))
            R["bitmap"] = o2.process(text, off, fold=True, extra=extra)
            R["kws"], R["rgx"] = kws, rgx
        assert (kws, rgx) == (R["kws"], R["rgx"])
        want = R["bitmap"]
        assert np.array_equal(bm, want) and bm.any()
    finally:
        f.close()
