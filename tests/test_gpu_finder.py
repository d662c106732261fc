"""finder.Finder mirror on the GPU, written after the reference's own tests (finder/finder_test.go,
group/finder/finder_test.go) and the examples/finder program; every solve runs in the HIP solver kernel."""
import numpy as np
import pytest

from conftest import load_golden
from gofindthem_amd import _lib
from gofindthem_amd.finder import (EmptyEngine, EmptyRgxEngine, Finder, FinderError, GpuEngine, Match, PyRegexpEngine)
from oracle.pyoracle import Oracle, pack_strings
from test_host_logic import make_mocked_finder

pytestmark = pytest.mark.gpu


# ---- finder/finder_test.go: TestProcessText (mocked engines) ---------------------------------------------------
@pytest.mark.parametrize("case", load_golden("process_text.json")["cases"], ids=lambda c: c["message"])
def test_process_text(case):
    f, sub, rgx = make_mocked_finder(case, allow_no_device=False)
    text = load_golden("process_text.json")["text"]
    if case["expectedErr"]:
        with pytest.raises(FinderError) as ei:
            f.ProcessText(text)
        assert str(ei.value) == case["expectedErr"]
        return
    res = f.ProcessText(text)
    assert [r.to_obj() for r in res] == case["expected"]
    # lazy build exactly when the dirty flag says so (finder.go:147-153,163-169)
    assert ("BuildEngine", case["keywords"]) in sub.calls or case["updatedSub"]
    assert sum(1 for c in sub.calls if c[0] == "BuildEngine") == (0 if case["updatedSub"] else 1)
    assert sum(1 for c in rgx.calls if c[0] == "BuildEngine") == (0 if case["updatedRgx"] else 1)
    assert ("FindSubstrings", text) in sub.calls and ("FindRegexes", text) in rgx.calls


# ---- finder/finder_test.go: TestAddMatchesToSolverMap + TestSolveExpressions, through ProcessText ----------------
class Fixed:
    def __init__(self, matches):
        self.matches = matches

    def BuildEngine(self, kws, cs):
        return None

    def FindSubstrings(self, text):
        return self.matches

    def FindRegexes(self, text):      # as a RegexEngine: report only the terms the finder registered as regexes
        return [m for m in self.matches if m.Term in self.regexes]


@pytest.mark.parametrize("case", load_golden("add_matches.json")["cases"], ids=lambda c: c["message"])
def test_add_matches_case_folding(case):
    """Term is lower-cased into the map key when case-insensitive (finder.go:186-188): "Showman" then feeds
    the same key as "showman"."""
    ms = [Match(m["Position"], m["Term"]) for m in case["matches"]]
    f = Finder(Fixed(ms), EmptyRgxEngine(), case["caseSensitive"])
    keys = sorted(case["expected"])
    for k in keys:
        f.AddExpression('"%s"' % k)
    f.AddExpression('"Showman"')
    f.AddExpression('inord("sharpest" and "words" and "showman")')
    got = [r.ExpresionIndex for r in f.ProcessText("irrelevant")]
    want = list(range(len(keys)))                 # every key of the expected map is present
    if case["caseSensitive"]:
        want.append(len(keys))                    # "Showman" stays its own key
    else:
        want.append(len(keys))                    # '"Showman"' was lower-cased by the parser -> key "showman"
    want.append(len(keys) + 1)                    # positions 1 < 7 < 9|10 are in order
    assert got == want


def test_solve_expressions():
    g = load_golden("solve_expressions.json")
    for c in g["cases"]:
        ms = [Match(0, k) for k in c["map"]]
        f = Finder(Fixed(ms), EmptyRgxEngine(), True)
        for e in g["expressions"]:
            f.AddExpressionWithTag(e["exprString"], e["tag"])
        assert [r.to_obj() for r in f.ProcessText("x")] == c["expected"], c["message"]


# ---- dsl/expression_test.go: all 31 Solve cases, map supplied through a fixed engine ----------------------------
@pytest.mark.parametrize("case", load_golden("solver.json")["cases"], ids=lambda c: c["message"])
def test_solver_table(case):
    """Keys with nil/empty position lists (expression_test.go:29-33) cannot come out of an engine -- an engine
    reports a key by reporting a match -- so each listed key gets its listed positions, or one position when the
    table says nil (truth of non-INORD expressions depends on key presence only, expression.go:68-72)."""
    ms = []
    for k, v in case["map"].items():
        for p in (v or [0]):
            ms.append(Match(p, k))
    ms.sort(key=lambda m: m.Position)
    sub, rgx = Fixed(ms), Fixed(ms)
    f = Finder(sub, rgx, True)
    f.AddExpression(case["expStr"])
    rgx.regexes = f.GetRegexes()
    sub.matches = [m for m in ms if m.Term not in rgx.regexes]
    res = f.ProcessText("x")
    assert bool(res) is case["expected"]


# ---- group/finder/finder_test.go: the real-engine cases -------------------------------------------------------
def test_engine_truth():
    for c in load_golden("engine_truth.json")["cases"]:
        f = Finder(GpuEngine(), EmptyRgxEngine(), c["caseSensitive"])
        f.AddExpression(c["expression"])
        assert bool(f.ProcessText(c["text"])) is c["expected_true"], c["text"]
        assert bool(f.ProcessText(c["text"].upper())) is c["expected_true"]      # case-insensitive finder


# ---- examples/finder/main.go -------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["case_sensitive", "case_insensitive"])
def test_examples_finder(which):
    g = load_golden("examples.json")
    sec = g[which]
    cs = which == "case_sensitive"
    f = Finder(GpuEngine(), PyRegexpEngine(), cs)
    for e, tag in sec["expressions"]:
        f.AddExpressionWithTag(e, tag)
    for text, want in zip(g["texts"], sec["expected_true"]):
        res = f.ProcessText(text)
        assert [r.ExpresionIndex for r in res] == want
        assert [r.Tag for r in res] == [sec["expressions"][i][1] for i in want]
        assert [r.ExpresionStr for r in res] == [sec["expressions"][i][0] for i in want]
    # batch extension agrees with the per-document calls
    bm = f.ProcessTexts(g["texts"])
    for d, want in enumerate(sec["expected_true"]):
        assert [i for i in range(len(sec["expressions"])) if bm[d, i >> 5] >> (i & 31) & 1] == want


def test_gpu_engine_find_substrings():
    e = GpuEngine()
    e.BuildEngine({"he", "she", "his", "hers"}, True)
    assert e.FindSubstrings("ushers") == [Match(1, "she"), Match(2, "he"), Match(2, "hers")]
    assert e.FindSubstrings("") == []


def test_solve_error_is_returned_like_the_reference():
    """`"a" "b" and "c"` parses (parser.go:220-233) but Solve fails on its UNSET node for every document
    (expression.go:139-141); ProcessText returns that error after the engine calls."""
    f = Finder(GpuEngine(), EmptyRgxEngine(), True)
    f.AddExpression('"x"')
    f.AddExpression('"a" "b" and "c"')
    with pytest.raises(FinderError) as ei:
        f.ProcessText("abc")
    assert str(ei.value) == "unable to process expression type 0"
    o = Oracle(["a", "b", "c", "x"])
    o.set_expressions(['"x"', '"a" "b" and "c"'], True)
    blob, off = pack_strings(["abc"])
    with pytest.raises(RuntimeError, match="unable to process expression type 0"):
        o.process(blob, off)


def test_unicode_case_folding_end_to_end():
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpression('"STRAßE" and inord("ÉCOLE" and "Ünïcode")')
    f.AddExpression('"ecole"')
    res = f.ProcessText("École de la Straße, ÜNÏCODE")
    assert [r.ExpresionIndex for r in res] == [0]
    bm = f.ProcessTexts(["École de la Straße, ÜNÏCODE", "ecole", "ÉCOLE ÜNÏCODE"])
    assert bm[:, 0].tolist() == [1, 2, 0]


def test_process_texts_matches_oracle_with_regex_terms():
    from gofindthem_amd.workload import Workload, make_expressions
    from oracle import dsl_ref
    w = Workload(300)
    terms = w.terms()
    rx = ["en.*nr", "po[a-z]+ud", "q+"]
    exprs = make_expressions(terms, 120, inord_fraction=0.4, regexes=rx)
    f = Finder(GpuEngine(), PyRegexpEngine(), False)
    f.AddExpressions(exprs)
    text, off = w.docs_host(0, 60)
    bm = f.ProcessTexts(blob=text, doc_off=off)
    # oracle side: same keyword set, regex hits computed by the same host regex stand-in
    kws = sorted(f.GetKeywords())
    o = Oracle(kws)
    o.set_expressions(exprs, False)
    eng = PyRegexpEngine()
    eng.BuildEngine(sorted(f.GetRegexes()), False)
    offs, lits, poss = [0], [], []
    for d in range(60):
        t = bytes(text[int(off[d]):int(off[d + 1])])
        for m in eng.FindRegexes(t):
            lits.append(o.literals.index(m.Term))
            poss.append(m.Position)
        offs.append(len(lits))
    extra = (np.asarray(offs, np.uint64), np.asarray(lits or [0], np.int32), np.asarray(poss or [0], np.int64))
    want = o.process(text, off, fold=True, extra=extra)
    assert np.array_equal(bm, want)
    assert any(len(x) for x in (lits,))


def _oracle_with_regex(f, exprs, text, off, n):
    """expected bitmap: CPU oracle + the host regex stand-in on EVERY document (what the reference does)"""
    kws = sorted(f.GetKeywords())
    o = Oracle(kws)
    o.set_expressions(exprs, False)
    eng = PyRegexpEngine()
    eng.BuildEngine(sorted(f.GetRegexes()), False)
    offs, lits, poss = [0], [], []
    for d in range(n):
        t = bytes(text[int(off[d]):int(off[d + 1])])
        for m in eng.FindRegexes(t):
            lits.append(o.literals.index(m.Term))
            poss.append(m.Position)
        offs.append(len(lits))
    extra = (np.asarray(offs, np.uint64), np.asarray(lits or [0], np.int32), np.asarray(poss or [0], np.int64))
    return o.process(text, off, fold=True, extra=extra), len(set(np.searchsorted(offs, range(len(lits)), side="right")))


@pytest.mark.parametrize("prefilter", ["1", "0"])
def test_regex_prefilter_same_results_fewer_regex_calls(prefilter, monkeypatch):
    """with the GPU engine, regexes whose matches must contain literal runs get hidden
    AND-of-literals programs; the host regex engine then only sees the documents where one fired.  Results are those of
    running it everywhere (the oracle side does exactly that); GFT_REGEX_PREFILTER=0 is the everywhere path."""
    from gofindthem_amd.workload import Workload, make_expressions
    monkeypatch.setenv("GFT_REGEX_PREFILTER", prefilter)
    w = Workload(300)
    terms = w.terms()
    text, off = w.docs_host(0, 200)
    sample = bytes(text[:int(off[40])]).decode("ascii").split()
    rng = np.random.default_rng(3)
    words = [x for x in sample if len(x) >= 6]
    rx = []
    for _ in range(6):                              # wA.*wB (README.md:14-15 shape) from words that do occur, plus rarer shapes
        a, b = words[int(rng.integers(len(words)))], words[int(rng.integers(len(words)))]
        rx.append("%s.*%s" % (a[:4], b[-4:]))
    rx += ["%s[a-z]+%s" % (words[5][:3], words[5][-2:]), "zzqq.*never", "%s[ ]+%s" % (sample[10], sample[11])]
    exprs = make_expressions(terms, 80, inord_fraction=0.3, regexes=rx)
    f = Finder(GpuEngine(), PyRegexpEngine(), False)
    f.AddExpressions(exprs)
    assert sorted(f.GetRegexes()) == sorted(set(rx))
    bm = f.ProcessTexts(blob=text, doc_off=off)
    want, docs_with_hits = _oracle_with_regex(f, exprs, text, off, 200)
    assert np.array_equal(bm, want)
    assert docs_with_hits > 0
    seen = int(_lib.load().gft_finder_last_regex_docs(f._h))
    if prefilter == "1":
        assert docs_with_hits <= seen < 200          # candidates: a superset of the documents with hits, not everything
    # keyword set is the user's, the hidden literals are not visible (finder.go:58-63 GetKeywords)
    assert sorted(f.GetKeywords()) == sorted(set(k for k in f.GetKeywords()))
    # single-document calls agree
    for d in (0, 7, 199):
        one = f.ProcessText(bytes(text[int(off[d]):int(off[d + 1])]))
        assert [r.ExpresionIndex for r in one] == [i for i in range(len(exprs)) if want[d, i >> 5] >> (i & 31) & 1]


def _device_batch(texts):
    import torch
    blob, off = pack_strings(texts)
    t = torch.from_numpy(np.concatenate([blob, np.zeros(64, np.uint8)])).cuda()      # 64 bytes of readable slack
    o = torch.from_numpy(off.astype(np.int64)).cuda()
    return t, o


def test_process_device_non_ascii_text_takes_the_host_tolower():
    """gft_finder_process_device folds A-Z on the device; text that leaves ASCII must still come out like the reference's
    strings.ToLower (finder.go:140-142): upper-case non-ASCII letters, and the Kelvin sign U+212A whose lower-case form
    'k' is shorter.  The kernels flag bytes >= 0x80, the finder repeats such a batch through the host path."""
    import torch
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    exprs = ['"école"', '"kelvin"', 'inord("la" and "école")', '"ecole"', '"k"']
    f.AddExpressions(exprs)
    texts = ["Vive la École", "273 Kelvin", "plain ascii ECOLE", "", "la école"]
    t, o = _device_batch(texts)
    words = 1
    bm = torch.zeros((len(texts), words), dtype=torch.int32, device="cuda")
    f.ProcessDevice(t.data_ptr(), o.data_ptr(), len(texts), bm.data_ptr())
    got = bm.cpu().numpy().astype(np.uint32)
    # oracle: the reference lower-cases the text first (Python's str.lower() agrees with Go on these strings)
    o2 = Oracle(sorted({"école", "kelvin", "la", "ecole", "k"}))
    o2.set_expressions(exprs, False)
    lb, lo = pack_strings([s.lower() for s in texts])
    want = o2.process(lb, lo, fold=False)
    assert np.array_equal(got, want)
    assert got[:, 0].tolist() == [0b00101, 0b10010, 0b01000, 0, 0b00101]
    assert _lib_load().gft_last_nonascii(f.engine_handle()) in (0, 1)
    # an ASCII batch stays on the device path
    t, o = _device_batch(["ECOLE kelvin", "K"])
    bm = torch.zeros((2, 1), dtype=torch.int32, device="cuda")
    f.ProcessDevice(t.data_ptr(), o.data_ptr(), 2, bm.data_ptr())
    assert bm.cpu().numpy().astype(np.uint32)[:, 0].tolist() == [0b11010, 0b10000]
    assert _lib_load().gft_last_nonascii(f.engine_handle()) == 0


def _lib_load():
    from gofindthem_amd import _lib
    return _lib.load()


def test_folded_batches_in_sequence():
    """ProcessDevice over batches that stay ASCII, leave it harmlessly (lower-case Latin-1: ASCII folding is the whole of
    strings.ToLower, the batch stays on the device) and leave it for good (upper-case non-ASCII letters: the finder
    repeats the batch through the host's ToLower), in every order: the engine keeps state from batch to batch (sizes,
    the one-launch unit table, the non-ASCII flag) and none of it may leak into the next result."""
    import torch
    L = _lib_load()
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    exprs = ['"école"', '"ecole" or "straße"', '"la" and not "k"']
    f.AddExpressions(exprs)
    o = Oracle(sorted({"école", "ecole", "straße", "la", "k"}))
    o.set_expressions(exprs, False)
    safe = ["vive la école", "LA STRAßE", "plain ECOLE", "", "à la carte"] * 40      # lower-case Latin-1 only: device path
    unsafe = ["Vive la École", "LA STRASSE École"] * 30                               # upper-case É: host path
    ascii_ = ["ECOLE la", "k LA"] * 10

    def run(texts, want_flag):
        t, off = _device_batch(texts)
        bm = torch.zeros((len(texts), 1), dtype=torch.int32, device="cuda")
        f.ProcessDevice(t.data_ptr(), off.data_ptr(), len(texts), bm.data_ptr())
        lb, lo = pack_strings([s.lower() for s in texts])
        assert np.array_equal(bm.cpu().numpy().astype(np.uint32), o.process(lb, lo, fold=False))
        if want_flag is not None:
            assert L.gft_last_nonascii(f.engine_handle()) == want_flag

    for _ in range(3):
        run(safe, 0)
    for _ in range(3):
        run(unsafe, None)     # (the finder has repeated the batch on the host path; results above are what counts)
    run(ascii_, 0)
    run(ascii_, 0)
    run(safe, 0)
    run(unsafe, None)
    run(safe, 0)


def test_process_device_one_read_back_per_batch_and_regrowth():
    """gft_process_device sizes the unit table and the match pool from the previous batch and reads the control block
    back once, after the solver; a batch that outgrows them (longer documents -> more units, a denser dictionary hit
    rate -> more matches) is run again with the right sizes.  Results must not depend on any of that."""
    import torch
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(500)
    terms = w.terms()
    exprs = make_expressions(terms, 64, inord_fraction=0.5, cover=True)
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(set(t.decode() for t in terms)))
    o.set_expressions(exprs, False)

    def run(texts):
        t, off = _device_batch(texts)
        bm = torch.zeros((len(texts), 2), dtype=torch.int32, device="cuda")
        f.ProcessDevice(t.data_ptr(), off.data_ptr(), len(texts), bm.data_ptr())
        lb, lo = pack_strings(texts)
        assert np.array_equal(bm.cpu().numpy().astype(np.uint32), o.process(lb, lo, fold=True))

    text, off = w.docs_host(0, 300)
    docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode() for d in range(300)]
    run(docs[:50])                       # first batch: sizes unknown, synchronous path
    run(docs[50:100])                    # same shape: deferred path
    run(docs[100:300])                   # more documents than the unit table holds: regrown, run again
    run([" ".join(docs[:40])] + docs[40:60])     # one 160 KB document -> many units
    dense = " ".join(t.decode() for t in terms[:200]) * 30
    run([dense, dense[:5000], docs[7]])  # far more matches per byte than the pool was sized for
    run(docs[:50])


def test_process_device_refuses_descending_offsets_on_the_deferred_path():
    """gft_process_device launches the scan BEFORE the host has read the bad-offsets flag once it runs deferred (from the
    second batch of a sequence on): offsets that descend -- a document "length" that wraps to >= 4 GiB -- must come back
    as GFT_E_INVALID with every unit still inside its document (k_unit_fill / k_units_single give such a document an empty
    unit), never as a read beyond the text.  Both unit-table paths: the general one (first deferred batches) and the
    one-launch table for batches of single-unit documents (from the third batch on)."""
    import torch
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(300)
    terms = w.terms()
    exprs = make_expressions(terms, 40, inord_fraction=0.3, cover=True)
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(set(t.decode() for t in terms)))
    o.set_expressions(exprs, False)
    text, off = w.docs_host(0, 1500)
    docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode() for d in range(1500)]

    def run(off_dev, t, n, want):
        bm = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
        if want is None:
            with pytest.raises(FinderError) as ei:
                f.ProcessDevice(t.data_ptr(), off_dev.data_ptr(), n, bm.data_ptr())
            assert "ascending" in str(ei.value)
        else:
            f.ProcessDevice(t.data_ptr(), off_dev.data_ptr(), n, bm.data_ptr())
            assert np.array_equal(bm.cpu().numpy().astype(np.uint32), want)

    t, good = _device_batch(docs)
    lb, lo = pack_strings(docs)
    want = o.process(lb, lo, fold=True)
    for step in range(5):
        run(good, t, 1500, want)                       # sizes learnt, then deferred, then the one-launch unit table
        bad = good.clone()
        bad[700] = bad[701] + 3                        # document 699 runs backwards: its length wraps
        run(bad, t, 1500, None)
    run(good, t, 1500, want)


def test_keyword_and_regex_with_the_same_literal_share_one_list():
    """A keyword "aa" and a regex r"aa" feed ONE map key in the reference (finder/finder.go:181-196): its list is the
    keyword positions followed by the regex positions, [2 3 4 5] ++ [2 4] for the text below -- not sorted, so
    getLowestIdxGTVal's binary search (dsl/expression.go:175-189) can miss an element that is there: inord("aaay" and
    "aa") asks for a position > 4 and the reference answers none, although 5 is in the list.  The device solver keeps one
    sorted position set per slot; documents in which a slot read by an INORD group gets such a list are therefore solved
    by the host solver (csrc/host_solve.cpp: the reference's algorithm on materialised lists) -- identical results."""
    exprs = ['inord("aaay" and "aa")', 'r"aa"', '"aa" and "aaay"', 'inord("aa" and "aaay")', 'inord("xx" and "aa")',
             'not (inord("aaay" and "aa"))']
    text = "xxaaaaay"
    f = Finder(GpuEngine(), PyRegexpEngine(), True)
    f.AddExpressions(exprs)
    got = [r.ExpresionIndex for r in f.ProcessText(text)]
    blob, off = pack_strings([text])
    want_bm, _ = _oracle_with_regex(f, exprs, blob, off, 1)
    want = [i for i in range(len(exprs)) if want_bm[0, 0] >> i & 1]
    assert want == [1, 2, 3, 4, 5]      # the reference: expression 0 is false (binary search over [2 3 4 5 2 4] misses 5)
    assert got == want
    # a batch: only the documents where the pair really occurs take the host's road, all rows equal the reference's
    texts = [text, "xxaay", "nothing here", "aaay aa", "xx aa aa aaay aa", "aaaaaaay xx aa", ""] * 9
    blob, off = pack_strings(texts)
    bm = f.ProcessTexts(texts)
    want_bm, _ = _oracle_with_regex(f, exprs, blob, off, len(texts))
    assert np.array_equal(bm, want_bm)
    # without the regex twin nothing is irregular: the device alone answers, and agrees
    f2 = Finder(GpuEngine(), PyRegexpEngine(), True)
    f2.AddExpressions(exprs[:1] + exprs[2:])
    o = Oracle(sorted(f2.GetKeywords()))
    o.set_expressions(exprs[:1] + exprs[2:], True)
    assert np.array_equal(f2.ProcessTexts(texts), o.process(blob, off))


def test_foreign_engines_with_the_same_literal_share_one_list():
    """the same with a foreign substring engine: every match arrives as a caller-supplied match, the keyword's hits and the
    regex's for one literal land in one slot in that order -- a list that is not ascending is spotted from the order of the
    caller's matches alone"""
    ms_sub = [Match(p, "aa") for p in (2, 3, 4, 5)] + [Match(4, "aaay")]
    ms_rgx = [Match(2, "aa"), Match(4, "aa")]
    sub, rgx = Fixed(ms_sub), Fixed(ms_rgx)
    f = Finder(sub, rgx, True)
    exprs = ['inord("aaay" and "aa")', 'inord("aaay" and r"aa")', '"aa"', 'inord("aa" and "aaay")']
    f.AddExpressions(exprs)
    rgx.regexes = f.GetRegexes()
    got = [r.ExpresionIndex for r in f.ProcessText("xxaaaaay")]
    assert got == [2, 3]                # [2 3 4 5 2 4]: nothing > 4 is found (dsl/expression.go:175-189)
    rgx.matches = []
    assert [r.ExpresionIndex for r in f.ProcessText("xxaaaaay")] == [0, 1, 2, 3]     # [2 3 4 5]: 5 > 4


def test_expressions_beyond_the_device_solver_limits_are_solved_on_the_host():
    """The device solver keeps an INORD group's (slot, threshold) pairs one per lane -- 64 --, a wider group's in a scratch
    region per wave (up to 8 192 pairs: `wide`, round 4), and nests 128 operands deep in its FUSED form (`deep_right`, 200
    operands waiting on the public postfix stack, is flat there); the reference's recursion has no limits at all
    (dsl/expression.go:66-142).  What is still beyond the device (`huge`: an INORD over 2 x 4 500 OR-ed leaves) does not make
    gft_set_programs fail: it is solved on the host from the scan's matches while the other expressions of the set run on
    the device as before."""
    import torch
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(2000)
    terms = [t.decode() for t in w.terms()]
    exprs = make_expressions(w.terms(), 1000, inord_fraction=0.3, cover=True)
    wide = "inord((%s) and (%s))" % (" or ".join('"%s"' % t for t in terms[:200]), " or ".join('"%s"' % t for t in terms[200:400]))
    deep = "(" * 150 + " or ".join('("%s" and "%s")' % (terms[2 * i], terms[2 * i + 1]) + ")" for i in range(150))
    deep_right = '"%s"' % terms[500]
    for i in range(200):                                 # nests to the right: 200 operands wait on the postfix stack
        deep_right = '("%s" %s %s)' % (terms[501 + i], "and" if i % 3 else "or", deep_right)
    huge = "inord((%s) and (%s))" % (" or ".join('"%s"' % terms[i % 700] for i in range(4500)),
                                     " or ".join('"%s"' % terms[700 + i % 700] for i in range(4500)))
    exprs = exprs[:400] + [wide] + exprs[400:900] + [deep_right, deep] + exprs[900:] + [huge]
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(f.GetKeywords()))
    o.set_expressions(exprs, False)
    text, off = w.docs_host(0, 300)
    want = o.process(text, off, fold=True)
    assert want[:, 400 >> 5].any() and np.array_equal(f.ProcessTexts(blob=text, doc_off=off), want)
    assert _lib_load().gft_n_host_exprs(f.engine_handle()) == 1          # `huge` alone (the programs are uploaded by now)
    one = f.ProcessText(bytes(text[int(off[5]):int(off[6])]))
    assert [r.ExpresionIndex for r in one] == [i for i in range(len(exprs)) if want[5, i >> 5] >> (i & 31) & 1]
    # device-resident corpus: the host's bits are patched into the device bitmap
    docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode() for d in range(300)]
    t, od = _device_batch(docs)
    words = (len(exprs) + 31) // 32
    for _ in range(3):                                   # (first call sizes the tables, the next ones would run deferred)
        bm = torch.zeros((300, words), dtype=torch.int32, device="cuda")
        f.ProcessDevice(t.data_ptr(), od.data_ptr(), 300, bm.data_ptr())
        assert np.array_equal(bm.cpu().numpy().astype(np.uint32), want)


@pytest.mark.parametrize("k", [100, 10000])
def test_inord_chains_of_the_reference_benchmarks(k):
    """benchmarks/benchmark_test.go:55-56, 132-134, 182-184, 438-462: exp100 / exp10000 are ONE `INORD` of 100 / 10 000 AND-ed
    terms over the ~1 MB document of createText (:66, :473-489).  Once with terms drawn at random like the reference does
    (a term that does not occur ends the chain: false), once with the terms planted in that order (true) -- every AND is a
    successor query on the document's matches (dsl/expression.go:87-95), the left-deep chain is as long as the reference's.
    Both documents in one batch, against the oracle."""
    from gofindthem_amd.workload import Workload
    w = Workload(10000)
    terms = [t.decode() for t in w.terms()]
    rng = np.random.default_rng(k)
    blob, off = w.docs_host(0, 250)
    random_doc = bytes(blob[:int(off[-1])]).decode()
    words = random_doc.split(" ")
    assert 900_000 < len(random_doc) < 1_300_000
    order = rng.permutation(len(terms))[:k].tolist()
    # the planted document: the same words with term order[i] inserted behind every (len(words) // k)-th word
    step = max(len(words) // k, 1)
    out, nxt = [], 0
    for i, wd in enumerate(words):
        out.append(wd)
        if i % step == step - 1 and nxt < k:
            out.append(terms[order[nxt]])
            nxt += 1
    out += [terms[j] for j in order[nxt:]]
    planted_doc = " ".join(out)
    chain = "INORD(" + " and ".join('"%s"' % terms[j] for j in order) + ")"
    drawn = "INORD(" + " and ".join('"%s"' % terms[j] for j in rng.integers(0, len(terms), k).tolist()) + ")"
    back = "INORD(" + " and ".join('"%s"' % terms[j] for j in order[::-1]) + ")"
    exprs = [chain, drawn, back, '"%s" and "%s"' % (terms[order[0]], terms[order[-1]])]
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(f.GetKeywords()))
    o.set_expressions(exprs, False)
    docs = [random_doc, planted_doc, "", planted_doc[: len(planted_doc) // 2]]
    lb, lo = pack_strings(docs)
    want = o.process(lb, lo, fold=True)
    got = f.ProcessTexts(docs)
    assert np.array_equal(got, want)
    assert int(want[1, 0]) & 1 == 1 and int(want[3, 0]) & 1 == 0          # in order on the planted document, cut short on its half
    one = f.ProcessText(planted_doc)
    assert [r.ExpresionIndex for r in one] == [i for i in range(len(exprs)) if int(want[1, 0]) >> i & 1]


def test_large_host_batch_folds_on_the_device_and_comes_back_when_it_cannot():
    """Finder.ProcessTexts on a batch of 16 MB or more does not scan the text for high bytes on the host first: it is
    uploaded as it is and folded by the scan kernels; only when they report text that ASCII folding does not lower-case the
    way strings.ToLower does (finder/finder.go:140-142) the batch is repeated through the host's ToLower.  Both ways must
    give the reference's rows."""
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(400)
    terms = w.terms()
    exprs = make_expressions(terms, 40, inord_fraction=0.3, cover=True) + ['"\u00e9cole"', '"ecole"']
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(k.encode("utf-8") for k in f.GetKeywords()))
    o.set_expressions(exprs, False)
    text, off = w.docs_host(0, 4300)
    assert int(off[-1]) >= 16 << 20
    docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode().upper() if d % 7 == 0 else bytes(text[int(off[d]):int(off[d + 1])]).decode()
            for d in range(4300)]
    lb, lo = pack_strings(docs)
    assert np.array_equal(f.ProcessTexts(docs), o.process(lb, lo, fold=True))          # ASCII: one pass, folded on the device
    docs[2999] = "\u00c9COLE normale " + docs[2999]                                        # an upper-case E-acute: ToLower's business
    docs[17] = "une \u00e9cole " + docs[17]                                                # lower-case: safe either way
    low = [d.lower() for d in docs]
    lb, lo = pack_strings(low)
    want = o.process(lb, lo, fold=False)
    got = f.ProcessTexts(docs)
    assert np.array_equal(got, want)
    assert want[2999, 41 >> 5] >> (40 & 31) & 1 and want[17, 40 >> 5] >> (40 & 31) & 1   # "\u00e9cole" is found in both


def test_process_device_begin_end_pipelines_two_batches():
    """gft_finder_process_device_begin / _end: two batches in flight, each with its own bitmap; _end
    completes the oldest and hands back ITS verdict.  Covered: steady state (deferred batches), an engine's first batches
    (completed inside _begin), a batch that outgrows the match pool the earlier ones left behind (run again inside its _end
    while a younger batch is in flight), the errors of calling out of order, and the synchronous entry point refusing to run
    beside batches in flight."""
    import torch
    from gofindthem_amd.workload import Workload, make_expressions
    L = _lib_load()
    w = Workload(1000)
    exprs = make_expressions(w.terms(), 200, inord_fraction=0.3, cover=True)
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(f.GetKeywords()))
    o.set_expressions(exprs, False)
    words = (len(exprs) + 31) // 32

    def batch(first, n):
        text, off = w.docs_host(first, n)
        want = o.process(text, off, fold=True)
        docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode() for d in range(n)]
        t, od = _device_batch(docs)
        return t, od, n, torch.zeros((n, words), dtype=torch.int32, device="cuda"), want

    small = [batch(100 * i, 100) for i in range(6)]
    big = batch(1000, 3000)                       # thirty times the matches of the batches that sized the pool
    with pytest.raises(FinderError):
        f.ProcessDeviceEnd()                      # nothing in flight
    seq = small[:4] + [big] + small[4:]
    inflight = []
    for b in seq:
        f.ProcessDeviceBegin(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3].data_ptr())
        inflight.append(b)
        if len(inflight) == 2:
            with pytest.raises(FinderError):      # a third batch does not fit
                f.ProcessDeviceBegin(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3].data_ptr())
            with pytest.raises(FinderError):      # ... and the synchronous form does not run beside them
                f.ProcessDevice(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3].data_ptr())
            done = inflight.pop(0)
            f.ProcessDeviceEnd()
            assert np.array_equal(done[3].cpu().numpy().astype(np.uint32), done[4])
    while inflight:
        done = inflight.pop(0)
        f.ProcessDeviceEnd()
        assert np.array_equal(done[3].cpu().numpy().astype(np.uint32), done[4])
    # the synchronous form works again once nothing is in flight
    b = small[0]
    b[3].zero_()
    f.ProcessDevice(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3].data_ptr())
    assert np.array_equal(b[3].cpu().numpy().astype(np.uint32), b[4])
    assert L.gft_last_nonascii(f.engine_handle()) == 0
    f.close()


def test_steady_state_batches_are_scanned_once():
    """A sequence of equal batches settles on ONE scan launch and ONE solver launch per batch: the
    deferred verdict (gft_api.cpp deferred_interpret) must not send a batch of single-unit documents around again, on the
    synchronous entry point and with two batches in flight alike -- a rerun is invisible in the results, so the profile
    counters (gft_profile_read) are what this test reads."""
    import ctypes as C
    import torch
    from gofindthem_amd.workload import Workload, make_expressions
    L = _lib_load()
    w = Workload(1000)
    exprs = make_expressions(w.terms(), 100, inord_fraction=0.2, cover=True)
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(f.GetKeywords()))
    o.set_expressions(exprs, False)
    words = (len(exprs) + 31) // 32
    text, off = w.docs_host(0, 400)
    want = o.process(text, off, fold=True)
    docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode() for d in range(400)]
    t, od = _device_batch(docs)
    bms = [torch.zeros((400, words), dtype=torch.int32, device="cuda") for _ in range(2)]
    for _ in range(6):                                 # sizes learnt, deferred, then the one-launch unit table
        f.ProcessDevice(t.data_ptr(), od.data_ptr(), 400, bms[0].data_ptr())
    eh = f.engine_handle()

    def launches(name):
        ms, n = C.c_double(), C.c_uint64()
        assert L.gft_profile_read(eh, name.encode(), C.byref(ms), C.byref(n)) == 0
        return n.value

    L.gft_profile_enable(eh, 1)
    L.gft_profile_reset(eh)
    for _ in range(12):
        f.ProcessDevice(t.data_ptr(), od.data_ptr(), 400, bms[0].data_ptr())
    assert launches("scan") == 12 and launches("solve") == 12
    L.gft_profile_reset(eh)
    for i in range(12):
        f.ProcessDeviceBegin(t.data_ptr(), od.data_ptr(), 400, bms[i % 2].data_ptr())
        if i:
            f.ProcessDeviceEnd()
    f.ProcessDeviceEnd()
    assert launches("scan") == 12 and launches("solve") == 12
    L.gft_profile_enable(eh, 0)
    for bm in bms:
        assert np.array_equal(bm.cpu().numpy().astype(np.uint32), want)


def test_cu_margin_leaves_results_alone():
    """gft_set_cu_margin (what bench.py sets for N > 1 so that RCCL's kernels find a CU beside the persistent scan and solver
    workgroups): the same bitmaps on 240 CUs, on ONE CU (a margin larger than the device) and on all of them again; refused
    while a batch is in flight."""
    import torch
    from gofindthem_amd.workload import Workload, make_expressions
    L = _lib_load()
    w = Workload(1000)
    exprs = make_expressions(w.terms(), 100, inord_fraction=0.3, cover=True)
    f = Finder(GpuEngine(), EmptyRgxEngine(), False)
    f.AddExpressions(exprs)
    o = Oracle(sorted(f.GetKeywords()))
    o.set_expressions(exprs, False)
    words = (len(exprs) + 31) // 32
    text, off = w.docs_host(0, 600)
    want = o.process(text, off, fold=True)
    docs = [bytes(text[int(off[d]):int(off[d + 1])]).decode() for d in range(600)]
    t, od = _device_batch(docs)
    bm = torch.zeros((600, words), dtype=torch.int32, device="cuda")
    f.ProcessDevice(t.data_ptr(), od.data_ptr(), 600, bm.data_ptr())       # (the engine exists from here on)
    eh = f.engine_handle()
    for margin in (16, 100000, 0):
        assert L.gft_set_cu_margin(eh, margin) == 0
        for _ in range(4):                                                 # first, deferred and one-launch-unit-table batches
            bm.zero_()
            f.ProcessDevice(t.data_ptr(), od.data_ptr(), 600, bm.data_ptr())
            assert np.array_equal(bm.cpu().numpy().astype(np.uint32), want)
    f.ProcessDeviceBegin(t.data_ptr(), od.data_ptr(), 600, bm.data_ptr())
    assert L.gft_set_cu_margin(eh, 8) != 0
    f.ProcessDeviceEnd()
    assert L.gft_set_cu_margin(eh, 8) == 0
