"""The oracle (oracle/) against the reference's own test tables (tests/golden/*.json).

This is what pins the CPU restatement: dsl/expression_test.go, dsl/parser_test.go, dsl/scanner_test.go,
finder/finder_test.go and the real-engine cases of group/finder/finder_test.go.  No GPU involved.
"""
import re

import numpy as np
import pytest

from conftest import load_golden
from oracle import dsl_ref
from oracle.pyoracle import Oracle, pack_strings, POS_START, POS_END


def docs(texts):
    blob, off = pack_strings(texts)
    return blob, off


# ---- dsl/scanner_test.go ------------------------------------------------------------------------------
@pytest.mark.parametrize("case", load_golden("scanner.json")["cases"], ids=lambda c: c["message"])
def test_scanner(case):
    sc = dsl_ref.Scanner(case["expStr"])
    for exp in case["expected"]:
        try:
            tok, lit = sc.scan()
            err = None
        except dsl_ref.DslError as e:
            tok, lit, err = dsl_ref.ILLEGAL, "", str(e)
        assert err == exp["Err"]
        assert dsl_ref.TOKEN_NAMES[tok] == exp["Tok"]
        assert lit == exp["Lit"]
        if err is not None or tok == dsl_ref.EOF:
            break


# ---- dsl/parser_test.go -------------------------------------------------------------------------------
@pytest.mark.parametrize("case", load_golden("parser.json")["cases"], ids=lambda c: c["message"] + "|" + c["expStr"])
def test_parser(case):
    try:
        exp, kws, rgx = dsl_ref.parse(case["expStr"], case["caseSense"])
        err = None
    except dsl_ref.DslError as e:
        err = str(e)
    assert err == case["error"]
    if err is None:
        assert exp.to_obj() == case["exp"]
        assert sorted(kws) == case["keywords"]
        assert sorted(rgx) == case["regexes"]


# ---- dsl/expression_test.go ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", load_golden("solver.json")["cases"], ids=lambda c: c["message"])
def test_solver(case):
    o = Oracle([])
    o.set_expressions([case["expStr"]], case_sensitive=True)
    assert o.solve(0, case["map"]) is case["expected"]


def test_readme_inord_example():
    ex = load_golden("examples.json")["readme_inord"]
    o = Oracle([])
    o.set_expressions([ex["expStr"]], True)
    assert o.solve(0, ex["map"]) is ex["expected"]


# ---- finder/finder_test.go: solveExpressions ----------------------------------------------------------
def test_solve_expressions():
    g = load_golden("solve_expressions.json")
    o = Oracle([])
    o.set_expressions([e["exprString"] for e in g["expressions"]], True)
    # the fixture's hand-built trees equal what the parser builds from the strings
    for e in g["expressions"]:
        assert dsl_ref.parse(e["exprString"], True)[0].to_obj() == e["expression"]
    for c in g["cases"]:
        got = [{"ExpresionIndex": i, "ExpresionStr": g["expressions"][i]["exprString"], "Tag": ""}
               for i in range(len(g["expressions"])) if o.solve(i, c["map"])]
        assert got == c["expected"], c["message"]


# ---- group/finder/finder_test.go: the only real-engine cases ------------------------------------------
@pytest.mark.parametrize("pos_mode", [POS_START, POS_END])
def test_engine_truth(pos_mode):
    g = load_golden("engine_truth.json")
    for c in g["cases"]:
        o = Oracle(["string"], pos_mode)
        o.set_expressions([c["expression"]], c["caseSensitive"])
        blob, off = docs([c["text"]])
        bm = o.process(blob, off, fold=not c["caseSensitive"])
        assert bool(bm[0, 0] & 1) is c["expected_true"], c["text"]


# ---- examples/finder/main.go (expected indices hand-derived, see fixture header) -----------------------
def _regex_extra(o, texts, fold):
    """host-side stand-in for RegexpEngine.FindRegexes (finder/regexEngine.go:36-47): start offsets of
    non-overlapping leftmost matches, keyed by the regex source text."""
    offs, lits, poss = [0], [], []
    for t in texts:
        t = t.lower() if fold else t
        for li, lit in enumerate(o.literals):
            if lit not in o._regexes:
                continue
            for m in re.finditer(lit, t):
                lits.append(li)
                poss.append(len(t[:m.start()].encode()))
        offs.append(len(lits))
    return (np.asarray(offs, np.uint64), np.asarray(lits or [0], np.int32), np.asarray(poss or [0], np.int64))


@pytest.mark.parametrize("which", ["case_sensitive", "case_insensitive"])
def test_examples_finder(which):
    g = load_golden("examples.json")
    cs = which == "case_sensitive"
    sec = g[which]
    exprs = [e for e, _tag in sec["expressions"]]
    kws = {}
    for e in exprs:
        kws.update(dict.fromkeys(dsl_ref.parse(e, cs)[1]))
    o = Oracle(list(kws))
    _, rgx = o.set_expressions(exprs, cs)
    o._regexes = set(rgx)
    blob, off = docs(g["texts"])
    bm = o.process(blob, off, fold=not cs, extra=_regex_extra(o, g["texts"], not cs))
    for d, want in enumerate(sec["expected_true"]):
        got = [i for i in range(len(exprs)) if bm[d, i >> 5] >> (i & 31) & 1]
        assert got == want, (which, d)


# ---- AC restatement vs the independent brute-force enumerator ----------------------------------------
def test_ushers_known_answer():
    # textbook case fixing the canonical order (end asc, length desc); build-defined, see SURVEY 8(c)
    o = Oracle(["he", "she", "his", "hers"], POS_START)
    assert o.terms() == [b"he", b"hers", b"his", b"she"]
    blob, off = docs(["ushers"])
    moff, tid, pos = o.scan(blob, off)
    assert list(zip(tid.tolist(), pos.tolist())) == [(3, 1), (0, 2), (1, 2)]
    o = Oracle(["he", "she", "his", "hers"], POS_END)
    moff, tid, pos = o.scan(blob, off)
    assert list(zip(tid.tolist(), pos.tolist())) == [(3, 3), (0, 3), (1, 5)]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("pos_mode", [POS_START, POS_END])
def test_scan_equals_bruteforce(seed, pos_mode):
    rng = np.random.default_rng(seed)
    alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][seed % 4]
    n_terms = [1, 5, 40, 300, 1000, 17][seed]
    terms = set()
    for _ in range(n_terms):
        L = int(rng.integers(1, 9 if seed != 4 else 5))
        terms.add(bytes(alpha[i] for i in rng.integers(0, len(alpha), L)))
    terms.add(b"")                     # legal keyword, must never match (dsl/scanner.go:211-212)
    o = Oracle(sorted(terms), pos_mode)
    texts = [bytes(alpha[i] for i in rng.integers(0, len(alpha), int(n))) for n in [0, 1, 2, 7, 64, 300, 2000]]
    blob, off = docs(texts)
    a = o.scan(blob, off)
    b = o.brute(blob, off)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[1].size > 0 or n_terms == 1


def test_duplicate_and_empty_dictionary():
    o = Oracle(["aa", "aa", "a"])
    assert o.n_terms == 2
    blob, off = docs(["aaa"])
    _, tid, pos = o.scan(blob, off)
    assert list(zip(tid.tolist(), pos.tolist())) == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2)]
    o = Oracle([])
    _, tid, _ = o.scan(blob, off)
    assert tid.size == 0
