"""Host-side logic of the product (no GPU): DSL front-end in libgft.so against the reference's tables, the
Finder registry / engine orchestration with mocked engines, C-ABI symbol coverage."""
import ctypes as C
import json
import os
import re

import pytest

from conftest import ROOT, load_golden
from gofindthem_amd import _lib
from gofindthem_amd.finder import (EmptyEngine, EmptyRgxEngine, Finder, FinderError, Match, RegexEngine,
                                   SubstringEngine)


def _call(fn, *a):
    need = C.c_uint64()
    buf = C.create_string_buffer(1 << 12)
    rc = fn(*a, buf, len(buf), C.byref(need))
    if rc != 0:
        buf = C.create_string_buffer(need.value + 16)
        rc = fn(*a, buf, len(buf), C.byref(need))
    assert rc == 0
    return buf.raw[:need.value]


def dsl_parse(expr, cs=True):
    b = expr.encode("utf-8")
    return json.loads(_call(_lib.load().gft_dsl_parse, b, len(b), 1 if cs else 0)[:-1])


# ---- the C ABI exports every symbol include/gft.h declares ---------------------------------------------------
def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "gft.h")).read()
    declared = set(re.findall(r"\b(gft_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"gft_emit_fn", "gft_engine_build_fn", "gft_engine_find_fn"}
    L = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_no_device_fails_loudly():
    """there is no CPU fallback: without a HIP device the compute entry points return GFT_E_HIP"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gofindthem_amd.engine import Engine, GftError
    with pytest.raises(GftError) as ei:
        Engine()
    assert ei.value.code == _lib.GFT_E_HIP


def test_loader_brings_torch_in_first_so_that_one_hip_runtime_is_mapped():
    """libgft.so links /opt/rocm's libamdhip64, PyTorch ships its own copy of the same soname; a process that
    mapped libgft.so BEFORE torch has carried two HIP runtimes.  The product loader imports torch
    first when it is importable -- checked in a fresh process that never names torch itself."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gofindthem_amd import _lib\n"
            "assert 'torch' not in sys.modules\n"
            "_lib.load()\n"
            "libs = _lib._mapped_hip_runtimes()\n"
            "print('torch' in sys.modules, len(libs))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["True", "1"], (out.stdout, out.stderr[-500:])


def test_missing_library_fails_loudly(monkeypatch):
    """... and without libgft.so the package does not fall back to anything either"""
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgft.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


# ---- dsl/scanner_test.go ------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", load_golden("scanner.json")["cases"], ids=lambda c: c["message"])
def test_scanner(case):
    b = case["expStr"].encode("utf-8")
    toks = json.loads(_call(_lib.load().gft_dsl_tokens, b, len(b))[:-1])
    exp = case["expected"]
    n = len(toks)
    assert toks == exp[:n]
    assert toks[-1]["Err"] is not None or toks[-1]["Tok"] == "EOF"


# ---- dsl/parser_test.go -------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", load_golden("parser.json")["cases"], ids=lambda c: c["message"] + "|" + c["expStr"])
def test_parser(case):
    r = dsl_parse(case["expStr"], case["caseSense"])
    assert r.get("error") == case["error"]
    if case["error"] is None:
        assert r["tree"] == case["exp"]
        assert sorted(r["keywords"]) == case["keywords"]
        assert sorted(r["regexes"]) == case["regexes"]


def test_parser_matches_oracle_on_generated_expressions():
    """two independent restatements of dsl/parser.go (C++ in the product, Python in the oracle) agree"""
    from gofindthem_amd.workload import Workload, make_expressions
    from helpers import tree_to_program
    from oracle import dsl_ref
    terms = Workload(200).terms()
    exprs = make_expressions(terms, 300, inord_fraction=0.3, regexes=["ab.*cd", "x+y"])
    exprs += ['"a" "b" "c"', '("a" and "b") "c" or "d"', '"a" and ("b") ("c")', 'not not "a"' if False else '"a"',
              '"A\\tb" AND R"Q\\\\d"', '(("a"))', '"a" or (not "b") and inord(("c") and "d" or "e")']
    seen_unset = []
    for e in exprs:
        for cs in (True, False):
            r = dsl_parse(e, cs)
            t, kw, rx = dsl_ref.parse(e, cs)
            assert r["tree"] == t.to_obj(), e
            assert r["keywords"] == kw and r["regexes"] == rx
            lits = kw + [x for x in rx if x not in kw]
            if r["solve_error"] is None:
                assert r["program"] == tree_to_program(t, lits.index)
            else:       # the parser accepted a tree that Expression.Solve rejects (UNSET node with two children)
                assert r["solve_error"] == "unable to process expression type 0" and '"UNSET"' in json.dumps(r["tree"])
                seen_unset.append(e)
    assert seen_unset


def test_deep_expression_compiles():
    e = "INORD(" + " AND ".join('"w%d"' % i for i in range(10000)) + ")"   # benchmarks/benchmark_test.go:56
    b = e.encode()
    raw = _call(_lib.load().gft_dsl_parse, b, len(b), 1)[:-1]
    # the tree is 10 000 levels deep (too deep for Python's json): read only the flat tail of the document
    r = json.loads(b"{" + raw[raw.index(b'"keywords":'):])
    assert len(r["keywords"]) == 10000 and len(r["program"]) == 2 * 10000 and r["solve_error"] is None


def test_to_lower_matches_unicode_simple_mapping():
    L = _lib.load()
    for s in ["Lorem IPSUM", "ÀÉÎÕÜ Straße", "ΣΑΣ", "ДОБРО Ёж", "İstanbul", "ǅ ǈ", "plain ascii 123"]:
        b = s.encode("utf-8")
        got = _call(L.gft_to_lower, b, len(b)).decode("utf-8")
        want = "".join("i" if ch == "İ" else (ch.lower() if len(ch.lower()) == 1 else ch) for ch in s)
        assert got == want, s
    bad = b"AB\xffCD\xc3"      # invalid UTF-8 decodes to U+FFFD per byte, as Go's strings.ToLower does
    assert _call(L.gft_to_lower, bad, len(bad)) == b"ab\xef\xbf\xbdcd\xef\xbf\xbd"


# ---- finder/finder_test.go: TestAddExpression ------------------------------------------------------------------
@pytest.mark.parametrize("case", load_golden("add_expression.json")["cases"], ids=lambda c: c["message"])
def test_add_expression(case):
    f = Finder(EmptyEngine(), EmptyRgxEngine(), case["caseSensitive"], allow_no_device=True)
    for expr, want in zip(case["expressions"], case["errors"]):
        try:
            f.AddExpression(expr)
            err = None
        except FinderError as e:
            err = str(e)
        assert err == want
    got = [dict(zip(("exprString", "tag", "expression"), (lambda s, t, j: (s, t, j))(*f.expression(i))))
           for i in range(f.n_expressions)]
    assert got == [{"exprString": w["exprString"], "tag": w["tag"], "expression": w["expression"]} for w in case["exprs"]]
    assert sorted(f.GetKeywords()) == case["keywords"]
    assert sorted(f.GetRegexes()) == case["regexes"]


# ---- finder/finder_test.go: TestProcessText, the error-propagation half (needs no solve) -------------------------
class SubMock(SubstringEngine):
    def __init__(self, build_err, find):
        self.build_err, self.find, self.calls = build_err, find, []

    def BuildEngine(self, keywords, caseSensitive):
        self.calls.append(("BuildEngine", sorted(keywords)))
        if self.build_err:
            raise Exception(self.build_err)

    def FindSubstrings(self, text):
        self.calls.append(("FindSubstrings", text))
        if self.find["err"]:
            raise Exception(self.find["err"])
        return [Match(m["Position"], m["Term"]) for m in self.find["matches"]]


class RgxMock(RegexEngine):
    def __init__(self, build_err, find):
        self.build_err, self.find, self.calls = build_err, find, []

    def BuildEngine(self, regexes, caseSensitive):
        self.calls.append(("BuildEngine", sorted(regexes)))
        if self.build_err:
            raise Exception(self.build_err)

    def FindRegexes(self, text):
        self.calls.append(("FindRegexes", text))
        if self.find["err"]:
            raise Exception(self.find["err"])
        return [Match(m["Position"], m["Term"]) for m in self.find["matches"]]


def make_mocked_finder(c, allow_no_device):
    sub, rgx = SubMock(c["buildSubErr"], c["findSub"]), RgxMock(c["buildRgxErr"], c["findRgx"])
    f = Finder(sub, rgx, True, allow_no_device=allow_no_device)
    for w in c["expressions"]:
        f.AddExpressionWithTag(w["exprString"], w["tag"])
    for k in c["keywords"]:
        f.debug_add_literal(0, k)
    for r in c["regexes"]:
        f.debug_add_literal(1, r)
    f.debug_set_updated(c["updatedSub"], c["updatedRgx"])
    return f, sub, rgx


@pytest.mark.parametrize("case", [c for c in load_golden("process_text.json")["cases"] if c["expectedErr"]],
                         ids=lambda c: c["message"])
def test_process_text_error_propagation(case):
    f, sub, rgx = make_mocked_finder(case, allow_no_device=True)
    with pytest.raises(FinderError) as ei:
        f.ProcessText(load_golden("process_text.json")["text"])
    assert str(ei.value) == case["expectedErr"]          # the engine's error value, unchanged (finder.go:149-158)
    assert ei.value.code == _lib.GFT_E_ENGINE


def test_force_build_quirk():
    """finder/finder.go:218-235: the regex branch sets updatedSubMachine; BuildEngine runs even with no regexes"""
    sub, rgx = SubMock(None, {"matches": [], "err": None}), RgxMock(None, {"matches": [], "err": None})
    f = Finder(sub, rgx, True, allow_no_device=True)
    f.AddExpression('"a" and r"b"')
    assert f.debug_get_updated() == (False, False)
    f.ForceBuild()
    assert sub.calls == [("BuildEngine", ["a"])] and rgx.calls == [("BuildEngine", ["b"])]
    assert f.debug_get_updated() == (True, False)        # sic
    f.ForceBuild()
    assert len(rgx.calls) == 2 and len(sub.calls) == 1
    f.AddExpression('"a"')                                # a known keyword still clears the flag (finder.go:123-126)
    assert f.debug_get_updated() == (False, False)


# ---- regex prefilter: required literals ------------------------------------------------------------
def _required(pattern):
    import ctypes as C
    import json
    L = _lib.load()
    p = pattern.encode("utf-8")
    buf = C.create_string_buffer(4096)
    need = C.c_uint64()
    assert L.gft_regex_required_literals(p, len(p), C.cast(buf, C.c_void_p), 4096, C.byref(need)) == 0
    return json.loads(buf.value.decode("utf-8"))


@pytest.mark.parametrize("pattern,want", [
    ("en.*nr", ["en", "nr"]), ("po[a-z]+ud", ["po", "ud"]), ("q+", []), ("abc", ["abc"]), ("ab?c", []), ("abc?d", ["ab"]),
    ("ab*cd", ["cd"]), ("abc+de", ["abc", "de"]), ("a{2}bc", ["bc"]), ("xy{0,3}zw", ["zw"]), ("xy{1,3}zw", ["xy", "zw"]),
    (r"foo\.bar", ["foo.bar"]), (r"foo\dbar", ["foo", "bar"]), (r"a\x41b", []), ("(ab|cd)ef", ["ef"]), ("ab|cd", []),
    ("(?i)abc", []), ("(?:ab)+cd", ["cd"]), ("^hello$", ["hello"]), ("he[l]lo wor.d", ["he", "lo wor"]), ("é+x", ["é"]),
    ("ééé?x", ["éé"]), ("ab)", []), ("a[bc", []), ("ab{x", []), (r"\Qab\E", []), ("héllo wörld", ["héllo wörld"]),
    (r"(?P<n>ab)cd\s+ef", ["cd", "ef"]), ("ab.*?cd", ["ab", "cd"]), ("[[:alpha:]]+foo", ["foo"]), (r"[\]]ab", ["ab"])])
def test_regex_required_literals(pattern, want):
    assert _required(pattern) == want


def test_regex_required_literals_are_required():
    """property: whenever the pattern matches a text, every extracted literal occurs in it (checked with Python's re
    on random texts over a small alphabet, for patterns both engines read the same way)"""
    import random
    import re
    rng = random.Random(7)
    pats = ["ab.*ba", "a+bb", "ab[ab]+ba", "(ab)+ba", "ab?ba", "aab*", "ba{2}b", "b.a.b", "abb|baa", "ab{1,2}a", "^ab.*b$",
            r"a\.?bb", "(?:ba)*ab", "bab+a", "a[^a]b", "abba?b"]
    for pat in pats:
        lits = _required(pat)
        rx = re.compile(pat)
        hits = 0
        for _ in range(3000):
            t = "".join(rng.choice("ab.") for _ in range(rng.randint(0, 12)))
            if rx.search(t):
                hits += 1
                for lit in lits:
                    assert lit in t, (pat, lits, t)
        assert hits > 0, pat
