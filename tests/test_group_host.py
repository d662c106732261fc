"""SURVEY.md 8(f) row 2, host half: the tag-rule DSL and the GroupFinder bookkeeping, written after the reference's own
tests (group/dsl/*_test.go, group/finder/finder_test.go, group/finder/internal_test.go; tests/golden/group_*.json).
Everything here runs without a GPU: the oracle restatement against the fixtures, and libgft.so's host-only entry points
(gft_group_dsl_*, gft_group_add_rule / _state / _evaluate, the JSON reader) against both."""
import json

import pytest

from conftest import load_golden
from gofindthem_amd import _lib, group
from gofindthem_amd.finder import Finder
from oracle import group_ref

PARSER = load_golden("group_parser.json")["cases"]
SCANNER = load_golden("group_scanner.json")["cases"]
SOLVER = load_golden("group_solver.json")["cases"]
FINDER = load_golden("group_finder.json")


# ---- group/dsl/parser_test.go ------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", PARSER, ids=lambda c: c["message"] + " " + c["expStr"][:20])
def test_parser_oracle_and_product(case):
    got = group.dsl_parse(case["expStr"])
    if case["error"] is not None:
        with pytest.raises(group_ref.GroupDslError) as ei:
            group_ref.parse(case["expStr"])
        assert str(ei.value) == case["error"]
        assert got == {"error": case["error"]}
        return
    e, tags, fields = group_ref.parse(case["expStr"])
    assert e.to_obj() == case["exp"]
    assert sorted(tags) == case["tags"] and sorted(fields) == case["fields"]
    assert got["tree"] == case["exp"]
    assert sorted(got["tags"]) == case["tags"] and sorted(got["fields"]) == case["fields"]


# ---- group/dsl/scanner_test.go -----------------------------------------------------------------------------------
@pytest.mark.parametrize("case", SCANNER, ids=lambda c: c["message"])
def test_scanner_oracle_and_product(case):
    want = case["expected"]
    # the reference's loop stops at the first error (scanner_test.go:117-122): later table rows are never reached
    for i, t in enumerate(want):
        if t["Err"] is not None:
            want = want[:i + 1]
            break
    assert group_ref.tokens(case["expStr"]) == want
    assert group.dsl_tokens(case["expStr"]) == want


def test_scanner_details_beyond_the_tables():
    """escapes, the space trim and the ':' hand-over between scanTag and scanFieldPath (scanner.go:177-244)"""
    for src in [r'"a\:b:c\"d"', '"  tag  :  f 1 "', r'"bad\q"', r'"t:bad\:"', '"t" :f"', '"é:ü"', '":f"', '""']:
        assert group.dsl_tokens(src) == group_ref.tokens(src), src
        try:
            e, tags, fields = group_ref.parse(src)
            want = {"tree": e.to_obj(), "tags": tags, "fields": fields}
        except group_ref.GroupDslError as x:
            want = {"error": str(x)}
        assert group.dsl_parse(src) == want, src


# ---- group/dsl/expression_test.go --------------------------------------------------------------------------------
def _finder_no_device():
    return Finder(None, None, False, allow_no_device=True)


@pytest.mark.parametrize("case", SOLVER, ids=lambda c: c["message"])
def test_solver_oracle_and_product(case):
    e, _, _ = group_ref.parse(case["expStr"])
    assert group_ref.solve(e, case["map"]) is case["expected"]
    g = group.GroupFinder(_finder_no_device())
    g.AddRule("r", [case["expStr"]])
    got = g.EvaluateRules(case["map"])
    assert got == ({"r": [case["expStr"]]} if case["expected"] else {})


# ---- group/finder/internal_test.go -------------------------------------------------------------------------------
@pytest.mark.parametrize("case", FINDER["valid_path"], ids=lambda c: c["message"])
def test_is_valid_field_path(case):
    assert group_ref.is_valid_field_path(case["fieldPath"], case["includePaths"], case["excludePaths"]) is case["expected"]
    # product: a document whose only leaf sits at that path is walked iff the path is valid (no expressions
    # registered, so nothing reaches the GPU; NOT-rules see the difference through an empty tag map either way) --
    # checked through the leaf counter
    g = group.GroupFinder(_finder_no_device())
    doc = {}
    cur = doc
    parts = case["fieldPath"].split(".")
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = "text"
    res = g.TagJsons([json.dumps(doc)], case["includePaths"], case["excludePaths"])
    leaves, _ = g.last_batch()
    assert leaves == (1 if case["expected"] else 0)
    assert ("error" in res[0]) == case["expected"]      # the one leaf needs the device, which is absent here


# ---- group/finder/finder_test.go: NewFinderWithRules / AddRule / AddRules ------------------------------------------
@pytest.mark.parametrize("case", FINDER["add_rules"], ids=lambda c: c["test"] + " " + c["message"])
def test_add_rules(case):
    ref = group_ref.GroupFinder(lambda t: [])
    g = group.GroupFinder(_finder_no_device())
    if case["error"] is not None:
        with pytest.raises(group_ref.GroupDslError) as ei:
            ref.add_rules(case["rules"])
        assert str(ei.value) == case["error"]
        with pytest.raises(group.GroupFinderError) as ei:
            g.AddRules(case["rules"])
        assert str(ei.value) == case["error"] and ei.value.code == _lib.GFT_E_PARSE
    else:
        ref.add_rules(case["rules"])
        g.AddRules(case["rules"])
    want = case["state"]
    assert sorted(ref.fields) == want["fields"] and sorted(ref.tags) == want["tags"]
    assert {k: [{"ExpressionString": s, "Expression": e.to_obj()} for s, e in v] for k, v in ref.rules.items()} == want["rules"]
    st = g.state()
    assert st["rules"] == want["rules"] and sorted(st["fields"]) == want["fields"] and sorted(st["tags"]) == want["tags"]
    assert sorted(g.GetFieldNames()) == want["fields"]


# ---- group/finder/finder_test.go: TestEvaluateRules -----------------------------------------------------------------
@pytest.mark.parametrize("case", FINDER["evaluate"], ids=lambda c: c["message"])
def test_evaluate_rules(case):
    ref = group_ref.GroupFinder(lambda t: [])
    ref.add_rules(case["rules"])
    assert ref.evaluate_rules(case["map"]) == case["expected"]
    g = group.NewFinderWithRules(_finder_no_device(), case["rules"])
    assert g.EvaluateRules(case["map"]) == case["expected"]


# ---- the JSON reader under TagJson (finder.go:80-92 hands the text to encoding/json) -------------------------------
def test_json_documents_without_string_leaves_need_no_device():
    g = group.NewFinderWithRules(_finder_no_device(), {"none": ['not "t"'], "some": ['"t"']})
    docs = ['{"a": 1, "b": [true, null, 2.5e3, {"c": {}}]}', "42", "[]", " {} ", "null", '{"a":{"a":{"a":[[[]]]}}}']
    res = g.ProcessJsons(docs)
    assert res == [{"rules": {"none": ['not "t"']}}] * len(docs)
    assert g.last_batch() == (0, 0)


@pytest.mark.parametrize("raw", ['{"a": }', '{"a" 1}', "[1, 2", '{"a": tru}', '{"a": "x\ny"}', '{"a": "\\q"}', "{} x", "",
                                 '{"a": 01}', '{"a": "\\u12G4"}', "[1,]", '{,}', '{"a": 1,}', "nul", "-", "1e", '"abc'])
def test_malformed_json_is_an_error_per_document(raw):
    with pytest.raises(ValueError):
        json.loads(raw)
    g = group.GroupFinder(_finder_no_device())
    res = g.ProcessJsons([raw, "{}"])
    assert "error" in res[0] and res[0]["error"]
    assert res[1] == {"rules": {}}
    with pytest.raises(group.GroupFinderError):
        g.ProcessJson(raw)


def test_json_string_decoding_matches_python():
    """leaves are decoded (escapes, surrogate pairs, duplicate keys: last wins) before they reach the finder; seen here
    through the byte count of the batch"""
    g = group.GroupFinder(_finder_no_device())
    cases = ['{"a": "\\u00e9\\ud83d\\ude00\\n\\"\\\\\\/\\b\\f\\r\\t"}', '{"k": "first", "k": "the last one wins"}',
             '["\\ud800", "\\udc00x", "\\ud800\\u0041"]', '{"a": {"b": ["x", {"c": "yz"}]}, "d": ""}']
    for raw in cases:
        want = []

        def leaves(v):
            if isinstance(v, str):
                want.append(v)
            elif isinstance(v, dict):
                for x in v.values():
                    leaves(x)
            elif isinstance(v, list):
                for x in v:
                    leaves(x)
        leaves(json.loads(raw))
        g.TagJsons([raw])
        n, nbytes = g.last_batch()
        assert n == len(want)
        # lone surrogates become U+FFFD (3 bytes) in Go; Python keeps them as surrogates
        assert nbytes == sum(len(s.encode("utf-8", "replace").replace(b"?", b"\xef\xbf\xbd")) for s in want), raw


def test_deeply_nested_json_on_a_small_stack():
    """untrusted JSON, 9 999 levels deep (encoding/json allows 10 000), evaluated on a thread with a 256 KiB stack: the
    parser, the teardown of the decoded value and the walk over it must not recurse per level"""
    import ctypes as C
    import threading
    from gofindthem_amd import _lib
    L = _lib.load()
    deep = ("[" * 9999 + "]" * 9999).encode()
    too_deep = ("[" * 10001 + "]" * 10001).encode()
    out = {}

    def run():
        buf = C.create_string_buffer(1 << 16)
        need = C.c_uint64(0)
        out["ok"] = L.gft_group_dsl_parse(b'"a"', 3, buf, len(buf), C.byref(need))
        # gft_to_lower etc. do not parse JSON; the group finder's evaluate entry point does
        fh = C.c_void_p()
        rc = L.gft_finder_create(C.byref(fh), 1, -1)
        gh = C.c_void_p()
        out["create"] = L.gft_group_create(C.byref(gh), fh)
        out["deep"] = L.gft_group_evaluate(gh, deep, len(deep), buf, len(buf), C.byref(need))
        out["deep_err"] = L.gft_group_last_error(gh)
        out["too_deep"] = L.gft_group_evaluate(gh, too_deep, len(too_deep), buf, len(buf), C.byref(need))
        out["too_deep_err"] = L.gft_group_last_error(gh)
        L.gft_group_destroy(gh)
        L.gft_finder_destroy(fh)
        out["rc"] = rc
    old = threading.stack_size(256 * 1024)
    try:
        t = threading.Thread(target=run)
        t.start()
        t.join()
    finally:
        threading.stack_size(old)
    assert out["deep"] == _lib.GFT_E_INVALID and b"expected" in out["deep_err"]           # parsed, then refused as a tag map
    assert out["too_deep"] == _lib.GFT_E_INVALID and b"exceeded max depth" in out["too_deep_err"]
