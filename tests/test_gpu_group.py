"""SURVEY.md 8(f) row 2 on the GPU: the group finder end to end (group/finder/finder_test.go:269-447,
examples/group/finder/main.go), every string leaf of a call scanned and solved by the HIP kernels in one batch.
Checked against the reference's tables (tests/golden/group_finder.json) and, on generated JSON documents, against the
oracle's restatement of the walk driven by the CPU oracle's ProcessText."""
import json

import numpy as np
import pytest

from conftest import load_golden
from gofindthem_amd import group
from gofindthem_amd.finder import EmptyRgxEngine, Finder, GpuEngine
from oracle import group_ref
from oracle.pyoracle import Oracle, pack_strings

pytestmark = pytest.mark.gpu
FIX = load_golden("group_finder.json")


def canon(tagmap):
    return {t: {f: sorted(v) for f, v in fs.items()} for t, fs in tagmap.items()}


class GoStruct:
    """stands in for a Go struct value: attributes with a lower-case first letter are unexported"""

    def __init__(self, d):
        for k, v in d.items():
            setattr(self, k, GoStruct(v) if isinstance(v, dict) else v)


def make_finder(expressions, case_sensitive):
    f = Finder(GpuEngine(), EmptyRgxEngine(), case_sensitive)
    for e, tag in expressions:
        f.AddExpressionWithTag(e, tag)
    return f


# ---- group/finder/finder_test.go: TestTagObject / TestTagText / TestTagJson ------------------------------------------
def test_tag_object_text_json_tables():
    sec = FIX["tag_object"]
    f = make_finder([(x["expression"], x["tag"]) for x in sec["finder_expressions"]], sec["case_sensitive"])
    g = group.NewFinderWithRules(f, sec["rules"])
    for c in sec["cases"]:
        obj = GoStruct(c["object"]) if c.get("struct") else c["object"]
        assert canon(g.TagObject(obj)) == c["expected"], c["message"]
    assert g.TagText(sec["tag_text"]["text"]) == sec["tag_text"]["expected"]
    # TestTagJson registers no expression at all: nothing can be tagged
    g0 = group.NewFinder(make_finder([], False))
    assert g0.TagJson(sec["tag_json_no_expressions"]["raw"]) == sec["tag_json_no_expressions"]["expected"]


# ---- examples/group/finder/main.go ---------------------------------------------------------------------------------
def test_group_example_program():
    ex = FIX["example"]
    f = make_finder([(e, tag) for tag, es in ex["finder_rules"].items() for e in es], ex["case_sensitive"])
    g = group.NewFinderWithRules(f, ex["rules"])
    names = g.GetFieldNames()
    assert sorted(names) == ["Field3", "Field3.SomeField1"]
    obj = GoStruct(ex["object"])
    assert canon(g.TagObject(obj, names, None)) == ex["expected_tags_with_field_names"]
    assert g.ProcessObject(obj, names, None) == ex["expected_rules_with_field_names"]
    assert canon(g.TagObject(obj)) == ex["expected_tags_all_fields"]
    arr = [GoStruct(x) for x in ex["array"]]
    assert canon(g.TagObject(arr)) == ex["expected_array_tags"]
    assert g.ProcessObject(arr) == ex["expected_array_rules"]
    assert canon(g.TagJson(ex["rawJson"], names, None)) == ex["expected_tags_with_field_names"]
    assert g.ProcessJson(ex["rawJson"], names, None) == ex["expected_rules_with_field_names"]
    # exclude wins over include (internal.go:100-107)
    assert g.ProcessJson(ex["rawJson"], names, ["Field3.SomeField1"]) == {
        "rule1": ['"tag1" or "tag2"'], "rule3": ['"tag3:Field3" or "tag4"']}
    # the batch extension returns the same per document
    res = g.ProcessJsons([ex["rawJson"], "{}", ex["rawJson"]], names, None)
    assert [r["rules"] for r in res] == [ex["expected_rules_with_field_names"], {}, ex["expected_rules_with_field_names"]]


# ---- generated documents against the oracle ---------------------------------------------------------------------------
def _random_docs(rng, w, n_docs):
    text, off = w.docs_host(0, 4 * n_docs)
    pieces = [bytes(text[int(off[i]):int(off[i + 1])]).decode("ascii") for i in range(4 * n_docs)]
    keys = ["Title", "Body", "Meta", "Notes", "Author", "items"]
    docs = []

    def value(depth, k):
        r = rng.random()
        p = pieces[int(rng.integers(len(pieces)))]
        s = p[:int(rng.integers(0, 400))]
        if r < 0.45 or depth > 2:
            return s.upper() if rng.random() < 0.2 else s
        if r < 0.55:
            return [int(rng.integers(100)), None, True][int(rng.integers(3))]
        if r < 0.8:
            return [value(depth + 1, k) for _ in range(int(rng.integers(0, 4)))]
        return {kk: value(depth + 1, kk) for kk in rng.choice(keys, int(rng.integers(1, 4)), replace=False)}
    for _ in range(n_docs):
        docs.append({kk: value(0, kk) for kk in rng.choice(keys, int(rng.integers(1, 6)), replace=False)})
    return docs


@pytest.mark.parametrize("seed", [0, 1])
def test_generated_json_documents_match_the_oracle(seed):
    from gofindthem_amd.workload import Workload, make_expressions
    rng = np.random.default_rng(seed)
    w = Workload(300)
    terms = w.terms()
    exprs = make_expressions(terms, 60, inord_fraction=0.3)
    tags = ["tag%d" % (i % 7) for i in range(len(exprs))]
    f = make_finder(list(zip(exprs, tags)), False)
    rules = {"r%d" % i: [r] for i, r in enumerate([
        '"tag0" and "tag1"', '"tag2:Body" or "tag3:Meta.Notes"', 'not "tag4" and ("tag5:items" or "tag6")',
        '"tag1:Title" and not "tag2:Body.index(0)"', '"tag0:Meta" or "tag0:Notes" or "tag0:Author"', 'not ("tag3" or "tag5")'])}
    g = group.NewFinderWithRules(f, rules)
    # oracle: the same expressions through the CPU restatement, one ProcessText per leaf like the reference
    o = Oracle(sorted(f.GetKeywords()))
    o.set_expressions(exprs, False)

    def process_text(text):
        blob, off = pack_strings([text.encode("utf-8")])
        bm = o.process(blob, off, fold=True)
        return [(tags[i], exprs[i]) for i in range(len(exprs)) if bm[0, i >> 5] >> (i & 31) & 1]
    ref = group_ref.GroupFinder(process_text)
    ref.add_rules(rules)
    docs = _random_docs(rng, w, 40)
    raws = [json.dumps(d) for d in docs]
    for inc, exc in [(None, None), (["Body", "Meta"], None), (None, ["Meta.Notes", "items"]), (g.GetFieldNames(), ["Body.index(1)"])]:
        got_tags = g.TagJsons(raws, inc, exc)
        got_rules = g.ProcessJsons(raws, inc, exc)
        n_hits = 0
        for d, raw in enumerate(raws):
            want = ref.tag_json(raw, inc, exc)
            assert canon(got_tags[d]["tags"]) == canon(want), (d, inc, exc)
            assert got_rules[d]["rules"] == ref.evaluate_rules(want), (d, inc, exc)
            n_hits += len(want)
        assert n_hits > 0
    leaves, nbytes = g.last_batch()
    assert leaves > 0 and nbytes > 0
    # objects take the same path as their JSON form
    assert g.ProcessObject(docs[0]) == ref.process_object(docs[0])
    assert g.ProcessObject(docs[3], ["Body"], None) == ref.process_object(docs[3], ["Body"], None)


def test_finder_errors_surface_per_document():
    """an expression that parses but cannot be solved (dsl/expression.go:139-141) fails every ProcessText, hence every
    document that has a taggable leaf; documents without one are still evaluated (internal.go:28-31)"""
    f = make_finder([('"a" "b" and "c"', "t")], True)
    g = group.NewFinderWithRules(f, {"r": ['not "t"']})
    res = g.ProcessJsons(['{"x": "abc"}', '{"x": 1}'])
    assert res[0] == {"error": "unable to process expression type 0"}
    assert res[1] == {"rules": {"r": ['not "t"']}}
