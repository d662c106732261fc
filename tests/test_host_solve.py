"""The product's host solver (csrc/host_solve.cpp, behind gft_debug_host_solve) -- what gft_process* run for the
(expression, document) pairs the device solver does not answer itself: expressions beyond its limits, and INORD expressions
over a slot whose position list is not ascending (a keyword and a regex with the same literal: finder/finder.go:181-196).

Pinned by the reference's own table (dsl/expression_test.go:21-313, all 31 cases incl. keys with nil lists) and, on lists
that are NOT sorted, by the oracle's literal restatement of dsl/expression.go:66-225 on random trees."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from gofindthem_amd import _lib
from helpers import tree_to_program
from oracle import dsl_ref
from oracle.pyoracle import Oracle


def host_solve(expr, mapping, case_sensitive=True):
    """mapping: literal -> list of positions (or None), in the order addMatchesToSolverMap would have appended them"""
    L = _lib.load()
    tree, kws, rgx = dsl_ref.parse(expr, case_sensitive)
    lits = list(dict.fromkeys(list(kws) + list(rgx) + list(mapping)))
    slot = {l: i for i, l in enumerate(lits)}
    words = np.asarray(tree_to_program(tree, lambda lit: slot[lit]), dtype=np.uint32)
    keys = list(mapping)
    slots = np.asarray([slot[k] for k in keys], dtype=np.uint32)
    lists = [list(mapping[k] or []) for k in keys]
    off = np.zeros(len(keys) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(v) for v in lists])
    pos = np.asarray([p for v in lists for p in v] + [0], dtype=np.int64)
    out = C.c_int(-1)
    rc = L.gft_debug_host_solve(words.ctypes.data, len(words), slots.ctypes.data, off.ctypes.data, pos.ctypes.data, len(keys), C.byref(out))
    assert rc == 0, rc
    return bool(out.value)


@pytest.mark.parametrize("case", load_golden("solver.json")["cases"], ids=lambda c: c["message"])
def test_reference_solver_table(case):
    assert host_solve(case["expStr"], case["map"]) is case["expected"]


def test_readme_inord_example():
    ex = load_golden("examples.json")["readme_inord"]
    assert host_solve(ex["expStr"], ex["map"]) is ex["expected"]


def test_unsorted_list_is_searched_like_the_reference():
    """keyword "aa" at [2 3 4 5] and regex r"aa" at [2 4] share the key "aa": the list is [2 3 4 5 2 4].  inord("aaay" and
    "aa") asks getLowestIdxGTVal for an element > 4 (dsl/expression.go:175-189): the binary search probes index 2 (4), 4
    (2), 5 (4) and finds none -- although 5 is in the list.  A sorted set would answer 5."""
    assert host_solve('inord("aaay" and "aa")', {"aa": [2, 3, 4, 5, 2, 4], "aaay": [4]}) is False
    assert host_solve('inord("aaay" and "aa")', {"aa": [2, 3, 4, 5], "aaay": [4]}) is True
    # ... and the other way round: the search lands on an element that a sorted set would not offer first, true both ways
    assert host_solve('inord("x" and "aa")', {"aa": [7, 8, 1, 9], "x": [3]}) is True
    # the FIRST element of the left list is compared, not its minimum (expression.go:89): [9 1] ++ nothing > 9
    assert host_solve('inord("aa" and "y")', {"aa": [9, 1], "y": [5]}) is False
    assert host_solve('inord("aa" and "y")', {"aa": [1, 9], "y": [5]}) is True


@pytest.mark.parametrize("seed", range(12))
def test_random_trees_on_unsorted_lists_against_the_oracle(seed):
    rng = np.random.default_rng(seed)
    lits = [chr(ord("a") + i) for i in range(6)]

    def tree(depth, inord):
        if depth == 0 or rng.integers(4) == 0:
            return '"%s"' % rng.choice(lits)
        k = rng.integers(5 if not inord else 2)
        if k == 0:
            return "(%s and %s)" % (tree(depth - 1, inord), tree(depth - 1, inord))
        if k == 1:
            return "(%s or %s)" % (tree(depth - 1, inord), tree(depth - 1, inord))
        if k == 2:
            return "not (%s)" % tree(depth - 1, inord)
        return "inord(%s)" % tree(depth - 1, True)

    o = Oracle([])
    for _ in range(40):
        e = tree(4, False)
        o.set_expressions([e], True)
        for _ in range(8):
            m = {}
            for l in lits:
                if rng.integers(3):
                    n = int(rng.integers(0, 6))
                    v = rng.integers(0, 12, n).tolist()
                    if rng.integers(2):
                        v.sort()
                    m[l] = v if (v or rng.integers(2)) else None
            assert host_solve(e, m) is o.solve(0, m), (e, m)


def test_left_deep_chain_of_10000_leaves_does_not_recurse():
    """benchmarks/benchmark_test.go:56,182-184: exp10000 is INORD of 10 000 AND-ed terms"""
    n = 10000
    words = [1 << 28 | 1 << 27 | 0]
    for i in range(1, n):
        words += [1 << 28 | 1 << 27 | i, 2 << 28 | 1 << 27]
    words.append(5 << 28)
    words = np.asarray(words, dtype=np.uint32)
    L = _lib.load()
    slots = np.arange(n, dtype=np.uint32)
    off = np.arange(n + 1, dtype=np.uint64)
    out = C.c_int(-1)
    pos = np.arange(n, dtype=np.int64)                          # term i at position i: in order
    assert L.gft_debug_host_solve(words.ctypes.data, len(words), slots.ctypes.data, off.ctypes.data, pos.ctypes.data, n, C.byref(out)) == 0
    assert out.value == 1
    pos[n // 2] = 0                                             # one term only occurs before its predecessor
    assert L.gft_debug_host_solve(words.ctypes.data, len(words), slots.ctypes.data, off.ctypes.data, pos.ctypes.data, n, C.byref(out)) == 0
    assert out.value == 0
