"""The solver's program compiler on the host (no GPU): gft_debug_eval_programs takes public postfix programs through the
steps of gft_set_programs -- check, fusion (NOT pushed to the leaves, leaf operands folded into their operators, the
deeper operand first, push + set in one word), control-bit device words -- and interprets the device words for one
document.  Checked against a direct evaluation of the reference-shaped trees (dsl/expression.go:66-127)."""
import ctypes as C

import numpy as np
import pytest

from gofindthem_amd import _lib
from helpers import tree_to_program
from oracle import dsl_ref


def truth(n, present):
    if n.Type == dsl_ref.UNIT_EXPR:
        return n.Literal in present
    if n.Type == dsl_ref.AND_EXPR:
        a, b = truth(n.LExpr, present), truth(n.RExpr, present)      # (every node is evaluated: no short-circuit)
        return a and b
    if n.Type == dsl_ref.OR_EXPR:
        a, b = truth(n.LExpr, present), truth(n.RExpr, present)
        return a or b
    if n.Type == dsl_ref.NOT_EXPR:
        return not truth(n.RExpr, present)
    if n.Type == dsl_ref.INORD_EXPR:                                 # (only one-leaf groups here: presence)
        return truth(n.RExpr, present)
    raise ValueError(n.Type)


def evaluate(exprs, letters, presence_sets):
    L = _lib.load()
    slot = {l: i for i, l in enumerate(letters)}
    trees = [dsl_ref.parse(e, True)[0] for e in exprs]
    progs = [tree_to_program(t, lambda lit: slot[lit]) for t in trees]
    words = np.asarray([w for p in progs for w in p], dtype=np.uint32)
    off = np.zeros(len(progs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in progs])
    depth = np.zeros(len(progs), dtype=np.uint32)
    for present in presence_sets:
        p8 = np.asarray([1 if l in present else 0 for l in letters], dtype=np.uint8)
        hit = np.zeros(len(progs), dtype=np.uint8)
        rc = L.gft_debug_eval_programs(words.ctypes.data, off.ctypes.data, len(progs), len(letters), p8.ctypes.data,
                                       hit.ctypes.data, depth.ctypes.data)
        assert rc == 0, rc
        want = [truth(t, present) for t in trees]
        assert hit.astype(bool).tolist() == want, [e for e, h, w in zip(exprs, hit, want) if bool(h) != w][:3]
    return depth


def nested(rng, letters, depth, side):
    """both operands of every operator are subtrees; the chain nests to the right, to the left, or on both sides"""
    def pair():
        a, b = rng.choice(letters, 2)
        return '(%s"%s" %s "%s")' % ("not " if rng.integers(3) == 0 else "", a, "and" if rng.integers(2) else "or", b)
    if side == "both":
        if depth == 0:
            return pair()
        return "(%s %s %s)" % (nested(rng, letters, depth - 1, side), "and" if rng.integers(2) else "or",
                               nested(rng, letters, depth - 1, side))
    e = pair()
    for _ in range(depth):
        op = "and" if rng.integers(2) else "or"
        e = "(%s %s %s)" % ((pair(), op, e) if side == "right" else (e, op, pair()))
        if rng.integers(4) == 0:
            e = "not " + e
    return e


def sets(rng, letters, n):
    return [set(l for l in letters if rng.integers(2)) for _ in range(n)] + [set(), set(letters)]


def test_random_expressions_match_the_tree_evaluation():
    rng = np.random.default_rng(5)
    letters = list("abcdefghij")
    exprs = [nested(rng, letters, d, s) for d in (0, 1, 2, 3, 5, 9, 30) for s in ("right", "left") for _ in range(6)]
    exprs += [nested(rng, letters, d, "both") for d in (1, 2, 3, 4, 5) for _ in range(4)]
    exprs += ['"a"', 'not "a"', '"a" and "b" or not "c" and "d"', 'not ("a" or "b")', 'not (not ("a" and not "b"))',
              'inord("a")', 'not (inord("a"))', '"b" and not (inord("a")) or "c"']
    evaluate(exprs, letters, sets(rng, letters, 40))


def test_operand_order_keeps_the_stack_shallow():
    """Sethi-Ullman: a chain of parentheses nested to ONE side needs one stack entry however long it is (the operand
    that needs the deeper stack goes first); only a balanced tree of 2^k subtrees gets k deep."""
    rng = np.random.default_rng(6)
    letters = list("abcdefgh")
    right = [nested(rng, letters, d, "right") for d in (1, 4, 17, 60)]
    left = [nested(rng, letters, d, "left") for d in (1, 4, 17, 60)]
    both = [nested(rng, letters, d, "both") for d in (1, 2, 3, 4, 5)]
    flat = ['"a" and "b" or "c" and not "d"', '"a"']
    depth = evaluate(right + left + both + flat, letters, sets(rng, letters, 5))
    assert depth[:8].tolist() == [1] * 8
    assert depth[8:13].tolist() == [1, 2, 3, 4, 5]
    assert depth[13:].tolist() == [0, 0]


def test_benchmark_expressions_fit_the_register_stack():
    from gofindthem_amd.workload import Workload, make_expressions
    w = Workload(2000)
    terms = [t.decode() for t in w.terms()]
    exprs = make_expressions(w.terms(), 300, cover=True)
    rng = np.random.default_rng(7)
    presence = [set(rng.choice(terms, 300).tolist()) for _ in range(6)]
    depth = evaluate(exprs, sorted(set(terms)), presence)
    assert depth.max() <= 2 and (depth == 0).sum() > 50


def test_malformed_programs_are_refused():
    L = _lib.load()
    hit = np.zeros(1, dtype=np.uint8)
    p8 = np.ones(2, dtype=np.uint8)
    for words in ([2 << 28], [1 << 28 | 0, 1 << 28 | 1], [1 << 28 | 5], [1 << 28 | 0, 1 << 28 | 1, 2 << 28, 5 << 28, 2 << 28]):
        w = np.asarray(words, dtype=np.uint32)
        off = np.asarray([0, len(words)], dtype=np.uint64)
        rc = L.gft_debug_eval_programs(w.ctypes.data, off.ctypes.data, 1, 2, p8.ctypes.data, hit.ctypes.data, None)
        assert rc != 0, words
