"""Shared test helpers (test infrastructure)."""
import numpy as np

from oracle import dsl_ref
from oracle.pyoracle import pack_strings

OP_UNIT, OP_AND, OP_OR, OP_NOT, OP_INORD = 1, 2, 3, 4, 5
INORD_FLAG = 1 << 27


def tree_to_program(e, slot_of):
    """dsl_ref.Expression -> postfix uint32 words of include/gft.h (independent of the product's own
    compiler in csrc/dsl_compile.cpp, so the two can be checked against each other)."""
    out = []

    def walk(n):
        fl = INORD_FLAG if n.Inord else 0
        if n.Type == dsl_ref.UNIT_EXPR:
            out.append(OP_UNIT << 28 | fl | slot_of(n.Literal))
        elif n.Type in (dsl_ref.AND_EXPR, dsl_ref.OR_EXPR):
            walk(n.LExpr)
            walk(n.RExpr)
            out.append((OP_AND if n.Type == dsl_ref.AND_EXPR else OP_OR) << 28 | fl)
        elif n.Type == dsl_ref.NOT_EXPR:
            walk(n.RExpr)
            out.append(OP_NOT << 28)
        elif n.Type == dsl_ref.INORD_EXPR:
            walk(n.RExpr)
            out.append(OP_INORD << 28)
        else:
            raise ValueError("unexpected node type %d" % n.Type)

    walk(e)
    return out


def docs(texts):
    return pack_strings(texts)


def csr_lists(moff, tid, pos):
    return [list(zip(tid[int(moff[d]):int(moff[d + 1])].tolist(), pos[int(moff[d]):int(moff[d + 1])].tolist()))
            for d in range(len(moff) - 1)]


def assert_csr_equal(a, b):
    for x, y, name in zip(a, b, ("match_off", "term_id", "pos")):
        assert x.shape == y.shape, (name, x.shape, y.shape)
        if not np.array_equal(x, y):
            bad = np.nonzero(x != y)[0][:5]
            raise AssertionError("%s differs at %s: %s vs %s" % (name, bad, x[bad], y[bad]))
