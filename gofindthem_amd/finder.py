"""Python mirror of the reference's finder package (finder/finder.go, substringEngine.go, regexEngine.go) over
libgft.so's gft_finder_* C ABI.  Same names and error behaviour; Go `error` values surface as FinderError whose
text is the reference's message (parser errors, injected-engine errors) or the library's (GPU errors).

All scanning and solving happens on the GPU inside libgft.so; this module only marshals arguments.
"""
import ctypes as C
import json
import re

import numpy as np

from . import _lib
from .engine import Engine, GftError, pack


class FinderError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class Match:
    """finder.Match (finder/finder.go:11-14)"""
    __slots__ = ("Position", "Term")

    def __init__(self, Position, Term):
        self.Position, self.Term = Position, Term

    def __eq__(self, o):
        return (self.Position, self.Term) == (o.Position, o.Term)

    def __repr__(self):
        return "Match(%d, %r)" % (self.Position, self.Term)


class ExpressionResult:
    """finder.ExpressionResult (finder/finder.go:25-29; field names as in the reference)"""
    __slots__ = ("ExpresionIndex", "ExpresionStr", "Tag")

    def __init__(self, ExpresionIndex, ExpresionStr, Tag):
        self.ExpresionIndex, self.ExpresionStr, self.Tag = ExpresionIndex, ExpresionStr, Tag

    def to_obj(self):
        return {"ExpresionIndex": self.ExpresionIndex, "ExpresionStr": self.ExpresionStr, "Tag": self.Tag}

    def __repr__(self):
        return "ExpressionResult(%r)" % (self.to_obj(),)


class SubstringEngine:
    """finder.SubstringEngine (finder/substringEngine.go:11-18): raise an Exception to return a Go error."""

    def BuildEngine(self, keywords, caseSensitive):
        raise NotImplementedError

    def FindSubstrings(self, text):
        raise NotImplementedError


class RegexEngine:
    """finder.RegexEngine (finder/regexEngine.go:8-15)"""

    def BuildEngine(self, regexes, caseSensitive):
        raise NotImplementedError

    def FindRegexes(self, text):
        raise NotImplementedError


class EmptyEngine(SubstringEngine):
    def BuildEngine(self, keywords, caseSensitive):
        return None

    def FindSubstrings(self, text):
        return []


class EmptyRgxEngine(RegexEngine):
    def BuildEngine(self, regexes, caseSensitive):
        return None

    def FindRegexes(self, text):
        return []


class GpuEngine(SubstringEngine):
    """Drop-in for finder.CloudflareForkEngine backed by the HIP kernels (one gft_engine)."""

    def __init__(self, device=-1):
        self.engine = Engine(device)

    def BuildEngine(self, keywords, caseSensitive):
        try:
            self.engine.build(sorted(keywords))
        except GftError as e:
            raise FinderError(e.code, e.msg)

    def FindSubstrings(self, text):
        b = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        blob, off = pack([b])
        mo, ti, po = self.engine.scan(blob, off)
        terms = {}
        out = []
        for t, p in zip(ti.tolist(), po.tolist()):
            if t not in terms:
                terms[t] = self.engine.term(t).decode("utf-8", "surrogateescape")
            out.append(Match(p, terms[t]))
        return out


class PyRegexpEngine(RegexEngine):
    """Host-side stand-in for finder.RegexpEngine (finder/regexEngine.go:17-47): Python `re` instead of Go
    `regexp`; Position = byte offset of the start of each non-overlapping leftmost match, Term = regex source."""

    def __init__(self):
        self.compiled = []

    def BuildEngine(self, regexes, caseSensitive):
        self.compiled = [(r, re.compile(r.encode("utf-8") if isinstance(r, str) else r)) for r in regexes]

    def FindRegexes(self, text):
        b = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        out = []
        for src, rx in self.compiled:
            for m in rx.finditer(b):
                out.append(Match(m.start(), src))
        return out


def _mk_build(obj):
    def cb(user, blob, off, n, cs, err, cap):
        try:
            offs = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(n + 1,))
            raw = C.string_at(blob, int(offs[n])) if n else b""
            items = [raw[int(offs[i]):int(offs[i + 1])].decode("utf-8", "surrogateescape") for i in range(n)]
            obj.BuildEngine(items, bool(cs))
            return 0
        except Exception as e:   # Go: return err
            msg = str(e).encode("utf-8")[:cap - 1] + b"\0"
            C.memmove(err, msg, len(msg))
            return 1
    return _lib.BUILD_FN(cb)


def _mk_find(obj, method):
    def cb(user, text, n, emit, sink, err, cap):
        try:
            t = C.string_at(text, n).decode("utf-8", "surrogateescape")
            for m in getattr(obj, method)(t) or []:
                term = m.Term.encode("utf-8", "surrogateescape") if isinstance(m.Term, str) else bytes(m.Term)
                emit(sink, term, len(term), int(m.Position))
            return 0
        except Exception as e:
            msg = str(e).encode("utf-8")[:cap - 1] + b"\0"
            C.memmove(err, msg, len(msg))
            return 1
    return _lib.FIND_FN(cb)


class Finder:
    """finder.Finder (finder/finder.go:32-240).  NewFinder(subEng, rgxEng, caseSensitive)."""

    def __init__(self, subEng=None, rgxEng=None, caseSensitive=True, device=-1, allow_no_device=False, devices=None):
        """allow_no_device: keep the handle when no HIP device exists, so the host-only half (expression
        registry, parser, engine-build orchestration) can be exercised; every GPU step then fails with GFT_E_HIP."""
        self._L = _lib.load()
        h = C.c_void_p()
        if devices is not None:        # one finder over several devices (gft_finder_create_multi)
            arr = (C.c_int * len(devices))(*devices)
            rc = self._L.gft_finder_create_multi(C.byref(h), 1 if caseSensitive else 0, C.cast(arr, C.c_void_p), len(devices))
        else:
            rc = self._L.gft_finder_create(C.byref(h), 1 if caseSensitive else 0, device)
        self._h = h
        if rc != 0 and not (allow_no_device and rc == _lib.GFT_E_HIP and h):
            msg = self._L.gft_finder_last_error(h).decode() if h else "gft_finder_create failed"
            self.close()
            raise FinderError(rc, msg)
        self.caseSensitive = caseSensitive
        self._keep = []
        if subEng is not None and not isinstance(subEng, GpuEngine):
            b, f = _mk_build(subEng), _mk_find(subEng, "FindSubstrings")
            self._keep += [b, f, subEng]
            self._check(self._L.gft_finder_set_substring_engine(self._h, C.cast(b, C.c_void_p), C.cast(f, C.c_void_p), None))
        if rgxEng is not None and not isinstance(rgxEng, EmptyRgxEngine):
            b, f = _mk_build(rgxEng), _mk_find(rgxEng, "FindRegexes")
            self._keep += [b, f, rgxEng]
            self._check(self._L.gft_finder_set_regex_engine(self._h, C.cast(b, C.c_void_p), C.cast(f, C.c_void_p), None))

    def close(self):
        if getattr(self, "_h", None):
            self._L.gft_finder_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise FinderError(rc, self._L.gft_finder_last_error(self._h).decode("utf-8", "replace"))

    # -- registry -------------------------------------------------------------------------------------
    def AddExpression(self, expression):
        return self.AddExpressionWithTag(expression, "")

    def AddExpressions(self, expressions):
        for e in expressions:
            self.AddExpressionWithTag(e, "")

    def AddExpressionsWithTag(self, expressions, tag):
        for e in expressions:
            self.AddExpressionWithTag(e, tag)

    def AddExpressionWithTag(self, expression, tag):
        e = expression.encode("utf-8") if isinstance(expression, str) else bytes(expression)
        t = tag.encode("utf-8") if isinstance(tag, str) else bytes(tag)
        self._check(self._L.gft_finder_add_expression(self._h, e, len(e), t, len(t)))

    def _literals(self, which):
        out = []
        p, n = C.c_void_p(), C.c_uint32()
        for i in range(self._L.gft_finder_n_literals(self._h, which)):
            self._check(self._L.gft_finder_literal(self._h, which, i, C.byref(p), C.byref(n)))
            out.append(C.string_at(p, n.value).decode("utf-8", "surrogateescape"))
        return out

    def GetKeywords(self):
        return set(self._literals(0))

    def GetRegexes(self):
        return set(self._literals(1))

    @property
    def n_expressions(self):
        return self._L.gft_finder_n_expressions(self._h)

    def expression(self, i, tree=True):
        """-> (exprString, tag, tree as dict); tree=False: source and tag only (the reference's own benchmarks build chains
        of 10 000 leaves, benchmarks/benchmark_test.go:56 -- a tree that deep is not something json.loads should recurse into)"""
        ptrs = [C.c_void_p() for _ in range(3)]
        lens = [C.c_uint32() for _ in range(3)]
        self._check(self._L.gft_finder_expression(self._h, i, C.byref(ptrs[0]), C.byref(lens[0]), C.byref(ptrs[1]),
                                                  C.byref(lens[1]), C.byref(ptrs[2]) if tree else None,
                                                  C.byref(lens[2]) if tree else None))
        s, t = (C.string_at(p, n.value) for p, n in zip(ptrs[:2], lens[:2]))
        j = json.loads(C.string_at(ptrs[2], lens[2].value).decode("utf-8")) if tree else None
        return s.decode("utf-8", "surrogateescape"), t.decode("utf-8", "surrogateescape"), j

    # -- processing -----------------------------------------------------------------------------------
    def ForceBuild(self):
        self._check(self._L.gft_finder_force_build(self._h))

    def ProcessText(self, text):
        """-> list of ExpressionResult for the expressions that are true (registration order)"""
        b = text.encode("utf-8", "surrogateescape") if isinstance(text, str) else bytes(text)
        cap = max(self.n_expressions, 1)
        idx = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint32()
        self._check(self._L.gft_finder_process_text(self._h, b, len(b), idx.ctypes.data, cap, C.byref(n)))
        cache = self.__dict__.setdefault("_expr_cache", [])      # (source, tag) per registered expression
        while len(cache) < self.n_expressions:
            s, t, _ = self.expression(len(cache), tree=False)
            cache.append((s, t))
        return [ExpressionResult(i, cache[i][0], cache[i][1]) for i in idx[:n.value].tolist()]

    def ProcessTexts(self, texts=None, blob=None, doc_off=None, out=None):
        """batch extension -> uint32 bitmap [n_docs, ceil(E/32)]; `out`: a caller's array of that shape to fill instead of a
        fresh one (a Go caller keeps its []uint32 from batch to batch: no page of the result is touched for the first time)"""
        if texts is not None:
            blob, doc_off = pack(texts)
        n_docs = len(doc_off) - 1
        words = (self.n_expressions + 31) // 32
        if out is not None:
            if out.dtype != np.uint32 or out.shape != (n_docs, words) or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous uint32 array of shape (n_docs, ceil(n_expressions / 32))")
            bm = out
        else:
            bm = np.zeros((n_docs, words), dtype=np.uint32)
        self._check(self._L.gft_finder_process_texts(self._h, blob.ctypes.data, doc_off.ctypes.data, n_docs,
                                                     bm.ctypes.data if bm.size else None))
        return bm

    def ProcessDevice(self, d_text_ptr, d_doc_off_ptr, n_docs, d_bitmap_ptr):
        self._check(self._L.gft_finder_process_device(self._h, d_text_ptr, d_doc_off_ptr, n_docs, d_bitmap_ptr))

    def ProcessDeviceBegin(self, d_text_ptr, d_doc_off_ptr, n_docs, d_bitmap_ptr):
        """pipelined ProcessDevice: enqueue the batch and return; ProcessDeviceEnd() completes the oldest batch begun (at
        most two in flight; inputs and bitmap stay untouched until then)"""
        self._check(self._L.gft_finder_process_device_begin(self._h, d_text_ptr, d_doc_off_ptr, n_docs, d_bitmap_ptr))

    def ProcessDeviceEnd(self):
        self._check(self._L.gft_finder_process_device_end(self._h))

    def engine_handle(self):
        return self._L.gft_finder_engine(self._h)

    # -- test hooks (finder_test.go pokes struct fields) --------------------------------------------------
    def debug_add_literal(self, which, lit):
        b = lit.encode("utf-8")
        self._check(self._L.gft_finder_debug_add_literal(self._h, which, b, len(b)))

    def debug_set_updated(self, sub, rgx):
        self._check(self._L.gft_finder_debug_set_updated(self._h, int(sub), int(rgx)))

    def debug_get_updated(self):
        a, b = C.c_int(), C.c_int()
        self._check(self._L.gft_finder_debug_get_updated(self._h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)


def NewFinder(subEng, rgxEng, caseSensitive, device=-1):
    return Finder(subEng, rgxEng, caseSensitive, device)
