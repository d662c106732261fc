"""oracle/pyoracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes driver for oracle/liboracle.so (the CPU restatement in oracle/ac_oracle.cpp) plus the glue that
feeds it trees from oracle/dsl_ref.py.  Importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import dsl_ref

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

POS_START, POS_END = 0, 1


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ac_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u8p, u64p, u32p, i32p, i64p = (C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int64))
        L.orc_create.restype = vp
        L.orc_create.argtypes = [vp, vp, C.c_uint32, C.c_int]
        L.orc_destroy.argtypes = [vp]
        L.orc_n_terms.restype = C.c_uint32
        L.orc_n_terms.argtypes = [vp]
        L.orc_n_states.restype = C.c_uint32
        L.orc_n_states.argtypes = [vp]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [vp]
        L.orc_term.restype = C.c_uint32
        L.orc_term.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
        for f in (L.orc_scan_batch, L.orc_brute_batch):
            f.restype = C.c_uint64
            f.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, vp, vp, vp, C.c_uint64]
        L.orc_set_expressions.restype = C.c_int
        L.orc_set_expressions.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_uint32]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp]
        L.orc_process_batch.restype = C.c_int
        L.orc_process_batch.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, vp, vp, vp, vp, C.c_int]
        L.orc_scan_count_batch.restype = C.c_uint64
        L.orc_scan_count_batch.argtypes = [vp, vp, vp, C.c_uint64, C.c_int]
        _LIB = L
    return _LIB


def pack_strings(strs):
    """list of bytes/str -> (blob uint8 array, offsets uint64 array[n+1])"""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strs]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, dtype=np.uint8)[:0]
    return blob, off


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """CPU restatement of CloudflareForkEngine + Finder.ProcessText (see ac_oracle.cpp header)."""

    def __init__(self, terms, pos_mode=POS_START):
        blob, off = pack_strings(terms)
        self._L = lib()
        self._h = self._L.orc_create(_p(blob), _p(off), len(terms), pos_mode)
        self.n_terms = self._L.orc_n_terms(self._h)
        self.n_states = self._L.orc_n_states(self._h)
        self.n_exprs = 0
        self.literals = []

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def terms(self):
        out, buf = [], np.zeros(1 << 16, dtype=np.uint8)
        for i in range(self.n_terms):
            n = self._L.orc_term(self._h, i, _p(buf), buf.size)
            out.append(bytes(buf[:n]))
        return out

    def _run_scan(self, fn, blob, doc_off, fold):
        n_docs = len(doc_off) - 1
        moff = np.zeros(n_docs + 1, dtype=np.uint64)
        cap = max(1024, int(blob.size) // 4)
        while True:
            tid = np.zeros(cap, dtype=np.uint32)
            pos = np.zeros(cap, dtype=np.uint32)
            tot = fn(self._h, _p(blob), _p(doc_off), n_docs, int(fold), _p(moff), _p(tid), _p(pos), cap)
            if tot <= cap:
                return moff, tid[:tot].copy(), pos[:tot].copy()
            cap = int(tot)

    def scan(self, blob, doc_off, fold=False):
        """FindSubstrings over a batch -> CSR (match_off u64[n+1], term_id u32[H], pos u32[H])."""
        return self._run_scan(self._L.orc_scan_batch, blob, doc_off, fold)

    def brute(self, blob, doc_off, fold=False):
        return self._run_scan(self._L.orc_brute_batch, blob, doc_off, fold)

    def scan_count(self, blob, doc_off, n_threads=1):
        return int(self._L.orc_scan_count_batch(self._h, _p(blob), _p(doc_off), len(doc_off) - 1, n_threads))

    def set_expression_trees(self, trees):
        nodes, roots, lits = dsl_ref.flatten_forest(trees)
        tab = np.asarray(nodes, dtype=np.int32).reshape(-1, 5) if nodes else np.zeros((0, 5), np.int32)
        r = np.asarray(roots, dtype=np.int32)
        lb, lo = pack_strings(lits)
        rc = self._L.orc_set_expressions(self._h, _p(tab), tab.shape[0], _p(r), len(roots), _p(lb), _p(lo), len(lits))
        assert rc == 0
        self.n_exprs = len(roots)
        self.literals = lits

    def set_expressions(self, exprs, case_sensitive=True):
        """expression strings -> reference-shaped trees (dsl_ref) -> oracle.  Returns (keywords, regexes)."""
        trees, kws, rgx = [], {}, {}
        for e in exprs:
            t, k, r = dsl_ref.parse(e, case_sensitive)
            trees.append(t)
            kws.update(dict.fromkeys(k))
            rgx.update(dict.fromkeys(r))
        self.set_expression_trees(trees)
        return list(kws), list(rgx)

    def solve(self, expr_index, solver_map):
        """Expression.Solve on an explicit map {key: [positions] | None}."""
        keys = list(solver_map.keys())
        kb, ko = pack_strings(keys)
        lists = [solver_map[k] or [] for k in keys]
        po = np.zeros(len(keys) + 1, dtype=np.uint64)
        if keys:
            po[1:] = np.cumsum([len(x) for x in lists], dtype=np.uint64)
        flat = np.asarray([p for x in lists for p in x], dtype=np.int64)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.int64)
        rc = self._L.orc_solve(self._h, expr_index, _p(kb), _p(ko), len(keys), _p(flat), _p(po))
        if rc < 0:
            raise RuntimeError(self._L.orc_last_error(self._h).decode())
        return bool(rc)

    def process(self, blob, doc_off, fold=False, n_threads=1, extra=None):
        """Batch ProcessText -> uint32 hit bitmap [n_docs, ceil(E/32)].
        extra = (extra_off u64[n+1], literal_index i32[], pos i64[]) regex-engine hits, or None."""
        n_docs = len(doc_off) - 1
        words = (self.n_exprs + 31) // 32
        bm = np.zeros((n_docs, max(words, 1)), dtype=np.uint32)[:, :words]
        bm = np.ascontiguousarray(bm)
        eo = el = ep = None
        if extra is not None:
            eo, el, ep = extra
        rc = self._L.orc_process_batch(self._h, _p(blob), _p(doc_off), n_docs, int(fold), _p(eo), _p(el), _p(ep),
                                       _p(bm) if bm.size else None, n_threads)
        if rc < 0:
            raise RuntimeError(self._L.orc_last_error(self._h).decode())
        return bm
