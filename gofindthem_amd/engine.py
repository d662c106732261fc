"""GpuEngine: the batch-capable SubstringEngine (finder/substringEngine.go:11-18) over libgft.so."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GftExtra, GftMatches


class GftError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gft error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def pack(strs):
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strs]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy()
    return blob, off


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """Thin object wrapper of the C ABI (one gft_engine handle)."""

    def __init__(self, device=-1, devices=None):
        """devices: a list of HIP device ordinals -> one handle over all of them (gft_engine_create_multi)"""
        self._L = _lib.load()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = self._L.gft_engine_create_multi(C.byref(h), C.cast(arr, C.c_void_p), len(devices))
        else:
            rc = self._L.gft_engine_create(C.byref(h), device)
        self._h = h
        self.warning = None
        if rc == _lib.GFT_W_NO_RCCL:          # a complete handle whose gathers are device-to-device copies, not RCCL's
            self.warning = self._L.gft_last_error(h).decode()
            rc = 0
        if rc != 0:
            msg = self._L.gft_last_error(h).decode() if h else "engine_create failed"
            self.close()
            raise GftError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gft_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise GftError(rc, self._L.gft_last_error(self._h).decode())

    def gather_mode(self):
        """"rccl" / "copy" / "" -- how gft_process_device_multi moves the shards' bitmaps (gft_gather_mode)"""
        return self._L.gft_gather_mode(self._h).decode()

    def set_stream(self, stream_ptr):
        self._check(self._L.gft_set_stream(self._h, stream_ptr))

    # -- SubstringEngine.BuildEngine ---------------------------------------------------------------
    def build(self, terms, pos_end=False):
        blob, off = pack(terms)
        self._check(self._L.gft_build(self._h, _p(blob), _p(off), len(terms), 1 if pos_end else 0))

    def export_tables(self):
        """the compiled tables of the current dictionary as bytes (gft_export_tables)"""
        need = C.c_uint64(0)
        self._L.gft_export_tables(self._h, None, 0, C.byref(need))
        if not need.value:
            self._check(_lib.GFT_E_NOT_BUILT)
        buf = (C.c_uint8 * need.value)()
        self._check(self._L.gft_export_tables(self._h, C.cast(buf, C.c_void_p), need.value, C.byref(need)))
        return bytes(buf)

    def import_tables(self, blob):
        """install tables written by export_tables: same effect as the build() that produced them"""
        self._check(self._L.gft_import_tables(self._h, blob, len(blob)))

    @property
    def n_terms(self):
        return self._L.gft_n_terms(self._h)

    @property
    def n_states(self):
        return self._L.gft_n_states(self._h)

    def term(self, i):
        p, n = C.c_void_p(), C.c_uint32()
        self._check(self._L.gft_term(self._h, i, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value)

    def terms(self):
        return [self.term(i) for i in range(self.n_terms)]

    def term_id(self, t):
        b = t.encode() if isinstance(t, str) else bytes(t)
        return int(self._L.gft_term_id(self._h, b, len(b)))

    # -- SubstringEngine.FindSubstrings, batched ------------------------------------------------------
    def scan(self, blob, doc_off, fold=False, unique=False, runes=False):
        """host numpy in -> CSR numpy out (match_off u64, term_id u32, pos u32).  unique=True: CloudflareEngine's output
        (GFT_SCAN_UNIQUE: every term once per document, first-occurrence order, positions 0)"""
        m = GftMatches()
        n_docs = len(doc_off) - 1
        self._check(self._L.gft_scan(self._h, _p(blob), _p(doc_off), n_docs, (1 if fold else 0) | (2 if unique else 0) | (4 if runes else 0), C.byref(m)))
        nm = int(m.n_matches)
        mo = np.ctypeslib.as_array(C.cast(m.match_off, C.POINTER(C.c_uint64)), shape=(n_docs + 1,)).copy()
        if nm == 0:
            return mo, np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        ti = np.ctypeslib.as_array(C.cast(m.term_id, C.POINTER(C.c_uint32)), shape=(nm,)).copy()
        po = np.ctypeslib.as_array(C.cast(m.pos, C.POINTER(C.c_uint32)), shape=(nm,)).copy()
        return mo, ti, po

    def scan_device(self, d_text_ptr, d_doc_off_ptr, n_docs, fold=False, unique=False):
        """device pointers in -> GftMatches with DEVICE pointers (valid until the next call)"""
        m = GftMatches()
        self._check(self._L.gft_scan_device(self._h, d_text_ptr, d_doc_off_ptr, n_docs, (1 if fold else 0) | (2 if unique else 0), C.byref(m)))
        return m

    # -- solver ---------------------------------------------------------------------------------------
    def set_programs(self, programs, n_extra=0):
        """programs: list of lists of uint32 words"""
        off = np.zeros(len(programs) + 1, dtype=np.uint64)
        if programs:
            off[1:] = np.cumsum([len(p) for p in programs], dtype=np.uint64)
        words = np.asarray([w for p in programs for w in p] or [0], dtype=np.uint32)
        self._check(self._L.gft_set_programs(self._h, _p(words), _p(off), len(programs), n_extra))

    @property
    def n_exprs(self):
        return self._L.gft_n_exprs(self._h)

    def process(self, blob, doc_off, fold=False, extra=None):
        """host numpy in -> uint32 bitmap [n_docs, ceil(E/32)]"""
        n_docs = len(doc_off) - 1
        words = (self.n_exprs + 31) // 32
        bm = np.zeros((n_docs, words), dtype=np.uint32)
        x = None
        if extra is not None:
            eo, es, ep = extra
            keep = (eo, es, ep)  # noqa: F841  (keep arrays alive across the call)
            x = GftExtra(eo.ctypes.data, es.ctypes.data, ep.ctypes.data)
        self._check(self._L.gft_process(self._h, _p(blob), _p(doc_off), n_docs, 1 if fold else 0,
                                        C.byref(x) if x is not None else None, _p(bm) if bm.size else None))
        return bm

    def process_device(self, d_text_ptr, d_doc_off_ptr, n_docs, d_bitmap_ptr, fold=False, d_extra=None):
        x = None
        if d_extra is not None:
            x = GftExtra(*d_extra)
        self._check(self._L.gft_process_device(self._h, d_text_ptr, d_doc_off_ptr, n_docs, 1 if fold else 0,
                                               C.byref(x) if x is not None else None, d_bitmap_ptr))

    # -- measurement ------------------------------------------------------------------------------------
    def profile(self, on=True):
        self._check(self._L.gft_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._L.gft_profile_reset(self._h))

    def profile_read(self, name):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._L.gft_profile_read(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)
