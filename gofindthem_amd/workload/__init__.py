"""Synthetic workload of SURVEY.md 8(d) (vocabulary, dictionary, documents, expressions).

Stands in for the reference's benchmark corpus (benchmarks/benchmark_test.go:22-67,438-489; its word list
files/words.txt is missing from the mount).  Bench/test utility: deterministic, integer-only, identical on the
host (corpus_host.cpp) and on the GPU (corpus_gen.hip).  Not part of the drop-in library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgfworkload.so")
_LIB = None

GOLDEN = 0x9E3779B97F4A7C15
BASE_SEED = GOLDEN ^ 1629074756677820700
VOCAB_WORDS = 466550
M64 = (1 << 64) - 1


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("corpus_gen.hip", "corpus_host.cpp", "corpus_rng.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    ho, go = os.path.join(_HERE, "corpus_host.o"), os.path.join(_HERE, "corpus_gen.o")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-c", srcs[1], "-o", ho])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-c", srcs[0], "-o", go])
    subprocess.check_call(["hipcc", "-shared", "-o", _SO, go, ho])
    return _SO


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.gfw_vocab_build.restype = C.c_uint32
        L.gfw_vocab_build.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(vp), C.POINTER(vp)]
        L.gfw_sample_dictionary.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, vp]
        L.gfw_doc_lengths_host.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint32, vp, C.c_uint32, vp]
        L.gfw_doc_fill_host.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp]
        L.gfw_doc_lengths_dev.restype = C.c_int
        L.gfw_doc_lengths_dev.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint32, vp, C.c_uint32, vp, vp]
        L.gfw_read_sum_dev.restype = C.c_int
        L.gfw_read_sum_dev.argtypes = [vp, C.c_uint64, vp, vp]
        L.gfw_doc_fill_dev.restype = C.c_int
        L.gfw_doc_fill_dev.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mix(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def stream(seed, i):
    return mix(seed + GOLDEN * (i + 1))


_LEET = {ord("a"): b"4", ord("e"): b"3", ord("i"): b"1", ord("o"): b"0", ord("s"): b"5", ord("t"): b"7", ord("b"): b"8",
         ord("g"): b"9", ord("z"): b"2", ord("l"): b"6"}
_UTF8 = {ord("a"): "ä", ord("e"): "é", ord("o"): "ö", ord("u"): "ü", ord("n"): "ñ", ord("c"): "ç", ord("s"): "ß", ord("i"): "í"}
_PUNCT = b"-_.'@#&+"


def _dress_vocabulary(blob, off, seed):
    """every word, deterministically by its index: maybe a capital / all capitals, a digit for a letter, a two-byte
    UTF-8 letter (lower-case forms only, so that ASCII case folding equals strings.ToLower on this corpus), a
    punctuation mark inside or behind it.  Returns (blob uint8[], off uint64[n+1])."""
    raw = blob.tobytes()
    out, offs = bytearray(), [0]
    key = mix(seed ^ 0x5EED)
    for i in range(len(off) - 1):
        w = bytearray(raw[int(off[i]):int(off[i + 1])])
        h = stream(key, i)
        r = [(h >> (8 * k)) & 0xFF for k in range(8)]
        if r[0] < 26:                                   # ~10 %: Capitalised
            w[0] = w[0] - 32
        elif r[0] < 34:                                 # ~3 %: SHOUTED
            w = bytearray(bytes(w).upper())
        if r[1] < 20:                                   # ~8 %: a digit for a letter
            k = r[2] % len(w)
            if w[k] in _LEET:
                w[k:k + 1] = _LEET[w[k]]
        if r[3] < 20:                                   # ~8 %: a two-byte letter
            k = r[4] % len(w)
            if w[k] in _UTF8:
                w[k:k + 1] = _UTF8[w[k]].encode("utf-8")
        if r[5] < 15:                                   # ~6 %: punctuation behind
            w.append(_PUNCT[r[6] % len(_PUNCT)])
        elif r[5] < 23 and len(w) >= 4:                 # ~3 %: punctuation inside
            k = 1 + r[6] % (len(w) - 2)
            if w[k] < 0x80 and w[k - 1] < 0xC0:
                w[k:k] = bytes([_PUNCT[r[7] % 3]])
        out += w
        offs.append(len(out))
    return np.frombuffer(bytes(out), dtype=np.uint8).copy(), np.asarray(offs, dtype=np.uint64)


class Workload:
    """vocabulary + an n_terms dictionary; generates documents on the host or straight into HBM."""

    def __init__(self, n_terms, base_seed=BASE_SEED, n_vocab=VOCAB_WORDS, alphabet="lower"):
        """alphabet = "lower": SURVEY.md 8(d), words over a-z.  "mixed": the same words dressed the way a real word list
        looks (benchmarks/benchmark_test.go:43,72-83 reads one): capitals, digits, punctuation, two-byte UTF-8 letters --
        more than 48 distinct bytes even after ASCII case folding, terms of 2-3 bytes included."""
        L = _lib()
        self.base_seed = base_seed
        self.alphabet = alphabet
        blob, off = C.c_void_p(), C.c_void_p()
        self.n_vocab = L.gfw_vocab_build(base_seed, n_vocab, C.byref(blob), C.byref(off))
        self.vocab_off = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(self.n_vocab + 1,))
        self.vocab_blob = np.ctypeslib.as_array(C.cast(blob, C.POINTER(C.c_uint8)), shape=(int(self.vocab_off[-1]),))
        if alphabet == "mixed":
            self.vocab_blob, self.vocab_off = _dress_vocabulary(self.vocab_blob, self.vocab_off, base_seed)
        elif alphabet != "lower":
            raise ValueError("alphabet must be 'lower' or 'mixed'")
        self.n_terms = n_terms
        self.dict_idx = np.zeros(max(n_terms, 1), dtype=np.uint32)
        if n_terms:
            L.gfw_sample_dictionary(base_seed, self.n_vocab, n_terms, _p(self.dict_idx))
        self.dict_idx = self.dict_idx[:n_terms]
        self._dev = None

    def word(self, i):
        return bytes(self.vocab_blob[int(self.vocab_off[i]):int(self.vocab_off[i + 1])])

    def terms(self):
        return [self.word(int(i)) for i in self.dict_idx]

    def docs_host(self, first, n):
        """-> (text uint8[N], doc_off uint64[n+1]) numpy"""
        L = _lib()
        lens = np.zeros(max(n, 1), dtype=np.uint32)
        di = self.dict_idx if self.n_terms else np.zeros(1, np.uint32)
        L.gfw_doc_lengths_host(self.base_seed, first, n, _p(self.vocab_off), self.n_vocab, _p(di), self.n_terms, _p(lens))
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens[:n], dtype=np.uint64)
        text = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        L.gfw_doc_fill_host(self.base_seed, first, n, _p(self.vocab_blob), _p(self.vocab_off), self.n_vocab, _p(di),
                            self.n_terms, _p(off), _p(text))
        return text[:int(off[-1])], off

    def docs_device(self, first, n, device="cuda"):
        """-> (text uint8 tensor, doc_off int64 tensor[n+1]) resident in HBM (torch = device-memory plumbing)."""
        import torch
        L = _lib()
        if self._dev is None:
            self._dev = (torch.from_numpy(self.vocab_blob.copy()).to(device),
                         torch.from_numpy(self.vocab_off.astype(np.int64)).to(device),
                         torch.from_numpy(self.dict_idx.astype(np.int32) if self.n_terms else np.zeros(1, np.int32)).to(device))
        vb, vo, di = self._dev
        st = torch.cuda.current_stream().cuda_stream
        lens = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        rc = L.gfw_doc_lengths_dev(self.base_seed, first, n, vo.data_ptr(), self.n_vocab, di.data_ptr(), self.n_terms,
                                   lens.data_ptr(), st)
        assert rc == 0, rc
        off = torch.zeros(n + 1, dtype=torch.int64, device=device)
        off[1:] = torch.cumsum(lens[:n].to(torch.int64), 0)
        total = int(off[-1].item())
        text = torch.empty(max(total, 1) + 64, dtype=torch.uint8, device=device)  # +64: slack for vector loads
        text[total:] = 0
        rc = L.gfw_doc_fill_dev(self.base_seed, first, n, vb.data_ptr(), vo.data_ptr(), self.n_vocab, di.data_ptr(),
                                self.n_terms, off.data_ptr(), text.data_ptr(), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return text[:total], off


def read_ceiling_gbps(buf, reps=5):
    """achievable HBM read bandwidth on this device: a coalesced sum over `buf` (a torch uint8 CUDA tensor), GB/s"""
    import torch
    L = _lib()
    scratch = torch.zeros(1, dtype=torch.int64, device=buf.device)
    st = torch.cuda.current_stream().cuda_stream
    nbytes = int(buf.numel()) & ~15
    L.gfw_read_sum_dev(buf.data_ptr(), nbytes, scratch.data_ptr(), st)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        rc = L.gfw_read_sum_dev(buf.data_ptr(), nbytes, scratch.data_ptr(), st)
        assert rc == 0
    ev1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


def make_expressions(terms, n_exprs, inord_fraction=0.0, seed=BASE_SEED + 3, regexes=(), cover=False):
    """Expression strings over `terms` (list of bytes), SURVEY.md 8(d).

    cover=False: each expression has 1 + r%10 leaves drawn uniformly from `terms` (benchmarks/benchmark_test.go:467).
    cover=True : the finder's dictionary is exactly the keyword set of its expressions (finder/finder.go:123-126), so
                 for the automaton to hold ALL `terms` ("10 k-term dictionary + 1 k expressions") the leaves walk a
                 shuffled cycle of `terms` and leaf counts are 1 + r%(2*len(terms)/n_exprs - 1), topped up until every
                 term is referenced at least once.
    A fraction `inord_fraction` is INORD(t1 AND ... AND tk) exactly like createRandExpressionAndSolverMap (:449-460);
    the rest combine leaves with random and/or/not and optional parentheses.  `regexes` (source strings) are mixed in
    as r"..." leaves when given."""
    key = mix(seed)
    ctr = [0]

    def rnd(n):
        ctr[0] += 1
        return stream(key, ctr[0]) % n

    def lit(t):
        s = t.decode("utf-8").replace("\\", "\\\\").replace('"', '\\"')
        return '"%s"' % s

    if cover:
        span = max(2 * len(terms) // max(n_exprs, 1) - 1, 1)
        counts = [1 + rnd(span) for _ in range(n_exprs)]
        while sum(counts) < len(terms):
            counts[rnd(n_exprs)] += 1
        perm = list(range(len(terms)))
        for i in range(len(perm) - 1, 0, -1):
            j = rnd(i + 1)
            perm[i], perm[j] = perm[j], perm[i]
        cur = [0]

        def next_term():
            t = terms[perm[cur[0] % len(perm)]]
            cur[0] += 1
            return t
    else:
        counts = [1 + rnd(10) for _ in range(n_exprs)]

        def next_term():
            return terms[rnd(len(terms))]

    def leaf():
        if regexes and rnd(8) == 0:
            return 'r"%s"' % regexes[rnd(len(regexes))]
        return lit(next_term())

    out = []
    for k in counts:
        if inord_fraction > 0 and rnd(1000) < int(inord_fraction * 1000):
            out.append("INORD(" + " AND ".join(lit(next_term()) for _ in range(k)) + ")")
            continue
        parts, depth = [], 0
        for i in range(k):
            if i:
                parts.append(" and " if rnd(2) else " or ")
            if rnd(4) == 0:
                parts.append("not ")
                if rnd(3) == 0 and i + 1 < k:
                    parts.append("(")
                    depth += 1
                    parts.append(leaf())
                    continue
            elif rnd(5) == 0 and i + 1 < k:
                parts.append("(")
                depth += 1
            parts.append(leaf())
            if depth and rnd(2):
                parts.append(")")
                depth -= 1
        parts.append(")" * depth)
        out.append("".join(parts))
    return out
