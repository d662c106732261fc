#!/usr/bin/env python3
"""Design study (CPU, numpy): flagged probes per document for candidate stride-2 filter layouts on the benchmark workload.
Not product code."""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gofindthem_amd.workload import Workload

n_terms = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = Workload(n_terms)
terms = sorted(set(w.terms()))
text, off = w.docs_host(0, n_docs)
cls = np.zeros(256, np.int64)
used = sorted(set(b for t in terms for b in t))
for i, b in enumerate(used): cls[b] = i + 1
K = len(used) + 1
print("K", K, "terms", len(terms), "bytes/doc", len(text) / n_docs)
c = cls[text]          # classes of the whole blob (docs concatenated: edge effects negligible)
N = len(c)
# unigram class frequencies from the text for the anchor cost model
freq = np.bincount(c, minlength=K) / N

def cl(t): return [int(cls[b]) for b in t]

def anchors_for(term):
    """two anchors (off even, off odd) -> list of (off, window classes(4, None=wildcard), front classes [c(i-3-1)=c[i-4]...])"""
    L = len(term); tc = cl(term)
    res = []
    for par in (0, 1):
        best = None
        for o in range(par, 5, 2):          # window ends o bytes before the term end
            if L - o < 4: continue
            win = tc[L - o - 4: L - o]
            nf = L - o - 4
            cost = np.prod([freq[x] for x in win]) * (freq[tc[L - o - 5]] if nf >= 1 else 1.0) * (freq[tc[L - o - 6]] if nf >= 2 else 1.0)
            if best is None or cost < best[0]: best = (cost, o)
        res.append(best[1] if best else None)
    return res

def build(variant, Mf=11):
    """returns dict: filter array + meta. variant 'F4' (K^4 bitmap), 'F5' (x3 word, front-pair-hash bit)"""
    if variant == 'F4':
        F = np.zeros(K ** 4, bool)
    else:
        F = np.zeros((K ** 3, 32), bool)
    allc = np.arange(K)
    def set4(win, front):   # win: list of 4 entries (int or None), front: (c[i-4], c[i-3]=win[0]) handled separately
        idx = [allc if x is None else np.array([x]) for x in win]
        g = np.stack(np.meshgrid(*idx, indexing='ij'), -1).reshape(-1, 4)
        if variant == 'F4':
            F[((g[:, 0] * K + g[:, 1]) * K + g[:, 2]) * K + g[:, 3]] = True
        else:
            x3 = (g[:, 1] * K + g[:, 2]) * K + g[:, 3]
            f4 = allc if front is None else np.array([front])
            if variant == 'F4p':
                F[x3, g[:, 0] % 27] = True
                for a in f4: F[x3, 27 + ((a * 6) >> 5) % 5] = True
            else:
                for a in f4:
                    F[x3, (a * Mf + g[:, 0]) & 31] = True
    for t in terms:
        L = len(t); tc = cl(t)
        if L >= 5:
            for o in anchors_for(t):
                win = tc[L - o - 4: L - o]
                front = tc[L - o - 5] if L - o - 5 >= 0 else None
                set4(win, front)
        elif L == 4:
            set4(tc, None)                              # ends at the probe
            set4(tc[1:] + [None], tc[0])                # ends one before the probe: (b,c,d,*), front pair (a, b)... c[i-4]=a, c[i-3]=b
        elif L == 3:
            set4([None] + tc, None)
            set4(tc + [None], None)
        elif L == 2:
            set4([None, None] + tc, None)
            set4([None] + tc + [None], None)
    return F

def flagged(variant, F, par, Mf=11):
    i = np.arange(4 + par, N, 2)
    c0, c1, c2, c3, cf = c[i - 3], c[i - 2], c[i - 1], c[i], c[i - 4]
    if variant == 'F4':
        fl = F[((c0 * K + c1) * K + c2) * K + c3]
    else:
        if variant == 'F4p':
            fl = F[(c1 * K + c2) * K + c3, c0 % 27] & F[(c1 * K + c2) * K + c3, 27 + ((cf * 6) >> 5) % 5]
        else:
            fl = F[(c1 * K + c2) * K + c3, (cf * Mf + c0) & 31]
    return i[fl]

# ground truth: end positions of short (<=3) matches and of long matches
short_end = np.zeros(N, bool); long_end = np.zeros(N, bool)
byL = collections.defaultdict(set)
for t in terms: byL[len(t)].add(t)
tb = text.tobytes()
for L, st in byL.items():
    for s in range(0, N - L + 1):
        pass
# faster: use python find per term
nshort = nlong = 0
for t in terms:
    s = tb.find(t)
    while s >= 0:
        if len(t) <= 3: short_end[s + len(t) - 1] = True; nshort += 1
        else: long_end[s + len(t) - 1] = True; nlong += 1
        s = tb.find(t, s + 1)
print("matches/doc short %.1f long %.1f" % (nshort / n_docs, nlong / n_docs))
for variant in ('F4', 'F4p'):
    F = build(variant)
    dens = F.mean()
    for par in (0, 1):
        fl = flagged(variant, F, par)
        sh = short_end[fl] | short_end[fl - 1]
        print("%s parity %d: density %.4f flagged/doc %.1f  of which short sites %.1f, other %.1f" %
              (variant, par, dens, len(fl) / n_docs, sh.sum() / n_docs, (~sh).sum() / n_docs))

# ---- attribution for F4: which entry types flag how many probes ----
print("attribution (F4, parity 0)")
types = collections.OrderedDict()
def add(name, win):
    types.setdefault(name, []).append(win)
for t in terms:
    L = len(t); tc = cl(t)
    if L >= 5:
        for o in anchors_for(t):
            add("L%d%s" % (min(L, 7), "+" if L >= 7 else ""), tc[L - o - 4: L - o])
    elif L == 4:
        add("L4A", tc); add("L4B", tc[1:] + [None])
    elif L == 3:
        add("L3A", [None] + tc); add("L3B", tc + [None])
    else:
        add("L2A", [None, None] + tc); add("L2B", [None] + tc + [None])
i = np.arange(4, N, 2)
key = ((c[i - 3] * K + c[i - 2]) * K + c[i - 1]) * K + c[i]
allc = np.arange(K)
for name, wins in types.items():
    F = np.zeros(K ** 4, bool)
    for win in wins:
        idx = [allc if x is None else np.array([x]) for x in win]
        g = np.stack(np.meshgrid(*idx, indexing='ij'), -1).reshape(-1, 4)
        F[((g[:, 0] * K + g[:, 1]) * K + g[:, 2]) * K + g[:, 3]] = True
    print("  %-5s entries %6d keys %7d flagged/doc %.1f" % (name, len(wins), F.sum(), F[key].sum() / n_docs))

# ---- anchor choice study: any offset 0..L-4, by model ----
print("anchor choice study (F4, L>=5 terms only, parity 0 probes)")
# dictionary n-gram model: 4-gram counts over the dictionary's own terms (+ smoothing by unigram product)
dict4 = collections.Counter()
for t in terms:
    tc = cl(t)
    for s in range(len(tc) - 3): dict4[tuple(tc[s:s + 4])] += 1
text4 = np.bincount(((c[3:] * 1 + c[2:-1] * K + c[1:-2] * K * K + c[:-3] * K ** 3)), minlength=K ** 4)  # key of window ending at i
def k4(win): return ((win[0] * K + win[1]) * K + win[2]) * K + win[3]
dfreq = np.zeros(K)
for t in terms:
    for x in cl(t): dfreq[x] += 1
dfreq /= dfreq.sum()
for model in ("uni_text_maxoff4", "uni_dict_any", "dict4_any", "text4_any(oracle)", "uni_dict_any_uniq"):
    F = np.zeros(K ** 4, bool)
    taken = set()
    offs = collections.Counter()
    for t in terms:
        L = len(t); tc = cl(t)
        if L < 5: continue
        for par in (0, 1):
            best = None
            rng = range(par, 5, 2) if model.endswith("maxoff4") else range(par, L - 3, 2)
            for o in rng:
                if L - o < 4: continue
                win = tc[L - o - 4: L - o]
                if model.startswith("uni_text"): cost = np.prod([freq[x] for x in win])
                elif model.startswith("uni_dict"): cost = np.prod([dfreq[x] for x in win])
                elif model.startswith("dict4"): cost = dict4[tuple(win)] + 1e3 * np.prod([dfreq[x] for x in win])
                else: cost = text4[k4(win)]
                if model.endswith("uniq") and k4(win) in taken: cost *= 0.3
                if best is None or cost < best[0]: best = (cost, o, win)
            F[k4(best[2])] = True; taken.add(k4(best[2])); offs[best[1]] += 1
    print("  %-22s keys %6d flagged/doc %.1f  offs %s" % (model, F.sum(), F[key].sum() / n_docs, sorted(offs.items())[:8]))
