// oracle/ac_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of gofindthem's ProcessText hot path, used only as the parity checker
// (tests/, __graft_entry__.smoke()) and as bench.py's `cpu_baseline` leg.  Nothing under
// gofindthem_amd/ may include, link or call this file.
//
// What it restates (citations relative to /root/reference):
//   * CloudflareForkEngine.BuildEngine / FindSubstrings   finder/substringEngine.go:98-119
//   * Finder.ProcessText / addMatchesToSolverMap /
//     solveExpressions                                     finder/finder.go:139-215
//   * Expression.Solve / solve / getLowestIdxGTVal /
//     mergeArraysSorted                                    dsl/expression.go:60-142,175-225
//   * the Aho-Corasick matcher behind MatchAll: third-party module
//     github.com/pedroegsilva/ahocorasick v0.1.0 (go.mod:9), a fork of
//     github.com/cloudflare/ahocorasick v0.0.0-20210425175752-730270c3e184 (go.mod:8).  Its
//     source is NOT in /root/reference; the published Cloudflare design is restated here: a
//     byte trie whose nodes carry a dense child[256] and a dense precomputed fails[256]
//     array, a `suffix` link to the nearest output node on the failure chain, and a matching
//     loop that, per input byte, takes fails[c] when there is no child, steps to child[c],
//     emits that node if it is an output and then walks the suffix links emitting each.
//
// PARITY PINNING STATUS
//   * solver + grouping + result shape: pinned by the reference's own test tables
//     (dsl/expression_test.go:21-313, finder/finder_test.go:407-578) -> tests/golden/*.json.
//   * engine truth: pinned by group/finder/finder_test.go:332-447 ("string" is found).
//   * engine match POSITIONS and emission order: **parity unpinned** -- no reference test
//     asserts a position and the third-party source is absent.  Position convention is the
//     named constant ORC_POS_START (start offset = end - len + 1, like the sibling engines
//     regexEngine.go:38-42 / substringEngine.go:45-49); ORC_POS_END is selectable.  The
//     restatement is cross-checked against an independent brute-force enumerator below.
//
// Build: see oracle/Makefile (g++ -O2 -shared -fPIC).  C ABI only, driven through ctypes.

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

enum { ORC_POS_START = 0, ORC_POS_END = 1 };

// ---- Cloudflare-shaped trie node (dense arrays; indices instead of Go pointers) -------------
struct Node {
    int32_t child[256];
    int32_t fails[256];  // precomputed "first node on the fail chain that has child c", else root
    int32_t fail;
    int32_t suffix;      // nearest output node on the fail chain, root if none
    int32_t index;       // dictionary index if output
    int32_t depth;
    bool output;
};

// expression tree node, mirrors dsl.Expression (dsl/expression.go:42-48)
enum ExprType { UNSET_EXPR = 0, AND_EXPR, OR_EXPR, NOT_EXPR, UNIT_EXPR, INORD_EXPR };
struct ExprNode {
    int32_t type;
    int32_t l, r;     // child node indices, -1 == nil
    int32_t lit;      // literal index into literals, -1 if none
    int32_t inord;
};

typedef std::unordered_map<std::string, std::vector<int64_t>> SolverMap;

struct Oracle {
    std::vector<std::string> dict;  // sorted, unique (the build's deterministic term-id space)
    std::vector<Node> trie;
    int pos_mode = ORC_POS_START;
    // expressions
    std::vector<ExprNode> nodes;
    std::vector<int32_t> roots;
    std::vector<std::string> literals;
    std::string err;
};

// NewStringMatcher restated: trie insert, BFS fail links, suffix links, dense fails[] rows.
void build_trie(Oracle& o) {
    o.trie.clear();
    o.trie.reserve(1024);
    auto new_node = [&](int depth) {
        Node n;
        memset(n.child, 0xff, sizeof(n.child));
        memset(n.fails, 0, sizeof(n.fails));
        n.fail = 0; n.suffix = 0; n.index = -1; n.depth = depth; n.output = false;
        o.trie.push_back(n);
        return (int32_t)o.trie.size() - 1;
    };
    new_node(0);  // root
    for (size_t i = 0; i < o.dict.size(); i++) {
        int32_t n = 0;
        for (unsigned char c : o.dict[i]) {
            if (o.trie[n].child[c] < 0) {
                int32_t nn = new_node(o.trie[n].depth + 1);
                o.trie[n].child[c] = nn;
            }
            n = o.trie[n].child[c];
        }
        if (!o.trie[n].output) { o.trie[n].output = true; o.trie[n].index = (int32_t)i; }
    }
    // BFS
    std::vector<int32_t> queue; queue.reserve(o.trie.size());
    queue.push_back(0);
    for (size_t qi = 0; qi < queue.size(); qi++) {
        int32_t n = queue[qi];
        for (int c = 0; c < 256; c++) {
            int32_t ch = o.trie[n].child[c];
            if (ch < 0) continue;
            queue.push_back(ch);
            // fail link: longest proper suffix that is a trie node
            int32_t f = 0;
            if (n != 0) {
                int32_t t = o.trie[n].fail;
                for (;;) {
                    if (o.trie[t].child[c] >= 0) { f = o.trie[t].child[c]; break; }
                    if (t == 0) break;
                    t = o.trie[t].fail;
                }
            }
            o.trie[ch].fail = f;
            o.trie[ch].suffix = o.trie[f].output && f != 0 ? f : o.trie[f].suffix;
        }
    }
    // dense fails[]: for node n and byte c without child: first node on n's fail chain with child c
    for (size_t qi = 0; qi < queue.size(); qi++) {  // BFS order => fail(n) already final
        int32_t n = queue[qi];
        for (int c = 0; c < 256; c++) {
            if (n == 0) { o.trie[n].fails[c] = 0; continue; }
            int32_t f = o.trie[n].fail;
            o.trie[n].fails[c] = (o.trie[f].child[c] >= 0) ? f : o.trie[f].fails[c];
        }
    }
}

struct Hit { uint32_t term; int64_t pos; };

// (*Matcher).MatchAll restated; emission order = node, then its suffix chain (len desc), per end offset.
template <class F>
inline void match_all(const Oracle& o, const uint8_t* in, size_t n, F&& emit) {
    const Node* T = o.trie.data();
    int32_t cur = 0;
    for (size_t i = 0; i < n; i++) {
        int c = in[i];
        if (cur != 0 && T[cur].child[c] < 0) cur = T[cur].fails[c];
        int32_t f = T[cur].child[c];
        if (f >= 0) {
            cur = f;
            if (T[f].output) emit(T[f].index, T[f].depth, i);
            while (T[f].suffix != 0) {
                f = T[f].suffix;
                emit(T[f].index, T[f].depth, i);
            }
        }
    }
}

inline int64_t report_pos(const Oracle& o, int depth, size_t end) {
    return o.pos_mode == ORC_POS_START ? (int64_t)end - depth + 1 : (int64_t)end;
}

// ---- dsl/expression.go:175-225 ---------------------------------------------------------------
int get_lowest_idx_gt_val(const std::vector<int64_t>& positions, size_t from, int64_t value) {
    // operates on the slice positions[from:], returns index relative to that slice
    int left = 0, right = (int)(positions.size() - from) - 1, lw = -1;
    while (left <= right) {
        int half = (left + right) >> 1;
        if (positions[from + half] > value) { lw = half; right = half - 1; }
        else left = half + 1;
    }
    return lw;
}

std::vector<int64_t> merge_arrays_sorted(const std::vector<int64_t>& l, const std::vector<int64_t>& r) {
    if (l.empty()) return r;
    if (r.empty()) return l;
    std::vector<int64_t> out(l.size() + r.size());
    size_t li = 0, ri = 0, cnt = 0;
    while (cnt < out.size()) {
        if (li == l.size()) out[cnt] = r[ri++];
        else if (ri == r.size()) out[cnt] = l[li++];
        else if (l[li] < r[ri]) out[cnt] = l[li++];
        else out[cnt] = r[ri++];
        cnt++;
    }
    return out;
}

// dsl/expression.go:66-142.  Returns 0 ok / -1 error (message in err).
int solve(const Oracle& o, int32_t ni, const SolverMap& m, bool& val, std::vector<int64_t>& pos,
          std::string& err) {
    pos.clear();
    if (ni < 0) { err = "nil expression"; return -1; }
    const ExprNode& e = o.nodes[ni];
    switch (e.type) {
    case UNIT_EXPR: {
        auto it = m.find(e.lit >= 0 ? o.literals[e.lit] : std::string());
        if (it != m.end()) { val = true; pos = it->second; return 0; }
        val = false; return 0;
    }
    case AND_EXPR: {
        if (e.l < 0 || e.r < 0) { err = "AND statment do not have rigth or left expression"; return -1; }
        bool lv, rv; std::vector<int64_t> lp, rp;
        if (solve(o, e.l, m, lv, lp, err)) return -1;
        if (solve(o, e.r, m, rv, rp, err)) return -1;
        if (e.inord && !lp.empty() && !rp.empty()) {
            int idx = get_lowest_idx_gt_val(rp, 0, lp[0]);
            if (idx >= 0) pos.assign(rp.begin() + idx, rp.end());
        }
        val = lv && rv; return 0;
    }
    case OR_EXPR: {
        if (e.l < 0 || e.r < 0) { err = "OR statment do not have rigth or left expression"; return -1; }
        bool lv, rv; std::vector<int64_t> lp, rp;
        if (solve(o, e.l, m, lv, lp, err)) return -1;
        if (solve(o, e.r, m, rv, rp, err)) return -1;
        if (e.inord) pos = merge_arrays_sorted(lp, rp);
        val = lv || rv; return 0;
    }
    case NOT_EXPR: {
        if (e.r < 0) { err = "NOT statement do not have expression"; return -1; }
        bool rv; std::vector<int64_t> rp;
        if (solve(o, e.r, m, rv, rp, err)) return -1;
        val = !rv; return 0;
    }
    case INORD_EXPR: {
        if (e.r < 0) { err = "INORD statement do not have expression"; return -1; }
        bool rv; std::vector<int64_t> rp;
        if (solve(o, e.r, m, rv, rp, err)) return -1;
        val = rv && !rp.empty(); return 0;
    }
    default:
        err = "unable to process expression type " + std::to_string(e.type);
        return -1;
    }
}

inline uint8_t fold_ascii(uint8_t b) { return (b >= 'A' && b <= 'Z') ? (uint8_t)(b + 32) : b; }

// Finder.ProcessText for one document (finder/finder.go:139-179).  `extra` = regex-engine hits
// (literal index, position) appended after the keyword hits exactly like finder.go:162-176.
int process_one(const Oracle& o, const uint8_t* text, size_t n, bool fold,
                const int32_t* extra_lit, const int64_t* extra_pos, size_t n_extra,
                uint32_t* bitmap_row, std::string& err) {
    std::vector<uint8_t> lowered;
    if (fold) {  // strings.ToLower restricted to ASCII (non-ASCII folding stays with the caller)
        lowered.resize(n);
        for (size_t i = 0; i < n; i++) lowered[i] = fold_ascii(text[i]);
        text = lowered.data();
    }
    SolverMap m;
    if (!o.dict.empty()) {
        match_all(o, text, n, [&](int32_t idx, int depth, size_t end) {
            m[o.dict[idx]].push_back(report_pos(o, depth, end));   // addMatchesToSolverMap
        });
    }
    for (size_t i = 0; i < n_extra; i++) m[o.literals[extra_lit[i]]].push_back(extra_pos[i]);
    for (size_t e = 0; e < o.roots.size(); e++) {
        bool v; std::vector<int64_t> p;
        if (solve(o, o.roots[e], m, v, p, err)) return -1;
        if (v) bitmap_row[e >> 5] |= 1u << (e & 31);
    }
    return 0;
}

}  // namespace

extern "C" {

void* orc_create(const uint8_t* blob, const uint64_t* off, uint32_t n_terms, int pos_mode) {
    Oracle* o = new Oracle();
    o->pos_mode = pos_mode;
    o->dict.reserve(n_terms);
    for (uint32_t i = 0; i < n_terms; i++)
        o->dict.emplace_back((const char*)blob + off[i], (size_t)(off[i + 1] - off[i]));
    std::sort(o->dict.begin(), o->dict.end());
    o->dict.erase(std::unique(o->dict.begin(), o->dict.end()), o->dict.end());
    build_trie(*o);
    return o;
}
void orc_destroy(void* h) { delete (Oracle*)h; }
uint32_t orc_n_terms(void* h) { return (uint32_t)((Oracle*)h)->dict.size(); }
uint32_t orc_n_states(void* h) { return (uint32_t)((Oracle*)h)->trie.size(); }
const char* orc_last_error(void* h) { return ((Oracle*)h)->err.c_str(); }
// copies term i (sorted order) into out (cap bytes), returns its length
uint32_t orc_term(void* h, uint32_t i, uint8_t* out, uint32_t cap) {
    const std::string& s = ((Oracle*)h)->dict[i];
    memcpy(out, s.data(), std::min<size_t>(cap, s.size()));
    return (uint32_t)s.size();
}

// Batch FindSubstrings -> CSR.  match_off has n_docs+1 entries.  If cap is too small nothing past
// cap is written but match_off is still complete; returns total hit count.
uint64_t orc_scan_batch(void* h, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, int fold,
                        uint64_t* match_off, uint32_t* term_id, uint32_t* pos, uint64_t cap) {
    Oracle& o = *(Oracle*)h;
    uint64_t total = 0;
    std::vector<uint8_t> lowered;
    for (uint64_t d = 0; d < n_docs; d++) {
        match_off[d] = total;
        const uint8_t* t = blob + doc_off[d];
        size_t n = (size_t)(doc_off[d + 1] - doc_off[d]);
        if (fold) {
            lowered.resize(n);
            for (size_t i = 0; i < n; i++) lowered[i] = fold_ascii(t[i]);
            t = lowered.data();
        }
        match_all(o, t, n, [&](int32_t idx, int depth, size_t end) {
            if (total < cap) { term_id[total] = (uint32_t)idx; pos[total] = (uint32_t)report_pos(o, depth, end); }
            total++;
        });
    }
    match_off[n_docs] = total;
    return total;
}

// Independent ground truth: for every end offset, every term (longest first) compared with memcmp.
uint64_t orc_brute_batch(void* h, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, int fold,
                         uint64_t* match_off, uint32_t* term_id, uint32_t* pos, uint64_t cap) {
    Oracle& o = *(Oracle*)h;
    std::vector<uint32_t> by_len(o.dict.size());
    for (uint32_t i = 0; i < by_len.size(); i++) by_len[i] = i;
    std::stable_sort(by_len.begin(), by_len.end(),
                     [&](uint32_t a, uint32_t b) { return o.dict[a].size() > o.dict[b].size(); });
    uint64_t total = 0;
    std::vector<uint8_t> lowered;
    for (uint64_t d = 0; d < n_docs; d++) {
        match_off[d] = total;
        const uint8_t* t = blob + doc_off[d];
        size_t n = (size_t)(doc_off[d + 1] - doc_off[d]);
        if (fold) {
            lowered.resize(n);
            for (size_t i = 0; i < n; i++) lowered[i] = fold_ascii(t[i]);
            t = lowered.data();
        }
        for (size_t end = 0; end < n; end++) {
            for (uint32_t ti : by_len) {
                const std::string& w = o.dict[ti];
                size_t L = w.size();
                if (L == 0 || L > end + 1) continue;
                if (memcmp(t + end + 1 - L, w.data(), L) == 0) {
                    if (total < cap) {
                        term_id[total] = ti;
                        pos[total] = (uint32_t)(o.pos_mode == ORC_POS_START ? end + 1 - L : end);
                    }
                    total++;
                }
            }
        }
    }
    match_off[n_docs] = total;
    return total;
}

// Expression forest: node table (type,l,r,lit,inord) x n_nodes, roots x n_exprs, literals blob.
int orc_set_expressions(void* h, const int32_t* node_tab, uint32_t n_nodes, const int32_t* roots, uint32_t n_exprs,
                        const uint8_t* lit_blob, const uint64_t* lit_off, uint32_t n_lits) {
    Oracle& o = *(Oracle*)h;
    o.nodes.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        o.nodes[i].type = node_tab[i * 5 + 0]; o.nodes[i].l = node_tab[i * 5 + 1];
        o.nodes[i].r = node_tab[i * 5 + 2];    o.nodes[i].lit = node_tab[i * 5 + 3];
        o.nodes[i].inord = node_tab[i * 5 + 4];
    }
    o.roots.assign(roots, roots + n_exprs);
    o.literals.clear();
    for (uint32_t i = 0; i < n_lits; i++)
        o.literals.emplace_back((const char*)lit_blob + lit_off[i], (size_t)(lit_off[i + 1] - lit_off[i]));
    return 0;
}

// Expression.Solve on an explicit map (fixture tests).  Map = n_keys keys (blob/off) each with a
// position list key_pos[key_pos_off[k] .. key_pos_off[k+1]).  Returns 1 true / 0 false / -1 error.
int orc_solve(void* h, uint32_t expr, const uint8_t* key_blob, const uint64_t* key_off, uint32_t n_keys,
              const int64_t* key_pos, const uint64_t* key_pos_off) {
    Oracle& o = *(Oracle*)h;
    SolverMap m;
    for (uint32_t k = 0; k < n_keys; k++) {
        std::string key((const char*)key_blob + key_off[k], (size_t)(key_off[k + 1] - key_off[k]));
        m[key].assign(key_pos + key_pos_off[k], key_pos + key_pos_off[k + 1]);
    }
    bool v; std::vector<int64_t> p;
    if (solve(o, o.roots[expr], m, v, p, o.err)) return -1;
    return v ? 1 : 0;
}

// Batch ProcessText -> hit bitmap [n_docs x ceil(E/32)] (must be zeroed by the caller).
// extra_* (nullable) = per-document regex hits in CSR form (literal index, position).
int orc_process_batch(void* h, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, int fold,
                      const uint64_t* extra_off, const int32_t* extra_lit, const int64_t* extra_pos,
                      uint32_t* bitmap, int n_threads) {
    Oracle& o = *(Oracle*)h;
    size_t words = (o.roots.size() + 31) / 32;
    if (n_threads < 1) n_threads = 1;
    std::vector<std::string> errs(n_threads);
    std::vector<int> rc(n_threads, 0);
    auto work = [&](int t) {
        uint64_t lo = n_docs * t / n_threads, hi = n_docs * (t + 1) / n_threads;
        for (uint64_t d = lo; d < hi; d++) {
            size_t ne = extra_off ? (size_t)(extra_off[d + 1] - extra_off[d]) : 0;
            if (process_one(o, blob + doc_off[d], (size_t)(doc_off[d + 1] - doc_off[d]), fold != 0,
                            extra_off ? extra_lit + extra_off[d] : nullptr,
                            extra_off ? extra_pos + extra_off[d] : nullptr, ne,
                            bitmap + d * words, errs[t])) { rc[t] = -1; return; }
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (int t = 0; t < n_threads; t++) if (rc[t]) { o.err = errs[t]; return -1; }
    return 0;
}

// Scan-only batch over threads, returns total hits (CPU baseline for the positions-only config).
uint64_t orc_scan_count_batch(void* h, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, int n_threads) {
    Oracle& o = *(Oracle*)h;
    if (n_threads < 1) n_threads = 1;
    std::vector<uint64_t> tot(n_threads, 0);
    auto work = [&](int t) {
        uint64_t lo = n_docs * t / n_threads, hi = n_docs * (t + 1) / n_threads, c = 0, x = 0;
        for (uint64_t d = lo; d < hi; d++)
            match_all(o, blob + doc_off[d], (size_t)(doc_off[d + 1] - doc_off[d]),
                      [&](int32_t idx, int depth, size_t end) { c++; x += idx + report_pos(o, depth, end); });
        tot[t] = c + (x == 0xdeadbeefcafef00dull);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
    for (auto& x : th) x.join();
    uint64_t s = 0; for (auto v : tot) s += v;
    return s;
}

}  // extern "C"

// ---- analysis helper (design studies; not used by tests' parity checks) -----------------------------------------
// histogram of automaton depth after each input byte, and of emitted matches by term length
extern "C" void orc_depth_hist(void* h, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs,
                               uint64_t* depth_hist /*[64]*/, uint64_t* emit_len_hist /*[64]*/) {
    Oracle& o = *(Oracle*)h;
    const Node* T = o.trie.data();
    for (uint64_t d = 0; d < n_docs; d++) {
        const uint8_t* in = blob + doc_off[d];
        size_t n = (size_t)(doc_off[d + 1] - doc_off[d]);
        int32_t cur = 0;
        for (size_t i = 0; i < n; i++) {
            int c = in[i];
            if (cur != 0 && T[cur].child[c] < 0) cur = T[cur].fails[c];
            int32_t f = T[cur].child[c];
            if (f >= 0) {
                cur = f;
                if (T[f].output) emit_len_hist[std::min(T[f].depth, 63)]++;
                while (T[f].suffix != 0) { f = T[f].suffix; emit_len_hist[std::min(T[f].depth, 63)]++; }
            }
            depth_hist[std::min(T[cur].depth, 63)]++;
        }
    }
}
