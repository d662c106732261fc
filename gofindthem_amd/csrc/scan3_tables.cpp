#include "scan3_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <unordered_map>

namespace gft {

namespace {

constexpr uint32_t kWin3 = 4;
constexpr int kMaxOff3 = (int)kScan2MaxOff;

struct Ent { uint32_t term_id, len1; int off; };   // len1 = length up to the end of the window = L - off

// a bucket-table slot for the anchor (len1, off) of term s.  The 20 bytes behind the header: front[k] = the term's bytes
// under text[p-7-4k .. p-4-4k] for k = 0..2 (3: as well, unless off > 0: then front[3] = the bytes behind the window,
// text[p+1 ..]), and front[4] = the window's own bytes text[p-3 .. p] -- under merged filter groups the key does not prove
// them.  A term up to 20 bytes (16 when shifted) is compared without leaving the slot.
Scan2Slot make_slot(uint32_t key, const Ent& e, const std::string& s) {
    const int L = (int)e.len1;
    Scan2Slot r{key, e.term_id, e.len1 | (uint32_t)(uint8_t)(int8_t)e.off << 24, {0, 0, 0, 0, 0}};
    for (int k = 0; k < (e.off > 0 ? 3 : 4); k++)
        for (int b = 0; b < 4; b++) {
            const int idx = L - 8 - 4 * k + b;             // term byte under text[p-7-4k+b]
            if (idx >= 0) r.front[k] |= (uint32_t)(uint8_t)s[(size_t)idx] << (8 * b);
        }
    for (int b = 0; b < e.off; b++) r.front[3] |= (uint32_t)(uint8_t)s[(size_t)(L + b)] << (8 * b);   // text[p+1+b]
    for (int b = 0; b < 4; b++) {
        const int idx = L - 4 + b;                         // term byte under text[p-3+b]; past the term's end (off = -1): free
        if (idx >= 0 && idx < (int)s.size()) r.front[4] |= (uint32_t)(uint8_t)s[(size_t)idx] << (8 * b);
    }
    return r;
}

}  // namespace

void build_scan3_tables(const AcTables& ac, Scan3Tables& t) {
    t = Scan3Tables();
    const uint32_t kp = ac.n_classes;
    if (kp == 0 || kp > 256) { t.why_not = "bad class count"; return; }

    // ---- filter groups ------------------------------------------------------------------------------------------------
    std::vector<uint32_t> group_of(kp, 0);
    std::vector<uint64_t> cnt(kp, 0);
    for (const auto& s : ac.terms) for (unsigned char ch : s) cnt[ac.byte_class[ch]]++;
    if (kp <= kScan3Groups) {
        t.G = kp;
        for (uint32_t c = 0; c < kp; c++) group_of[c] = c;
    } else {
        // class 0 (bytes of no term) keeps group 0; the others go, most frequent first, to the lightest of the remaining
        // groups (longest-processing-time rule): the frequent bytes end up alone, the rare ones share
        t.G = kScan3Groups;
        t.grouped = true;
        std::vector<uint32_t> order;
        for (uint32_t c = 1; c < kp; c++) order.push_back(c);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });
        std::vector<uint64_t> mass(t.G, 0);
        for (uint32_t c : order) {
            uint32_t best = 1;
            for (uint32_t g = 2; g < t.G; g++) if (mass[g] < mass[best]) best = g;
            group_of[c] = best;
            mass[best] += cnt[c] + 1;
        }
    }
    const uint32_t G = t.G;
    for (int b = 0; b < 256; b++) t.cls[b] = (uint8_t)group_of[ac.byte_class[b]];
    for (int b = 0; b < 256; b++) t.cls_fold[b] = t.cls[(b >= 'A' && b <= 'Z') ? b + 32 : b];
    auto grp = [&](char ch) { return (uint32_t)t.cls[(uint8_t)ch]; };

    const uint64_t G2 = (uint64_t)G * G, G3 = G2 * G, G4 = G3 * G;
    t.filter.assign((size_t)((G4 + 31) / 32), 0);
    auto set_filter = [&](uint64_t x) { t.filter[(size_t)(x >> 5)] |= 1u << (x & 31); };

    // unigram model of the text for the choice of anchors: group frequencies of the dictionary itself
    std::vector<double> logp(G, 0.0);
    {
        std::vector<uint64_t> gc(G, 0);
        uint64_t tot = 0;
        for (uint32_t c = 0; c < kp; c++) { gc[group_of[c]] += cnt[c]; tot += cnt[c]; }
        for (uint32_t g = 0; g < G; g++) logp[g] = std::log(((double)gc[g] + 0.5) / ((double)tot + 0.5 * G));
    }
    auto window_key = [&](const std::string& s, uint32_t len1) {
        uint32_t key = 0;
        for (uint32_t i = len1 - kWin3; i < len1; i++) key = key * G + grp(s[i]);
        return key;
    };

    // ---- term blob (4 bytes of slack in front of every term: the kernel compares unaligned dwords) -------------------
    for (size_t id = 0; id < ac.terms.size(); id++) {
        t.term_blob.insert(t.term_blob.end(), 4, 0);
        t.term_off.push_back((uint32_t)t.term_blob.size());
        t.term_blob.insert(t.term_blob.end(), ac.terms[id].begin(), ac.terms[id].end());
    }
    t.term_blob.insert(t.term_blob.end(), 8, 0);
    t.term_off.push_back((uint32_t)t.term_blob.size());

    // ---- anchors of the terms of length >= 4, short terms by the 3-group window they end -------------------------------
    std::unordered_map<uint32_t, std::vector<Ent>> buckets;
    buckets.reserve(ac.terms.size() * 3);
    std::unordered_map<uint32_t, uint32_t> taken;             // window key -> anchors there so far
    std::unordered_map<uint32_t, std::vector<uint32_t>> cells;   // 3-group window -> short terms ending there
    std::vector<std::pair<uint32_t, uint32_t>> bloom_items;   // (window key, front group or ~0u = any)
    bool any_short = false;
    // (4-byte terms first: their odd anchors take G window keys each, which the free anchors of the longer terms then avoid)
    std::vector<size_t> term_order;
    for (size_t id = 0; id < ac.terms.size(); id++) if (ac.terms[id].size() == kWin3) term_order.push_back(id);
    for (size_t id = 0; id < ac.terms.size(); id++) if (ac.terms[id].size() != kWin3) term_order.push_back(id);
    for (size_t id : term_order) {
        const std::string& s = ac.terms[id];
        const uint32_t L = (uint32_t)s.size();
        if (L == 0) continue;                                  // the empty keyword never matches
        if (L > kScan2LenMask) { t.why_not = "term longer than 16 MiB"; return; }
        if (L < kWin3) {
            if (id >= (1u << 28)) { t.why_not = "term id too large for a short-term record"; return; }
            any_short = true;
            uint32_t tail = 0, scale = 1, combos = 1;
            for (uint32_t i = 0; i < L; i++) { tail = tail * G + grp(s[i]); scale *= G; }
            for (uint32_t i = L; i < 3; i++) combos *= G;
            for (uint32_t pre = 0; pre < combos; pre++) cells[pre * scale + tail].push_back((uint32_t)id);
            continue;
        }
        for (int par = 0; par < 2; par++) {
            if (L == kWin3 && par == 1) {
                // the odd anchor of a 4-byte term: the window one byte further on, its last group is free
                uint32_t k3 = 0;
                for (uint32_t i = 1; i < 4; i++) k3 = k3 * G + grp(s[i]);
                for (uint32_t g = 0; g < G; g++) {
                    const uint32_t key = k3 * G + g;
                    buckets[key].push_back(Ent{(uint32_t)id, 5, -1});
                    taken[key]++;
                    bloom_items.push_back({key, grp(s[0])});
                    t.n_anchors++;
                }
                continue;
            }
            // cost of an anchor under the unigram model: every text position with these four groups costs a stage-A
            // check, and one whose front byte's group fits as well costs a bucket-table probe
            int best = -1;
            double best_eff = 0;
            for (int off = par; off <= kMaxOff3 && (uint32_t)off + kWin3 <= L; off += 2) {
                const uint32_t len1 = L - (uint32_t)off;
                double w = 0;
                for (uint32_t i = len1 - kWin3; i < len1; i++) w += logp[grp(s[i])];
                const double pf = len1 > kWin3 ? std::exp(logp[grp(s[len1 - kWin3 - 1])]) : 1.0;
                const double cost = std::exp(w) * (1.0 + 4.0 * pf);
                auto it = taken.find(window_key(s, len1));
                const double eff = cost * (it == taken.end() ? 1.0 : 3.0) * (off == par ? 0.999 : 1.0);
                if (best < 0 || eff < best_eff) { best = off; best_eff = eff; }
            }
            if (best < 0) continue;   // (cannot happen: off = par fits every term of length >= 5, and par = 0 one of 4)
            const uint32_t len1 = L - (uint32_t)best;
            const uint32_t key = window_key(s, len1);
            taken[key]++;
            buckets[key].push_back(Ent{(uint32_t)id, len1, best});
            bloom_items.push_back({key, len1 > kWin3 ? grp(s[len1 - kWin3 - 1]) : ~0u});
            t.n_anchors++;
        }
    }
    t.n_keys = buckets.size();

    // ---- short3 + records ------------------------------------------------------------------------------------------------
    t.srec.assign(kScan3RecWords, 0);          // record 0: empty
    if (any_short) {
        t.short3.assign(((size_t)G3 + 15) & ~(size_t)15, 0);
        auto entry_words = [&](uint32_t id, uint32_t* w) {
            const std::string& s = ac.terms[id];
            const uint32_t L = (uint32_t)s.size();
            w[0] = id | L << 28;
            w[1] = 0;
            for (uint32_t k = 0; k < L; k++) w[1] |= (uint32_t)(uint8_t)s[k] << (8 * (4 - L + k));   // as text[e-3 .. e]
        };
        // cells in descending order of expected text frequency get the LDS record ids
        std::vector<std::pair<double, uint32_t>> by_freq;
        for (auto& kv : cells) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) {
                return ac.terms[a].size() != ac.terms[b].size() ? ac.terms[a].size() > ac.terms[b].size() : a < b;
            });
            const uint32_t x3 = kv.first;
            by_freq.push_back({-(logp[x3 / G2] + logp[(x3 / G) % G] + logp[x3 % G]), x3});
        }
        std::sort(by_freq.begin(), by_freq.end());
        std::map<std::vector<uint32_t>, uint32_t> ids, big_ids;
        for (const auto& bf : by_freq) {
            const uint32_t x3 = bf.second;
            const auto& v = cells[x3];
            uint32_t id = 255;
            if (v.size() <= 3) {
                auto it = ids.find(v);
                if (it != ids.end()) id = it->second;
                else if (ids.size() < kScan3RecLds) {
                    id = (uint32_t)ids.size() + 1;
                    ids.emplace(v, id);
                    uint32_t w[kScan3RecWords] = {0, 0, 0, 0, 0, 0};
                    for (size_t i = 0; i < v.size(); i++) entry_words(v[i], w + 2 * i);
                    t.srec.insert(t.srec.end(), w, w + kScan3RecWords);
                }
            }
            t.short3[x3] = (uint8_t)id;
            if (id == 255) {
                if (t.short3_big.empty()) t.short3_big.assign(t.short3.size(), 0);
                auto it = big_ids.find(v);
                if (it == big_ids.end()) {
                    it = big_ids.emplace(v, (uint32_t)t.srec_big.size()).first;
                    t.srec_big.push_back((uint32_t)v.size());
                    for (uint32_t term : v) { uint32_t w[2]; entry_words(term, w); t.srec_big.push_back(w[0]); t.srec_big.push_back(w[1]); }
                }
                t.short3_big[x3] = it->second;
            }
            // a probe at p answers for ends at p (any group in front of the 3-window) and at p-1 (any group behind it)
            for (uint32_t g = 0; g < G; g++) { set_filter((uint64_t)g * G3 + x3); set_filter((uint64_t)x3 * G + g); }
        }
    }
    if (t.srec_big.empty()) t.srec_big.push_back(0);

    // ---- bucket table (scan2's 32-byte slots, two-choice placement) ----------------------------------------------------
    std::vector<Scan2Slot> items;        // one per key: the term itself, or the header of a multi-term bucket
    for (auto& kv : buckets) {
        auto& v = kv.second;
        std::stable_sort(v.begin(), v.end(), [](const Ent& a, const Ent& b) { return a.len1 > b.len1; });
        const uint32_t key = kv.first;
        set_filter(key);
        if (v.size() == 1) {
            items.push_back(make_slot(key, v[0], ac.terms[v[0].term_id]));
        } else {
            items.push_back(Scan2Slot{key, kScan2Multi | (uint32_t)t.more.size(), (uint32_t)v.size(), {0, 0, 0, 0, 0}});
            for (const Ent& e : v) t.more.push_back(make_slot(key, e, ac.terms[e.term_id]));
        }
    }
    {
        uint32_t lg = 10;
        while ((1ull << lg) < 2 * items.size()) lg++;        // load <= 0.5 (two choices place that easily)
        for (uint32_t attempt = 0;; attempt++) {
            if (attempt && attempt % 8 == 0) lg++;             // eight seeds per size, then the next size
            if (lg > 28) { t.why_not = "bucket table too large"; return; }
            t.slot_shift = 32 - lg;
            t.slot_seed = (attempt % 8) * 0x9E37u;
            t.slots.assign((size_t)1 << lg, Scan2Slot{kScan2EmptyKey, 0, 0, {0, 0, 0, 0, 0}});
            uint32_t rng = 0x2545F491u;
            bool ok = true;
            for (const Scan2Slot& it : items) {
                Scan2Slot cur = it;
                bool placed = false;
                for (int kick = 0; kick < 1000 && !placed; kick++) {
                    const uint32_t h0 = scan2_slot_hash(cur.key, 0, t.slot_shift, t.slot_seed),
                                   h1 = scan2_slot_hash(cur.key, 1, t.slot_shift, t.slot_seed);
                    if (t.slots[h0].key == kScan2EmptyKey) { t.slots[h0] = cur; placed = true; break; }
                    if (t.slots[h1].key == kScan2EmptyKey) { t.slots[h1] = cur; placed = true; break; }
                    rng = rng * 1664525u + 1013904223u;
                    std::swap(cur, t.slots[(rng >> 16) & 1 ? h1 : h0]);
                }
                if (!placed) { ok = false; break; }
            }
            if (ok) break;
        }
    }
    if (t.more.empty()) t.more.push_back(Scan2Slot{kScan2EmptyKey, 0, 0, {0, 0, 0, 0, 0}});
    if (t.more.size() >= (1u << 31)) { t.why_not = "bucket table too large"; return; }

    // ---- bloom: two bits per (window key, front group); an anchor without a byte in front owns every front group -------
    {
        size_t n_items = 0;
        for (const auto& it : bloom_items) n_items += it.second == ~0u ? G : 1;
        t.bloom_lg = kScan3BloomLdsLg;
        if (n_items > 60000) while (((size_t)1 << t.bloom_lg) * 4 < n_items && t.bloom_lg < 26) t.bloom_lg++;
        t.bloom.assign((size_t)1 << t.bloom_lg, 0);
        auto put = [&](uint32_t key, uint32_t g) {
            const uint32_t key5 = key * G + g;
            t.bloom[scan3_bloom_cell(key5, t.bloom_lg)] |= scan3_bloom_bits(key5);
        };
        for (const auto& it : bloom_items) {
            if (it.second == ~0u) for (uint32_t g = 0; g < G; g++) put(it.first, g);
            else put(it.first, it.second);
        }
    }
    t.supported = true;
}

// ---- host emulation of the kernel's table walk (tests of the table compiler) --------------------------------------------
namespace {
inline uint8_t foldb(uint8_t b, bool fold) { return fold && b >= 'A' && b <= 'Z' ? (uint8_t)(b + 32) : b; }

// does the anchor described by slot e sit with its window ending at p of text[0..n)?  Mirrors entry_ok (gft_scan3.hip)
bool entry_ok_host(const Scan3Tables& t, const Scan2Slot& e, const uint8_t* text, uint32_t n, uint32_t p, bool fold,
                   uint32_t lo, uint32_t hi) {
    const uint32_t L1 = e.len & kScan2LenMask;
    const int off = (int)(int8_t)(e.len >> 24);
    if (L1 > p + 1) return false;
    const int64_t pe = (int64_t)p + off;
    if (pe < (int64_t)lo || pe >= (int64_t)hi) return false;
    const uint32_t L = (uint32_t)((int)L1 + off);
    const uint8_t* tb = t.term_blob.data() + t.term_off[e.info];
    const uint32_t start = p + 1 - L1;
    if ((uint64_t)start + L > n) return false;
    for (uint32_t i = 0; i < L; i++) if (foldb(text[start + i], fold) != tb[i]) return false;
    // the inline copy must agree with the blob (checks make_slot): front bytes, tail and window
    const int nfront = (int)L1 - 4;
    for (int k = 0; k < (off > 0 ? 3 : 4); k++)
        for (int b = 0; b < 4; b++) {
            const int idx = (int)L1 - 8 - 4 * k + b;
            if (idx >= 0 && idx < nfront && (uint8_t)(e.front[k] >> (8 * b)) != tb[idx]) return false;
        }
    for (int b = 0; b < off; b++) if ((uint8_t)(e.front[3] >> (8 * b)) != tb[L1 + (uint32_t)b]) return false;
    for (int b = 0; b < (off < 0 ? 3 : 4); b++) if ((uint8_t)(e.front[4] >> (8 * b)) != tb[L1 - 4 + (uint32_t)b]) return false;
    return true;
}
}  // namespace

void scan3_emulate(const Scan3Tables& t, const uint8_t* text, uint32_t n, uint32_t lo, bool fold, bool pos_end,
                   std::vector<Scan3Hit>& out) {
    // one unit [lo, n) of a document text[0..n): probes at lo - 3, lo - 1 (border, long anchors only), lo + 1, lo + 3, ...
    const uint32_t G = t.G, hi = n;
    const uint8_t* cls = fold ? t.cls_fold : t.cls;
    auto g_at = [&](int64_t i) -> uint32_t { return i >= 0 && i < (int64_t)n ? cls[text[i]] : 0u; };
    auto byte_at = [&](int64_t i) -> uint32_t { return i >= 0 && i < (int64_t)n ? foldb(text[i], fold) : 0u; };
    for (int64_t p = (int64_t)lo - 3; p - 1 < (int64_t)hi; p += 2) {
        if (p < 0) continue;
        const bool regular = p > (int64_t)lo;
        const uint32_t g0 = g_at(p - 4), g1 = g_at(p - 3), g2 = g_at(p - 2), g3 = g_at(p - 1), g4 = g_at(p);
        const uint32_t X = ((g1 * G + g2) * G + g3) * G + g4;
        if (regular && !(t.filter[X >> 5] >> (X & 31) & 1)) continue;
        // short terms ending at p and at p - 1
        for (int which = 0; which < 2 && regular && !t.short3.empty(); which++) {
            const int64_t e = p - which;
            if (e < (int64_t)lo || e >= (int64_t)hi) continue;
            const uint32_t x3 = which ? (g1 * G + g2) * G + g3 : (g2 * G + g3) * G + g4;
            const uint32_t sid = t.short3[x3];
            if (!sid) continue;
            const uint32_t W = byte_at(e - 3) | byte_at(e - 2) << 8 | byte_at(e - 1) << 16 | byte_at(e) << 24;
            const uint32_t* rec;
            uint32_t cnt;
            if (sid == 255) { rec = t.srec_big.data() + t.short3_big[x3] + 1; cnt = rec[-1]; }
            else { rec = t.srec.data() + (size_t)sid * kScan3RecWords; cnt = 3; }
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t w0 = rec[2 * j], w1 = rec[2 * j + 1], L = w0 >> 28;
                if (!w0 || L > e + 1) continue;
                if (((W ^ w1) >> (8 * (4 - L))) != 0) continue;
                out.push_back(Scan3Hit{w0 & 0x0FFFFFFFu, (uint32_t)(pos_end ? e : e + 1 - L)});
            }
        }
        // terms of length >= 4 anchored at p
        const uint32_t key5 = X * G + g0;
        const uint32_t bits = scan3_bloom_bits(key5);
        if ((t.bloom[scan3_bloom_cell(key5, t.bloom_lg)] & bits) != bits) continue;
        const Scan2Slot* hit = nullptr;
        for (int which = 0; which < 2 && !hit; which++) {
            const Scan2Slot& s = t.slots[scan2_slot_hash(X, which, t.slot_shift, t.slot_seed)];
            if (s.key == X) hit = &s;
        }
        if (!hit) continue;
        const Scan2Slot* ents = hit;
        uint32_t n_ent = 1;
        if (hit->info & kScan2Multi) { ents = t.more.data() + (hit->info & ~kScan2Multi); n_ent = hit->len; }
        for (uint32_t j = 0; j < n_ent; j++) {
            const Scan2Slot& e = ents[j];
            if (!entry_ok_host(t, e, text, n, (uint32_t)p, fold, lo, hi)) continue;
            const uint32_t L1 = e.len & kScan2LenMask;
            const int off = (int)(int8_t)(e.len >> 24);
            out.push_back(Scan3Hit{e.info, (uint32_t)(pos_end ? (int64_t)p + off : (int64_t)p + 1 - L1)});
        }
    }
}

}  // namespace gft
