#include "scan2_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>

namespace gft {

void build_scan2_tables(const AcTables& ac, Scan2Tables& t) {
    t = Scan2Tables();
    // Positions before the document start behave like bytes that occur in no term, so they share class 0 ("other");
    // that class exists unless the dictionary uses all 256 byte values.
    bool has_other = false;
    for (int b = 0; b < 256; b++) has_other |= ac.byte_class[b] == 0 && ac.n_classes < 256;
    if (!has_other || ac.n_classes > 255) { t.why_not = "dictionary uses all 256 byte values"; return; }
    const uint32_t kp = ac.n_classes;
    t.kp = kp;
    t.pad_class = 0;
    memcpy(t.cls, ac.byte_class, 256);
    for (int b = 0; b < 256; b++) t.cls_fold[b] = ac.byte_class[(b >= 'A' && b <= 'Z') ? b + 32 : b];

    const uint64_t kp3 = (uint64_t)kp * kp * kp;
    bool any_short = false;
    for (const auto& s : ac.terms) any_short |= !s.empty() && s.size() < kWin;
    // (too many byte classes for a direct K^3 table of the short terms: gft_scan2.hip cannot run, the tables of the long terms
    // are built all the same -- gft_scan5.hip takes its short terms from Scan3Tables then)
    t.short_direct = !(any_short && kp3 > kScan2Short3Max);
    if (!t.short_direct) t.why_not = "terms shorter than 4 bytes over a large alphabet";

    // ---- terms of length >= 4: buckets keyed by the four classes of their window ----------------------------------
    // The window ends `off` bytes before the term's end (gft_kernels.hpp, kScan2MaxOff).  Every text position whose last
    // four classes equal a window costs a stage-A check, and one that also matches the bytes in front of it costs a
    // bucket-table probe, so the build picks the window with the least expected cost under a unigram model of the text
    // (class frequencies of the dictionary itself: dictionaries and the text they are run on share an alphabet).
    std::vector<double> logp(kp, 0.0);
    {
        std::vector<uint64_t> cnt(kp, 0);
        uint64_t tot = 0;
        for (const auto& s : ac.terms) for (unsigned char ch : s) { cnt[ac.byte_class[ch]]++; tot++; }
        for (uint32_t cl = 0; cl < kp; cl++) logp[cl] = std::log(((double)cnt[cl] + 0.5) / ((double)tot + 0.5 * kp));
    }
    auto window_key = [&](const std::string& s, uint32_t L1) {
        uint32_t key = 0;
        for (uint32_t i = L1 - kWin; i < L1; i++) key = key * kp + ac.byte_class[(uint8_t)s[i]];
        return key;
    };
    // ... and, among windows of similar cost, one that no other term has taken: a bucket with a single term is verified
    // in the same trip that finds it, the entries of a shared bucket need another round of loads
    // ... and one whose PAIR of the bucket table still has room: a key's two candidate slots are the two slots of one 64-byte
    // pair (gft_kernels.hpp scan2_pair_slot), a pair holds two keys, and no key lives anywhere else -- so a window whose pair
    // is full is not a choice.  Terms of exactly four bytes have one window; they are placed first.  Returns false when some
    // term is left without a window (the caller tries another seed, then a larger table).
    std::unordered_map<uint32_t, uint32_t> taken;                // window key -> terms anchored there so far
    std::vector<uint8_t> pair_keys;                              // keys per pair of the table being planned
    std::vector<uint32_t> off_of(ac.terms.size(), 0);
    auto plan_offsets = [&](uint32_t lg, uint32_t seed) -> bool {
        taken.clear();
        pair_keys.assign((size_t)1 << (lg - 1), 0);
        const uint32_t shift = 32 - lg;
        auto pick_off = [&](const std::string& s, uint32_t* out) -> bool {
            const uint32_t L = (uint32_t)s.size();
            double cost[kScan2MaxOff + 1];
            uint32_t n = 0;
            for (uint32_t off = 0; off <= kScan2MaxOff && off + kWin <= L; off++, n++) {
                const uint32_t L1 = L - off;
                double w = 0, f = 0;
                for (uint32_t i = L1 - kWin; i < L1; i++) w += logp[ac.byte_class[(uint8_t)s[i]]];
                for (uint32_t i = 0; i < 3 && i + kWin < L1; i++) f += logp[ac.byte_class[(uint8_t)s[L1 - kWin - 1 - i]]];
                cost[off] = std::exp(w) * (1.0 + 3.0 * std::max(std::exp(f), 1.0 / 32));
            }
            bool any = false;
            uint32_t best = 0;
            double best_eff = 0;
            for (uint32_t off = 0; off < n; off++) {
                const uint32_t key = window_key(s, L - off);
                auto it = taken.find(key);
                if (it == taken.end() && pair_keys[scan2_pair_slot(key, 0, shift, seed) >> 1] >= 2) continue;   // (its pair is full)
                // sharing a bucket is worth avoiding unless the free window is several times more frequent in text
                const double eff = cost[off] * (it == taken.end() ? 1.0 : 4.0) * (off == 0 ? 0.999 : 1.0);
                if (!any || eff < best_eff) { best = off; best_eff = eff; any = true; }
            }
            if (!any) return false;
            const uint32_t key = window_key(s, L - best);
            if (taken[key]++ == 0) pair_keys[scan2_pair_slot(key, 0, shift, seed) >> 1]++;
            *out = best;
            return true;
        };
        for (int pass = 0; pass < 2; pass++)                     // the terms without a choice first
            for (size_t id = 0; id < ac.terms.size(); id++) {
                const std::string& s = ac.terms[id];
                if (s.size() < kWin || (s.size() == kWin) != (pass == 0)) continue;
                if (!pick_off(s, &off_of[id])) return false;
            }
        return true;
    };
    uint32_t n_long = 0;
    for (const auto& s : ac.terms) n_long += s.size() >= kWin;
    uint32_t plan_lg = 10;
    while ((1ull << plan_lg) < 2ull * n_long) plan_lg++;         // load <= 0.5 (as the two-choice placement before it)
    uint32_t plan_seed = 0;
    for (uint32_t attempt = 0;; attempt++) {
        if (attempt && attempt % 8 == 0) plan_lg++;              // eight seeds per size, then the next size
        if (plan_lg > 28) { t.why_not = "bucket table too large"; return; }
        plan_seed = (attempt % 8) * 0x9E37u;
        if (plan_offsets(plan_lg, plan_seed)) break;
    }
    struct Ent { uint32_t term_id, len, off; };   // len = len1: the term up to the end of its window
    std::unordered_map<uint32_t, std::vector<Ent>> buckets;
    buckets.reserve(ac.terms.size() * 2);
    std::vector<std::vector<uint32_t>> content;   // per 3-window: short terms ending there
    if (any_short && t.short_direct) content.resize((size_t)kp3);
    // term blob: 4 bytes of slack in front of every term (the kernel compares unaligned dwords that may start up to
    // 3 bytes before a term); term_off[id] points at the term's first byte
    t.term_off.clear();
    for (size_t id = 0; id < ac.terms.size(); id++) {
        const std::string& s = ac.terms[id];
        t.term_blob.insert(t.term_blob.end(), 4, 0);
        t.term_off.push_back((uint32_t)t.term_blob.size());
        t.term_blob.insert(t.term_blob.end(), s.begin(), s.end());
        const uint32_t L = (uint32_t)s.size();
        if (L == 0) continue;   // the empty keyword never matches
        if (L > kScan2LenMask) { t.why_not = "term longer than 16 MiB"; return; }
        const uint32_t off = L >= kWin ? off_of[id] : 0, L1 = L - off;
        uint32_t tail = 0;      // radix value of the window's classes (the whole term when it is shorter)
        const uint32_t m = std::min(L1, kWin);
        for (uint32_t i = L1 - m; i < L1; i++) tail = tail * kp + ac.byte_class[(uint8_t)s[i]];
        if (L >= kWin) {
            buckets[tail].push_back(Ent{(uint32_t)id, L1, off});
        } else if (t.short_direct) {
            // a short term ends every 3-window whose last L classes are the term
            uint32_t scale = 1, combos = 1;
            for (uint32_t i = 0; i < L; i++) scale *= kp;
            for (uint32_t i = L; i < 3; i++) combos *= kp;
            for (uint32_t pre = 0; pre < combos; pre++) content[(size_t)pre * scale + tail].push_back((uint32_t)id);
        }
    }
    t.n_keys = buckets.size();

    // ---- filter -----------------------------------------------------------------------------------------------------
    const uint64_t direct_bits = kp3 * kp;
    t.hashed = direct_bits > kFilterDirectMaxBits;
    t.filter_bits = t.hashed ? kFilterHashedBits : (uint32_t)((direct_bits + 31) & ~31ull);
    t.hash_shift = 32;
    if (t.hashed) { uint32_t lg = 0; while ((1u << lg) < t.filter_bits) lg++; t.hash_shift = 32 - lg; }
    t.filter.assign(t.filter_bits / 32, 0);
    auto set_filter = [&](uint32_t key) {
        const uint32_t fi = t.hashed ? (key * kGold) >> t.hash_shift : key;
        t.filter[fi >> 5] |= 1u << (fi & 31);
    };

    // ---- short3: records of up to three short terms, longest first -------------------------------------------------------
    t.shorts.assign(1, Scan2Short{0, {0, 0, 0}, {0, 0, 0}, 0});
    if (any_short && t.short_direct) {
        t.short3.assign(((size_t)kp3 + 15) & ~(size_t)15, 0);
        std::map<std::vector<uint32_t>, uint32_t> ids;
        for (size_t w = 0; w < content.size(); w++) {
            auto& v = content[w];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return ac.terms[a].size() > ac.terms[b].size(); });
            auto it = ids.find(v);
            if (it == ids.end()) {
                Scan2Short r{(uint32_t)v.size(), {0, 0, 0}, {0, 0, 0}, 0};
                for (size_t i = 0; i < v.size() && i < 3; i++) { r.term[i] = v[i]; r.len[i] = (uint32_t)ac.terms[v[i]].size(); }
                it = ids.emplace(v, (uint32_t)t.shorts.size()).first;
                t.shorts.push_back(r);
            }
            // the LDS byte table names the first 254 records; 255 = "look the id up in short3_big" (global memory)
            t.short3[w] = (uint8_t)std::min<uint32_t>(it->second, 255);
            if (it->second >= 255) {
                if (t.short3_big.empty()) t.short3_big.assign(t.short3.size(), 0);
                t.short3_big[w] = it->second;
            }
            for (uint32_t c0 = 0; c0 < kp; c0++) set_filter((uint32_t)(c0 * kp3 + w));   // any class may precede
        }
    }

    for (const Scan2Short& r : t.shorts)
        for (uint32_t j = 0; j < 3; j++) {
            if (j < r.n && r.term[j] >= (1u << 28)) { t.why_not = "term id too large for a short-term record"; return; }
            t.shorts_packed.push_back(j < r.n ? r.term[j] | r.len[j] << 28 : 0u);
        }

    // ---- bucket table (32-byte slots, two-choice placement) + fingerprint items ------------------------------------
    auto make_slot = [&](uint32_t key, const Ent& e) {
        const std::string& s = ac.terms[e.term_id];
        const int L = (int)e.len;                              // the window ends at byte L - 1 of the term
        Scan2Slot r{key, e.term_id, e.len | e.off << 24, {0, 0, 0, 0, 0}};
        for (int k = 0; k < (e.off ? 4 : 5); k++)
            for (int b = 0; b < 4; b++) {
                const int idx = L - 8 - 4 * k + b;             // term byte under text[p-7-4k+b]
                if (idx >= 0) r.front[k] |= (uint32_t)(uint8_t)s[idx] << (8 * b);
            }
        for (uint32_t b = 0; b < e.off; b++) r.front[4] |= (uint32_t)(uint8_t)s[L + b] << (8 * b);   // text[p+1+b]
        return r;
    };
    std::vector<Scan2Slot> items;        // one per key: the term itself, or the header of a multi-term bucket
    for (auto& kv : buckets) {
        auto& v = kv.second;
        std::stable_sort(v.begin(), v.end(), [](const Ent& a, const Ent& b) { return a.len > b.len; });
        const uint32_t key = kv.first;
        if (key == kScan2EmptyKey) { t.why_not = "window key collides with the empty marker"; return; }
        set_filter(key);
        if (v.size() == 1) {
            items.push_back(make_slot(key, v[0]));
        } else {
            items.push_back(Scan2Slot{key, kScan2Multi | (uint32_t)t.more.size(), (uint32_t)v.size(), {0, 0, 0, 0, 0}});
            for (const Ent& e : v) t.more.push_back(make_slot(key, e));
        }
    }
    {
        // every key into its pair (plan_offsets made sure that no pair gets a third)
        t.slot_shift = 32 - plan_lg;
        t.slot_seed = plan_seed;
        t.slots.assign((size_t)1 << plan_lg, Scan2Slot{kScan2EmptyKey, 0, 0, {0, 0, 0, 0, 0}});
        for (const Scan2Slot& it : items) {
            const uint32_t p0 = scan2_pair_slot(it.key, 0, t.slot_shift, t.slot_seed);
            if (t.slots[p0].key == kScan2EmptyKey) t.slots[p0] = it;
            else if (t.slots[p0 + 1].key == kScan2EmptyKey) t.slots[p0 + 1] = it;
            else { t.why_not = "bucket pair overflow (internal)"; return; }
        }
    }
    // ---- fingerprint table: one cell per term (gft_kernels.hpp) ---------------------------------------------------------
    {
        size_t n_items = 0;
        for (const auto& kv : buckets) n_items += kv.second.size();
        t.fpt_lg = 0;
        const bool force_global = getenv("GFT_SCAN_FPT_GLOBAL") != nullptr;     // timing studies
        if (n_items > kScan2FptLdsItems || force_global) { t.fpt_lg = 15; while (((size_t)1 << t.fpt_lg) * 2 < n_items * 5 && t.fpt_lg < 28) t.fpt_lg++; }
        const uint32_t flg = t.fpt_lg;
        const size_t n_cells = flg ? (size_t)1 << flg : kScan2FptSize;
        t.fpt.assign(n_cells, 0);
        struct GItem { uint32_t x, b1n; uint8_t val; };
        std::vector<uint32_t> ambiguous;                       // cells that must pass everything
        std::vector<uint8_t> pinned(n_cells, 0);
        std::map<std::pair<uint32_t, uint32_t>, uint8_t> groups;   // (key, b1n) -> the window's tag
        for (const auto& kv : buckets) {
            const uint32_t x = kv.first;
            const uint8_t tag = (uint8_t)scan2_fpt_xbyte(scan2_fpt_xmix(x));
            for (const Ent& e : kv.second) {
                const std::string& s = ac.terms[e.term_id];
                const int L = (int)e.len;
                if (L == 4) {                                  // no byte in front of the window: keyed by the window alone
                    const uint32_t c = scan2_fpt_xcell(x, flg);
                    if (t.fpt[c] != 0 && t.fpt[c] != tag) ambiguous.push_back(c);
                    t.fpt[c] = tag;
                    pinned[c] = 1;
                    continue;
                }
                groups[{x, (uint32_t)((uint8_t)s[L - 5] & 0xDFu)}] = tag;
            }
        }
        std::vector<GItem> singles;
        for (const auto& kv : groups) singles.push_back(GItem{kv.first.first, kv.first.second, kv.second});
        std::vector<GItem> owner(n_cells, GItem{0, 0, 0});
        uint32_t rng = 0x12345u;
        for (GItem cur : singles) {
            bool placed = false;
            for (int kick = 0; kick < 500 && !placed; kick++) {
                const uint32_t c0 = scan2_fpt_gcell(cur.x, cur.b1n, 0, flg), c1 = scan2_fpt_gcell(cur.x, cur.b1n, 1, flg);
                if (t.fpt[c0] == 0) { t.fpt[c0] = cur.val; owner[c0] = cur; placed = true; break; }
                if (t.fpt[c1] == 0) { t.fpt[c1] = cur.val; owner[c1] = cur; placed = true; break; }
                rng = rng * 1664525u + 1013904223u;
                uint32_t victim = (rng >> 16) & 1 ? c1 : c0;
                if (pinned[victim]) victim = victim == c0 ? c1 : c0;
                if (pinned[victim]) break;                      // both cells are taken by items that cannot move
                std::swap(cur, owner[victim]);                  // the newcomer moves in, the old tenant looks on
                t.fpt[victim] = owner[victim].val;
            }
            if (!placed) ambiguous.push_back(scan2_fpt_gcell(cur.x, cur.b1n, 0, flg));
        }
        // cells that pass everything are set last: whatever lived there passes too
        for (uint32_t c : ambiguous) t.fpt[c] = kScan2FptAmbiguous;
    }
    t.term_blob.insert(t.term_blob.end(), 8, 0);
    t.term_off.push_back((uint32_t)t.term_blob.size());
    if (t.more.empty()) t.more.push_back(Scan2Slot{kScan2EmptyKey, 0, 0, {0, 0, 0, 0, 0}});
    t.long_ok = t.more.size() < (1u << 31);
    if (!t.long_ok) t.why_not = "bucket table too large";
    t.supported = t.long_ok && t.short_direct;
}

}  // namespace gft

namespace gft {

void build_scan5_tables(const AcTables& ac, const Scan2Tables& s2, uint32_t G, Scan5Tables& t) {
    t = Scan5Tables();
    const uint32_t kp = s2.kp;
    if (G > kp) G = kp;
    if (G > 32) G = 32;              // (a filter word holds one bit per group)
    if (G < 2 && kp > 1) G = 2;      // (group 0 is class 0's alone: the other classes need one of their own)
    if (G < 1) G = 1;
    t.G = G;
    // class -> group.  Weights: how often a class occurs in the dictionary (the build's only model of the text, as in
    // pick_off); class 0 is "everything else" -- blanks, punctuation, the bytes of other alphabets -- and frequent in any text.
    std::vector<uint64_t> cnt(kp, 0);
    for (const auto& s : ac.terms) for (unsigned char ch : s) cnt[ac.byte_class[ch]]++;
    std::vector<uint32_t> order(kp);
    for (uint32_t c = 0; c < kp; c++) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if ((a == 0) != (b == 0)) return a == 0;
        return cnt[a] > cnt[b];
    });
    std::vector<uint32_t> group_of(kp, 0);
    std::vector<uint64_t> load(G, 0);
    for (uint32_t i = 0; i < kp; i++) {
        uint32_t g = i;
        if (i >= G) {
            g = 1;                                             // (group 0 is class 0's alone)
            for (uint32_t k = 1; k < G; k++) if (load[k] < load[g]) g = k;
        }
        group_of[order[i]] = g;
        load[g] += order[i] == 0 ? ~0ull / 4 : cnt[order[i]] + 1;
    }
    t.pad_group = group_of[s2.pad_class];
    for (int b = 0; b < 256; b++) { t.grp[b] = (uint8_t)group_of[s2.cls[b]]; t.grp_fold[b] = (uint8_t)group_of[s2.cls_fold[b]]; }
    // Every anchor window (a, b, c, d), in group space, under both of its 3-grams: bit a of the low word of entry (b, c, d) --
    // the probe at the window's last byte --, bit d of the high word of entry (a, b, c) -- the probe one byte earlier.  The
    // windows of the long terms are the keys of the bucket table; a term of length L <= 3 ends every window whose last L
    // classes are the term (what Scan2Tables::filter holds as bits, when it is direct; built here from the terms themselves,
    // so that an alphabet too large for that filter is served as well).
    t.filter.assign((size_t)G * G * G, 0);
    auto add_window = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        t.filter[((size_t)b * G + c) * G + d] |= 1ull << a;
        t.filter[((size_t)a * G + b) * G + c] |= 1ull << (32 + d);
    };
    for (const Scan2Slot& sl : s2.slots) {
        if (sl.key == kScan2EmptyKey) continue;
        const uint64_t key = sl.key;
        add_window(group_of[key / ((uint64_t)kp * kp * kp)], group_of[key / ((uint64_t)kp * kp) % kp], group_of[key / kp % kp], group_of[key % kp]);
    }
    const uint64_t all_lo = (1ull << G) - 1, all_hi = all_lo << 32;
    for (const auto& term : ac.terms) {
        const size_t L = term.size();
        if (L == 0 || L >= kWin) continue;
        uint32_t g[3] = {0, 0, 0};
        for (size_t i = 0; i < L; i++) g[i] = group_of[ac.byte_class[(uint8_t)term[i]]];
        if (L == 3) {
            t.filter[((size_t)g[0] * G + g[1]) * G + g[2]] |= all_lo;                                       // (any, t0, t1, t2)
            for (uint32_t a = 0; a < G; a++) t.filter[((size_t)a * G + g[0]) * G + g[1]] |= 1ull << (32 + g[2]);
        } else if (L == 2) {
            for (uint32_t b = 0; b < G; b++) {
                t.filter[((size_t)b * G + g[0]) * G + g[1]] |= all_lo;                                      // (any, any, t0, t1)
                for (uint32_t a = 0; a < G; a++) t.filter[((size_t)a * G + b) * G + g[0]] |= 1ull << (32 + g[1]);
            }
        } else {
            for (size_t e = 0; e < t.filter.size(); e++) t.filter[e] |= 1ull << (32 + g[0]);              // (any, any, any, t0)
            for (uint32_t b = 0; b < G; b++)
                for (uint32_t c = 0; c < G; c++) t.filter[((size_t)b * G + c) * G + g[0]] |= all_lo;
        }
    }
    (void)all_hi;
}

}  // namespace gft
