#include "scan2_tables.hpp"

#include <algorithm>
#include <cstring>
#include <unordered_map>

namespace gft {

void build_scan2_tables(const AcTables& ac, Scan2Tables& t) {
    t = Scan2Tables();
    const uint32_t kp = ac.n_classes + 1;
    if (kp > 256) { t.why_not = "dictionary uses all 256 byte values"; return; }
    t.kp = kp;
    t.pad_class = ac.n_classes;
    memcpy(t.cls, ac.byte_class, 256);
    for (int b = 0; b < 256; b++) t.cls_fold[b] = ac.byte_class[(b >= 'A' && b <= 'Z') ? b + 32 : b];

    // budget for entering short terms under every window they end
    uint64_t expand = 0;
    for (const auto& s : ac.terms) {
        if (s.empty() || s.size() >= kWin) continue;
        uint64_t n = 1;
        for (size_t i = s.size(); i < kWin; i++) n *= kp;
        expand += n;
        if (expand > kMaxWindowKeys) { t.why_not = "too many short terms for this alphabet size"; return; }
    }

    std::unordered_map<uint32_t, std::vector<Scan2Entry>> buckets;
    buckets.reserve(ac.terms.size() * 2 + (size_t)expand);
    t.term_off.assign(1, 0);
    for (size_t id = 0; id < ac.terms.size(); id++) {
        const std::string& s = ac.terms[id];
        t.term_blob.insert(t.term_blob.end(), s.begin(), s.end());
        t.term_off.push_back((uint32_t)t.term_blob.size());
        const uint32_t L = (uint32_t)s.size();
        if (L == 0) continue;   // the empty keyword never matches
        Scan2Entry e{(uint32_t)id, L, 0, 0};
        for (int k = 0; k < 4; k++) {
            const int idx = (int)L - 8 + k;
            if (idx >= 0 && idx < (int)L - 4) {
                e.cmp_val |= (uint32_t)(uint8_t)s[idx] << (8 * k);
                e.cmp_mask |= 0xFFu << (8 * k);
            }
        }
        uint32_t tail = 0;            // radix value of the term's last min(L, 4) classes
        const uint32_t m = std::min(L, kWin);
        for (uint32_t i = L - m; i < L; i++) tail = tail * kp + ac.byte_class[(uint8_t)s[i]];
        if (L >= kWin) {
            buckets[tail].push_back(e);
        } else {
            uint32_t scale = 1, combos = 1;
            for (uint32_t i = 0; i < m; i++) scale *= kp;
            for (uint32_t i = m; i < kWin; i++) combos *= kp;
            for (uint32_t pre = 0; pre < combos; pre++) buckets[pre * scale + tail].push_back(e);
        }
    }
    t.n_keys = buckets.size();

    // filter
    uint64_t direct_bits = (uint64_t)kp * kp * kp * kp;
    t.hashed = direct_bits > kFilterDirectMaxBits;
    t.filter_bits = t.hashed ? kFilterHashedBits : (uint32_t)((direct_bits + 31) & ~31ull);
    t.hash_shift = 32;
    if (t.hashed) { uint32_t lg = 0; while ((1u << lg) < t.filter_bits) lg++; t.hash_shift = 32 - lg; }
    t.filter.assign(t.filter_bits / 32, 0);

    // slots: 16 bytes each; a bucket with one plain entry is answered by the slot alone
    uint32_t lg = 10;
    while ((1ull << lg) < 2 * t.n_keys) lg++;
    t.slot_shift = 32 - lg;
    t.slots.assign((size_t)1 << lg, Scan2Slot{kScan2EmptyKey, 0, 0, 0});
    const uint32_t smask = (1u << lg) - 1;
    for (auto& kv : buckets) {
        auto& v = kv.second;
        std::stable_sort(v.begin(), v.end(), [](const Scan2Entry& a, const Scan2Entry& b) { return a.len > b.len; });
        const uint32_t key = kv.first;
        const uint32_t fi = t.hashed ? (key * kGold) >> t.hash_shift : key;
        t.filter[fi >> 5] |= 1u << (fi & 31);
        uint32_t h = (key * kGold) >> t.slot_shift;
        while (t.slots[h].key != kScan2EmptyKey) h = (h + 1) & smask;
        Scan2Slot& s = t.slots[h];
        s.key = key;
        if (v.size() == 1 && v[0].len <= 255 && v[0].term_id < (1u << 23)) {
            s.cmp_val = v[0].cmp_val; s.cmp_mask = v[0].cmp_mask;
            s.info = kScan2Simple | v[0].len << 23 | v[0].term_id;
        } else {
            s.cmp_val = 0; s.cmp_mask = 0;
            s.info = (uint32_t)t.more.size();                       // header {count} then the entries, longest first
            t.more.push_back(Scan2Entry{(uint32_t)v.size(), 0, 0, 0});
            t.more.insert(t.more.end(), v.begin(), v.end());
        }
    }
    if (t.more.empty()) t.more.push_back(Scan2Entry{0, 0, 0, 0});
    t.supported = t.more.size() < (1u << 31);
    if (!t.supported) t.why_not = "bucket table too large";
}

}  // namespace gft
