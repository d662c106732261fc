#include "ac_tables.hpp"

#include <algorithm>
#include <cstring>

namespace gft {

void build_ac_tables(std::vector<std::string> terms, AcTables& t) {
    std::sort(terms.begin(), terms.end());
    terms.erase(std::unique(terms.begin(), terms.end()), terms.end());
    t.terms = std::move(terms);
    const size_t n_terms = t.terms.size();

    // byte classes: bytes seen in any term get classes 1..k in byte order
    bool seen[256] = {false};
    t.max_term_len = 0;
    t.term_len.resize(n_terms);
    for (size_t i = 0; i < n_terms; i++) {
        t.term_len[i] = (uint32_t)t.terms[i].size();
        t.max_term_len = std::max(t.max_term_len, t.term_len[i]);
        for (unsigned char c : t.terms[i]) seen[c] = true;
    }
    int n_seen = 0;
    for (int b = 0; b < 256; b++) n_seen += seen[b];
    if (n_seen == 256) {   // every byte occurs in some term: no "other" class needed, identity map
        t.n_classes = 256;
        for (int b = 0; b < 256; b++) t.byte_class[b] = (uint8_t)b;
    } else {
        t.n_classes = 1;
        for (int b = 0; b < 256; b++) t.byte_class[b] = seen[b] ? (uint8_t)(t.n_classes++) : 0;
    }
    const uint32_t ncls = t.n_classes;

    // Level-by-level trie construction over the SORTED term list: every state is a contiguous range of terms
    // sharing a prefix, so children come out grouped, in byte (== class) order, and BFS-numbered for free.
    struct Range { uint32_t lo, hi; };
    std::vector<Range> range;          // per state
    range.push_back({0, (uint32_t)n_terms});
    t.depth.assign(1, 0);
    t.out_term.assign(1, kNoTerm);
    t.in_class.assign(1, 0);
    t.child_begin.clear();
    std::vector<uint32_t> parent(1, 0);
    for (uint32_t s = 0; s < range.size(); s++) {
        const uint32_t d = t.depth[s];
        uint32_t lo = range[s].lo, hi = range[s].hi;
        t.child_begin.push_back((uint32_t)range.size());
        // terms of length exactly d sort first in the range
        if (lo < hi && t.terms[lo].size() == d) {
            if (d > 0) t.out_term[s] = lo;
            lo++;
        }
        while (lo < hi) {
            unsigned char c = (unsigned char)t.terms[lo][d];
            uint32_t e = lo + 1;
            while (e < hi && (unsigned char)t.terms[e][d] == c) e++;
            range.push_back({lo, e});
            t.depth.push_back(d + 1);
            t.out_term.push_back(kNoTerm);
            t.in_class.push_back(t.byte_class[c]);
            parent.push_back(s);
            lo = e;
        }
    }
    const uint32_t n_states = (uint32_t)range.size();
    t.child_begin.push_back(n_states);
    t.n_states = n_states;

    // failure links + full DFA rows in BFS (== id) order
    t.fail.assign(n_states, 0);
    t.out_link.assign(n_states, 0);
    t.delta.assign((size_t)n_states * ncls, 0);
    for (uint32_t s = 0; s < n_states; s++) {
        uint32_t* row = &t.delta[(size_t)s * ncls];
        if (s == 0) {
            // root: missing transitions stay at root
        } else {
            const uint32_t f = t.fail[s];
            memcpy(row, &t.delta[(size_t)f * ncls], sizeof(uint32_t) * ncls);
            t.out_link[s] = (t.out_term[f] != kNoTerm) ? f : t.out_link[f];
        }
        for (uint32_t ch = t.child_begin[s]; ch < t.child_begin[s + 1]; ch++) {
            const uint32_t c = t.in_class[ch];
            t.fail[ch] = (s == 0) ? 0 : (row[c] & ~kOutFlag);  // row[c] still holds delta[fail(s)][c]
            row[c] = ch;
        }
    }
    // flag transitions whose target emits
    for (size_t i = 0; i < t.delta.size(); i++) {
        const uint32_t tgt = t.delta[i];
        if (t.out_term[tgt] != kNoTerm || t.out_link[tgt] != 0) t.delta[i] = tgt | kOutFlag;
    }
}

}  // namespace gft
