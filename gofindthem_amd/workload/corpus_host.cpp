// Host half of the synthetic workload generator: vocabulary, dictionary sampling and a CPU
// document generator that is the bit-exact twin of the HIP generator in corpus_gen.hip.  Bench/test utility,
// not part of the drop-in library.
#include "corpus_rng.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

namespace {
// English-like unigram weights a..z, parts per 10000
const uint16_t kLetterW[26] = {817, 149, 278, 425, 1270, 223, 202, 609, 697, 15, 77, 403, 241,
                               675, 751, 193, 10,  599, 633, 906, 276, 98, 236, 15, 197, 5};
struct Vocab {
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off;
};
Vocab* g_vocab = nullptr;

void make_vocab(uint64_t base_seed, uint32_t n_words, Vocab& v) {
    uint16_t cum[26]; uint32_t s = 0;
    for (int i = 0; i < 26; i++) { s += kLetterW[i]; cum[i] = (uint16_t)s; }   // s == 10000
    std::unordered_set<std::string> seen;
    seen.reserve(n_words * 2);
    v.blob.clear(); v.off.assign(1, 0);
    for (uint64_t cand = 0; v.off.size() - 1 < n_words; cand++) {
        uint64_t key = gfw_mix(base_seed + GFW_GOLDEN * 7 + cand);
        // length ~ clamp(round(N(9.4, 2.9)), 2, 24): Irwin-Hall sum of 12 16-bit uniforms (mean 393210, sd 65536)
        int64_t sum = 0;
        for (int j = 0; j < 3; j++) {
            uint64_t r = gfw_stream(key, j);
            sum += (r & 0xffff) + ((r >> 16) & 0xffff) + ((r >> 32) & 0xffff) + (r >> 48);
        }
        int64_t t = 99 * 65536 + 29 * (sum - 393210);       // (9.4 + 0.5 + 2.9 z) * 10 * 65536
        int64_t len = t >= 0 ? t / 655360 : -((-t + 655359) / 655360);
        if (len < 2) len = 2;
        if (len > 24) len = 24;
        std::string w((size_t)len, 'a');
        for (int64_t k = 0; k < len; k++) {
            uint32_t x = (uint32_t)(gfw_stream(key, 3 + k) % 10000);
            int c = 0;
            while (x >= cum[c]) c++;
            w[(size_t)k] = (char)('a' + c);
        }
        if (seen.insert(w).second) {
            v.blob.insert(v.blob.end(), w.begin(), w.end());
            v.off.push_back(v.blob.size());
        }
    }
}
}  // namespace

extern "C" {

// Builds (once) the vocabulary; returns number of words.  Blob/offset pointers stay valid until gfw_vocab_free.
uint32_t gfw_vocab_build(uint64_t base_seed, uint32_t n_words, const uint8_t** blob, const uint64_t** off) {
    if (!g_vocab) { g_vocab = new Vocab(); make_vocab(base_seed, n_words, *g_vocab); }
    *blob = g_vocab->blob.data();
    *off = g_vocab->off.data();
    return (uint32_t)g_vocab->off.size() - 1;
}
void gfw_vocab_free(void) { delete g_vocab; g_vocab = nullptr; }

// n_terms vocabulary indices sampled without replacement (partial Fisher-Yates, stream seed base+1)
void gfw_sample_dictionary(uint64_t base_seed, uint32_t n_vocab, uint32_t n_terms, uint32_t* out_idx) {
    std::vector<uint32_t> idx(n_vocab);
    for (uint32_t i = 0; i < n_vocab; i++) idx[i] = i;
    uint64_t key = gfw_mix(base_seed + 1);
    for (uint32_t i = 0; i < n_terms; i++) {
        uint32_t j = i + (uint32_t)(gfw_stream(key, i) % (n_vocab - i));
        std::swap(idx[i], idx[j]);
        out_idx[i] = idx[i];
    }
}

// Document lengths for docs [first, first+n): words (each followed by one space) appended until >= L bytes.
void gfw_doc_lengths_host(uint64_t base_seed, uint64_t first, uint64_t n, const uint64_t* vocab_off, uint32_t n_vocab,
                          const uint32_t* dict_idx, uint32_t n_terms, uint32_t* len_out) {
    for (uint64_t d = 0; d < n; d++) {
        uint64_t key = gfw_doc_key(base_seed, first + d);
        uint32_t L = gfw_doc_target_len(key), cur = 0;
        for (uint32_t i = 0; cur < L; i++) {
            int is_term; uint32_t w = gfw_doc_word(key, i, n_vocab, n_terms, &is_term);
            if (is_term) w = dict_idx[w];
            cur += (uint32_t)(vocab_off[w + 1] - vocab_off[w]) + 1;
        }
        len_out[d] = cur;
    }
}

void gfw_doc_fill_host(uint64_t base_seed, uint64_t first, uint64_t n, const uint8_t* vocab_blob,
                       const uint64_t* vocab_off, uint32_t n_vocab, const uint32_t* dict_idx, uint32_t n_terms,
                       const uint64_t* doc_off, uint8_t* text) {
    for (uint64_t d = 0; d < n; d++) {
        uint64_t key = gfw_doc_key(base_seed, first + d);
        uint32_t L = gfw_doc_target_len(key), cur = 0;
        uint8_t* dst = text + doc_off[d];
        for (uint32_t i = 0; cur < L; i++) {
            int is_term; uint32_t w = gfw_doc_word(key, i, n_vocab, n_terms, &is_term);
            if (is_term) w = dict_idx[w];
            uint32_t wl = (uint32_t)(vocab_off[w + 1] - vocab_off[w]);
            memcpy(dst + cur, vocab_blob + vocab_off[w], wl);
            dst[cur + wl] = ' ';
            cur += wl + 1;
        }
    }
}

}  // extern "C"
