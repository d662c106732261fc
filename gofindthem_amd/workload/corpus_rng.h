// Synthetic workload of SURVEY.md 8(d) -- shared host/device arithmetic (all integer, bit-reproducible).
//
// Stands in for the reference's benchmark corpus (benchmarks/benchmark_test.go:22-67,473-489), whose word list
// benchmarks/files/words.txt is missing from the reference mount.  Everything is counter-based splitmix64 so any
// document (and any word of it) can be generated independently on the CPU or by one GPU wave.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GFW_HD __host__ __device__ __forceinline__
#else
#define GFW_HD static inline
#endif

#define GFW_GOLDEN 0x9E3779B97F4A7C15ull
// reference's rand.Seed (benchmarks/benchmark_test.go:23) folded into the base seed
#define GFW_BASE_SEED (GFW_GOLDEN ^ 1629074756677820700ull)
#define GFW_VOCAB_WORDS 466550u   /* benchmarks/benchmark_test.go:43,72 */
#define GFW_PLANT_ONE_IN 20u      /* benchmarks/benchmark_test.go:479 */
#define GFW_DOC_MIN_BYTES 3072u
#define GFW_DOC_SPAN_BYTES 2049u  /* L ~ U[3072, 5120] */

GFW_HD uint64_t gfw_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// i-th output (i >= 0) of the splitmix64 stream seeded with `seed`
GFW_HD uint64_t gfw_stream(uint64_t seed, uint64_t i) { return gfw_mix(seed + GFW_GOLDEN * (i + 1)); }

GFW_HD uint64_t gfw_doc_key(uint64_t base_seed, uint64_t doc_id) { return gfw_mix(base_seed + 2 + doc_id); }
GFW_HD uint32_t gfw_doc_target_len(uint64_t key) { return GFW_DOC_MIN_BYTES + (uint32_t)(gfw_stream(key, 0) % GFW_DOC_SPAN_BYTES); }

// word i of a document: returns the index into the dictionary (planted, *is_term=1) or into the vocabulary
GFW_HD uint32_t gfw_doc_word(uint64_t key, uint32_t i, uint32_t n_vocab, uint32_t n_terms, int* is_term) {
    uint64_t r = gfw_stream(key, 1 + (uint64_t)i);
    uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
    if (n_terms > 0 && lo % GFW_PLANT_ONE_IN == 0) { *is_term = 1; return hi % n_terms; }
    *is_term = 0;
    return hi % n_vocab;
}
