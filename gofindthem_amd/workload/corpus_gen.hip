// Device half of the synthetic workload generator: writes the document blob straight into
// HBM so that a 1 M-document / ~4 GB corpus never crosses PCIe.  Bit-exact twin of corpus_host.cpp.
// Bench/test utility, not part of the drop-in library.
#include <hip/hip_runtime.h>

#include "corpus_rng.h"

namespace {

__global__ void __launch_bounds__(256) k_doc_lengths(uint64_t base_seed, uint64_t first, uint64_t n,
                                                     const uint64_t* __restrict__ vocab_off, uint32_t n_vocab,
                                                     const uint32_t* __restrict__ dict_idx, uint32_t n_terms,
                                                     uint32_t* __restrict__ len_out) {
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    uint64_t key = gfw_doc_key(base_seed, first + d);
    uint32_t L = gfw_doc_target_len(key), cur = 0;
    for (uint32_t i = 0; cur < L; i++) {
        int is_term; uint32_t w = gfw_doc_word(key, i, n_vocab, n_terms, &is_term);
        if (is_term) w = dict_idx[w];
        cur += (uint32_t)(vocab_off[w + 1] - vocab_off[w]) + 1;
    }
    len_out[d] = cur;
}

// one wave per document; lane i of batch b owns word b*64+i
__global__ void __launch_bounds__(256) k_doc_fill(uint64_t base_seed, uint64_t first, uint64_t n,
                                                  const uint8_t* __restrict__ vocab_blob,
                                                  const uint64_t* __restrict__ vocab_off, uint32_t n_vocab,
                                                  const uint32_t* __restrict__ dict_idx, uint32_t n_terms,
                                                  const uint64_t* __restrict__ doc_off, uint8_t* __restrict__ text) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t d = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (d >= n) return;
    uint64_t key = gfw_doc_key(base_seed, first + d);
    uint32_t L = gfw_doc_target_len(key);
    uint8_t* dst = text + doc_off[d];
    uint32_t base = 0;
    for (uint32_t b = 0; base < L; b++) {
        int is_term; uint32_t w = gfw_doc_word(key, b * 64 + lane, n_vocab, n_terms, &is_term);
        if (is_term) w = dict_idx[w];
        uint64_t so = vocab_off[w];
        uint32_t wl = (uint32_t)(vocab_off[w + 1] - so);
        uint32_t incl = wl + 1;
        for (int s = 1; s < 64; s <<= 1) {
            uint32_t v = __shfl_up(incl, s, 64);
            if ((int)lane >= s) incl += v;
        }
        uint32_t start = base + incl - (wl + 1);
        if (start < L) {
            const uint8_t* src = vocab_blob + so;
            for (uint32_t k = 0; k < wl; k++) dst[start + k] = src[k];
            dst[start + wl] = ' ';
        }
        base += __shfl(incl, 63, 64);
    }
}

// achievable-read ceiling: every lane sums 16-byte loads of a grid-strided slice of the buffer ("a
// trivial coalesced sum-reduction kernel over the same blob")
__global__ void __launch_bounds__(256) k_read_sum(const uint4* __restrict__ p, uint64_t n16, uint64_t* __restrict__ out) {
    // eight 16-byte loads in flight per lane, each wave instruction one contiguous KiB
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (uint64_t)v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc += (uint64_t)a.x + a.y + a.z + a.w; }
    for (int s = 32; s; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0 && acc == 0x0123456789ABCDEFull) out[0] = acc;    // keep the loads alive
}

}  // namespace

extern "C" {

int gfw_read_sum_dev(const void* d_buf, uint64_t bytes, uint64_t* d_scratch, void* stream) {
    const uint64_t n16 = bytes / 16;
    if (!n16) return 0;
    k_read_sum<<<dim3(256 * 8), dim3(256), 0, (hipStream_t)stream>>>((const uint4*)d_buf, n16, d_scratch);
    return (int)hipGetLastError();
}

int gfw_doc_lengths_dev(uint64_t base_seed, uint64_t first, uint64_t n, const uint64_t* d_vocab_off, uint32_t n_vocab,
                        const uint32_t* d_dict_idx, uint32_t n_terms, uint32_t* d_len_out, void* stream) {
    if (n == 0) return 0;
    k_doc_lengths<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        base_seed, first, n, d_vocab_off, n_vocab, d_dict_idx, n_terms, d_len_out);
    return (int)hipGetLastError();
}

int gfw_doc_fill_dev(uint64_t base_seed, uint64_t first, uint64_t n, const uint8_t* d_vocab_blob,
                     const uint64_t* d_vocab_off, uint32_t n_vocab, const uint32_t* d_dict_idx, uint32_t n_terms,
                     const uint64_t* d_doc_off, uint8_t* d_text, void* stream) {
    if (n == 0) return 0;
    k_doc_fill<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        base_seed, first, n, d_vocab_blob, d_vocab_off, n_vocab, d_dict_idx, n_terms, d_doc_off, d_text);
    return (int)hipGetLastError();
}

}  // extern "C"
