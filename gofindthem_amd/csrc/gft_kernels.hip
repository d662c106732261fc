// gft_kernels.hip -- HIP kernels (gfx950 / CDNA4, wave64) for the ProcessText hot path.
//
//   k_scan_units   Aho-Corasick traversal, one wavefront per work unit (a document, or a slice of a long one).
//                  Replaces (*Matcher).MatchAll as called from CloudflareForkEngine.FindSubstrings
//                  (finder/substringEngine.go:110-119).
//   k_gather       unit slabs -> canonical CSR (document order, reference emission order inside a document).
//   (the solver kernel lives in gft_solve.hip)
//   k_*scan*       exclusive prefix sums used to lay the CSR out.
//
// No MFMA anywhere: this is a byte automaton walk plus boolean/integer evaluation (HBM/LDS bound).
#include <hip/hip_runtime.h>

#include "gft_kernels.hpp"

namespace gft {

namespace {

constexpr uint32_t kLane = 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t l = lane_id();
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t o = __shfl_up(v, s, 64);
        if ((int)l >= s) v += o;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------------
// work units
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unit_count(const uint64_t* __restrict__ doc_off, uint64_t n_docs,
                                                    uint32_t unit_max, uint32_t* __restrict__ cnt,
                                                    uint32_t* __restrict__ bad) {
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    uint64_t n = doc_off[d + 1] - doc_off[d];
    // positions are u32 offsets into the document: longer documents (and descending offsets, which wrap) are refused
    if (n > 0xFFFFFFFFull) { atomicOr(bad, 1u); n = 0; }
    cnt[d] = n <= unit_max ? 1u : (uint32_t)((n + unit_max - 1) / unit_max);
}

// one thread: what the host wants to know after the unit prefix sum, side by side, so that it is ONE small copy
__global__ void k_pack_ctl(const uint64_t* __restrict__ unit_base, const uint64_t* __restrict__ doc_off, uint64_t n_docs,
                           uint64_t* __restrict__ out) {
    out[0] = unit_base[n_docs];
    out[1] = doc_off[0];
    out[2] = doc_off[n_docs];
}

// max_units: capacity of `units` (entries beyond it are dropped: the host sized the table before it knew the count, sees
// the true count afterwards and runs the batch again with a table that holds it)
__global__ void __launch_bounds__(256) k_unit_fill(const uint64_t* __restrict__ doc_off, uint64_t n_docs,
                                                   const uint64_t* __restrict__ unit_base, Unit* __restrict__ units,
                                                   uint64_t max_units, uint32_t unit_max) {
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    uint64_t n = doc_off[d + 1] - doc_off[d];
    // as k_unit_count: a document of 4 GiB or more (descending offsets wrap to that) has been flagged there and gets ONE
    // EMPTY unit -- the scan kernels trust hi - lo <= unit_max, and on the deferred path they are launched before the host
    // has read the flag
    if (n > 0xFFFFFFFFull) n = 0;
    uint64_t b = unit_base[d];
    uint32_t k = (uint32_t)(unit_base[d + 1] - b);
    if (k == 0) return;
    uint64_t per = (n + k - 1) / k;
    if (per > unit_max) per = unit_max;                 // (never taken when cnt came from k_unit_count with the same unit_max)
    for (uint32_t i = 0; i < k; i++) {
        uint64_t lo = (uint64_t)i * per, hi = lo + per < n ? lo + per : n;
        if (lo > n) lo = n;
        if (b + i < max_units) units[b + i] = Unit{(uint32_t)d, (uint32_t)lo, (uint32_t)hi};
    }
}

// The whole unit table in one launch for the batch in which every document is ONE unit (the common shape: documents up to
// unit_max bytes): units[d] = {d, 0, length}, unit_base[d] = d, and the three numbers k_pack_ctl reports.  A longer
// document only raises ctl32[7]: the host finds it with the batch's one read-back and runs the batch again on the
// general path (count, prefix sum, fill).
__global__ void __launch_bounds__(256) k_units_single(const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint32_t unit_max,
                                                      Unit* __restrict__ units, uint64_t* __restrict__ unit_base,
                                                      uint32_t* __restrict__ ctl32, uint32_t epoch) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d == 0) {
        // the control block of the batch: cursor, match count and the non-ASCII word start at zero (nobody else in this
        // launch touches them); the two flags are not cleared but RAISED TO THE BATCH'S EPOCH by whoever has cause to
        uint64_t* ctl64 = reinterpret_cast<uint64_t*>(ctl32);
        unit_base[n_docs] = n_docs;
        ctl32[1] = 0; ctl64[1] = 0; ctl64[2] = 0; ctl32[6] = 0;
        ctl64[4] = n_docs; ctl64[5] = doc_off[0]; ctl64[6] = doc_off[n_docs];
    }
    if (d >= n_docs) return;
    uint64_t n = doc_off[d + 1] - doc_off[d];
    if (n > 0xFFFFFFFFull) { atomicMax(ctl32, epoch); n = 0; }       // (as k_unit_count)
    if (n > unit_max) { atomicMax(ctl32 + 7, epoch); n = unit_max; }  // (no kernel ever sees a unit longer than unit_max)
    units[d] = Unit{(uint32_t)d, 0u, (uint32_t)n};
    unit_base[d] = d;
}

// words[idx[i]] = (words[idx[i]] & ~clr[i]) | set[i]: the bits of (expression, document) pairs the host solved
// (host_solve.hpp), into a device-resident bitmap; several patches may name one word
__global__ void __launch_bounds__(256) k_patch_words(uint32_t* __restrict__ words, const uint64_t* __restrict__ idx,
                                                     const uint32_t* __restrict__ clr, const uint32_t* __restrict__ set, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (clr[i]) atomicAnd(&words[idx[i]], ~clr[i]);
    if (set[i]) atomicOr(&words[idx[i]], set[i]);
}

// unit_base[i] = min(unit_base[i], cap): after a unit table that was too small, every consumer stays inside it
__global__ void __launch_bounds__(256) k_clamp_u64(uint64_t* __restrict__ v, uint64_t n, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && v[i] > cap) v[i] = cap;
}

// ------------------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64 (out has n+1 entries, out[n] = total).  Three small kernels, 4096 items / block.
// ------------------------------------------------------------------------------------------------------
constexpr uint32_t kScanItems = 16, kScanBlock = 256, kScanTile = kScanItems * kScanBlock;

__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t* total) {
    __shared__ uint64_t wsum[kScanBlock / 64];
    const uint32_t l = lane_id(), w = threadIdx.x >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint64_t o = __shfl_up(incl, s, 64);
        if ((int)l >= s) incl += o;
    }
    if (l == 63) wsum[w] = incl;
    __syncthreads();
    uint64_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < kScanBlock / 64; i++) {
        if (i < w) off += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total = tot;
    return off + incl - v;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_partials(const uint32_t* __restrict__ in, uint64_t n,
                                                              uint64_t* __restrict__ partial) {
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint64_t s = 0;
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++)
        if (base + i < n) s += in[base + i];
    uint64_t tot;
    block_excl_scan_u64(s, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_spine(uint64_t* __restrict__ partial, uint64_t n_part) {
    uint64_t carry = 0;
    for (uint64_t b = 0; b < n_part; b += kScanBlock) {
        uint64_t i = b + threadIdx.x;
        uint64_t v = i < n_part ? partial[i] : 0, tot;
        uint64_t ex = block_excl_scan_u64(v, &tot);
        if (i < n_part) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) partial[n_part] = carry;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_final(const uint32_t* __restrict__ in, uint64_t n,
                                                           const uint64_t* __restrict__ partial, uint64_t n_part,
                                                           uint64_t* __restrict__ out) {
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint64_t s = 0;
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_u64(s, &tot) + partial[blockIdx.x];
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = partial[n_part];
}

// ------------------------------------------------------------------------------------------------------
// k_scan_units (v1): two-tier class-compressed DFA, hot rows in LDS, the rest read through L2.
// One wave per unit: the unit's bytes are staged into the wave's LDS buffer with coalesced loads, lane k walks
// bytes [k*C, (k+1)*C) after a (max_term_len-1)-byte warm-up from the root and reports matches whose END lies in
// its own range.  Two walks per unit (count, then write after a ballot-free wave prefix sum) keep the unit's
// matches in text order; a single global cursor hands out the unit's slab.
// ------------------------------------------------------------------------------------------------------
template <bool WRITE>
__device__ __forceinline__ uint32_t walk_chunk(const ScanParams& P, const uint8_t* __restrict__ buf,
                                               const uint8_t* __restrict__ cls_lds,
                                               const uint32_t* __restrict__ delta_lds, uint32_t seg_lo,
                                               uint32_t walk_from, uint32_t my_lo, uint32_t my_hi, uint64_t out_idx) {
    uint32_t state = 0, cnt = 0;
    const uint32_t ncls = P.n_classes;
    for (uint32_t p = walk_from; p < my_hi; p++) {
        const uint32_t c = cls_lds[buf[p - seg_lo]];
        const uint32_t e = state < P.n_lds_states ? delta_lds[state * ncls + c] : P.delta[(size_t)state * ncls + c];
        state = e & 0x7FFFFFFFu;
        if ((e >> 31) && p >= my_lo) {
            uint32_t s = state;
            do {
                const uint32_t t = P.out_term[s];
                if (t != 0xFFFFFFFFu) {
                    if (WRITE) {
                        P.pool_term[out_idx + cnt] = t;
                        P.pool_pos[out_idx + cnt] = P.pos_end ? p : p + 1 - P.term_len[t];
                    }
                    cnt++;
                }
                s = P.out_link[s];
            } while (s != 0);
        }
    }
    return cnt;
}

__global__ void __launch_bounds__(kScanBlockThreads) k_scan_units(const ScanParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* cls_lds = smem;
    uint32_t* delta_lds = reinterpret_cast<uint32_t*>(smem + 256);
    const uint32_t tier_words = P.n_lds_states * P.n_classes;
    uint8_t* text_lds = smem + 256 + (((size_t)tier_words * 4 + 15) & ~(size_t)15);

    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls_lds[i] = P.byte_class[i];
    for (uint32_t i = threadIdx.x; i < tier_words; i += blockDim.x) delta_lds[i] = P.delta[i];
    __syncthreads();

    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint8_t* buf = text_lds + (size_t)wave * kTextBuf;
    const uint32_t warm = P.max_term_len > 0 ? P.max_term_len - 1 : 0;
    bool told_nonascii = false;

    for (uint64_t u = (uint64_t)blockIdx.x * wpb + wave; u < P.n_units; u += (uint64_t)gridDim.x * wpb) {
        const Unit un = P.units[u];
        const uint64_t dstart = P.doc_off[un.doc];
        const uint32_t seg_lo = un.lo > warm ? un.lo - warm : 0;
        const uint32_t nbytes = un.hi - seg_lo;       // <= kTextBuf by construction (host checks)
        const uint8_t* src = P.text + dstart + seg_lo;
        uint32_t hib = 0;
        for (uint32_t i = lane; i < nbytes; i += kLane) {
            uint8_t b = src[i];
            hib |= b;
            if (P.fold && b >= 'A' && b <= 'Z') b += 32;
            buf[i] = b;
        }
        if (P.fold && P.nonascii && !told_nonascii && __any((hib & 0x80u) != 0)) { told_nonascii = true; if (lane == 0) atomicOr(P.nonascii, 1u); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const uint32_t own = un.hi - un.lo;
        uint32_t C = ((own + 63) / 64 + 3) & ~3u;      // bytes per lane, multiple of 4
        if (((C >> 2) & 1u) == 0) C += 4;             // odd dword stride between lanes: conflict-free LDS reads
        const uint32_t my_lo = un.lo + lane * C;
        const uint32_t my_hi = my_lo + C < un.hi ? my_lo + C : un.hi;
        const bool active = my_lo < un.hi;
        const uint32_t walk_from = my_lo > seg_lo + warm ? my_lo - warm : seg_lo;

        uint32_t cnt = 0;
        if (active) cnt = walk_chunk<false>(P, buf, cls_lds, delta_lds, seg_lo, walk_from, my_lo, my_hi, 0);
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t total = __shfl(incl, 63, 64);
        uint64_t base = 0;
        if (lane == 0) {
            base = total ? atomicAdd(reinterpret_cast<unsigned long long*>(P.cursor), (unsigned long long)total) : 0;
            P.unit_start[u] = base;
            P.unit_count[u] = base + total <= P.pool_cap ? total : 0u;   // (beyond the pool: nothing is written, the host runs the batch again)
        }
        base = __shfl(base, 0, 64);
        if (active && cnt && base + total <= P.pool_cap)
            walk_chunk<true>(P, buf, cls_lds, delta_lds, seg_lo, walk_from, my_lo, my_hi, base + incl - cnt);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------
// k_gather: one wave per unit copies its slab to the final CSR position.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather(const uint64_t* __restrict__ unit_start,
                                                const uint32_t* __restrict__ unit_count,
                                                const uint64_t* __restrict__ unit_out, uint64_t n_units,
                                                const uint32_t* __restrict__ pool_term,
                                                const uint32_t* __restrict__ pool_pos, uint32_t* __restrict__ term_id,
                                                uint32_t* __restrict__ pos) {
    const uint32_t lane = lane_id(), wpb = blockDim.x >> 6;
    for (uint64_t u = (uint64_t)blockIdx.x * wpb + (threadIdx.x >> 6); u < n_units; u += (uint64_t)gridDim.x * wpb) {
        const uint64_t s = unit_start[u], d = unit_out[u];
        const uint32_t n = unit_count[u];
        for (uint32_t i = lane; i < n; i += kLane) {
            term_id[d + i] = pool_term[s + i];
            pos[d + i] = pool_pos[s + i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_gather_sorted: the same for slabs written by gft_scan2 (any order inside a unit; every match lies in the unit
// that holds its end position): a unit's matches are put into the reference's emission order -- end offset
// ascending, longer term first -- on the way.  One wave per unit.  Keys (end - unit.lo) << 18 | (2^18 - 1 - len) are
// distinct: two matches with the same end and length are the same term.  Units of up to kScan2FifoCap matches (the
// rule): bucketed rank sort, see below.  Larger units: plain rank sort, keys staged in LDS kScan2FifoCap at a time,
// every lane counts the keys below each of its own (several passes).
// ------------------------------------------------------------------------------------------------------
constexpr uint32_t kSortBins = 256;

__global__ void __launch_bounds__(256) k_gather_sorted(const Unit* __restrict__ units,
                                                       const uint64_t* __restrict__ unit_start,
                                                       const uint32_t* __restrict__ unit_count,
                                                       const uint64_t* __restrict__ unit_out, uint64_t n_units,
                                                       const uint32_t* __restrict__ pool_term,
                                                       const uint32_t* __restrict__ pool_pos,
                                                       const uint32_t* __restrict__ term_len, uint32_t pos_end,
                                                       uint32_t* __restrict__ term_id, uint32_t* __restrict__ pos) {
    __shared__ uint32_t keys_all[4][kScan2FifoCap];
    __shared__ uint32_t bins_all[4][2 * kSortBins];                  // per wave: count per bin, then start per bin
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint32_t* keys = keys_all[wave];
    uint32_t* bcnt = bins_all[wave];
    uint32_t* bstart = bins_all[wave] + kSortBins;
    constexpr uint32_t kPer = kScan2FifoCap / kLane;
    for (uint64_t u = (uint64_t)blockIdx.x * wpb + wave; u < n_units; u += (uint64_t)gridDim.x * wpb) {
        const uint64_t s = unit_start[u], d = unit_out[u];
        const uint32_t n = unit_count[u];
        const uint32_t lo = units[u].lo;
        auto key_of = [&](uint32_t i, uint32_t& t, uint32_t& p) {
            t = pool_term[s + i];
            p = pool_pos[s + i];
            const uint32_t L = term_len[t];
            const uint32_t end = pos_end ? p : p + L - 1;
            return (end - lo) << 18 | (0x3FFFFu - (L < 0x3FFFFu ? L : 0x3FFFFu));
        };
        if (n <= kScan2FifoCap) {
            // The common case, at most four matches per lane.  Bucketed rank sort: the end offsets of a unit span at most
            // 8 KiB, so 256 bins by end offset hold a few matches each; a match's rank = the start of its bin (histogram
            // + prefix sum) + the number of smaller keys inside the bin (a loop over the largest bin only, instead of over
            // all n keys).
            const uint32_t own = units[u].hi - lo;
            uint32_t shift = 0;
            while ((own >> shift) > kSortBins) shift++;
            uint32_t t[kPer], p[kPer], k[kPer], bin[kPer], slot[kPer];
            for (uint32_t i = lane; i < kSortBins; i += kLane) bcnt[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++) {
                const uint32_t i = lane + q * kLane;
                k[q] = 0xFFFFFFFFu; t[q] = 0; p[q] = 0; bin[q] = 0; slot[q] = 0;
                if (i < n) {
                    k[q] = key_of(i, t[q], p[q]);
                    bin[q] = (k[q] >> 18) >> shift;
                    if (bin[q] >= kSortBins) bin[q] = kSortBins - 1;
                    slot[q] = atomicAdd(&bcnt[bin[q]], 1u);              // arrival order inside the bin
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // exclusive prefix sum over the bins: four consecutive bins per lane, a wave scan over the lane sums
            constexpr uint32_t kBinsPerLane = kSortBins / kLane;
            uint32_t c[kBinsPerLane], mine = 0, cmax = 0;
#pragma unroll
            for (uint32_t j = 0; j < kBinsPerLane; j++) {
                c[j] = bcnt[lane * kBinsPerLane + j];
                mine += c[j];
                cmax = c[j] > cmax ? c[j] : cmax;
            }
            uint32_t run = wave_incl_scan(mine) - mine;
#pragma unroll
            for (uint32_t j = 0; j < kBinsPerLane; j++) { bstart[lane * kBinsPerLane + j] = run; run += c[j]; }
            for (int sh = 32; sh; sh >>= 1) { const uint32_t o = __shfl_xor(cmax, sh, 64); cmax = o > cmax ? o : cmax; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t base[kPer], cnt_b[kPer], rank[kPer];
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++) {
                base[q] = bstart[bin[q]];
                cnt_b[q] = bcnt[bin[q]];
                rank[q] = 0;
                if (lane + q * kLane < n) keys[base[q] + slot[q]] = k[q];       // keys grouped by bin
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t j = 0; j < cmax; j++) {
#pragma unroll
                for (uint32_t q = 0; q < kPer; q++)
                    if (j < cnt_b[q] && lane + q * kLane < n) rank[q] += keys[base[q] + j] < k[q] ? 1u : 0u;
            }
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++)
                if (lane + q * kLane < n) { term_id[d + base[q] + rank[q]] = t[q]; pos[d + base[q] + rank[q]] = p[q]; }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        for (uint32_t b0 = 0; b0 < n; b0 += kScan2FifoCap) {           // this pass's own items
            uint32_t t[kPer], p[kPer], k[kPer], rank[kPer];
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++) {
                const uint32_t i = b0 + lane + q * kLane;
                k[q] = 0xFFFFFFFFu; rank[q] = 0; t[q] = 0; p[q] = 0;
                if (i < n) k[q] = key_of(i, t[q], p[q]);
            }
            for (uint32_t t0 = 0; t0 < n; t0 += kScan2FifoCap) {       // all keys, a tile at a time
                __builtin_amdgcn_wave_barrier();
                if (t0 == b0) {
#pragma unroll
                    for (uint32_t q = 0; q < kPer; q++) keys[lane + q * kLane] = k[q];
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < kPer; q++) {
                        const uint32_t j = t0 + lane + q * kLane;
                        uint32_t tt, pp;
                        keys[lane + q * kLane] = j < n ? key_of(j, tt, pp) : 0xFFFFFFFFu;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t tn = n - t0 < kScan2FifoCap ? n - t0 : kScan2FifoCap;
                const uint32_t rounds = (((n - b0 < kScan2FifoCap ? n - b0 : kScan2FifoCap)) + kLane - 1) / kLane;   // own items in use
                for (uint32_t j = 0; j < tn; j++) {
                    const uint32_t kj = keys[j];                        // broadcast read
#pragma unroll
                    for (uint32_t q = 0; q < kPer; q++)
                        if (q < rounds) rank[q] += kj < k[q] ? 1u : 0u;
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++)
                if (b0 + lane + q * kLane < n) { term_id[d + rank[q]] = t[q]; pos[d + rank[q]] = p[q]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(256) k_match_off(const uint64_t* __restrict__ unit_base,
                                                   const uint64_t* __restrict__ unit_out, uint64_t n_docs,
                                                   uint64_t* __restrict__ match_off) {
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d <= n_docs) match_off[d] = unit_out[unit_base[d]];
}

// ---- unique terms per document (CloudflareEngine.FindSubstrings, finder/substringEngine.go:77-86: every term once, in
// the order of its first occurrence, Position 0).  One workgroup per document at a time; first[t] = index of the first
// match of term t (a row of n_terms words per workgroup, all ones between documents).  Pass 1 counts, pass 2 writes.
template <bool WRITE>
__global__ void __launch_bounds__(256) k_unique_terms(const uint64_t* __restrict__ match_off, const uint32_t* __restrict__ term,
                                                      uint64_t n_docs, uint32_t n_terms, uint32_t* __restrict__ first_all,
                                                      uint32_t* __restrict__ cnt, const uint64_t* __restrict__ out_off,
                                                      uint32_t* __restrict__ out_term) {
    uint32_t* first = first_all + (size_t)blockIdx.x * n_terms;
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t run;
    for (uint64_t d = blockIdx.x; d < n_docs; d += gridDim.x) {
        const uint64_t a = match_off[d], b = match_off[d + 1];
        for (uint64_t i = a + threadIdx.x; i < b; i += blockDim.x) atomicMin(&first[term[i]], (uint32_t)(i - a));
        __syncthreads();
        if (threadIdx.x == 0) run = 0;
        __syncthreads();
        // in order, 256 matches per round: a match stays if it is the first of its term
        uint32_t total = 0;
        for (uint64_t i0 = a; i0 < b; i0 += blockDim.x) {
            const uint64_t i = i0 + threadIdx.x;
            const bool keep = i < b && first[term[i]] == (uint32_t)(i - a);
            const uint64_t m = __ballot(keep);
            const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
            if (l == 0) wsum[w] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = run;
            for (uint32_t k = 0; k < w; k++) before += wsum[k];
            if (WRITE && keep) out_term[out_off[d] + before + (uint32_t)__popcll(m & ((1ull << l) - 1))] = term[i];
            total = run + wsum[0] + wsum[1] + wsum[2] + wsum[3];
            __syncthreads();
            if (threadIdx.x == 0) run = total;
            __syncthreads();
        }
        if (!WRITE && threadIdx.x == 0) cnt[d] = total;
        for (uint64_t i = a + threadIdx.x; i < b; i += blockDim.x) first[term[i]] = 0xFFFFFFFFu;
        __syncthreads();
    }
}

// ---- is ASCII case folding the whole of strings.ToLower for this text?  (finder/finder.go:140-142 lower-cases with
// strings.ToLower; the scan kernels fold A-Z only.)  Run when a folded scan saw bytes >= 0x80.  Accepted without a
// host round: ASCII, and the two-byte sequences C2 80..BF (Latin-1 signs: no case) and C3 9F..BF / C3 97 (Latin-1
// LOWER-case letters and the multiplication sign).  Anything else -- upper-case Latin-1 (C3 80..9E), every other lead
// byte, a continuation byte out of place (Go rewrites invalid UTF-8 to U+FFFD) -- sets *flag: the finder then lower-cases
// that batch on the host.  The rule looks at byte pairs (previous, this) only, so every thread checks its 16 bytes
// with one byte of context.
__global__ void __launch_bounds__(256) k_fold_safe(const uint8_t* __restrict__ text, uint64_t lo, uint64_t hi, uint32_t* __restrict__ flag) {
    // 16-byte blocks aligned in memory (one 16-byte load each); a block without a high bit -- nearly all of them even in
    // text that leaves ASCII now and then -- costs the load and two ORs, so the pass runs at streaming speed
    const uint64_t a0 = ((uint64_t)(uintptr_t)text + lo) & ~(uint64_t)15;          // address of the first block
    const uint64_t blk = a0 + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const uint64_t end = (uint64_t)(uintptr_t)text + hi, beg = (uint64_t)(uintptr_t)text + lo;
    bool bad = false;
    if (blk < end) {
        const uint4 v = *reinterpret_cast<const uint4*>((uintptr_t)blk);           // (reads < 16 bytes outside [lo, hi): inside the allocation's slack)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        // the byte in front of the block decides about the block's first byte (a lead byte that ends a block is judged
        // by the next block's thread): after a lead byte, even an ASCII byte is a violation
        const uint32_t prev = blk > beg ? *reinterpret_cast<const uint8_t*>((uintptr_t)(blk - 1)) : 0u;
        auto byte_at = [&](uint32_t k) { return (w[k >> 2] >> (8 * (k & 3))) & 0xFFu; };
        const uint32_t k_lo = blk < beg ? (uint32_t)(beg - blk) : 0u, k_hi = end - blk < 16 ? (uint32_t)(end - blk) : 16u;   // in-range bytes
        if ((prev == 0xC2u || prev == 0xC3u) && k_lo == 0 && byte_at(0) < 0x80u) bad = true;
        // one bit per byte that has its high bit set; the rule is evaluated for those bytes only
        uint32_t m = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) m |= ((((w[d] & 0x80808080u) >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * d);
        m &= (0xFFFFu << k_lo) & (0xFFFFu >> (16 - k_hi));
        while (m) {
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            const uint32_t b = byte_at(k), p = k > k_lo ? byte_at(k - 1) : (k == 0 ? prev : 0u);
            const bool p_lead = p == 0xC2u || p == 0xC3u;
            if (b == 0xC2u || b == 0xC3u) {
                // a lead byte: not behind another lead, and the byte behind it (if this block holds it) continues it
                bad |= p_lead;
                if (k + 1 < k_hi) bad |= (byte_at(k + 1) & 0xC0u) != 0x80u;
                else if (blk + k + 1 == end) bad = true;                            // a lead byte at the very end
            } else if ((b & 0xC0u) == 0x80u) {
                bad |= !p_lead || (p == 0xC3u && !(b >= 0x9Fu || b == 0x97u));     // C3 80..9E: upper-case Latin-1
            } else {
                bad = true;                                                         // any other lead byte, or 0xC0 / 0xC1 / 0xF8+
            }
        }
    }
    if (__any(bad) && (threadIdx.x & 63u) == 0) atomicOr(flag, 2u);
}
// ... and documents are lower-cased one by one: a pair that k_fold_safe accepts across a document border is a lead byte
// without continuation in one document and a continuation byte without lead in the next
__global__ void __launch_bounds__(256) k_fold_doc_edges(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doc_off, uint64_t n_docs,
                                                        uint32_t* __restrict__ flag) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (d < n_docs) {
        const uint64_t a = doc_off[d], b = doc_off[d + 1];
        if (b > a) {
            const uint32_t first = text[a], last = text[b - 1];
            bad = (first & 0xC0u) == 0x80u || last == 0xC2u || last == 0xC3u;
        }
    }
    if (__any(bad) && (threadIdx.x & 63u) == 0) atomicOr(flag, 2u);
}

// ---- rune offsets (GFT_POS_RUNES): AnknownEngine reports Position over []rune(text) (finder/substringEngine.go:44-53) ----
// A byte starts a rune of Go's decoder (range over a string: utf8.DecodeRuneInString, an invalid byte is one U+FFFD of
// width 1) unless it is a continuation byte that a VALID multi-byte sequence up to three bytes in front of it covers; a
// lead byte is never inside another sequence, so the rule is local.  Rune index of byte p = rune starts in [0, p).
constexpr uint32_t kRuneBlock = 64;         // bytes per counted block (a document's blocks count from its first byte)
// length of the valid UTF-8 sequence that starts at t[j] (1 for ASCII, and for anything Go decodes as a width-1 error);
// end = the document's end
__device__ __forceinline__ uint32_t utf8_valid_len(const uint8_t* __restrict__ t, uint64_t j, uint64_t end) {
    const uint32_t b0 = t[j];
    if (b0 < 0xC2u || b0 > 0xF4u) return 1;
    const uint32_t n = b0 < 0xE0u ? 2u : b0 < 0xF0u ? 3u : 4u;
    if (j + n > end) return 1;
    const uint32_t b1 = t[j + 1];
    const uint32_t lo = b0 == 0xE0u ? 0xA0u : b0 == 0xF0u ? 0x90u : 0x80u, hi = b0 == 0xEDu ? 0x9Fu : b0 == 0xF4u ? 0x8Fu : 0xBFu;   // (utf8.acceptRanges)
    if (b1 < lo || b1 > hi) return 1;
    for (uint32_t k = 2; k < n; k++)
        if ((t[j + k] & 0xC0u) != 0x80u) return 1;
    return n;
}
__device__ __forceinline__ bool rune_start(const uint8_t* __restrict__ t, uint64_t beg, uint64_t end, uint64_t i) {
    if ((t[i] & 0xC0u) != 0x80u) return true;
    for (uint32_t k = 1; k <= 3 && i >= beg + k; k++) {
        const uint64_t j = i - k;
        if ((t[j] & 0xC0u) != 0x80u) return utf8_valid_len(t, j, end) <= k;      // the only byte that could lead a sequence over i
    }
    return true;
}
__global__ void __launch_bounds__(256) k_rune_doc_blocks(const uint64_t* __restrict__ doc_off, uint64_t n_docs, uint32_t* __restrict__ cnt) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_docs) cnt[d] = (uint32_t)((doc_off[d + 1] - doc_off[d] + kRuneBlock - 1) / kRuneBlock);
}
// one thread per block of kRuneBlock bytes: the rune starts in it.  blk_base[d] = first block of document d
__global__ void __launch_bounds__(256) k_rune_block_starts(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doc_off,
                                                           const uint64_t* __restrict__ blk_base, uint64_t n_docs, uint64_t n_blocks,
                                                           uint32_t* __restrict__ starts) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint64_t lo = 0, hi = n_docs;                       // the document of block b: last d with blk_base[d] <= b
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) / 2; if (blk_base[mid] <= b) lo = mid; else hi = mid; }
    const uint64_t beg = doc_off[lo], end = doc_off[lo + 1];
    const uint64_t a = beg + (b - blk_base[lo]) * kRuneBlock, z = a + kRuneBlock < end ? a + kRuneBlock : end;
    uint32_t c = 0;
    for (uint64_t i = a; i < z; i++) c += rune_start(text, beg, end, i) ? 1u : 0u;
    starts[b] = c;
}
// one thread per match: byte offset -> rune offset
__global__ void __launch_bounds__(256) k_pos_to_rune(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doc_off,
                                                     const uint64_t* __restrict__ blk_base, const uint64_t* __restrict__ blk_prefix,
                                                     const uint64_t* __restrict__ match_off, uint64_t n_docs, uint64_t n_matches,
                                                     uint32_t* __restrict__ pos) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_matches) return;
    uint64_t lo = 0, hi = n_docs;                       // the document of match m: last d with match_off[d] <= m
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) / 2; if (match_off[mid] <= m) lo = mid; else hi = mid; }
    const uint64_t beg = doc_off[lo], end = doc_off[lo + 1];
    const uint32_t p = pos[m];
    const uint64_t g = p / kRuneBlock;
    uint64_t r = blk_prefix[blk_base[lo] + g] - blk_prefix[blk_base[lo]];
    for (uint64_t i = beg + g * kRuneBlock; i < beg + p; i++) r += rune_start(text, beg, end, i) ? 1u : 0u;
    pos[m] = (uint32_t)r;
}

inline unsigned grid_for(uint64_t n, unsigned per_block, unsigned cap) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
hipError_t launch_unit_count(const uint64_t* d_doc_off, uint64_t n_docs, uint32_t unit_max, uint32_t* d_cnt,
                             uint32_t* d_bad, hipStream_t st) {
    if (!n_docs) return hipSuccess;
    k_unit_count<<<dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, st>>>(d_doc_off, n_docs, unit_max, d_cnt, d_bad);
    return hipGetLastError();
}

hipError_t launch_pack_ctl(const uint64_t* d_unit_base, const uint64_t* d_doc_off, uint64_t n_docs, uint64_t* d_out, hipStream_t st) {
    k_pack_ctl<<<dim3(1), dim3(1), 0, st>>>(d_unit_base, d_doc_off, n_docs, d_out);
    return hipGetLastError();
}

hipError_t launch_patch_words(uint32_t* d_words, const uint64_t* d_idx, const uint32_t* d_clr, const uint32_t* d_set, uint64_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    k_patch_words<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_words, d_idx, d_clr, d_set, n);
    return hipGetLastError();
}

hipError_t launch_unit_fill(const uint64_t* d_doc_off, uint64_t n_docs, const uint64_t* d_unit_base, Unit* d_units,
                            uint32_t unit_max, hipStream_t st, uint64_t max_units) {
    if (!n_docs) return hipSuccess;
    k_unit_fill<<<dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, st>>>(d_doc_off, n_docs, d_unit_base, d_units, max_units, unit_max);
    return hipGetLastError();
}

hipError_t launch_units_single(const uint64_t* d_doc_off, uint64_t n_docs, uint32_t unit_max, Unit* d_units, uint64_t* d_unit_base,
                               uint32_t* d_ctl32, uint32_t epoch, hipStream_t st) {
    if (!n_docs) return hipSuccess;
    k_units_single<<<dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, st>>>(d_doc_off, n_docs, unit_max, d_units, d_unit_base, d_ctl32, epoch);
    return hipGetLastError();
}

hipError_t launch_clamp_u64(uint64_t* d_v, uint64_t n, uint64_t cap, hipStream_t st) {
    if (!n) return hipSuccess;
    k_clamp_u64<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_v, n, cap);
    return hipGetLastError();
}

uint64_t scan_partials_needed(uint64_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

hipError_t launch_exclusive_scan(const uint32_t* d_in, uint64_t n, uint64_t* d_out, uint64_t* d_partial,
                                 hipStream_t st) {
    const uint64_t n_part = (n + kScanTile - 1) / kScanTile;
    if (n_part == 0) {
        return hipMemsetAsync(d_out, 0, sizeof(uint64_t), st);
    }
    k_scan_partials<<<dim3((unsigned)n_part), dim3(kScanBlock), 0, st>>>(d_in, n, d_partial);
    k_scan_spine<<<dim3(1), dim3(kScanBlock), 0, st>>>(d_partial, n_part);
    k_scan_final<<<dim3((unsigned)n_part), dim3(kScanBlock), 0, st>>>(d_in, n, d_partial, n_part, d_out);
    return hipGetLastError();
}

size_t scan_units_lds_bytes(uint32_t n_lds_states, uint32_t n_classes) {
    return 256 + (((size_t)n_lds_states * n_classes * 4 + 15) & ~(size_t)15) +
           (size_t)(kScanBlockThreads / 64) * kTextBuf;
}

hipError_t launch_scan_units(const ScanParams& P, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const size_t lds = scan_units_lds_bytes(P.n_lds_states, P.n_classes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_units),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const unsigned wpb = kScanBlockThreads / 64;
    const unsigned grid = grid_for(P.n_units, wpb, n_cus * 2);
    k_scan_units<<<dim3(grid), dim3(kScanBlockThreads), lds, st>>>(P);
    return hipGetLastError();
}

hipError_t launch_gather(const uint64_t* d_unit_start, const uint32_t* d_unit_count, const uint64_t* d_unit_out,
                         uint64_t n_units, const uint32_t* d_pool_term, const uint32_t* d_pool_pos, uint32_t* d_term,
                         uint32_t* d_pos, const uint64_t* d_unit_base, uint64_t n_docs, uint64_t* d_match_off,
                         unsigned n_cus, hipStream_t st, const Unit* d_units_to_sort, const uint32_t* d_term_len,
                         uint32_t pos_end) {
    k_match_off<<<dim3((unsigned)((n_docs + 1 + 255) / 256)), dim3(256), 0, st>>>(d_unit_base, d_unit_out, n_docs,
                                                                                  d_match_off);
    if (n_units && d_units_to_sort)
        k_gather_sorted<<<dim3(grid_for(n_units, 4, n_cus * 16)), dim3(256), 0, st>>>(
            d_units_to_sort, d_unit_start, d_unit_count, d_unit_out, n_units, d_pool_term, d_pool_pos, d_term_len, pos_end,
            d_term, d_pos);
    else if (n_units)
        k_gather<<<dim3(grid_for(n_units, 4, n_cus * 16)), dim3(256), 0, st>>>(
            d_unit_start, d_unit_count, d_unit_out, n_units, d_pool_term, d_pool_pos, d_term, d_pos);
    return hipGetLastError();
}

hipError_t launch_fold_safe(const uint8_t* d_text, uint64_t lo, uint64_t hi, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_flag, hipStream_t st) {
    if (hi <= lo) return hipSuccess;
    const uint64_t a0 = ((uint64_t)(uintptr_t)d_text + lo) & ~(uint64_t)15, a1 = (uint64_t)(uintptr_t)d_text + hi;
    const uint64_t n = (a1 - a0 + 15) / 16;
    k_fold_safe<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_text, lo, hi, d_flag);
    if (d_doc_off && n_docs) k_fold_doc_edges<<<dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, st>>>(d_text, d_doc_off, n_docs, d_flag);
    return hipGetLastError();
}

hipError_t launch_rune_doc_blocks(const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_cnt, hipStream_t st) {
    if (!n_docs) return hipSuccess;
    k_rune_doc_blocks<<<dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, st>>>(d_doc_off, n_docs, d_cnt);
    return hipGetLastError();
}
hipError_t launch_rune_block_starts(const uint8_t* d_text, const uint64_t* d_doc_off, const uint64_t* d_blk_base, uint64_t n_docs,
                                    uint64_t n_blocks, uint32_t* d_starts, hipStream_t st) {
    if (!n_blocks) return hipSuccess;
    k_rune_block_starts<<<dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st>>>(d_text, d_doc_off, d_blk_base, n_docs, n_blocks, d_starts);
    return hipGetLastError();
}
hipError_t launch_pos_to_rune(const uint8_t* d_text, const uint64_t* d_doc_off, const uint64_t* d_blk_base, const uint64_t* d_blk_prefix,
                              const uint64_t* d_match_off, uint64_t n_docs, uint64_t n_matches, uint32_t* d_pos, hipStream_t st) {
    if (!n_matches) return hipSuccess;
    k_pos_to_rune<<<dim3((unsigned)((n_matches + 255) / 256)), dim3(256), 0, st>>>(d_text, d_doc_off, d_blk_base, d_blk_prefix, d_match_off, n_docs,
                                                                                  n_matches, d_pos);
    return hipGetLastError();
}

hipError_t launch_unique_terms(bool write, const uint64_t* d_match_off, const uint32_t* d_term, uint64_t n_docs, uint32_t n_terms,
                               uint32_t* d_first, unsigned grid, uint32_t* d_cnt, const uint64_t* d_out_off, uint32_t* d_out_term,
                               hipStream_t st) {
    if (!n_docs) return hipSuccess;
    if (write) k_unique_terms<true><<<dim3(grid), dim3(256), 0, st>>>(d_match_off, d_term, n_docs, n_terms, d_first, d_cnt, d_out_off, d_out_term);
    else k_unique_terms<false><<<dim3(grid), dim3(256), 0, st>>>(d_match_off, d_term, n_docs, n_terms, d_first, d_cnt, d_out_off, d_out_term);
    return hipGetLastError();
}

}  // namespace gft
