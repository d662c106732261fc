// gft_scan3.hip -- suffix-window Aho-Corasick scan, process-path form (matches of a unit in any order).
// Replaces (*Matcher).MatchAll behind CloudflareForkEngine.FindSubstrings (finder/substringEngine.go:110-119) for
// callers that only group matches by term (Finder.addMatchesToSolverMap, finder/finder.go:181-196 -> the solver kernel).
// Same tables as gft_scan2.hip (scan2_tables.hpp); gft_scan2.hip keeps the text-ordered form used for CSR results.
//
// What is different from gft_scan2.hip, and why (all from rocprofv3 counters: the kernel was bound by the ~100 G/s
// rate of L2 line requests, 68 % of wave cycles waiting, not by HBM bytes or ALU):
//   * the unit's text is read in rounds of 1 KiB with lane k taking bytes [16k, 16k+16): one fully coalesced
//     16-byte-per-lane load per round (8 lines) instead of 64 strided lane chunks (~50 line requests per load);
//     the 8 bytes of history a lane needs come from its neighbour lane with two __shfl_up;
//   * flagged positions are resolved while their bytes are still in registers: window key, the LDS short-term
//     record and the LDS front-byte fingerprint need no second look at the text;
//   * only positions that survive the fingerprint (about 1/3 of the flagged ones) are queued in LDS (key, front
//     bytes, position) and, densely packed over the lanes, go to the L2 bucket table;
//   * matches go to a per-wave LDS fifo and leave as one coalesced run; a unit with more matches than the fifo holds
//     is re-run writing directly (its size is known by then).
// No MFMA (byte automaton, not a contraction).
#include <hip/hip_runtime.h>

#include "gft_kernels.hpp"

namespace gft {

namespace {

struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U128u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U64u { uint32_t lo, hi; };

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) { return reinterpret_cast<const U32u*>(p)->v; }

// ASCII lower-casing of four packed bytes (finder/finder.go:140-142 for ASCII text)
__device__ __forceinline__ uint32_t fold4(uint32_t w) {
    const uint32_t h = w & 0x7F7F7F7Fu;
    const uint32_t ge_a = h + 0x3F3F3F3Fu;          // bit 7 set where byte >= 'A'
    const uint32_t gt_z = h + 0x25252525u;          // bit 7 set where byte >  'Z'
    const uint32_t up = ge_a & ~gt_z & ~w & 0x80808080u;
    return w | (up >> 2);
}
__device__ __forceinline__ uint32_t fold1(uint32_t b) { return (b - 'A' < 26u) ? b + 32 : b; }

struct Out {
    const Scan2Params& P;
    uint2* fifo;        // LDS
    uint32_t* fcnt;     // LDS
    bool direct;        // second run of an overflowing unit: write straight to the pool at `base`
    uint64_t base;
};

__device__ __forceinline__ void emit(const Out& o, uint32_t term, uint32_t pos) {
    const uint32_t idx = __hip_atomic_fetch_add(o.fcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (o.direct) {
        if (o.base + idx < o.P.pool_cap) { o.P.pool_term[o.base + idx] = term; o.P.pool_pos[o.base + idx] = pos; }
    } else if (idx < kScan2FifoCap) {
        o.fifo[idx] = make_uint2(term, pos);
    }
}

// terms of length >= 4 ending at document position p (window key x, the 4 bytes in front of the window in tw)
// `slot` = the first probe of the bucket table, loaded ahead of time by the caller
__device__ __forceinline__ void long_terms(const Out& o, const uint8_t* dbase, uint64_t doc_abs, uint32_t p, uint32_t x,
                                           uint32_t tw, uint4 slot) {
    const Scan2Params& P = o.P;
    uint32_t h = (x * kGoldDev) >> P.slot_shift;
    while (slot.x != x) {
        if (slot.x == kScan2EmptyKey) return;     // fingerprint / hashed-filter false positive
        h = (h + 1) & P.slot_mask;
        slot = *reinterpret_cast<const uint4*>(&P.slots[h]);
    }
    const bool simple = (slot.w & kScan2Simple) != 0;
    uint32_t n_ent = 1, more_at = 0;
    uint4 e;
    if (simple) {
        e = make_uint4(slot.w & 0x7FFFFFu, (slot.w >> 23) & 0xFFu, slot.y, slot.z);
    } else {
        more_at = slot.w;
        n_ent = P.more[more_at].term_id;
        e = *reinterpret_cast<const uint4*>(&P.more[more_at + 1]);
    }
    for (uint32_t j = 0;;) {
        const uint32_t L = e.y;
        bool ok = L <= p + 1 && ((tw ^ e.z) & e.w) == 0;
        if (ok && L > 8) {
            // the first L-8 bytes of the term against text[p+1-L .. p-8], four bytes at a time from the end; the loads
            // are independent.  term_blob has 4 bytes of slack before every term; the text needs 3 before the match.
            const uint8_t* tb = P.term_blob + P.term_off[e.x];
            const uint8_t* tp = dbase + (int64_t)p + 1 - L;
            const uint32_t n = L - 8;
            if (doc_abs + p + 1 - L >= 3) {
                uint32_t diff = 0;
                for (uint32_t c = 0; c * 4 < n; c++) {
                    const int32_t at = (int32_t)n - 4 - (int32_t)(c * 4);
                    uint32_t tv = load_u32_unaligned(tp + at);
                    const uint32_t wv = load_u32_unaligned(tb + at);
                    if (P.fold) tv = fold4(tv);
                    const uint32_t mask = at >= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (uint32_t)(-at));
                    diff |= (tv ^ wv) & mask;
                }
                ok = diff == 0;
            } else {
                for (uint32_t i = 0; i < n && ok; i++) {
                    uint32_t b = tp[i];
                    if (P.fold) b = fold1(b);
                    ok = b == tb[i];
                }
            }
        }
        if (ok) emit(o, e.x, P.pos_end ? p : p + 1 - L);
        if (++j >= n_ent) break;
        e = *reinterpret_cast<const uint4*>(&P.more[more_at + 1 + j]);
    }
}

template <bool HASHED>
__global__ void __launch_bounds__(kScan2Threads) k_scan3(const Scan2Params P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* cls = smem;
    uint32_t* filt = reinterpret_cast<uint32_t*>(smem + 256);
    uint8_t* short3 = smem + 256 + (size_t)P.filter_words * 4;
    uint8_t* fpt = short3 + P.short3_bytes;
    uint8_t* wave_lds_all = fpt + kScan2FptSize;

    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = P.cls[i];
    for (uint32_t i = threadIdx.x; i < P.filter_words; i += blockDim.x) filt[i] = P.filter[i];
    for (uint32_t i = threadIdx.x; i < P.short3_bytes / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(short3)[i] = reinterpret_cast<const uint32_t*>(P.short3)[i];
    for (uint32_t i = threadIdx.x; i < kScan2FptSize / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(fpt)[i] = reinterpret_cast<const uint32_t*>(P.fpt)[i];
    __syncthreads();
    const bool have_short = P.short3_bytes != 0;

    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t n_waves = blockDim.x >> 6;
    // per-wave LDS region: match fifo | this round's text (8 bytes of history + 1 KiB) | flagged-position list | counter
    uint8_t* wl = wave_lds_all + (size_t)wave * kScan2WaveLds;
    uint2* fifo = reinterpret_cast<uint2*>(wl);
    uint32_t* txt = reinterpret_cast<uint32_t*>(wl + kScan2FifoCap * 8);          // byte 16 + q = round offset q
    uint16_t* cand = reinterpret_cast<uint16_t*>(wl + kScan2FifoCap * 8 + kScan3TextBytes);
    uint32_t* fcnt = reinterpret_cast<uint32_t*>(wl + kScan2WaveLds - 16);
    const uint32_t kp = P.kp, kp2 = kp * kp, kp3 = kp2 * kp;
    const uint64_t lt_mask = (1ull << lane) - 1;

    uint64_t slab_next = 0, wave_matches = 0;   // wave-uniform
    uint32_t slab_left = 0;

    for (uint64_t u = (uint64_t)blockIdx.x * n_waves + wave; u < P.n_units; u += (uint64_t)gridDim.x * n_waves) {
        const Unit un = P.units[u];
        const uint64_t doc_abs = P.doc_off[un.doc];
        const uint8_t* dbase = P.text + doc_abs;
        const uint32_t own = un.hi - un.lo;
        const uint32_t rounds = (own + 1023) >> 10;
        Out o{P, fifo, fcnt, false, 0};

        for (int attempt = 0; attempt < 2; attempt++) {
            if (lane == 0) *fcnt = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // history of lane 0 in the first round: the 8 bytes in front of the unit, or "no-term" bytes at a document start
            uint32_t carry0, carry1, carry2, carry3;
            carry0 = carry1 = P.pad_byte * 0x01010101u;
            if (un.lo >= 16) {
                const U64u hv0 = *reinterpret_cast<const U64u*>(dbase + un.lo - 16);
                carry0 = hv0.lo; carry1 = hv0.hi;
            }
            if (un.lo >= 8) {
                const U64u hv = *reinterpret_cast<const U64u*>(dbase + un.lo - 8);
                carry2 = hv.lo; carry3 = hv.hi;
            } else {
                carry2 = carry3 = P.pad_byte * 0x01010101u;
                for (uint32_t q = 0; q < un.lo; q++) {       // un.lo in 1..7 does not occur for units cut by k_unit_fill
                    carry2 = (carry2 >> 8) | (carry3 << 24);
                    carry3 = (carry3 >> 8) | ((uint32_t)dbase[q] << 24);
                }
            }

            // a bucket probe in flight per lane (see stage B)
            bool pend = false;
            uint32_t pend_x = 0, pend_tw = 0, pend_p = 0;
            uint4 pend_slot = make_uint4(0, 0, 0, 0);
            auto finish_pending = [&]() {
                if (__any(pend)) {
                    if (pend) long_terms(o, dbase, doc_abs, pend_p, pend_x, pend_tw, pend_slot);
                    pend = false;
                }
            };

            for (uint32_t r = 0; r < rounds; r++) {
                const uint32_t p0 = un.lo + (r << 10) + lane * 16;
                const uint32_t nvalid = p0 < un.hi ? (un.hi - p0 < 16 ? un.hi - p0 : 16) : 0;
                uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                if (nvalid) {
                    const U128u v = *reinterpret_cast<const U128u*>(dbase + p0);
                    w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
                }
                uint32_t h0 = __shfl_up(w2, 1, 64), h1 = __shfl_up(w3, 1, 64);
                if (lane == 0) { h0 = carry2; h1 = carry3; }
                const uint32_t g0 = carry0, g1 = carry1;       // 8 more bytes of history for the fingerprints (lane 0 only)
                carry0 = __shfl(w0, 63, 64); carry1 = __shfl(w1, 63, 64);
                carry2 = __shfl(w2, 63, 64);
                carry3 = __shfl(w3, 63, 64);

                // ---- filter: 16 independent probes of the LDS bit table ------------------------------------------------
                uint32_t a1 = __umul24(cls[h1 >> 24], kp), a2 = __umul24(cls[(h1 >> 16) & 0xFF], kp2),
                         a3 = __umul24(cls[(h1 >> 8) & 0xFF], kp3);
                uint32_t acc = 0;
                const uint32_t wv[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int d = 0; d < 4; d++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t cl = cls[(wv[d] >> (8 * b)) & 0xFF];
                        const uint32_t x = a3 + a2 + a1 + cl;
                        a3 = __umul24(a2, kp); a2 = __umul24(a1, kp); a1 = __umul24(cl, kp);
                        const uint32_t fi = HASHED ? (x * kGoldDev) >> P.hash_shift : x;
                        const uint32_t fw = filt[fi >> 5];
                        acc = __builtin_amdgcn_alignbit(fw >> (fi & 31), acc, 1);
                    }
                }
                uint32_t fm = (acc >> 16) & ((1u << nvalid) - 1);
                if (P.dbg & 1) fm = 0;        // timing study: filter only

                // ---- this round's text to LDS: flagged positions are resolved from there, balanced over the lanes -----------
                if (lane == 0) { txt[0] = g0; txt[1] = g1; txt[2] = h0; txt[3] = h1; }
                if (nvalid) *reinterpret_cast<uint4*>(&txt[4 + lane * 4]) = make_uint4(w0, w1, w2, w3);
                const uint32_t f = __popc(fm);
                uint32_t fincl = f;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const uint32_t up = __shfl_up(fincl, sft, 64);
                    if ((int)lane >= sft) fincl += up;
                }
                const uint32_t ftotal = __shfl(fincl, 63, 64);
                const uint32_t round_base = un.lo + (r << 10);
                // 8 bytes ending at round offset q: tw = bytes q-7..q-4, win = q-3..q (LDS text, byte 16 + q = offset q)
                auto context = [&](uint32_t q, uint32_t& tw, uint32_t& tw2, uint32_t& x, uint32_t& x3) {
                    const uint32_t s0 = q + 9, sh = (s0 & 3) * 8;
                    const uint32_t dm = txt[(s0 >> 2) - 1], d0 = txt[s0 >> 2], d1 = txt[(s0 >> 2) + 1], d2 = txt[(s0 >> 2) + 2];
                    tw2 = __builtin_amdgcn_alignbit(d0, dm, sh);      // bytes q-11..q-8 (q >= 0: s0 >= 9, index >= 1)
                    tw = __builtin_amdgcn_alignbit(d1, d0, sh);
                    const uint32_t win = __builtin_amdgcn_alignbit(d2, d1, sh);
                    if (P.fold) { tw = fold4(tw); tw2 = fold4(tw2); }
                    x3 = (cls[(win >> 8) & 0xFF] * kp + cls[(win >> 16) & 0xFF]) * kp + cls[win >> 24];
                    x = cls[win & 0xFF] * kp3 + x3;
                };
                for (uint32_t l0 = 0; l0 < 64 && ftotal;) {       // passes over lane ranges that fit the LDS list
                    const uint32_t before = l0 ? __shfl(fincl, (int)l0 - 1, 64) : 0;
                    const bool fits = lane >= l0 && fincl - before <= kScan3ListCap;
                    const uint64_t fmask = __ballot(fits) >> l0;
                    const uint32_t nl = fmask == ~0ull >> l0 ? 64 - l0 : (uint32_t)__builtin_ctzll(~fmask);
                    const uint32_t l1 = l0 + nl;
                    const uint32_t ptotal = __shfl(fincl, (int)l1 - 1, 64) - before;
                    if (lane >= l0 && lane < l1) {
                        uint32_t wpos = fincl - f - before, mk = fm;
                        while (mk) {
                            const uint32_t i = __builtin_ctz(mk);
                            mk &= mk - 1;
                            cand[wpos++] = (uint16_t)(lane * 16 + i);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // stage A: LDS-only decisions; short terms are emitted, positions that may end a term of length >= 4
                    // are compacted in place to the front of the list (write index <= read index)
                    uint32_t ns = 0;
                    for (uint32_t i0 = 0; i0 < ptotal; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        const bool on = i < ptotal;
                        const uint32_t q = on ? cand[i] : 0;
                        uint32_t tw, tw2, x, x3;
                        context(q, tw, tw2, x, x3);
                        const uint32_t p = round_base + q;
                        const uint32_t sid = on && have_short ? short3[x3] : 0;
                        const uint32_t xm = scan2_fpt_xmix(x);
                        const bool go0 = scan2_fpt_pass(fpt[scan2_fpt_cell(x, 0)], xm, tw);
                        const bool go1 = scan2_fpt_pass(fpt[scan2_fpt_cell(x, 1)], xm, tw);
                        bool go_long = on && (go0 || go1);
                        if (P.dbg & 4) go_long = false;
                        if (sid && !(P.dbg & 8)) {   // terms of length <= 3 (record array: tiny, L1 resident)
                            const Scan2Short rec = P.shorts[sid];
#pragma unroll
                            for (uint32_t j = 0; j < 3; j++)
                                if (j < rec.n && rec.len[j] <= p + 1) emit(o, rec.term[j], P.pos_end ? p : p + 1 - rec.len[j]);
                        }
                        const uint64_t sb = __ballot(go_long);
                        if (go_long) cand[ns + (uint32_t)__popcll(sb & lt_mask)] = (uint16_t)q;
                        ns += (uint32_t)__popcll(sb);
                    }
                    if ((P.dbg & 2) && lane == 0) {
                        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg_counters), (unsigned long long)ptotal);
                        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg_counters + 1), (unsigned long long)ns);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // stage B: the survivors, densely packed over the lanes, go to the L2 bucket table.  Software pipeline:
                    // the probe is only ISSUED here; it is consumed one trip later (normally in the next round, after that
                    // round's filter and stage A), so its L2 latency is off the critical path.
                    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
                        finish_pending();
                        const uint32_t i = i0 + lane;
                        if (i < ns) {
                            const uint32_t q = cand[i];
                            uint32_t x3, tw2;
                            context(q, pend_tw, tw2, pend_x, x3);
                            pend_p = round_base + q;
                            pend_slot = *reinterpret_cast<const uint4*>(&P.slots[(pend_x * kGoldDev) >> P.slot_shift]);
                            pend = true;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    l0 = l1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();      // the next round overwrites the LDS text
            }
            finish_pending();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t nh = *fcnt;
            if (o.direct) break;
            // room for the unit's matches: from the wave's slab, one global atomic per P.slab matches
            if (nh > slab_left) {
                const uint32_t want = nh > P.slab ? nh : P.slab;
                uint64_t nb = 0;
                if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(P.cursor), (unsigned long long)want);
                slab_next = __shfl(nb, 0, 64);
                slab_left = want;
            }
            const uint64_t base = slab_next;
            slab_next += nh;
            slab_left -= nh;
            wave_matches += nh;
            if (lane == 0) { P.unit_start[u] = base; P.unit_count[u] = nh; }
            if (nh <= kScan2FifoCap) {
                if (base + nh <= P.pool_cap)
                    for (uint32_t i = lane; i < nh; i += 64) {
                        const uint2 rr = fifo[i];
                        P.pool_term[base + i] = rr.x;
                        P.pool_pos[base + i] = rr.y;
                    }
                break;
            }
            o.direct = true;      // more matches than the fifo holds: run the unit again, writing in place
            o.base = base;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && wave_matches)
        atomicAdd(reinterpret_cast<unsigned long long*>(P.n_matches), (unsigned long long)wave_matches);
}

}  // namespace

hipError_t launch_scan3(const Scan2Params& P, uint32_t waves, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const size_t lds = scan2_lds_bytes(P.filter_words, P.short3_bytes, waves);
    const void* fn = P.hashed ? reinterpret_cast<const void*>(k_scan3<true>) : reinterpret_cast<const void*>(k_scan3<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t g = (P.n_units + waves - 1) / waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    if (P.hashed) k_scan3<true><<<dim3(grid), dim3(waves * 64), lds, st>>>(P);
    else k_scan3<false><<<dim3(grid), dim3(waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
