// gft_scan5.hip -- the suffix-window Aho-Corasick scan with ONE filter probe per TWO text bytes (tables: scan2_tables.hpp,
// build_scan5_tables).  Replaces (*Matcher).MatchAll behind CloudflareForkEngine.FindSubstrings
// (finder/substringEngine.go:110-119), as gft_scan2.hip does; same bucket / fingerprint / short-term tables, same results.
//
// gft_scan2.hip is bound by the CU's LDS unit and its vector ALU together: 983 LDS cycles per 4 KB document,
// 547 of them the filter's probes -- one ds_read_b32 at a random address per text byte, 8.5 cycles each with the bank
// conflicts that 64 random addresses have.  The filter here answers two end positions with one read:
//   FILTER   the table is indexed by a 3-gram of byte GROUPS (g[j-2], g[j-1], g[j]) and holds 64 bits: bit g[j-3] of the low
//            word says whether some term's anchor window is (g[j-3], g[j-2], g[j-1], g[j]) -- a window that ENDS at j --, bit
//            g[j+1] of the high word whether one is (g[j-2], g[j-1], g[j], g[j+1]) -- a window that ends at j+1.  One
//            ds_read_b64 per two bytes (half the probes, 11 VALU instructions per two bytes instead of 14), the same
//            per-position flags as gft_scan2's filter.  G^3 x 8 bytes: the byte classes are merged down to G groups (the
//            classes that are rare in the dictionary share: 27 -> 22 groups = 83 KB for 4 % more flagged positions); the
//            room comes from 4-byte fifo entries.  The keys of the verification stages stay EXACT classes, so everything
//            behind the filter (fingerprints, bucket slots, short records) is gft_scan2's.
//   STAGE A / B  gft_scan2.hip's (gft_scan2_dev.hpp).
//   OUTPUT   matches go to a 4-byte-per-entry LDS fifo (term id, or term id | relative position << term_bits), flushed
//            coalesced into the wave's slab.  A unit whose matches outgrow the fifo is verified a second time with the
//            appends going straight to a pool region of the counted size (no per-lane staging path).
// HBM traffic: text once + 4 B (8 B with positions) per match.  No MFMA (byte automaton, not a contraction).
// Measured and set aside (git history: 31cb86c): the same kernel with the unit's text resident in LDS -- the unaligned
// ds_read_b64 / b128 of the verification stages replay at 64 cycles each, and LDS cycles are what the kernel is short of.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gft_kernels.hpp"

namespace gft {


namespace {

#include "gft_scan2_dev.hpp"
#include "gft_foldsafe_dev.hpp"

constexpr int kWays5 = 2;           // stage A: candidates a lane works on at once (3 / 4 / 6 measured in round 4: + 1 / + 3 % / slower)

struct Ctx5 {
    uint32_t* fifo;              // LDS
    uint32_t fifo_cap;
    uint32_t nf;                 // matches of this unit so far (wave-uniform)
    uint32_t npend;              // short-term jobs parked at the END of the fifo (entry i = fifo[cap - 1 - i]; wave-uniform)
    bool lost;                   // wave-uniform: a match did not fit the fifo (or a parked job took its place): walk again
    bool direct;                 // wave-uniform: the second walk of a unit that outgrew the fifo -- appends go to the pool
    uint64_t dbase;              // ... at this entry
    uint32_t term_bits, pos_base;
};

__device__ __forceinline__ void out_append(const Scan2Params& P, Ctx5& o, bool em, uint32_t term, uint32_t pos) {
    const uint64_t mask = __ballot(em);
    if (em) {
        const uint32_t idx = o.nf + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (!o.direct) {
            if (idx < o.fifo_cap - o.npend) o.fifo[idx] = P.want_pos ? term | (pos - o.pos_base) << o.term_bits : term;
        } else {
            KARG(pool_term)[o.dbase + idx] = term;
            if (P.want_pos) KARG(pool_pos)[o.dbase + idx] = pos;
        }
    }
    o.nf += (uint32_t)__popcll(mask);
    if (!o.direct && o.nf + o.npend > o.fifo_cap) o.lost = true;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Short terms are emitted DENSELY: stage A only parks {position, record id} of the candidates whose 3-window ends one
// (a tenth of them: emitting in place cost three appends per 64 candidates, whoever had a hit), 64 parked jobs make one
// trip.  The jobs wait at the end of the match fifo, which gives up that room (a unit whose matches then do not fit is
// walked again with the appends going to the pool: the fifo is all theirs then).
__device__ __forceinline__ void short_trip(const Ctx& c, Ctx5& o, uint32_t ubase, uint32_t n) {       // the n (<= 64) most recently parked jobs
    const uint32_t lane = lane_id();
    const bool on = lane < n;
    const uint32_t job = o.fifo[o.fifo_cap - 1 - (o.npend - n) - (on ? lane : 0)];
    o.npend -= n;
    const uint32_t p = ubase + (job & 0xFFFFu), sid = on ? job >> 16 : 0;
    uint32_t r[3] = {0, 0, 0};
    if (sid) {
        const uint32_t* src = c.lrec + 3 * sid;
        r[0] = src[0]; r[1] = src[1]; r[2] = src[2];
    }
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        if (j && !__any(r[j] != 0)) break;
        const uint32_t L = r[j] >> 28;
        out_append(c.P, o, r[j] != 0 && L <= p + 1, r[j] & 0x0FFFFFFFu, c.P.pos_end ? p : p + 1 - L);
    }
}
// sid != 0: this lane's candidate (list entry rel) ends short terms; wave-uniform call
__device__ __forceinline__ void park_short(const Ctx& c, Ctx5& o, uint32_t ubase, uint32_t rel, uint32_t sid, uint32_t x3) {
    if (!__any(sid != 0)) return;
    if (__builtin_expect(__any(sid == 255) && KARG(short3_big) != nullptr, 0)) {
        // records beyond the LDS ids (a dictionary with hundreds of distinct short-term sets): emitted in place
        uint32_t r[3] = {0, 0, 0};
        const uint32_t p = ubase + rel;
        if (sid) short_record(c, sid, x3, r);
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) {
            const uint32_t L = r[j] >> 28;
            out_append(c.P, o, r[j] != 0 && L <= p + 1, r[j] & 0x0FFFFFFFu, c.P.pos_end ? p : p + 1 - L);
        }
        return;
    }
    const uint64_t m = __ballot(sid != 0);
    const uint32_t n = (uint32_t)__popcll(m);
    if (o.npend + n > o.fifo_cap / 2) {                            // (never in practice: keep half of the fifo for matches)
        wave_lds_sync();
        while (o.npend) short_trip(c, o, ubase, o.npend < 64 ? o.npend : 64);
        wave_lds_sync();
    }
    if (!o.direct && o.nf + o.npend + n > o.fifo_cap) o.lost = true;   // (the jobs go where matches are)
    if (sid) o.fifo[o.fifo_cap - 1 - o.npend - __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = rel | sid << 16;
    o.npend += n;
}

// ---- short terms over an alphabet too large for the direct K^3 table (SG = true): the tables of the stride-2 kernel
// (scan3_tables.hpp) -- a record id per 3-window of byte GROUPS, records of up to three {term | len << 28, the term's bytes
// as text[e-3 .. e]} -- so a record's terms are verified against the text (gft_scan3.hip stage_s).  Stage A parks {position,
// record id}; ids of 255 name cells whose records live in global memory (short3_big / srec_big) ------------------------------
__device__ __forceinline__ void short_trip_g(const Ctx& c, Ctx5& o, lds_u8* lsg, uint32_t ubase, uint32_t n) {
    const Scan2Params& P = c.P;
    const uint32_t lane = lane_id();
    const bool on = lane < n;
    const uint32_t job = o.fifo[o.fifo_cap - 1 - (o.npend - n) - (on ? lane : 0)];
    o.npend -= n;
    const uint32_t p = ubase + (job & 0xFFFFu);
    uint32_t sid = on ? job >> 16 : 0;
    const Text8 t8 = cand_load(c, p);
    const uint32_t W = P.fold ? fold4(t8.w) : t8.w;               // text[p-3 .. p]
    if (__builtin_expect(__any(sid == 255), 0)) {
        const uint32_t G = P.s5_sG;
        const uint32_t x3 = (lsg[(t8.w >> 8) & 0xFF] * G + lsg[(t8.w >> 16) & 0xFF]) * G + lsg[t8.w >> 24];
        const bool b = sid == 255;
        uint32_t cnt = 0;
        const uint32_t* g = nullptr;
        if (b) { g = KARG(s5_srec_big) + KARG(short3_big)[x3]; cnt = g[0]; }
        for (uint32_t j = 0; __any(j < cnt); j++) {
            uint32_t w0 = 0, w1 = 0;
            if (j < cnt) { w0 = g[1 + 2 * j]; w1 = g[2 + 2 * j]; }
            const uint32_t L = w0 >> 28;
            const bool ok = w0 != 0 && L <= p + 1 && ((W ^ w1) >> ((32 - 8 * L) & 31)) == 0;
            out_append(P, o, ok, w0 & 0x0FFFFFFFu, P.pos_end ? p : p + 1 - L);
        }
        if (b) sid = 0;
    }
    if (!__any(sid != 0)) return;
    const uint32_t* r = c.lrec + sid * kScan3RecWords;
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        const uint32_t w0 = sid ? r[2 * j] : 0;
        if (j && !__any(w0 != 0)) break;
        const uint32_t L = w0 >> 28;
        const bool ok = w0 != 0 && L <= p + 1 && ((W ^ r[2 * j + 1]) >> ((32 - 8 * L) & 31)) == 0;
        out_append(P, o, ok, w0 & 0x0FFFFFFFu, P.pos_end ? p : p + 1 - L);
    }
}
__device__ __forceinline__ void park_short_g(const Ctx& c, Ctx5& o, lds_u8* lsg, uint32_t ubase, uint32_t rel, uint32_t sid) {
    if (!__any(sid != 0)) return;
    const uint64_t m = __ballot(sid != 0);
    const uint32_t n = (uint32_t)__popcll(m);
    if (o.npend + n > o.fifo_cap / 2) {
        wave_lds_sync();
        while (o.npend) short_trip_g(c, o, lsg, ubase, o.npend < 64 ? o.npend : 64);
        wave_lds_sync();
    }
    if (!o.direct && o.nf + o.npend + n > o.fifo_cap) o.lost = true;   // (the jobs go where matches are)
    if (sid) o.fifo[o.fifo_cap - 1 - o.npend - __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = rel | sid << 16;
    o.npend += n;
}

// gft_scan2_dev.hpp finish_long / drain_deferred with the text in LDS and the 4-byte fifo
__device__ __forceinline__ void finish_long5(const Ctx& c, Ctx5& o, bool on, uint32_t rel, const Cand& k, const Slot& s0, const Slot& s1,
                                             Front t, uint32_t tl, Deferred& d) {
    const Scan2Params& P = c.P;
    Slot e;
    const bool have = slot_pick(k.x, s0, s1, e) && on;
    if (!__any(have)) return;
    const bool multi = have && (e.a.y & kScan2Multi);
    uint32_t folded = 0;
    {   // one-term buckets
        const bool act = have && !multi;
        const uint32_t kmax = wave_kmax(act ? e.a.z & kScan2LenMask : 0);
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = act && entry_ok(c, k.p, t, tl, e, kmax);
        out_append(P, o, ok, e.a.y, match_pos(P, k.p, e.a.z));
    }
    if (!__any(multi)) return;
    const uint32_t n_ent = multi ? e.a.z : 0, more_at = e.a.y & ~kScan2Multi;
    const uint32_t tot = lane_value(wave_incl_scan(n_ent), 63);
    if (tot <= d.cap - d.n) {
        for (uint32_t j = 0; __any(j < n_ent); j++) {
            const uint64_t m = __ballot(j < n_ent);
            if (j < n_ent)
                d.list[d.n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] =
                    make_uint2(rel, more_at + j);
            d.n += (uint32_t)__popcll(m);
        }
        return;
    }
    Slot cur = e;
    if (multi) cur = slot_load(&P.more[more_at]);
    for (uint32_t j = 0; __any(j < n_ent); j++) {
        const bool act = j < n_ent;
        Slot nxt = cur;
        if (j + 1 < n_ent) nxt = slot_load(&P.more[more_at + j + 1]);
        const uint32_t kmax = wave_kmax(act ? cur.a.z & kScan2LenMask : 0);
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = act && entry_ok(c, k.p, t, tl, cur, kmax);
        out_append(P, o, ok, cur.a.y, match_pos(P, k.p, cur.a.z));
        cur = nxt;
    }
}


__device__ __forceinline__ void drain_deferred5(const Ctx& c, Ctx5& o, uint32_t ubase, Deferred& d) {
    const Scan2Params& P = c.P;
    const uint32_t lane = lane_id();
    wave_lds_sync();
    for (uint32_t i0 = 0; i0 < d.n; i0 += 64) {
        const bool on = i0 + lane < d.n;
        const uint2 it = d.list[on ? i0 + lane : 0];
        const uint32_t p = ubase + it.x;
        const Slot e = slot_load(&P.more[it.y]);
        const Text8 t8 = cand_load(c, p);
        Front t = front_load(c, p, t8.tw);
        const uint32_t tl = tail_load(c, p);
        const uint32_t kmax = wave_kmax(on ? e.a.z & kScan2LenMask : 0);
        uint32_t folded = 0;
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = on && entry_ok(c, p, t, tl, e, kmax);
        out_append(P, o, ok, e.a.y, match_pos(P, p, e.a.z));
    }
    d.n = 0;
    __builtin_amdgcn_wave_barrier();
}

// FPT_LDS: the fingerprint table is staged in LDS; DBG: the timing-study instantiation (GFT_SCAN_DEBUG); SG: the short terms
// come from the group-indexed tables of the stride-2 kernel (an alphabet of more than 32 byte classes: short_trip_g)
template <bool FPT_LDS, bool DBG, bool SG>
__global__ void __launch_bounds__(kScan2Threads) k_scan5(const Scan2Params P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* grp = smem;                                          // byte -> filter group
    uint8_t* cls = smem + 256;                                    // byte -> exact class
    uint8_t* sgrp = smem + 512;                                   // SG: byte -> group of the short-term tables
    uint64_t* dual = reinterpret_cast<uint64_t*>(smem + 768);     // [G^3] the filter
    uint8_t* short3 = reinterpret_cast<uint8_t*>(dual + P.s5_dual);
    uint8_t* fpt = short3 + P.short3_bytes;                       // (short3_bytes is a multiple of 16)
    // (the fingerprint table in global memory: its place holds the Bloom level in front of it, if there is one)
    const uint32_t bloom_lg = FPT_LDS ? 0u : __builtin_amdgcn_readfirstlane(P.s5_bloom_lg);
    const uint32_t bloom_bytes = bloom_lg ? (1u << bloom_lg) / 8 : 0u;
    uint32_t* lrec = reinterpret_cast<uint32_t*>(fpt + (FPT_LDS ? kScan2FptSize : bloom_bytes));
    uint32_t* wg_next = reinterpret_cast<uint32_t*>(smem + (((size_t)(reinterpret_cast<uint8_t*>(lrec) - smem) + P.shorts_words * 4 + 15) & ~(size_t)15));
    uint8_t* wave_lds_all = reinterpret_cast<uint8_t*>(wg_next) + 16;

    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) { grp[i] = P.s5_grp[i]; cls[i] = P.cls[i]; sgrp[i] = SG ? P.s5_sgrp[i] : 0; }
    for (uint32_t i = threadIdx.x; i < P.s5_dual; i += blockDim.x) dual[i] = P.s5_filter[i];
    for (uint32_t i = threadIdx.x; i < P.short3_bytes / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(short3)[i] = reinterpret_cast<const uint32_t*>(P.short3)[i];
    for (uint32_t i = threadIdx.x; FPT_LDS && i < kScan2FptSize / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(fpt)[i] = reinterpret_cast<const uint32_t*>(P.fpt)[i];
    for (uint32_t i = threadIdx.x; !FPT_LDS && i < bloom_bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(fpt)[i] = P.s5_bloom[i];
    for (uint32_t i = threadIdx.x; i < P.shorts_words; i += blockDim.x) lrec[i] = P.shorts_packed[i];
    if (threadIdx.x == 0) { wg_next[0] = kScan5Waves; wg_next[1] = wg_next[2] = wg_next[3] = 0; }
    __syncthreads();
    if (DBG && (P.dbg & 128)) return;                            // timing study: launch + table staging alone

    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per-wave LDS region: [fifo: fifo_cap x 4 B][window keys of the first survivors: kScan5SurvX x 4 B][candidate list: cand_cap x 2 B]
    uint8_t* wave_lds = wave_lds_all + (size_t)wave * (P.s5_fifo_cap * 4 + kScan5SurvX * 4 + ((P.cand_cap * 2 + 15) & ~15u));
    uint32_t* fifo = reinterpret_cast<uint32_t*>(wave_lds);
    uint32_t* survx = fifo + P.s5_fifo_cap;
    uint16_t* cand = reinterpret_cast<uint16_t*>(survx + kScan5SurvX);
    const uint32_t G = __builtin_amdgcn_readfirstlane(P.s5_G);
    const uint32_t kp = __builtin_amdgcn_readfirstlane(P.kp), kp2 = __builtin_amdgcn_readfirstlane(kp * kp);
    lds_u8* lgrp = (lds_u8*)0;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const u32x2 lds_u64;
    lds_u64* ldual = (lds_u64*)768;
    lds_u8* lsg = (lds_u8*)512;
    if ((uint32_t)(uintptr_t)(lds_u8*)smem != 0) __builtin_trap();   // see lds_u8

    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = DBG ? clock64() : 0;
    auto mark = [&](int ph) {
        if (DBG && (P.dbg & 64)) { const unsigned long long now = clock64(); tl[ph] += now - tprev; tprev = now; }
    };
    // (the slabs that every wave of the grid owns from the start: the cursor counts what is taken behind them)
    auto static_slabs = [&]() { return (uint64_t)gridDim.x * kScan5Waves * KARG(slab); };
    uint64_t slab_next = ((uint64_t)blockIdx.x * kScan5Waves + wave) * KARG(slab);   // wave-uniform
    uint32_t wave_matches = 0;                                    // (a wave's share of a launch stays far below 2^32)
    bool told_nonascii = false;
    uint32_t slab_left = KARG(slab);

    // work distribution as in gft_scan2.hip: the workgroup owns the units b * waves + k * (grid * waves) + [0, waves) of
    // every round k, its waves take them one by one from a counter in LDS
    constexpr uint32_t wg_waves = kScan5Waves;                    // (a constant: item / wg_waves is a shift, not a division per unit)
    const uint64_t round_units = (uint64_t)gridDim.x * wg_waves, wg_first = (uint64_t)blockIdx.x * wg_waves;
    auto unit_of = [&](uint32_t item) -> uint64_t { return (uint64_t)(item / wg_waves) * round_units + wg_first + item % wg_waves; };
    uint64_t u = unit_of(wave), nu = 0;                           // wave-uniform
    Unit un_n{0, 0, 0};
    uint64_t abs_n = 0, end_n = 0;                               // the next unit's document: blob offsets of its first byte and of the byte behind it
    if (u < P.n_units) { un_n = P.units[u]; abs_n = P.doc_off[un_n.doc]; end_n = P.doc_off[un_n.doc + 1]; }
    for (; u < P.n_units; u = nu) {
        const Unit un{(uint32_t)__builtin_amdgcn_readfirstlane(un_n.doc), (uint32_t)__builtin_amdgcn_readfirstlane(un_n.lo),
                      (uint32_t)__builtin_amdgcn_readfirstlane(un_n.hi)};
        const uint64_t doc_abs = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(abs_n >> 32)) << 32 |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)abs_n);
        const uint64_t end_v = end_n;                             // (the document's end is read by the fold-safety check only: kept in its vector register)
        {
            uint32_t item = 0;
            if (lane == 0) item = __hip_atomic_fetch_add(wg_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nu = unit_of((uint32_t)__builtin_amdgcn_readfirstlane(item));
        }
        mark(7);
        const bool more_units = nu < P.n_units;
        if (more_units) un_n = P.units[nu];
        const Ctx c{P, cls, nullptr, P.short3_bytes ? short3 : nullptr, fpt, lrec, P.text + doc_abs, doc_abs, kp2,
                    doc_abs < 7, doc_abs < 23, un.lo, un.hi, doc_abs + un.hi + 4 > P.text_bytes, DBG ? P.dbg : 0u};
        const uint32_t nborder = un.lo < kScan2MaxOff ? un.lo : kScan2MaxOff;
        const uint32_t ubase = un.lo - kScan2MaxOff;               // candidate lists hold p - ubase (may wrap; p never does)
        const uint32_t own = un.hi - un.lo;
        const uint32_t C = ((own + 63) / 64 + 3) & ~3u;            // bytes per lane (multiple of 4, <= 128)
        const uint32_t my_lo = un.lo + lane * C;
        const uint32_t my_hi = my_lo + C < un.hi ? my_lo + C : un.hi;
        const uint32_t nvalid = my_lo < un.hi ? my_hi - my_lo : 0;
        Ctx5 o{fifo, P.s5_fifo_cap, 0, 0, false, false, 0, P.s5_term_bits, un.lo - P.s5_pos_bias};

        // ---- FILTER -----------------------------------------------------------------------------------------------------
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (own) {
            // the groups of the three bytes in front of the lane's range (the pad group in front of the document)
            uint32_t h1 = P.s5_pad_g, h2 = P.s5_pad_g, h3 = P.s5_pad_g, pq = 0;      // g[j-1], g[j-2], g[j-3]; pq = h2 * G + h1
            (void)h2;
            const uint8_t* src = c.dbase + my_lo;
            U128u nxt{0, 0, 0, 0};
            if (nvalid) {
                uint32_t hist = 0;
                if (doc_abs + my_lo >= 4) hist = load_u32_unaligned(src - 4);
                else for (uint32_t i = 1; i <= 3 && i <= doc_abs + my_lo; i++) hist |= (uint32_t)src[-(int)i] << (32 - 8 * i);
                nxt = *reinterpret_cast<const U128u*>(src);
                if (my_lo >= 1) h1 = lgrp[hist >> 24];
                if (my_lo >= 2) h2 = lgrp[(hist >> 16) & 0xFF];
                if (my_lo >= 3) h3 = lgrp[(hist >> 8) & 0xFF];
            }
            pq = mad24s(h2, G, h1);
            mark(0);
            uint32_t acc = 0, njobs = 0, hib = 0;                // hib: OR of the lane's text (a byte >= 0x80 anywhere?)
            const bool want_fold = P.fold && P.nonascii;
            const uint32_t ndw = C >> 2;                         // dwords per lane (wave-uniform, <= 32)
            const uint32_t npieces = (ndw + 3) >> 2;
            for (uint32_t q = 0; q < npieces; q++) {
                const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
                if (q * 16 < nvalid) hib |= (w[0] | w[1]) | (w[2] | w[3]);   // (may take in up to 15 bytes behind the lane's range: conservative)
                // A dictionary over a large alphabet (SG) usually meets text that leaves ASCII: there the pieces that hold a
                // high byte are noted one by one as they pass (a ballot per piece; a sixth of the fold jobs of "every piece of a
                // lane that saw one").  The all-ASCII instantiation keeps the two ORs: the extra loop-carried state cost the
                // headline 1.2 % when both shared the code (round 3)
                if constexpr (SG) {
                    if (want_fold && !told_nonascii)
                        fold_job_push(q * 16 < nvalid && (((w[0] | w[1]) | (w[2] | w[3])) & 0x80808080u) != 0, lane * C + q * 16, cand, P.cand_cap, njobs);
                }
                if (q + 1 < npieces && (q + 1) * 16 < nvalid) nxt = *reinterpret_cast<const U128u*>(src + (q + 1) * 16);
                const uint32_t nd = ndw - 4 * q;                 // dwords of this piece that belong to the lane (>= 1)
                // probe at byte 0 of a dword: 3-gram (h2, h1, c0); the window that ends there has h3 in front, the window that
                // ends at byte 1 has c1 behind.  Probe at byte 2: 3-gram (c0, c1, c2), h1 in front, c3 behind
                auto dword = [&](uint32_t wd) {
                    const uint32_t c0 = lgrp[wd & 0xFF], c1 = lgrp[(wd >> 8) & 0xFF], c2 = lgrp[(wd >> 16) & 0xFF], c3 = lgrp[wd >> 24];
                    const uint32_t xa = mad24s(pq, G, c0);
                    const uint32_t xb = mad24s(mad24s(c0, G, c1), G, c2);
                    const u32x2 fa = ldual[xa], fb = ldual[xb];
                    acc = __builtin_amdgcn_alignbit(fa.x >> h3, acc, 1);
                    acc = __builtin_amdgcn_alignbit(fa.y >> c1, acc, 1);
                    acc = __builtin_amdgcn_alignbit(fb.x >> h1, acc, 1);
                    acc = __builtin_amdgcn_alignbit(fb.y >> c3, acc, 1);
                    pq = mad24s(c2, G, c3);
                    h3 = c1; h2 = c2; h1 = c3;
                };
                if (nd >= 4) {                                   // a whole piece: sixteen lookups, then eight probes in flight together
                    uint32_t cc[19];                             // cc[3 + i] = group of byte i; cc[0..2] = h3, h2, h1
                    cc[0] = h3; cc[1] = h2; cc[2] = h1;
#pragma unroll
                    for (int i = 0; i < 16; i++) cc[3 + i] = lgrp[(w[i >> 2] >> (8 * (i & 3))) & 0xFF];
                    uint32_t xk[8];
                    xk[0] = mad24s(pq, G, cc[3]);
#pragma unroll
                    for (int t = 1; t < 8; t++) xk[t] = mad24s(mad24s(cc[1 + 2 * t], G, cc[2 + 2 * t]), G, cc[3 + 2 * t]);
                    u32x2 fk[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) fk[t] = ldual[xk[t]];
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        acc = __builtin_amdgcn_alignbit(fk[t].x >> cc[2 * t], acc, 1);         // window ends at byte 2t: cc[2t] stands in front
                        acc = __builtin_amdgcn_alignbit(fk[t].y >> cc[4 + 2 * t], acc, 1);     // ... at byte 2t + 1: that byte's group behind
                    }
                    pq = mad24s(cc[17], G, cc[18]);
                    h3 = cc[16]; h2 = cc[17]; h1 = cc[18];
                } else {
                    dword(w[0]);
                    if (nd >= 2) dword(w[1]);
                    if (nd >= 3) dword(w[2]);
                }
                if ((q & 1) && nd >= 4) {                        // 32 positions complete
                    if ((q >> 1) == 0) m0 = acc; else if ((q >> 1) == 1) m1 = acc; else if ((q >> 1) == 2) m2 = acc; else m3 = acc;
                }
            }
            if (ndw & 7) {                                       // the last, partial group of 32 positions
                const uint32_t v = acc >> (32 - 4 * (ndw & 7));
                const uint32_t k = ndw >> 3;
                if (k == 0) m0 = v; else if (k == 1) m1 = v; else if (k == 2) m2 = v; else m3 = v;
            }
            // ASCII folding is not strings.ToLower once the text leaves ASCII (finder.go:140-142): the pieces of the lanes that
            // met a byte >= 0x80 are judged now (gft_foldsafe_dev.hpp; which of a lane's pieces it was is not kept -- text that
            // leaves ASCII is the exception for the dictionaries this kernel serves, the filter loop pays two ORs for it)
            if (!SG && want_fold && !told_nonascii && __any((hib & 0x80808080u) != 0)) {
                for (uint32_t q = 0; q < npieces; q++)
                    fold_job_push((hib & 0x80808080u) != 0 && q * 16 < nvalid, lane * C + q * 16, cand, P.cand_cap, njobs);
            }
            if (njobs && !told_nonascii) {
                // bit 1: a piece breaks the rule; bit 0: more pieces than the list holds -- the host then checks the text itself
                // (judged here and now: dictionaries over few byte classes rarely meet such text, and carrying the pieces
                // through the list build as gft_scan3.hip does costs the all-ASCII case registers)
                uint32_t bits = 1u;
                if (njobs <= P.cand_cap) {
                    FOLD_JOB_VARS(fj_);
                    bits = fold_jobs_begin(P.text, end_v, doc_abs + un.lo, own, un.lo == 0, cand, njobs, FOLD_JOB_PASS(fj_)) || fold_jobs_finish(FOLD_JOB_PASS(fj_)) ? 2u : 0u;
                }
                if (bits) { told_nonascii = true; if (lane == 0) atomicOr(P.nonascii, bits); }
            }
            // positions past the lane's range carry garbage flags
            m0 = nvalid >= 32 ? m0 : (nvalid ? m0 & ((1u << nvalid) - 1) : 0);
            m1 = nvalid >= 64 ? m1 : (nvalid > 32 ? m1 & ((1u << (nvalid - 32)) - 1) : 0);
            m2 = nvalid >= 96 ? m2 : (nvalid > 64 ? m2 & ((1u << (nvalid - 64)) - 1) : 0);
            m3 = nvalid >= 128 ? m3 : (nvalid > 96 ? m3 & ((1u << (nvalid - 96)) - 1) : 0);
        }
        mark(1);
        if (more_units) { abs_n = P.doc_off[un_n.doc]; end_n = P.doc_off[un_n.doc + 1]; }

        if (DBG && P.dbg) {
            if (P.dbg & 2) {
                uint32_t f = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
                for (int s = 32; s; s >>= 1) f += __shfl_xor(f, s, 64);
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters)), (unsigned long long)f);
            }
            if (P.dbg & 1) m0 = m1 = m2 = m3 = 0;
        }

        // ---- VERIFY: balance the flagged positions over the lanes through an LDS candidate list ---------------------------
        const uint32_t f = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3) + (lane == 0 ? nborder : 0);
        const uint32_t fincl = wave_incl_scan(f);
        const uint32_t ftotal = lane_value(fincl, 63);
        for (uint32_t walk = 0; walk < 2 && ftotal; walk++) {
            o.nf = 0; o.npend = 0; o.lost = false;
            for (uint32_t l0 = 0; l0 < 64;) {
                const uint32_t before = l0 ? lane_value(fincl, l0 - 1) : 0;
                const bool fits = lane >= l0 && fincl - before <= P.cand_cap;
                const uint64_t fm = __ballot(fits) >> l0;
                const uint32_t nl = fm == ~0ull >> l0 ? 64 - l0 : (uint32_t)__builtin_ctzll(~fm);   // lanes in this pass (>= 1)
                const uint32_t l1 = l0 + nl;
                const uint32_t ptotal = lane_value(fincl, l1 - 1) - before;
                if (lane >= l0 && lane < l1) {
                    uint32_t wpos = fincl - f - before;
                    if (lane == 0)
                        for (uint32_t i = 0; i < nborder; i++) cand[wpos++] = (uint16_t)(kScan2MaxOff - nborder + i);
                    uint32_t mm[4] = {m0, m1, m2, m3};
                    const uint32_t rel = lane * C + kScan2MaxOff;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t mk = mm[k];
                        while (mk) {
                            const uint32_t i = __builtin_ctz(mk);
                            mk &= mk - 1;
                            cand[wpos++] = (uint16_t)(rel + 32 * k + i);
                        }
                    }
                }
                wave_lds_sync();
                // stage A: every flagged position -> LDS-only decisions; short terms are emitted here, positions that may end a
                // term of length >= 4 are compacted in place to the front of the list (write index <= read index)
                mark(2);
                uint32_t ns = 0;
                bool n_on[kWays5];
                uint32_t n_rel[kWays5];
                Text8 n_tx[kWays5];
                auto fetch = [&](uint32_t i0) {
#pragma unroll
                    for (int q = 0; q < kWays5; q++) {
                        const uint32_t i = i0 + 64 * q + lane;
                        n_on[q] = i < ptotal;
                        n_rel[q] = cand[n_on[q] ? i : 0];
                    }
#pragma unroll
                    for (int q = 0; q < kWays5; q++) n_tx[q] = cand_load(c, ubase + n_rel[q]);
                };
                fetch(0);
                for (uint32_t i0 = 0; i0 < ptotal; i0 += 64 * kWays5) {
                    bool on[kWays5];
                    uint32_t rel[kWays5];
                    Text8 tx[kWays5];
                    Cand k[kWays5];
#pragma unroll
                    for (int q = 0; q < kWays5; q++) { on[q] = n_on[q]; rel[q] = n_rel[q]; tx[q] = n_tx[q]; }
                    if (i0 + 64 * kWays5 < ptotal) fetch(i0 + 64 * kWays5);
#pragma unroll
                    for (int q = 0; q < kWays5; q++) {
                        cand_keys<!SG>(c, ubase + rel[q], tx[q], k[q]);
                        if (SG && P.short3_bytes) {
                            const uint32_t wq = tx[q].w, sG = P.s5_sG;
                            k[q].sid = short3[mad24s(mad24s(lsg[(wq >> 8) & 0xFF], sG, lsg[(wq >> 16) & 0xFF]), sG, lsg[wq >> 24])];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < kWays5; q++) {
                        if (FPT_LDS || !bloom_lg) cand_decide<FPT_LDS>(c, k[q]);
                        else {
                            // large dictionary: the three cells of the fingerprint table are L2 gathers -- only for the positions
                            // whose (window, byte in front) or window alone is some term's, by the Bloom level in LDS
                            const uint32_t* bl = reinterpret_cast<const uint32_t*>(fpt);
                            const uint32_t hg = scan5_bloom_g(k[q].x, (k[q].tw >> 24) & 0xDFu, bloom_lg), hx = scan5_bloom_x(k[q].x, bloom_lg);
                            const uint32_t wg = bl[hg >> 5], wx = bl[hx >> 5];
                            k[q].go_long = false;
                            if (((wg >> (hg & 31)) | (wx >> (hx & 31))) & 1u) cand_decide<false>(c, k[q]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < kWays5; q++)
                        if (i0 + 64 * q < ptotal)                  // (positions in front of the unit: long terms only)
                        {
                            const uint32_t sidq = on[q] && rel[q] >= kScan2MaxOff ? k[q].sid : 0;
                            if (SG) park_short_g(c, o, lsg, ubase, rel[q], sidq); else park_short(c, o, ubase, rel[q], sidq, k[q].x3);
                        }
                    const uint64_t below = (1ull << lane) - 1;
#pragma unroll
                    for (int q = 0; q < kWays5; q++) {
                        const bool keep = on[q] && k[q].go_long;
                        const uint64_t sb = __ballot(keep);
                        if (keep) {
                            const uint32_t at = ns + (uint32_t)__popcll(sb & below);
                            cand[at] = (uint16_t)rel[q];
                            if (at < kScan5SurvX) survx[at] = k[q].x;   // (stage B's first trip then needs no text to name its slots)
                        }
                        ns += (uint32_t)__popcll(sb);
                    }
                }
                wave_lds_sync();
                mark(6);                                         // (stage A's trips end here; what follows until mark(3) is the short-term trips)
                // stage B: the survivors, densely packed over the lanes, go to the L2 bucket table.  The first trip's loads -- the
                // slots its keys name (they came along from stage A) and the text around the positions, the longest single wait
                // of a unit -- are issued in FRONT of the short-term trips, which touch LDS only, and consumed behind them
                static_assert(kScan5SurvX >= 64, "the first stage-B trip takes its keys from survx");
                const bool b_on = lane < ns;
                const uint32_t b_rel = ns ? cand[b_on ? lane : 0] : 0;
                Cand b_k;
                b_k.p = ubase + b_rel; b_k.x3 = 0; b_k.sid = 0; b_k.go_long = true; b_k.x = 0; b_k.tw = 0;
                Slot b_s0{}, b_s1{};
                Text8 b_t8{0, 0};
                Front b_fr{};
                uint32_t b_tl = 0;
                if (ns) {
                    b_k.x = survx[b_on ? lane : 0];
                    b_s0 = slot_load(&P.slots[scan2_pair_slot(b_k.x, 0, P.slot_shift, P.slot_seed)]);
                    b_s1 = slot_load(&P.slots[scan2_pair_slot(b_k.x, 1, P.slot_shift, P.slot_seed)]);
                    if (!c.near24 && !c.near_end) {
                        // the 32 bytes around the position as two 16-byte loads -- text[p-23 .. p-8] and text[p-7 .. p+8] -- instead
                        // of three (window + front 8, front 16, tail 4): one request less per survivor into an L1 whose
                        // pending-request queue is full half of the time
                        const U128u v1 = *reinterpret_cast<const U128u*>(c.dbase + (int64_t)b_k.p - 23);
                        const U128u v2 = *reinterpret_cast<const U128u*>(c.dbase + (int64_t)b_k.p - 7);
                        b_fr.f[4] = v1.x; b_fr.f[3] = v1.y; b_fr.f[2] = v1.z; b_fr.f[1] = v1.w;
                        b_t8.tw = v2.x; b_t8.w = v2.y;
                        b_tl = v2.z;
                    } else {
                        b_t8 = cand_load(c, b_k.p);
                        b_fr = front_load(c, b_k.p, 0);
                        b_tl = tail_load(c, b_k.p);
                    }
                }
                while (o.npend) { if (SG) short_trip_g(c, o, lsg, ubase, o.npend < 64 ? o.npend : 64); else short_trip(c, o, ubase, o.npend < 64 ? o.npend : 64); }
                if (DBG && (P.dbg & 2)) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 2), (unsigned long long)ns); }
                mark(3);
                Deferred dfr;
                dfr.list = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(cand) + ((ns * 2 + 7) & ~7u));
                dfr.cap = (P.cand_cap * 2 - ((ns * 2 + 7) & ~7u)) / 8;
                dfr.n = 0;
                if (ns) {
                    b_k.tw = b_t8.tw;
                    b_fr.f[0] = b_t8.tw;
                    finish_long5(c, o, b_on, b_rel, b_k, b_s0, b_s1, b_fr, b_tl, dfr);
                }
                for (uint32_t i0 = 64; i0 < ns; i0 += 64) {
                    const bool on = i0 + lane < ns;
                    const uint32_t rel = cand[on ? i0 + lane : 0];
                    const uint32_t p = ubase + rel;
                    Cand k;
                    k.p = p; k.x3 = 0; k.sid = 0; k.go_long = true;
                    if (i0 + 64 <= kScan5SurvX) {
                        // the survivors' window keys came along from stage A: slots and text leave together, one round trip
                        k.x = survx[on ? i0 + lane : 0];
                        const Slot s0 = slot_load(&P.slots[scan2_pair_slot(k.x, 0, P.slot_shift, P.slot_seed)]);
                        const Slot s1 = slot_load(&P.slots[scan2_pair_slot(k.x, 1, P.slot_shift, P.slot_seed)]);
                        const Text8 t8 = cand_load(c, p);
                        const Front fr = front_load(c, p, t8.tw);
                        const uint32_t tl5 = tail_load(c, p);
                        k.tw = t8.tw;
                        finish_long5(c, o, on, rel, k, s0, s1, fr, tl5, dfr);
                    } else {
                        const Text8 t8 = cand_load(c, p);
                        const Front fr = front_load(c, p, t8.tw);
                        const uint32_t tl5 = tail_load(c, p);
                        cand_keys<false>(c, p, t8, k);
                        const Slot s0 = slot_load(&P.slots[scan2_pair_slot(k.x, 0, P.slot_shift, P.slot_seed)]);
                        const Slot s1 = slot_load(&P.slots[scan2_pair_slot(k.x, 1, P.slot_shift, P.slot_seed)]);
                        finish_long5(c, o, on, rel, k, s0, s1, fr, tl5, dfr);
                    }
                }
                if (dfr.n) drain_deferred5(c, o, ubase, dfr);
                wave_lds_sync();
                l0 = l1;
                mark(4);
            }
            if (o.direct || !o.lost) break;
            // the unit's matches outgrew the fifo: their number is known now -- walk it again into a region of that size
            {
                const uint32_t nh = o.nf;
                if (nh > slab_left) {
                    const uint32_t want = nh > KARG(slab) ? nh : KARG(slab);
                    uint64_t nb = 0;
                    if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)want);
                    slab_next = static_slabs() + __shfl(nb, 0, 64);
                    slab_left = want;
                }
                o.dbase = slab_next;
                o.direct = true;
                if (slab_next + nh > KARG(pool_cap)) break;        // (beyond the pool: nothing is written, the host runs the batch again)
            }
        }
        if (ftotal) {
            const uint32_t nh = o.nf;
            if (!o.direct && nh > slab_left) {
                const uint32_t want = nh > KARG(slab) ? nh : KARG(slab);
                uint64_t nb = 0;
                if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)want);
                slab_next = static_slabs() + __shfl(nb, 0, 64);
                slab_left = want;
            }
            const uint64_t base = slab_next;
            slab_next += nh;
            slab_left -= nh;
            wave_matches += nh;
            const bool room = base + nh <= KARG(pool_cap);
            if (lane == 0) { KARG(unit_start)[u] = base; KARG(unit_count)[u] = room ? nh : 0u; }
            if (room && !o.direct) {
                const uint32_t tmask = (1u << o.term_bits) - 1u;
                for (uint32_t i = lane; i < nh; i += 64) {
                    const uint32_t e = fifo[i];
                    // (streaming stores: the pool is read by the NEXT kernel, its lines should not push the units' text out of L2)
                    if (P.want_pos) {
                        __builtin_nontemporal_store(e & tmask, &KARG(pool_term)[base + i]);
                        __builtin_nontemporal_store(o.pos_base + (e >> o.term_bits), &KARG(pool_pos)[base + i]);
                    } else {
                        __builtin_nontemporal_store(e, &KARG(pool_term)[base + i]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            mark(5);
        } else if (lane == 0) {
            KARG(unit_start)[u] = slab_next; KARG(unit_count)[u] = 0;
        }
    }
    if (lane == 0) {
        if (wave_matches)
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(wg_next + 2), (unsigned long long)wave_matches, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t done = __hip_atomic_fetch_add(wg_next + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) & 0x7FFFFFFFu;
        if (done + 1 == kScan5Waves) {
            const unsigned long long all = __hip_atomic_load(reinterpret_cast<unsigned long long*>(wg_next + 2), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP);
            if (all) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(n_matches)), all);
        }
    }
    if (DBG && (P.dbg & 64) && lane == 0 && KARG(dbg_counters)) {
        unsigned long long all = 0;
        for (int ph = 0; ph < 8; ph++) { atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 4 + ph), tl[ph]); all += tl[ph]; }
        atomicMax(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 12), all);
        atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 13), 1ull);
    }
}

}  // namespace

static size_t scan5_fixed_lds(uint32_t dual_entries, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes) {
    return ((768 + (size_t)dual_entries * 8 + short3_bytes + fpt_lds_bytes + (size_t)shorts_words * 4 + 15) & ~(size_t)15) + 16;
}
static size_t scan5_wave_lds(uint32_t fifo_cap, uint32_t cand_cap) {
    return (size_t)fifo_cap * 4 + kScan5SurvX * 4 + (((size_t)cand_cap * 2 + 15) & ~(size_t)15);
}

bool scan5_plan(uint32_t kp, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max, uint32_t fifo_cap, Scan5Plan* out) {
    // as many filter groups as fit next to a candidate list of kScan5CandCapMin entries (every merged class flags more
    // positions: tools/sim); what is left goes to the list
    const uint32_t g_hi = kp < kScan5MaxGroups ? kp : kScan5MaxGroups;
    const uint32_t g_lo = g_hi > 8 + 12 ? g_hi - 8 : (g_hi < 12 ? g_hi : 12);
    for (uint32_t G = g_hi; G >= g_lo; G--) {
        const uint32_t ent = G * G * G;
        const size_t fixed = scan5_fixed_lds(ent, short3_bytes, shorts_words, fpt_lds_bytes);
        const size_t need = fixed + (size_t)kScan5Waves * scan5_wave_lds(fifo_cap, kScan5CandCapMin);
        if (need > lds_max) continue;
        const size_t spare = ((lds_max - need) / kScan5Waves / 2) & ~(size_t)7;      // entries the candidate list can grow by
        out->G = G; out->dual_entries = ent;
        out->cand_cap = (uint32_t)std::min<size_t>(kScan5CandCapMin + spare, 2048);
        out->fifo_cap = fifo_cap;
        return true;
    }
    return false;
}

hipError_t launch_scan5(const Scan2Params& P, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const bool fl = P.fpt_lg == 0;
    const size_t lds = scan5_fixed_lds(P.s5_dual, P.short3_bytes, P.shorts_words, fl ? kScan2FptSize : P.s5_bloom_lg ? (1u << P.s5_bloom_lg) / 8 : 0) +
                       (size_t)kScan5Waves * scan5_wave_lds(P.s5_fifo_cap, P.cand_cap);
    using Kern = void (*)(const Scan2Params);
    // (the timing-study instantiations exist for the direct short-term table only)
#ifdef GFT_S5_SG_CLOCKS       // (study builds, tools/build_variant.sh: the phase clocks of the large-alphabet instantiation)
    const Kern fn = P.s5_sG ? (P.dbg && fl ? k_scan5<true, true, true> : fl ? k_scan5<true, false, true> : k_scan5<false, false, true>)
#else
    const Kern fn = P.s5_sG ? (fl ? k_scan5<true, false, true> : k_scan5<false, false, true>)
#endif
                    : P.dbg ? (fl ? k_scan5<true, true, false> : k_scan5<false, true, false>) : (fl ? k_scan5<true, false, false> : k_scan5<false, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t g = (P.n_units + kScan5Waves - 1) / kScan5Waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    fn<<<dim3(grid), dim3(kScan5Waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
