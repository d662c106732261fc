// gft_scan5.hip -- the suffix-window Aho-Corasick scan with the unit's text resident in LDS (tables: scan2_tables.hpp,
// build_scan5_tables).  Replaces (*Matcher).MatchAll behind CloudflareForkEngine.FindSubstrings
// (finder/substringEngine.go:110-119), as gft_scan2.hip does; same bucket / fingerprint / short-term tables, same results.
//
// What gft_scan2.hip pays for outside its instruction stream (profiles/r3_tcc_counters.json): the
// verification stages read the text around every flagged position again from global memory, 6-10 gathers per unit whose
// 32 lines have by then half left the XCD's L2 (16 line re-fetches per 4 KB document, 2 GB per launch), and every such
// gather is a dependent round trip in front of the LDS decisions.  Here the text is read from HBM exactly once:
//   FILTER   coalesced rounds of 1 KiB -- lane k owns bytes [1024 r + 16 k, +16) of round r, one 16-byte load per lane and
//            round, every line requested by exactly one instruction -- and each piece goes to the wave's text buffer in
//            LDS on its way through the filter.  The room comes from the filter itself: byte classes are merged down to G
//            filter GROUPS (the rare classes share), G^4 bits instead of K^4 (27 -> 22 classes: 66 -> 29 KB for 2 % more
//            flagged positions), and from the short-term table as a bitmap with ranks (K^3 bits + a byte per set cell
//            instead of K^3 bytes).  The keys of the verification stages stay EXACT classes, so everything behind the
//            filter (fingerprints, bucket slots, short records) is gft_scan2's.
//   STAGE A / B  as in gft_scan2.hip, but every text access is an (unaligned) LDS read: window + front bytes ds_read_b64,
//            the 16 bytes in front ds_read_b128, the tail ds_read_b32.  No vector-memory access in stage A at all, stage B
//            waits for its bucket slots only.
//   PREFETCH with stage A free of vector memory the next unit's first round can be requested in front of it: by the time
//            stage B waits for its slots (results return in order) that load has long landed.
//   OUTPUT   matches go to a 4-byte-per-entry LDS fifo (term id, or term id | relative position << term_bits), flushed
//            coalesced into the wave's slab.  A unit whose matches outgrow the fifo is verified a second time with the
//            appends going straight to a pool region of the counted size.
// HBM traffic: text once + 4 B (8 B with positions) per match.  No MFMA (byte automaton, not a contraction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gft_kernels.hpp"

namespace gft {

namespace {

#include "gft_scan2_dev.hpp"

typedef __attribute__((address_space(3))) uint8_t lds_wb;
// (un)aligned LDS accesses by address: gfx950 reads LDS at any alignment (ds_read_b64 / ds_read_b128 / ds_read_b32)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 __attribute__((aligned(1))) u32x2_u;
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef uint32_t __attribute__((aligned(1))) u32_u;
__device__ __forceinline__ void lds_store16(uint32_t addr, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    *(__attribute__((address_space(3))) u32x4*)(uintptr_t)addr = u32x4{x, y, z, w};
}

// the wave's text buffer: LDS byte address of document position p is tq + p (tq = buffer + kScan5Lead - unit.lo, mod 2^32)
__device__ __forceinline__ Text8 text8(uint32_t tq, uint32_t p) {
    const u32x2 v = *(__attribute__((address_space(3))) const u32x2_u*)(uintptr_t)(tq + p - 7);
    return Text8{v.x, v.y};
}
__device__ __forceinline__ Front front5(uint32_t tq, uint32_t p, uint32_t tw) {
    const u32x4 v = *(__attribute__((address_space(3))) const u32x4_u*)(uintptr_t)(tq + p - 23);
    Front t;
    t.f[0] = tw; t.f[4] = v.x; t.f[3] = v.y; t.f[2] = v.z; t.f[1] = v.w;
    return t;
}
__device__ __forceinline__ uint32_t tail5(uint32_t tq, uint32_t p) {
    return *(__attribute__((address_space(3))) const u32_u*)(uintptr_t)(tq + p + 1);
}

struct Ctx5 {
    uint32_t tq;                 // see text8
    const uint2* s3cell;         // LDS: {bits of 32 consecutive 3-windows, rank of the first} (nullptr: no short terms)
    const uint8_t* s3ids;        // LDS: record id per set bit, in rank order
    uint32_t* fifo;              // LDS
    uint32_t fifo_cap;
    uint32_t nf;                 // matches of this unit so far (wave-uniform)
    bool direct;                 // wave-uniform: the second walk of a unit that outgrew the fifo -- appends go to the pool
    uint64_t dbase;              // ... at this entry
    uint32_t term_bits, pos_base;
};

__device__ __forceinline__ void out_append(const Scan2Params& P, Ctx5& o, bool em, uint32_t term, uint32_t pos) {
    const uint64_t mask = __ballot(em);
    if (em) {
        const uint32_t idx = o.nf + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (!o.direct) {
            if (idx < o.fifo_cap) o.fifo[idx] = P.want_pos ? term | (pos - o.pos_base) << o.term_bits : term;
        } else {
            KARG(pool_term)[o.dbase + idx] = term;
            if (P.want_pos) KARG(pool_pos)[o.dbase + idx] = pos;
        }
    }
    o.nf += (uint32_t)__popcll(mask);
}

// record id of the short terms that end with 3-window x3 (0: none); wave-uniform call
__device__ __forceinline__ uint32_t short_id(const Ctx5& o, uint32_t x3) {
    if (!o.s3cell) return 0;
    const uint2 cell = o.s3cell[x3 >> 5];
    const bool hit = (cell.x >> (x3 & 31)) & 1u;
    uint32_t sid = 0;
    if (__any(hit)) {
        const uint32_t rank = cell.y + __popc(cell.x & ((1u << (x3 & 31)) - 1u));
        sid = o.s3ids[hit ? rank : 0];
        if (!hit) sid = 0;
    }
    return sid;
}

template <bool WANT_SID>
__device__ __forceinline__ void cand_keys5(const Ctx& c, const Ctx5& o, uint32_t p, const Text8 t, Cand& k) {
    cand_keys<false>(c, p, t, k);
    k.sid = WANT_SID ? short_id(o, k.x3) : 0;
}

__device__ __forceinline__ void finish_short5(const Ctx& c, Ctx5& o, uint32_t p, uint32_t sid, uint32_t x3) {
    if (!__any(sid != 0)) return;
    uint32_t r[3] = {0, 0, 0};
    if (sid) short_record(c, sid, x3, r);
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        if (j && !__any(r[j] != 0)) break;
        const uint32_t L = r[j] >> 28;
        out_append(c.P, o, r[j] != 0 && L <= p + 1, r[j] & 0x0FFFFFFFu, c.P.pos_end ? p : p + 1 - L);
    }
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

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
        const Text8 t8 = text8(o.tq, p);
        Front t = front5(o.tq, p, t8.tw);
        const uint32_t tl = tail5(o.tq, p);
        const uint32_t kmax = wave_kmax(on ? e.a.z & kScan2LenMask : 0);
        uint32_t folded = 0;
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = on && entry_ok(c, p, t, tl, e, kmax);
        out_append(P, o, ok, e.a.y, match_pos(P, p, e.a.z));
    }
    d.n = 0;
    __builtin_amdgcn_wave_barrier();
}

// FPT_LDS: the fingerprint table is staged in LDS; DBG: the timing-study instantiation (GFT_SCAN_DEBUG)
template <bool FPT_LDS, bool DBG>
__global__ void __launch_bounds__(kScan2Threads) k_scan5(const Scan2Params P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* grp = smem;                                          // byte -> filter group
    uint8_t* cls = smem + 256;                                    // byte -> exact class
    uint32_t* filt = reinterpret_cast<uint32_t*>(smem + 512);
    uint2* s3cell = reinterpret_cast<uint2*>(smem + 512 + (size_t)P.filter_words * 4);   // (filter_words is even: 8-byte aligned)
    uint8_t* s3ids = reinterpret_cast<uint8_t*>(s3cell + P.s5_cells);
    uint8_t* fpt = s3ids + ((P.s5_ids + 3) & ~3u);
    uint32_t* lrec = reinterpret_cast<uint32_t*>(fpt + (FPT_LDS ? kScan2FptSize : 0));
    uint32_t* wg_next = reinterpret_cast<uint32_t*>(smem + (((size_t)(reinterpret_cast<uint8_t*>(lrec) - smem) + P.shorts_words * 4 + 15) & ~(size_t)15));
    uint8_t* wave_lds_all = reinterpret_cast<uint8_t*>(wg_next) + 16;

    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) { grp[i] = P.s5_grp[i]; cls[i] = P.cls[i]; }
    for (uint32_t i = threadIdx.x; i < P.filter_words; i += blockDim.x) filt[i] = P.filter[i];
    for (uint32_t i = threadIdx.x; i < P.s5_cells; i += blockDim.x) reinterpret_cast<uint64_t*>(s3cell)[i] = P.s5_cell[i];
    for (uint32_t i = threadIdx.x; i < P.s5_ids; i += blockDim.x) s3ids[i] = P.s5_id[i];
    for (uint32_t i = threadIdx.x; FPT_LDS && i < kScan2FptSize / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(fpt)[i] = reinterpret_cast<const uint32_t*>(P.fpt)[i];
    for (uint32_t i = threadIdx.x; i < P.shorts_words; i += blockDim.x) lrec[i] = P.shorts_packed[i];
    if (threadIdx.x == 0) { wg_next[0] = blockDim.x >> 6; wg_next[1] = wg_next[2] = wg_next[3] = 0; }
    __syncthreads();

    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per-wave LDS region: [text: kScan5Lead + text_cap + 16][fifo: fifo_cap x 4 B][candidate list: cand_cap x 2 B]
    const uint32_t text_bytes_lds = kScan5Lead + P.s5_text_cap + 16;
    uint8_t* wave_lds = wave_lds_all + (size_t)wave * (text_bytes_lds + P.s5_fifo_cap * 4 + ((P.cand_cap * 2 + 15) & ~15u));
    uint8_t* tbuf = wave_lds;
    uint32_t* fifo = reinterpret_cast<uint32_t*>(wave_lds + text_bytes_lds);
    uint16_t* cand = reinterpret_cast<uint16_t*>(wave_lds + text_bytes_lds + P.s5_fifo_cap * 4);
    const uint32_t tb = (uint32_t)(uintptr_t)(lds_wb*)tbuf;       // LDS byte address of the text buffer
    const uint32_t G = __builtin_amdgcn_readfirstlane(P.s5_G), G2 = __builtin_amdgcn_readfirstlane(G * G);
    const uint32_t kp = __builtin_amdgcn_readfirstlane(P.kp), kp2 = __builtin_amdgcn_readfirstlane(kp * kp);
    lds_u8* lgrp = (lds_u8*)0;
    lds_u32* lfilt = (lds_u32*)512;
    if ((uint32_t)(uintptr_t)(lds_u8*)smem != 0) __builtin_trap();   // see lds_u8

    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = DBG ? clock64() : 0;
    auto mark = [&](int ph) {
        if (DBG && (P.dbg & 64)) { const unsigned long long now = clock64(); tl[ph] += now - tprev; tprev = now; }
    };
    const uint64_t static_slabs = (uint64_t)gridDim.x * (blockDim.x >> 6) * KARG(slab);
    uint64_t slab_next = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave) * KARG(slab), wave_matches = 0;   // wave-uniform
    bool told_nonascii = false;
    uint32_t slab_left = KARG(slab);

    // work distribution as in gft_scan2.hip: the workgroup owns the units b * waves + k * (grid * waves) + [0, waves) of
    // every round k, its waves take them one by one from a counter in LDS
    const uint32_t wg_waves = blockDim.x >> 6;
    const uint64_t round_units = (uint64_t)gridDim.x * wg_waves, wg_first = (uint64_t)blockIdx.x * wg_waves;
    auto unit_of = [&](uint32_t item) { return (uint64_t)(item / wg_waves) * round_units + wg_first + item % wg_waves; };
    uint64_t u = wg_first + wave, nu = 0;                         // wave-uniform
    Unit un_n{0, 0, 0};
    uint64_t abs_n = 0;
    if (u < P.n_units) { un_n = P.units[u]; abs_n = P.doc_off[un_n.doc]; }
    // the first round of the next unit, requested in front of stage A (see the file comment); `have_pf` = pf_* belong to
    // the unit that the next iteration works on
    U128u pf_piece{0, 0, 0, 0}, pf_lead{0, 0, 0, 0};
    uint32_t pf_hist = 0;
    bool have_pf = false;
    for (; u < P.n_units; u = nu) {
        const Unit un{(uint32_t)__builtin_amdgcn_readfirstlane(un_n.doc), (uint32_t)__builtin_amdgcn_readfirstlane(un_n.lo),
                      (uint32_t)__builtin_amdgcn_readfirstlane(un_n.hi)};
        const uint64_t doc_abs = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(abs_n >> 32)) << 32 |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)abs_n);
        {
            uint32_t item = 0;
            if (lane == 0) item = __hip_atomic_fetch_add(wg_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nu = unit_of((uint32_t)__builtin_amdgcn_readfirstlane(item));
        }
        mark(7);
        const bool more_units = nu < P.n_units;
        if (more_units) un_n = P.units[nu];
        const Ctx c{P, cls, filt, nullptr, fpt, lrec, P.text + doc_abs, doc_abs, kp2,
                    false, false, un.lo, un.hi, false, DBG ? P.dbg : 0u};
        const uint32_t nborder = un.lo < kScan2MaxOff ? un.lo : kScan2MaxOff;
        const uint32_t ubase = un.lo - kScan2MaxOff;               // candidate lists hold p - ubase
        const uint32_t own = un.hi - un.lo;                        // <= s5_text_cap (the unit table is built with that limit)
        const uint32_t nr = (own + 1023) >> 10;                    // rounds
        Ctx5 o{tb + kScan5Lead - un.lo, P.s5_cells ? s3cell : nullptr, s3ids, fifo, P.s5_fifo_cap, 0, false, 0,
               P.s5_term_bits, un.lo - P.s5_pos_bias};

        // ---- FILTER -----------------------------------------------------------------------------------------------------
        if (P.prio) __builtin_amdgcn_s_setprio(0);
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        if (own) {
            const uint8_t* ubeg = c.dbase + un.lo;
            const uint8_t* src = ubeg + lane * 16;
            const uint64_t ab = doc_abs + un.lo;                   // blob offset of the unit's first byte
            U128u nxt{0, 0, 0, 0}, lead{0, 0, 0, 0};
            uint32_t hist = 0;
            if (have_pf) { nxt = pf_piece; lead = pf_lead; hist = pf_hist; }
            else {
                if (lane * 16 < own) nxt = *reinterpret_cast<const U128u*>(src);
                // the kScan5Lead bytes in front of the unit (the bytes of the document before it, or of the previous document:
                // gft_scan2_dev.hpp cand_load; zeros in front of the blob) and, once more, the last four of them for the
                // window that rolls into the unit
                if (ab >= kScan5Lead) {
                    if (lane < kScan5Lead / 16) lead = *reinterpret_cast<const U128u*>(ubeg - kScan5Lead + lane * 16);
                    hist = load_u32_unaligned(ubeg - 4);
                } else {
                    uint32_t lw[4] = {0, 0, 0, 0};
                    if (lane < kScan5Lead / 16)
                        for (uint32_t i = 0; i < 16; i++) {
                            const uint32_t back = kScan5Lead - (lane * 16 + i);      // this byte sits `back` bytes in front of the unit
                            if (back <= ab) lw[i >> 2] |= (uint32_t)ubeg[-(int)back] << (8 * (i & 3));
                        }
                    lead = U128u{lw[0], lw[1], lw[2], lw[3]};
                    for (uint32_t i = 1; i <= 3 && i <= ab; i++) hist |= (uint32_t)ubeg[-(int)i] << (32 - 8 * i);
                }
            }
            have_pf = false;
            if (lane < kScan5Lead / 16) lds_store16(tb + lane * 16, lead.x, lead.y, lead.z, lead.w);
            // the window that rolls into lane 0's first piece: groups of the three bytes in front of the unit (the pad group
            // in front of the blob -- what precedes a document only matters to short terms, whose filter bits have every
            // group in front of them)
            const uint32_t pad_g = P.s5_pad_g;
            const uint32_t k1 = ab >= 1 ? lgrp[hist >> 24] : pad_g, k2 = ab >= 2 ? lgrp[(hist >> 16) & 0xFF] : pad_g,
                           k3 = ab >= 3 ? lgrp[(hist >> 8) & 0xFF] : pad_g;
            uint32_t carry_g = __builtin_amdgcn_readfirstlane(k1);
            uint32_t carry_p1 = __builtin_amdgcn_readfirstlane(mad24s(k2, G, k1));      // pair(lo-1)
            uint32_t carry_p2 = __builtin_amdgcn_readfirstlane(mad24s(k3, G, k2));      // pair(lo-2)
            mark(0);
            uint32_t acc = 0, hib = 0;
            for (uint32_t r = 0; r < nr; r++) {
                const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
                const bool mine = r * 1024 + lane * 16 < own;
                if (mine) hib |= (w[0] | w[1]) | (w[2] | w[3]);   // (up to 15 bytes behind the unit: conservative)
                if (r + 1 < nr && (r + 1) * 1024 + lane * 16 < own) nxt = *reinterpret_cast<const U128u*>(src + (r + 1) * 1024);
                // (the next unit's document offset: its record was requested at the top, the prefetch in front of stage A needs both)
                if (r == 0 && more_units) abs_n = P.doc_off[un_n.doc];
                lds_store16(tb + kScan5Lead + r * 1024 + lane * 16, w[0], w[1], w[2], w[3]);
                uint32_t g[16];
#pragma unroll
                for (int d = 0; d < 4; d++)
#pragma unroll
                    for (int b = 0; b < 4; b++) g[4 * d + b] = lgrp[(w[d] >> (8 * b)) & 0xFF];
                // the group in front of the lane's piece: the previous lane's last one (lane 0: the previous round's lane 63)
                const uint32_t gin = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_g, (int)g[15], 0x138, 0xF, 0xF, false);   // wave_shr:1
                carry_g = __builtin_amdgcn_readlane(g[15], 63);
                uint32_t pr[16];
                pr[0] = mad24s(gin, G, g[0]);
#pragma unroll
                for (int i = 1; i < 16; i++) pr[i] = mad24s(g[i - 1], G, g[i]);
                const uint32_t p1in = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_p1, (int)pr[15], 0x138, 0xF, 0xF, false);
                const uint32_t p2in = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_p2, (int)pr[14], 0x138, 0xF, 0xF, false);
                carry_p1 = __builtin_amdgcn_readlane(pr[15], 63);
                carry_p2 = __builtin_amdgcn_readlane(pr[14], 63);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const uint32_t x = mad24s(i == 0 ? p2in : i == 1 ? p1in : pr[i - 2], G2, pr[i]);
                    const uint32_t fw = lfilt[x >> 5];
                    acc = __builtin_amdgcn_alignbit(fw >> (x & 31), acc, 1);
                }
                if (r == 1) m0 = acc; else if (r == 3) m1 = acc; else if (r == 5) m2 = acc;
            }
            if (nr & 1) { const uint32_t v = acc >> 16; if (nr == 1) m0 = v; else if (nr == 3) m1 = v; else m2 = v; }
            if (P.fold && P.nonascii && !told_nonascii && __any((hib & 0x80808080u) != 0)) {
                told_nonascii = true;
                if (lane == 0 && !(__hip_atomic_fetch_or(wg_next + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 31))
                    atomicOr(P.nonascii, 1u);
            }
            // positions of the last round at or beyond the unit's end carry garbage flags
            {
                const uint32_t lr = nr - 1;
                const int32_t avail = (int32_t)own - (int32_t)(lr * 1024 + lane * 16);
                const uint32_t vr = avail <= 0 ? 0u : avail >= 16 ? 16u : (uint32_t)avail;
                const uint32_t keep = ~((0xFFFFu & ~((1u << vr) - 1u)) << (16 * (lr & 1)));
                if ((lr >> 1) == 0) m0 &= keep; else if ((lr >> 1) == 1) m1 &= keep; else m2 &= keep;
            }
        }
        mark(1);
        if (P.prio) __builtin_amdgcn_s_setprio(1);
        if (!own && more_units) abs_n = P.doc_off[un_n.doc];

        if (DBG && P.dbg) {
            if (P.dbg & 2) {
                uint32_t f = __popc(m0) + __popc(m1) + __popc(m2);
                for (int s = 32; s; s >>= 1) f += __shfl_xor(f, s, 64);
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters)), (unsigned long long)f);
            }
            if (P.dbg & 1) m0 = m1 = m2 = 0;
        }

        // ---- VERIFY: balance the flagged positions over the lanes through an LDS candidate list ---------------------------
        const uint32_t f = __popc(m0) + __popc(m1) + __popc(m2) + (lane == 0 ? nborder : 0);
        const uint32_t fincl = wave_incl_scan(f);
        const uint32_t ftotal = lane_value(fincl, 63);
        wave_lds_sync();                                           // the text buffer is complete
        // the next unit's first round goes out now: stage A below touches LDS only
        if (P.s5_prefetch && more_units) {
            const Unit nn{(uint32_t)__builtin_amdgcn_readfirstlane(un_n.doc), (uint32_t)__builtin_amdgcn_readfirstlane(un_n.lo),
                          (uint32_t)__builtin_amdgcn_readfirstlane(un_n.hi)};
            const uint64_t nabs = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(abs_n >> 32)) << 32 |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)abs_n);
            const uint64_t nab = nabs + nn.lo;
            if (nn.hi > nn.lo && nab >= kScan5Lead) {              // (a unit at the very start of the blob loads its own bytes)
                const uint8_t* nbeg = P.text + nab;
                pf_piece = U128u{0, 0, 0, 0};
                if (lane * 16 < nn.hi - nn.lo) pf_piece = *reinterpret_cast<const U128u*>(nbeg + lane * 16);
                pf_lead = U128u{0, 0, 0, 0};
                if (lane < kScan5Lead / 16) pf_lead = *reinterpret_cast<const U128u*>(nbeg - kScan5Lead + lane * 16);
                pf_hist = load_u32_unaligned(nbeg - 4);
                have_pf = true;
            }
        }
        for (uint32_t walk = 0; walk < 2 && ftotal; walk++) {
            o.nf = 0;
            for (uint32_t l0 = 0; l0 < 64;) {
                const uint32_t before = l0 ? lane_value(fincl, l0 - 1) : 0;
                const bool fits = lane >= l0 && fincl - before <= P.cand_cap;
                const uint64_t fm = __ballot(fits) >> l0;
                const uint32_t nl = fm == ~0ull >> l0 ? 64 - l0 : (uint32_t)__builtin_ctzll(~fm);   // lanes in this pass (>= 1)
                const uint32_t l1 = l0 + nl;
                const uint32_t ptotal = lane_value(fincl, l1 - 1) - before;
                if (lane >= l0 && lane < l1) {
                    uint32_t wpos = fincl - f - before;
                    const uint32_t rel = lane * 16 + kScan2MaxOff;
                    if (lane == 0)
                        for (uint32_t i = 0; i < nborder; i++) cand[wpos++] = (uint16_t)(kScan2MaxOff - nborder + i);
                    uint32_t mm[3] = {m0, m1, m2};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        uint32_t mk = mm[k];
                        while (mk) {
                            const uint32_t i = __builtin_ctz(mk);
                            mk &= mk - 1;
                            // bit i of mask k: round 2 k + (i >> 4), byte i & 15 of the lane's piece
                            cand[wpos++] = (uint16_t)(rel + 2048 * k + ((i & 16) << 6) + (i & 15));
                        }
                    }
                }
                wave_lds_sync();
                // stage A: every flagged position -> LDS-only decisions; short terms are emitted here, positions that may end a
                // term of length >= 4 are compacted in place to the front of the list (write index <= read index)
                mark(2);
                if (P.prio) __builtin_amdgcn_s_setprio(2);
                uint32_t ns = 0;
                bool n_on[kStageAWays];
                uint32_t n_rel[kStageAWays];
                Text8 n_tx[kStageAWays];
                auto fetch = [&](uint32_t i0) {
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        const uint32_t i = i0 + 64 * q + lane;
                        n_on[q] = i < ptotal;
                        n_rel[q] = cand[n_on[q] ? i : 0];
                    }
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) n_tx[q] = text8(o.tq, ubase + n_rel[q]);
                };
                fetch(0);
                for (uint32_t i0 = 0; i0 < ptotal; i0 += 64 * kStageAWays) {
                    bool on[kStageAWays];
                    uint32_t rel[kStageAWays];
                    Text8 tx[kStageAWays];
                    Cand k[kStageAWays];
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) { on[q] = n_on[q]; rel[q] = n_rel[q]; tx[q] = n_tx[q]; }
                    if (i0 + 64 * kStageAWays < ptotal) fetch(i0 + 64 * kStageAWays);
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) cand_keys5<true>(c, o, ubase + rel[q], tx[q], k[q]);
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) cand_decide<FPT_LDS>(c, k[q]);
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++)
                        if (i0 + 64 * q < ptotal)                  // (positions in front of the unit: long terms only)
                            finish_short5(c, o, k[q].p, on[q] && rel[q] >= kScan2MaxOff ? k[q].sid : 0, k[q].x3);
                    const uint64_t below = (1ull << lane) - 1;
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        const bool keep = on[q] && k[q].go_long;
                        const uint64_t sb = __ballot(keep);
                        if (keep) cand[ns + __popcll(sb & below)] = (uint16_t)rel[q];
                        ns += (uint32_t)__popcll(sb);
                    }
                }
                wave_lds_sync();
                if (DBG && (P.dbg & 2)) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 2), (unsigned long long)ns); }
                mark(3);
                if (P.prio) __builtin_amdgcn_s_setprio(3);
                // stage B: the survivors, densely packed over the lanes, go to the L2 bucket table
                Deferred dfr;
                dfr.list = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(cand) + ((ns * 2 + 7) & ~7u));
                dfr.cap = (P.cand_cap * 2 - ((ns * 2 + 7) & ~7u)) / 8;
                dfr.n = 0;
                for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
                    const bool on = i0 + lane < ns;
                    const uint32_t rel = cand[on ? i0 + lane : 0];
                    const uint32_t p = ubase + rel;
                    const Text8 t8 = text8(o.tq, p);
                    const Front fr = front5(o.tq, p, t8.tw);
                    const uint32_t tl5 = tail5(o.tq, p);
                    Cand k;
                    cand_keys<false>(c, p, t8, k);
                    const Slot s0 = slot_load(&P.slots[scan2_slot_hash(k.x, 0, P.slot_shift, P.slot_seed)]);
                    const Slot s1 = slot_load(&P.slots[scan2_slot_hash(k.x, 1, P.slot_shift, P.slot_seed)]);
                    finish_long5(c, o, on, rel, k, s0, s1, fr, tl5, dfr);
                }
                if (dfr.n) drain_deferred5(c, o, ubase, dfr);
                wave_lds_sync();
                l0 = l1;
                mark(4);
            }
            if (o.direct || o.nf <= o.fifo_cap) break;
            // the unit's matches outgrew the fifo: their number is known now -- walk it again into a region of that size
            {
                const uint32_t nh = o.nf;
                if (nh > slab_left) {
                    const uint32_t want = nh > KARG(slab) ? nh : KARG(slab);
                    uint64_t nb = 0;
                    if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)want);
                    slab_next = static_slabs + __shfl(nb, 0, 64);
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
                slab_next = static_slabs + __shfl(nb, 0, 64);
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
        if (done + 1 == (blockDim.x >> 6)) {
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

static size_t scan5_fixed_lds(uint32_t filter_words, uint32_t cells, uint32_t ids, uint32_t shorts_words, uint32_t fpt_lds_bytes) {
    return ((512 + (size_t)filter_words * 4 + (size_t)cells * 8 + ((ids + 3) & ~3u) + fpt_lds_bytes + (size_t)shorts_words * 4 + 15) & ~(size_t)15) + 16;
}
static size_t scan5_wave_lds(uint32_t text_cap, uint32_t fifo_cap, uint32_t cand_cap) {
    return kScan5Lead + text_cap + 16 + (size_t)fifo_cap * 4 + (((size_t)cand_cap * 2 + 15) & ~(size_t)15);
}

bool scan5_plan(uint32_t kp, uint32_t cells, uint32_t ids, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max, Scan5Plan* out) {
    // whole rounds of text first (a document that does not fit a unit costs a second unit's fixed work), then as many
    // filter groups as fit (every merged class flags more positions: tools/sim)
    const uint32_t g_hi = kp < kScan5MaxGroups ? kp : kScan5MaxGroups;
    const uint32_t g_lo = g_hi > 8 + 12 ? g_hi - 8 : (g_hi < 12 ? g_hi : 12);
    for (uint32_t rounds = kScan5MaxRounds; rounds >= 2; rounds--)
        for (uint32_t G = g_hi; G >= g_lo; G--) {
            const uint64_t bits = (uint64_t)G * G * G * G;
            const uint32_t fw = (uint32_t)((bits + 63) / 64 * 2);
            const size_t fixed = scan5_fixed_lds(fw, cells, ids, shorts_words, fpt_lds_bytes);
            const size_t need = fixed + (size_t)kScan5Waves * scan5_wave_lds(rounds * 1024, kScan2FifoCap, kScan5CandCapMin);
            if (need > lds_max) continue;
            size_t spare = (lds_max - need) / kScan5Waves / 2 & ~(size_t)7;      // entries the candidate list can grow by
            out->G = G; out->rounds = rounds; out->filter_words = fw;
            out->cand_cap = (uint32_t)std::min<size_t>(kScan5CandCapMin + spare, 1024);
            out->fifo_cap = kScan2FifoCap;
            return true;
        }
    return false;
}

hipError_t launch_scan5(const Scan2Params& P, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const bool fl = P.fpt_lg == 0;
    const size_t lds = scan5_fixed_lds(P.filter_words, P.s5_cells, P.s5_ids, P.shorts_words, fl ? kScan2FptSize : 0) +
                       (size_t)kScan5Waves * scan5_wave_lds(P.s5_text_cap, P.s5_fifo_cap, P.cand_cap);
    using Kern = void (*)(const Scan2Params);
    const Kern fn = P.dbg ? (fl ? k_scan5<true, true> : k_scan5<false, true>) : (fl ? k_scan5<true, false> : k_scan5<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t g = (P.n_units + kScan5Waves - 1) / kScan5Waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    fn<<<dim3(grid), dim3(kScan5Waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
