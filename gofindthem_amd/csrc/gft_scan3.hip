// gft_scan3.hip -- the Aho-Corasick scan for gfx950, stride-2 suffix-window form (tables: scan3_tables.hpp).
// Replaces (*Matcher).MatchAll behind CloudflareForkEngine.FindSubstrings (finder/substringEngine.go:110-119).
//
// One wavefront per work unit (a document, or a slice of a long one), 16 waves per workgroup share the LDS tables.
//   FILTER   the unit is read in coalesced rounds of 1 KiB: lane k owns bytes [1024 r + 16 k, +16) of round r (one 16-byte
//            load per lane and round, the next round in flight while this one is filtered).  Bytes map to filter groups
//            through a 256-byte LDS table; every OTHER position p is probed: the 4-group window ending at p indexes a
//            bit table in LDS (one random LDS probe per two text bytes -- the scarce resource, tools/ubench).  The pair
//            in front of a lane's piece comes from its neighbour by DPP, no byte is looked up twice.
//   LIST     flagged probes are listed in LDS (a wave prefix sum over the lanes' flag counts) and dealt densely to lanes.
//   STAGE A  per flagged probe, LDS only: terms of length <= 3 ending at p or p-1 (short3 records, emitted here), and a
//            Bloom cell keyed by (window, group of the byte in front) that decides whether a longer term can be anchored
//            here at all.  Survivors are parked with their window key.
//   STAGE B  survivors, 64 per trip: both candidate 32-byte slots of the key and the text around the window are loaded at
//            once (the key is known: no dependent lookup in front of the slot loads), the slot carries the term's bytes.
//   OUTPUT   lanes that found a match take consecutive cells (ballot + mbcnt) of the slab the wave reserved in the match
//            pool with one global atomic per ~4 K matches: every append is one store instruction to consecutive
//            addresses, nothing is staged in LDS.  A unit whose matches do not fit what is left of the slab is walked a
//            second time into a fresh one (same traversal, same indices).  Nothing is truncated: the host re-runs with
//            a larger pool if the cursor overran.
// HBM traffic: text once + 8 B per match (4 B in presence-only mode); tables are LDS / L2 resident.  No MFMA (byte
// automaton, not a contraction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gft_kernels.hpp"

namespace gft {

namespace {

#include "gft_foldsafe_dev.hpp"

struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U128u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U64u { uint32_t lo, hi; };

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// a * b + c on 24-bit operands with the multiplier in a scalar register: one full-rate instruction
__device__ __forceinline__ uint32_t mad24s(uint32_t a, uint32_t sb, uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sb), "v"(c));
    return d;
}
// LDS tables at fixed addresses (the kernel's only LDS object is the dynamic array, which starts at 0; checked at kernel
// entry): constant bases fold into the ds_read offset field
typedef __attribute__((address_space(3))) const uint8_t lds_u8;
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) { return reinterpret_cast<const U32u*>(p)->v; }

// kernel arguments that are needed once per unit or less are read from the kernarg segment where they are used (see
// gft_scan2.hip: the unit loop keeps more values alive than there are SGPRs)
typedef __attribute__((address_space(4))) const uint8_t karg_u8;
template <class T>
__device__ __forceinline__ T karg_field(uint32_t off) {
    karg_u8* ka = (karg_u8*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(__attribute__((address_space(4))) const T*)(ka + off);
}
#define KARG(field) karg_field<decltype(Scan3Params::field)>((uint32_t)offsetof(Scan3Params, field))

// ASCII lower-casing of four packed bytes (finder/finder.go:140-142 for ASCII text)
__device__ __forceinline__ uint32_t fold4(uint32_t w) {
    const uint32_t h = w & 0x7F7F7F7Fu;
    const uint32_t ge_a = h + 0x3F3F3F3Fu;          // bit 7 set where byte >= 'A'
    const uint32_t gt_z = h + 0x25252525u;          // bit 7 set where byte >  'Z'
    const uint32_t up = ge_a & ~gt_z & ~w & 0x80808080u;
    return w | (up >> 2);
}
__device__ __forceinline__ uint32_t fold1(uint32_t b) { return (b - 'A' < 26u) ? b + 32 : b; }

// inclusive prefix sum over the 64 lanes with DPP moves
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t l) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Ctx {
    const Scan3Params& P;
    const uint8_t* dbase;      // first byte of the document
    uint64_t doc_abs;          // offset of the document inside the text blob
    bool near0;                // wave-uniform: the document starts within 7 bytes of the blob start
    bool near24;               // ... within 23 bytes
    uint32_t lo, hi;           // the unit: a match belongs to the unit that holds its END position
    bool near_end;             // wave-uniform: the unit ends within 8 bytes of the blob end
};

// ---- output.  `nf` (matches of the unit so far) is wave-uniform: every append happens in wave-uniform control flow, lanes
// that have something to append take consecutive cells of the wave's slab (ballot + mbcnt).  Cells beyond `room` are
// counted, not written: the unit is then walked again into a slab that holds all of it ---------------------------------------
struct Out {
    uint32_t* term;            // pool_term + base of the unit
    uint32_t* pos;             // pool_pos + base (nullptr: presence only)
    uint32_t room;             // cells of the slab (and of the pool) this unit may write
    uint32_t nf;
    __device__ __forceinline__ void append(bool em, uint32_t t, uint32_t p) {
        const uint64_t mask = __ballot(em);
        if (em) {
            const uint32_t idx = nf + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (idx < room) {
                term[idx] = t;
                if (pos) pos[idx] = p;
            }
        }
        nf += (uint32_t)__popcll(mask);
    }
};

// ---- text around a probe -------------------------------------------------------------------------------------------------
// one 8-byte load: tw = text[p-7 .. p-4], w = text[p-3 .. p] (the window).  Bytes before the document start (the previous
// document's, or zeros in front of the blob) can only change keys of windows that reach across the start, and every term
// such a window may name is longer than p + 1 and is dropped by the length check at emission.
struct Text8 { uint32_t tw, w; };
__device__ __forceinline__ Text8 cand_load_slow(const Ctx& c, uint32_t p) {   // within 7 bytes of the blob start
    Text8 t{0, 0};
    const uint64_t ab = c.doc_abs + p;
    if (ab >= 7) {
        const U64u v = *reinterpret_cast<const U64u*>(c.dbase + (int64_t)p - 7);
        t.tw = v.lo; t.w = v.hi;
    } else {
        for (uint32_t i = 0; i <= (uint32_t)ab; i++) {        // oldest byte first; byte p ends up on top of w
            t.tw = t.tw >> 8 | t.w << 24;
            t.w = t.w >> 8 | (uint32_t)c.dbase[(int64_t)p - (int64_t)ab + i] << 24;
        }
    }
    return t;
}
__device__ __forceinline__ Text8 cand_load(const Ctx& c, uint32_t p) {
    if (__builtin_expect(c.near0, 0)) return cand_load_slow(c, p);
    const U64u v = *reinterpret_cast<const U64u*>(c.dbase + (int64_t)p - 7);
    return Text8{v.lo, v.hi};
}
// the 20 bytes in front of the window as the slots store them, f[k] = text[p-7-4k .. p-4-4k]
struct Front { uint32_t f[5]; };
__device__ __forceinline__ Front front_load(const Ctx& c, uint32_t p, uint32_t tw) {
    Front t;
    t.f[0] = tw;
    if (__builtin_expect(c.near24, 0)) {                      // wave-uniform: first document of the blob
        const uint64_t ab = c.doc_abs + p;
#pragma unroll
        for (int k = 1; k < 5; k++) {
            uint32_t v = 0;
            for (int b = 0; b < 4; b++)
                if (ab >= (uint64_t)(7 + 4 * k - b)) v |= (uint32_t)c.dbase[(int64_t)p - 7 - 4 * k + b] << (8 * b);
            t.f[k] = v;
        }
    } else {
        const U128u v = *reinterpret_cast<const U128u*>(c.dbase + (int64_t)p - 23);
        t.f[4] = v.x; t.f[3] = v.y; t.f[2] = v.z; t.f[1] = v.w;
    }
    return t;
}
// the (up to) four bytes behind position p, text[p+1 .. p+4]; bytes past the blob end read as zero (a tail that reached
// there would end outside the unit and is dropped by the range check anyway)
__device__ __forceinline__ uint32_t tail_load(const Ctx& c, uint32_t p) {
    if (__builtin_expect(c.near_end, 0)) {
        uint32_t v = 0;
        for (uint32_t b = 0; b < 4; b++)
            if (c.doc_abs + p + 1 + b < c.P.text_bytes) v |= (uint32_t)c.dbase[(uint64_t)p + 1 + b] << (8 * b);
        return v;
    }
    return load_u32_unaligned(c.dbase + (uint64_t)p + 1);
}

// ---- bucket table ---------------------------------------------------------------------------------------------------------
struct Slot { uint4 a, b; };      // a = {key, info, len, front[0]}, b = front[1..4]
__device__ __forceinline__ Slot slot_load(const Scan2Slot* s) {
    const uint4* q = reinterpret_cast<const uint4*>(s);
    return Slot{q[0], q[1]};
}
__device__ __forceinline__ bool slot_pick(uint32_t x, const Slot& s0, const Slot& s1, Slot& out) {
    const bool use1 = s1.a.x == x;
    out.a = use1 ? s1.a : s0.a;
    out.b = use1 ? s1.b : s0.b;
    return use1 || s0.a.x == x;
}
__device__ __forceinline__ int32_t slot_off(uint32_t lw) { return (int32_t)lw >> 24; }          // signed: -1 .. kScan2MaxOff
// reported position of a match whose window ends at p (lw = the slot's len word)
__device__ __forceinline__ uint32_t match_pos(const Scan3Params& P, uint32_t p, uint32_t lw) {
    const uint32_t L1 = lw & kScan2LenMask;
    return P.pos_end ? p + (uint32_t)slot_off(lw) : p + 1 - L1;
}

// does the anchor described by e have its window end at p (and the term its end inside the unit)?  t = the (folded) bytes
// in front of the window, w = the (folded) window, tl = the raw bytes behind it.  kmax: dwords of `front` to look at.
// Slot layout (scan3_tables.cpp make_slot): a = {key, info, len, front[0]}, b = {front[1], front[2], front[3] or tail, window}
__device__ __forceinline__ bool entry_ok(const Ctx& c, uint32_t p, const Front& t, uint32_t w, uint32_t tl, const Slot& e, uint32_t kmax) {
    const Scan3Params& P = c.P;
    const uint32_t L = e.a.z & kScan2LenMask;          // the term up to the end of its window
    const int32_t off = slot_off(e.a.z);
    const int32_t nfront = (int32_t)L - 4;
    // the window's own bytes (the key names their filter groups only); a window with off = -1 holds three term bytes
    uint32_t diff = (w ^ e.b.w) & (off < 0 ? 0x00FFFFFFu : 0xFFFFFFFFu);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((uint32_t)k < kmax) {
            int32_t nb = min(max(nfront - 4 * k, 0), 4);                           // bytes of this dword the term owns
            if (k == 3 && off > 0) nb = 0;                                         // front[3] holds the tail instead
            const uint32_t mask = (uint32_t)(0xFFFFFFFF00000000ull >> (8 * nb));   // ... the ones next to the window
            const uint32_t fk = k == 0 ? e.a.w : k == 1 ? e.b.x : k == 2 ? e.b.y : e.b.z;
            diff |= (t.f[k] ^ fk) & mask;
        }
    }
    if (off > 0) {
        const uint32_t tv = P.fold ? fold4(tl) : tl;
        diff |= (tv ^ e.b.z) & (0xFFFFFFFFu >> (8 * (4 - off)));
    }
    const uint32_t pe = p + (uint32_t)off;                                         // where the term ends
    bool ok = L <= p + 1 && diff == 0 && pe >= c.lo && pe < c.hi;
    const uint32_t inl = off > 0 ? 16u : 20u;                                      // bytes the slot holds (window included)
    if (ok && L > inl) {
        // the first L-inl bytes of the term against text[p+1-L .. p-inl], four bytes at a time from the end; term_blob
        // carries 4 bytes of slack in front of every term, the text side needs 3 bytes of slack before the match
        const uint8_t* tb = KARG(term_blob) + KARG(term_off)[e.a.y];
        const uint8_t* tp = c.dbase + (int64_t)p + 1 - L;
        const uint32_t n = L - inl;
        if (c.doc_abs + p + 1 - L >= 3) {
            uint32_t d2 = 0;
            for (uint32_t j = 0; j * 4 < n; j++) {
                const int32_t at = (int32_t)n - 4 - (int32_t)(j * 4);       // may be -1..-3 for the last chunk
                uint32_t tv = load_u32_unaligned(tp + at);
                const uint32_t wv = load_u32_unaligned(tb + at);
                if (P.fold) tv = fold4(tv);
                const uint32_t mask = at >= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (uint32_t)(-at));
                d2 |= (tv ^ wv) & mask;
            }
            ok = d2 == 0;
        } else {
            for (uint32_t i = 0; i < n && ok; i++) {
                uint32_t b = tp[i];
                if (P.fold) b = fold1(b);
                ok = b == tb[i];
            }
        }
    }
    return ok;
}

// dwords of front bytes a term of length L owns (0 for lanes that are not active)
__device__ __forceinline__ uint32_t wave_kmax(uint32_t L) {
    return __any(L > 16) ? 4 : __any(L > 12) ? 3 : __any(L > 8) ? 2 : 1;
}
__device__ __forceinline__ void front_fold_upto(Front& t, uint32_t& done, uint32_t kmax) {
#pragma unroll
    for (int k = 0; k < 4; k++)
        if ((uint32_t)k >= done && (uint32_t)k < kmax) t.f[k] = fold4(t.f[k]);
    done = kmax > done ? kmax : done;
}

// A wave's deferred bucket entries: {probe entry, index into `more`} pairs parked in LDS so that the entries of
// multi-term buckets are verified densely (64 distinct entries per trip) instead of one round per bucket depth.
struct Deferred { uint2* list; uint32_t cap, n; };

// position of a listed probe: entries 0 and 1 are the border probes lo - 3 and lo - 1, entry t + 2 is probe t at lo + 2 t + 1
__device__ __forceinline__ uint32_t probe_pos(const Ctx& c, uint32_t entry) { return c.lo + 2 * entry - 3; }

// terms of length >= 4 anchored at the lanes' probes (`on`: this lane has a candidate); wave-uniform call
__device__ __forceinline__ void finish_long(const Ctx& c, bool on, uint32_t entry, uint32_t p, uint32_t x, const Slot& s0, const Slot& s1,
                                            Front t, uint32_t w, uint32_t tl, Out& fifo, Deferred& d) {
    const Scan3Params& P = c.P;
    Slot e;
    const bool have = slot_pick(x, s0, s1, e) && on;
    if (!__any(have)) return;
    const bool multi = have && (e.a.y & kScan2Multi);
    uint32_t folded = 0;
    if (P.fold) w = fold4(w);
    {   // one-term buckets
        const bool act = have && !multi;
        const uint32_t kmax = wave_kmax(act ? e.a.z & kScan2LenMask : 0);
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = act && entry_ok(c, p, t, w, tl, e, kmax);
        fifo.append(ok, e.a.y, match_pos(P, p, e.a.z));
    }
    if (!__any(multi)) return;
    const uint32_t n_ent = multi ? e.a.z : 0, more_at = e.a.y & ~kScan2Multi;
    const uint32_t tot = lane_value(wave_incl_scan(n_ent), 63);
    if (tot <= d.cap - d.n) {
        // park every entry: lanes take consecutive cells, entry after entry
        for (uint32_t j = 0; __any(j < n_ent); j++) {
            const uint64_t m = __ballot(j < n_ent);
            if (j < n_ent)
                d.list[d.n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] =
                    make_uint2(entry, more_at + j);
            d.n += (uint32_t)__popcll(m);
        }
        return;
    }
    // no room (a very deep bucket): verify in place, one round per entry
    Slot cur = e;
    if (multi) cur = slot_load(&P.more[more_at]);
    for (uint32_t j = 0; __any(j < n_ent); j++) {
        const bool act = j < n_ent;
        Slot nxt = cur;
        if (j + 1 < n_ent) nxt = slot_load(&P.more[more_at + j + 1]);      // in flight during the compare
        const uint32_t kmax = wave_kmax(act ? cur.a.z & kScan2LenMask : 0);
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = act && entry_ok(c, p, t, w, tl, cur, kmax);
        fifo.append(ok, cur.a.y, match_pos(P, p, cur.a.z));
        cur = nxt;
    }
}

// verify the parked entries, 64 per trip
__device__ __forceinline__ void drain_deferred(const Ctx& c, Out& fifo, Deferred& d) {
    const Scan3Params& P = c.P;
    const uint32_t lane = lane_id();
    wave_lds_fence();
    for (uint32_t i0 = 0; i0 < d.n; i0 += 64) {
        const bool on = i0 + lane < d.n;
        const uint2 it = d.list[on ? i0 + lane : 0];
        const uint32_t p = probe_pos(c, it.x);
        const Slot e = slot_load(&P.more[it.y]);
        const Text8 t8 = cand_load(c, p);
        Front t = front_load(c, p, t8.tw);
        const uint32_t tl = tail_load(c, p);
        const uint32_t kmax = wave_kmax(on ? e.a.z & kScan2LenMask : 0);
        uint32_t folded = 0;
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = on && entry_ok(c, p, t, P.fold ? fold4(t8.w) : t8.w, tl, e, kmax);
        fifo.append(ok, e.a.y, match_pos(P, p, e.a.z));
    }
    d.n = 0;
    __builtin_amdgcn_wave_barrier();
}

// ---- stage B: the parked survivors {probe entry, window key}, at most kScan3SurvCap = two trips of 64 ------------------------
__device__ __forceinline__ void stage_b(const Ctx& c, uint2* surv, uint32_t& ns, Out& fifo) {
    const Scan3Params& P = c.P;
    const uint32_t lane = lane_id();
    if (!ns) return;
    wave_lds_fence();
    uint2 its[2];
#pragma unroll
    for (int h = 0; h < 2; h++) its[h] = surv[64 * h + lane < ns ? 64 * h + lane : 0];
    __builtin_amdgcn_wave_barrier();                    // every lane holds its entries: the list's room parks bucket entries now
    Deferred dfr{surv, kScan3SurvCap, 0};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h && ns <= 64) break;
        const bool on = 64 * h + lane < ns;
        const uint2 it = its[h];
        const uint32_t p = probe_pos(c, it.x);
        const Slot s0 = slot_load(&P.slots[scan2_slot_hash(it.y, 0, P.slot_shift, P.slot_seed)]);
        const Slot s1 = slot_load(&P.slots[scan2_slot_hash(it.y, 1, P.slot_shift, P.slot_seed)]);
        const Text8 t8 = cand_load(c, p);
        const Front fr = front_load(c, p, t8.tw);
        const uint32_t tl = tail_load(c, p);
        finish_long(c, on, it.x, p, it.y, s0, s1, fr, t8.w, tl, fifo, dfr);
    }
    if (dfr.n) drain_deferred(c, fifo, dfr);
    ns = 0;
    wave_lds_fence();
}

// ---- stage A decisions for one listed probe --------------------------------------------------------------------------------
struct CandKeys { uint32_t x, key5, x3p, x3m; };
__device__ __forceinline__ CandKeys cand_keys(lds_u8* lcls, uint32_t G, uint32_t G2, const Text8 t) {
    const uint32_t g0 = lcls[t.tw >> 24], g1 = lcls[t.w & 0xFF], g2 = lcls[(t.w >> 8) & 0xFF], g3 = lcls[(t.w >> 16) & 0xFF],
                   g4 = lcls[t.w >> 24];
    const uint32_t qa = mad24s(g1, G, g2), qb = mad24s(g3, G, g4);
    CandKeys k;
    k.x = mad24s(qa, G2, qb);                 // window text[p-3 .. p]
    k.x3p = mad24s(g2, G2, qb);               // 3-window ending at p
    k.x3m = mad24s(qa, G, g3);                // 3-window ending at p - 1
    k.key5 = mad24s(k.x, G, g0);              // window + the group in front of it
    return k;
}

// ---- stage S: the short jobs parked by stage A, 64 per trip.  A job = probe entry | sid0 << 16 | sid1 << 24: the short3
// cells of the 3-windows ending at p and at p - 1 (0: nothing ends there).  A record holds up to three {term | len << 28,
// bytes}; the bytes are compared only under merged groups (with one byte class per group the 3-window proves them) ------------
typedef uint32_t __attribute__((may_alias)) u32a;
__device__ __forceinline__ void stage_s(const Ctx& c, lds_u8* lcls, lds_u32* lrec, u32a* jobs, uint32_t& nj, Out& fifo) {
    const Scan3Params& P = c.P;
    const uint32_t lane = lane_id();
    if (!nj) return;
    wave_lds_fence();
    for (uint32_t i0 = 0; i0 < nj; i0 += 64) {
        // (lanes past the end of the list work on a copy of job 0's probe with empty cells and stay silent)
        const uint32_t job = i0 + lane < nj ? jobs[i0 + lane] : jobs[0] & 0xFFFFu;
        const uint32_t p = probe_pos(c, job & 0xFFFFu);
        uint32_t sid[2] = {(job >> 16) & 0xFFu, job >> 24};
        uint32_t W[2] = {0, 0};
        const bool big = sid[0] == 255 || sid[1] == 255;
        const bool need_text = P.grouped || __any(big);
        if (need_text) {
            const Text8 t8 = cand_load(c, p);
            W[0] = t8.w;                                                  // text[p-3 .. p]
            W[1] = __builtin_amdgcn_alignbyte(t8.w, t8.tw, 3);           // text[p-4 .. p-1]
            if (P.fold) { W[0] = fold4(W[0]); W[1] = fold4(W[1]); }
            if (__builtin_expect(__any(big), 0)) {
                // records that did not get an LDS id: {n, n x 2 words} in global memory, one entry per round; the cell is
                // found again from the text (rare path)
                const uint32_t G = P.G;
                const uint32_t g1 = lcls[t8.w & 0xFF], g2 = lcls[(t8.w >> 8) & 0xFF], g3 = lcls[(t8.w >> 16) & 0xFF], g4 = lcls[t8.w >> 24];
                const uint32_t x3[2] = {(g2 * G + g3) * G + g4, (g1 * G + g2) * G + g3};
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const bool b = sid[h] == 255;
                    uint32_t n = 0;
                    const uint32_t* g = nullptr;
                    if (b) { g = KARG(srec_big) + KARG(short3_big)[x3[h]]; n = g[0]; }
                    const uint32_t e = p - h;
                    for (uint32_t j = 0; __any(j < n); j++) {
                        uint32_t w0 = 0, w1 = 0;
                        if (j < n) { w0 = g[1 + 2 * j]; w1 = g[2 + 2 * j]; }
                        const uint32_t L = w0 >> 28;
                        const bool ok = w0 != 0 && L <= e + 1 && ((W[h] ^ w1) >> ((32 - 8 * L) & 31)) == 0;
                        fifo.append(ok, w0 & 0x0FFFFFFFu, P.pos_end ? e : e + 1 - L);
                    }
                    if (b) sid[h] = 0;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (!__any(sid[h] != 0)) continue;
            lds_u32* r = lrec + sid[h] * kScan3RecWords;
            const uint32_t e = p - h;
#pragma unroll
            for (uint32_t j = 0; j < 3; j++) {
                const uint32_t w0 = sid[h] ? r[2 * j] : 0;
                if (j && !__any(w0 != 0)) break;
                const uint32_t L = w0 >> 28;
                bool ok = w0 != 0 && L <= e + 1;
                if (P.grouped) ok = ok && ((W[h] ^ r[2 * j + 1]) >> ((32 - 8 * L) & 31)) == 0;
                fifo.append(ok, w0 & 0x0FFFFFFFu, P.pos_end ? e : e + 1 - L);
            }
        }
    }
    nj = 0;
    wave_lds_fence();
}

constexpr int kStageAWays = 2;      // stage A: candidates a lane works on at once

// BLOOM_LDS: the Bloom table is staged in LDS (dictionaries up to ~11 k long terms); otherwise read in place (L2)
// DBG:       timing studies only (GFT_SCAN_DEBUG, wrong results): P.dbg = 1 filter only, 2 + list and stage A decisions
//            without any emission, 3 + short-term emission (no stage B).  Production launches use DBG = false, where
//            none of this is compiled in
template <bool BLOOM_LDS, bool DBG>
__global__ void __launch_bounds__(kScan3Threads) k_scan3(const Scan3Params P) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t off_filt = 256, off_s3 = off_filt + P.filter_words * 4, off_rec = off_s3 + P.short3_bytes,
                   off_bloom = off_rec + ((P.srec_words * 4 + 15) & ~15u),
                   off_next = (off_bloom + (BLOOM_LDS ? 4u << P.bloom_lg : 0u) + 15u) & ~15u,   // the workgroup's bookkeeping (16 B, aligned)
                   off_wave = off_next + 16;
    {
        uint32_t* s32 = reinterpret_cast<uint32_t*>(smem);
        for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) s32[i] = reinterpret_cast<const uint32_t*>(P.cls)[i];
        for (uint32_t i = threadIdx.x; i < P.filter_words; i += blockDim.x) s32[off_filt / 4 + i] = P.filter[i];
        for (uint32_t i = threadIdx.x; i < P.short3_bytes / 4; i += blockDim.x) s32[off_s3 / 4 + i] = reinterpret_cast<const uint32_t*>(P.short3)[i];
        for (uint32_t i = threadIdx.x; i < P.srec_words; i += blockDim.x) s32[off_rec / 4 + i] = P.srec[i];
        for (uint32_t i = threadIdx.x; BLOOM_LDS && i < (1u << P.bloom_lg); i += blockDim.x) s32[off_bloom / 4 + i] = P.bloom[i];
        if (threadIdx.x == 0) {                                      // [0] work counter (every wave starts with the item of its own
            s32[off_next / 4] = blockDim.x >> 6;                     // number), [1] waves done, [2..3] matches
            s32[off_next / 4 + 1] = s32[off_next / 4 + 2] = s32[off_next / 4 + 3] = 0;
        }
    }
    __syncthreads();
    if ((uint32_t)(uintptr_t)(lds_u8*)smem != 0) __builtin_trap();   // see lds_u8

    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t kWaves = blockDim.x >> 6;
    // per-wave LDS region: [survivors: kScan3SurvCap x 8 B][candidate list: cand_cap x 2 B]
    const uint32_t wave_bytes = kScan3SurvCap * 8 + P.cand_cap * 2;
    uint8_t* wave_lds = smem + off_wave + (size_t)wave * wave_bytes;
    uint2* surv = reinterpret_cast<uint2*>(wave_lds);
    uint16_t* cand = reinterpret_cast<uint16_t*>(wave_lds + kScan3SurvCap * 8);
    u32a* jobs = reinterpret_cast<u32a*>(cand);                  // short jobs overlay the part of the list that has been read
    const uint32_t G = __builtin_amdgcn_readfirstlane(P.G), G2 = __builtin_amdgcn_readfirstlane(G * G);
    lds_u8* lcls = (lds_u8*)0;
    lds_u32* lfilt = (lds_u32*)(uintptr_t)off_filt;
    lds_u8* ls3 = (lds_u8*)(uintptr_t)off_s3;
    lds_u32* lrec = (lds_u32*)(uintptr_t)off_rec;
    lds_u32* lbloom = (lds_u32*)(uintptr_t)off_bloom;
    const bool have_short = P.short3_bytes != 0;

    // match pool as in gft_scan2.hip: every wave of the grid owns one slab from the start, further slabs come from the cursor
    // behind those; the match count goes through LDS, one global atomic per workgroup
    const uint64_t static_slabs = (uint64_t)gridDim.x * kWaves * KARG(slab);
    uint64_t slab_next = ((uint64_t)blockIdx.x * kWaves + wave) * KARG(slab), wave_matches = 0;   // wave-uniform
    bool told_nonascii = false;                  // (at most one LDS atomic per wave, not one per unit)
    uint32_t slab_left = KARG(slab);

    // Work distribution as in gft_scan2.hip: in round k the workgroup owns the units  k * (grid * waves) + b * waves +
    // [0, waves), and its waves take them one by one from a counter in LDS (item i = round i / waves, slot i % waves), so
    // a wave that drew cheap documents simply takes more.
    // The next unit's record and document offset are fetched while the current unit is processed.
    uint32_t* wg_book = reinterpret_cast<uint32_t*>(__builtin_assume_aligned(smem + off_next, 16));   // work counter, waves done, matches
    uint32_t* wg_next = wg_book;
    const uint64_t round_units = (uint64_t)gridDim.x * kWaves, wg_first = (uint64_t)blockIdx.x * kWaves;
    // (16, 8 or 4 waves: a shift instead of a 32-bit division per unit)
    const bool waves_p2 = (kWaves & (kWaves - 1)) == 0;
    const uint32_t waves_lg = (uint32_t)__builtin_ctz(kWaves);
    auto unit_of = [&](uint32_t item) {
        const uint32_t q = waves_p2 ? item >> waves_lg : item / kWaves;
        return (uint64_t)q * round_units + wg_first + (item - q * kWaves);
    };
    uint64_t u = wg_first + wave, nu = 0;                         // wave-uniform
    Unit un_n{0, 0, 0};
    uint64_t abs_n = 0, end_n = 0;                               // the next unit's document: blob offsets of its first byte and of the byte behind it
    if (u < P.n_units) { un_n = P.units[u]; abs_n = P.doc_off[un_n.doc]; end_n = P.doc_off[un_n.doc + 1]; }
    for (; u < P.n_units; u = nu) {
        const Unit un{(uint32_t)__builtin_amdgcn_readfirstlane(un_n.doc), (uint32_t)__builtin_amdgcn_readfirstlane(un_n.lo),
                      (uint32_t)__builtin_amdgcn_readfirstlane(un_n.hi)};
        const uint64_t doc_abs = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(abs_n >> 32)) << 32 |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)abs_n);
        const uint64_t doc_end = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(end_n >> 32)) << 32 |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)end_n);
        {
            uint32_t item = 0;
            if (lane == 0) item = __hip_atomic_fetch_add(wg_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nu = unit_of((uint32_t)__builtin_amdgcn_readfirstlane(item));
        }
        const bool more_units = nu < P.n_units;
        if (more_units) un_n = P.units[nu];
        const Ctx c{P, P.text + doc_abs, doc_abs, doc_abs < 7, doc_abs < 23, un.lo, un.hi, doc_abs + un.hi + 8 > P.text_bytes};
        const uint32_t own = un.hi - un.lo;
        const uint32_t nr = (own + 1023) >> 10;                      // rounds (<= 8)

        // ---- FILTER --------------------------------------------------------------------------------------------------
        if (P.prio) __builtin_amdgcn_s_setprio(0);
        uint32_t m0 = 0, m1 = 0;
        FOLD_JOB_VARS(fj_);
        bool fold_pending = false;
        if (own) {
            const uint8_t* src = c.dbase + un.lo + lane * 16;
            U128u nxt{0, 0, 0, 0};
            if (lane * 16 < own) nxt = *reinterpret_cast<const U128u*>(src);
            // the pair in front of the unit (groups of text[lo-2], text[lo-1]); nothing there at the blob start
            uint32_t carry = 0;
            if (doc_abs + un.lo >= 2) {
                const uint8_t* hp = c.dbase + un.lo;
                carry = mad24s(lcls[hp[-2]], G, lcls[hp[-1]]);
            }
            carry = __builtin_amdgcn_readfirstlane(carry);
            uint32_t acc = 0, njobs = 0;                            // njobs: pieces that hold a byte >= 0x80 (noted in the candidate list, idle until the filter is done)
            const bool want_fold = P.fold && P.nonascii;
            for (uint32_t r = 0; r < nr; r++) {
                const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
                // ASCII folding is not strings.ToLower once the text leaves ASCII (finder.go:140-142): gft_foldsafe_dev.hpp
                if (want_fold) fold_job_push(r * 1024 + lane * 16 < own && (((w[0] | w[1]) | (w[2] | w[3])) & 0x80808080u) != 0, r * 1024 + lane * 16, cand, P.cand_cap, njobs);
                if (r + 1 < nr && (r + 1) * 1024 + lane * 16 < own) nxt = *reinterpret_cast<const U128u*>(src + (r + 1) * 1024);
                uint32_t q[8];
#pragma unroll
                for (int d = 0; d < 4; d++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t co = lcls[(w[d] >> (16 * h)) & 0xFF];
                        const uint32_t ce = lcls[(w[d] >> (16 * h + 8)) & 0xFF];
                        q[2 * d + h] = mad24s(co, G, ce);
                    }
                // the pair in front of the lane's piece: the previous lane's last pair (lane 0: the previous round's lane 63)
                uint32_t qm = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)q[7], 0x138, 0xF, 0xF, false);   // wave_shr:1
                carry = __builtin_amdgcn_readlane(q[7], 63);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t x = mad24s(qm, G2, q[k]);
                    qm = q[k];
                    const uint32_t fw = lfilt[x >> 5];
                    acc = __builtin_amdgcn_alignbit(fw >> (x & 31), acc, 1);
                }
                if (r == 3) { m0 = acc; acc = 0; }
            }
            if (njobs && !told_nonascii) {
                // bit 1: a piece breaks the rule; bit 0: more pieces than the list holds -- the host then checks the text itself
                fold_pending = njobs <= P.cand_cap;
                const uint32_t bits = !fold_pending ? 1u : fold_jobs_begin(P.text, doc_end, doc_abs + un.lo, own, un.lo == 0, cand, njobs, FOLD_JOB_PASS(fj_)) ? 2u : 0u;
                if (bits) { told_nonascii = true; if (lane == 0) atomicOr(P.nonascii, bits); }
            }
            // probe t of the unit sits in round t / 512, lane (t / 8) % 64, bit 8 * (round % 4) + t % 8 of m0 (rounds 0-3) or
            // m1 (rounds 4-7); probes of the last round that start at or beyond the unit's end carry garbage
            const uint32_t lr = nr - 1;                               // the last round
            const int32_t avail = (int32_t)own - (int32_t)(lr * 1024 + lane * 16);
            const uint32_t vr = avail <= 0 ? 0u : avail >= 16 ? 8u : (uint32_t)(avail + 1) >> 1;
            uint32_t part = acc >> ((32 - 8 * (nr & 3)) & 31);        // the rounds since the last full mask word, low bits first
            if ((nr & 3) == 0) part = nr == 4 ? m0 : acc;
            const uint32_t keep = ~(((0xFFu << vr) & 0xFFu) << (8 * (lr & 3)));
            part &= keep;
            if (nr <= 4) m0 = part; else m1 = part;
        }

        if (P.prio) __builtin_amdgcn_s_setprio(1);
        if (more_units) { abs_n = P.doc_off[un_n.doc]; end_n = P.doc_off[un_n.doc + 1]; }

        // a unit that continues a document also probes the two positions in front of it that continue its parity (terms
        // whose window ends there but that end inside the unit): entries 0 (lo - 3) and 1 (lo - 1), long anchors only
        const uint32_t nborder = un.lo >= 3 ? 2u : un.lo >= 1 ? 1u : 0u;
        const uint32_t f = __popc(m0) + __popc(m1) + (lane == 0 ? nborder : 0);
        const uint32_t fincl = wave_incl_scan(f);
        const uint32_t ftotal = lane_value(fincl, 63);

        // room for the unit's matches: what is left of the wave's slab (a fresh one when little is left, so that a second
        // walk stays the exception)
        if (ftotal && slab_left < kScan3MinRoom) {
            uint64_t nb = 0;
            if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)KARG(slab));
            slab_next = static_slabs + __shfl(nb, 0, 64);
            slab_left = KARG(slab);
        }
        Out fifo{nullptr, nullptr, 0, 0};
        uint32_t nh = 0;
        if (DBG && P.dbg == 1) { if (lane == 0) { KARG(unit_start)[u] = slab_next; KARG(unit_count)[u] = ftotal & 0; } continue; }
        for (uint32_t walk = 0; walk < 2 && ftotal; walk++) {
            {
                const uint64_t cap = P.pool_cap;
                const uint64_t pool_room = cap > slab_next ? cap - slab_next : 0;
                fifo.term = P.pool_term + slab_next;
                fifo.pos = P.want_pos ? P.pool_pos + slab_next : nullptr;
                fifo.room = pool_room < slab_left ? (uint32_t)pool_room : slab_left;
                fifo.nf = 0;
            }
            uint32_t ns = 0, nj = 0;                               // parked survivors, parked short jobs
            // passes over lane ranges whose flagged probes fit the LDS list (one pass for a typical unit)
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
                        for (uint32_t i = 0; i < nborder; i++) cand[wpos++] = (uint16_t)(2 - nborder + i);
                    const uint32_t e0 = 8 * lane + 2;              // entry of the lane's probe 0 in round 0
                    uint32_t mm[2] = {m0, m1};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        uint32_t mk = mm[k];
                        while (mk) {
                            const uint32_t i = __builtin_ctz(mk);
                            mk &= mk - 1;
                            cand[wpos++] = (uint16_t)(e0 + 2048 * k + ((i & 24) << 6) + (i & 7));
                        }
                    }
                }
                wave_lds_fence();
                if (fold_pending) {                                // the pieces with high bytes whose loads went out in front of the list build
                    fold_pending = false;
                    if (fold_jobs_finish(FOLD_JOB_PASS(fj_)) && !told_nonascii) {
                        told_nonascii = true;
                        if (lane == 0) atomicOr(P.nonascii, 2u);
                    }
                }
                // ---- STAGE A: every listed probe -> LDS-only decisions; kStageAWays probes per lane and trip, the list
                // entries and text of trip t + 1 are fetched while trip t is worked on ------------------------------------
                if (P.prio) __builtin_amdgcn_s_setprio(2);
                bool n_on[kStageAWays];
                uint32_t n_ent[kStageAWays];
                Text8 n_tx[kStageAWays];
                auto fetch = [&](uint32_t i0) {
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        const uint32_t i = i0 + 64 * q + lane;
                        n_on[q] = i < ptotal;
                        n_ent[q] = cand[n_on[q] ? i : 0];
                    }
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) n_tx[q] = cand_load(c, probe_pos(c, n_ent[q]));
                };
                fetch(0);
                for (uint32_t i0 = 0; i0 < ptotal; i0 += 64 * kStageAWays) {
                    bool on[kStageAWays];
                    uint32_t ent[kStageAWays];
                    Text8 tx[kStageAWays];
                    CandKeys k[kStageAWays];
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) { on[q] = n_on[q]; ent[q] = n_ent[q]; tx[q] = n_tx[q]; }
                    if (i0 + 64 * kStageAWays < ptotal) fetch(i0 + 64 * kStageAWays);
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) k[q] = cand_keys(lcls, G, G2, tx[q]);
                    // Bloom cell of (window, front group): can a term of length >= 4 be anchored here at all?
                    bool go[kStageAWays];
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        const uint32_t cell_i = scan3_bloom_cell(k[q].key5, P.bloom_lg);
                        const uint32_t cell = BLOOM_LDS ? lbloom[cell_i] : P.bloom[cell_i];
                        const uint32_t h = k[q].key5 * kScan3BloomMul2;
                        go[q] = on[q] && ((cell >> (h >> 27)) & (cell >> ((h >> 22) & 31)) & 1u);
                    }
                    // terms of length <= 3 ending at p and at p - 1 (regular probes only: entries >= 2): the probes whose short3
                    // cells are not empty are parked as jobs -- over the part of the candidate list that has been read
                    // already -- and emitted densely by stage S
                    if (have_short) {
                        // (entries up to i0 + 2 trips are in registers; once the whole list is, all of its room is free)
                        const uint32_t job_cap = i0 + 128 * kStageAWays < ptotal ? (i0 + 128 * kStageAWays) / 2 : P.cand_cap / 2;
                        if (nj + 64 * kStageAWays > job_cap) {
                            if (!(DBG && P.dbg == 2)) stage_s(c, lcls, lrec, jobs, nj, fifo); else nj = 0;
                        }
#pragma unroll
                        for (int q = 0; q < kStageAWays; q++) {
                            if (i0 + 64 * q >= ptotal) continue;
                            const bool reg = on[q] && ent[q] >= 2;
                            const uint32_t sid0 = reg && probe_pos(c, ent[q]) < c.hi ? ls3[k[q].x3p] : 0u;
                            const uint32_t sid1 = reg ? ls3[k[q].x3m] : 0u;
                            const bool has = (sid0 | sid1) != 0;
                            const uint64_t jb = __ballot(has);
                            if (has) jobs[nj + __builtin_amdgcn_mbcnt_hi((uint32_t)(jb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)jb, 0))] =
                                ent[q] | sid0 << 16 | sid1 << 24;
                            nj += (uint32_t)__popcll(jb);
                        }
                    }
                    if (DBG && (P.dbg == 2 || P.dbg == 3)) { fifo.nf += (uint32_t)__popcll(__ballot(go[0] && go[1] && k[0].x == ~0u)); continue; }
                    // park the survivors with their window keys; a full list goes through stage B first
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        const uint64_t sb = __ballot(go[q]);
                        const uint32_t cnt = (uint32_t)__popcll(sb);
                        if (ns + cnt > kScan3SurvCap) {
                            if (P.prio) __builtin_amdgcn_s_setprio(3);
                            stage_b(c, surv, ns, fifo);
                            if (P.prio) __builtin_amdgcn_s_setprio(2);
                        }
                        if (go[q]) surv[ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0))] =
                            make_uint2(ent[q], k[q].x);
                        ns += cnt;
                    }
                }
                if (P.prio) __builtin_amdgcn_s_setprio(3);
                if (!(DBG && P.dbg == 2)) stage_s(c, lcls, lrec, jobs, nj, fifo); else nj = 0;
                stage_b(c, surv, ns, fifo);
                wave_lds_fence();
                l0 = l1;
            }
            nh = fifo.nf;
            if (nh <= slab_left) break;                            // everything fits what the unit was given (or would have, had
                                                                   // the pool not run out: the host re-runs then)
            // ---- the unit's matches exceed the rest of the slab: a fresh slab that holds all of them, and the same walk again
            const uint32_t want = nh > KARG(slab) ? nh : KARG(slab);
            uint64_t nb = 0;
            if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)want);
            slab_next = static_slabs + __shfl(nb, 0, 64);
            slab_left = want;
        }
        // (a unit whose cells lie beyond the pool wrote nothing: the host sees the cursor and runs the batch again)
        if (fold_pending) {                                        // (a unit without a flagged probe never reached stage A)
            fold_pending = false;
            if (fold_jobs_finish(FOLD_JOB_PASS(fj_)) && !told_nonascii) {
                told_nonascii = true;
                if (lane == 0) atomicOr(P.nonascii, 2u);
            }
        }
        if (lane == 0) { KARG(unit_start)[u] = slab_next; KARG(unit_count)[u] = slab_next + nh <= P.pool_cap ? nh : 0u; }
        slab_next += nh;
        slab_left -= nh;
        wave_matches += nh;
    }
    if (lane == 0) {
        uint32_t* wg = wg_book;
        if (wave_matches)
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(wg + 2), (unsigned long long)wave_matches, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t done = __hip_atomic_fetch_add(wg + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) & 0x7FFFFFFFu;
        if (done + 1 == kWaves) {
            const unsigned long long all = __hip_atomic_load(reinterpret_cast<unsigned long long*>(wg + 2), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP);
            if (all) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(n_matches)), all);
        }
    }
}

}  // namespace

static size_t scan3_fixed_lds(uint32_t filter_words, uint32_t short3_bytes, uint32_t srec_words, uint32_t bloom_lds_bytes) {
    return ((256 + (size_t)filter_words * 4 + short3_bytes + (((size_t)srec_words * 4 + 15) & ~(size_t)15) + bloom_lds_bytes + 15) & ~(size_t)15) +
           16;       // the workgroup's bookkeeping (aligned)
}

bool scan3_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t srec_words, uint32_t bloom_lds_bytes, size_t lds_max,
                uint32_t* waves, uint32_t* cand_cap) {
    const size_t fixed = scan3_fixed_lds(filter_words, short3_bytes, srec_words, bloom_lds_bytes);
    for (uint32_t w : {16u, 12u, 8u, 4u}) {
        const size_t per_min = kScan3SurvCap * 8 + kScan3CandCapMin * 2;
        if (fixed + (size_t)w * per_min > lds_max) continue;
        const size_t per = ((lds_max - fixed) / w) & ~(size_t)15;
        const size_t cap = (per - kScan3SurvCap * 8) / 2;
        *waves = w;
        *cand_cap = (uint32_t)(cap > 4096 + 2 ? 4096 + 2 : cap);
        return true;
    }
    return false;
}

hipError_t launch_scan3(const Scan3Params& P, uint32_t waves, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const size_t lds = scan3_fixed_lds(P.filter_words, P.short3_bytes, P.srec_words, P.bloom_lds ? 4u << P.bloom_lg : 0) +
                       (size_t)waves * (kScan3SurvCap * 8 + P.cand_cap * 2);
    using Kern = void (*)(const Scan3Params);
    const Kern fn = P.dbg ? (P.bloom_lds ? k_scan3<true, true> : k_scan3<false, true>) : (P.bloom_lds ? k_scan3<true, false> : k_scan3<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t g = (P.n_units + waves - 1) / waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    fn<<<dim3(grid), dim3(waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
