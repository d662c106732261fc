// gft_scan2_dev.hpp -- device helpers of the suffix-window scan kernels (gft_scan2.hip: one unit per wave iteration;
// gft_scan4.hip: the streaming form): text loads around a flagged position, window keys, the fingerprint decision, bucket
// slots and their byte compares, short-term records, the LDS match fifo.  Included inside namespace gft { namespace { ... } }
// of a .hip file, after gft_kernels.hpp; KARG's parameter block is Scan2Params.
#pragma once
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U128u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U64u { uint32_t lo, hi; };

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// a * b + c on 24-bit operands: one full-rate instruction (the compiler's own choice is v_mul_lo_u32 / v_mad_u64_u32);
// the wave-uniform multiplier comes straight from a scalar register (no v_mov to materialise it)
__device__ __forceinline__ uint32_t mad24s(uint32_t a, uint32_t sb, uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sb), "v"(c));
    return d;
}
// LDS tables at fixed addresses (the kernel's only LDS object is the dynamic array, which starts at 0; checked at
// kernel entry): constant bases fold into the ds_read offset field
typedef __attribute__((address_space(3))) const uint8_t lds_u8;
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) { return reinterpret_cast<const U32u*>(p)->v; }

// Kernel arguments that are needed once per unit or less (output buffers, tables of the rare paths) are read from the
// kernarg segment where they are used instead of living in scalar registers for the whole kernel: the unit loop keeps
// more values alive than there are SGPRs, and every spilled one costs VALU slots (v_writelane / v_readlane).  The asm
// makes the address opaque, so the load can be neither merged with the preloaded arguments nor hoisted.
typedef __attribute__((address_space(4))) const uint8_t karg_u8;
template <class T>
__device__ __forceinline__ T karg_field(uint32_t off) {
    karg_u8* ka = (karg_u8*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(__attribute__((address_space(4))) const T*)(ka + off);
}
#define KARG(field) karg_field<decltype(Scan2Params::field)>((uint32_t)offsetof(Scan2Params, field))

// ASCII lower-casing of four packed bytes (finder/finder.go:140-142 for ASCII text)
__device__ __forceinline__ uint32_t fold4(uint32_t w) {
    const uint32_t h = w & 0x7F7F7F7Fu;
    const uint32_t ge_a = h + 0x3F3F3F3Fu;          // bit 7 set where byte >= 'A'
    const uint32_t gt_z = h + 0x25252525u;          // bit 7 set where byte >  'Z'
    const uint32_t up = ge_a & ~gt_z & ~w & 0x80808080u;
    return w | (up >> 2);
}
__device__ __forceinline__ uint32_t fold1(uint32_t b) { return (b - 'A' < 26u) ? b + 32 : b; }

// inclusive prefix sum over the 64 lanes with DPP moves (row shifts inside the four rows of 16 lanes, then the row
// totals are broadcast into the rows behind them): ten VALU instructions, no LDS traffic, no index arithmetic
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}
// value of lane l (wave-uniform l) -- a scalar read instead of an LDS permute
__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t l) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}

struct Ctx {
    const Scan2Params& P;
    const uint8_t* cls;        // LDS
    const uint32_t* filt;      // LDS
    const uint8_t* short3;     // LDS (nullptr: the dictionary has no term shorter than the window)
    const uint8_t* fpt;        // LDS (fpt_lg == 0) or the global table
    const uint32_t* lrec;      // LDS: short-term records, 3 words each
    const uint8_t* dbase;      // first byte of the document
    uint64_t doc_abs;          // offset of the document inside the text blob
    uint32_t kp2;              // kp * kp
    bool near0;                // wave-uniform: the document starts within 7 bytes of the blob start
    bool near24;               // ... within 23 bytes
    uint32_t lo, hi;           // the unit: a match belongs to the unit that holds its END position
    bool near_end;             // wave-uniform: the unit ends within 4 bytes of the blob end
    uint32_t dbg;              // GFT_SCAN_DEBUG bits in the timing-study instantiations, the constant 0 in production
};

// the (up to) four bytes behind position p, text[p+1 .. p+4], for the tails of shifted terms; bytes past the blob end
// read as zero (a tail that reached there would end outside the unit and is dropped by the range check anyway)
__device__ __forceinline__ uint32_t tail_load(const Ctx& c, uint32_t p) {
    if (__builtin_expect(c.near_end, 0)) {
        uint32_t v = 0;
        for (uint32_t b = 0; b < 4; b++)
            if (c.doc_abs + p + 1 + b < c.P.text_bytes) v |= (uint32_t)c.dbase[(uint64_t)p + 1 + b] << (8 * b);
        return v;
    }
    return load_u32_unaligned(c.dbase + (uint64_t)p + 1);
}
// reported position of a match whose window ends at p (lw = the slot's len word)
__device__ __forceinline__ uint32_t match_pos(const Scan2Params& P, uint32_t p, uint32_t lw) {
    const uint32_t off = lw >> 24, L1 = lw & kScan2LenMask;
    return P.pos_end ? p + off : p + 1 - L1;
}

// ---- verification of one flagged position p, in three separable steps so that several candidates can have their
// the cheap LDS-only decisions (stage A) and the L2 bucket probes (stage B) can run as separate, dense passes ------------

// step 1: one 8-byte load brings the window (bytes p-3..p) and the 4 bytes in front of it (p-7..p-4).
// Positions before the document start need no special casing here: the bytes there (the previous document's, or
// zeros in front of the blob) can only change keys of windows that reach across the start, and every term such a
// window may name is longer than p + 1 and is dropped by the length check at emission.
constexpr int kStageAWays = 2;      // stage A: candidates a lane works on at once
struct Cand { uint32_t p, x, x3, tw, sid; bool go_long; };
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
    if (__builtin_expect(c.near0, 0)) return cand_load_slow(c, p);        // wave-uniform: first document of the blob
    const U64u v = *reinterpret_cast<const U64u*>(c.dbase + (int64_t)p - 7);
    return Text8{v.lo, v.hi};
}
// window key, bucket hash and the short-term record id
template <bool WANT_SID = true>
__device__ __forceinline__ void cand_keys(const Ctx& c, uint32_t p, const Text8 t, Cand& k) {
    const uint32_t kp = c.P.kp, w = t.w;
    k.p = p;
    k.tw = t.tw;        // raw: the fingerprint ignores the case bit, the bucket compare folds when asked to
    const uint32_t c0 = c.cls[w & 0xFF], c1 = c.cls[(w >> 8) & 0xFF], c2 = c.cls[(w >> 16) & 0xFF];
    const uint32_t lo = mad24s(c2, kp, c.cls[w >> 24]);          // key = (c0 kp + c1) kp^2 + (c2 kp + c3)
    const uint32_t x3 = mad24s(c1, c.kp2, lo);
    k.x = mad24s(mad24s(c0, kp, c1), c.kp2, lo);
    k.x3 = x3;
    k.sid = WANT_SID && c.short3 ? c.short3[x3] : 0;      // (stage B has no use for the short-term record: one LDS probe less)
}
// LDS-only decision: can a term of length >= 4 end here at all (fingerprint of the bytes in front of the window)?
// Most flagged positions stop here without touching L2.
template <bool FPT_LDS>
__device__ __forceinline__ void cand_decide(const Ctx& c, Cand& k) {
    const uint32_t b1n = (k.tw >> 24) & 0xDFu;
    const uint32_t flg = FPT_LDS ? 0u : c.P.fpt_lg;
    const uint8_t* f = FPT_LDS ? c.fpt : c.P.fpt;
    const uint32_t cx = f[scan2_fpt_xcell(k.x, flg)], cg0 = f[scan2_fpt_gcell(k.x, b1n, 0, flg)], cg1 = f[scan2_fpt_gcell(k.x, b1n, 1, flg)];
    k.go_long = scan2_fpt_pass(cx, cg0, cg1, scan2_fpt_xmix(k.x), k.tw);
    if (c.dbg & 12) {           // timing studies (wrong results): 4 = no bucket-table access, 8 = no short-term records
        if (c.dbg & 4) k.go_long = false;
        if (c.dbg & 8) k.sid = 0;
    }
}
template <bool FPT_LDS>
__device__ __forceinline__ void cand_text(const Ctx& c, uint32_t p, Cand& k) {
    cand_keys(c, p, cand_load(c, p), k);
    cand_decide<FPT_LDS>(c, k);
}

// step 2 (bucket table): the 20 bytes in front of the window as the slots store them, f[k] = text[p-7-4k .. p-4-4k]
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
__device__ __forceinline__ void front_fold(Front& t) {
#pragma unroll
    for (int k = 0; k < 5; k++) t.f[k] = fold4(t.f[k]);
}

struct Slot { uint4 a, b; };      // a = {key, info, len, front[0]}, b = front[1..4]
__device__ __forceinline__ Slot slot_load(const Scan2Slot* s) {
    const uint4* q = reinterpret_cast<const uint4*>(s);
    return Slot{q[0], q[1]};
}
// both candidate slots of key x -> the one that holds it; false: no term ends with this window
__device__ __forceinline__ bool slot_pick(uint32_t x, const Slot& s0, const Slot& s1, Slot& out) {
    const bool use1 = s1.a.x == x;
    out.a = use1 ? s1.a : s0.a;
    out.b = use1 ? s1.b : s0.b;
    return use1 || s0.a.x == x;
}

// does the term described by e have its window end at p (and its own end inside the unit)?  t = the (folded) bytes in
// front of the window, tl = the raw bytes behind it.  kmax: dwords of `front` to look at (wave-uniform bound, or 5)
// pd = p counted from the start of its document, [lo, hi) = where the term must END to be this caller's (one unit per wave
// iteration: pd = p, the unit; streaming: per lane -- gft_scan4.hip)
__device__ __forceinline__ bool entry_ok_x(const Ctx& c, uint32_t p, uint32_t pd, uint32_t lo, uint32_t hi, const Front& t, uint32_t tl,
                                           const Slot& e, uint32_t kmax) {
    const Scan2Params& P = c.P;
    const uint32_t L = e.a.z & kScan2LenMask, off = e.a.z >> 24;      // L: the term up to the end of its window
    const int32_t nfront = (int32_t)L - 4;
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if ((uint32_t)k < kmax) {
            int32_t nb = min(max(nfront - 4 * k, 0), 4);                           // bytes of this dword the term owns
            if (k == 4 && off) nb = 0;                                             // front[4] holds the tail instead
            const uint32_t mask = (uint32_t)(0xFFFFFFFF00000000ull >> (8 * nb));   // ... the ones next to the window
            const uint32_t fk = k == 0 ? e.a.w : k == 1 ? e.b.x : k == 2 ? e.b.y : k == 3 ? e.b.z : e.b.w;
            diff |= (t.f[k] ^ fk) & mask;
        }
    }
    if (off) {
        const uint32_t tv = P.fold ? fold4(tl) : tl;
        diff |= (tv ^ e.b.w) & (0xFFFFFFFFu >> (8 * (4 - off)));
    }
    const uint32_t pe = p + off;                                                   // where the term ends
    bool ok = L <= pd + 1 && diff == 0 && pe >= lo && pe < hi;
    const uint32_t inl = off ? kScan2InlineLen - 4 : kScan2InlineLen;
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
__device__ __forceinline__ bool entry_ok(const Ctx& c, uint32_t p, const Front& t, uint32_t tl, const Slot& e, uint32_t kmax) {
    return entry_ok_x(c, p, p, c.lo, c.hi, t, tl, e, kmax);
}

// the three words of short-term record `sid` of 3-window x3 (LDS; ids beyond 254 live in global memory)
__device__ __forceinline__ void short_record(const Ctx& c, uint32_t sid, uint32_t x3, uint32_t (&r)[3]) {
    // two separate accesses (an LDS read, and -- rarely -- a global one): a pointer that may be either would turn both
    // into flat loads, which wait on the vector-memory AND the LDS counters
    const bool big = sid == 255 && KARG(short3_big) != nullptr;
    const uint32_t* src = c.lrec + 3 * (big ? 0 : sid);
    r[0] = src[0]; r[1] = src[1]; r[2] = src[2];
    if (__builtin_expect(big, 0)) {
        const uint32_t* g = KARG(shorts_packed) + 3 * (size_t)KARG(short3_big)[x3];
        r[0] = g[0]; r[1] = g[1]; r[2] = g[2];
    }
}

// step 3 (ordered path): all terms that end at p, longest first.  MODE 0: count and stage per lane in LDS;
// MODE 1: write to the pool at out_base.
template <int MODE>
__device__ __forceinline__ void cand_finish(const Ctx& c, const Cand& k, uint32_t& cnt, uint2* stage, uint64_t out_base) {
    const Scan2Params& P = c.P;
    const uint32_t p = k.p;
    auto emit = [&](uint32_t term, uint32_t lw) {
        const uint32_t pos = match_pos(P, p, lw);
        if (MODE == 0) {
            if (cnt < kScan2StageCap) stage[cnt * 64] = make_uint2(term, pos);
        } else {
            KARG(pool_term)[out_base + cnt] = term;
            if (P.want_pos) KARG(pool_pos)[out_base + cnt] = pos;
        }
        cnt++;
    };
    // ---- terms of length >= 4, longest first ---------------------------------------------------------------------
    if (k.go_long) {
        const Slot s0 = slot_load(&P.slots[scan2_pair_slot(k.x, 0, P.slot_shift, P.slot_seed)]);
        const Slot s1 = slot_load(&P.slots[scan2_pair_slot(k.x, 1, P.slot_shift, P.slot_seed)]);
        Front t = front_load(c, p, k.tw);
        const uint32_t tl = tail_load(c, p);
        if (P.fold) front_fold(t);
        Slot e;
        if (slot_pick(k.x, s0, s1, e)) {
            uint32_t n_ent = 1, more_at = 0;
            if (e.a.y & kScan2Multi) {
                more_at = e.a.y & ~kScan2Multi;
                n_ent = e.a.z;
                e = slot_load(&P.more[more_at]);
            }
            for (uint32_t j = 0;;) {
                if (entry_ok(c, p, t, tl, e, 5)) emit(e.a.y, e.a.z);
                if (++j >= n_ent) break;
                e = slot_load(&P.more[more_at + j]);
            }
        }
    }
    // ---- terms of length <= 3 (records from the LDS 3-window table) -------------------------------------------------------
    if (k.sid) {
        uint32_t r[3];
        short_record(c, k.sid, k.x3, r);
#pragma unroll
        for (uint32_t j = 0; j < 3; j++)
            if (r[j] && (r[j] >> 28) <= p + 1) emit(r[j] & 0x0FFFFFFFu, r[j] >> 28);
    }
}

// ---- unordered path: matches go to the wave's LDS fifo.  `nf` (matches so far) is wave-uniform: every append happens
// in wave-uniform control flow, lanes that have something to append take consecutive cells (ballot + mbcnt) -------------
__device__ __forceinline__ void fifo_append(bool em, uint32_t term, uint32_t pos, uint2* fifo, uint32_t& nf) {
    const uint64_t mask = __ballot(em);
    if (em) {
        const uint32_t idx = nf + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (idx < kScan2FifoCap) fifo[idx] = make_uint2(term, pos);
    }
    nf += (uint32_t)__popcll(mask);
}

// terms of length <= 3 ending at the lanes' positions (sid = 0: none); wave-uniform call
__device__ __forceinline__ void finish_short(const Ctx& c, uint32_t p, uint32_t sid, uint32_t x3, uint2* fifo, uint32_t& nf) {
    if (!__any(sid != 0)) return;
    uint32_t r[3] = {0, 0, 0};
    if (sid) short_record(c, sid, x3, r);
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        if (j && !__any(r[j] != 0)) break;
        const uint32_t L = r[j] >> 28;
        fifo_append(r[j] != 0 && L <= p + 1, r[j] & 0x0FFFFFFFu, c.P.pos_end ? p : p + 1 - L, fifo, nf);
    }
}

// dwords of front bytes a term of length L owns (0 for lanes that are not active)
__device__ __forceinline__ uint32_t wave_kmax(uint32_t L) {
    return __any(L > 20) ? 5 : __any(L > 16) ? 4 : __any(L > 12) ? 3 : __any(L > 8) ? 2 : 1;
}
// fold the first kmax dwords of t that are not folded yet (`done` = how many are; wave-uniform)
__device__ __forceinline__ void front_fold_upto(Front& t, uint32_t& done, uint32_t kmax) {
#pragma unroll
    for (int k = 0; k < 5; k++)
        if ((uint32_t)k >= done && (uint32_t)k < kmax) t.f[k] = fold4(t.f[k]);
    done = kmax > done ? kmax : done;
}

// A wave's deferred bucket entries: {candidate position, index into `more`} pairs parked in LDS so that the entries
// of multi-term buckets are verified densely (64 distinct entries per trip) instead of one round per bucket depth.
struct Deferred { uint2* list; uint32_t cap, n; };

// terms of length >= 4 ending at the lanes' positions (`on`: this lane has a candidate); wave-uniform call.
// s0, s1: the key's two candidate slots; t: the bytes in front of the window, not yet folded.  One-term buckets are
// verified here; the entries of multi-term buckets are deferred (or, if the list is full, verified in place).
__device__ __forceinline__ void finish_long(const Ctx& c, bool on, uint32_t rel, const Cand& k, const Slot& s0, const Slot& s1,
                                            Front t, uint32_t tl, uint2* fifo, uint32_t& nf, Deferred& d) {
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
        fifo_append(ok, e.a.y, match_pos(P, k.p, e.a.z), fifo, nf);
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
                    make_uint2(rel, more_at + j);
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
        const bool ok = act && entry_ok(c, k.p, t, tl, cur, kmax);
        fifo_append(ok, cur.a.y, match_pos(P, k.p, cur.a.z), fifo, nf);
        cur = nxt;
    }
}

// verify the parked entries, 64 per trip
__device__ __forceinline__ void drain_deferred(const Ctx& c, uint32_t unit_lo, uint2* fifo, uint32_t& nf, Deferred& d) {
    const Scan2Params& P = c.P;
    const uint32_t lane = lane_id();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t i0 = 0; i0 < d.n; i0 += 64) {
        const bool on = i0 + lane < d.n;
        const uint2 it = d.list[on ? i0 + lane : 0];
        const uint32_t p = unit_lo + it.x;
        const Slot e = slot_load(&P.more[it.y]);
        const Text8 t8 = cand_load(c, p);
        Front t = front_load(c, p, t8.tw);
        const uint32_t tl = tail_load(c, p);
        const uint32_t kmax = wave_kmax(on ? e.a.z & kScan2LenMask : 0);
        uint32_t folded = 0;
        if (P.fold) front_fold_upto(t, folded, kmax);
        const bool ok = on && entry_ok(c, p, t, tl, e, kmax);
        fifo_append(ok, e.a.y, match_pos(P, p, e.a.z), fifo, nf);
    }
    d.n = 0;
    __builtin_amdgcn_wave_barrier();
}

