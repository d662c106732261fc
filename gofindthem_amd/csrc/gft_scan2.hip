// gft_scan2.hip -- the Aho-Corasick scan for gfx950, "suffix-window" form (tables: scan2_tables.hpp).
// Replaces (*Matcher).MatchAll behind CloudflareForkEngine.FindSubstrings (finder/substringEngine.go:110-119).
//
// One wavefront per work unit (a document, or a slice of a long one); lane k owns C consecutive bytes.
//   phase 1  FILTER  every lane streams its bytes from HBM with 16-byte loads, maps them to byte classes through a
//            256-byte LDS table and probes the LDS-resident 4-window filter once per byte.  Neighbouring bytes are
//            independent (the depth-4 automaton is 4-local), so there is no dependent lookup chain; the per-byte flag
//            is shifted into a lane-private bit mask with v_alignbit.
//   phase 2  VERIFY  flagged positions (~8 % of the text) are listed in LDS and dealt densely to the lanes.  Stage A:
//            LDS-only checks -- terms of length <= 3 from the short3 records, and the per-term fingerprint table (keyed by
//            the window and the byte in front of it) decides whether a longer term can be anchored here at all.  Stage B:
//            the survivors (~1.3 %) go to the L2-resident bucket table: both candidate 32-byte slots of the window key are
//            loaded at once, a slot holds a whole term up to 24 bytes, so a lookup is two loads deep (text, slot).
//            A term's window is not necessarily its last four bytes (shifted anchors, gft_kernels.hpp): the bytes behind
//            the window are compared together with the bytes in front of it, and a match belongs to the unit that holds
//            its END -- a unit that continues a document also verifies the four positions in front of it.
//   output   matches are appended to a per-wave LDS fifo (ballot + mbcnt) and flushed coalesced, in any order (the solver
//            does not care; CSR results are sorted per unit by k_gather_sorted).  Units whose matches overflow the fifo
//            take the staging path: matches are staged per lane, a DPP prefix sum gives every lane its offset.  The wave
//            takes its room from a private slab (one global atomic per ~4 K matches).  Nothing is ever truncated: the
//            host re-runs with a larger pool if the cursor overran.
// HBM traffic: text once + 8 B per match (4 B in presence-only mode); tables are LDS / L2 resident.  No MFMA (byte
// automaton, not a contraction).  The kernel is VALU-issue bound (profiles/r1_sq_counters.json).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gft_kernels.hpp"

namespace gft {

namespace {

#include "gft_scan2_dev.hpp"

template <int MODE, bool FPT_LDS>
__device__ __forceinline__ void verify(const Ctx& c, uint32_t p, uint32_t& cnt, uint2* stage, uint64_t out_base) {
    Cand k;
    cand_text<FPT_LDS>(c, p, k);
    if (p < c.lo) k.sid = 0;            // a position in front of the unit: only terms that end inside it
    if (k.go_long || k.sid) cand_finish<MODE>(c, k, cnt, stage, out_base);
}

template <int MODE, bool FPT_LDS>
__device__ __forceinline__ void verify_masks(const Ctx& c, uint32_t my_lo, uint32_t mb, uint32_t m0, uint32_t m1, uint32_t m2,
                                             uint32_t m3, uint32_t& cnt, uint2* stage, uint64_t out_base) {
    // one copy of the (large) verification body: the 32-position block index k is a scalar loop variable; block 0 =
    // the positions in front of the unit (mb: bit i = position lo - kScan2MaxOff + i, lane 0 only)
    for (uint32_t k = 0; k < 5; k++) {
        uint32_t mk = k == 0 ? mb : k == 1 ? m0 : k == 2 ? m1 : k == 3 ? m2 : m3;
        const uint32_t at = k == 0 ? c.lo - kScan2MaxOff : my_lo + 32 * (k - 1);
        while (__any(mk != 0)) {
            if (mk) {
                const uint32_t i = __builtin_ctz(mk);
                mk &= mk - 1;
                verify<MODE, FPT_LDS>(c, at + i, cnt, stage, out_base);
            }
        }
    }
}

// ORDERED: matches of a unit leave in text order (CSR results); otherwise any order (solver input)
// FPT_LDS: the fingerprint table is staged in LDS (dictionaries up to kScan2FptLdsItems long terms); otherwise it is read
// in place from global memory -- the spill path of large dictionaries, which also leaves more LDS to the candidate lists
// DBG: the timing-study knock-outs (GFT_SCAN_DEBUG) exist only in the DBG = true instantiations; production launches
// run DBG = false, where `dbg` is the constant 0 and every such branch is compiled out
template <bool HASHED, bool ORDERED, bool FPT_LDS, bool DBG>
__global__ void __launch_bounds__(kScan2Threads) k_scan2(const Scan2Params P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* cls = smem;
    uint32_t* filt = reinterpret_cast<uint32_t*>(smem + 256);
    uint8_t* short3 = smem + 256 + (size_t)P.filter_words * 4;
    uint8_t* fpt = short3 + P.short3_bytes;
    uint32_t* lrec = reinterpret_cast<uint32_t*>(fpt + (FPT_LDS ? kScan2FptSize : 0));
    // the workgroup's 16 bytes of bookkeeping (work counter, waves done, match count: a 64-bit LDS atomic), 16-byte aligned
    // whatever the sizes of the tables in front of it
    uint32_t* wg_next = reinterpret_cast<uint32_t*>(smem + (((size_t)(reinterpret_cast<uint8_t*>(lrec) - smem) + P.shorts_words * 4 + 15) & ~(size_t)15));
    uint8_t* wave_lds_all = reinterpret_cast<uint8_t*>(wg_next) + 16;

    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = P.cls[i];
    for (uint32_t i = threadIdx.x; i < P.filter_words; i += blockDim.x) filt[i] = P.filter[i];
    for (uint32_t i = threadIdx.x; i < P.short3_bytes / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(short3)[i] = reinterpret_cast<const uint32_t*>(P.short3)[i];
    for (uint32_t i = threadIdx.x; FPT_LDS && i < kScan2FptSize / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(fpt)[i] = reinterpret_cast<const uint32_t*>(P.fpt)[i];
    for (uint32_t i = threadIdx.x; i < P.shorts_words; i += blockDim.x) lrec[i] = P.shorts_packed[i];
    if (threadIdx.x == 0) { wg_next[0] = blockDim.x >> 6; wg_next[1] = wg_next[2] = wg_next[3] = 0; }   // [0] work counter (every
                                                                 // wave starts with the item of its own number), [1] waves done, [2..3] matches
    __syncthreads();
    if (DBG && (P.dbg & 128)) return;                            // timing study: launch + table staging alone

    // (the wave index is the same in all lanes: as a scalar, the unit bookkeeping below stays off the vector ALU)
    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per-wave LDS region: [fifo / ordered staging: kScan2FifoCap x 8 B][candidate list: cand_cap x 2 B]
    uint8_t* wave_lds = wave_lds_all + (size_t)wave * (kScan2FifoCap * 8 + P.cand_cap * 2);
    uint2* fifo = reinterpret_cast<uint2*>(wave_lds);
    uint2* stage = fifo + lane;                                   // ordered path: entry k of this lane is stage[k * 64]
    uint16_t* cand = reinterpret_cast<uint16_t*>(wave_lds + kScan2FifoCap * 8);
    const uint32_t kp = __builtin_amdgcn_readfirstlane(P.kp), kp2 = __builtin_amdgcn_readfirstlane(kp * kp);
    lds_u8* lcls = (lds_u8*)0;
    lds_u32* lfilt = (lds_u32*)256;
    if ((uint32_t)(uintptr_t)(lds_u8*)smem != 0) __builtin_trap();   // see lds_u8

    // phase clocks of the timing studies (GFT_SCAN_DEBUG & 64): cycles of this wave per phase, summed into dbg_counters[4..11]
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = DBG ? clock64() : 0;
    auto mark = [&](int ph) {
        if (DBG && (P.dbg & 64)) { const unsigned long long now = clock64(); tl[ph] += now - tprev; tprev = now; }
    };
    // Match pool: every wave of the grid owns one slab from the start (wave g: [g * slab, (g + 1) * slab)), further slabs come
    // from the cursor behind those.  Same-address atomics retire at ~ 15 ns each across the device: 4 096 waves asking for
    // their first slab in the same microsecond, and adding their match counts when they all finish, were 0.1 ms of every
    // launch -- a fifth of a 125 000-document batch.  (The host adds the static slabs to the cursor it reads back.)
    const uint64_t static_slabs = (uint64_t)gridDim.x * (blockDim.x >> 6) * KARG(slab);
    uint64_t slab_next = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave) * KARG(slab), wave_matches = 0;   // wave-uniform
    bool told_nonascii = false;                  // (at most one LDS atomic per wave, not one per unit)
    uint32_t slab_left = KARG(slab);

    // Work distribution.  The workgroup owns the units  b * waves + k * (grid * waves) + [0, waves)  of every round k -- the
    // 4 096 waves of the grid move through the text side by side -- and its waves take them one by one from a counter in
    // LDS (item i = round i / waves, slot i % waves), so a wave that drew cheap documents simply takes more: with one fixed
    // slot per wave the slowest of the 4 096 waves took 8.7 % longer than the average one at 244 units per wave, 50 % at
    // 30, while the sums over a whole workgroup scatter by 1.6 %.  (A counter in global memory does not work: same-address
    // atomics retire at ~ 15 ns each across the device, 45 000 tickets for 9 % of the units cost 0.46 ms.)
    // The next unit's record and document offset are fetched while the current unit is processed.
    const uint32_t wg_waves = blockDim.x >> 6;
    const uint64_t round_units = (uint64_t)gridDim.x * wg_waves, wg_first = (uint64_t)blockIdx.x * wg_waves;
    auto unit_of = [&](uint32_t item) { return (uint64_t)(item / wg_waves) * round_units + wg_first + item % wg_waves; };
    uint64_t u = wg_first + wave, nu = 0;                         // wave-uniform
    Unit un_n{0, 0, 0};
    uint64_t abs_n = 0;
    if (u < P.n_units) { un_n = P.units[u]; abs_n = P.doc_off[un_n.doc]; }
    for (; u < P.n_units; u = nu) {
        // the unit's record is the same in all lanes: in scalar registers the bookkeeping below costs no VALU slots
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
        const Ctx c{P, cls, filt, P.short3_bytes ? short3 : nullptr, fpt, lrec, P.text + doc_abs, doc_abs, kp2,
                    doc_abs < 7, doc_abs < 23, un.lo, un.hi, doc_abs + un.hi + 4 > P.text_bytes, DBG ? P.dbg : 0u};
        // A term whose window ends up to kScan2MaxOff bytes before the unit may itself end inside it: those positions
        // (units that continue a document only) join the candidates unconditionally, for such terms only
        const uint32_t nborder = un.lo < kScan2MaxOff ? un.lo : kScan2MaxOff;
        const uint32_t ubase = un.lo - kScan2MaxOff;               // candidate lists hold p - ubase (may wrap; p never does)
        const uint32_t own = un.hi - un.lo;
        const uint32_t C = ((own + 63) / 64 + 3) & ~3u;            // bytes per lane (multiple of 4, <= 128)
        const uint32_t my_lo = un.lo + lane * C;
        const uint32_t my_hi = my_lo + C < un.hi ? my_lo + C : un.hi;
        const uint32_t nvalid = my_lo < un.hi ? my_hi - my_lo : 0;

        // ---- phase 1: filter ------------------------------------------------------------------------------
        // wave priorities, graded by how latency-bound a stage is: filter 0 < list build 1 < stage A 2 < stage B and flush 3.
        // A wave that waits for L2 in the verification stages issues the moment its data arrives; the other waves' filter
        // loops fill the remaining slots.
        if (P.prio) __builtin_amdgcn_s_setprio(0);
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (own) {
            // running window key: x(i) = pair(i-2) * kp^2 + pair(i), pair(i) = class(i-1) * kp + class(i)
            uint32_t cp = 0, pm1 = 0, pm2 = 0;
            const uint8_t* src = c.dbase + my_lo;
            U128u nxt{0, 0, 0, 0};
            if (nvalid) {
                // the three bytes in front of the lane's range (one unaligned dword) and the first piece, together
                uint32_t hist = 0;
                if (doc_abs + my_lo >= 4) hist = load_u32_unaligned(src - 4);
                else for (uint32_t i = 1; i <= 3 && i <= doc_abs + my_lo; i++) hist |= (uint32_t)src[-(int)i] << (32 - 8 * i);
                nxt = *reinterpret_cast<const U128u*>(src);
                const uint32_t k1 = my_lo >= 1 ? lcls[hist >> 24] : P.pad_class, k2 = my_lo >= 2 ? lcls[(hist >> 16) & 0xFF] : P.pad_class,
                               k3 = my_lo >= 3 ? lcls[(hist >> 8) & 0xFF] : P.pad_class;
                cp = k1; pm1 = mad24s(k2, kp, k1); pm2 = mad24s(k3, kp, k2);
            }
            mark(0);
            uint32_t acc = 0, hib = 0;                           // hib: OR of the lane's text (a byte >= 0x80 anywhere?)
            const uint32_t ndw = C >> 2;                         // dwords per lane (wave-uniform, <= 32)
            const uint32_t npieces = (ndw + 3) >> 2;
            // the 16-byte piece q+1 is in flight while piece q is filtered
            for (uint32_t q = 0; q < npieces; q++) {
                const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
                if (q * 16 < nvalid) hib |= (w[0] | w[1]) | (w[2] | w[3]);   // (may take in up to 15 bytes behind the lane's range: conservative)
                if (q + 1 < npieces && (q + 1) * 16 < nvalid) nxt = *reinterpret_cast<const U128u*>(src + (q + 1) * 16);
                const uint32_t nd = ndw - 4 * q;                 // dwords of this piece that belong to the lane (>= 1)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    if ((uint32_t)d < nd) {
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const uint32_t cl = lcls[(w[d] >> (8 * b)) & 0xFF];
                            const uint32_t pair = mad24s(cp, kp, cl);
                            const uint32_t x = mad24s(pm2, kp2, pair);
                            pm2 = pm1; pm1 = pair; cp = cl;
                            const uint32_t fi = HASHED ? (x * kGoldDev) >> P.hash_shift : x;
                            const uint32_t fw = lfilt[fi >> 5];
                            acc = __builtin_amdgcn_alignbit(fw >> (fi & 31), acc, 1);
                        }
                    }
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
            // ASCII folding is not strings.ToLower once the text leaves ASCII (finder.go:140-142): tell the host
            if (P.fold && P.nonascii && !told_nonascii && __any((hib & 0x80808080u) != 0)) {
                // (one global atomic per workgroup: bit 31 of its "waves done" word says that somebody has told already)
                told_nonascii = true;
                if (lane == 0 && !(__hip_atomic_fetch_or(wg_next + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 31))
                    atomicOr(P.nonascii, 1u);
            }
            // positions past the lane's range carry garbage flags
            m0 = nvalid >= 32 ? m0 : (nvalid ? m0 & ((1u << nvalid) - 1) : 0);
            m1 = nvalid >= 64 ? m1 : (nvalid > 32 ? m1 & ((1u << (nvalid - 32)) - 1) : 0);
            m2 = nvalid >= 96 ? m2 : (nvalid > 64 ? m2 & ((1u << (nvalid - 64)) - 1) : 0);
            m3 = nvalid >= 128 ? m3 : (nvalid > 96 ? m3 & ((1u << (nvalid - 96)) - 1) : 0);
        }

        mark(1);
        if (P.prio) __builtin_amdgcn_s_setprio(1);
        if (more_units) abs_n = P.doc_off[un_n.doc];

        // ---- phase 2: verify flagged positions, stage matches in LDS ----------------------------------------
        uint32_t cnt = 0;
        if (DBG && P.dbg) {   // timing studies (GFT_SCAN_DEBUG)
            if (P.dbg & 2) {
                uint32_t f = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
                uint32_t mx = f;
                for (int s = 32; s; s >>= 1) { f += __shfl_xor(f, s, 64); uint32_t o = __shfl_xor(mx, s, 64); mx = o > mx ? o : mx; }
                if (lane == 0) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters)), (unsigned long long)f);
                    atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 1), (unsigned long long)mx);
                }
            }
            if (P.dbg & 1) m0 = m1 = m2 = m3 = 0;
        }
        // ---- phase 2, unordered fast path (the solver does not need text order): balance the flagged positions over
        // the lanes through an LDS candidate list, append matches to an LDS fifo, flush the fifo coalesced ----------------
        if (!ORDERED) {
            const uint32_t f = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3) + (lane == 0 ? nborder : 0);
            const uint32_t fincl = wave_incl_scan(f);
            const uint32_t ftotal = lane_value(fincl, 63);
            bool done = ftotal == 0;
            if (ftotal) {
                uint32_t nf = 0;                                   // matches in the fifo (wave-uniform)
                // passes over lane ranges whose flagged positions fit the LDS list (one pass for a typical unit)
                for (uint32_t l0 = 0; l0 < 64;) {
                    const uint32_t before = l0 ? lane_value(fincl, l0 - 1) : 0;
                    const bool fits = lane >= l0 && fincl - before <= P.cand_cap;
                    const uint64_t fm = __ballot(fits) >> l0;
                    const uint32_t nl = fm == ~0ull >> l0 ? 64 - l0 : (uint32_t)__builtin_ctzll(~fm);   // lanes in this pass (>= 1)
                    const uint32_t l1 = l0 + nl;
                    const uint32_t ptotal = lane_value(fincl, l1 - 1) - before;
                    if (lane >= l0 && lane < l1) {
                        uint32_t wpos = fincl - f - before;
                        const uint32_t rel = lane * C + kScan2MaxOff;
                        if (lane == 0)
                            for (uint32_t i = 0; i < nborder; i++) cand[wpos++] = (uint16_t)(kScan2MaxOff - nborder + i);
                        uint32_t mm[4] = {m0, m1, m2, m3};
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
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // stage A: every flagged position -> LDS-only decisions; short terms are emitted here, positions that may
                    // end a term of length >= 4 are compacted in place to the front of the list (write index <= read index)
                    // (kStageAWays candidates per lane and trip: their loads and table lookups are in flight together;
                    // lanes past the end of the list work on a copy of entry 0 and stay silent)
                    mark(2);
                    if (P.prio) __builtin_amdgcn_s_setprio(2);
                    uint32_t ns = 0;
                    // (the list entries and text of trip t + 1 are fetched while trip t is worked on)
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
                        for (int q = 0; q < kStageAWays; q++) n_tx[q] = cand_load(c, ubase + n_rel[q]);
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
                        for (int q = 0; q < kStageAWays; q++) cand_keys(c, ubase + rel[q], tx[q], k[q]);
#pragma unroll
                        for (int q = 0; q < kStageAWays; q++) cand_decide<FPT_LDS>(c, k[q]);
#pragma unroll
                        for (int q = 0; q < kStageAWays; q++)
                            if (i0 + 64 * q < ptotal)              // (positions in front of the unit: long terms only)
                                finish_short(c, k[q].p, on[q] && rel[q] >= kScan2MaxOff ? k[q].sid : 0, k[q].x3, fifo, nf);
                        // all reads of this trip are done (every read index >= every write index below)
                        const uint64_t below = (1ull << lane) - 1;
#pragma unroll
                        for (int q = 0; q < kStageAWays; q++) {
                            const bool keep = on[q] && k[q].go_long;
                            const uint64_t sb = __ballot(keep);
                            if (keep) cand[ns + __popcll(sb & below)] = (uint16_t)rel[q];
                            ns += (uint32_t)__popcll(sb);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (DBG && (P.dbg & 2)) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 2), (unsigned long long)ns); }
                    mark(3);
                    if (P.prio) __builtin_amdgcn_s_setprio(3);
                    // stage B: the survivors, densely packed over the lanes, go to the L2 bucket table; the room behind
                    // them in the candidate list parks the entries of multi-term buckets
                    Deferred dfr;
                    dfr.list = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(cand) + ((ns * 2 + 7) & ~7u));
                    dfr.cap = (P.cand_cap * 2 - ((ns * 2 + 7) & ~7u)) / 8;
                    dfr.n = 0;
                    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
                        const bool on = i0 + lane < ns;
                        const uint32_t rel = cand[on ? i0 + lane : 0];
                        const uint32_t p = ubase + rel;
                        const Text8 t8 = cand_load(c, p);
                        const Front fr = front_load(c, p, t8.tw);
                        const uint32_t tl = tail_load(c, p);
                        Cand k;
                        cand_keys<false>(c, p, t8, k);
                        const Slot s0 = slot_load(&P.slots[scan2_pair_slot(k.x, 0, P.slot_shift, P.slot_seed)]);
                        const Slot s1 = slot_load(&P.slots[scan2_pair_slot(k.x, 1, P.slot_shift, P.slot_seed)]);
                        finish_long(c, on, rel, k, s0, s1, fr, tl, fifo, nf, dfr);
                    }
                    if (dfr.n) drain_deferred(c, ubase, fifo, nf, dfr);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    l0 = l1;
                    mark(4);
                }
                const uint32_t nh = nf;
                if (nh <= kScan2FifoCap) {
                    if (nh > slab_left) {
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
                    if (lane == 0) { KARG(unit_start)[u] = base; KARG(unit_count)[u] = base + nh <= KARG(pool_cap) ? nh : 0u; }   // (beyond the pool: nothing was written, the host runs the batch again)
                    if (base + nh <= KARG(pool_cap))
                        for (uint32_t i = lane; i < nh; i += 64) {
                            const uint2 r = fifo[i];
                            // (streaming stores: the pool is read by the NEXT kernel, its lines should not push the units'
                            // text out of L2 before the verification stages have re-read it)
                            __builtin_nontemporal_store(r.x, &KARG(pool_term)[base + i]);
                            if (P.want_pos) __builtin_nontemporal_store(r.y, &KARG(pool_pos)[base + i]);
                        }
                    done = true;
                }
                __builtin_amdgcn_wave_barrier();
                mark(5);
            }
            if (ftotal == 0 && lane == 0) { KARG(unit_start)[u] = slab_next; KARG(unit_count)[u] = 0; }
            if (done) continue;
        }

        // ---- phase 2, ordered path: every lane verifies its own positions in text order, stages matches in LDS ---------
        const uint32_t mb = lane == 0 ? ((1u << nborder) - 1) << (kScan2MaxOff - nborder) : 0;
        verify_masks<0, FPT_LDS>(c, my_lo, mb, m0, m1, m2, m3, cnt, stage, 0);

        // ---- output -------------------------------------------------------------------------------------------------
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t total = lane_value(incl, 63);
        if (total > slab_left) {          // wave-uniform: take a new slab
            const uint32_t want = total > KARG(slab) ? total : KARG(slab);
            uint64_t nb = 0;
            if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)want);
            slab_next = static_slabs + __shfl(nb, 0, 64);
            slab_left = want;
        }
        const uint64_t base = slab_next;
        slab_next += total;
        slab_left -= total;
        wave_matches += total;
        if (lane == 0) { KARG(unit_start)[u] = base; KARG(unit_count)[u] = base + total <= KARG(pool_cap) ? total : 0u; }
        if (total && base + total <= KARG(pool_cap)) {
            const uint64_t mine = base + incl - cnt;
            if (cnt <= kScan2StageCap) {
#pragma unroll
                for (uint32_t k = 0; k < kScan2StageCap; k++)
                    if (k < cnt) {
                        const uint2 r = stage[k * 64];
                        KARG(pool_term)[mine + k] = r.x;
                        if (P.want_pos) KARG(pool_pos)[mine + k] = r.y;
                    }
            }
            // lanes whose matches did not fit the staging area run the verification again, writing directly
            const bool over = cnt > kScan2StageCap;
            if (__any(over)) {
                uint32_t c2 = 0;
                verify_masks<1, FPT_LDS>(c, my_lo, over ? mb : 0, over ? m0 : 0, over ? m1 : 0, over ? m2 : 0, over ? m3 : 0, c2, stage, mine);
            }
        }
    }
    // the match count: summed in LDS, one global atomic per workgroup by the wave that finishes last
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
        atomicMax(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 12), all);      // the wave that finishes last
        atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 13), 1ull);
    }
}

}  // namespace

static size_t scan2_fixed_lds(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes) {
    return ((256 + (size_t)filter_words * 4 + short3_bytes + fpt_lds_bytes + (size_t)shorts_words * 4 + 15) & ~(size_t)15) +
           16;       // the workgroup's bookkeeping (aligned)
}

bool scan2_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max,
                uint32_t* waves, uint32_t* cand_cap) {
    const size_t fixed = scan2_fixed_lds(filter_words, short3_bytes, shorts_words, fpt_lds_bytes);
    for (uint32_t w : {16u, 12u, 8u, 4u}) {
        if (fixed + (size_t)w * (kScan2FifoCap * 8 + kScan2CandCapMin * 2) > lds_max) continue;
        size_t per = ((lds_max - fixed) / w) & ~(size_t)15;
        size_t cap = (per - kScan2FifoCap * 8) / 2;
        *waves = w;
        *cand_cap = (uint32_t)(cap > 4096 ? 4096 : cap);
        return true;
    }
    return false;
}

hipError_t launch_scan2(const Scan2Params& P, uint32_t waves, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const bool fl = P.fpt_lg == 0;
    const size_t lds = scan2_fixed_lds(P.filter_words, P.short3_bytes, P.shorts_words, fl ? kScan2FptSize : 0) +
                       (size_t)waves * (kScan2FifoCap * 8 + P.cand_cap * 2);
    using Kern = void (*)(const Scan2Params);
    static const Kern table[2][2][2] = {
        {{k_scan2<false, false, false, false>, k_scan2<false, false, true, false>}, {k_scan2<false, true, false, false>, k_scan2<false, true, true, false>}},
        {{k_scan2<true, false, false, false>, k_scan2<true, false, true, false>}, {k_scan2<true, true, false, false>, k_scan2<true, true, true, false>}}};
    // timing studies: the benchmark's shape only (direct filter, balanced path)
    const Kern fn = P.dbg && !P.hashed && !P.ordered ? (fl ? k_scan2<false, false, true, true> : k_scan2<false, false, false, true>)
                                                     : table[P.hashed ? 1 : 0][P.ordered ? 1 : 0][fl ? 1 : 0];
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t g = (P.n_units + waves - 1) / waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    fn<<<dim3(grid), dim3(waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
