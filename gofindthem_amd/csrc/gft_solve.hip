// gft_solve.hip -- the solver half of ProcessText on gfx950: addMatchesToSolverMap + solveExpressions
// (finder/finder.go:181-215) and Expression.solve (dsl/expression.go:66-142) as ONE data-parallel kernel.
//
// Bit-sliced evaluation over groups of 64 documents.  One workgroup owns a group:
//   1. presence matrix P[slot] = 64-bit mask "which of my 64 documents contain this slot", built in LDS from the
//      scan kernel's match slabs with ds_or (UNIT truth == key presence, dsl/expression.go:68-72);
//   2. every lane interprets ONE expression over 64-bit masks, so each AND/OR/NOT evaluates 64 documents at once
//      (the reference evaluates every node, no short-circuit, so this is the same function, expression.go:74-127);
//      programs are fused at gft_set_programs time (gft_api.cpp fuse_program), staged in LDS and handed to the waves
//      sorted by length;
//   3. INORD(...) groups: the boolean value of the group's subtree gives the candidate documents; only for those
//      the position algebra runs, per document, on (slot, theta) pairs with successor queries;
//   4. the 64 x 64 result tile of a wave is transposed (six masked exchange steps) so that lane d holds the two
//      bitmap words of document d, staged in LDS and written out as full rows.
// No MFMA: boolean algebra on bit masks; LDS- and latency-bound.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "gft_kernels.hpp"

namespace gft {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// A wave-uniform word of a read-only table through the scalar unit (constant address space: s_load, counted by lgkmcnt).
// As a plain global load it would be a VECTOR load plus v_readfirstlane -- and its s_waitcnt vmcnt(0) would wait for every
// prefetch in flight as well.
__device__ __forceinline__ uint32_t uniform_word(const uint32_t* table, size_t i) {
    using ConstWord = const __attribute__((address_space(4))) uint32_t;
    return reinterpret_cast<ConstWord*>(reinterpret_cast<uintptr_t>(table))[i];
}

// ---- per-document view of the matches (slabs of the scan kernel + caller-supplied matches) -------------------
struct DocHits {
    const uint64_t* unit_start;
    const uint32_t* unit_count;
    const uint32_t* term;
    const uint32_t* pos;
    uint64_t u0, u1;            // units of this document
    const uint32_t* xslot;
    const uint32_t* xpos;
    uint32_t nx;
    uint32_t per;               // size of the document's slices (units), for a document of kUnitsPerLaneMode units or more
    uint32_t back;              // a match's position is at most this far in front of the slice it ends in (0: positions ARE ends)
};

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;      // pair that never matches: the empty position list
constexpr uint32_t kUnitsPerLaneMode = 8;      // documents with at least this many units: one unit per lane

__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
    // Every caller's values are match positions (32 bits, below 2^32 - 1: a document is at most 4 GiB - 1 bytes) or INT64_MAX
    // ("none"): the reduction runs on 32 bits with DPP moves -- running minimum along the four rows of 16 lanes, row results
    // broadcast into the rows behind them, lane 63 ends up with the minimum of all -- instead of twelve xor shuffles through
    // the LDS permute network (round 3), which cost as much as the loads they follow
    uint32_t x = v == INT64_MAX ? 0xFFFFFFFFu : (uint32_t)v;
    auto step = [](uint32_t y, auto ctrl, auto rows) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)y, decltype(ctrl)::value, decltype(rows)::value, 0xF, false);
        return o < y ? o : y;
    };
    x = step(x, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{});      // row_shr:1
    x = step(x, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{});      // row_shr:2
    x = step(x, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{});      // row_shr:4
    x = step(x, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{});      // row_shr:8
    x = step(x, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});      // row_bcast:15 -> rows 1 and 3
    x = step(x, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});      // row_bcast:31 -> rows 2 and 3
    const uint32_t m = __builtin_amdgcn_readlane(x, 63);
    return m == 0xFFFFFFFFu ? INT64_MAX : (int64_t)m;
}
__device__ __forceinline__ int64_t readlane_i64(int64_t v, uint32_t l) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// The pairs (slot, theta) of an INORD group live one per lane (pair i in lane i; kMaxPairs == 64).  All 64 lanes call
// this together: min over the pairs [pb, pb + pc) of succ(slot, theta) = the first position of `slot` that is > theta
// (INT64_MAX if none).  The document's matches are read ONCE per call, spread over the lanes: documents of many units
// (a 1 MB document is > 100 units) give every lane whole units, small documents are strided inside each unit.  A
// keyword and a regex with the same literal share one map key (finder/finder.go:181-196): the caller maps both onto
// one slot, both lists are read.
__device__ int64_t wave_succ_min(const DocHits& M, uint32_t my_slot, int64_t my_theta, uint32_t pb, uint32_t pc) {
    const uint32_t lane = lane_id();
    int64_t best = INT64_MAX;
    const uint32_t sl0 = __builtin_amdgcn_readlane(my_slot, pb);
    const int64_t th0 = readlane_i64(my_theta, pb);
    // the empty list (the single pair that no match satisfies): nothing to walk -- what is left of a chain behind a term
    // that does not occur costs nothing (benchmarks/benchmark_test.go:438-462 builds chains of 10 000 terms)
    if (pc == 1 && sl0 == kNoSlot) return INT64_MAX;
    auto test = [&](uint32_t t, uint32_t p) {
        if (t == sl0 && (int64_t)p > th0 && (int64_t)p < best) best = p;
        for (uint32_t k = 1; k < pc; k++) {
            const uint32_t sl = __builtin_amdgcn_readlane(my_slot, pb + k);
            const int64_t th = readlane_i64(my_theta, pb + k);
            if (t == sl && (int64_t)p > th && (int64_t)p < best) best = p;
        }
    };
    if (M.u1 - M.u0 >= kUnitsPerLaneMode) {
        // A long document: its units are equal slices in position order, a unit holds the matches that END in its slice.
        // Nothing at or below the smallest threshold can answer, so the walk starts at the unit that holds that position
        // -- and it stops as soon as no later unit can hold a smaller answer (a match of a later unit ends behind this
        // one's slice and starts at most max_term_len - 1 bytes earlier): a chain of k queries over n matches walks
        // O(n + k x one unit) entries, not k x n.
        int64_t thmin = th0;
        for (uint32_t k = 1; k < pc; k++) { const int64_t th = readlane_i64(my_theta, pb + k); thmin = th < thmin ? th : thmin; }
        const uint64_t per = M.per;                                                   // (all but the last slice have this size)
        const uint64_t k0 = thmin < 0 ? 0 : (uint64_t)thmin / per;                    // the slice that holds position thmin
        uint64_t ub = M.u0 + (k0 < M.u1 - M.u0 ? k0 : M.u1 - M.u0 - 1);
        int64_t sofar = INT64_MAX;
        // From that unit on FOUR units at a time, sixteen lanes per unit, four of a unit's matches in flight per lane: a term
        // of a rule set recurs every few units, so the answer is in the first step or the second -- and a step is a unit's
        // matches / 64 round trips, not a unit's matches / 8 as with one unit per lane (round 3: 13 000 cycles per query of the
        // reference benchmark's 45-term chain, `profiles/r4_c1_solver_phase_clocks.txt`; the chain IS the call's time)
        while (ub < M.u1) {
            const int64_t next_lo = (int64_t)((ub - M.u0) * per);                      // positions of the next unit's slice begin here
            if (sofar != INT64_MAX && next_lo - (int64_t)M.back > sofar) break;
            const uint64_t u = ub + (lane >> 4);
            const uint32_t member = lane & 15u;
            if (u < M.u1) {
                const uint64_t s = M.unit_start[u];
                const uint32_t n = M.unit_count[u];
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    uint32_t t[4], p[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { const uint32_t i = i0 + 16 * q + member; t[q] = i < n ? M.term[s + i] : kNoSlot - 1; p[q] = i < n ? M.pos[s + i] : 0u; }
#pragma unroll
                    for (int q = 0; q < 4; q++) test(t[q], p[q]);
                }
            }
            ub += 4;
            sofar = wave_min_i64(best);
        }
    } else {
        for (uint64_t u = M.u0; u < M.u1; u++) {
            const uint64_t s = M.unit_start[u];
            const uint32_t n = M.unit_count[u];
            for (uint32_t i = lane; i < n; i += 64) test(M.term[s + i], M.pos[s + i]);
        }
    }
    for (uint32_t i = lane; i < M.nx; i += 64) test(M.xslot[i], M.xpos[i]);
    return wave_min_i64(best);
}

// Position algebra of one INORD group for one document (dsl/expression.go:87-95,111-116,129-137), evaluated by a whole
// wave.  `prog` is the group's subtree in the public postfix form (UNIT/AND/OR words carrying GFT_K_INORD_FLAG); control
// flow is wave-uniform.  Returns len(rpos) > 0.
//   UNIT t    -> {(t, -1)}
//   OR        -> union of the pair sets (only minimum and emptiness are ever observed, duplicates are harmless)
//   AND(L, R) -> m = min over L of succ(t, theta); {} if m = +inf, else {(t, max(theta, m)) : (t, theta) in R}
//                == rpos[getLowestIdxGTVal(rpos, lpos[0]):]   (expression.go:87-93,175-189)
// The operand stack is a sequence of adjacent, non-empty pair ranges [.., top): bit b of `starts` = a range starts at
// pair b.  The empty list is the single pair (kNoSlot, -1), which no match satisfies.
__device__ bool inord_group_wave(const uint32_t* __restrict__ prog, uint32_t len, const DocHits& M) {
    const uint32_t lane = lane_id();
    uint32_t my_slot = kNoSlot;
    int64_t my_theta = -1;
    uint64_t starts = 0;
    uint32_t top = 0;
    for (uint32_t pc = 0; pc < len; pc++) {
        const uint32_t w = prog[pc];
        const uint32_t op = w >> 28;
        if (op == 1) {                                           // UNIT
            if (lane == top) { my_slot = w & GFT_K_SLOT_MASK; my_theta = -1; }
            starts |= 1ull << top;
            top++;
        } else if (op == 2) {                                    // AND
            const uint32_t rb = 63u - (uint32_t)__builtin_clzll(starts);
            const uint64_t rest = starts & ~(1ull << rb);
            const uint32_t lb = 63u - (uint32_t)__builtin_clzll(rest);
            const uint32_t lc = rb - lb, rc = top - rb;
            const int64_t m = wave_succ_min(M, my_slot, my_theta, lb, lc);
            const uint32_t r_slot = __shfl(my_slot, (int)((lane + lc) & 63u), 64);   // R's pairs move down to lb
            const int64_t r_theta = __shfl(my_theta, (int)((lane + lc) & 63u), 64);
            if (m == INT64_MAX) {
                if (lane == lb) { my_slot = kNoSlot; my_theta = -1; }
                top = lb + 1;
            } else {
                if (lane >= lb && lane < lb + rc) { my_slot = r_slot; my_theta = r_theta < m ? m : r_theta; }
                top = lb + rc;
            }
            starts = rest;
        } else if (op == 3) {                                    // OR: the two ranges are adjacent, union == concatenation
            starts &= ~(1ull << (63u - (uint32_t)__builtin_clzll(starts)));
        }
    }
    if (!top) return false;
    const uint32_t b = 63u - (uint32_t)__builtin_clzll(starts);
    return wave_succ_min(M, my_slot, my_theta, b, top - b) != INT64_MAX;
}

// The same algebra for a group of more than kMaxPairs pairs alive at once (or a pair stack deeper than kMaxPairDepth): the
// pairs live in this wave's scratch region in HBM (ws / wt, SolveParams::wide_*) and pass through the lanes 64 at a time;
// the operand stack -- the first pair of every range -- is one entry per lane (kMaxPairDepthWide).  A rare shape (an INORD
// over an OR of dozens of terms): correctness and no host round trip matter here, not the last cycle; not inlined, so
// that the register allocation of the common path stays what it is.  The scratch region is this wave's alone; its lanes
// exchange pairs through it, so stores are made visible (workgroup scope: the CU's own L1) before other lanes load them.
__device__ __attribute__((noinline)) bool inord_group_wide(const uint32_t* __restrict__ prog, uint32_t len, const DocHits* Mp,
                                                            uint32_t* ws, long long* wt) {
    const DocHits& M = *Mp;
    const uint32_t lane = lane_id();
    uint32_t st = 0;                                             // lane k: first pair of stack entry k
    uint32_t sp = 0, top = 0;
    auto visible = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto range_min = [&](uint32_t b, uint32_t e) {               // min over the pairs [b, e) of succ(slot, theta)
        int64_t m = INT64_MAX;
        for (uint32_t c0 = b; c0 < e; c0 += 64) {
            const uint32_t n = e - c0 < 64u ? e - c0 : 64u;
            const uint32_t sl = lane < n ? ws[c0 + lane] : kNoSlot;
            const int64_t th = lane < n ? (int64_t)wt[c0 + lane] : -1;
            const int64_t r = wave_succ_min(M, sl, th, 0, n);
            m = r < m ? r : m;
        }
        return m;
    };
    uint32_t wl = 0;                                             // the group's words, 64 at a time, one per lane
    for (uint32_t pc = 0; pc < len; pc++) {
        if ((pc & 63u) == 0) wl = pc + lane < len ? prog[pc + lane] : 0u;
        const uint32_t w = __builtin_amdgcn_readlane(wl, pc & 63u);
        const uint32_t op = w >> 28;
        if (op == 1) {                                           // UNIT
            if (lane == 0) { ws[top] = w & GFT_K_SLOT_MASK; wt[top] = -1; }
            if (lane == sp) st = top;
            sp++; top++;
        } else if (op == 2) {                                    // AND
            const uint32_t rb = __builtin_amdgcn_readlane(st, sp - 1), lb = __builtin_amdgcn_readlane(st, sp - 2);
            const uint32_t rc = top - rb;
            visible();
            const int64_t m = range_min(lb, rb);
            if (m == INT64_MAX) {
                if (lane == 0) { ws[lb] = kNoSlot; wt[lb] = -1; }
                top = lb + 1;
            } else {
                // R's pairs move down to lb, thresholds raised to m (ascending chunks: a chunk's target lies below every
                // pair that is still to be read)
                for (uint32_t c0 = 0; c0 < rc; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    uint32_t sl = 0;
                    int64_t th = 0;
                    if (i < rc) { sl = ws[rb + i]; th = (int64_t)wt[rb + i]; }
                    visible();                                   // (every lane has its pair before a lane overwrites one)
                    if (i < rc) { ws[lb + i] = sl; wt[lb + i] = th < m ? m : th; }
                }
                top = lb + rc;
            }
            sp--;                                                // (entry sp - 1 keeps its start, lb)
        } else if (op == 3) {                                    // OR: adjacent ranges, union == concatenation
            sp--;
        }
    }
    if (!top || !sp) return false;
    visible();
    return range_min(__builtin_amdgcn_readlane(st, sp - 1), top) != INT64_MAX;
}

// One expression with a WIDE INORD group for ONE document, by a whole wave, from the expression's public postfix words
// (dsl/expression.go:66-142 in one pass: every word moves the boolean stack, the words inside an INORD group -- they carry
// GFT_K_INORD_FLAG -- move the pair stack of inord_group_wave as well, and the group's INORD word joins the two:
// rval && len(rpos) > 0, :137).  What makes a wide group fit a wave's 64 lanes after all: of an OR over hundreds of terms a
// document holds a handful, and the presence matrix says which.  The words pass through the lanes 128 at a time; every
// UNIT lane looks its slot up in the presence matrix (Pw: LDS or HBM, element slot * G + j for document j of the group); an
// absent UNIT that an OR follows is dropped together with that OR (X or {} == X, for truth values and for position lists
// alike), ranges that are the empty list (the dummy pair) are remembered (`empt`, by their first pair) and vanish in an OR:
// the pairs alive are the PRESENT leaves plus a dummy per empty operand.  A group that still has more than 64 alive (a
// document that holds dozens of its terms) is answered by inord_group_wide's scratch path when its INORD word comes.
// The boolean stack is a bit per entry (gft_set_programs sends deeper expressions to the host).
__device__ __forceinline__ bool wide_expr_doc(const uint32_t* __restrict__ prog, uint32_t len, const DocHits& M, const uint32_t* Pw,
                                              uint32_t G, uint32_t j, uint32_t* ws, long long* wt) {
    const uint32_t lane = lane_id();
    uint32_t my_slot = kNoSlot;
    int64_t my_theta = -1;
    uint64_t starts = 0, empt = 0, bst = 0;                       // bst: the boolean stack, entry k = bit k
    uint32_t top = 0, bsp = 0;
    bool ovf = false;                                            // this group outgrew the lanes: pairs are not tracked any more
    auto fetch = [&](uint32_t at) { return at + lane < len ? prog[at + lane] : 0u; };
    uint32_t n0 = fetch(0), n1 = fetch(64);
    for (uint32_t c0 = 0; c0 < len; c0 += 128) {
        const uint32_t w2[2] = {n0, n1};
        n0 = fetch(c0 + 128); n1 = fetch(c0 + 192);              // (the next round's words travel while this one is interpreted)
        bool pres[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t bp = (w2[h] & GFT_K_SLOT_MASK) * G + j;
            // (agent scope: a presence matrix in HBM was built by other waves' atomics; for the LDS one this is a plain read)
            pres[h] = (w2[h] >> 28) == 1 && ((__hip_atomic_load(&Pw[bp >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bp & 31)) & 1u);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t base = c0 + 64 * h;
            if (base >= len) break;
            const uint32_t wl = w2[h];
            const uint64_t mU = __ballot((wl >> 28) == 1), mP = __ballot(pres[h]), mO = __ballot((wl >> 28) == 3);
            uint64_t skip = (mU & ~mP) & (mO >> 1);              // an absent UNIT and the OR behind it
            skip |= skip << 1;
            uint64_t todo = ~skip & (len - base >= 64 ? ~0ull : (1ull << (len - base)) - 1);
            while (todo) {
                const uint32_t i = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t w = __builtin_amdgcn_readlane(wl, i);
                const uint32_t op = w >> 28;
                const bool fl = (w & GFT_K_INORD_FLAG) != 0;
                if (op == 1) {                                   // UNIT
                    const bool here = (mP >> i) & 1;
                    bst = (bst & ~(1ull << bsp)) | ((uint64_t)here << bsp);
                    bsp++;
                    if (fl && !ovf) {
                        if (top >= 64) ovf = true;
                        else {
                            if (lane == top) { my_slot = here ? (w & GFT_K_SLOT_MASK) : kNoSlot; my_theta = -1; }
                            starts |= 1ull << top;
                            if (!here) empt |= 1ull << top;
                            top++;
                        }
                    }
                } else if (op == 2 || op == 3) {
                    const uint64_t b = (bst >> (bsp - 1)) & 1, a = (bst >> (bsp - 2)) & 1, r = op == 2 ? (a & b) : (a | b);
                    bsp--;
                    bst = (bst & ~(1ull << (bsp - 1))) | (r << (bsp - 1));
                    if (fl && !ovf) {
                        const uint32_t rb = 63u - (uint32_t)__builtin_clzll(starts);
                        const uint64_t rest = starts & ~(1ull << rb);
                        const uint32_t lb = 63u - (uint32_t)__builtin_clzll(rest);
                        const uint32_t lc = rb - lb, rc = top - rb;
                        const bool le = (empt >> lb) & 1, re = (empt >> rb) & 1;
                        empt &= ~((1ull << lb) | (1ull << rb));
                        if (op == 3) {                           // OR: union == concatenation, empty operands vanish
                            if (re) { top = rb; if (le) empt |= 1ull << lb; }
                            else if (le) {                       // (the dummy at lb goes: R moves down by one)
                                const uint32_t r_slot = __shfl(my_slot, (int)((lane + 1) & 63u), 64);
                                const int64_t r_theta = __shfl(my_theta, (int)((lane + 1) & 63u), 64);
                                if (lane >= lb && lane < lb + rc) { my_slot = r_slot; my_theta = r_theta; }
                                top = lb + rc;
                            }
                        } else {                                 // AND
                            int64_t m = INT64_MAX;
                            if (!le && !re) m = wave_succ_min(M, my_slot, my_theta, lb, lc);
                            const uint32_t r_slot = __shfl(my_slot, (int)((lane + lc) & 63u), 64);
                            const int64_t r_theta = __shfl(my_theta, (int)((lane + lc) & 63u), 64);
                            if (m == INT64_MAX) {
                                if (lane == lb) { my_slot = kNoSlot; my_theta = -1; }
                                top = lb + 1;
                                empt |= 1ull << lb;
                            } else {
                                if (lane >= lb && lane < lb + rc) { my_slot = r_slot; my_theta = r_theta < m ? m : r_theta; }
                                top = lb + rc;
                            }
                        }
                        starts = rest;
                    }
                } else if (op == 4) {                            // NOT (never inside a group: the parser rejects it)
                    bst ^= 1ull << (bsp - 1);
                } else if (op == 5) {                            // INORD: rval && len(rpos) > 0
                    bool some = false;
                    if ((bst >> (bsp - 1)) & 1) {                // (a false rval needs no positions)
                        if (ovf) {
                            // the group's words: the flagged run that ends in front of this word
                            const uint32_t pc = base + i;
                            uint32_t gs = pc;
                            while (gs > 0 && (prog[gs - 1] & GFT_K_INORD_FLAG) != 0 && (prog[gs - 1] >> 28) != 5) gs--;
                            some = ws != nullptr && inord_group_wide(prog + gs, pc - gs, &M, ws, wt);
                        } else if (top) {
                            const uint32_t gb = 63u - (uint32_t)__builtin_clzll(starts);
                            some = !((empt >> gb) & 1) && wave_succ_min(M, my_slot, my_theta, gb, top - gb) != INT64_MAX;
                        }
                    }
                    bst = (bst & ~(1ull << (bsp - 1))) | ((uint64_t)some << (bsp - 1));
                    top = 0; starts = 0; empt = 0; ovf = false;
                }
            }
        }
    }
    return bsp == 1 && (bst & 1);
}

// 64 x 64 bit-matrix transpose across a wave: lane i holds row i; afterwards lane j holds column j (bit i = old row
// i's bit j).  Six exchange steps with the partner lane i ^ s, swapping the off-diagonal s x s blocks -- all in the
// vector ALU (no LDS permutes, the LDS pipe is busy enough here):
//   s = 32: the low word of the upper lanes against the high word of the lower lanes IS v_permlane32_swap;
//   s < 32, per 32-bit half h with the partner's t:  new = keep ? h : rot(t), bit by bit, where the lower lane of a pair
//           keeps the bits of m = the low s of every 2s and takes t << s, the upper lane keeps ~m and takes t >> s; both
//           shifts as rotations (v_alignbit), what wraps around lands on kept bits.
//   partner fetch: v_permlane16_swap (s = 16), DPP row_ror:8 (s = 8), row_shl:4 / row_shr:4 by bank (s = 4), quad_perm (2, 1).
template <int SH>
__device__ __forceinline__ uint32_t partner(uint32_t h, uint32_t lane) {
    if constexpr (SH == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(h, h, false, false);   // [0]: odd rows <- even rows, [1]: even rows <- odd rows
        return (lane & 16) ? r[0] : r[1];
    } else if constexpr (SH == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x128, 0xF, 0xF, false);               // row_ror:8
    } else if constexpr (SH == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)h, 0x104, 0xF, 0x5, false);                  // row_shl:4 -> lanes 0-3, 8-11
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)h, 0x114, 0xF, 0xA, false);               // row_shr:4 -> lanes 4-7, 12-15
    } else if constexpr (SH == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x4E, 0xF, 0xF, false);                // quad_perm:[2,3,0,1]
    } else {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0xB1, 0xF, 0xF, false);                // quad_perm:[1,0,3,2]
    }
}
template <int SH>
__device__ __forceinline__ void transpose_step(uint32_t& lo, uint32_t& hi, uint32_t lane, uint32_t m) {
    const bool upper = (lane & SH) != 0;
    const uint32_t keep = upper ? ~m : m, rot = upper ? (uint32_t)SH : 32u - (uint32_t)SH;             // rotate right by `rot`
    const uint32_t tl = partner<SH>(lo, lane), th = partner<SH>(hi, lane);
    lo = (keep & lo) | (~keep & __builtin_amdgcn_alignbit(tl, tl, rot));
    hi = (keep & hi) | (~keep & __builtin_amdgcn_alignbit(th, th, rot));
}
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x) {
    const uint32_t lane = lane_id();
    const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)(x >> 32), false, false);
    uint32_t lo = r[0], hi = r[1];
    transpose_step<16>(lo, hi, lane, 0x0000FFFFu);
    transpose_step<8>(lo, hi, lane, 0x00FF00FFu);
    transpose_step<4>(lo, hi, lane, 0x0F0F0F0Fu);
    transpose_step<2>(lo, hi, lane, 0x33333333u);
    transpose_step<1>(lo, hi, lane, 0x55555555u);
    return ((uint64_t)hi << 32) | lo;
}

// INORD words of a wave's current program step.  All 64 lanes call this together; `is_inord` marks the lanes whose word
// closes a group (operand `grp`), `cand` their candidate documents (bit j = document d0 + j: the documents where the
// group's boolean value is true, rval of expression.go:137).  The (lane, document) pairs are taken one after the other
// and each is evaluated by the whole wave, so a large document's matches are scanned 64 wide and a wave with few
// candidates does not leave 63 lanes idle.  Returns the documents whose position list is non-empty.
__device__ __forceinline__ uint64_t inord_wave(const SolveParams& S, bool is_inord, uint32_t grp, uint64_t cand, uint64_t d0) {
    const uint32_t lane = lane_id();
    uint64_t res = 0;
    uint64_t todo = __ballot(is_inord && cand != 0);
    while (todo) {
        const uint32_t L = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t g = __builtin_amdgcn_readlane(grp, L);
        uint64_t c = (uint64_t)readlane_i64((int64_t)cand, L);
        const uint32_t goff = S.groups[g * 2], glen = S.groups[g * 2 + 1];
        while (c) {
            const uint32_t j = (uint32_t)__builtin_ctzll(c);
            c &= c - 1;
            const uint64_t d = d0 + j;
            DocHits M;
            M.unit_start = S.unit_start; M.unit_count = S.unit_count;
            M.term = S.term; M.pos = S.pos;
            M.u0 = S.doc_unit_base[d]; M.u1 = S.doc_unit_base[d + 1];
            M.nx = 0; M.xslot = nullptr; M.xpos = nullptr;
            M.back = S.pos_back;
            M.per = 1;
            if (M.u1 - M.u0 >= kUnitsPerLaneMode) {                                   // (read once per document, not once per query)
                const Unit first = S.units[M.u0];
                const uint32_t per32 = __builtin_amdgcn_readfirstlane(first.hi - first.lo);
                M.per = per32 ? per32 : 1u;
            }
            if (S.x_off) {
                const uint64_t x0 = S.x_off[d];
                M.xslot = S.x_slot + x0; M.xpos = S.x_pos + x0; M.nx = (uint32_t)(S.x_off[d + 1] - x0);
            }
            const bool r = inord_group_wave(S.gprog + goff, glen, M);
            if (r && lane == L) res |= 1ull << j;
        }
    }
    return res;
}

// maximum over the 64 lanes with DPP moves: running maximum along the four rows of 16 lanes, row results broadcast into
// the rows behind them, lane 63 ends up with the maximum of all (no LDS permutes, no index arithmetic)
template <int CTRL, int ROWS, bool ZERO_FILL>
__device__ __forceinline__ uint32_t dpp_max_step(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, ZERO_FILL);
    return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = dpp_max_step<0x111, 0xF, true>(v);      // row_shr:1
    v = dpp_max_step<0x112, 0xF, true>(v);      // row_shr:2
    v = dpp_max_step<0x114, 0xF, true>(v);      // row_shr:4
    v = dpp_max_step<0x118, 0xF, true>(v);      // row_shr:8
    v = dpp_max_step<0x142, 0xA, false>(v);     // row_bcast:15 -> rows 1 and 3
    v = dpp_max_step<0x143, 0xC, false>(v);     // row_bcast:31 -> rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// 64 document bits as two 32-bit halves: the interpreter's bitwise steps are then plain 32-bit operations, which the
// compiler folds into v_bfi_b32 / v_and_or_b32 (it does not for 64-bit values)
struct Mask64 {
    uint32_t lo, hi;
    Mask64() = default;                                        // (no zero fill: the scratch stack of the deep interpreter)
    __device__ __forceinline__ Mask64(uint32_t l, uint32_t h) : lo(l), hi(h) {}
    __device__ __forceinline__ explicit Mask64(uint64_t v) : lo((uint32_t)v), hi((uint32_t)(v >> 32)) {}
    __device__ __forceinline__ explicit operator uint64_t() const { return ((uint64_t)hi << 32) | lo; }
    __device__ __forceinline__ explicit operator bool() const { return (lo | hi) != 0; }
};
__device__ __forceinline__ Mask64 operator&(Mask64 a, Mask64 b) { return Mask64(a.lo & b.lo, a.hi & b.hi); }
__device__ __forceinline__ Mask64 operator|(Mask64 a, Mask64 b) { return Mask64(a.lo | b.lo, a.hi | b.hi); }
__device__ __forceinline__ Mask64 operator^(Mask64 a, Mask64 b) { return Mask64(a.lo ^ b.lo, a.hi ^ b.hi); }
__device__ __forceinline__ Mask64 operator~(Mask64 a) { return Mask64(~a.lo, ~a.hi); }

// a control bit of a device word (gft_kernels.hpp kDw*) as a mask over the group's documents: v_bfe_i32 gives 0 or ~0
// (both halves of a Mask64 are that one register)
template <class AT, uint32_t BIT>
__device__ __forceinline__ AT bit_mask(uint32_t w) {
    const uint32_t m = (uint32_t)((int32_t)(w << (31u - BIT)) >> 31);
    if constexpr (sizeof(AT) == 8) return AT(m, m);
    else return (AT)m;
}
template <class AT>
__device__ __forceinline__ AT pick(AT m, AT a, AT b) { return (m & a) | (~m & b); }       // v_bfi_b32: m ? a : b, bit by bit
constexpr uint32_t bit_index(uint32_t m) { return m <= 1 ? 0 : 1 + bit_index(m >> 1); }

// One fused program over the group's documents (bit j of every mask = document d0 + j).  Lanes of a wave run different
// programs, so the interpreter has no branches on the word: the control bits of a device word become lane masks and one
// step is a handful of v_bfi / v_and_or on them plus one presence read (gft_kernels.hpp, "What the kernel reads").
// The accumulator stack lives in R registers -- none for the blocks of flat programs (no push / pop: half the work per
// word), 2 for the ordinary blocks, 4 for the blocks whose programs nest deeper, which also keep a depth counter and spill
// to scratch beyond (a wave-uniform slow path, like the NOT / INORD words).
// ALL 64 lanes of a wave call this together (lanes without a program pass chunks = 0): the trip count is the wave's
// maximum, finished lanes run no-op words, so the INORD steps can use the whole wave.
template <bool P_LDS, uint32_t R, int RARE, class PT, class AT>
__device__ __forceinline__ AT run_program(const SolveParams& S, const PT* P, const uint4* prog, uint32_t stride, uint32_t chunks,
                                          AT valid, uint64_t d0) {
    static_assert(R == 0 || R == kSolveRegStack || R == kSolveRegStackDeep, "three interpreters");
    constexpr bool DEEP = R > kSolveRegStack;
    // HBM-resident P was written with L2 atomics by other waves: read it past this CU's L1
    // (LDS: the field of a word is the slot's byte offset in a P of 8-byte elements, one v_and away)
    auto ld = [&](uint32_t w) -> AT {
        constexpr uint32_t kDown = sizeof(PT) == 8 ? 0 : sizeof(PT) == 4 ? 1 : sizeof(PT) == 2 ? 2 : 3;
        const uint32_t byte = (w & kDwFieldMask) >> kDown;
        // (P sits at LDS address 0 -- checked in the kernel --, so the byte offset IS the address: no add of the base)
        if (P_LDS) return (AT)*reinterpret_cast<const __attribute__((address_space(3))) PT*>((size_t)byte);
        return (AT)__hip_atomic_load(&P[byte / sizeof(PT)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    AT acc = AT(0), s0 = AT(0), s1 = AT(0), s2 = AT(0), s3 = AT(0), deep[DEEP ? kMaxBoolDepth : 1];
    uint32_t sp = 0;                                            // DEEP: entries on the stack (registers + scratch)
    uint4 nx = chunks ? prog[0] : make_uint4(kDwNop, kDwNop, kDwNop, kDwNop);
    uint4 nx2 = chunks > 1 ? prog[stride] : make_uint4(kDwNop, kDwNop, kDwNop, kDwNop);   // two chunks ahead (programs in L2)
    const uint32_t wchunks = wave_max_u32(chunks);
    // four words per trip: the next two chunks and this chunk's four presence reads are in flight together, so a trip
    // exposes one memory round trip instead of four (programs are padded to whole chunks with no-op words)
    for (uint32_t c = 0; c < wchunks; c++) {
        const uint32_t w[4] = {nx.x, nx.y, nx.z, nx.w};
        nx = nx2;
        nx2 = make_uint4(kDwNop, kDwNop, kDwNop, kDwNop);
        if (c + 2 < chunks) nx2 = prog[(size_t)(c + 2) * stride];
        AT pv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pv[q] = ld(RARE && (int32_t)w[q] < 0 ? 0u : w[q]);            // (a rare word's field is a group, not a slot)
        }
        // wave-uniform: a NOT / INORD word in this chunk, or a lane whose stack leaves the registers in it
        bool careful = RARE && __any((int32_t)(w[0] | w[1] | w[2] | w[3]) < 0);
        if (DEEP) {
            uint32_t high = sp;                                  // an upper bound of the depth inside this chunk
#pragma unroll
            for (int q = 0; q < 4; q++) high += (w[q] >> bit_index(kDwPush)) & 1u;
            careful |= __any(high > R);
        }
        auto step = [&](const uint32_t wq, const AT pvq, const bool care) __attribute__((always_inline)) {
            const AT neg = bit_mask<AT, bit_index(kDwNeg)>(wq), sel = bit_mask<AT, bit_index(kDwSel)>(wq);
            const AT ones = bit_mask<AT, bit_index(kDwOnes)>(wq), orr = bit_mask<AT, bit_index(kDwOr)>(wq);
            const AT pop = bit_mask<AT, bit_index(kDwPop)>(wq), push = bit_mask<AT, bit_index(kDwPush)>(wq);
            const AT v = pvq ^ neg;                             // (bits past the group are never stored)
            const AT x = R ? pick(pop, s0, v) : v;
            const AT before = acc;
            acc = (acc & pick(sel, x, ones)) | (x & orr);
            if (R == 0) {
                // flat programs: no push, no pop
            } else if (!DEEP) {
                const AT n0 = pick(push, before, pick(pop, s1, s0));
                s1 = pick(push, s0, s1);
                s0 = n0;
            } else {
                if (care && push && sp >= R) deep[sp - R] = s3;
                const AT n0 = pick(push, before, pick(pop, s1, s0)), n1 = pick(push, s0, pick(pop, s2, s1));
                const AT n2 = pick(push, s1, pick(pop, s3, s2)), n3 = pick(push, s2, s3);
                s0 = n0; s1 = n1; s2 = n2; s3 = n3;
                sp = sp - bit_mask<uint32_t, bit_index(kDwPush)>(wq) + bit_mask<uint32_t, bit_index(kDwPop)>(wq);   // 0 or ~0 == -1
                if (care && pop && sp >= R) {
                    s3 = deep[sp - R];
                    // (waited for here: a scratch load left pending would have the compiler wait for ALL vector loads, the
                    // prefetch of the next group included, wherever the fast path touches the same register)
                    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
                }
            }
            if (RARE && care) {
                const bool rare = (int32_t)wq < 0, is_not = rare && (wq & kDwNeg), is_inord = rare && !(wq & kDwNeg);
                if (is_not) acc = ~acc;
                // candidates: documents where the group's boolean value is true (rval, expression.go:137)
                const AT in = AT(inord_wave(S, is_inord, (wq & kDwFieldMask) >> kDwFieldShift, (uint64_t)(acc & valid), d0));
                if (is_inord) acc = in;
            }
        };
        if (!careful) {
#pragma unroll
            for (int q = 0; q < 4; q++) step(w[q], pv[q], false);
        } else {
            // one copy of the slow code: the chunk's words one after the other in a rolled loop.  The words rotate through
            // fixed registers -- indexing w[] / pv[] with the loop counter would move both arrays to scratch memory, for
            // the fast path above as well
            uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            AT p0 = pv[0], p1 = pv[1], p2 = pv[2], p3 = pv[3];
#pragma nounroll
            for (int q = 0; q < 4; q++) {
                step(w0, p0, true);
                w0 = w1; w1 = w2; w2 = w3;
                p0 = p1; p1 = p2; p2 = p3;
            }
        }
    }
    return acc;
}

// The same interpreter for programs that are read from global memory (a 100 000-term dictionary's do not fit LDS): FOUR
// chunks are requested in front of the one being interpreted, each in a register set of its own that is loaded again as soon
// as its words are copied out.  run_program's rotation (nx = nx2) copies a set that a load is still filling and so waits for
// that load: its distance is one trip, which does not cover an L2 round trip -- the longest program of a group was a chain of
// exposed round trips.  (configs[4]: solve 3.03 -> 2.89 ms with 50 % INORD expressions, 2.82 -> 2.60 without.  What remains
// is instruction issue: ~34 vector instructions per program word at 8 documents per group; leaving a block of INORD
// conjunctions as soon as no document is left in any accumulator removes 45 % of that work and not a microsecond -- the
// group waits for its longest program, 59 trips that one wave issues alone.)
template <bool P_LDS, uint32_t R, int RARE, class PT, class AT>
__device__ __forceinline__ AT run_program_far(const SolveParams& S, const PT* P, const uint4* prog, uint32_t stride, uint32_t chunks,
                                              AT valid, uint64_t d0) {
    static_assert(R == 0 || R == kSolveRegStack || R == kSolveRegStackDeep, "three interpreters");
    constexpr bool DEEP = R > kSolveRegStack;
    // HBM-resident P was written with L2 atomics by other waves: read it past this CU's L1
    // (LDS: the field of a word is the slot's byte offset in a P of 8-byte elements, one v_and away)
    auto ld = [&](uint32_t w) -> AT {
        constexpr uint32_t kDown = sizeof(PT) == 8 ? 0 : sizeof(PT) == 4 ? 1 : sizeof(PT) == 2 ? 2 : 3;
        const uint32_t byte = (w & kDwFieldMask) >> kDown;
        // (P sits at LDS address 0 -- checked in the kernel --, so the byte offset IS the address: no add of the base)
        if (P_LDS) return (AT)*reinterpret_cast<const __attribute__((address_space(3))) PT*>((size_t)byte);
        return (AT)__hip_atomic_load(&P[byte / sizeof(PT)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    AT acc = AT(0), s0 = AT(0), s1 = AT(0), s2 = AT(0), s3 = AT(0), deep[DEEP ? kMaxBoolDepth : 1];
    uint32_t sp = 0;                                            // DEEP: entries on the stack (registers + scratch)
    const uint4 nops = make_uint4(kDwNop, kDwNop, kDwNop, kDwNop);
    uint4 q0 = chunks > 0 ? prog[0] : nops, q1 = chunks > 1 ? prog[stride] : nops;
    uint4 q2 = chunks > 2 ? prog[2 * (size_t)stride] : nops, q3 = chunks > 3 ? prog[3 * (size_t)stride] : nops;
    const uint32_t wchunks = wave_max_u32(chunks);
    auto trip = [&](uint4& q, const uint32_t c) __attribute__((always_inline)) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        q = nops;
        if (c + 4 < chunks) q = prog[(size_t)(c + 4) * stride];
        AT pv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pv[q] = ld(RARE && (int32_t)w[q] < 0 ? 0u : w[q]);            // (a rare word's field is a group, not a slot)
        }
        // wave-uniform: a NOT / INORD word in this chunk, or a lane whose stack leaves the registers in it
        bool careful = RARE && __any((int32_t)(w[0] | w[1] | w[2] | w[3]) < 0);
        if (DEEP) {
            uint32_t high = sp;                                  // an upper bound of the depth inside this chunk
#pragma unroll
            for (int q = 0; q < 4; q++) high += (w[q] >> bit_index(kDwPush)) & 1u;
            careful |= __any(high > R);
        }
        auto step = [&](const uint32_t wq, const AT pvq, const bool care) __attribute__((always_inline)) {
            const AT neg = bit_mask<AT, bit_index(kDwNeg)>(wq), sel = bit_mask<AT, bit_index(kDwSel)>(wq);
            const AT ones = bit_mask<AT, bit_index(kDwOnes)>(wq), orr = bit_mask<AT, bit_index(kDwOr)>(wq);
            const AT pop = bit_mask<AT, bit_index(kDwPop)>(wq), push = bit_mask<AT, bit_index(kDwPush)>(wq);
            const AT v = pvq ^ neg;                             // (bits past the group are never stored)
            const AT x = R ? pick(pop, s0, v) : v;
            const AT before = acc;
            acc = (acc & pick(sel, x, ones)) | (x & orr);
            if (R == 0) {
                // flat programs: no push, no pop
            } else if (!DEEP) {
                const AT n0 = pick(push, before, pick(pop, s1, s0));
                s1 = pick(push, s0, s1);
                s0 = n0;
            } else {
                if (care && push && sp >= R) deep[sp - R] = s3;
                const AT n0 = pick(push, before, pick(pop, s1, s0)), n1 = pick(push, s0, pick(pop, s2, s1));
                const AT n2 = pick(push, s1, pick(pop, s3, s2)), n3 = pick(push, s2, s3);
                s0 = n0; s1 = n1; s2 = n2; s3 = n3;
                sp = sp - bit_mask<uint32_t, bit_index(kDwPush)>(wq) + bit_mask<uint32_t, bit_index(kDwPop)>(wq);   // 0 or ~0 == -1
                if (care && pop && sp >= R) {
                    s3 = deep[sp - R];
                    // (waited for here: a scratch load left pending would have the compiler wait for ALL vector loads, the
                    // prefetch of the next group included, wherever the fast path touches the same register)
                    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
                }
            }
            if (RARE && care) {
                const bool rare = (int32_t)wq < 0, is_not = rare && (wq & kDwNeg), is_inord = rare && !(wq & kDwNeg);
                if (is_not) acc = ~acc;
                // candidates: documents where the group's boolean value is true (rval, expression.go:137)
                const AT in = AT(inord_wave(S, is_inord, (wq & kDwFieldMask) >> kDwFieldShift, (uint64_t)(acc & valid), d0));
                if (is_inord) acc = in;
            }
        };
        if (!careful) {
#pragma unroll
            for (int q = 0; q < 4; q++) step(w[q], pv[q], false);
        } else {
            // one copy of the slow code: the chunk's words one after the other in a rolled loop.  The words rotate through
            // fixed registers -- indexing w[] / pv[] with the loop counter would move both arrays to scratch memory, for
            // the fast path above as well
            uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            AT p0 = pv[0], p1 = pv[1], p2 = pv[2], p3 = pv[3];
#pragma nounroll
            for (int q = 0; q < 4; q++) {
                step(w0, p0, true);
                w0 = w1; w1 = w2; w2 = w3;
                p0 = p1; p1 = p2; p2 = p3;
            }
        }
    };
    for (uint32_t c = 0; c < wchunks; c += 4) {
        trip(q0, c);
        if (c + 1 < wchunks) trip(q1, c + 1);
        if (c + 2 < wchunks) trip(q2, c + 2);
        if (c + 3 < wchunks) trip(q3, c + 3);
    }
    return acc;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a barrier plus a fence over ALL memory: the compiler
// puts s_waitcnt vmcnt(0) in front of it, i.e. every barrier would wait for the prefetch loads in flight and a group would
// pay their whole latency (measured: 4 400 of 24 000 cycles per group).  The loads' registers are tracked by the
// compiler as usual and waited for where they are read.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// atomic OR into the LDS word at byte offset `at` (the presence matrix starts at LDS address 0: checked in the kernel)
using lds_word = __attribute__((address_space(3))) uint32_t;
__device__ __forceinline__ void lds_or(uint32_t at, uint32_t bits) {
    __hip_atomic_fetch_or(reinterpret_cast<lds_word*>((size_t)at), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// PROG_LDS: the fused programs (and their offsets) are staged in LDS once per workgroup, so the interpreter's
// dependent word-after-word fetches cost an LDS round trip instead of an L2 one
template <int G> struct PType { using type = uint64_t; };
template <> struct PType<32> { using type = uint32_t; };
template <> struct PType<16> { using type = uint16_t; };
template <> struct PType<8> { using type = uint8_t; };
template <int G> struct AType { using type = uint32_t; };
template <> struct AType<64> { using type = Mask64; };

// G = documents per group = bits of a presence-matrix element: 64 when 8 bytes per slot fit LDS, else 32 / 16 / 8 so that
// large dictionaries still keep P in LDS (the evaluation then covers fewer documents per operation, but P stops being an
// L2 ping-pong of atomics and random reads)
template <bool P_LDS, bool PROG_LDS, int G, int RARE, bool DBG = false>   // RARE: 0 no NOT / INORD word, 1 some, 2 some and a wide INORD group
__global__ void __launch_bounds__(kSolveBlockThreads) k_solve_groups(const SolveParams S) {
    const uint32_t dbg = DBG ? S.dbg : 0u;      // timing-study knock-outs (GFT_SOLVE_DEBUG): compiled out of production launches
    using PT = typename PType<G>::type;
    using AT = typename AType<G>::type;                         // document masks of the evaluation
    constexpr uint32_t kTeam = 16, kTeams = kSolveBlockThreads / kTeam;   // lanes that share one unit while P is built
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the same in all lanes
    constexpr uint32_t kWaves = kSolveBlockThreads / 64;
    const uint32_t tile_words = S.tile_words;                   // bitmap words covered by one pass (<= kSolveTileWords)
    const uint32_t bm_words = (S.n_exprs + 31) / 32;
    PT* P = P_LDS ? reinterpret_cast<PT*>(smem) : reinterpret_cast<PT*>(S.p_scratch + (size_t)blockIdx.x * S.n_slots);
    uint32_t* O = reinterpret_cast<uint32_t*>(smem + (P_LDS ? (((size_t)S.n_slots * sizeof(PT) + 15) & ~(size_t)15) : 0));   // [64][tile_words | 1]
    uint32_t* Pw = reinterpret_cast<uint32_t*>(P);
    if (P_LDS && (uint32_t)(uintptr_t)(lds_word*)smem != 0) __builtin_trap();   // lds_or: P at LDS address 0 (no static LDS in this file)
    const uint32_t ostride = tile_words | 1u;                   // odd row stride: the transposed columns land in different banks
    uint64_t* R = reinterpret_cast<uint64_t*>(O + 64 * ostride);      // [tile_words * 32] results by expression
    uint32_t* lprog = reinterpret_cast<uint32_t*>(R + tile_words * 32);  // [fprog_words] when PROG_LDS
    uint32_t* loff = lprog + S.fprog_words;                     // [n_exprs + 1]
    uint32_t* lorder = loff + S.n_exprs + 1;                    // [n_exprs]
    if (PROG_LDS) {
        for (uint32_t i = threadIdx.x; i < S.fprog_words; i += kSolveBlockThreads) lprog[i] = S.fprog[i];
        for (uint32_t i = threadIdx.x; i <= S.n_exprs; i += kSolveBlockThreads) loff[i] = (uint32_t)S.fprog_off[i];
        for (uint32_t i = threadIdx.x; i < S.n_exprs; i += kSolveBlockThreads) lorder[i] = S.order[i];
    }

    for (uint32_t i = threadIdx.x; i < S.n_slots; i += kSolveBlockThreads) {
        if (P_LDS) P[i] = 0;
        else __hip_atomic_store(&P[i], (PT)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    const uint64_t n_groups = (S.n_docs + G - 1) / G;
    // The units of a group's documents are consecutive (doc_unit_base), so the presence matrix is built unit by unit:
    // a team of kTeam lanes per unit, kTeams units at once -- a document of many units (a 1 MB document is > 100) is
    // spread over the whole workgroup instead of queueing behind one team.  What the build needs is three dependent
    // reads away (range of units -> slab and count of this thread's unit -> its entries); each is issued one group
    // ahead of the next, behind the evaluation, so none of them ever waits for the one before it:
    //   group g + 3: its range of units                                  (`rng`)
    //   group g + 2: this thread's first unit: slab, count, document     (`nxt`)
    //   group g + 1: the first kPfTerms slab entries of that unit        (`pf_t`; a unit of up to 192 matches)
    // -- the build of the common case is LDS atomics on registers.  All of it stays in VECTOR registers and every load
    // is unconditional (indices clamped, validity checked where the value is used): a scalar value or a predicate derived
    // from a load would make the wave wait for it on the spot.
    // PIPE: the kernel variant with the NOT / INORD path has no registers to spare (the compiler would spill the ones the
    // prefetch is landing in, and wait for them on the spot): there only the range of units travels ahead, the unit and its
    // entries are read where they are needed, two round trips in front of every build (keeping the unit in flight as
    // well already costs more in spills than it hides: 0.81 against 0.66 ms with 50 % INORD expressions)
    constexpr bool PIPE = !RARE;
    constexpr int kPfTerms = 12;
    struct UnitPf { uint64_t u, U1, s; uint32_t n, doc; };      // u: this thread's first unit (valid if u < U1)
    struct Range { uint64_t U0, U1; };
    const uint32_t member = threadIdx.x % kTeam;
    auto launder = [](uint64_t v) { asm volatile("" : "+v"(v)); return v; };   // (keeps a uniform index out of the scalar unit)
    auto fetch_range = [&](uint64_t g, Range& r) {                // past the last group: an empty range
        const uint64_t da = g < n_groups ? g * G : S.n_docs, db = da + G < S.n_docs ? da + G : S.n_docs;
        r.U0 = S.doc_unit_base[launder(da)]; r.U1 = S.doc_unit_base[launder(db)];
    };
    auto fetch_unit = [&](const Range& r, UnitPf& m) {           // r: fetch_range, arrived
        m.u = r.U0 + threadIdx.x / kTeam; m.U1 = r.U1;
        const uint64_t uc = m.u < m.U1 ? m.u : 0;                 // (unit 0 exists: every document has at least one)
        m.s = S.unit_start[uc]; m.n = S.unit_count[uc]; m.doc = S.units[uc].doc;
    };
    uint32_t pf_t[kPfTerms];
    auto fetch_terms = [&](const UnitPf& m) {                    // m: fetch_unit, arrived
        const uint32_t n = m.u < m.U1 ? m.n : 0u;
        uint64_t at[kPfTerms];
#pragma unroll
        for (int q = 0; q < kPfTerms; q++) at[q] = member + q * kTeam < n ? m.s + member + q * kTeam : 0;    // (entry 0 of the pool exists)
        __builtin_amdgcn_sched_barrier(0);                       // addresses (and whatever they need reloaded) first, then the loads
#pragma unroll
        for (int q = 0; q < kPfTerms; q++) pf_t[q] = S.term[at[q]];
    };
    // phase clocks of the timing studies (GFT_SOLVE_DEBUG & 8)
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = DBG ? clock64() : 0;
    auto mark = [&](int ph) {
        if (DBG && (dbg & 8)) { const unsigned long long now = clock64(); tl[ph] += now - tprev; tprev = now; }
    };
    // rows of an output tile -> global bitmap: a wave stores whole rows -- two at a time while a row is at most 32
    // words -- so no index is divided
    auto store_rows = [&](uint64_t d0, uint32_t nd, uint32_t w0, uint32_t tw) {
        const uint32_t per = tw <= 32 ? 2u : 1u, c = per == 2 ? lane & 31u : lane;
        for (uint32_t j = wave * per + (per == 2 ? lane >> 5 : 0u); j < nd; j += kWaves * per)
            if (c < tw) S.bitmap[(d0 + j) * bm_words + w0 + c] = O[j * ostride + c];
    };
    uint64_t pend_d0 = 0;
    uint32_t pend_nd = 0, pend_w0 = 0, pend_tw = 0;             // the last tile of the group before (nd = 0: none)
    UnitPf cur, nxt;
    Range rng;
    fetch_range(blockIdx.x, rng);
    if (PIPE) {
        fetch_unit(rng, cur);
        fetch_range((uint64_t)blockIdx.x + gridDim.x, rng);
        fetch_unit(rng, nxt);
        fetch_range((uint64_t)blockIdx.x + 2 * (uint64_t)gridDim.x, rng);
        fetch_terms(cur);
    }
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t d0 = g * G;
        const uint32_t nd = (uint32_t)(S.n_docs - d0 < (uint64_t)G ? S.n_docs - d0 : (uint64_t)G);

        // ---- 1. presence matrix ------------------------------------------------------------------------------
        mark(7);
        if (!PIPE) { fetch_unit(rng, cur); fetch_terms(cur); }
        if (!(dbg & 1)) {
            bool pf = true;                                       // first (normally only) unit of this team: prefetched
            for (uint64_t u = cur.u; u < cur.U1; u += kTeams, pf = false) {
                const uint64_t s = pf ? cur.s : S.unit_start[u];
                const uint32_t n = pf ? cur.n : S.unit_count[u];
                const uint32_t j = (uint32_t)((pf ? cur.doc : S.units[u].doc) - d0);
                const uint32_t j_word = (j >> 5) * 4, j_bit = 1u << (j & 31);
                // kPfTerms slab entries per lane are in flight at a time.  The LDS unit retires about two lane-atomics per
                // cycle whatever the addresses are, so only the lanes with an entry issue one
                for (uint32_t i = member; i < n; i += kPfTerms * kTeam) {
                    uint32_t t[kPfTerms];
                    if (pf && i == member) {
#pragma unroll
                        for (int q = 0; q < kPfTerms; q++) t[q] = pf_t[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < kPfTerms; q++) t[q] = i + q * kTeam < n ? S.term[s + i + q * kTeam] : 0u;
                    }
#pragma unroll
                    for (int q = 0; q < kPfTerms; q++)
                        if (i + q * kTeam < n) {
                            // (timing study 16: the same number of atomics, scattered -- same-word collisions cost nothing)
                            const uint32_t tq = DBG && (dbg & 16) ? (t[q] + threadIdx.x * 37u) % (S.n_slots - 1) : t[q];
                            if (G == 64 && P_LDS) {                                 // bit j of element t: word t * 2 + j / 32
                                // (timing study 32: a plain store in place of the atomic -- what does the read-modify-write cost?)
                                if (DBG && (dbg & 32)) *reinterpret_cast<lds_word*>((size_t)(tq * 8 + j_word)) = j_bit;
                                else lds_or(tq * 8 + j_word, j_bit);
                            } else {
                                const uint32_t bp = tq * (uint32_t)G + j;          // (n_slots * G < 2^32)
                                if (P_LDS) lds_or((bp >> 5) * 4, 1u << (bp & 31));
                                else atomicOr(&Pw[bp >> 5], 1u << (bp & 31));
                            }
                        }
                }
            }
            if (S.x_off) {
                for (uint32_t j = threadIdx.x / kTeam; j < nd; j += kTeams) {
                    const uint64_t x0 = S.x_off[d0 + j], x1 = S.x_off[d0 + j + 1];
                    for (uint64_t i = x0 + member; i < x1; i += kTeam) {
                        const size_t bp = (size_t)S.x_slot[i] * G + j;
                        if (P_LDS) __hip_atomic_fetch_or(&Pw[bp >> 5], 1u << (bp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else atomicOr(&Pw[bp >> 5], 1u << (bp & 31));
                    }
                }
            }
        }
        mark(0);
        if (P_LDS) lds_barrier();
        else { __threadfence_block(); __syncthreads(); }       // (P in HBM: the other waves' atomics must be visible)
        mark(1);
        store_rows(pend_d0, pend_nd, pend_w0, pend_tw);          // (the group before; see `pend` below)
        pend_nd = 0;
        mark(6);

        // the next group's slab entries, the unit of the group after it, the range of the one after that: in flight
        // during the evaluation (no __syncthreads() from here to the next build: lds_barrier does not wait for them).
        // Straight-line and unconditional, so that the loaded registers ARE the loop-carried ones: a copy of a register
        // that a load is still filling makes the wave wait for the load on the spot.
        if (PIPE) {
            cur = nxt;
            fetch_terms(cur);
            fetch_unit(rng, nxt);
            fetch_range(g + 3 * (uint64_t)gridDim.x, rng);
        } else fetch_range(g + gridDim.x, rng);

        // ---- 2. expressions, tile by tile over the bitmap words ------------------------------------------------------
        const uint64_t valid = nd == 64 ? ~0ull : ((1ull << nd) - 1);
        for (uint32_t w0 = 0; w0 < bm_words; w0 += tile_words) {
            const uint32_t tw = bm_words - w0 < tile_words ? bm_words - w0 : tile_words;
            const uint32_t e0 = w0 << 5;                                          // first expression of the tile
            const uint32_t ne = S.n_exprs - e0 < (tw << 5) ? S.n_exprs - e0 : (tw << 5);
            // 2a. evaluation in sorted order: 64 programs of similar length and one interpreter class per wave, sixteen
            // blocks at a time, dealt to the waves by gft_set_programs so that the four SIMDs get similar sums of work
            const uint32_t nblk = (ne + 63) / 64;
            for (uint32_t b16 = 0; b16 < nblk && !(dbg & 2); b16 += kWaves) {
                const uint32_t b = uniform_word(S.wave_blk, (e0 >> 6) + b16 + wave);
                const uint32_t i = b * 64 + lane;
                if (b < nblk) {
                    const bool has = i < ne;                                      // lanes past the tile: no program
                    const uint32_t e = PROG_LDS ? lorder[e0 + (has ? i : 0)] : S.order[e0 + (has ? i : 0)];
                    const uint64_t po = PROG_LDS ? loff[e] : S.fprog_off[e];
                    const uint32_t len = (uint32_t)((PROG_LDS ? loff[e + 1] : S.fprog_off[e + 1]) - po);
                    // programs in LDS: linear; in global memory: the block's 4-word chunks transposed ([chunk][lane]) so
                    // that the lanes of a wave read consecutive 16-byte pieces
                    const uint4* prog = reinterpret_cast<const uint4*>(PROG_LDS ? lprog + po : S.fprog_t + uniform_word(S.fblk_off, (e0 >> 6) + b)) +
                                        (PROG_LDS ? 0 : lane);
                    const uint32_t stride = PROG_LDS ? 1u : 64u;
                    const uint32_t cls = uniform_word(S.blk_class, (e0 >> 6) + b);
                    const uint32_t chunks = has ? len / 4 : 0;
                    AT r;
                    if constexpr (PROG_LDS)
                        r = cls == 0   ? run_program<P_LDS, 0, RARE, PT, AT>(S, P, prog, stride, chunks, AT(valid), d0)
                            : cls == 1 ? run_program<P_LDS, kSolveRegStack, RARE, PT, AT>(S, P, prog, stride, chunks, AT(valid), d0)
                                       : run_program<P_LDS, kSolveRegStackDeep, RARE, PT, AT>(S, P, prog, stride, chunks, AT(valid), d0);
                    else
                        r = cls == 0   ? run_program_far<P_LDS, 0, RARE, PT, AT>(S, P, prog, stride, chunks, AT(valid), d0)
                            : cls == 1 ? run_program_far<P_LDS, kSolveRegStack, RARE, PT, AT>(S, P, prog, stride, chunks, AT(valid), d0)
                                       : run_program_far<P_LDS, kSolveRegStackDeep, RARE, PT, AT>(S, P, prog, stride, chunks, AT(valid), d0);
                    if (has) R[e - e0] = (uint64_t)r;
                }
            }
            if constexpr (RARE == 2) {
                // 2a'. the expressions with a wide INORD group (stand-ins in the fused form: their R is 0 by now): a document per
                // wave, all sixteen waves on one expression -- the candidates of ONE lane's program would otherwise queue on one
                // wave while fifteen idle
                lds_barrier();
                const uint64_t wv = (uint64_t)blockIdx.x * kWaves + wave;
                for (uint32_t k = 0; k < S.n_wide; k++) {
                    const uint32_t e = uniform_word(S.wide_list, 3 * k);
                    if (e < e0 || e >= e0 + ne) continue;
                    const uint32_t goff = uniform_word(S.wide_list, 3 * k + 1), glen = uniform_word(S.wide_list, 3 * k + 2);
                    for (uint32_t j = wave; j < nd; j += kWaves) {
                        const uint64_t d = d0 + j;
                        DocHits M;
                        M.unit_start = S.unit_start; M.unit_count = S.unit_count;
                        M.term = S.term; M.pos = S.pos;
                        M.u0 = S.doc_unit_base[d]; M.u1 = S.doc_unit_base[d + 1];
                        M.nx = 0; M.xslot = nullptr; M.xpos = nullptr;
                        M.back = S.pos_back;
                        M.per = 1;
                        if (M.u1 - M.u0 >= kUnitsPerLaneMode) {
                            const Unit first = S.units[M.u0];
                            const uint32_t per32 = __builtin_amdgcn_readfirstlane(first.hi - first.lo);
                            M.per = per32 ? per32 : 1u;
                        }
                        if (S.x_off) {
                            const uint64_t x0 = S.x_off[d];
                            M.xslot = S.x_slot + x0; M.xpos = S.x_pos + x0; M.nx = (uint32_t)(S.x_off[d + 1] - x0);
                        }
                        const bool r = wide_expr_doc(S.gprog + goff, glen, M, reinterpret_cast<const uint32_t*>(P), (uint32_t)G, j,
                                                     S.wide_slot ? S.wide_slot + wv * S.wide_cap : nullptr, S.wide_theta ? S.wide_theta + wv * S.wide_cap : nullptr);
                        if (r && lane == 0)
                            __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(&R[e - e0]), 1ull << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            mark(2);
            lds_barrier();                                       // R is complete (LDS); P was read by atomic loads if in HBM
            mark(3);
            // 2b. transpose in natural order: lane j ends up with the two bitmap words of document j for 64 expressions
            const uint32_t rounds = (tw + 1) / 2;
            for (uint32_t r = wave; r < rounds; r += kWaves) {
                const uint32_t el = r * 64 + lane;
                const uint64_t acc = (el < ne && !(dbg & 2)) ? R[el] : 0;
                const uint64_t mine = (dbg & 4) ? 0 : wave_transpose64(acc);
                O[lane * ostride + r * 2] = (uint32_t)mine;
                if (r * 2 + 1 < tw) O[lane * ostride + r * 2 + 1] = (uint32_t)(mine >> 32);
            }
            // ---- 3. (LDS) the last tile's evaluation was the last reader of P: wipe it for the next group in the same
            // phase -- a handful of wide stores per lane, no re-read of the matches
            if (P_LDS && w0 + tile_words >= bm_words) {
                uint4* P4 = reinterpret_cast<uint4*>(P);
                for (uint32_t i = threadIdx.x; i < (uint32_t)(((size_t)S.n_slots * sizeof(PT) + 15) / 16); i += kSolveBlockThreads) P4[i] = make_uint4(0, 0, 0, 0);
            }
            mark(4);
            lds_barrier();
            mark(5);
            // rows of the tile -> global bitmap.  Those of a group's last (normally only) tile wait until the next group's
            // presence matrix is built (`pend`): stores issued HERE would be younger than the prefetch loads in flight, and
            // the s_waitcnt vmcnt(0) in front of the next build would wait for them to reach L2 (1 500 cycles per group).
            // No barrier behind either: O is written again only after the barrier that follows the next evaluation.
            if (w0 + tile_words >= bm_words) { pend_d0 = d0; pend_nd = nd; pend_w0 = w0; pend_tw = tw; }
            else store_rows(d0, nd, w0, tw);
            mark(6);
        }

        // ---- 3. (HBM) clear the touched entries of P for the next group ----------------------------------------------
        if (!P_LDS) {
            for (uint32_t j = threadIdx.x / kTeam; j < nd; j += kSolveBlockThreads / kTeam) {
                const uint32_t member = threadIdx.x % kTeam;
                const uint64_t d = d0 + j;
                const uint64_t u0 = S.doc_unit_base[d], u1 = S.doc_unit_base[d + 1];
                for (uint64_t u = u0; u < u1; u++) {
                    const uint64_t s = S.unit_start[u];
                    const uint32_t n = S.unit_count[u];
                    for (uint32_t i = member; i < n; i += kTeam)
                        __hip_atomic_store(&P[S.term[s + i]], (PT)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (S.x_off) {
                    const uint64_t x0 = S.x_off[d], x1 = S.x_off[d + 1];
                    for (uint64_t i = x0 + member; i < x1; i += kTeam)
                        __hip_atomic_store(&P[S.x_slot[i]], (PT)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    store_rows(pend_d0, pend_nd, pend_w0, pend_tw);
    if (DBG && (dbg & 8) && lane == 0 && S.dbg_out)
        for (int ph = 0; ph < 8; ph++) atomicAdd(&S.dbg_out[wave * 8 + ph], tl[ph]);
}

}  // namespace

size_t solve_lds_bytes(uint32_t n_slots, uint32_t tile_words, uint32_t group_docs, bool p_in_lds, uint32_t prog_words,
                       uint32_t n_exprs, bool prog_in_lds) {
    return (p_in_lds ? (((size_t)n_slots * (group_docs / 8) + 15) & ~(size_t)15) : 0) + (size_t)64 * (tile_words | 1u) * 4 +
           (size_t)tile_words * 32 * 8 + (prog_in_lds ? ((size_t)prog_words + 2 * (size_t)n_exprs + 1) * 4 : 0);
}

namespace {
template <int G>
hipError_t launch_g(const SolveParams& S, bool p_in_lds, bool prog_in_lds, unsigned grid, size_t lds, hipStream_t st) {
    using Kern = void (*)(const SolveParams);
    const bool io = S.has_rare != 0;
    const Kern fn = p_in_lds ? (prog_in_lds ? (io ? k_solve_groups<true, true, G, 1> : k_solve_groups<true, true, G, 0>)
                                            : (io ? k_solve_groups<true, false, G, 1> : k_solve_groups<true, false, G, 0>))
                             : (prog_in_lds ? (io ? k_solve_groups<false, true, 64, 1> : k_solve_groups<false, true, 64, 0>)
                                            : (io ? k_solve_groups<false, false, 64, 1> : k_solve_groups<false, false, 64, 0>));
    Kern run = fn;
    // a program set with a wide INORD group (SolveParams::wide_cap): the variant that can call the wide paths; its programs
    // are read from L2 (the caller plans LDS accordingly)
    if (S.wide_cap) run = p_in_lds ? k_solve_groups<true, false, G, 2> : k_solve_groups<false, false, 64, 2>;
    // timing studies: the benchmark's shape only (presence matrix and programs in LDS, 64 documents per group)
    if (S.dbg && !S.wide_cap && G == 64 && p_in_lds && prog_in_lds) run = io ? k_solve_groups<true, true, 64, 1, true> : k_solve_groups<true, true, 64, 0, true>;
    // ... and the shape of a 100 000-term dictionary (8 documents per group, programs in L2)
    if (S.dbg && !S.wide_cap && G == 8 && p_in_lds && !prog_in_lds) run = io ? k_solve_groups<true, false, 8, 1, true> : k_solve_groups<true, false, 8, 0, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(run), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    run<<<dim3(grid), dim3(kSolveBlockThreads), lds, st>>>(S);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_solve(const SolveParams& S, uint32_t group_docs, bool p_in_lds, bool prog_in_lds, unsigned grid, hipStream_t st) {
    if (!S.n_docs || !S.n_exprs) return hipSuccess;
    const size_t lds = solve_lds_bytes(S.n_slots, S.tile_words, group_docs, p_in_lds, S.fprog_words, S.n_exprs, prog_in_lds);
    switch (p_in_lds ? group_docs : 64) {
    case 64: return launch_g<64>(S, p_in_lds, prog_in_lds, grid, lds, st);
    case 32: return launch_g<32>(S, p_in_lds, prog_in_lds, grid, lds, st);
    case 16: return launch_g<16>(S, p_in_lds, prog_in_lds, grid, lds, st);
    default: return launch_g<8>(S, p_in_lds, prog_in_lds, grid, lds, st);
    }
}

}  // namespace gft
