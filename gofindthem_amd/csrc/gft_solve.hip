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

#include "gft_kernels.hpp"

namespace gft {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

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
};

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;      // pair that never matches: the empty position list
constexpr uint32_t kUnitsPerLaneMode = 8;      // documents with at least this many units: one unit per lane

__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
#pragma unroll
    for (int s = 32; s; s >>= 1) {
        const int64_t o = __shfl_xor(v, s, 64);
        v = o < v ? o : v;
    }
    // the same value in every lane: tell the compiler, so that what is derived from it stays in scalar registers
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
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
    auto test = [&](uint32_t t, uint32_t p) {
        if (t == sl0 && (int64_t)p > th0 && (int64_t)p < best) best = p;
        for (uint32_t k = 1; k < pc; k++) {
            const uint32_t sl = __builtin_amdgcn_readlane(my_slot, pb + k);
            const int64_t th = readlane_i64(my_theta, pb + k);
            if (t == sl && (int64_t)p > th && (int64_t)p < best) best = p;
        }
    };
    if (M.u1 - M.u0 >= kUnitsPerLaneMode) {
        for (uint64_t u = M.u0 + lane; u < M.u1; u += 64) {
            const uint64_t s = M.unit_start[u];
            const uint32_t n = M.unit_count[u];
            uint32_t i = 0;
            for (; i + 4 <= n; i += 4) {                         // four matches in flight per lane
                uint32_t t[4], p[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { t[q] = M.term[s + i + q]; p[q] = M.pos[s + i + q]; }
#pragma unroll
                for (int q = 0; q < 4; q++) test(t[q], p[q]);
            }
            for (; i < n; i++) test(M.term[s + i], M.pos[s + i]);
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

// 64 x 64 bit-matrix transpose across a wave: lane i holds row i; afterwards lane j holds column j (bit i = old row
// i's bit j).  Six exchange steps with the partner lane i ^ s, swapping the off-diagonal s x s blocks.
template <int SH>
__device__ __forceinline__ uint64_t transpose_step(uint64_t x, uint32_t lane, uint64_t m) {
    const uint64_t t = __shfl_xor(x, SH, 64);
    return (lane & SH) ? ((t >> SH) & m) | (x & ~m) : (x & m) | ((t & m) << SH);
}
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x) {
    const uint32_t lane = lane_id();
    x = transpose_step<32>(x, lane, 0x00000000FFFFFFFFull);
    x = transpose_step<16>(x, lane, 0x0000FFFF0000FFFFull);
    x = transpose_step<8>(x, lane, 0x00FF00FF00FF00FFull);
    x = transpose_step<4>(x, lane, 0x0F0F0F0F0F0F0F0Full);
    x = transpose_step<2>(x, lane, 0x3333333333333333ull);
    x = transpose_step<1>(x, lane, 0x5555555555555555ull);
    return x;
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

// One fused program over 64 documents (bit j of every mask = document d0 + j).  Lanes of a wave run different
// programs, so the interpreter is predicated rather than branched: every word costs one presence read and a handful of
// selects.  DEEP = false keeps the accumulator stack in two registers (programs that nest deeper are sorted into
// blocks of their own and take DEEP = true: four registers backed by scratch).
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

// ALL 64 lanes of a wave call this together (lanes without a program pass chunks = 0): the trip count is the wave's
// maximum, finished lanes run kFopNop words, so the INORD steps can use the whole wave.
template <bool P_LDS, bool DEEP, bool INORD, class PT, class AT>
__device__ __forceinline__ AT run_program(const SolveParams& S, const PT* P, const uint4* prog, uint32_t stride, uint32_t chunks,
                                          AT valid, uint64_t d0) {
    // HBM-resident P was written with L2 atomics by other waves: read it past this CU's L1
    auto ld = [&](uint32_t slot) -> AT {
        return P_LDS ? (AT)P[slot] : (AT)__hip_atomic_load(&P[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    AT acc = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, deep[DEEP ? kMaxBoolDepth : 1];
    uint32_t sp = 0;
    constexpr uint32_t kNop = (uint32_t)kFopNop << 28;
    uint4 nx = chunks ? prog[0] : make_uint4(kNop, kNop, kNop, kNop);
    uint4 nx2 = chunks > 1 ? prog[stride] : make_uint4(kNop, kNop, kNop, kNop);     // two chunks ahead (programs in L2)
    const uint32_t wchunks = wave_max_u32(chunks);
    // four words per trip: the next two chunks and this chunk's four presence reads are in flight together, so a trip
    // exposes one memory round trip instead of four (programs are padded to whole chunks with kFopNop)
    for (uint32_t c = 0; c < wchunks; c++) {
        const uint32_t w[4] = {nx.x, nx.y, nx.z, nx.w};
        nx = nx2;
        nx2 = make_uint4(kNop, kNop, kNop, kNop);
        if (c + 2 < chunks) nx2 = prog[(size_t)(c + 2) * stride];
        AT pv[4];
        bool rare = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t op = w[q] >> 28;
            pv[q] = ld(op < kFopAndPop ? (w[q] & 0x0FFFFFFFu) : 0);
            rare |= op == kFopNot || op == kFopInord;
        }
        const bool any_rare = INORD && __any(rare);             // wave-uniform: only around INORD groups (INORD = false:
                                                                // the program set has none, the path is compiled out)
        auto step = [&](const uint32_t wq, const AT pvq, const bool with_rare) __attribute__((always_inline)) {
            const uint32_t op = wq >> 28, a = wq & 0x0FFFFFFFu;
            const bool is_s = op < kFopAndPop;
            const AT v = pvq ^ (AT)(0 - (AT)((op >> 2) & 1));   // SetN / AndNS / OrNS (bits past the group are never stored)
            const uint32_t k = op & 3;
            const AT t = k == 3 ? (acc | v) : (acc & v);
            const AT sacc = k == 1 ? v : t;
            const bool is_pop = (op & 14) == kFopAndPop, is_push = op == kFopPush;
            const AT pacc = (op & 1) ? (acc | s0) : (acc & s0);
            const AT before = acc;
            acc = is_s ? sacc : is_pop ? pacc : acc;
            if (!DEEP) {
                const AT n0 = is_push ? before : is_pop ? s1 : s0;
                s1 = is_push ? s0 : s1;
                s0 = n0;
            } else {
                if (is_push) {
                    if (sp >= 4) deep[sp - 4] = s3;
                    s3 = s2; s2 = s1; s1 = s0; s0 = before;
                    sp++;
                }
                if (is_pop) {
                    s0 = s1; s1 = s2; s2 = s3;
                    sp--;
                    if (sp >= 4) s3 = deep[sp - 4];
                }
            }
            if (with_rare) {
                if (op == kFopNot) acc = ~acc;
                // candidates: documents where the group's boolean value is true (rval, expression.go:137)
                const AT in = (AT)inord_wave(S, op == kFopInord, a, (uint64_t)(acc & valid), d0);
                if (op == kFopInord) acc = in;
            }
        };
        if (!any_rare) {
#pragma unroll
            for (int q = 0; q < 4; q++) step(w[q], pv[q], false);
        } else {
            // one copy of the INORD code: the chunk's words one after the other in a rolled loop
#pragma nounroll
            for (int q = 0; q < 4; q++) {
                const uint32_t wq = q == 0 ? w[0] : q == 1 ? w[1] : q == 2 ? w[2] : w[3];
                const AT pvq = q == 0 ? pv[0] : q == 1 ? pv[1] : q == 2 ? pv[2] : pv[3];
                step(wq, pvq, true);
            }
        }
    }
    return acc;
}

// PROG_LDS: the fused programs (and their offsets) are staged in LDS once per workgroup, so the interpreter's
// dependent word-after-word fetches cost an LDS round trip instead of an L2 one
template <int G> struct PType { using type = uint64_t; };
template <> struct PType<32> { using type = uint32_t; };
template <> struct PType<16> { using type = uint16_t; };
template <> struct PType<8> { using type = uint8_t; };
template <int G> struct AType { using type = uint32_t; };
template <> struct AType<64> { using type = uint64_t; };

// G = documents per group = bits of a presence-matrix element: 64 when 8 bytes per slot fit LDS, else 32 / 16 / 8 so that
// large dictionaries still keep P in LDS (the evaluation then covers fewer documents per operation, but P stops being an
// L2 ping-pong of atomics and random reads)
template <bool P_LDS, bool PROG_LDS, int G, bool INORD, bool DBG = false>
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
    uint32_t* O = reinterpret_cast<uint32_t*>(smem + (P_LDS ? (((size_t)S.n_slots * sizeof(PT) + 15) & ~(size_t)15) : 0));   // [64][tile_words]
    uint32_t* Pw = reinterpret_cast<uint32_t*>(P);
    uint64_t* R = reinterpret_cast<uint64_t*>(O + 64 * tile_words);   // [tile_words * 32] results by expression
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
    // spread over the whole workgroup instead of queueing behind one team.  This thread's first unit of the NEXT group
    // (range of units, slab, count, document) is fetched while the current group is evaluated (pf_*).
    uint64_t pf_U0 = 0, pf_U1 = 0, pf_s = 0;
    uint32_t pf_n = 0, pf_doc = 0;
    auto prefetch = [&](uint64_t g) {
        pf_U0 = pf_U1 = 0; pf_n = 0;
        if (g < n_groups) {
            const uint64_t da = g * G, db = da + G < S.n_docs ? da + G : S.n_docs;
            pf_U0 = S.doc_unit_base[da]; pf_U1 = S.doc_unit_base[db];
            const uint64_t u = pf_U0 + threadIdx.x / kTeam;
            if (u < pf_U1) { pf_s = S.unit_start[u]; pf_n = S.unit_count[u]; pf_doc = S.units[u].doc; }
        }
    };
    // ... and, once those have arrived (after the evaluation), the first four slab entries of that unit
    uint32_t pf_t[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto prefetch_terms = [&]() {
        const uint32_t member = threadIdx.x % kTeam;
#pragma unroll
        for (int q = 0; q < 4; q++) pf_t[q] = member + q * kTeam < pf_n ? S.term[pf_s + member + q * kTeam] : 0xFFFFFFFFu;
    };
    prefetch(blockIdx.x);
    prefetch_terms();
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t d0 = g * G;
        const uint32_t nd = (uint32_t)(S.n_docs - d0 < (uint64_t)G ? S.n_docs - d0 : (uint64_t)G);

        // ---- 1. presence matrix ------------------------------------------------------------------------------
        if (!(dbg & 1)) {
            const uint32_t member = threadIdx.x % kTeam;
            const uint64_t U0 = pf_U0, U1 = pf_U1;
            for (uint64_t u = U0 + threadIdx.x / kTeam; u < U1; u += kTeams) {
                const bool pf = u < U0 + kTeams;                   // first (normally only) unit of this team
                const uint64_t s = pf ? pf_s : S.unit_start[u];
                const uint32_t n = pf ? pf_n : S.unit_count[u];
                const uint32_t j = (uint32_t)((pf ? pf_doc : S.units[u].doc) - d0);
                // twelve slab entries per lane are in flight at a time (the first four came with the prefetch): a unit of
                // up to 192 matches is one round trip
                for (uint32_t i = member; i < n; i += 12 * kTeam) {
                    uint32_t t[12];
#pragma unroll
                    for (int q = 0; q < 12; q++)
                        t[q] = (q < 4 && pf && i == member) ? pf_t[q] : i + q * kTeam < n ? S.term[s + i + q * kTeam] : 0xFFFFFFFFu;
#pragma unroll
                    for (int q = 0; q < 12; q++)
                        if (t[q] != 0xFFFFFFFFu) {
                            const size_t bp = (size_t)t[q] * G + j;              // bit j of element t
                            if (P_LDS) __hip_atomic_fetch_or(&Pw[bp >> 5], 1u << (bp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            else atomicOr(&Pw[bp >> 5], 1u << (bp & 31));
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
        if (!P_LDS) __threadfence_block();
        __syncthreads();

        prefetch(g + gridDim.x);

        // ---- 2. expressions, tile by tile over the bitmap words ------------------------------------------------------
        const uint64_t valid = nd == 64 ? ~0ull : ((1ull << nd) - 1);
        for (uint32_t w0 = 0; w0 < bm_words; w0 += tile_words) {
            const uint32_t tw = bm_words - w0 < tile_words ? bm_words - w0 : tile_words;
            const uint32_t e0 = w0 << 5;                                          // first expression of the tile
            const uint32_t ne = S.n_exprs - e0 < (tw << 5) ? S.n_exprs - e0 : (tw << 5);
            // 2a. evaluation in sorted order (gft_set_programs): 64 programs of similar length per wave.  Sixteen blocks
            // at a time, longest first, dealt to the waves in a snake over the four SIMDs so that the SIMDs get similar sums
            const uint32_t nblk = (ne + 63) / 64;
            for (uint32_t b16 = 0; b16 < nblk && !(dbg & 2); b16 += kWaves) {
                const uint32_t row = wave >> 2, c4 = wave & 3;
                const uint32_t b = b16 + row * 4 + ((row & 1) ? 3 - c4 : c4);
                const uint32_t i = b * 64 + lane;
                if (b < nblk) {                                                   // wave-uniform
                    const bool has = i < ne;                                      // lanes past the tile: no program
                    const uint32_t e = PROG_LDS ? lorder[e0 + (has ? i : 0)] : S.order[e0 + (has ? i : 0)];
                    const uint64_t po = PROG_LDS ? loff[e] : S.fprog_off[e];
                    const uint32_t len = (uint32_t)((PROG_LDS ? loff[e + 1] : S.fprog_off[e + 1]) - po);
                    // programs in LDS: linear; in global memory: the block's 4-word chunks transposed ([chunk][lane]) so
                    // that the lanes of a wave read consecutive 16-byte pieces
                    const uint4* prog = reinterpret_cast<const uint4*>(PROG_LDS ? lprog + po : S.fprog_t + S.fblk_off[(e0 >> 6) + b]) +
                                        (PROG_LDS ? 0 : lane);
                    const uint32_t stride = PROG_LDS ? 1u : 64u;
                    const bool deep = S.blk_deep[(e0 >> 6) + b] != 0;              // wave-uniform
                    const uint32_t chunks = has ? len / 4 : 0;
                    const AT r = deep ? run_program<P_LDS, true, INORD, PT, AT>(S, P, prog, stride, chunks, (AT)valid, d0)
                                      : run_program<P_LDS, false, INORD, PT, AT>(S, P, prog, stride, chunks, (AT)valid, d0);
                    if (has) R[e - e0] = r;
                }
            }
            __syncthreads();
            // 2b. transpose in natural order: lane j ends up with the two bitmap words of document j for 64 expressions
            const uint32_t rounds = (tw + 1) / 2;
            for (uint32_t r = wave; r < rounds; r += kWaves) {
                const uint32_t el = r * 64 + lane;
                const uint64_t acc = (el < ne && !(dbg & 2)) ? R[el] : 0;
                const uint64_t mine = (dbg & 4) ? 0 : wave_transpose64(acc);
                O[lane * tile_words + r * 2] = (uint32_t)mine;
                if (r * 2 + 1 < tw) O[lane * tile_words + r * 2 + 1] = (uint32_t)(mine >> 32);
            }
            // ---- 3. (LDS) the last tile's evaluation was the last reader of P: wipe it for the next group in the same
            // phase -- a handful of wide stores per lane, no re-read of the matches
            if (P_LDS && w0 + tile_words >= bm_words) {
                uint4* P4 = reinterpret_cast<uint4*>(P);
                for (uint32_t i = threadIdx.x; i < (uint32_t)(((size_t)S.n_slots * sizeof(PT) + 15) / 16); i += kSolveBlockThreads) P4[i] = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
            // rows of the tile -> global bitmap.  No barrier behind it: O is written again only after the barrier that
            // follows the next evaluation, and the next group's presence build touches P alone
            for (uint32_t i = threadIdx.x; i < nd * tw; i += kSolveBlockThreads) {
                const uint32_t j = i / tw, c = i - j * tw;
                S.bitmap[(d0 + j) * bm_words + w0 + c] = O[j * tile_words + c];
            }
        }

        prefetch_terms();

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
}

}  // namespace

size_t solve_lds_bytes(uint32_t n_slots, uint32_t tile_words, uint32_t group_docs, bool p_in_lds, uint32_t prog_words,
                       uint32_t n_exprs, bool prog_in_lds) {
    return (p_in_lds ? (((size_t)n_slots * (group_docs / 8) + 15) & ~(size_t)15) : 0) + (size_t)64 * tile_words * 4 +
           (size_t)tile_words * 32 * 8 + (prog_in_lds ? ((size_t)prog_words + 2 * (size_t)n_exprs + 1) * 4 : 0);
}

namespace {
template <int G>
hipError_t launch_g(const SolveParams& S, bool p_in_lds, bool prog_in_lds, unsigned grid, size_t lds, hipStream_t st) {
    using Kern = void (*)(const SolveParams);
    const bool io = S.has_inord != 0;
    const Kern fn = p_in_lds ? (prog_in_lds ? (io ? k_solve_groups<true, true, G, true> : k_solve_groups<true, true, G, false>)
                                            : (io ? k_solve_groups<true, false, G, true> : k_solve_groups<true, false, G, false>))
                             : (prog_in_lds ? (io ? k_solve_groups<false, true, 64, true> : k_solve_groups<false, true, 64, false>)
                                            : (io ? k_solve_groups<false, false, 64, true> : k_solve_groups<false, false, 64, false>));
    Kern run = fn;
    // timing studies: the benchmark's shape only (presence matrix and programs in LDS, 64 documents per group)
    if (S.dbg && G == 64 && p_in_lds && prog_in_lds) run = io ? k_solve_groups<true, true, 64, true, true> : k_solve_groups<true, true, 64, false, true>;
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
