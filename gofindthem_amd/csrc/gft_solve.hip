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

// first position of `slot` that is > theta (theta == -1: any), or INT64_MAX.  A keyword and a regex with the same
// literal share one map key (finder/finder.go:181-196): the caller maps both onto one slot, both lists are read.
__device__ int64_t succ_query(const DocHits& M, uint32_t slot, int64_t theta) {
    int64_t best = INT64_MAX;
    for (uint64_t u = M.u0; u < M.u1; u++) {
        const uint64_t s = M.unit_start[u];
        const uint32_t n = M.unit_count[u];
        for (uint32_t i = 0; i < n; i++)
            if (M.term[s + i] == slot) {
                const int64_t p = M.pos[s + i];
                if (p > theta && p < best) best = p;
            }
    }
    for (uint32_t i = 0; i < M.nx; i++)
        if (M.xslot[i] == slot) {
            const int64_t p = M.xpos[i];
            if (p > theta && p < best) best = p;
        }
    return best;
}

struct Pair { uint32_t slot; int32_t theta; };

// Position algebra of one INORD group for one document (dsl/expression.go:87-95,111-116,129-137).  `prog` is the
// group's subtree in the public postfix form (UNIT/AND/OR words carrying GFT_K_INORD_FLAG).  Returns len(rpos) > 0.
//   UNIT t    -> {(t, -1)}
//   OR        -> union of the pair sets (only minimum and emptiness are ever observed, duplicates are harmless)
//   AND(L, R) -> m = min over L of succ(t, theta); {} if m = +inf, else {(t, max(theta, m)) : (t, theta) in R}
//                == rpos[getLowestIdxGTVal(rpos, lpos[0]):]   (expression.go:87-93,175-189)
__device__ bool inord_group_nonempty(const uint32_t* __restrict__ prog, uint32_t len, const DocHits& M) {
    Pair pairs[kMaxPairs];
    uint16_t rbeg[kMaxPairDepth], rcnt[kMaxPairDepth];
    uint32_t psp = 0;
    for (uint32_t pc = 0; pc < len; pc++) {
        const uint32_t w = prog[pc];
        switch (w >> 28) {
        case 1: {  // UNIT
            const uint32_t b = psp ? rbeg[psp - 1] + rcnt[psp - 1] : 0;
            pairs[b] = Pair{w & GFT_K_SLOT_MASK, -1};
            rbeg[psp] = (uint16_t)b; rcnt[psp] = 1; psp++;
            break;
        }
        case 2: {  // AND
            const uint32_t lb = rbeg[psp - 2], lc = rcnt[psp - 2], rb = rbeg[psp - 1], rc = rcnt[psp - 1];
            int64_t m = INT64_MAX;
            for (uint32_t i = 0; i < lc; i++) {
                const int64_t s = succ_query(M, pairs[lb + i].slot, pairs[lb + i].theta);
                if (s < m) m = s;
            }
            uint32_t nc = 0;
            if (m != INT64_MAX)
                for (uint32_t i = 0; i < rc; i++) {
                    Pair q = pairs[rb + i];
                    if ((int64_t)q.theta < m) q.theta = (int32_t)m;
                    pairs[lb + nc++] = q;
                }
            psp--;
            rcnt[psp - 1] = (uint16_t)nc;
            break;
        }
        case 3:  // OR: the two ranges are adjacent, union == concatenation
            psp--;
            rcnt[psp - 1] = (uint16_t)(rcnt[psp - 1] + rcnt[psp]);
            break;
        default:
            break;
        }
    }
    if (!psp) return false;
    const uint32_t b = rbeg[psp - 1], c = rcnt[psp - 1];
    for (uint32_t i = 0; i < c; i++)
        if (succ_query(M, pairs[b + i].slot, pairs[b + i].theta) != INT64_MAX) return true;
    return false;
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

// documents of `cand` (bit j = document d0 + j) whose INORD group `grp` has a non-empty position list
__device__ uint64_t inord_docs(const SolveParams& S, uint32_t grp, uint64_t cand, uint64_t d0) {
    uint64_t res = 0;
    const uint32_t goff = S.groups[grp * 2], glen = S.groups[grp * 2 + 1];
    while (cand) {
        const uint32_t j = (uint32_t)__builtin_ctzll(cand);
        cand &= cand - 1;
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
        if (inord_group_nonempty(S.gprog + goff, glen, M)) res |= 1ull << j;
    }
    return res;
}

// One fused program over 64 documents (bit j of every mask = document d0 + j).  Lanes of a wave run different
// programs, so the interpreter is predicated rather than branched: every word costs one presence read and a handful of
// selects.  DEEP = false keeps the accumulator stack in two registers (programs that nest deeper are sorted into
// blocks of their own and take DEEP = true: four registers backed by scratch).
template <bool P_LDS, bool DEEP, class PT, class AT>
__device__ __forceinline__ AT run_program(const SolveParams& S, const PT* P, const uint4* prog, uint32_t stride, uint32_t chunks,
                                          AT valid, uint64_t d0) {
    // HBM-resident P was written with L2 atomics by other waves: read it past this CU's L1
    auto ld = [&](uint32_t slot) -> AT {
        return P_LDS ? (AT)P[slot] : (AT)__hip_atomic_load(&P[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    AT acc = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, deep[DEEP ? kMaxBoolDepth : 1];
    uint32_t sp = 0;
    uint4 nx = chunks ? prog[0] : make_uint4(0, 0, 0, 0);
    // four words per trip: the next chunk and this chunk's four presence reads are in flight together, so a trip
    // exposes one memory round trip instead of four (programs are padded to whole chunks with kFopNop)
    for (uint32_t c = 0; c < chunks; c++) {
        const uint32_t w[4] = {nx.x, nx.y, nx.z, nx.w};
        if (c + 1 < chunks) nx = prog[(size_t)(c + 1) * stride];
        AT pv[4];
        bool rare = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t op = w[q] >> 28;
            pv[q] = ld(op < kFopAndPop ? (w[q] & 0x0FFFFFFFu) : 0);
            rare |= op == kFopNot || op == kFopInord;
        }
        const bool any_rare = __any(rare);                      // wave-uniform: only around INORD groups
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t op = w[q] >> 28, a = w[q] & 0x0FFFFFFFu;
            const bool is_s = op < kFopAndPop;
            const AT v = pv[q] ^ (AT)(0 - (AT)((op >> 2) & 1)); // SetN / AndNS / OrNS (bits past the group are never stored)
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
            if (any_rare) {
                if (op == kFopNot) acc = ~acc;
                // candidates: documents where the group's boolean value is true (rval, expression.go:137)
                if (op == kFopInord) acc = (AT)inord_docs(S, a, (uint64_t)(acc & valid), d0);
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
template <bool P_LDS, bool PROG_LDS, int G>
__global__ void __launch_bounds__(kSolveBlockThreads) k_solve_groups(const SolveParams S) {
    using PT = typename PType<G>::type;
    using AT = typename AType<G>::type;                         // document masks of the evaluation
    constexpr uint32_t kTeam = kSolveBlockThreads / G;          // lanes that share one document while P is built
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
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
    // unit range + first slab of this thread's document in the NEXT group (consumed by step 1 of that group)
    uint64_t pf_u0 = 0, pf_u1 = 0, pf_s = 0;
    uint32_t pf_n = 0;
    auto prefetch = [&](uint64_t g) {
        const uint64_t d = g * G + threadIdx.x / kTeam;
        pf_u0 = pf_u1 = 0; pf_n = 0;
        if (g < n_groups && d < S.n_docs) {
            pf_u0 = S.doc_unit_base[d]; pf_u1 = S.doc_unit_base[d + 1];
            if (pf_u1 > pf_u0) { pf_s = S.unit_start[pf_u0]; pf_n = S.unit_count[pf_u0]; }
        }
    };
    prefetch(blockIdx.x);
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t d0 = g * G;
        const uint32_t nd = (uint32_t)(S.n_docs - d0 < (uint64_t)G ? S.n_docs - d0 : (uint64_t)G);

        // ---- 1. presence matrix ------------------------------------------------------------------------------
        // a team of kTeam lanes per document, all documents of the group at once.  The document's unit range and its
        // first unit's slab were fetched while the previous group was evaluated (pf_*); four slab entries per lane are
        // in flight at a time
        for (uint32_t j = threadIdx.x / kTeam; j < nd && !(S.dbg & 1); j += kSolveBlockThreads / kTeam) {
            const uint32_t member = threadIdx.x % kTeam;
            const uint64_t d = d0 + j;
            const bool pf = j == threadIdx.x / kTeam;              // first (normally only) document of this team
            const uint64_t u0 = pf ? pf_u0 : S.doc_unit_base[d], u1 = pf ? pf_u1 : S.doc_unit_base[d + 1];
            for (uint64_t u = u0; u < u1; u++) {
                const uint64_t s = (pf && u == u0) ? pf_s : S.unit_start[u];
                const uint32_t n = (pf && u == u0) ? pf_n : S.unit_count[u];
                for (uint32_t i = member; i < n; i += 4 * kTeam) {
                    uint32_t t[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) t[q] = i + q * kTeam < n ? S.term[s + i + q * kTeam] : 0xFFFFFFFFu;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (t[q] != 0xFFFFFFFFu) {
                            const size_t bp = (size_t)t[q] * G + j;              // bit j of element t
                            if (P_LDS) __hip_atomic_fetch_or(&Pw[bp >> 5], 1u << (bp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            else atomicOr(&Pw[bp >> 5], 1u << (bp & 31));
                        }
                }
            }
            if (S.x_off) {
                const uint64_t x0 = S.x_off[d], x1 = S.x_off[d + 1];
                for (uint64_t i = x0 + member; i < x1; i += kTeam) {
                    const size_t bp = (size_t)S.x_slot[i] * G + j;
                    if (P_LDS) __hip_atomic_fetch_or(&Pw[bp >> 5], 1u << (bp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else atomicOr(&Pw[bp >> 5], 1u << (bp & 31));
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
            for (uint32_t b16 = 0; b16 < nblk && !(S.dbg & 2); b16 += kWaves) {
                const uint32_t row = wave >> 2, c4 = wave & 3;
                const uint32_t b = b16 + row * 4 + ((row & 1) ? 3 - c4 : c4);
                const uint32_t i = b * 64 + lane;
                if (b < nblk && i < ne) {
                    const uint32_t e = PROG_LDS ? lorder[e0 + i] : S.order[e0 + i];
                    const uint64_t po = PROG_LDS ? loff[e] : S.fprog_off[e];
                    const uint32_t len = (uint32_t)((PROG_LDS ? loff[e + 1] : S.fprog_off[e + 1]) - po);
                    // programs in LDS: linear; in global memory: the block's 4-word chunks transposed ([chunk][lane]) so
                    // that the lanes of a wave read consecutive 16-byte pieces
                    const uint4* prog = reinterpret_cast<const uint4*>(PROG_LDS ? lprog + po : S.fprog_t + S.fblk_off[(e0 >> 6) + b]) +
                                        (PROG_LDS ? 0 : lane);
                    const uint32_t stride = PROG_LDS ? 1u : 64u;
                    const bool deep = S.blk_deep[(e0 >> 6) + b] != 0;              // wave-uniform
                    R[e - e0] = deep ? run_program<P_LDS, true, PT, AT>(S, P, prog, stride, len / 4, (AT)valid, d0)
                                     : run_program<P_LDS, false, PT, AT>(S, P, prog, stride, len / 4, (AT)valid, d0);
                }
            }
            __syncthreads();
            // 2b. transpose in natural order: lane j ends up with the two bitmap words of document j for 64 expressions
            const uint32_t rounds = (tw + 1) / 2;
            for (uint32_t r = wave; r < rounds; r += kWaves) {
                const uint32_t el = r * 64 + lane;
                const uint64_t acc = (el < ne && !(S.dbg & 2)) ? R[el] : 0;
                const uint64_t mine = (S.dbg & 4) ? 0 : wave_transpose64(acc);
                O[lane * tile_words + r * 2] = (uint32_t)mine;
                if (r * 2 + 1 < tw) O[lane * tile_words + r * 2 + 1] = (uint32_t)(mine >> 32);
            }
            __syncthreads();
            // rows of the tile -> global bitmap
            for (uint32_t i = threadIdx.x; i < nd * tw; i += kSolveBlockThreads) {
                const uint32_t j = i / tw, c = i - j * tw;
                S.bitmap[(d0 + j) * bm_words + w0 + c] = O[j * tile_words + c];
            }
            __syncthreads();
        }

        // ---- 3. clear the touched entries of P for the next group ---------------------------------------------------
        if (P_LDS) {
            // LDS: wiping the whole matrix is a handful of wide stores per lane, no HBM re-read of the matches
            uint4* P4 = reinterpret_cast<uint4*>(P);
            for (uint32_t i = threadIdx.x; i < (uint32_t)(((size_t)S.n_slots * sizeof(PT) + 15) / 16); i += kSolveBlockThreads) P4[i] = make_uint4(0, 0, 0, 0);
        } else {
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
        }
        __syncthreads();
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
    const Kern fn = p_in_lds ? (prog_in_lds ? k_solve_groups<true, true, G> : k_solve_groups<true, false, G>)
                             : (prog_in_lds ? k_solve_groups<false, true, 64> : k_solve_groups<false, false, 64>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    fn<<<dim3(grid), dim3(kSolveBlockThreads), lds, st>>>(S);
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
