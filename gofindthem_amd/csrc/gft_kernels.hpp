// Kernel parameter blocks + launcher prototypes shared by gft_kernels.hip and gft_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gft {

constexpr uint32_t kScanBlockThreads = 512;      // 8 waves: one work unit per wave
constexpr uint32_t kTextBuf = 8448;              // per-wave LDS text buffer (bytes)
constexpr uint32_t kSolveBlockThreads = 1024;    // 16 waves: 64 documents x 16-lane teams; one expression round per wave
constexpr uint32_t kMaxPairs = 64;               // (slot, theta) pairs alive inside one INORD group
constexpr uint32_t kMaxPairDepth = 32;           // operand-stack depth inside one INORD group
constexpr uint32_t kMaxBoolDepth = 128;          // operand-stack depth of a whole program
// An expression with a wider INORD group is answered by the solver kernel's second phase (gft_solve.hip wide_expr_doc: pairs
// compacted by presence in the lanes, a per-wave scratch region in HBM behind that).  Beyond these limits the expression is
// solved on the host (host_solve.hpp)
constexpr uint32_t kMaxPairsWide = 8192;
constexpr uint32_t kMaxPairDepthWide = 64;       // (a stack entry per lane)

#define GFT_K_INORD_FLAG (1u << 27)
#define GFT_K_SLOT_MASK ((1u << 27) - 1u)

// a work unit: matches whose END offset lies in [lo, hi) of document `doc`
struct Unit { uint32_t doc, lo, hi; };

struct ScanParams {
    const uint8_t* text;
    const uint64_t* doc_off;
    const Unit* units;
    uint64_t n_units;
    // automaton
    const uint8_t* byte_class;   // [256]
    const uint32_t* delta;       // [n_states * n_classes]
    const uint32_t* out_term;
    const uint32_t* out_link;
    const uint32_t* term_len;
    uint32_t n_classes, n_states, n_lds_states, max_term_len;
    uint32_t pos_end, fold;
    uint32_t* nonascii;          // fold only: *nonascii |= 1 when the text holds a byte >= 0x80 (nullptr: not wanted)
    // output
    uint64_t* cursor;            // pool allocation cursor (entries)
    uint64_t pool_cap;
    uint32_t* pool_term;
    uint32_t* pool_pos;
    uint64_t* unit_start;
    uint32_t* unit_count;
};

// internal ("fused") form of an expression program, produced from the public postfix words at
// gft_set_programs time.  An operand that is a plain (or negated) UNIT folds into its operator -- AND / OR commute, so
// this works on either side -- and the accumulator is pushed only between two operands that are both subtrees.  The
// common shapes (the parser is precedence-free and left-associative) are straight runs of ops 1..5.
//   word = op << 28 | operand
enum FusedOp : uint32_t {     // ops < 8 read one presence row: bit 2 = negate it, low two bits = set / and / or
    kFopSet = 1,      // acc = P[slot]
    kFopAndS = 2,     // acc &= P[slot]
    kFopOrS = 3,      // acc |= P[slot]
    kFopSetN = 5,     // acc = ~P[slot]
    kFopAndNS = 6,    // acc &= ~P[slot]
    kFopOrNS = 7,     // acc |= ~P[slot]
    kFopAndPop = 8,   // acc = pop & acc
    kFopOrPop = 9,    // acc = pop | acc
    kFopNot = 10,     // acc = ~acc (only on top of an INORD group: NOT is pushed down to the leaves otherwise)
    kFopInord = 11,   // acc = documents of acc whose INORD group `operand` has a non-empty position list
    kFopPush = 12,    // push acc
    kFopNop = 13,     // padding: every program is a whole number of 4-word chunks
    kFopPushSet = 14, // push acc; acc = P[slot]   (what follows a push is always the first leaf of the next subtree)
    kFopPushSetN = 15 // push acc; acc = ~P[slot]
};
constexpr uint32_t kSolveRegStack = 2;           // accumulator-stack entries the fast interpreter keeps in registers
constexpr uint32_t kSolveRegStackDeep = 4;       // ... and the interpreter of the blocks that nest deeper (beyond: scratch)

// What the kernel reads: the fused words re-coded as CONTROL BITS (fused_to_device, at upload), so that one step of the
// interpreter is straight-line mask arithmetic -- v_bfe_i32 turns a bit into a lane mask, v_bfi / v_and_or do the rest:
//   v = P[field] ^ NEG;  x = POP ? s0 : v;  A = SEL ? x : ONES;  B = OR ? x : 0;  acc = (acc & A) | B
//   set: B = v, A = 0   and: A = v   or: A = ~0, B = v   and-pop / or-pop: the same on s0   push, nop: A = ~0, B = 0
// RARE words (NOT, INORD group `field`) are no-ops of that data flow and take the wave-uniform slow path.
constexpr uint32_t kDwNeg = 1u << 0, kDwSel = 1u << 1, kDwOnes = 1u << 2, kDwOr = 1u << 28, kDwPop = 1u << 29,
                   kDwPush = 1u << 30, kDwRare = 1u << 31;
constexpr uint32_t kDwFieldShift = 3, kDwFieldBits = 25;        // bits 3..27: slot (x 8 = its byte offset in a 64-document P)
constexpr uint32_t kDwFieldMask = ((1u << kDwFieldBits) - 1u) << kDwFieldShift;
constexpr uint32_t kDwNop = kDwOnes;
inline uint32_t fused_to_device(uint32_t fw) {
    const uint32_t field = (fw << kDwFieldShift) & kDwFieldMask;
    switch (fw >> 28) {
    case kFopSet: return field | kDwOr;
    case kFopSetN: return field | kDwOr | kDwNeg;
    case kFopAndS: return field | kDwSel;
    case kFopAndNS: return field | kDwSel | kDwNeg;
    case kFopOrS: return field | kDwOnes | kDwOr;
    case kFopOrNS: return field | kDwOnes | kDwOr | kDwNeg;
    case kFopAndPop: return kDwPop | kDwSel;
    case kFopOrPop: return kDwPop | kDwOnes | kDwOr;
    case kFopPush: return kDwPush | kDwOnes;
    case kFopPushSet: return field | kDwPush | kDwOr;
    case kFopPushSetN: return field | kDwPush | kDwOr | kDwNeg;
    case kFopNot: return kDwRare | kDwOnes | kDwNeg;
    case kFopInord: return kDwRare | kDwOnes | field;
    default: return kDwNop;
    }
}
constexpr uint32_t kSolveTileWords = 64;         // bitmap words (x32 expressions) evaluated per LDS output tile

struct SolveParams {
    // matches: document d owns units [doc_unit_base[d], doc_unit_base[d+1]); unit u owns pool entries
    // [unit_start[u], unit_start[u] + unit_count[u])
    const uint64_t* doc_unit_base;
    const uint64_t* unit_start;
    const uint32_t* unit_count;
    const uint32_t* term;
    const uint32_t* pos;
    const uint64_t* x_off;       // caller-supplied matches (absolute slots), nullable
    const uint32_t* x_slot;
    const uint32_t* x_pos;
    uint64_t n_docs;
    const uint32_t* fprog;       // fused programs, device form (fused_to_device)
    const uint64_t* fprog_off;
    const uint32_t* gprog;       // public postfix words (INORD group subtrees are interpreted from these)
    const uint32_t* groups;      // [n_groups][2] = offset, length into gprog
    const uint32_t* order;       // [n_exprs] evaluation order inside every output tile (gft_set_programs)
    const Unit* units;           // the scan's work units (document of every unit)
    uint32_t has_rare;           // some program holds a NOT or INORD word (0: the kernel variant without the position algebra)
    uint32_t pos_back;           // a match's reported position lies at most this far in front of the unit it ends in: max_term_len - 1
                                 // (GFT_POS_START), 0 (GFT_POS_END) -- where a successor walk over a long document may stop
    const uint32_t* blk_class;   // per 64 sorted programs: the interpreter they need (0 flat, 1 register stack, 2 deep)
    const uint32_t* wave_blk;    // per tile and round of 16 blocks: the block (inside the tile) of every wave, or ~0
    const uint32_t* fprog_t;     // the same programs per sorted block, transposed by chunk: words 4c..4c+3 of lane l at fblk_off[b] + (c * 64 + l) * 4
    const uint32_t* fblk_off;    // (read when the programs do not fit LDS: coalesced instead of one stream per lane)
    uint32_t n_exprs, n_slots, tile_words;
    uint32_t fprog_words;        // total words of fprog (staged in LDS when they fit)
    uint32_t dbg;                // GFT_SOLVE_DEBUG bits (timing studies): 1 skip presence build, 2 skip evaluation, 4 skip transpose/output,
                                 // 8 phase clocks: cycles per phase and wave summed into dbg_out[wave * 8 + phase]
    unsigned long long* dbg_out;
    uint64_t* p_scratch;         // presence matrix in HBM when it does not fit LDS: [grid][n_slots]
    uint32_t* wide_slot;         // pairs of wide INORD groups: [grid * waves][wide_cap] slots ...
    long long* wide_theta;       // ... and thresholds (null: no program has a wide group)
    uint32_t wide_cap;
    const uint32_t* wide_list;   // [n_wide][3] = expression, offset and length of its public words in gprog: the expressions
    uint32_t n_wide;             // with a wide INORD group, which the kernel's second phase answers (wide_expr_doc)
    uint32_t* bitmap;
};

#if defined(__HIPCC__)
#define GFT_HD __host__ __device__
#else
#define GFT_HD
#endif

// ---- suffix-window scan (gft_scan2.hip; tables built by scan2_tables.cpp) ---------------------------------------
constexpr uint32_t kScan2Threads = 1024;         // 16 waves per workgroup share one LDS copy of the filter
constexpr uint32_t kScan2StageCap = 6;           // matches a lane can stage in LDS before the direct-write path
constexpr uint32_t kScan2FifoCap = 256;          // unordered path: matches of one unit buffered in LDS (>= 64 * kScan2StageCap)
constexpr uint32_t kScan2CandCapMin = 512;       // unordered path: flagged positions of one unit listed in LDS (the actual
                                                 // capacity is whatever LDS is left, Scan2Params::cand_cap)
constexpr uint32_t kScan2Slab = 4096;            // pool entries a wave reserves per global atomic
constexpr uint32_t kScan2UnitMax = 8192;         // bytes per work unit (128 per lane)
constexpr uint32_t kGoldDev = 0x9E3779B1u;

// Bucket table: window key -> the terms of length >= 4 that end with that window.  32-byte slots, two-choice (cuckoo)
// placement: a key lives in slot h0(key) or h1(key), the kernel loads both at once, so a lookup is never a chain of
// dependent probes.  A slot describes ONE term completely for lengths <= kScan2InlineLen: the bytes in front of the
// window are stored the way the text loads see them, front[k] = text[p-7-4k .. p-4-4k] as a little-endian dword
// (zero where the term has no byte); longer terms compare the rest against term_blob.  A bucket with several terms
// has kScan2Multi set in `info`: the slot is then only a header, info & ~kScan2Multi indexes `more` (entries in the
// same 32-byte format, longest first == the reference's emission order) and `len` is their count.
constexpr uint32_t kScan2EmptyKey = 0xFFFFFFFFu;   // never a window key of a term (checked at build time)
constexpr uint32_t kScan2Multi = 0x80000000u;
constexpr uint32_t kScan2InlineLen = 24;           // 4 window bytes + 20 front bytes
// Shifted anchors: the window of a term need not be its LAST four bytes.  Among the windows that end `off` = 0 ..
// kScan2MaxOff bytes before the term's end the build picks the one that is rarest in text (fewest flagged positions);
// the bytes behind the window (`tail`) are then checked like the bytes in front of it.  The slot describes the term up
// to the end of its window (len1 = length - off), keeps off in the top byte of `len`, and stores the tail in front[4]
// (so a shifted term carries 16 front bytes inline instead of 20).
constexpr uint32_t kScan2MaxOff = 4;
constexpr uint32_t kScan2LenMask = 0x00FFFFFFu;    // len word: len1 | off << 24
struct __attribute__((aligned(32))) Scan2Slot {
    uint32_t key;
    uint32_t info;       // one term: term_id (< 2^31); several: kScan2Multi | index into `more`
    uint32_t len;        // one term: len1 | off << 24; several: number of entries
    uint32_t front[5];   // off > 0: front[4] = the off bytes behind the window, text[p+1 ..] as a little-endian dword
};
// short3 record: the (up to three) terms of length <= 3 that end at a 3-window, longest first
struct Scan2Short {
    uint32_t n;
    uint32_t term[3];
    uint32_t len[3];
    uint32_t pad;
};
constexpr uint32_t kScan2Short3Max = 32768;      // bytes of LDS a direct 3-window table may take (K' <= 32)
constexpr uint32_t kScan2FptSize = 20480;        // cells (one byte each) of the LDS-resident table
constexpr uint32_t kScan2FptLdsItems = 11000;    // more terms of length >= 4 than this: the table moves to global memory (L2),
                                                 // 2^fpt_lg cells at load <= 0.4 -- the LDS-spill path of large dictionaries
constexpr uint32_t kScan2FptAmbiguous = 0xFF;    // always go to the bucket table
// Fingerprint table: the LDS-only answer to "can a term of length >= 4 be anchored here at all?" for a position whose
// window passed the filter.  One cell per (window, byte in front of it) that some term has (cuckoo placement at build
// time):
//   * a term whose window is its first four bytes has no byte in front: it owns the cell x-hash(key);
//   * any other term is keyed by (window key, b1 = the byte in front of the window): it owns one of the two cells
//     g-hash_{0,1}(key, b1).  Terms that share a window almost always differ in b1; those that do not share the cell.
// The cell byte is a 7-bit tag of the window key (1..128), so the cell index proves b1 and the byte proves the window;
// 0 = empty cell, kScan2FptAmbiguous = pass everything (placement failures).  b1 is taken with its case bit cleared on
// both sides so the same table serves exact and ASCII-folded scans (the bucket table does the exact compare).
// A position passes if its x-cell or either g-cell carries its tag.  All hashing is 24-bit multiplies (v_mul_u32_u24:
// full rate).  (A longer fingerprint -- five bits over up to three more front bytes -- rejected 5 more positions per
// 4 KB document before the bucket table but cost twice the instructions in stage A: measured slower.)
// low 32 bits of (a mod 2^24) * (C mod 2^24).  On the device this must be v_mul_u32_u24 (full rate); the compiler
// tends to pick the quarter-rate v_mul_lo_u32 for the generic form, hence the explicit instruction.
template <uint32_t C>
GFT_HD inline uint32_t scan2_mul24c(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "n"(C), "v"(a));
    return d;
#else
    return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(C & 0xFFFFFFu));
#endif
}
// the two candidate slots of a window key (table of 2^lg slots, shift = 32 - lg).  `seed` is chosen by the table builder:
// a handful of structured keys can share both slots with others; another seed separates them without growing the table
GFT_HD inline uint32_t scan2_slot_hash(uint32_t x, int which, uint32_t shift, uint32_t seed) {
    const uint32_t xf = (x ^ (x >> 20)) + seed;    // keys beyond 24 bits (hashed alphabets) keep their top bits in play
    return (which ? scan2_mul24c<0x85EBCBu>(xf) : scan2_mul24c<0x9E3779u>(xf)) >> shift;
}
// Scan2Tables (round 4): a key's two candidate slots are the two slots of ONE 64-byte pair -- one request of the L1 instead of
// two at random places of the table.  A pair holds two keys; the table builder gives a term another anchor window when its
// pair is full (scan2_tables.cpp), so no key ever lives anywhere else and a lookup is still one round trip.  (Scan3Tables
// keeps the two independent choices of scan2_slot_hash: two anchors per term leave it no window to trade.)
GFT_HD inline uint32_t scan2_pair_slot(uint32_t x, int which, uint32_t shift, uint32_t seed) {
    const uint32_t xf = (x ^ (x >> 20)) + seed;
    return ((scan2_mul24c<0x9E3779u>(xf) >> shift) & ~1u) | (uint32_t)which;
}
// fpt_lg == 0: the LDS table of kScan2FptSize cells; else a global table of 2^fpt_lg cells
GFT_HD inline uint32_t scan2_fpt_index(uint32_t h, uint32_t fpt_lg) {
    return fpt_lg ? h >> (32 - fpt_lg) : scan2_mul24c<kScan2FptSize>(h >> 16) >> 16;
}
GFT_HD inline uint32_t scan2_fpt_xcell(uint32_t x, uint32_t fpt_lg) { return scan2_fpt_index(scan2_mul24c<0x3779B1u>(x), fpt_lg); }
GFT_HD inline uint32_t scan2_fpt_gcell(uint32_t x, uint32_t b1n, int which, uint32_t fpt_lg) {
    return scan2_fpt_index(which ? scan2_mul24c<0xB2AE35u>(x) + scan2_mul24c<0x9E4F2Du>(b1n)
                                 : scan2_mul24c<0xC2B2AFu>(x) + scan2_mul24c<0x27D4EBu>(b1n), fpt_lg);
}
GFT_HD inline uint32_t scan2_fpt_xmix(uint32_t x) { return scan2_mul24c<0xD4EB2Fu>(x); }
GFT_HD inline uint32_t scan2_fpt_xbyte(uint32_t xmix) { return 1u + (xmix >> 25); }   // 1..128: never empty, never ambiguous
GFT_HD inline bool scan2_fpt_pass(uint32_t cx, uint32_t cg0, uint32_t cg1, uint32_t xmix, uint32_t tw) {
    (void)tw;
    const uint32_t tag = scan2_fpt_xbyte(xmix);
    const bool amb = (cx == kScan2FptAmbiguous) | (cg0 == kScan2FptAmbiguous) | (cg1 == kScan2FptAmbiguous);
    return (cx == tag) | (cg0 == tag) | (cg1 == tag) | amb;
}

struct Scan2Params {
    const uint8_t* text;
    const uint64_t* doc_off;
    const Unit* units;
    uint64_t n_units;
    const uint32_t* filter;
    uint32_t filter_words, hashed, hash_shift;
    const uint8_t* short3;       // [short3_bytes] record id per 3-window, copied to LDS (short3_bytes == 0: no short terms)
    uint32_t short3_bytes;
    const uint32_t* shorts_packed;   // 3 words per record (term_id | len << 28, longest first, 0 = none), copied to LDS
    uint32_t shorts_words;       // words staged in LDS (at most 255 records)
    const uint32_t* short3_big;  // full record id per 3-window for LDS byte 255 (nullptr: every record has an LDS id)
    uint32_t cand_cap;           // entries of a wave's LDS candidate list (scan2_plan)
    const uint8_t* fpt;          // fpt_lg == 0: [kScan2FptSize], copied to LDS; else [2^fpt_lg], read in place (L2)
    uint32_t fpt_lg;
    const Scan2Slot* slots;      // 2^lg slots, slot_shift = 32 - lg
    uint32_t slot_shift, slot_seed;
    const Scan2Slot* more;
    const uint8_t* cls;          // [256] byte -> class (the folded table when GFT_FOLD_ASCII)
    const uint8_t* term_blob;
    const uint32_t* term_off;
    uint32_t kp, pad_class, fold, pos_end;
    uint64_t* cursor;            // pool allocation cursor (entries, slab granular)
    uint64_t pool_cap;
    uint32_t* pool_term;
    uint32_t* pool_pos;
    uint64_t* unit_start;
    uint32_t* unit_count;
    uint64_t* n_matches;         // exact number of matches (the cursor includes slab slack)
    uint32_t slab;               // pool entries a wave reserves per global atomic (<= kScan2Slab)
    uint32_t ordered;            // 1: per-lane staging path for every unit (cross-check); 0: balanced path, staging only on overflow
    uint64_t text_bytes;         // size of the text blob (loads behind a window must not run past it)
    uint32_t prio;               // 1 (default): graded wave priorities (s_setprio) -- filter 0 < list build 1 < stage A 2 <
                                 // stage B 3; 0: off
    uint32_t want_pos;           // 0: presence only -- no expression has an INORD group, pool_pos is not written
    uint32_t dbg;                // GFT_SCAN_DEBUG bits (timing studies only): 1 = skip verification, 2 = count flags
    uint32_t* nonascii;          // fold only: *nonascii |= 1 when the text holds a byte >= 0x80 (nullptr: not wanted)
    uint64_t* dbg_counters;      // [4] when dbg & 2: flagged positions, table probes, entries compared, -
    // gft_scan4.hip (the streaming form) only:
    uint32_t chunk_units;        // units a wave streams through in one go (1 .. kScan4ChunkUnits)
    uint32_t bound_q16, bound_add;   // a unit's region of the match pool: bytes * bound_q16 / 65536 + bound_add entries
    uint32_t round_c;            // bytes per lane and round (16, 32, 48 or 64): a round is 64 x round_c bytes of the stream
    // gft_scan5.hip (one filter probe per two bytes) only:
    const uint8_t* s5_grp;       // [256] byte -> filter group (the folded table when GFT_FOLD_ASCII), copied to LDS
    uint32_t s5_G, s5_pad_g;     // groups; the group of class 0 (what stands in front of a document)
    const uint64_t* s5_filter;   // [s5_dual = G^3] per 3-gram of groups: low word bit a = some anchor window is (a, 3-gram), high
    uint32_t s5_dual;            // word bit d = some anchor window is (3-gram, d); copied to LDS
    uint32_t s5_fifo_cap;        // entries (4 B) of a wave's LDS match fifo
    // ... over an alphabet too large for the direct short-term table (s5_sG != 0): short3 / shorts_packed / short3_big are the
    // group-indexed tables of scan3_tables.hpp (short3, srec, short3_big), plus:
    const uint8_t* s5_sgrp;      // [256] byte -> group of those tables (the folded table when GFT_FOLD_ASCII), copied to LDS
    uint32_t s5_sG;              // their number of groups (0: the direct table of exact classes)
    const uint32_t* s5_srec_big; // records of the cells with id 255 ({n, n x 2 words})
    uint32_t s5_term_bits, s5_pos_bias;   // with positions a fifo entry is term | (pos - (unit.lo - pos_bias)) << term_bits
    // ... with the fingerprint table in global memory (fpt_lg != 0): a Bloom level of 2^s5_bloom_lg bits in LDS in front of it
    const uint32_t* s5_bloom;    // bit scan5_bloom_g(key, b1) / scan5_bloom_x(key) set for every cell owner of the fingerprint table
    uint32_t s5_bloom_lg;        // 0: none
};
// Bloom level of gft_scan5.hip's stage A for large dictionaries (one bit per (window, byte in front) that some term of
// length >= 4 has, one per window that is a term's first four bytes): a position whose two bits are both clear cannot anchor
// a long term and spares the three L2 gathers of the global fingerprint table
GFT_HD inline uint32_t scan5_bloom_g(uint32_t x, uint32_t b1n, uint32_t lg) {
    return (scan2_mul24c<0xC2B2AFu>(x ^ (x >> 20)) + scan2_mul24c<0x27D4EBu>(b1n + 1)) >> (32 - lg);
}
GFT_HD inline uint32_t scan5_bloom_x(uint32_t x, uint32_t lg) { return scan2_mul24c<0x3779B1u>(x ^ (x >> 20)) >> (32 - lg); }
// waves per workgroup (16, 12, 8 or 4) and candidate-list capacity that fit lds_max; false if nothing fits
bool scan2_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max,
                uint32_t* waves, uint32_t* cand_cap);
hipError_t launch_scan2(const Scan2Params& P, uint32_t waves, unsigned n_cus, hipStream_t st);
// gft_scan4.hip: the same tables, streamed chunk by chunk.  A chunk's positions are 16-bit: units of at most kScan4UnitMax bytes
constexpr uint32_t kScan4ChunkUnits = 8;
constexpr uint32_t kScan4UnitMax = 8176;
// waves per workgroup and the match fifo's capacity (entries; goes into Scan2Params::cand_cap) that fit lds_max
bool scan4_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max, bool want_pos,
                uint32_t* waves, uint32_t* fifo_cap);
hipError_t launch_scan4(const Scan2Params& P, uint32_t waves, unsigned n_cus, hipStream_t st);
// gft_scan5.hip: the same tables behind a filter over 3-grams of merged byte classes that is probed every other byte
constexpr uint32_t kScan5Waves = 16;             // waves per workgroup (one LDS copy of the tables per CU)
constexpr uint32_t kScan5MaxGroups = 27;         // filter groups: G^3 x 8 bytes of LDS (G <= 32: a group is a bit of a 32-bit word)
constexpr uint32_t kScan5SurvX = 64;              // stage A hands the window keys of a unit's first survivors to stage B through LDS
constexpr uint32_t kScan5CandCapMin = 384;       // flagged positions of one unit listed in LDS at least (more when LDS is left)
struct Scan5Plan { uint32_t G, dual_entries, cand_cap, fifo_cap; };
// filter groups and list capacities that fit lds_max with kScan5Waves waves; false if nothing fits
// fifo_cap: entries of a wave's match fifo (the unit size follows it: a unit's matches should fit)
bool scan5_plan(uint32_t kp, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max, uint32_t fifo_cap, Scan5Plan* out);
hipError_t launch_scan5(const Scan2Params& P, unsigned n_cus, hipStream_t st);


// ---- stride-2 suffix-window scan (gft_scan3.hip; tables built by scan3_tables.cpp) ---------------------------------
// The text is probed at every OTHER position: a probe at p looks at the window text[p-3 .. p] and finds every term that
// ends at p or at p-1 (terms of length <= 3, from the short3 records) and every term of length >= 4 that has one of its
// two anchor windows end at p.  Each such term has two anchors whose distances to the term's end differ in parity, so
// whatever the parity of an occurrence, exactly one of its anchors falls on a probe.  Byte classes are merged down to
// at most kScan3Groups filter groups, so the LDS tables are direct-indexed for every alphabet; exactness comes from the
// byte compares of the bucket table (and of the short records), never from a hash.
constexpr uint32_t kScan3Groups = 27;            // filter groups: G^4 bits = 66 KB, G^3 bytes = 20 KB of LDS at 27
constexpr uint32_t kScan3Threads = 1024;         // 16 waves per workgroup share one LDS copy of the tables
constexpr uint32_t kScan3UnitMax = 8192;         // bytes per work unit: up to 8 coalesced rounds of 1 KiB (16 B per lane)
constexpr uint32_t kScan3RecWords = 6;           // short record: 3 entries x {term_id | len << 28, bytes}
constexpr uint32_t kScan3RecLds = 254;           // record ids 1..254 live in LDS; cell 255 = look the cell up in short3_big
constexpr uint32_t kScan3BloomLdsLg = 13;        // Bloom table in LDS: 2^13 cells x 4 B = 32 KB; larger dictionaries: global
constexpr uint32_t kScan3SurvCap = 128;          // stage-B candidates parked per wave ({entry, window key}, 8 B each)
constexpr uint32_t kScan3MinRoom = 512;          // a unit starts with at least this much room in the wave's slab
constexpr uint32_t kScan3CandCapMin = 256;       // per-wave candidate list (u16 probe indices); the rest of LDS goes here
constexpr uint32_t kScan3BloomMul1 = 0x9E3779B1u, kScan3BloomMul2 = 0x85EBCA6Bu;
// Bloom cell and the two bits of a 5-group key (window key * G + group of the byte in front of the window)
GFT_HD inline uint32_t scan3_bloom_cell(uint32_t key5, uint32_t lg) { return (key5 * kScan3BloomMul1) >> (32 - lg); }
GFT_HD inline uint32_t scan3_bloom_bits(uint32_t key5) {
    const uint32_t h = key5 * kScan3BloomMul2;
    return 1u << (h >> 27) | 1u << ((h >> 22) & 31);
}

struct Scan3Params {
    const uint8_t* text;
    const uint64_t* doc_off;
    const Unit* units;
    uint64_t n_units;
    uint64_t text_bytes;
    const uint8_t* cls;          // [256] byte -> filter group (the folded table when GFT_FOLD_ASCII), copied to LDS
    const uint32_t* filter;      // G^4 bits, copied to LDS
    uint32_t filter_words;
    const uint8_t* short3;       // [short3_bytes] record id per 3-group END window (0 none, 255 -> short3_big), copied to LDS
    uint32_t short3_bytes;       // 0: no term shorter than 4 bytes
    const uint32_t* srec;        // LDS records, kScan3RecWords words each (record 0 = empty)
    uint32_t srec_words;
    const uint32_t* short3_big;  // [G^3] offset into srec_big for cells whose record is not in LDS (nullptr: none)
    const uint32_t* srec_big;    // {n, n x {term_id | len << 28, bytes}} records in global memory
    const uint32_t* bloom;       // 2^bloom_lg cells
    uint32_t bloom_lg, bloom_lds;
    const Scan2Slot* slots;
    uint32_t slot_shift, slot_seed;
    const Scan2Slot* more;
    const uint8_t* term_blob;
    const uint32_t* term_off;
    uint32_t G, fold, pos_end, want_pos, grouped, prio;
    uint32_t cand_cap;           // entries of a wave's LDS candidate list (scan3_plan)
    uint32_t dbg;                // GFT_SCAN_DEBUG (timing studies; selects the kernel instantiation that has the knock-outs)
    uint32_t* nonascii;          // fold only: *nonascii |= 1 when the text holds a byte >= 0x80 (nullptr: not wanted)
    uint64_t* cursor;
    uint64_t pool_cap;
    uint32_t* pool_term;
    uint32_t* pool_pos;
    uint64_t* unit_start;
    uint32_t* unit_count;
    uint64_t* n_matches;
    uint32_t slab;
};
// waves per workgroup and candidate-list capacity that fit lds_max; false if nothing fits
bool scan3_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t srec_words, uint32_t bloom_lds_bytes, size_t lds_max,
                uint32_t* waves, uint32_t* cand_cap);
hipError_t launch_scan3(const Scan3Params& P, uint32_t waves, unsigned n_cus, hipStream_t st);

// *d_bad |= 1 when a document is longer than 2^32 - 1 bytes (or its offsets descend); such documents get zero units
hipError_t launch_unit_count(const uint64_t* d_doc_off, uint64_t n_docs, uint32_t unit_max, uint32_t* d_cnt,
                             uint32_t* d_bad, hipStream_t st);
// d_out[0..2] = number of units, first and last text offset (read back by the host in one copy)
hipError_t launch_pack_ctl(const uint64_t* d_unit_base, const uint64_t* d_doc_off, uint64_t n_docs, uint64_t* d_out, hipStream_t st);
// unit_max: as given to launch_unit_count (no unit is longer than that, whatever the offsets say)
hipError_t launch_unit_fill(const uint64_t* d_doc_off, uint64_t n_docs, const uint64_t* d_unit_base, Unit* d_units,
                            uint32_t unit_max, hipStream_t st, uint64_t max_units = ~0ull);
hipError_t launch_units_single(const uint64_t* d_doc_off, uint64_t n_docs, uint32_t unit_max, Unit* d_units, uint64_t* d_unit_base,
                               uint32_t* d_ctl32, uint32_t epoch, hipStream_t st);
hipError_t launch_patch_words(uint32_t* d_words, const uint64_t* d_idx, const uint32_t* d_clr, const uint32_t* d_set, uint64_t n, hipStream_t st);
hipError_t launch_clamp_u64(uint64_t* d_v, uint64_t n, uint64_t cap, hipStream_t st);
uint64_t scan_partials_needed(uint64_t n);
// d_out has n+1 entries; d_partial has scan_partials_needed(n) entries
hipError_t launch_exclusive_scan(const uint32_t* d_in, uint64_t n, uint64_t* d_out, uint64_t* d_partial,
                                 hipStream_t st);
size_t scan_units_lds_bytes(uint32_t n_lds_states, uint32_t n_classes);
hipError_t launch_scan_units(const ScanParams& P, unsigned n_cus, hipStream_t st);
// d_units_to_sort != nullptr: the slabs come from gft_scan2's balanced path and are sorted into emission order on the way
hipError_t launch_gather(const uint64_t* d_unit_start, const uint32_t* d_unit_count, const uint64_t* d_unit_out,
                         uint64_t n_units, const uint32_t* d_pool_term, const uint32_t* d_pool_pos, uint32_t* d_term,
                         uint32_t* d_pos, const uint64_t* d_unit_base, uint64_t n_docs, uint64_t* d_match_off,
                         unsigned n_cus, hipStream_t st, const Unit* d_units_to_sort = nullptr,
                         const uint32_t* d_term_len = nullptr, uint32_t pos_end = 0);
// *d_flag |= 2 unless ASCII case folding equals strings.ToLower on every document of text[lo, hi) (see k_fold_safe)
hipError_t launch_fold_safe(const uint8_t* d_text, uint64_t lo, uint64_t hi, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_flag, hipStream_t st);
// unique terms per document of a CSR result (first occurrence order); d_first: grid x n_terms words, all ones
hipError_t launch_unique_terms(bool write, const uint64_t* d_match_off, const uint32_t* d_term, uint64_t n_docs, uint32_t n_terms,
                               uint32_t* d_first, unsigned grid, uint32_t* d_cnt, const uint64_t* d_out_off, uint32_t* d_out_term,
                               hipStream_t st);
// GFT_POS_RUNES: byte offsets of a CSR result -> offsets over []rune(text) (finder/substringEngine.go:44-53)
hipError_t launch_rune_doc_blocks(const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_cnt, hipStream_t st);
hipError_t launch_rune_block_starts(const uint8_t* d_text, const uint64_t* d_doc_off, const uint64_t* d_blk_base, uint64_t n_docs,
                                    uint64_t n_blocks, uint32_t* d_starts, hipStream_t st);
hipError_t launch_pos_to_rune(const uint8_t* d_text, const uint64_t* d_doc_off, const uint64_t* d_blk_base, const uint64_t* d_blk_prefix,
                              const uint64_t* d_match_off, uint64_t n_docs, uint64_t n_matches, uint32_t* d_pos, hipStream_t st);
size_t solve_lds_bytes(uint32_t n_slots, uint32_t tile_words, uint32_t group_docs, bool p_in_lds, uint32_t prog_words,
                       uint32_t n_exprs, bool prog_in_lds);
// group_docs: documents per group (64, 32, 16 or 8 = bits per presence-matrix element)
hipError_t launch_solve(const SolveParams& S, uint32_t group_docs, bool p_in_lds, bool prog_in_lds, unsigned grid, hipStream_t st);

}  // namespace gft
