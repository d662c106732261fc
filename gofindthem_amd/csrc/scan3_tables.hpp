// Host-side compiler for the stride-2 suffix-window scan kernel (gft_scan3.hip).
//
// Same idea as scan2_tables.hpp -- the Aho-Corasick automaton cut at depth 4 is a 4-local machine, so which terms can be
// anchored at a text position is a function of the last few bytes only and can be tabulated -- re-cut for the costs
// measured on gfx950 (tools/ubench): a random LDS probe costs a CU as much as ~12 VALU instructions, so the text is
// probed at every OTHER position and every probe answers for two end positions:
//   * groups   byte classes merged down to <= kScan3Groups filter groups (most frequent classes keep a group of their
//              own), so every LDS table below is direct-indexed whatever the alphabet; a dictionary over <= 26 distinct
//              bytes is filtered exactly, a larger one approximately (the byte compares downstream stay exact).
//   * filter   one bit per 4-group window ending at a probe position p: some term ends at p or p-1 (length <= 3), or a
//              term of length >= 4 has an anchor window here.  LDS.
//   * anchors  every term of length >= 5 has TWO anchor windows, ending off0 (even) and off1 (odd) bytes before its end:
//              whatever the parity of an occurrence's end, exactly one of them ends at a probe.  A 4-byte term is its own
//              even anchor; its odd one is the window one byte further on (off = -1: three term bytes + any byte).
//   * short3   byte per 3-group window ending at e: id of the record of the terms of length <= 3 that end there (record =
//              up to three {term, bytes}; the bytes make the table exact under merged groups).  LDS.
//   * bloom    two bits per (anchor window, group of the byte in front of it) in a 32-bit cell: one LDS probe decides
//              whether a flagged window goes to the bucket table at all.  LDS up to ~11 k long terms, else global (L2).
//   * slots    window key -> terms anchored there (scan2's 32-byte two-choice bucket table, unchanged format; off is a
//              signed byte now).  L2.
// Same inputs as NewStringMatcher (finder/substringEngine.go:103); same outputs as MatchAll (:111-116).
#pragma once
#include <cstdint>
#include <vector>

#include "ac_tables.hpp"
#include "gft_kernels.hpp"

namespace gft {

struct Scan3Tables {
    bool supported = false;
    const char* why_not = "";
    uint32_t G = 1;                  // filter groups in use (<= kScan3Groups); group 0 = bytes that occur in no term
    bool grouped = false;            // several byte classes share a group: window bytes are compared explicitly
    uint8_t cls[256];                // byte -> group
    uint8_t cls_fold[256];           // byte -> group of its ASCII lower-case form
    std::vector<uint32_t> filter;    // G^4 bits
    std::vector<uint8_t> short3;     // [G^3 rounded up to 16] record id per 3-group window (empty: no short terms)
    std::vector<uint32_t> srec;      // LDS records (kScan3RecWords each), record 0 empty
    std::vector<uint32_t> short3_big;    // [short3.size()] offset into srec_big (cells with id 255); empty if unused
    std::vector<uint32_t> srec_big;      // {n, n x 2 words}
    std::vector<uint32_t> bloom;     // 2^bloom_lg cells
    uint32_t bloom_lg = kScan3BloomLdsLg;
    uint32_t slot_shift = 0, slot_seed = 0;
    std::vector<Scan2Slot> slots, more;
    std::vector<uint8_t> term_blob;  // raw term bytes, 4 bytes of slack in front of every term
    std::vector<uint32_t> term_off;  // n_terms + 1
    uint64_t n_keys = 0, n_anchors = 0;
};

void build_scan3_tables(const AcTables& ac, Scan3Tables& out);

// Host emulation of the kernel's table walk for ONE document (test infrastructure for the table compiler; the product
// path is the HIP kernel): every (term_id, position) the tables yield, in no particular order.  `lo` is the parity
// origin (probes at lo + 1, lo + 3, ...), so a test can check both parities.
struct Scan3Hit { uint32_t term, pos; };
void scan3_emulate(const Scan3Tables& t, const uint8_t* text, uint32_t n, uint32_t lo, bool fold, bool pos_end,
                   std::vector<Scan3Hit>& out);

}  // namespace gft
