// Host-side compiler for the "suffix-window" scan kernel (gft_scan2.hip).
//
// The Aho-Corasick automaton truncated at depth 4 is a 4-local machine: which dictionary terms can have a given one of
// their 4-byte windows END at text position p is a function of the last four byte classes only.  That function is
// tabulated:
//   * filter   one bit per 4-class window (direct-indexed, or hashed when the alphabet is large): "some term may be
//              anchored here".  LDS, probed once per text byte, no dependent chain between bytes.
//   * short3   byte per 3-class window: which terms of length <= 3 end here (id of a small record holding up to three
//              terms, longest first).  LDS; answers the bulk of all matches without leaving the CU.
//   * fpt      byte per (window, byte in front of the window) that some term of length >= 4 has (cuckoo placement): a
//              7-bit tag of the window key; terms whose window is their first four bytes are keyed by the window alone.
//              LDS; rejects most positions where the window matches but the byte in front does not, before any L2
//              access -- also for windows shared by several terms.
//   * slots    window -> the terms of length >= 4 anchored at exactly that window, longest first.  Normally one: the
//              build shifts a term's window up to kScan2MaxOff bytes away from its end to the rarest, untaken one
//              (pick_off).  L2, 32-byte slots, two-choice placement; a slot carries the term's bytes in front of the
//              window and behind it (terms up to 24 bytes inline, 20 when shifted), so every match is found from one
//              fixed position inside it and no failure links are needed.
// Same inputs as NewStringMatcher (finder/substringEngine.go:103); same outputs as MatchAll (:111-116).
#pragma once
#include <cstdint>
#include <vector>

#include "ac_tables.hpp"
#include "gft_kernels.hpp"

namespace gft {

constexpr uint32_t kWin = 4;                        // window length in bytes
constexpr uint32_t kFilterDirectMaxBits = 1u << 19 | 1u << 18;   // 768 Kbit = 96 KB of LDS for a direct-indexed filter
constexpr uint32_t kFilterHashedBits = 1u << 19;    // 64 KB when hashed
constexpr uint64_t kMaxWindowKeys = 1u << 21;       // budget for the expansion of terms shorter than the window

struct Scan2Tables {
    bool supported = false;          // gft_scan2.hip / gft_scan4.hip can run on these tables
    bool long_ok = false;            // ... the tables of the terms of length >= 4 (slots, more, fpt, classes) are complete: what
                                     // gft_scan5.hip needs; with short_direct == false its short terms come from Scan3Tables
    bool short_direct = true;        // short3 (K^3 bytes, exact classes) exists; false: too many byte classes for it
    const char* why_not = "";
    uint32_t kp = 0;                 // K' = n_classes + 1 (the extra class is PAD = "before the document start")
    uint32_t pad_class = 0;
    bool hashed = false;
    uint32_t filter_bits = 0;        // number of bits in the filter
    uint32_t hash_shift = 0;         // hashed: index = (key * kGold) >> hash_shift
    std::vector<uint32_t> filter;    // filter_bits / 32 words
    std::vector<uint8_t> short3;     // [kp^3 rounded up to 16] record id (0 = none) per 3-window
    std::vector<Scan2Short> shorts;  // record 0 unused
    std::vector<uint32_t> short3_big;      // [short3.size()] full record id per 3-window; empty unless > 254 records exist
    std::vector<uint32_t> shorts_packed;   // 3 words per record: term_id | len << 28, longest first, 0 = none
    std::vector<uint8_t> fpt;        // [kScan2FptSize] (fpt_lg == 0, LDS) or [2^fpt_lg] (global)
    uint32_t fpt_lg = 0;
    uint32_t slot_shift = 0;         // slot index = scan2_pair_slot(key, 0 or 1, slot_shift, slot_seed): the two slots of one 64-byte pair
    uint32_t slot_seed = 0;
    std::vector<Scan2Slot> slots;    // power-of-two table, terms of length >= 4 only
    std::vector<Scan2Slot> more;     // entry lists of multi-term buckets
    uint8_t cls[256];                // byte -> class
    uint8_t cls_fold[256];           // byte -> class of its ASCII lower-case form
    std::vector<uint8_t> term_blob;  // raw term bytes (for terms longer than 8)
    std::vector<uint32_t> term_off;  // n_terms + 1
    uint64_t n_keys = 0;
};

constexpr uint32_t kGold = kGoldDev;

void build_scan2_tables(const AcTables& ac, Scan2Tables& out);

// What gft_scan5.hip needs on top of Scan2Tables (derived, never serialised): the filter over 3-grams of G merged byte
// classes (one probe answers two end positions).
struct Scan5Tables {
    uint32_t G = 0, pad_group = 0;
    uint8_t grp[256];                // byte -> filter group
    uint8_t grp_fold[256];           // ... of its ASCII lower-case form
    std::vector<uint64_t> filter;    // [G^3] per 3-gram (b, c, d) of groups: bit a = (a, b, c, d) is flagged by Scan2Tables::filter;
                                     // bit 32 + e = (b, c, d, e) is
};
// G <= s2.kp groups: the classes that are rare in the dictionary share groups (longest-processing-time rule), class 0
// ("other": every byte that no term has, and what stands in front of the blob) keeps one of its own
void build_scan5_tables(const AcTables& ac, const Scan2Tables& s2, uint32_t G, Scan5Tables& out);

}  // namespace gft
