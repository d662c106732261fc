// Host-side compiler: keyword set -> flattened Aho-Corasick tables for the HIP kernels.
// Replaces forkahocorasick.NewStringMatcher (call site finder/substringEngine.go:103) with a layout made for
// the GPU: BFS-numbered states (shallow = small ids, children of a state contiguous in class order),
// byte -> equivalence-class remap, class-compressed full DFA rows, dictionary-suffix ("output") links.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gft {

constexpr uint32_t kNoTerm = 0xFFFFFFFFu;
constexpr uint32_t kOutFlag = 0x80000000u;  // delta entry: target state emits at least one term

struct AcTables {
    std::vector<std::string> terms;    // sorted unique; index == term id
    uint32_t n_classes = 1;            // class 0 = every byte that occurs in no term
    uint8_t byte_class[256] = {0};
    uint32_t n_states = 1;
    uint32_t max_term_len = 0;
    std::vector<uint32_t> delta;       // [n_states * n_classes], next state | kOutFlag
    std::vector<uint32_t> out_term;    // [n_states] term id ending exactly at this state, or kNoTerm
    std::vector<uint32_t> out_link;    // [n_states] nearest proper-suffix state that is terminal, 0 = none
    std::vector<uint32_t> term_len;    // [n_terms]
    std::vector<uint32_t> depth;       // [n_states]
    std::vector<uint32_t> fail;        // [n_states]
    std::vector<uint32_t> child_begin; // [n_states + 1] children of s are states [child_begin[s], child_begin[s+1])
    std::vector<uint8_t> in_class;     // [n_states] class of the edge entering the state (root: 0)
};

// terms may contain duplicates and the empty string (which never matches, dsl/scanner.go:211-212)
void build_ac_tables(std::vector<std::string> terms, AcTables& out);

}  // namespace gft
