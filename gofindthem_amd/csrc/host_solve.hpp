// host_solve.hpp -- Expression.solve (dsl/expression.go:66-142, 175-225) on the host, for the few (expression, document)
// pairs the device solver does not answer itself:
//   * expressions beyond the device solver's limits (include/gft.h: an INORD group wider than 64 pairs, an operand stack
//     deeper than 128) -- the reference's recursion has no such limits;
//   * INORD expressions over a slot whose position list is NOT ascending in some document: a keyword and a regex with the same
//     literal share one map key, keyword positions first, regex positions behind them (finder/finder.go:181-196), and
//     getLowestIdxGTVal's binary search (dsl/expression.go:175-189) then runs over an unsorted list.  The device keeps one
//     sorted position set per slot (successor queries); on such a list the two differ, and the reference's answer is the
//     one that counts.
// This is the reference's algorithm restated over the public postfix words of include/gft.h -- lists are materialised,
// AND keeps a suffix of the right list, OR merges with the reference's tie rule -- so it reproduces the reference on ANY
// input, sorted or not.  Product code (not the test oracle); it runs on matches the GPU scan produced.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace gft {

// one document's sortedMatchesByKeyword: slot -> positions in the order addMatchesToSolverMap appended them
using SlotLists = std::unordered_map<uint32_t, std::vector<int64_t>>;

// Solve of one expression (public postfix words, include/gft.h) over one document's map
bool host_solve(const uint32_t* words, uint64_t len, const SlotLists& m);

// what gft_set_programs learns about one program besides its validity
struct ProgramTraits {
    bool over_limit = false;                 // exceeds a limit of the device solver: always solved on the host
    uint32_t wide_pairs = 0;                 // > 0: an INORD group of more than kMaxPairs (slot, theta) pairs alive at once (or a pair
                                             // stack deeper than kMaxPairDepth) -- the device keeps such a group's pairs in a scratch
                                             // region of this many pairs per wave instead of one pair per lane
    std::vector<uint64_t> wide_groups;       // ... which groups: word index (inside the program) of their closing INORD word, ascending
    std::vector<uint32_t> inord_slots;       // slots read inside INORD groups of more than one leaf (sorted, unique):
                                             // documents in which one of them has a non-ascending list go to the host
};

}  // namespace gft
