// host_solve.cpp -- see host_solve.hpp.  dsl/expression.go:66-142 (solve), :175-189 (getLowestIdxGTVal), :192-225
// (mergeArraysSorted), evaluated over the postfix form with an explicit stack: the reference's own benchmark builds
// left-deep chains of 10 000 leaves (benchmarks/benchmark_test.go:56), which must not recurse here.
#include "host_solve.hpp"

#include <deque>

#include "../../include/gft.h"

namespace gft {

namespace {

struct Val {
    bool v;
    const int64_t* pos;      // the node's position list ([]int of the reference); nullptr / 0 = nil
    size_t n;
};

// dsl/expression.go:175-189: index of the lowest element greater than `value` -- by binary search, whatever the order of
// the list is (on an unsorted list this is NOT the first such element, and the reference's answer is what it finds)
int64_t lowest_idx_gt(const int64_t* a, size_t n, int64_t value) {
    int64_t left = 0, right = (int64_t)n - 1, found = -1;
    while (left <= right) {
        const int64_t half = (left + right) >> 1;
        if (a[half] > value) { found = half; right = half - 1; }
        else left = half + 1;
    }
    return found;
}

}  // namespace

bool host_solve(const uint32_t* w, uint64_t len, const SlotLists& m) {
    std::vector<Val> st;
    std::deque<std::vector<int64_t>> merged;         // lists that OR nodes created (stable addresses)
    for (uint64_t pc = 0; pc < len; pc++) {
        const uint32_t op = w[pc] >> 28;
        const bool inord = (w[pc] & GFT_INORD_FLAG) != 0;
        switch (op) {
        case GFT_OP_UNIT: {                          // :68-72: key presence; the list as it stands
            const auto it = m.find(w[pc] & GFT_SLOT_MASK);
            if (it == m.end()) st.push_back(Val{false, nullptr, 0});
            else st.push_back(Val{true, it->second.data(), it->second.size()});
            break;
        }
        case GFT_OP_AND: {                           // :74-95
            const Val r = st.back(); st.pop_back();
            const Val l = st.back(); st.pop_back();
            Val o{l.v && r.v, nullptr, 0};
            if (inord && l.n > 0 && r.n > 0) {
                const int64_t idx = lowest_idx_gt(r.pos, r.n, l.pos[0]);
                if (idx >= 0) { o.pos = r.pos + idx; o.n = r.n - (size_t)idx; }
            }
            st.push_back(o);
            break;
        }
        case GFT_OP_OR: {                            // :97-116
            const Val r = st.back(); st.pop_back();
            const Val l = st.back(); st.pop_back();
            Val o{l.v || r.v, nullptr, 0};
            if (inord) {
                if (l.n == 0) { o.pos = r.pos; o.n = r.n; }              // :195-200: the other slice itself
                else if (r.n == 0) { o.pos = l.pos; o.n = l.n; }
                else {
                    merged.emplace_back(l.n + r.n);
                    std::vector<int64_t>& out = merged.back();
                    size_t li = 0, ri = 0;
                    for (size_t c = 0; c < out.size(); c++) {            // :206-222: on a tie the RIGHT element goes first
                        if (li == l.n) out[c] = r.pos[ri++];
                        else if (ri == r.n) out[c] = l.pos[li++];
                        else if (l.pos[li] < r.pos[ri]) out[c] = l.pos[li++];
                        else out[c] = r.pos[ri++];
                    }
                    o.pos = out.data(); o.n = out.size();
                }
            }
            st.push_back(o);
            break;
        }
        case GFT_OP_NOT: {                           // :118-127
            const Val r = st.back(); st.pop_back();
            st.push_back(Val{!r.v, nullptr, 0});
            break;
        }
        case GFT_OP_INORD: {                         // :129-137
            const Val r = st.back(); st.pop_back();
            st.push_back(Val{r.v && r.n > 0, nullptr, 0});
            break;
        }
        default:
            break;                                   // (programs are validated by gft_set_programs)
        }
    }
    return !st.empty() && st.back().v;
}

}  // namespace gft
