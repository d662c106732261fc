// gft_api.cpp -- the C ABI of include/gft.h: engine handle, table upload, workspace management and the
// kernel pipelines behind gft_scan* / gft_process*.  Host orchestration only; all per-byte and per-match work
// happens in gft_kernels.hip.  There is no CPU fallback: every compute entry point needs a HIP device.
#include "../../include/gft.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <cstdlib>

#include "ac_tables.hpp"
#include "copy_pool.hpp"
#include "gft_guard.hpp"
#include "gft_kernels.hpp"
#include "host_solve.hpp"
#include "scan2_tables.hpp"
#include "scan3_tables.hpp"

using namespace gft;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct ProfCat {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};

}  // namespace

#ifndef GFT_EXTRA_KERNELS
// The earlier suffix-window kernels (gft_scan2.hip, gft_scan4.hip) are cross-checks and study objects: a product build
// does not carry them (python -m gofindthem_amd.build with GFT_EXTRA_KERNELS=1 does).  Without them nothing "fits".
namespace gft {
bool scan2_plan(uint32_t, uint32_t, uint32_t, uint32_t, size_t, uint32_t*, uint32_t*) { return false; }
hipError_t launch_scan2(const Scan2Params&, uint32_t, unsigned, hipStream_t) { return hipErrorNotSupported; }
bool scan4_plan(uint32_t, uint32_t, uint32_t, uint32_t, size_t, bool, uint32_t*, uint32_t*) { return false; }
hipError_t launch_scan4(const Scan2Params&, uint32_t, unsigned, hipStream_t) { return hipErrorNotSupported; }
}  // namespace gft
static constexpr bool kExtraKernels = false;
#else
static constexpr bool kExtraKernels = true;
#endif

struct gft_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    // host -> device staging of large caller buffers: two pinned bounce buffers, filled by a few copy threads while the
    // previous one is on the wire (a hipMemcpy from pageable memory stages through one thread)
    void* pin[2] = {nullptr, nullptr};
    uint64_t* pin_rb = nullptr;            // pinned landing place of the per-batch read-back of the control block
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    std::unique_ptr<gft::CopyPool> copy_pool;   // the threads that fill / empty the bounce buffers (created with the first large copy)
    bool own_stream = false;
    unsigned n_cus = 256;                  // CUs the persistent kernels fill: the device's minus cu_margin
    unsigned n_cus_hw = 256, cu_margin = 0;
    size_t lds_max = 65536;
    mutable std::string err;

    // automaton
    AcTables tab;
    bool built = false;
    uint32_t build_flags = 0;
    uint32_t n_lds_states = 0;
    DevBuf d_byte_class, d_delta, d_out_term, d_out_link, d_term_len;
    // suffix-window scan (gft_scan2.hip); the two-tier DFA kernel above stays as the general fallback
    Scan2Tables s2;
    bool use_scan2 = false;
    DevBuf d_s2_filter, d_s2_slots, d_s2_more, d_s2_cls, d_s2_cls_fold, d_s2_term_blob, d_s2_term_off, d_dbg;
    // stride-2 suffix-window scan (gft_scan3.hip): the default kernel
    Scan3Tables s3;
    bool use_scan3 = false;
    DevBuf d_s3_filter, d_s3_short3, d_s3_srec, d_s3_short3_big, d_s3_srec_big, d_s3_bloom, d_s3_slots, d_s3_more, d_s3_cls,
        d_s3_cls_fold, d_s3_term_blob, d_s3_term_off;
    uint32_t scan3_waves = 0, scan3_cand_cap = 0;
    // control block (64 B): [0] u32 bad-offsets flag, [8] u64 pool cursor, [16] u64 exact match count, [24] u32 a folded scan
    // saw a byte >= 0x80, [32] u64 n_units,
    // [40] u64 first text offset, [48] u64 last text offset -- one memset per batch, one read-back per synchronisation
    DevBuf d_ctl;
    DevBuf d_s2_short3, d_s2_shorts_packed, d_s2_short3_big, d_s2_fpt;
    uint32_t scan2_short3_bytes = 0;
    uint32_t scan2_k2_waves = 0, scan2_cand_cap = 0;    // scan2_plan
    // the streaming form of the suffix-window kernel (gft_scan4.hip): same tables, its own LDS plan; a unit's region of the
    // match pool is sized from the match density (matches per text byte) of the batches before
    bool use_scan4 = false;
    uint32_t scan4_waves = 0, scan4_fifo[2] = {0, 0};   // fifo entries without / with positions
    double scan4_density = 0.06;
    // the suffix-window kernel with the unit's text in LDS (gft_scan5.hip): scan2's tables + a filter over merged classes
    bool use_scan5 = false;
    Scan5Plan s5plan{0, 0, 0, 0};
    Scan5Tables s5;
    DevBuf d_s5_grp, d_s5_grp_fold, d_s5_filter, d_s5_bloom;
    std::vector<uint32_t> s5_bloom;                     // the Bloom level in front of a global fingerprint table (gft_kernels.hpp scan5_bloom_g)
    uint32_t s5_bloom_lg = 0;                           // 2^lg bits; 0: none
    uint32_t opt_scan5_bloom_kb = 32;                   // GFT_SCAN5_BLOOM_KB: its size in LDS (0: none; a power of two up to 64)
    uint32_t s5_term_bits = 0, s5_pos_bias = 0;
    bool s5_short_groups = false;                       // scan5 takes its short terms from the stride-2 kernel's tables (> 32 byte classes)
    uint32_t opt_scan5_large = 1;                       // GFT_SCAN5_LARGE=0: dictionaries over more than 32 byte classes stay on the stride-2 kernel
    uint32_t opt_scan5_fifo = 0;                        // GFT_SCAN5_FIFO: entries of a wave's match fifo (0: 256; timing study)
    uint32_t opt_scan5_groups = 0;                      // GFT_SCAN5_GROUPS: forced number of filter groups (tests)
    uint64_t scan_valid_docs = ~0ull;                   // documents of the last gft_process scan still in the pool (~0: none)
    uint32_t scan2_unit_max = kScan2UnitMax;            // bytes per work unit (adapts to the match density)
    // a scan launched without knowing the unit count / pool need (gft_process*: one read-back per batch, after the solver)
    bool deferred = false;
    uint64_t deferred_unit_cap = 0;
    uint32_t last_nonascii_bits = 0;                    // what the scan kernels said: 1 = bytes >= 0x80 seen, not judged; 2 = judged unsafe
    bool last_nonascii = false;                         // the last GFT_FOLD_ASCII scan ran over text that ASCII folding does not
                                                        // lower-case the way strings.ToLower does (gft_last_nonascii)
    uint64_t last_text_lo = 0, last_text_hi = 0;        // text range of the last scan
    // Environment switches (cross-checks and timing studies) are read when the handle is created and again
    // by gft_build / gft_import_tables / gft_set_programs -- never on the per-batch path
    uint32_t opt_scan_dbg = 0;                          // GFT_SCAN_DEBUG (timing studies)
    uint32_t opt_scan_prio = 1;                         // graded wave priorities in the scan kernels (GFT_SCAN_PRIO=0: off)
    uint32_t opt_scan_ordered = 0;                      // GFT_SCAN_ORDERED=1: scan2's per-lane staging path for every unit
    uint32_t opt_scan4_round = 0;                       // GFT_SCAN4_ROUND: bytes per lane and round of the streaming kernel (0: 64)
    uint32_t opt_scan4_chunk = 0;                       // GFT_SCAN4_CHUNK: units per chunk of the streaming kernel (0: by batch size)
    uint32_t opt_solve_dbg = 0;                         // GFT_SOLVE_DEBUG (timing studies)
    int opt_solve_group = -1;                           // GFT_SOLVE_GROUP_DOCS: forced group width (-1: the widest that fits)
    // one caller at a time per handle: every entry point that touches the device state takes this
    mutable std::recursive_mutex mu;
    // multi-device handle (gft_engine_create_multi): this engine serves devices[0], `peers` the others.  Tables and
    // programs are replicated, a batch is cut into contiguous document ranges of near-equal text bytes, every device
    // has its own host thread and stream for the duration of a call
    std::vector<gft_engine*> peers;
    std::vector<uint64_t> shard_cut;                     // document cuts of the last multi-device gft_process
    bool in_multi = false;                               // set while a multi-device call runs this engine's own share
    std::vector<void*> comms;                            // RCCL communicators (ncclCommInitAll), one per device; empty: none
    bool rccl_self = false;                              // GFT_RCCL_SELF=1 over one device named several times: ONE communicator of one rank
    void* rccl_lib = nullptr;

    // programs
    bool have_programs = false;
    uint32_t n_exprs = 0, n_extra = 0;
    // what the HOST solves (host_solve.hpp): expressions beyond the device solver's limits, and INORD expressions in the
    // documents where one of their slots has a position list that is not ascending (a keyword and a regex with the same
    // literal: finder/finder.go:181-196).  Host copy of the public programs + what gft_set_programs learnt about them.
    std::vector<uint32_t> h_prog;
    std::vector<uint64_t> h_prog_off;
    std::vector<ProgramTraits> traits;
    std::vector<uint32_t> host_only;       // expressions that are always solved on the host (over a device limit)
    std::vector<uint32_t> inord_exprs;     // expressions with a multi-leaf INORD group (candidates for irregular documents)
    std::vector<uint8_t> inord_slot;       // [n_slots]: 1 = the slot is read inside such a group
    uint64_t last_n_units = 0, last_total = 0;   // of the last completed scan (csr_from_pool)
    bool csr_valid = false;                // d_match_off / d_term / d_pos hold the last scan's canonical CSR
    DevBuf d_patch;                        // bit patches of host-solved results for a device-resident bitmap
    DevBuf d_prog, d_prog_off;            // public postfix words (INORD group subtrees are read from these)
    DevBuf d_fprog, d_fprog_off, d_groups; // fused internal form + INORD group table
    DevBuf d_wide_slot, d_wide_theta;      // pairs of wide INORD groups, a region per wave of the solver's grid
    DevBuf d_solve_dbg;                    // GFT_SOLVE_DEBUG & 8: phase clocks
    // batches in a row that were one unit per document (k_units_single serves the next one from 2 on; a batch that took
    // that path and held a longer document after all sets it well below zero, so that a corpus whose batches alternate does
    // not pay for the miss every other time)
    int single_streak = 0;
    uint64_t last_static_slabs = 0;        // pool entries the waves of the last scan launch owned from the start (gft_scan2 / 3)
    bool deferred_single = false;          // ... and this one took that path
    uint32_t ctl_epoch = 1, deferred_epoch = 0;   // k_units_single batches are numbered from 2 (their control-block flags)
    uint64_t deferred_n_docs = 0;
    // gft_process_device_begin / _end: up to two batches enqueued, their read-backs landing in pinned slots of their own
    struct Pending {
        bool done = false;                 // completed inside begin (a batch that could not be deferred): rc is its status
        int rc = 0;
        const uint8_t* d_text = nullptr; const uint64_t* d_doc_off = nullptr; uint64_t n_docs = 0; uint32_t flags = 0;
        uint32_t* d_bitmap = nullptr;
        bool single = false; uint32_t epoch = 0; uint64_t n_docs_cap = 0, unit_cap = 0, static_slabs = 0;     // deferred_check's view of the launch
        uint64_t* rb = nullptr; hipEvent_t ev = nullptr;
    };
    Pending pend[2];
    unsigned pend_head = 0, pend_count = 0;
    DevBuf d_order, d_blk_class, d_wave_blk;            // evaluation order of the programs (gft_set_programs)
    uint32_t last_solve_group_docs = 64;   // documents per solver group of the last launch (0 = presence matrix in HBM)
    DevBuf d_fprog_t, d_fblk_off;          // fused programs per sorted block of 64, transposed (read when they do not fit LDS)
    uint32_t fprog_words = 0;
    uint32_t n_inord_groups = 0;           // fused INORD ops: 0 = the solver never reads positions
    uint32_t wide_pairs = 0;               // the widest INORD group the device solves through its scratch path (0: none); d_wide_*
    uint32_t n_wide = 0;                   // expressions with such a group: answered by the solver's second phase (d_wide_list)
    DevBuf d_wide_list;
    uint32_t n_rare_words = 0;             // fused NOT + INORD ops: 0 = the solver variant without their slow path
    DevBuf d_pscratch;                    // HBM presence matrices when n_slots * 8 B does not fit LDS

    // workspace
    DevBuf d_unit_cnt, d_unit_base, d_units, d_partial, d_pool_term, d_pool_pos, d_unit_start,
        d_unit_count, d_unit_out, d_term, d_pos, d_match_off;
    uint64_t pool_cap = 0;
    // staging for the host-buffer entry points
    DevBuf d_text, d_doc_off, d_bitmap, d_xoff, d_xslot, d_xpos;
    DevBuf d_uq_first, d_uq_cnt, d_uq_off, d_uq_term;   // GFT_SCAN_UNIQUE: per-workgroup first-occurrence rows, unique CSR
    DevBuf d_rn_cnt, d_rn_base, d_rn_starts, d_rn_prefix;   // GFT_POS_RUNES: blocks per document, their rune starts, prefix sums
    std::vector<uint64_t> h_match_off;
    std::vector<uint32_t> h_term, h_pos;

    // profiling
    int profiling = 0;                     // gft_profile_enable: 0 off, 1 every category, 2 the scan kernel only
    std::vector<hipEvent_t> prof_pool;     // events given back by gft_profile_reset
    std::map<std::string, ProfCat> prof;
};

namespace {

void refresh_options(gft_engine* e) {
    auto num = [](const char* name, long dflt) { const char* v = getenv(name); return v ? atol(v) : dflt; };
    e->opt_scan_dbg = (uint32_t)num("GFT_SCAN_DEBUG", 0);
    e->opt_scan_prio = num("GFT_SCAN_PRIO", 1) ? 1u : 0u;
    e->opt_scan_ordered = getenv("GFT_SCAN_ORDERED") ? 1u : 0u;
    e->opt_scan4_chunk = (uint32_t)num("GFT_SCAN4_CHUNK", 0);
    e->opt_scan4_round = (uint32_t)num("GFT_SCAN4_ROUND", 0);
    e->opt_scan5_groups = (uint32_t)num("GFT_SCAN5_GROUPS", 0);
    e->opt_scan5_large = num("GFT_SCAN5_LARGE", 1) ? 1u : 0u;
    e->opt_scan5_bloom_kb = (uint32_t)std::min<long>(std::max<long>(num("GFT_SCAN5_BLOOM_KB", 32), 0), 64);
    e->opt_scan5_fifo = (uint32_t)std::min<long>(std::max<long>(num("GFT_SCAN5_FIFO", 0), 0), 4096) & ~63u;
    e->opt_solve_dbg = (uint32_t)num("GFT_SOLVE_DEBUG", 0);
    e->opt_solve_group = (int)num("GFT_SOLVE_GROUP_DOCS", -1);
}
#define GFT_LOCK(e) std::lock_guard<std::recursive_mutex> _gft_lock((e)->mu)

int fail(const gft_engine* e, int code, const std::string& msg) {
    e->err = msg;
    return code;
}
int fail_hip(const gft_engine* e, hipError_t h, const char* what) {
    e->err = std::string(what) + ": " + hipGetErrorString(h);
    return GFT_E_HIP;
}

#define HIP_TRY(expr, what)                                   \
    do {                                                      \
        hipError_t _h = (expr);                               \
        if (_h != hipSuccess) return fail_hip(e, _h, what);   \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Entry points that hand host memory (the caller's buffers, or temporaries of their own) to asynchronous copies: whatever
// path they leave by -- an error in the middle included -- the stream has drained before that memory can go away.
struct SyncOnExit {
    gft_engine* e;
    explicit SyncOnExit(gft_engine* e_) : e(e_) {}
    ~SyncOnExit() { if (e->device >= 0 && e->stream) (void)hipStreamSynchronize(e->stream); }
};

struct ProfScope {
    gft_engine* e;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(gft_engine* e_, const char* cat) : e(e_) {
        if (!e->profiling || (e->profiling == 2 && std::strcmp(cat, "scan") != 0)) return;
        auto get = [&](hipEvent_t* ev) {
            if (!e->prof_pool.empty()) { *ev = e->prof_pool.back(); e->prof_pool.pop_back(); return true; }
            return hipEventCreate(ev) == hipSuccess;
        };
        if (!get(&a) || !get(&b)) { a = b = nullptr; return; }
        (void)hipEventRecord(a, e->stream);
        e->prof[cat].ev.emplace_back(a, b);
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, e->stream); }
};

template <class T>
int upload(gft_engine* e, DevBuf& buf, const std::vector<T>& v, const char* what) {
    HIP_TRY(buf.ensure(std::max<size_t>(v.size() * sizeof(T), 16)), what);
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, e->stream), what);
    return GFT_OK;
}

// validates one postfix program and measures its stack needs
// traits != nullptr: a program beyond the device solver's limits is not refused but marked (it is solved on the host),
// and the slots of its multi-leaf INORD groups are listed
int check_program(const gft_engine* e, const uint32_t* w, uint64_t len, uint32_t n_slots, uint32_t idx, ProgramTraits* traits = nullptr) {
    uint32_t sp = 0, psp = 0, g_tot = 0, g_psp = 0, max_sp = 0;   // g_*: the most pairs / the deepest pair stack of the group being read
    std::vector<uint32_t> group_slots;
    std::vector<uint32_t> pcnt;   // pair counts of the INORD operand stack
    bool in_group = false;
    auto bad = [&](const char* m) {
        return fail(e, GFT_E_INVALID, "program " + std::to_string(idx) + ": " + m);
    };
    for (uint64_t pc = 0; pc < len; pc++) {
        const uint32_t op = w[pc] >> 28;
        const bool fl = (w[pc] & GFT_INORD_FLAG) != 0;
        switch (op) {
        case GFT_OP_UNIT:
            if ((w[pc] & GFT_SLOT_MASK) >= n_slots) return bad("slot out of range");
            sp++;
            if (fl) { pcnt.push_back(1); in_group = true; group_slots.push_back(w[pc] & GFT_SLOT_MASK); }
            break;
        case GFT_OP_AND:
        case GFT_OP_OR:
            if (sp < 2) return bad("operand stack underflow");
            sp--;
            if (fl) {
                if (pcnt.size() < 2) return bad("INORD operand stack underflow");
                uint32_t r = pcnt.back(); pcnt.pop_back();
                if (op == GFT_OP_AND) pcnt.back() = r; else pcnt.back() += r;
            }
            break;
        case GFT_OP_NOT:
            if (sp < 1) return bad("operand stack underflow");
            if (in_group) return bad("NOT inside INORD");
            break;
        case GFT_OP_INORD:
            if (sp < 1 || pcnt.size() != 1) return bad("malformed INORD group");
            pcnt.clear(); in_group = false;
            if (traits && (g_tot > kMaxPairs || g_psp > kMaxPairDepth) && g_tot <= kMaxPairsWide && g_psp <= kMaxPairDepthWide) {
                traits->wide_pairs = std::max(traits->wide_pairs, g_tot);
                traits->wide_groups.push_back(pc);
            }
            g_tot = g_psp = 0;
            // (a group of ONE leaf is true exactly when the leaf is present: no position is ever compared)
            if (traits && group_slots.size() > 1) traits->inord_slots.insert(traits->inord_slots.end(), group_slots.begin(), group_slots.end());
            group_slots.clear();
            break;
        default:
            return bad("unknown opcode");
        }
        // (the depth of the PUBLIC postfix form binds only a caller without traits; gft_set_programs judges the depth of the
        // fused form, which is what the device interprets: operands are reordered there, a chain nested to one side is flat)
        max_sp = std::max(max_sp, sp);
        if (sp > kMaxBoolDepth && !traits)
            return fail(e, GFT_E_UNSUPPORTED, "program " + std::to_string(idx) + ": operand stack deeper than " +
                                                   std::to_string(kMaxBoolDepth));
        uint32_t tot = 0;
        for (uint32_t c : pcnt) tot += c;
        psp = (uint32_t)pcnt.size();
        g_tot = std::max(g_tot, tot); g_psp = std::max(g_psp, psp);
        if (tot > kMaxPairs || psp > kMaxPairDepth) {
            if (!traits) return fail(e, GFT_E_UNSUPPORTED, "program " + std::to_string(idx) + ": INORD group too wide for the device solver");
            // (more than a pair per lane: the device's scratch path up to kMaxPairsWide, the host beyond)
            if (tot > kMaxPairsWide || psp > kMaxPairDepthWide) traits->over_limit = true;
        }
    }
    if (sp != 1 || !pcnt.empty()) return bad("program does not reduce to one value");
    if (traits && traits->wide_pairs && max_sp > kMaxPairDepthWide) traits->over_limit = true;   // (the wide evaluator's boolean stack: a bit per entry)
    if (traits) {
        std::sort(traits->inord_slots.begin(), traits->inord_slots.end());
        traits->inord_slots.erase(std::unique(traits->inord_slots.begin(), traits->inord_slots.end()), traits->inord_slots.end());
    }
    return GFT_OK;
}

// public postfix words -> fused internal form (gft_kernels.hpp FusedOp) + INORD group table.
// `gbase` = offset of this program inside the uploaded public word array.
// public postfix words -> fused words (gft_kernels.hpp FusedOp); returns the deepest the accumulator stack gets
uint32_t fuse_program(const uint32_t* w, uint64_t len, uint64_t gbase, std::vector<uint32_t>& out,
                      std::vector<uint32_t>& groups) {
    // postfix -> tree (node = operator or leaf, with the range of public words it covers)
    struct Node { uint32_t op, slot; int64_t l, r; uint64_t s, e; };
    std::vector<Node> nodes;
    std::vector<int64_t> st;
    for (uint64_t i = 0; i < len; i++) {
        const uint32_t op = w[i] >> 28;
        switch (op) {
        case GFT_OP_UNIT:
            nodes.push_back(Node{op, w[i] & GFT_SLOT_MASK, -1, -1, i, i});
            st.push_back((int64_t)nodes.size() - 1);
            break;
        case GFT_OP_AND:
        case GFT_OP_OR: {
            const int64_t r = st.back(); st.pop_back();
            const int64_t l = st.back(); st.pop_back();
            nodes.push_back(Node{op, 0, l, r, nodes[l].s, i});
            st.push_back((int64_t)nodes.size() - 1);
            break;
        }
        case GFT_OP_NOT:
        case GFT_OP_INORD: {
            const int64_t c = st.back(); st.pop_back();
            nodes.push_back(Node{op, 0, c, -1, nodes[c].s, i});
            st.push_back((int64_t)nodes.size() - 1);
            break;
        }
        default:
            break;
        }
    }
    if (st.empty()) return 0;
    const size_t out0 = out.size();
    // Code generation with an explicit job stack (left-deep chains of 10 000 leaves must not recurse).
    //  * NOT is pushed down to the leaves (De Morgan; every node is evaluated anyway, the reference does not
    //    short-circuit), so it only survives on top of an INORD group;
    //  * AND / OR commute: the operand that is a leaf goes second and folds into the operator word;
    //  * the accumulator is pushed only between two operands that are both subtrees -- and of those the one that needs
    //    the deeper stack goes FIRST (Sethi-Ullman), so a chain of parentheses nested to the right stays one entry deep
    //    and only a balanced tree of 2^k subtrees gets k deep: real rule sets fit the interpreter's register stack.
    auto strip = [&](int64_t n, bool& neg) {        // skip NOT chains
        while (nodes[n].op == GFT_OP_NOT) { neg = !neg; n = nodes[n].l; }
        return n;
    };
    std::vector<uint32_t> need(nodes.size(), 0);    // stack entries the subtree's code needs (children come before parents)
    for (size_t n = 0; n < nodes.size(); n++) {
        const Node& nd = nodes[n];
        bool dummy = false;
        if (nd.op == GFT_OP_NOT || nd.op == GFT_OP_INORD) need[n] = need[nd.l];
        else if (nd.op == GFT_OP_AND || nd.op == GFT_OP_OR) {
            const int64_t l = strip(nd.l, dummy), r = strip(nd.r, dummy);
            if (nodes[r].op == GFT_OP_UNIT) need[n] = need[nd.l];
            else if (nodes[l].op == GFT_OP_UNIT) need[n] = need[nd.r];
            else need[n] = need[nd.l] == need[nd.r] ? need[nd.l] + 1 : std::max(need[nd.l], need[nd.r]);
        }
    }
    struct Job { int64_t n; int phase; bool neg; };
    std::vector<Job> jobs{{st.back(), 0, false}};
    uint32_t depth = 0, max_depth = 0;
    while (!jobs.empty()) {
        Job j = jobs.back(); jobs.pop_back();
        bool neg = j.neg;
        const int64_t n = j.phase == 0 ? strip(j.n, neg) : j.n;
        const Node& nd = nodes[n];
        switch (nd.op) {
        case GFT_OP_UNIT:
            out.push_back((neg ? kFopSetN : kFopSet) << 28 | nd.slot);
            break;
        case GFT_OP_INORD:
            if (j.phase == 0) { jobs.push_back({n, 1, neg}); jobs.push_back({nd.l, 0, false}); }
            else {
                // (a group with a single leaf has a non-empty position list exactly when the leaf is present: every
                // reported key carries >= 1 position, so no position check is needed)
                if (nodes[nd.l].op != GFT_OP_UNIT) {
                    out.push_back(kFopInord << 28 | (uint32_t)(groups.size() / 2));
                    groups.push_back((uint32_t)(gbase + nodes[nd.l].s));
                    groups.push_back((uint32_t)(nodes[nd.l].e - nodes[nd.l].s + 1));
                }
                if (neg) out.push_back(kFopNot << 28);
            }
            break;
        case GFT_OP_AND:
        case GFT_OP_OR: {
            const bool is_and = (nd.op == GFT_OP_AND) != neg;        // not (a and b) == not a or not b
            if (j.phase == 0) {
                bool ln = neg, rn = neg;
                const int64_t l = strip(nd.l, ln), r = strip(nd.r, rn);
                if (nodes[r].op == GFT_OP_UNIT) { jobs.push_back({n, 1, neg}); jobs.push_back({nd.l, 0, neg}); }
                else if (nodes[l].op == GFT_OP_UNIT) { jobs.push_back({n, 2, neg}); jobs.push_back({nd.r, 0, neg}); }
                else {
                    const bool left_first = need[nd.l] >= need[nd.r];
                    jobs.push_back({n, 4, neg}); jobs.push_back({left_first ? nd.r : nd.l, 0, neg});
                    jobs.push_back({n, 3, neg}); jobs.push_back({left_first ? nd.l : nd.r, 0, neg});
                }
            } else if (j.phase == 1 || j.phase == 2) {
                bool ln = neg;
                const int64_t leaf = strip(j.phase == 1 ? nd.r : nd.l, ln);
                out.push_back((is_and ? (ln ? kFopAndNS : kFopAndS) : (ln ? kFopOrNS : kFopOrS)) << 28 | nodes[leaf].slot);
            } else if (j.phase == 3) {
                out.push_back(kFopPush << 28);
                max_depth = std::max(max_depth, ++depth);
            } else {
                out.push_back((is_and ? kFopAndPop : kFopOrPop) << 28);
                depth--;
            }
            break;
        }
        default:
            break;
        }
    }
    // a push is always followed by the first leaf of the next subtree: one word does both
    size_t k = out0;
    for (size_t i = out0; i < out.size(); i++) {
        const uint32_t op = out[i] >> 28, nx = i + 1 < out.size() ? out[i + 1] >> 28 : 0u;
        if (op == kFopPush && (nx == kFopSet || nx == kFopSetN)) {
            out[k++] = (nx == kFopSet ? kFopPushSet : kFopPushSetN) << 28 | (out[i + 1] & 0x0FFFFFFFu);
            i++;
        } else out[k++] = out[i];
    }
    out.resize(k);
    return max_depth;
}

int ensure_pool(gft_engine* e, uint64_t entries) {
    if (entries <= e->pool_cap) return GFT_OK;
    HIP_TRY(e->d_pool_term.ensure(entries * 4), "pool alloc");
    HIP_TRY(e->d_pool_pos.ensure(entries * 4), "pool alloc");
    e->pool_cap = std::min(e->d_pool_term.cap, e->d_pool_pos.cap) / 4;
    return GFT_OK;
}

// slabs of the last completed scan -> canonical CSR in e->d_match_off / d_term / d_pos (document order, the reference's
// emission order inside a document).  The unit table, the slabs and the counts of that scan are still in the engine
// (last_n_units, last_total); positions must have been written (want_pos).
int csr_from_pool(gft_engine* e, uint64_t n_docs) {
    hipStream_t st = e->stream;
    const uint64_t n_units = e->last_n_units, total = e->last_total;
    HIP_TRY(e->d_term.ensure(std::max<uint64_t>(total, 1) * 4), "result alloc");
    HIP_TRY(e->d_pos.ensure(std::max<uint64_t>(total, 1) * 4), "result alloc");
    HIP_TRY(e->d_unit_out.ensure((n_units + 1) * 8), "unit alloc");
    HIP_TRY(e->d_partial.ensure(scan_partials_needed(std::max(n_units, n_docs)) * 8), "unit alloc");
    ProfScope ps(e, "aux");
    HIP_TRY(launch_exclusive_scan(e->d_unit_count.as<uint32_t>(), n_units, e->d_unit_out.as<uint64_t>(),
                                  e->d_partial.as<uint64_t>(), st), "unit_out scan");
    // (the suffix-window kernels leave a unit's matches in any order -- shifted anchors report a term from another position
    // than its end, also on scan2's per-lane path: the gather sorts them)
    const bool sort_units = e->use_scan2 || e->use_scan3;
    HIP_TRY(launch_gather(e->d_unit_start.as<uint64_t>(), e->d_unit_count.as<uint32_t>(),
                          e->d_unit_out.as<uint64_t>(), n_units, e->d_pool_term.as<uint32_t>(),
                          e->d_pool_pos.as<uint32_t>(), e->d_term.as<uint32_t>(), e->d_pos.as<uint32_t>(),
                          e->d_unit_base.as<uint64_t>(), n_docs, e->d_match_off.as<uint64_t>(), e->n_cus, st,
                          sort_units ? e->d_units.as<Unit>() : nullptr,
                          e->d_term_len.as<uint32_t>(), (e->build_flags & GFT_POS_END) ? 1u : 0u),
            "gather");
    e->csr_valid = true;
    return GFT_OK;
}

// The device pipeline shared by scan and process.  On success the canonical CSR sits in e->d_match_off /
// d_term / d_pos and *n_matches is set.
constexpr uint64_t kHostUnitDocs = 1024;   // batches up to this many documents get their unit table from the host

// defer_ok: the caller reads the control block back itself after its last kernel (deferred_check) -- the unit table and
// the match pool are then sized from the previous batch, and a batch that outgrew them is run again.
int scan_pipeline(gft_engine* e, const uint8_t* d_text, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t flags,
                  bool need_csr, uint64_t* n_matches, const uint64_t* h_doc_off = nullptr, bool defer_ok = false) {
    hipStream_t st = e->stream;
    *n_matches = 0;
    e->deferred = false;
    e->deferred_single = false;
    e->scan_valid_docs = ~0ull;           // the pool is about to be overwritten
    e->csr_valid = false;
    HIP_TRY(e->d_match_off.ensure((n_docs + 1) * 8), "match_off alloc");
    if (n_docs == 0) {
        HIP_TRY(hipMemsetAsync(e->d_match_off.p, 0, 8, st), "memset");
        HIP_TRY(hipStreamSynchronize(st), "sync");
        e->last_n_units = e->last_total = 0;
        // (nothing was scanned: the verdict and the text range of the batch BEFORE must not be judged against this one's
        // pointers -- refine_nonascii would launch k_fold_safe over a stale range of a text that may be NULL)
        e->last_nonascii = false;
        e->last_nonascii_bits = 0;
        e->last_text_lo = e->last_text_hi = 0;
        return GFT_OK;
    }
    const uint32_t warm = e->tab.max_term_len ? e->tab.max_term_len - 1 : 0;
    // gft_scan2: a unit's matches should fit the wave's LDS fifo (kScan2FifoCap), so the unit size follows the match
    // density the previous call saw (dense dictionaries -> smaller units); results do not depend on it
    const uint32_t unit_max = e->use_scan3 ? kScan3UnitMax : e->use_scan4 ? kScan4UnitMax : e->use_scan2 ? e->scan2_unit_max : kTextBuf - warm;

    // 1. work units
    HIP_TRY(e->d_ctl.ensure(64), "control alloc");
    HIP_TRY(e->d_unit_cnt.ensure(n_docs * 4), "unit alloc");
    HIP_TRY(e->d_unit_base.ensure((n_docs + 1) * 8), "unit alloc");
    HIP_TRY(e->d_partial.ensure(scan_partials_needed(n_docs) * 8), "unit alloc");
    uint64_t n_units = 0, text_lo = 0, text_hi = 0;
    // small batches from host memory (a single ProcessText / FindSubstrings call is the reference's own shape): the unit
    // table is a few entries, computed here and uploaded instead of five kernel launches and a synchronising read-back
    const bool host_units = h_doc_off != nullptr && n_docs <= kHostUnitDocs;
    // The batch before was one unit per document: this one gets its unit table from ONE launch on that assumption
    // (k_units_single) instead of count + prefix sum + fill + clamp; deferred_check learns whether it held.  That launch
    // also clears the control block (its two flags are raised to the batch's EPOCH, a number no earlier batch wrote there,
    // so they need no clearing): one node less on the stream of every batch
    bool units_single = !host_units && defer_ok && !need_csr && e->single_streak >= 2 && e->pool_cap > 0 &&
                        std::min(std::min(e->d_units.cap / sizeof(Unit), e->d_unit_start.cap / 8), e->d_unit_count.cap / 4) >= n_docs;
    if (units_single && ++e->ctl_epoch < 2) { e->ctl_epoch = 1; units_single = false; }      // (wrapped: this batch the general way)
    if (!units_single) HIP_TRY(hipMemsetAsync(e->d_ctl.p, 0, 32, st), "memset");     // flag, cursor, match count, non-ASCII flag
    std::vector<uint64_t> hub;
    std::vector<Unit> hun;
    struct DrainIf { gft_engine* e; bool on; ~DrainIf() { if (on && e->stream) (void)hipStreamSynchronize(e->stream); } } drain_units{e, host_units};
    if (host_units) {
        hub.assign(n_docs + 1, 0);
        for (uint64_t d = 0; d < n_docs; d++) {
            if (h_doc_off[d + 1] < h_doc_off[d]) return fail(e, GFT_E_INVALID, "doc_off is not ascending");
            const uint64_t n = h_doc_off[d + 1] - h_doc_off[d];
            if (n > 0xFFFFFFFFull) return fail(e, GFT_E_UNSUPPORTED, "a document is longer than 4 GiB - 1 bytes (positions are 32-bit)");
            const uint64_t k = n <= unit_max ? 1 : (n + unit_max - 1) / unit_max;
            hub[d + 1] = hub[d] + k;
            const uint64_t per = (n + k - 1) / k;                 // as k_unit_fill
            for (uint64_t i = 0; i < k; i++) {
                uint64_t lo = i * per, hi = lo + per < n ? lo + per : n;
                if (lo > n) lo = n;
                hun.push_back(Unit{(uint32_t)d, (uint32_t)lo, (uint32_t)hi});
            }
        }
        n_units = hub[n_docs]; text_lo = h_doc_off[0]; text_hi = h_doc_off[n_docs];
        HIP_TRY(hipMemcpyAsync(e->d_unit_base.p, hub.data(), (n_docs + 1) * 8, hipMemcpyHostToDevice, st), "unit upload");
    } else if (units_single) {
        ProfScope ps(e, "aux");
        HIP_TRY(launch_units_single(d_doc_off, n_docs, unit_max, e->d_units.as<Unit>(), e->d_unit_base.as<uint64_t>(),
                                    e->d_ctl.as<uint32_t>(), e->ctl_epoch, st), "unit table");
        n_units = n_docs; text_lo = 0; text_hi = ~0ull;
        e->deferred = true; e->deferred_single = true; e->deferred_epoch = e->ctl_epoch;
        e->deferred_unit_cap = n_docs; e->deferred_n_docs = n_docs;
    } else {
        {
            ProfScope ps(e, "aux");
            HIP_TRY(launch_unit_count(d_doc_off, n_docs, unit_max, e->d_unit_cnt.as<uint32_t>(), e->d_ctl.as<uint32_t>(), st), "unit_count");
            HIP_TRY(launch_exclusive_scan(e->d_unit_cnt.as<uint32_t>(), n_docs, e->d_unit_base.as<uint64_t>(),
                                          e->d_partial.as<uint64_t>(), st), "unit scan");
            HIP_TRY(launch_pack_ctl(e->d_unit_base.as<uint64_t>(), d_doc_off, n_docs, e->d_ctl.as<uint64_t>() + 4, st), "unit scan");
        }
        // No read-back when the caller checks afterwards: the tables keep the size the last batch gave them (a document
        // is one unit unless it is longer than unit_max), units beyond the table are dropped and every index is clamped
        // into it -- deferred_check sees the true count and has the batch run again
        const uint64_t cap_units = std::min(std::min(e->d_units.cap / sizeof(Unit), e->d_unit_start.cap / 8), e->d_unit_count.cap / 4);
        e->deferred = defer_ok && !need_csr && cap_units >= n_docs && e->pool_cap > 0;
        if (e->deferred) {
            n_units = cap_units; text_lo = 0; text_hi = ~0ull;       // (the text blob is readable 64 bytes past its end: gft.h)
            e->deferred_unit_cap = cap_units; e->deferred_n_docs = n_docs;
        } else {
            uint64_t rb[7] = {0, 0, 0, 0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(rb, e->d_ctl.p, sizeof rb, hipMemcpyDeviceToHost, st), "readback");
            HIP_TRY(hipStreamSynchronize(st), "sync");
            n_units = rb[4]; text_lo = rb[5]; text_hi = rb[6];
            e->single_streak = n_units == n_docs ? e->single_streak + 1 : std::min(e->single_streak, 0);
            const uint32_t bad_doc = (uint32_t)rb[0];
            if (text_hi < text_lo) return fail(e, GFT_E_INVALID, "doc_off is not ascending");
            if (bad_doc) return fail(e, GFT_E_INVALID, "doc_off is not ascending, or a document is longer than 4 GiB - 1 bytes (positions are 32-bit)");
        }
    }

    if (!e->deferred) { e->last_text_lo = text_lo; e->last_text_hi = text_hi; }
    HIP_TRY(e->d_units.ensure(n_units * sizeof(Unit)), "unit alloc");
    HIP_TRY(e->d_unit_start.ensure(n_units * 8), "unit alloc");
    HIP_TRY(e->d_unit_count.ensure(n_units * 4), "unit alloc");
    HIP_TRY(e->d_unit_out.ensure((n_units + 1) * 8), "unit alloc");
    HIP_TRY(e->d_partial.ensure(scan_partials_needed(std::max(n_units, n_docs)) * 8), "unit alloc");
    {
        ProfScope ps(e, "aux");
        if (host_units) {
            if (n_units) HIP_TRY(hipMemcpyAsync(e->d_units.p, hun.data(), n_units * sizeof(Unit), hipMemcpyHostToDevice, st), "unit upload");
        } else if (e->deferred_single) {
            // (k_units_single has filled the table)
        } else if (e->deferred) {
            HIP_TRY(hipMemsetAsync(e->d_units.p, 0, n_units * sizeof(Unit), st), "memset");      // empty units behind the real ones
            HIP_TRY(launch_unit_fill(d_doc_off, n_docs, e->d_unit_base.as<uint64_t>(), e->d_units.as<Unit>(), unit_max, st, n_units), "unit_fill");
            HIP_TRY(launch_clamp_u64(e->d_unit_base.as<uint64_t>(), n_docs + 1, n_units, st), "unit clamp");
        } else {
            HIP_TRY(launch_unit_fill(d_doc_off, n_docs, e->d_unit_base.as<uint64_t>(), e->d_units.as<Unit>(), unit_max, st), "unit_fill");
        }
    }

    // 2. automaton walk into the slab pool; grow the pool and re-run if it overflowed (never truncate)
    int rc = e->deferred ? GFT_OK : ensure_pool(e, std::max<uint64_t>(1u << 20, (text_hi - text_lo) / 16));
    if (!rc && !e->deferred && (e->use_scan2 || e->use_scan3)) {
        // (every wave of the grid owns a slab from the start: the pool holds those twice over, or a small batch on a fresh
        // engine would overflow it before it had written a match)
        // (scan4: a slab holds at least one chunk's regions -- up to eight units of unit_max bytes at 1.6 x the density seen)
        const uint64_t wpw = e->use_scan3 ? e->scan3_waves : e->use_scan4 ? e->scan4_waves : e->use_scan5 ? kScan5Waves : e->scan2_k2_waves;
        const uint64_t min_slab = e->use_scan3 ? 2 * kScan3MinRoom
                                  : e->use_scan4 ? kScan4ChunkUnits * ((uint64_t)(unit_max * e->scan4_density * 1.6) + 49) : 64;
        const uint64_t waves = std::min<uint64_t>(std::max<uint64_t>((n_units + wpw - 1) / wpw, 1), e->n_cus) * wpw;
        rc = ensure_pool(e, 2 * waves * min_slab);
    }
    if (rc) return rc;
    uint64_t total = 0;
    for (int attempt = 0; attempt < 3 && e->use_scan3; attempt++) {
        if (attempt) HIP_TRY(hipMemsetAsync(e->d_ctl.as<uint8_t>() + 8, 0, 16, st), "memset");
        Scan3Params P;
        P.text = d_text; P.doc_off = d_doc_off; P.units = e->d_units.as<Unit>(); P.n_units = n_units; P.text_bytes = text_hi;
        P.fold = (flags & GFT_FOLD_ASCII) ? 1 : 0;
        P.cls = P.fold ? e->d_s3_cls_fold.as<uint8_t>() : e->d_s3_cls.as<uint8_t>();
        P.filter = e->d_s3_filter.as<uint32_t>(); P.filter_words = (uint32_t)e->s3.filter.size();
        P.short3 = e->d_s3_short3.as<uint8_t>(); P.short3_bytes = (uint32_t)e->s3.short3.size();
        P.srec = e->d_s3_srec.as<uint32_t>(); P.srec_words = (uint32_t)e->s3.srec.size();
        P.short3_big = e->s3.short3_big.empty() ? nullptr : e->d_s3_short3_big.as<uint32_t>();
        P.srec_big = e->d_s3_srec_big.as<uint32_t>();
        P.bloom = e->d_s3_bloom.as<uint32_t>(); P.bloom_lg = e->s3.bloom_lg; P.bloom_lds = e->s3.bloom_lg <= kScan3BloomLdsLg ? 1 : 0;
        P.slots = e->d_s3_slots.as<Scan2Slot>(); P.slot_shift = e->s3.slot_shift; P.slot_seed = e->s3.slot_seed;
        P.more = e->d_s3_more.as<Scan2Slot>();
        P.term_blob = e->d_s3_term_blob.as<uint8_t>(); P.term_off = e->d_s3_term_off.as<uint32_t>();
        P.G = e->s3.G; P.grouped = e->s3.grouped ? 1 : 0;
        P.pos_end = (e->build_flags & GFT_POS_END) ? 1 : 0;
        // presence-only mode: positions are only read by INORD groups (and by CSR callers)
        P.want_pos = (need_csr || e->n_inord_groups > 0) ? 1 : 0;
        P.prio = e->opt_scan_prio;
        P.dbg = e->opt_scan_dbg;
        P.nonascii = e->d_ctl.as<uint32_t>() + 6;
        P.cand_cap = e->scan3_cand_cap;
        P.cursor = e->d_ctl.as<uint64_t>() + 1; P.pool_cap = e->pool_cap;
        P.pool_term = e->d_pool_term.as<uint32_t>(); P.pool_pos = e->d_pool_pos.as<uint32_t>();
        P.unit_start = e->d_unit_start.as<uint64_t>(); P.unit_count = e->d_unit_count.as<uint32_t>();
        P.n_matches = e->d_ctl.as<uint64_t>() + 2;
        // slab slack is at most one slab per resident wave: keep it below half the pool
        const uint64_t n_waves = (uint64_t)e->n_cus * e->scan3_waves;
        P.slab = (uint32_t)std::min<uint64_t>(kScan2Slab, std::max<uint64_t>(2 * kScan3MinRoom, e->pool_cap / (2 * n_waves)));
        // every wave of the grid owns one slab from the start; the cursor counts what is taken behind those
        e->last_static_slabs = std::min<uint64_t>(std::max<uint64_t>((n_units + e->scan3_waves - 1) / e->scan3_waves, 1), e->n_cus) * e->scan3_waves * P.slab;
        {
            ProfScope ps(e, "scan");
            HIP_TRY(launch_scan3(P, e->scan3_waves, e->n_cus, st), "scan kernel launch");
        }
        if (e->deferred) return GFT_OK;                       // (deferred_check reads the cursor after the solver)
        uint64_t ct[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyAsync(ct, e->d_ctl.as<uint8_t>() + 8, 24, hipMemcpyDeviceToHost, st), "readback");
        HIP_TRY(hipStreamSynchronize(st), "scan kernel");
        const uint64_t cursor = ct[0] + e->last_static_slabs;
        total = ct[1];
        e->last_nonascii_bits = (uint32_t)ct[2]; e->last_nonascii = e->last_nonascii_bits != 0;
        if (cursor <= e->pool_cap) break;
        if (attempt == 2) return fail(e, GFT_E_HIP, "match pool overflow persisted");
        rc = ensure_pool(e, cursor + cursor / 16);
        if (rc) return rc;
    }
    for (int attempt = 0; attempt < 3 && e->use_scan2 && !e->use_scan3; attempt++) {
        if (attempt) HIP_TRY(hipMemsetAsync(e->d_ctl.as<uint8_t>() + 8, 0, 16, st), "memset");
        Scan2Params P;
        P.text = d_text; P.doc_off = d_doc_off; P.units = e->d_units.as<Unit>(); P.n_units = n_units;
        P.filter = e->d_s2_filter.as<uint32_t>(); P.filter_words = (uint32_t)e->s2.filter.size();
        P.hashed = e->s2.hashed ? 1 : 0; P.hash_shift = e->s2.hash_shift;
        P.short3 = e->d_s2_short3.as<uint8_t>(); P.short3_bytes = e->scan2_short3_bytes;
        P.fpt = e->d_s2_fpt.as<uint8_t>(); P.fpt_lg = e->s2.fpt_lg;
        P.shorts_packed = e->d_s2_shorts_packed.as<uint32_t>(); P.shorts_words = (uint32_t)std::min<size_t>(e->s2.shorts_packed.size(), 255 * 3);
        P.short3_big = e->s2.short3_big.empty() ? nullptr : e->d_s2_short3_big.as<uint32_t>();
        P.cand_cap = e->scan2_cand_cap;
        P.slots = e->d_s2_slots.as<Scan2Slot>(); P.slot_shift = e->s2.slot_shift; P.slot_seed = e->s2.slot_seed;
        P.more = e->d_s2_more.as<Scan2Slot>();
        P.text_bytes = text_hi;
        P.fold = (flags & GFT_FOLD_ASCII) ? 1 : 0;
        P.cls = P.fold ? e->d_s2_cls_fold.as<uint8_t>() : e->d_s2_cls.as<uint8_t>();
        P.term_blob = e->d_s2_term_blob.as<uint8_t>(); P.term_off = e->d_s2_term_off.as<uint32_t>();
        P.kp = e->s2.kp; P.pad_class = e->s2.pad_class;
        P.pos_end = (e->build_flags & GFT_POS_END) ? 1 : 0;
        P.cursor = e->d_ctl.as<uint64_t>() + 1; P.pool_cap = e->pool_cap;
        P.pool_term = e->d_pool_term.as<uint32_t>(); P.pool_pos = e->d_pool_pos.as<uint32_t>();
        P.unit_start = e->d_unit_start.as<uint64_t>(); P.unit_count = e->d_unit_count.as<uint32_t>();
        P.n_matches = e->d_ctl.as<uint64_t>() + 2;
        // slab slack is at most one slab per resident wave: keep it below half the pool
        const uint64_t n_waves = (uint64_t)e->n_cus * e->scan2_k2_waves;
        P.slab = (uint32_t)std::min<uint64_t>(kScan2Slab, std::max<uint64_t>(64, e->pool_cap / (2 * n_waves)));
        // every wave of the grid owns one slab from the start; the cursor counts what is taken behind those
        e->last_static_slabs = std::min<uint64_t>(std::max<uint64_t>((n_units + e->scan2_k2_waves - 1) / e->scan2_k2_waves, 1), e->n_cus) * e->scan2_k2_waves * P.slab;
        // the balanced path serves both callers: the solver reads presence / successor positions in any order, and
        // CSR results are put into emission order by the gather (k_gather_sorted).  GFT_SCAN_ORDERED=1 sends every unit
        // through the kernel's per-lane staging path (normally the fallback for units whose matches overflow the LDS
        // fifo): a second implementation of the verification, kept as a cross-check
        P.ordered = (need_csr && e->opt_scan_ordered) ? 1 : 0;
        // presence-only mode: positions are only read by INORD groups (and by CSR callers)
        P.want_pos = (need_csr || e->n_inord_groups > 0) ? 1 : 0;
        P.dbg = e->opt_scan_dbg;
        P.nonascii = e->d_ctl.as<uint32_t>() + 6;
        // wave priorities: the latency-bound verification stages overtake the filter phase of the other waves (5 % on
        // the benchmark; GFT_SCAN_PRIO=0 switches it off)
        P.prio = e->opt_scan_prio;
        P.dbg_counters = nullptr;
        if (P.dbg & (2 | 64)) {
            HIP_TRY(e->d_dbg.ensure(128), "debug alloc");
            HIP_TRY(hipMemsetAsync(e->d_dbg.p, 0, 128, st), "memset");
            P.dbg_counters = e->d_dbg.as<uint64_t>();
        }
        if (e->use_scan4) {
            // the streaming form: chunks of up to eight units (fewer when the batch is small: every wave should get several
            // chunks), a fifo in place of the candidate list, per-unit regions sized from the density seen so far
            const uint64_t n_waves4 = (uint64_t)e->n_cus * e->scan4_waves;
            P.chunk_units = (uint32_t)std::min<uint64_t>(kScan4ChunkUnits, std::max<uint64_t>(1, n_docs / (n_waves4 * 4)));
            if (e->opt_scan4_chunk) P.chunk_units = std::min<uint32_t>(e->opt_scan4_chunk, kScan4ChunkUnits);      // (GFT_SCAN4_CHUNK: tests)
            P.cand_cap = e->scan4_fifo[P.want_pos ? 1 : 0];
            P.bound_q16 = (uint32_t)std::min<double>(e->scan4_density * 1.6 * 65536.0 + 1.0, 4.0e9);
            P.bound_add = 48;
            P.round_c = e->opt_scan4_round ? std::min<uint32_t>(64, std::max<uint32_t>(16, e->opt_scan4_round & ~15u)) : 64;   // (GFT_SCAN4_ROUND: timing studies)
            // a slab should hold a few chunks' regions (the rest of a slab that the next chunk does not fit is lost)
            const uint64_t chunk_need = (uint64_t)P.chunk_units * (((uint64_t)unit_max * P.bound_q16 >> 16) + P.bound_add);
            P.slab = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(kScan2Slab, 4 * chunk_need), std::max<uint64_t>(chunk_need, e->pool_cap / (2 * n_waves4)));
            const uint64_t n_chunks = (n_units + P.chunk_units - 1) / P.chunk_units;
            e->last_static_slabs = std::min<uint64_t>(std::max<uint64_t>((n_chunks + e->scan4_waves - 1) / e->scan4_waves, 1), e->n_cus) * e->scan4_waves * P.slab;
            ProfScope ps(e, "scan");
            HIP_TRY(launch_scan4(P, e->scan4_waves, e->n_cus, st), "scan kernel launch");
        } else if (e->use_scan5) {
            // one filter probe per two bytes: scan2's tables behind the 3-gram filter over merged classes
            P.s5_filter = e->d_s5_filter.as<uint64_t>(); P.s5_dual = (uint32_t)e->s5.filter.size();
            P.s5_grp = P.fold ? e->d_s5_grp_fold.as<uint8_t>() : e->d_s5_grp.as<uint8_t>();
            P.s5_G = e->s5.G; P.s5_pad_g = e->s5.pad_group;
            P.s5_fifo_cap = e->s5plan.fifo_cap; P.cand_cap = e->s5plan.cand_cap;
            P.s5_sG = 0; P.s5_sgrp = nullptr; P.s5_srec_big = nullptr;
            if (e->s5_short_groups) {
                // more than 32 byte classes: the short terms through the group-indexed tables of the stride-2 kernel's set
                P.short3 = e->d_s3_short3.as<uint8_t>(); P.short3_bytes = (uint32_t)e->s3.short3.size();
                P.shorts_packed = e->d_s3_srec.as<uint32_t>(); P.shorts_words = (uint32_t)e->s3.srec.size();
                P.short3_big = e->s3.short3_big.empty() ? nullptr : e->d_s3_short3_big.as<uint32_t>();
                P.s5_srec_big = e->d_s3_srec_big.as<uint32_t>();
                P.s5_sgrp = P.fold ? e->d_s3_cls_fold.as<uint8_t>() : e->d_s3_cls.as<uint8_t>();
                P.s5_sG = e->s3.G;
            }
            P.s5_term_bits = e->s5_term_bits; P.s5_pos_bias = e->s5_pos_bias;
            P.s5_bloom = e->s5_bloom_lg ? e->d_s5_bloom.as<uint32_t>() : nullptr; P.s5_bloom_lg = e->s5_bloom_lg;
            const uint64_t n_waves5 = (uint64_t)e->n_cus * kScan5Waves;
            P.slab = (uint32_t)std::min<uint64_t>(kScan2Slab, std::max<uint64_t>(64, e->pool_cap / (2 * n_waves5)));
            e->last_static_slabs = std::min<uint64_t>(std::max<uint64_t>((n_units + kScan5Waves - 1) / kScan5Waves, 1), e->n_cus) * kScan5Waves * P.slab;
            ProfScope ps(e, "scan");
            HIP_TRY(launch_scan5(P, e->n_cus, st), "scan kernel launch");
        } else {
            ProfScope ps(e, "scan");
            // gft_scan2 serves both paths (ordered for CSR results, unordered + balanced for the solver)
            HIP_TRY(launch_scan2(P, e->scan2_k2_waves, e->n_cus, st), "scan kernel launch");
        }
        if (P.dbg & 64) {
            // phase clocks: a wave's cycles per unit (0 first bytes, 1 filter, 2 candidate list, 3 stage A, 4 stage B, 5 flush,
            // 7 unit record), averaged over all units
            uint64_t t[16];
            HIP_TRY(hipMemcpyAsync(t, e->d_dbg.p, sizeof t, hipMemcpyDeviceToHost, st), "debug read-back");
            HIP_TRY(hipStreamSynchronize(st), "debug read-back");
            if (e->use_scan5) fprintf(stderr, "[gft scan debug] scan5 (G=%u, list %u):\n", e->s5.G, e->s5plan.cand_cap);
            if (e->use_scan4)
                fprintf(stderr, "[gft scan debug] scan4 wave cycles per unit: chunk set-up %.0f, filter %.0f, queue push %.0f, stage A issue %.0f, stage A %.0f, stage B %.0f, flush %.0f, unit records %.0f\n",
                        (double)t[4] / n_units, (double)t[5] / n_units, (double)t[6] / n_units, (double)t[7] / n_units, (double)t[8] / n_units,
                        (double)t[9] / n_units, (double)t[10] / n_units, (double)t[11] / n_units);
            else
            fprintf(stderr, "[gft scan debug] wave cycles per unit: first bytes %.0f, filter %.0f, list %.0f, stage A %.0f (scan5: trips %.0f + stage-B issue and short-term trips %.0f), stage B %.0f, flush %.0f, unit record %.0f\n",
                    (double)t[4] / n_units, (double)t[5] / n_units, (double)t[6] / n_units, (double)(t[7] + t[10]) / n_units, (double)t[10] / n_units, (double)t[7] / n_units,
                    (double)t[8] / n_units, (double)t[9] / n_units, (double)t[11] / n_units);
            double sum = 0;
            for (int k = 4; k < 12; k++) sum += (double)t[k];
            if (t[13]) fprintf(stderr, "[gft scan debug] %llu waves: mean %.0f cycles in all, the slowest %.0f (+%.1f %%)\n", (unsigned long long)t[13],
                               sum / (double)t[13], (double)t[12], 100.0 * ((double)t[12] * (double)t[13] / sum - 1.0));
        }
        if (e->deferred) return GFT_OK;                       // (deferred_check reads the cursor after the solver)
        uint64_t ct[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyAsync(ct, e->d_ctl.as<uint8_t>() + 8, 24, hipMemcpyDeviceToHost, st), "readback");
        HIP_TRY(hipStreamSynchronize(st), "scan kernel");
        const uint64_t cursor = ct[0] + e->last_static_slabs;
        total = ct[1];
        e->last_nonascii_bits = (uint32_t)ct[2]; e->last_nonascii = e->last_nonascii_bits != 0;
        if (P.dbg & 2) {
            uint64_t c4[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpy(c4, e->d_dbg.p, 32, hipMemcpyDeviceToHost), "debug readback");
            fprintf(stderr, "[gft scan debug] units=%llu flagged=%llu sum_of_per_unit_max_lane=%llu to_bucket_table=%llu matches=%llu\n",
                    (unsigned long long)n_units, (unsigned long long)c4[0], (unsigned long long)c4[1],
                    (unsigned long long)c4[2], (unsigned long long)total);
        }
        if (cursor <= e->pool_cap) {
            if (e->use_scan4 && text_hi > text_lo) e->scan4_density = std::max(0.002, (double)total / (double)(text_hi - text_lo));
            if (P.ordered == 0 && text_hi > text_lo) {
                // a unit of maximal size should fill ~75 % of the fifo
                const double per_byte = (double)total / (double)(text_hi - text_lo);
                const double want = per_byte > 0 ? 0.75 * (e->use_scan5 ? e->s5plan.fifo_cap : kScan2FifoCap) / per_byte : (double)kScan2UnitMax;
                uint32_t um = want >= kScan2UnitMax ? kScan2UnitMax : (uint32_t)want & ~255u;
                e->scan2_unit_max = std::max<uint32_t>(512, um);
            }
            break;
        }
        if (attempt == 2) return fail(e, GFT_E_HIP, "match pool overflow persisted");
        rc = ensure_pool(e, cursor + cursor / 16);
        if (rc) return rc;
    }
    for (int attempt = 0; attempt < 3 && !e->use_scan2 && !e->use_scan3; attempt++) {
        if (attempt) HIP_TRY(hipMemsetAsync(e->d_ctl.as<uint8_t>() + 8, 0, 16, st), "memset");
        ScanParams P;
        P.text = d_text; P.doc_off = d_doc_off; P.units = e->d_units.as<Unit>(); P.n_units = n_units;
        P.byte_class = e->d_byte_class.as<uint8_t>(); P.delta = e->d_delta.as<uint32_t>();
        P.out_term = e->d_out_term.as<uint32_t>(); P.out_link = e->d_out_link.as<uint32_t>();
        P.term_len = e->d_term_len.as<uint32_t>();
        P.n_classes = e->tab.n_classes; P.n_states = e->tab.n_states; P.n_lds_states = e->n_lds_states;
        P.max_term_len = e->tab.max_term_len;
        P.pos_end = (e->build_flags & GFT_POS_END) ? 1 : 0;
        P.fold = (flags & GFT_FOLD_ASCII) ? 1 : 0;
        P.nonascii = e->d_ctl.as<uint32_t>() + 6;
        P.cursor = e->d_ctl.as<uint64_t>() + 1; P.pool_cap = e->pool_cap;
        P.pool_term = e->d_pool_term.as<uint32_t>(); P.pool_pos = e->d_pool_pos.as<uint32_t>();
        P.unit_start = e->d_unit_start.as<uint64_t>(); P.unit_count = e->d_unit_count.as<uint32_t>();
        e->last_static_slabs = 0;                                // (this kernel's cursor counts matches)
        {
            ProfScope ps(e, "scan");
            HIP_TRY(launch_scan_units(P, e->n_cus, st), "scan kernel launch");
        }
        if (e->deferred) return GFT_OK;
        uint64_t ct[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyAsync(ct, e->d_ctl.as<uint8_t>() + 8, 24, hipMemcpyDeviceToHost, st), "readback");
        HIP_TRY(hipStreamSynchronize(st), "scan kernel");
        total = ct[0];
        e->last_nonascii_bits = (uint32_t)ct[2]; e->last_nonascii = e->last_nonascii_bits != 0;
        if (total <= e->pool_cap) break;
        if (attempt == 2) return fail(e, GFT_E_HIP, "match pool overflow persisted");
        rc = ensure_pool(e, total + total / 16);
        if (rc) return rc;
    }

    *n_matches = total;
    e->last_n_units = n_units; e->last_total = total;
    if (!need_csr) return GFT_OK;   // the solver reads the slabs in place (doc -> units -> pool)
    return csr_from_pool(e, n_docs);
}

// After the last kernel of a batch whose scan was launched blind (scan_pipeline, defer_ok): ONE read-back of the control
// block -- the batch's only host synchronisation.  *again = the unit table or the match pool was too small (they have
// been grown): the caller runs the batch once more, this time with the sizes known.
// (what the launch of a deferred batch knew: the engine's fields at that time, or a pipelined batch's snapshot of them)
struct DeferredLaunch { bool single; uint64_t n_docs, unit_cap, static_slabs; uint32_t epoch; };   // epoch: of a k_units_single batch (its flags)
int deferred_interpret(gft_engine* e, const uint64_t* rb, const DeferredLaunch& dl, bool* again);

int deferred_check(gft_engine* e, bool* again) {
    *again = false;
    if (!e->deferred) return GFT_OK;
    e->deferred = false;
    // (into pinned memory: a copy to pageable memory is staged by the runtime, ten microseconds on every batch)
    if (!e->pin_rb) HIP_TRY(hipHostMalloc((void**)&e->pin_rb, 64, hipHostMallocDefault), "pinned alloc");
    uint64_t* rb = e->pin_rb;
    HIP_TRY(hipMemcpyAsync(rb, e->d_ctl.p, 7 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream), "readback");
    HIP_TRY(hipStreamSynchronize(e->stream), "process pipeline");
    return deferred_interpret(e, rb, DeferredLaunch{e->deferred_single, e->deferred_n_docs, e->deferred_unit_cap, e->last_static_slabs, e->deferred_single ? e->deferred_epoch : 0u}, again);
}

int deferred_interpret(gft_engine* e, const uint64_t* rb, const DeferredLaunch& dl, bool* again) {
    *again = false;
    const uint64_t cursor = rb[1] + dl.static_slabs, total = rb[2], n_units = rb[4], text_lo = rb[5], text_hi = rb[6];
    e->last_nonascii_bits = (uint32_t)rb[3]; e->last_nonascii = e->last_nonascii_bits != 0;
    // (a k_units_single batch raises its flags to its epoch; what an earlier batch left there is smaller)
    const bool flag_bad = dl.single ? (uint32_t)rb[0] == dl.epoch : (uint32_t)rb[0] != 0;
    if (dl.single && (uint32_t)(rb[3] >> 32) == dl.epoch) {      // a document of more than one unit: the general path
        e->single_streak = -8;
        *again = true;
        return GFT_OK;
    }
    e->single_streak = n_units == dl.n_docs ? e->single_streak + 1 : std::min(e->single_streak, 0);
    e->last_text_lo = text_lo; e->last_text_hi = text_hi;
    e->last_n_units = n_units; e->last_total = total;
    if (text_hi < text_lo) return fail(e, GFT_E_INVALID, "doc_off is not ascending");
    if (flag_bad) return fail(e, GFT_E_INVALID, "doc_off is not ascending, or a document is longer than 4 GiB - 1 bytes (positions are 32-bit)");
    // (the DFA kernel's cursor counts matches, the suffix-window kernels' slabs: both must fit the pool)
    if (n_units > dl.unit_cap || cursor > e->pool_cap) {
        if (cursor > e->pool_cap) { int rc = ensure_pool(e, cursor + cursor / 16); if (rc) return rc; }
        *again = true;
        return GFT_OK;
    }
    if (e->use_scan4 && text_hi > text_lo) e->scan4_density = std::max(0.002, (double)total / (double)(text_hi - text_lo));
    if (e->use_scan2 && !e->use_scan3 && !e->opt_scan_ordered && text_hi > text_lo) {
        // scan2: a unit of maximal size should fill ~75 % of the fifo
        const double per_byte = (double)total / (double)(text_hi - text_lo);
        const double want = per_byte > 0 ? 0.75 * (e->use_scan5 ? e->s5plan.fifo_cap : kScan2FifoCap) / per_byte : (double)kScan2UnitMax;
        const uint32_t um = want >= kScan2UnitMax ? kScan2UnitMax : (uint32_t)want & ~255u;
        e->scan2_unit_max = std::max<uint32_t>(512, um);
    }
    return GFT_OK;
}

// A folded scan that met bytes >= 0x80: is ASCII folding still the whole of strings.ToLower for this text (k_fold_safe)?
// One more pass over the text and one more read-back, for such batches only.
int refine_nonascii(gft_engine* e, const uint8_t* d_text, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t flags) {
    if (!(flags & GFT_FOLD_ASCII)) { e->last_nonascii = false; return GFT_OK; }
    if (!e->last_nonascii) return GFT_OK;
    // (gft_scan3 / gft_scan5 judge the pieces that hold high bytes themselves -- gft_foldsafe_dev.hpp -- and say "unsafe"
    // or nothing; the other kernels only say that they saw some)
    if (!(e->last_nonascii_bits & 1u)) { e->last_nonascii = (e->last_nonascii_bits & 2u) != 0; return GFT_OK; }
    uint32_t flag = 0;
    {
        ProfScope ps(e, "aux");
        HIP_TRY(launch_fold_safe(d_text, e->last_text_lo, e->last_text_hi, d_doc_off, n_docs, e->d_ctl.as<uint32_t>() + 6, e->stream), "fold check");
    }
    HIP_TRY(hipMemcpyAsync(&flag, e->d_ctl.as<uint32_t>() + 6, 4, hipMemcpyDeviceToHost, e->stream), "readback");
    HIP_TRY(hipStreamSynchronize(e->stream), "fold check");
    e->last_nonascii = (flag & 2u) != 0;
    return GFT_OK;
}

// GFT_SCAN_UNIQUE: the canonical CSR in d_match_off / d_term -> every term once per document, first occurrences in order.
// The result replaces d_match_off / d_term (positions: zeros in d_pos); *n_matches = new total.
int unique_pipeline(gft_engine* e, uint64_t n_docs, uint64_t* n_matches) {
    if (!n_docs) return GFT_OK;
    hipStream_t st = e->stream;
    const uint32_t n_terms = std::max<uint32_t>((uint32_t)e->tab.terms.size(), 1);
    // as many workgroups as 256 MB of first-occurrence rows allow, at most 4 per CU
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n_docs, (uint64_t)e->n_cus * 4), (256ull << 20) / ((uint64_t)n_terms * 4)));
    HIP_TRY(e->d_uq_first.ensure((size_t)grid * n_terms * 4), "unique alloc");
    HIP_TRY(e->d_uq_cnt.ensure(n_docs * 4), "unique alloc");
    HIP_TRY(e->d_uq_off.ensure((n_docs + 1) * 8), "unique alloc");
    HIP_TRY(e->d_partial.ensure(scan_partials_needed(n_docs) * 8), "unique alloc");
    HIP_TRY(hipMemsetAsync(e->d_uq_first.p, 0xFF, (size_t)grid * n_terms * 4, st), "memset");
    ProfScope ps(e, "aux");
    HIP_TRY(launch_unique_terms(false, e->d_match_off.as<uint64_t>(), e->d_term.as<uint32_t>(), n_docs, n_terms, e->d_uq_first.as<uint32_t>(), grid,
                                e->d_uq_cnt.as<uint32_t>(), nullptr, nullptr, st), "unique count");
    HIP_TRY(launch_exclusive_scan(e->d_uq_cnt.as<uint32_t>(), n_docs, e->d_uq_off.as<uint64_t>(), e->d_partial.as<uint64_t>(), st), "unique scan");
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, e->d_uq_off.as<uint64_t>() + n_docs, 8, hipMemcpyDeviceToHost, st), "readback");
    HIP_TRY(hipStreamSynchronize(st), "unique scan");
    HIP_TRY(e->d_uq_term.ensure(std::max<uint64_t>(total, 1) * 4), "unique alloc");
    HIP_TRY(launch_unique_terms(true, e->d_match_off.as<uint64_t>(), e->d_term.as<uint32_t>(), n_docs, n_terms, e->d_uq_first.as<uint32_t>(), grid,
                                nullptr, e->d_uq_off.as<uint64_t>(), e->d_uq_term.as<uint32_t>(), st), "unique write");
    // the caller-visible buffers: offsets and terms are swapped in, positions are all zero (substringEngine.go:83)
    std::swap(e->d_match_off, e->d_uq_off);
    std::swap(e->d_term, e->d_uq_term);
    HIP_TRY(e->d_pos.ensure(std::max<uint64_t>(total, 1) * 4), "unique alloc");
    HIP_TRY(hipMemsetAsync(e->d_pos.p, 0, std::max<uint64_t>(total, 1) * 4, st), "memset");
    *n_matches = total;
    return GFT_OK;
}

// GFT_POS_RUNES: the positions of the canonical CSR in d_pos become offsets over []rune(text), what AnknownEngine reports
// (finder/substringEngine.go:44-53: MultiPatternSearch([]rune(text), ...), Position = m.Pos)
int rune_pipeline(gft_engine* e, const uint8_t* d_text, const uint64_t* d_doc_off, uint64_t n_docs, uint64_t n_matches) {
    if (!n_docs || !n_matches) return GFT_OK;
    hipStream_t st = e->stream;
    HIP_TRY(e->d_rn_cnt.ensure(n_docs * 4), "rune alloc");
    HIP_TRY(e->d_rn_base.ensure((n_docs + 1) * 8), "rune alloc");
    HIP_TRY(e->d_partial.ensure(scan_partials_needed(n_docs) * 8), "rune alloc");
    ProfScope ps(e, "aux");
    HIP_TRY(launch_rune_doc_blocks(d_doc_off, n_docs, e->d_rn_cnt.as<uint32_t>(), st), "rune blocks");
    HIP_TRY(launch_exclusive_scan(e->d_rn_cnt.as<uint32_t>(), n_docs, e->d_rn_base.as<uint64_t>(), e->d_partial.as<uint64_t>(), st), "rune scan");
    uint64_t n_blocks = 0;
    HIP_TRY(hipMemcpyAsync(&n_blocks, e->d_rn_base.as<uint64_t>() + n_docs, 8, hipMemcpyDeviceToHost, st), "readback");
    HIP_TRY(hipStreamSynchronize(st), "rune scan");
    HIP_TRY(e->d_rn_starts.ensure(std::max<uint64_t>(n_blocks, 1) * 4), "rune alloc");
    HIP_TRY(e->d_rn_prefix.ensure((n_blocks + 1) * 8), "rune alloc");
    HIP_TRY(e->d_partial.ensure(scan_partials_needed(std::max(n_blocks, n_docs)) * 8), "rune alloc");
    HIP_TRY(launch_rune_block_starts(d_text, d_doc_off, e->d_rn_base.as<uint64_t>(), n_docs, n_blocks, e->d_rn_starts.as<uint32_t>(), st), "rune starts");
    HIP_TRY(launch_exclusive_scan(e->d_rn_starts.as<uint32_t>(), n_blocks, e->d_rn_prefix.as<uint64_t>(), e->d_partial.as<uint64_t>(), st), "rune scan");
    HIP_TRY(launch_pos_to_rune(d_text, d_doc_off, e->d_rn_base.as<uint64_t>(), e->d_rn_prefix.as<uint64_t>(), e->d_match_off.as<uint64_t>(), n_docs,
                               n_matches, e->d_pos.as<uint32_t>(), st), "rune offsets");
    return GFT_OK;
}

int solve_pipeline(gft_engine* e, uint64_t n_docs, const gft_extra_matches* d_extra, uint32_t* d_bitmap) {
    if (!n_docs || !e->n_exprs) return GFT_OK;
    SolveParams S;
    S.doc_unit_base = e->d_unit_base.as<uint64_t>();
    S.unit_start = e->d_unit_start.as<uint64_t>(); S.unit_count = e->d_unit_count.as<uint32_t>();
    S.units = e->d_units.as<Unit>();
    S.has_rare = e->n_rare_words > 0 ? 1u : 0u;
    S.pos_back = (e->build_flags & GFT_POS_END) ? 0u : (e->tab.max_term_len ? e->tab.max_term_len - 1 : 0u);
    S.term = e->d_pool_term.as<uint32_t>(); S.pos = e->d_pool_pos.as<uint32_t>();
    S.x_off = d_extra ? d_extra->off : nullptr;
    S.x_slot = d_extra ? d_extra->slot : nullptr;
    S.x_pos = d_extra ? d_extra->pos : nullptr;
    S.n_docs = n_docs;
    S.fprog = e->d_fprog.as<uint32_t>(); S.fprog_off = e->d_fprog_off.as<uint64_t>();
    S.gprog = e->d_prog.as<uint32_t>(); S.groups = e->d_groups.as<uint32_t>();
    S.order = e->d_order.as<uint32_t>(); S.blk_class = e->d_blk_class.as<uint32_t>(); S.wave_blk = e->d_wave_blk.as<uint32_t>();
    S.fprog_t = e->d_fprog_t.as<uint32_t>(); S.fblk_off = e->d_fblk_off.as<uint32_t>();
    S.n_exprs = e->n_exprs;
    S.n_slots = (uint32_t)e->tab.terms.size() + e->n_extra + 1;
    S.tile_words = std::min<uint32_t>(kSolveTileWords, (e->n_exprs + 31) / 32);
    S.bitmap = d_bitmap;
    S.p_scratch = nullptr;
    S.dbg = e->opt_solve_dbg;
    S.dbg_out = nullptr;
    if (S.dbg & 8) {
        HIP_TRY(e->d_solve_dbg.ensure(128 * 8), "debug alloc");
        HIP_TRY(hipMemsetAsync(e->d_solve_dbg.p, 0, 128 * 8, e->stream), "memset");
        S.dbg_out = e->d_solve_dbg.as<unsigned long long>();
    }
    // Presence matrix in LDS next to the output tile: G documents per group = G / 8 bytes per slot, the widest G of
    // 64 / 32 / 16 / 8 that fits (GFT_SOLVE_GROUP_DOCS forces one, for tests); beyond that in HBM (served by L2), G = 64
    S.fprog_words = e->fprog_words;
    uint32_t group_docs = 64;
    bool p_in_lds = false;
    for (uint32_t G : {64u, 32u, 16u, 8u}) {
        if (e->opt_solve_group >= 0 && (uint32_t)e->opt_solve_group != G) continue;
        if (solve_lds_bytes(S.n_slots, S.tile_words, G, true, 0, 0, false) + 1024 <= e->lds_max) { group_docs = G; p_in_lds = true; break; }
    }
    // ... and the fused programs too, if there is room left (the interpreter fetches them word after word)
    // (a set with a wide INORD group runs the kernel variant that reads its programs from L2: launch_g)
    const bool prog_in_lds = !e->wide_pairs && solve_lds_bytes(S.n_slots, S.tile_words, group_docs, p_in_lds, S.fprog_words, S.n_exprs, true) + 1024 <= e->lds_max;
    const uint64_t n_groups = (n_docs + group_docs - 1) / group_docs;
    const size_t lds_need = solve_lds_bytes(S.n_slots, S.tile_words, group_docs, p_in_lds, S.fprog_words, S.n_exprs, prog_in_lds) + 512;
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, e->lds_max / lds_need));
    unsigned grid = (unsigned)std::min<uint64_t>(n_groups, (uint64_t)e->n_cus * per_cu);
    if (!p_in_lds) {
        HIP_TRY(e->d_pscratch.ensure((size_t)grid * S.n_slots * 8), "presence scratch alloc");
        S.p_scratch = e->d_pscratch.as<uint64_t>();
    }
    S.wide_slot = nullptr; S.wide_theta = nullptr; S.wide_cap = 0; S.wide_list = nullptr; S.n_wide = 0;
    if (e->wide_pairs) {
        // (a region per wave of the grid; 12 bytes per pair: 8 192 pairs x 4 096 waves = 400 MB at the very most)
        const uint64_t n_waves = (uint64_t)grid * (kSolveBlockThreads / 64);
        S.wide_cap = (e->wide_pairs + 63u) & ~63u;
        HIP_TRY(e->d_wide_slot.ensure(n_waves * S.wide_cap * 4), "INORD scratch alloc");
        HIP_TRY(e->d_wide_theta.ensure(n_waves * S.wide_cap * 8), "INORD scratch alloc");
        S.wide_slot = e->d_wide_slot.as<uint32_t>();
        S.wide_theta = e->d_wide_theta.as<long long>();
        S.wide_list = e->d_wide_list.as<uint32_t>(); S.n_wide = e->n_wide;
    }
    e->last_solve_group_docs = p_in_lds ? group_docs : 0;
    ProfScope ps(e, "solve");
    HIP_TRY(launch_solve(S, group_docs, p_in_lds, prog_in_lds, grid, e->stream), "solve kernel launch");
    if (S.dbg & 8) {
        // phase clocks: cycles per group and wave (0 build, 1 barrier, 2 evaluation, 3 barrier, 4 transpose + wipe, 5 barrier,
        // 6 bitmap rows, 7 loop head), averaged over the workgroups
        unsigned long long t[128];
        HIP_TRY(hipMemcpyAsync(t, e->d_solve_dbg.p, sizeof t, hipMemcpyDeviceToHost, e->stream), "debug read-back");
        HIP_TRY(hipStreamSynchronize(e->stream), "debug read-back");
        fprintf(stderr, "[gft solve debug] cycles per group: wave | build bar eval bar transpose bar rows head\n");
        for (int w = 0; w < 16; w++) {
            fprintf(stderr, "[gft solve debug] %2d |", w);
            for (int ph = 0; ph < 8; ph++) fprintf(stderr, " %7.0f", (double)t[w * 8 + ph] / (double)n_groups);
            fprintf(stderr, "\n");
        }
    }
    return GFT_OK;
}

// ---- what the host solves (host_solve.hpp) ---------------------------------------------------------------------------
struct HostPlan {
    bool all_docs = false;                     // some expression is beyond the device solver's limits: every document
    std::vector<uint64_t> irregular;           // documents in which a slot read by an INORD group may have a non-ascending list
    bool empty() const { return !all_docs && irregular.empty(); }
};

// extra: the caller's matches as HOST arrays (nullable).  A slot's list is what addMatchesToSolverMap builds
// (finder/finder.go:181-196): the scan's positions of the term, then the caller's in the order given.  It can only be out
// of order when the caller's matches name a dictionary term (a regex with the text of a keyword), or are themselves not
// ascending (a foreign engine's keyword hits followed by the regex engine's for the same literal).
void plan_host(const gft_engine* e, const gft_extra_matches* extra, uint64_t n_docs, HostPlan& plan) {
    plan.all_docs = !e->host_only.empty();
    plan.irregular.clear();
    if (!extra || !extra->off || e->inord_exprs.empty() || !n_docs) return;
    const uint32_t n_terms = (uint32_t)e->tab.terms.size();
    std::vector<std::pair<uint32_t, uint32_t>> seen;          // (slot, last position) of this document: a handful
    for (uint64_t d = 0; d < n_docs; d++) {
        seen.clear();
        bool irr = false;
        for (uint64_t i = extra->off[d]; i < extra->off[d + 1] && !irr; i++) {
            const uint32_t sl = extra->slot[i];
            if (sl >= e->inord_slot.size() || !e->inord_slot[sl]) continue;      // (range errors are upload_extra's to report)
            if (sl < n_terms) { irr = true; break; }
            size_t k = 0;
            while (k < seen.size() && seen[k].first != sl) k++;
            if (k == seen.size()) seen.emplace_back(sl, extra->pos[i]);
            else { irr = extra->pos[i] < seen[k].second; seen[k].second = extra->pos[i]; }
        }
        if (irr) plan.irregular.push_back(d);
    }
}

// Solve the planned (expression, document) pairs on the host from the scan's matches and the caller's, and put their bits
// into the bitmap: h_bitmap (host rows, already downloaded) or d_bitmap (device rows, patched by a small kernel).
int host_eval(gft_engine* e, const gft_extra_matches* extra, uint64_t n_docs, const HostPlan& plan, uint32_t* h_bitmap,
              uint32_t* d_bitmap) {
    if (plan.empty() || !n_docs || !e->n_exprs) return GFT_OK;
    hipStream_t st = e->stream;
    if (!e->csr_valid) { int rc = csr_from_pool(e, n_docs); if (rc) return rc; }
    std::vector<uint64_t> mo(n_docs + 1);
    HIP_TRY(hipMemcpyAsync(mo.data(), e->d_match_off.p, (n_docs + 1) * 8, hipMemcpyDeviceToHost, st), "download");
    HIP_TRY(hipStreamSynchronize(st), "host solve");
    // the matches of the documents in question: all of them, or the irregular documents' ranges
    std::vector<uint64_t> docs;
    if (plan.all_docs) { docs.resize(n_docs); for (uint64_t d = 0; d < n_docs; d++) docs[d] = d; }
    else docs = plan.irregular;
    std::vector<uint32_t> ti, po;
    std::vector<uint64_t> at(docs.size() + 1, 0);           // document k's matches: [at[k], at[k + 1]) of ti / po
    for (size_t k = 0; k < docs.size(); k++) at[k + 1] = at[k] + (mo[docs[k] + 1] - mo[docs[k]]);
    ti.resize(at.back() + 1); po.resize(at.back() + 1);
    if (plan.all_docs) {
        if (at.back()) {
            HIP_TRY(hipMemcpyAsync(ti.data(), e->d_term.p, at.back() * 4, hipMemcpyDeviceToHost, st), "download");
            HIP_TRY(hipMemcpyAsync(po.data(), e->d_pos.p, at.back() * 4, hipMemcpyDeviceToHost, st), "download");
        }
    } else {
        for (size_t k = 0; k < docs.size(); k++) {
            const uint64_t n = at[k + 1] - at[k];
            if (!n) continue;
            HIP_TRY(hipMemcpyAsync(ti.data() + at[k], e->d_term.as<uint32_t>() + mo[docs[k]], n * 4, hipMemcpyDeviceToHost, st), "download");
            HIP_TRY(hipMemcpyAsync(po.data() + at[k], e->d_pos.as<uint32_t>() + mo[docs[k]], n * 4, hipMemcpyDeviceToHost, st), "download");
        }
    }
    HIP_TRY(hipStreamSynchronize(st), "host solve");
    const uint64_t words = (e->n_exprs + 31) / 32;
    std::vector<uint64_t> pw;                                // patches for a device bitmap: word index, bits to clear, bits to set
    std::vector<uint32_t> pclr, pset;
    SlotLists lists;
    size_t ir = 0;                                           // next irregular document
    for (size_t k = 0; k < docs.size(); k++) {
        const uint64_t d = docs[k];
        while (ir < plan.irregular.size() && plan.irregular[ir] < d) ir++;
        const bool irregular = ir < plan.irregular.size() && plan.irregular[ir] == d;
        // sortedMatchesByKeyword of this document (finder/finder.go:181-196): the engine's matches first (emission order:
        // ascending per term), the caller's behind them in the order given
        lists.clear();
        for (uint64_t i = at[k]; i < at[k + 1]; i++) lists[ti[i]].push_back((int64_t)po[i]);
        if (extra && extra->off)
            for (uint64_t i = extra->off[d]; i < extra->off[d + 1]; i++) lists[extra->slot[i]].push_back((int64_t)extra->pos[i]);
        auto solve_one = [&](uint32_t x) {
            const bool hit = host_solve(e->h_prog.data() + e->h_prog_off[x], e->h_prog_off[x + 1] - e->h_prog_off[x], lists);
            const uint64_t w = d * words + (x >> 5);
            const uint32_t bit = 1u << (x & 31);
            if (h_bitmap) h_bitmap[w] = hit ? h_bitmap[w] | bit : h_bitmap[w] & ~bit;
            else { pw.push_back(w); pclr.push_back(hit ? 0u : bit); pset.push_back(hit ? bit : 0u); }
        };
        for (uint32_t x : e->host_only) solve_one(x);
        if (irregular) for (uint32_t x : e->inord_exprs) solve_one(x);
    }
    if (!h_bitmap && !pw.empty()) {
        if (!d_bitmap) return fail(e, GFT_E_INVALID, "null bitmap");
        const size_t n = pw.size();
        HIP_TRY(e->d_patch.ensure(n * 16), "patch alloc");
        uint8_t* base = e->d_patch.as<uint8_t>();
        HIP_TRY(hipMemcpyAsync(base, pw.data(), n * 8, hipMemcpyHostToDevice, st), "patch upload");
        HIP_TRY(hipMemcpyAsync(base + n * 8, pclr.data(), n * 4, hipMemcpyHostToDevice, st), "patch upload");
        HIP_TRY(hipMemcpyAsync(base + n * 12, pset.data(), n * 4, hipMemcpyHostToDevice, st), "patch upload");
        HIP_TRY(launch_patch_words(d_bitmap, reinterpret_cast<const uint64_t*>(base), reinterpret_cast<const uint32_t*>(base + n * 8),
                                   reinterpret_cast<const uint32_t*>(base + n * 12), n, st), "patch");
        HIP_TRY(hipStreamSynchronize(st), "patch");
    }
    return GFT_OK;
}

}  // namespace

// ---- compiled tables as one blob (BuildEngine for a large dictionary is paid once) -----------------
namespace {
constexpr uint32_t kTablesMagic = 0x54544647u;   // "GFTT"
constexpr uint32_t kTablesVersion = 9;           // bump when a table layout or a hash function changes

struct Writer {
    std::vector<uint8_t> b;
    void raw(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    template <class T> void vec(const std::vector<T>& v) { u64(v.size()); if (!v.empty()) raw(v.data(), v.size() * sizeof(T)); }
};
struct Reader {
    const uint8_t* p; uint64_t n, i = 0; bool ok = true;
    bool raw(void* d, size_t k) { if (!ok || k > n - i) { ok = false; return false; } memcpy(d, p + i, k); i += k; return true; }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    template <class T> void vec(std::vector<T>& v) {
        const uint64_t k = u64();
        if (!ok || k > (n - i) / sizeof(T)) { ok = false; return; }
        v.resize((size_t)k);
        if (k) raw(v.data(), (size_t)k * sizeof(T));
    }
};
// Every index a kernel follows must stay inside the table it indexes: a blob that passes the checksum may still be stale
// (another library build) or crafted.  Returns what is wrong, or nullptr.
const char* validate_tables(const AcTables& a, const Scan2Tables& t, const Scan3Tables& u) {
    const size_t n_terms = a.terms.size();
    if (a.n_classes == 0 || a.n_classes > 256) return "class count";
    for (int b = 0; b < 256; b++) if (a.byte_class[b] >= a.n_classes) return "byte class";
    for (uint32_t d : a.delta) if ((d & ~kOutFlag) >= a.n_states) return "DFA target";
    for (uint32_t x : a.out_term) if (x != kNoTerm && x >= n_terms) return "DFA output term";
    for (uint32_t x : a.out_link) if (x >= a.n_states) return "DFA output link";
    for (size_t i = 0; i < n_terms; i++) if (a.term_len[i] != a.terms[i].size()) return "term length";
    auto slots_ok = [&](const std::vector<Scan2Slot>& slots, const std::vector<Scan2Slot>& more, uint32_t shift, const std::vector<uint8_t>& blob,
                        const std::vector<uint32_t>& off) -> const char* {
        if (shift < 1 || shift > 31 || slots.size() != ((size_t)1 << (32 - shift))) return "bucket table size";
        if (off.size() != n_terms + 1) return "term offsets";
        for (size_t i = 0; i < n_terms; i++)
            if (off[i] < 4 || (uint64_t)off[i] + a.terms[i].size() + 8 > blob.size()) return "term offset";
        auto entry_ok = [&](const Scan2Slot& s) {
            const uint32_t len1 = s.len & kScan2LenMask;
            const int off8 = (int)(int8_t)(s.len >> 24);
            return s.info < n_terms && off8 >= -1 && off8 <= (int)kScan2MaxOff && (int64_t)len1 + off8 == (int64_t)a.terms[s.info].size() && len1 >= 4;
        };
        for (const Scan2Slot& s : slots) {
            if (s.key == kScan2EmptyKey) continue;
            if (s.info & kScan2Multi) {
                const uint64_t at = s.info & ~kScan2Multi;
                if (at + s.len > more.size() || s.len == 0) return "bucket list";
            } else if (!entry_ok(s)) return "bucket entry";
        }
        for (const Scan2Slot& s : more) if (s.key != kScan2EmptyKey && !entry_ok(s)) return "bucket list entry";
        return nullptr;
    };
    if (t.supported) {
        if (t.kp == 0 || t.kp > 256 || t.pad_class >= t.kp) return "scan2 classes";
        for (int b = 0; b < 256; b++) if (t.cls[b] >= t.kp || t.cls_fold[b] >= t.kp) return "scan2 byte class";
        // build_scan5_tables (run on imported tables too) indexes its class counters by the automaton's byte classes and splits
        // every bucket key into four classes: the two class maps must be one, and a key must be four classes
        if (t.kp != a.n_classes) return "scan2 class count differs from the automaton's";
        for (int b = 0; b < 256; b++) if (t.cls[b] != a.byte_class[b]) return "scan2 byte class differs from the automaton's";
        // (a key lives in ITS pair of the bucket table and nowhere else: the kernels look nowhere else)
        if (t.slot_shift < 1 || t.slot_shift > 31) return "bucket table size";
        for (size_t i = 0; i < t.slots.size(); i++)
            if (t.slots[i].key != kScan2EmptyKey && (scan2_pair_slot(t.slots[i].key, 0, t.slot_shift, t.slot_seed) | 1u) != ((uint32_t)i | 1u)) return "bucket placement";
        {
            const uint64_t kp4 = (uint64_t)t.kp * t.kp * t.kp * t.kp;
            for (const Scan2Slot& s : t.slots) if (s.key != kScan2EmptyKey && s.key >= kp4) return "bucket key";
            for (const Scan2Slot& s : t.more) if (s.key != kScan2EmptyKey && s.key >= kp4) return "bucket list key";
        }
        if (t.hashed ? (t.hash_shift < 1 || t.hash_shift > 31 || t.filter_bits != (1u << (32 - t.hash_shift)))
                     : (uint64_t)t.kp * t.kp * t.kp * t.kp > t.filter_bits) return "scan2 filter size";
        if (!t.short3.empty() && t.short3.size() < (uint64_t)t.kp * t.kp * t.kp) return "scan2 short3 size";
        if (!t.short3_big.empty() && t.short3_big.size() != t.short3.size()) return "scan2 short3_big size";
        if (t.shorts_packed.size() != t.shorts.size() * 3) return "scan2 short records";
        for (uint8_t id : t.short3) if (id != 255 && id >= t.shorts.size()) return "scan2 short record id";
        for (uint32_t id : t.short3_big) if (id >= t.shorts.size()) return "scan2 short record id";
        for (uint32_t w : t.shorts_packed) if (w && ((w & 0x0FFFFFFFu) >= n_terms || (w >> 28) > 3)) return "scan2 short record";
        if (const char* why = slots_ok(t.slots, t.more, t.slot_shift, t.term_blob, t.term_off)) return why;
    }
    if (u.supported) {
        if (u.G == 0 || u.G > kScan3Groups) return "scan3 groups";
        const uint64_t G3 = (uint64_t)u.G * u.G * u.G;
        for (int b = 0; b < 256; b++) if (u.cls[b] >= u.G || u.cls_fold[b] >= u.G) return "scan3 byte group";
        if (u.filter.size() != (size_t)((G3 * u.G + 31) / 32)) return "scan3 filter size";
        if (!u.short3.empty() && (u.short3.size() < G3 || u.short3.size() % 16)) return "scan3 short3 size";
        if (!u.short3_big.empty() && u.short3_big.size() != u.short3.size()) return "scan3 short3_big size";
        if (u.srec.size() % kScan3RecWords || u.srec.empty() || u.srec.size() / kScan3RecWords > kScan3RecLds + 1) return "scan3 records";
        for (uint8_t id : u.short3) if (id != 255 && id >= u.srec.size() / kScan3RecWords) return "scan3 record id";
        for (size_t i = 0; i < u.short3.size(); i++) if (u.short3[i] == 255 && (u.short3_big.empty() || u.short3_big[i] >= u.srec_big.size())) return "scan3 big record";
        for (size_t i = 0; i < u.srec.size(); i += 2) if (u.srec[i] && ((u.srec[i] & 0x0FFFFFFFu) >= n_terms || (u.srec[i] >> 28) > 3)) return "scan3 record entry";
        for (size_t at = 0; at < u.srec_big.size();) {
            const uint64_t n = u.srec_big[at];
            if (at + 1 + 2 * n > u.srec_big.size()) return "scan3 big record length";
            for (uint64_t j = 0; j < n; j++) { const uint32_t w = u.srec_big[at + 1 + 2 * j]; if (w && ((w & 0x0FFFFFFFu) >= n_terms || (w >> 28) > 3)) return "scan3 big record entry"; }
            at += 1 + 2 * n;
        }
        if (u.bloom_lg < 1 || u.bloom_lg > 28 || u.bloom.size() != ((size_t)1 << u.bloom_lg)) return "scan3 bloom size";
        if (const char* why = slots_ok(u.slots, u.more, u.slot_shift, u.term_blob, u.term_off)) return why;
    }
    return nullptr;
}


void destroy_multi(gft_engine* e);
// multi-device dispatch (definitions behind the single-device entry points)
int multi_process(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                  const gft_extra_matches* extra, uint32_t* hit_bitmap);
int multi_process_again(gft_engine* e, uint64_t n_docs, const gft_extra_matches* extra, uint32_t* hit_bitmap);
int multi_scan(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags, gft_matches* out);
int multi_set_programs(gft_engine* e, const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs, uint32_t n_extra);
int multi_build(gft_engine* e, const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, uint32_t flags);
int multi_import_tables(gft_engine* e, const uint8_t* blob, uint64_t len);

}  // namespace


extern "C" {

int gft_engine_create(gft_engine** out, int device) try {
    if (!out) return GFT_E_INVALID;
    *out = nullptr;
    std::unique_ptr<gft_engine> owner(new gft_engine());     // (released into *out on every regular way out)
    gft_engine* e = owner.get();
    int count = 0;
    hipError_t h = hipGetDeviceCount(&count);
    if (h != hipSuccess || count == 0) {
        // keep the handle so the caller can read the message, but every compute call will fail loudly
        e->device = -1;
        e->err = std::string("no HIP device available: ") + (h != hipSuccess ? hipGetErrorString(h) : "device count is 0");
        *out = owner.release();
        return GFT_E_HIP;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    e->device = device;
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        e->n_cus_hw = prop.multiProcessorCount > 0 ? (unsigned)prop.multiProcessorCount : 256;
        if (const char* m = getenv("GFT_CU_MARGIN")) { const int v = atoi(m); if (v > 0) e->cu_margin = (unsigned)v; }
        e->n_cus = e->n_cus_hw > e->cu_margin ? e->n_cus_hw - e->cu_margin : 1;
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) != hipSuccess) optin = 0;
        e->lds_max = std::max<size_t>(prop.sharedMemPerBlock, (size_t)std::max(optin, 0));
        if (std::string(prop.gcnArchName).find("gfx950") != std::string::npos) e->lds_max = std::max<size_t>(e->lds_max, 160 * 1024);
    }
    // a blocking stream: ordered with the legacy default stream, which is where a caller that never names a stream
    // (torch's default stream, plain hipMemcpy) produces the device buffers it hands to *_device entry points
    if (hipStreamCreateWithFlags(&e->stream, hipStreamDefault) == hipSuccess) e->own_stream = true;
    else e->stream = nullptr;
    refresh_options(e);
    *out = owner.release();
    return GFT_OK;
} GFT_CATCH(nullptr)

void gft_engine_destroy(gft_engine* e) {
    if (!e) return;
    destroy_multi(e);
    if (e->device >= 0) {
        DeviceGuard g(e->device);
        if (e->stream) (void)hipStreamSynchronize(e->stream);
        for (auto& kv : e->prof)
            for (auto& p : kv.second.ev) { e->prof_pool.push_back(p.first); e->prof_pool.push_back(p.second); }
        for (hipEvent_t ev : e->prof_pool) (void)hipEventDestroy(ev);
        DevBuf* all[] = {&e->d_byte_class, &e->d_delta, &e->d_out_term, &e->d_out_link, &e->d_term_len, &e->d_prog,
                         &e->d_prog_off, &e->d_fprog, &e->d_fprog_off, &e->d_groups, &e->d_wide_slot, &e->d_wide_theta, &e->d_wide_list, &e->d_order, &e->d_blk_class, &e->d_wave_blk, &e->d_fprog_t, &e->d_fblk_off, &e->d_pscratch, &e->d_solve_dbg, &e->d_s2_filter,
                         &e->d_s2_slots, &e->d_s2_more, &e->d_s2_cls, &e->d_s2_cls_fold, &e->d_s2_term_blob,
                         &e->d_s2_term_off, &e->d_ctl, &e->d_dbg, &e->d_s2_short3, &e->d_s2_shorts_packed, &e->d_s2_short3_big, &e->d_s2_fpt,
                         &e->d_s3_filter, &e->d_s3_short3, &e->d_s3_srec, &e->d_s3_short3_big, &e->d_s3_srec_big, &e->d_s3_bloom, &e->d_s3_slots,
                         &e->d_s3_more, &e->d_s3_cls, &e->d_s3_cls_fold, &e->d_s3_term_blob, &e->d_s3_term_off,
                         &e->d_s5_grp, &e->d_s5_grp_fold, &e->d_s5_filter, &e->d_s5_bloom,
&e->d_unit_cnt, &e->d_unit_base, &e->d_units, &e->d_partial,
                         &e->d_pool_term, &e->d_pool_pos, &e->d_unit_start, &e->d_unit_count, &e->d_unit_out,
                         &e->d_term, &e->d_pos, &e->d_match_off, &e->d_text, &e->d_doc_off, &e->d_bitmap, &e->d_xoff,
                         &e->d_xslot, &e->d_xpos, &e->d_uq_first, &e->d_uq_cnt, &e->d_uq_off, &e->d_uq_term, &e->d_patch, &e->d_rn_cnt, &e->d_rn_base, &e->d_rn_starts, &e->d_rn_prefix};
        for (DevBuf* b : all) b->release();
        for (int k = 0; k < 2; k++) {
            if (e->pin[k]) (void)hipHostFree(e->pin[k]);
            if (k == 0 && e->pin_rb) (void)hipHostFree(e->pin_rb);
            if (e->pin_ev[k]) (void)hipEventDestroy(e->pin_ev[k]);
            if (e->pend[k].rb) (void)hipHostFree(e->pend[k].rb);
            if (e->pend[k].ev) (void)hipEventDestroy(e->pend[k].ev);
        }
        if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    }
    delete e;
}

const char* gft_last_error(const gft_engine* e) { return e ? e->err.c_str() : "null engine"; }

int gft_set_stream(gft_engine* e, void* hip_stream) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    DeviceGuard g(e->device);
    if (e->own_stream && e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    e->own_stream = false;
    e->stream = (hipStream_t)hip_stream;
    if (!hip_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamDefault), "stream create");
        e->own_stream = true;
    }
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))


int gft_set_cu_margin(gft_engine* e, uint32_t margin) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (e->pend_count) return fail(e, GFT_E_INVALID, "batches are in flight (gft_process_device_end first)");
    e->cu_margin = margin;
    e->n_cus = e->n_cus_hw > margin ? e->n_cus_hw - margin : 1;
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

// e->tab / e->s2 hold compiled tables (from gft_build or gft_import_tables): check them against the device, upload
static int install_tables(gft_engine* e, uint32_t flags) {
    if (e->tab.max_term_len + 1024 > kTextBuf)
        return fail(e, GFT_E_UNSUPPORTED, "keyword longer than " + std::to_string(kTextBuf - 1024) + " bytes");
    if (e->tab.n_states >= 0x7FFFFFFFu) return fail(e, GFT_E_UNSUPPORTED, "automaton too large");
    e->build_flags = flags;
    refresh_options(e);

    DeviceGuard g(e->device);
    const size_t fixed = 256 + (size_t)(kScanBlockThreads / 64) * kTextBuf + 1024;
    if (e->lds_max < fixed + (size_t)e->tab.n_classes * 4)
        return fail(e, GFT_E_UNSUPPORTED, "device LDS too small for the scan kernel");
    size_t rows = (e->lds_max - fixed) / ((size_t)e->tab.n_classes * 4);
    e->n_lds_states = (uint32_t)std::min<size_t>(rows, e->tab.n_states);

    std::vector<uint8_t> bc(e->tab.byte_class, e->tab.byte_class + 256), c1, c2, g1, g2, s3v, s5g, s5gf;
    SyncOnExit drained(e);                              // (declared behind the temporaries the uploads read from: it goes first)
    int rc;
    if ((rc = upload(e, e->d_byte_class, bc, "table upload"))) return rc;
    if ((rc = upload(e, e->d_delta, e->tab.delta, "table upload"))) return rc;
    if ((rc = upload(e, e->d_out_term, e->tab.out_term, "table upload"))) return rc;
    if ((rc = upload(e, e->d_out_link, e->tab.out_link, "table upload"))) return rc;
    if ((rc = upload(e, e->d_term_len, e->tab.term_len, "table upload"))) return rc;
    // GFT_SCAN_KERNEL=dfa forces the general two-tier DFA kernel
    const char* force = getenv("GFT_SCAN_KERNEL");
    if (!kExtraKernels && force && (std::string(force) == "scan2" || std::string(force) == "scan4"))
        return fail(e, GFT_E_UNSUPPORTED, std::string("GFT_SCAN_KERNEL=") + force + ": this library was built without the cross-check kernels (GFT_EXTRA_KERNELS=1 python -m gofindthem_amd.build --force)");
    const bool k2_fits = e->s2.supported && scan2_plan((uint32_t)e->s2.filter.size(), (uint32_t)e->s2.short3.size(),
                                                        (uint32_t)std::min<size_t>(e->s2.shorts_packed.size(), 255 * 3),
                                                        e->s2.fpt_lg ? 0u : kScan2FptSize, e->lds_max - 512,
                                                        &e->scan2_k2_waves, &e->scan2_cand_cap);
    e->use_scan2 = k2_fits && !(force && std::string(force) == "dfa");
    {
        uint32_t w0 = 0, w1 = 0;
        const bool k4_fits = e->s2.supported &&
                             scan4_plan((uint32_t)e->s2.filter.size(), (uint32_t)e->s2.short3.size(), (uint32_t)std::min<size_t>(e->s2.shorts_packed.size(), 255 * 3),
                                        e->s2.fpt_lg ? 0u : kScan2FptSize, e->lds_max - 512, false, &w0, &e->scan4_fifo[0]) &&
                             scan4_plan((uint32_t)e->s2.filter.size(), (uint32_t)e->s2.short3.size(), (uint32_t)std::min<size_t>(e->s2.shorts_packed.size(), 255 * 3),
                                        e->s2.fpt_lg ? 0u : kScan2FptSize, e->lds_max - 512, true, &w1, &e->scan4_fifo[1]);
        e->scan4_waves = std::min(w0, w1);
        // (the fifo capacities belong to the smaller of the two wave counts; with fewer waves there is only more room)
        e->use_scan4 = e->use_scan2 && k4_fits && force && std::string(force) == "scan4";
        e->scan4_density = 0.06;
    }
    // The suffix-window kernel with one filter probe per two bytes: the default for every dictionary whose long-term tables
    // exist (GFT_SCAN_KERNEL=scan2: the one-probe-per-byte kernel, scan3: the stride-2 kernel).  It runs on scan2's tables; a
    // fifo entry of 32 bits holds term id and relative position.  With more than 32 byte classes there is no
    // direct short-term table (Scan2Tables::short_direct): the group-indexed one of the stride-2 kernel's tables serves then
    // (GFT_SCAN5_LARGE=0 leaves such dictionaries to the stride-2 kernel).
    e->use_scan5 = false;
    e->s5_short_groups = false;
    const bool want5 = !force || std::string(force) == "scan5" || std::string(force) == "auto" || !*force;
    const bool large5 = e->s2.long_ok && !e->s2.short_direct && e->s3.supported && e->opt_scan5_large;
    // (the tables that kernel needs: the direct short-term table of small alphabets, or the group-indexed one)
    const bool direct5 = e->s2.supported && !(force && std::string(force) == "dfa");
    if ((direct5 || large5) && e->s2.long_ok && want5) {
        uint32_t tb = 1;
        while ((1ull << tb) < e->tab.terms.size()) tb++;
        e->s5_term_bits = tb;
        e->s5_pos_bias = e->tab.max_term_len + kScan2MaxOff;
        const bool packs = (uint64_t)kScan2UnitMax + e->s5_pos_bias + 8 < (1ull << (32 - tb));
        const uint32_t short_bytes = large5 ? (uint32_t)e->s3.short3.size() : (uint32_t)e->s2.short3.size();
        const uint32_t rec_words = large5 ? (uint32_t)e->s3.srec.size() : (uint32_t)std::min<size_t>(e->s2.shorts_packed.size(), 255 * 3);
        // a fingerprint table too large for LDS (fpt_lg != 0) gets a Bloom level there instead: 2^lg bits, as large as
        // GFT_SCAN5_BLOOM_KB allows but not more than eight bits per item would take
        e->s5_bloom_lg = 0;
        e->s5_bloom.clear();
        if (e->s2.fpt_lg && e->opt_scan5_bloom_kb) {
            uint32_t lg = 13;
            while ((2u << lg) / 8 <= e->opt_scan5_bloom_kb * 1024u && (1ull << lg) < 8 * e->s2.n_keys) lg++;
            e->s5_bloom_lg = lg;
        }
        bool fits5 = false;
        for (int attempt = 0; attempt < 2 && packs && !fits5; attempt++) {
            const uint32_t in_lds = e->s2.fpt_lg ? (e->s5_bloom_lg ? (1u << e->s5_bloom_lg) / 8 : 0u) : kScan2FptSize;
            fits5 = scan5_plan(e->s2.kp, short_bytes, rec_words, in_lds, e->lds_max - 512, e->opt_scan5_fifo ? e->opt_scan5_fifo : kScan2FifoCap, &e->s5plan);
            if (!fits5) e->s5_bloom_lg = 0;                      // (no room: without the Bloom level)
        }
        if (fits5) {
            if (e->opt_scan5_groups && e->opt_scan5_groups < e->s5plan.G) {      // (tests: more merging than LDS asks for)
                e->s5plan.G = std::max<uint32_t>(e->opt_scan5_groups, 2);
                e->s5plan.dual_entries = e->s5plan.G * e->s5plan.G * e->s5plan.G;
            }
            build_scan5_tables(e->tab, e->s2, e->s5plan.G, e->s5);
            if (e->s5_bloom_lg) {
                // one bit per owner of a fingerprint cell, read off the bucket table: (window key, byte in front of the
                // window with its case bit cleared), or the window key alone where the window is the term's first four bytes
                e->s5_bloom.assign((size_t)1 << (e->s5_bloom_lg - 5), 0u);
                auto set = [&](uint32_t h) { e->s5_bloom[h >> 5] |= 1u << (h & 31); };
                auto add = [&](const Scan2Slot& t) {
                    if ((t.len & kScan2LenMask) == 4) set(scan5_bloom_x(t.key, e->s5_bloom_lg));
                    else set(scan5_bloom_g(t.key, (t.front[0] >> 24) & 0xDFu, e->s5_bloom_lg));
                };
                for (const Scan2Slot& sl : e->s2.slots) {
                    if (sl.key == kScan2EmptyKey) continue;
                    if (!(sl.info & kScan2Multi)) { add(sl); continue; }
                    for (uint32_t j = 0; j < sl.len; j++) add(e->s2.more[(sl.info & ~kScan2Multi) + j]);
                }
            }
            e->use_scan5 = true;
            e->s5_short_groups = large5;
            e->use_scan2 = true;           // (the scan2 family's path through scan_pipeline; gft_scan2.hip itself runs only when s2.supported)
            if (!k2_fits) { e->scan2_k2_waves = kScan5Waves; e->scan2_cand_cap = e->s5plan.cand_cap; }   // (what that path sizes slabs by)
        }
    }
    // Kernel choice: the suffix-window kernel (scan2) where its direct tables apply -- small alphabets, the benchmark's
    // shape --, the stride-2 kernel (scan3: any alphabet, merged filter groups) everywhere else; the DFA kernel only as
    // a cross-check.  GFT_SCAN_KERNEL=scan2 / scan3 / dfa forces one (read here, i.e. by gft_build / gft_import_tables)
    const uint32_t bloom_lds_bytes = e->s3.supported && e->s3.bloom_lg <= kScan3BloomLdsLg ? 4u << e->s3.bloom_lg : 0u;
    const bool k3_fits = e->s3.supported && scan3_plan((uint32_t)e->s3.filter.size(), (uint32_t)e->s3.short3.size(), (uint32_t)e->s3.srec.size(),
                                                        bloom_lds_bytes, e->lds_max - 512, &e->scan3_waves, &e->scan3_cand_cap);
    const std::string forced = force ? force : "";
    e->use_scan3 = k3_fits && forced != "dfa" && forced != "scan2" && forced != "scan4" && (forced == "scan3" || !e->use_scan2);   // (scan5 asked for but not applicable: the stride-2 kernel, as by default)
    if (e->use_scan3) {
        if ((rc = upload(e, e->d_s3_filter, e->s3.filter, "table upload"))) return rc;
        s3v = e->s3.short3;
        if (s3v.empty()) s3v.assign(16, 0);
        if ((rc = upload(e, e->d_s3_short3, s3v, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_srec, e->s3.srec, "table upload"))) return rc;
        if (!e->s3.short3_big.empty() && (rc = upload(e, e->d_s3_short3_big, e->s3.short3_big, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_srec_big, e->s3.srec_big, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_bloom, e->s3.bloom, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_slots, e->s3.slots, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_more, e->s3.more, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_term_blob, e->s3.term_blob, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_term_off, e->s3.term_off, "table upload"))) return rc;
        g1.assign(e->s3.cls, e->s3.cls + 256); g2.assign(e->s3.cls_fold, e->s3.cls_fold + 256);
        if ((rc = upload(e, e->d_s3_cls, g1, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_cls_fold, g2, "table upload"))) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream), "table upload");      // g1 / g2 / s3v are locals
        if (e->opt_scan_dbg || getenv("GFT_SCAN_DEBUG"))
            fprintf(stderr, "[gft build debug] scan3: G=%u%s keys=%llu anchors=%llu slots=%zu more=%zu bloom 2^%u (%s) short cells: lds records %zu, big words %zu; waves=%u cand_cap=%u\n",
                    e->s3.G, e->s3.grouped ? " (merged classes)" : "", (unsigned long long)e->s3.n_keys, (unsigned long long)e->s3.n_anchors,
                    e->s3.slots.size(), e->s3.more.size(), e->s3.bloom_lg, bloom_lds_bytes ? "LDS" : "global", e->s3.srec.size() / kScan3RecWords - 1,
                    e->s3.srec_big.size(), e->scan3_waves, e->scan3_cand_cap);
    }
    if (e->use_scan2 && getenv("GFT_SCAN_DEBUG")) {
        size_t n_ff = 0, n_used = 0, n_simple = 0, n_slots = 0;
        for (uint8_t b : e->s2.fpt) { n_ff += b == 0xFF; n_used += b != 0; }
        for (const auto& s : e->s2.slots) { n_slots += s.key != kScan2EmptyKey; n_simple += s.key != kScan2EmptyKey && !(s.info & kScan2Multi); }
        fprintf(stderr, "[gft build debug] kp=%u keys=%zu (simple %zu) slots=%zu fpt: used=%zu always-pass=%zu of %u; shorts=%zu filter=%s %u bits waves=%u\n",
                e->s2.kp, n_slots, n_simple, e->s2.slots.size(), n_used, n_ff, (unsigned)e->s2.fpt.size(), e->s2.shorts.size() - 1,
                e->s2.hashed ? "hashed" : "direct", e->s2.filter_bits, e->scan2_k2_waves);
        fprintf(stderr, "[gft build debug] candidate list capacity %u per wave\n", e->scan2_cand_cap);
    }
    if (e->use_scan5 && e->s5_short_groups && !e->use_scan3) {
        // the short-term tables of the stride-2 kernel's set, which is not uploaded as a whole then
        s3v = e->s3.short3;
        if (s3v.empty()) s3v.assign(16, 0);
        if ((rc = upload(e, e->d_s3_short3, s3v, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_srec, e->s3.srec, "table upload"))) return rc;
        if (!e->s3.short3_big.empty() && (rc = upload(e, e->d_s3_short3_big, e->s3.short3_big, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_srec_big, e->s3.srec_big, "table upload"))) return rc;
        g1.assign(e->s3.cls, e->s3.cls + 256); g2.assign(e->s3.cls_fold, e->s3.cls_fold + 256);
        if ((rc = upload(e, e->d_s3_cls, g1, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s3_cls_fold, g2, "table upload"))) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream), "table upload");      // g1 / g2 / s3v are locals
    }
    if (e->use_scan2) {
        e->scan2_short3_bytes = (uint32_t)e->s2.short3.size();
        if (e->s2.short3.empty()) e->s2.short3.assign(16, 0);   // placeholder upload; short3_bytes stays 0
        if ((rc = upload(e, e->d_s2_short3, e->s2.short3, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_shorts_packed, e->s2.shorts_packed, "table upload"))) return rc;
        if (!e->s2.short3_big.empty() && (rc = upload(e, e->d_s2_short3_big, e->s2.short3_big, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_fpt, e->s2.fpt, "table upload"))) return rc;
        c1.assign(e->s2.cls, e->s2.cls + 256); c2.assign(e->s2.cls_fold, e->s2.cls_fold + 256);
        if ((rc = upload(e, e->d_s2_cls, c1, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_cls_fold, c2, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_filter, e->s2.filter, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_slots, e->s2.slots, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_more, e->s2.more, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_term_blob, e->s2.term_blob, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s2_term_off, e->s2.term_off, "table upload"))) return rc;
    }
    if (e->use_scan5) {
        s5g.assign(e->s5.grp, e->s5.grp + 256); s5gf.assign(e->s5.grp_fold, e->s5.grp_fold + 256);
        if ((rc = upload(e, e->d_s5_grp, s5g, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s5_grp_fold, s5gf, "table upload"))) return rc;
        if ((rc = upload(e, e->d_s5_filter, e->s5.filter, "table upload"))) return rc;
        if (e->s5_bloom_lg && (rc = upload(e, e->d_s5_bloom, e->s5_bloom, "table upload"))) return rc;
        if (getenv("GFT_SCAN_DEBUG")) {
            size_t set_bits = 0;
            for (uint32_t w : e->s5_bloom) set_bits += (size_t)__builtin_popcount(w);
            fprintf(stderr, "[gft build debug] scan5: G=%u of %u classes, filter %zu entries, list %u, term bits %u; Bloom level 2^%u bits, %.1f %% set\n",
                    e->s5.G, e->s2.kp, e->s5.filter.size(), e->s5plan.cand_cap, e->s5_term_bits, e->s5_bloom_lg,
                    e->s5_bloom_lg ? 100.0 * (double)set_bits / (double)(1ull << e->s5_bloom_lg) : 0.0);
        }
    }
    HIP_TRY(hipStreamSynchronize(e->stream), "table upload");
    e->built = true;
    e->scan_valid_docs = ~0ull;
    e->scan2_unit_max = kScan2UnitMax;
    e->have_programs = false;   // slots refer to the dictionary: programs must be set again
    e->n_exprs = 0;
    return GFT_OK;
}


int gft_build(gft_engine* e, const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, uint32_t flags) try {
    if (!e || (n_terms && (!terms_blob || !term_off))) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty() && !e->in_multi) return multi_build(e, terms_blob, term_off, n_terms, flags);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    std::vector<std::string> terms;
    terms.reserve(n_terms);
    for (uint32_t i = 0; i < n_terms; i++) {
        if (term_off[i + 1] < term_off[i]) return fail(e, GFT_E_INVALID, "term_off is not ascending");
        terms.emplace_back((const char*)terms_blob + term_off[i], (size_t)(term_off[i + 1] - term_off[i]));
    }
    e->built = false;
    build_ac_tables(std::move(terms), e->tab);
    build_scan2_tables(e->tab, e->s2);     // suffix-window tables (scan2, kept as a cross-check)
    build_scan3_tables(e->tab, e->s3);     // stride-2 suffix-window tables (the fast path)
    return install_tables(e, flags);
} GFT_CATCH((e ? &e->err : nullptr))

uint32_t gft_n_terms(const gft_engine* e) { return e ? (uint32_t)e->tab.terms.size() : 0; }
uint32_t gft_n_states(const gft_engine* e) { return e ? e->tab.n_states : 0; }
uint32_t gft_n_exprs(const gft_engine* e) { return e ? e->n_exprs : 0; }
uint32_t gft_n_host_exprs(const gft_engine* e) { return e ? (uint32_t)e->host_only.size() : 0; }
int gft_last_nonascii(const gft_engine* e) { return e && e->last_nonascii ? 1 : 0; }
const char* gft_build_info(void) { return kExtraKernels ? "gfx950 extra_kernels=1" : "gfx950 extra_kernels=0"; }
const char* gft_scan_kernel(const gft_engine* e) {
    if (!e || !e->built) return "";
    return e->use_scan3 ? "scan3" : e->use_scan4 ? "scan4" : e->use_scan5 ? "scan5" : e->use_scan2 ? "scan2" : "dfa";
}

int gft_term(const gft_engine* e, uint32_t term_id, const uint8_t** ptr, uint32_t* len) try {
    if (!e || !ptr || !len) return GFT_E_INVALID;
    if (term_id >= e->tab.terms.size()) return fail(e, GFT_E_INVALID, "term id out of range");
    *ptr = (const uint8_t*)e->tab.terms[term_id].data();
    *len = (uint32_t)e->tab.terms[term_id].size();
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int64_t gft_term_id(const gft_engine* e, const uint8_t* term, uint32_t len) try {
    if (!e) return -1;
    std::string s((const char*)term, len);
    auto it = std::lower_bound(e->tab.terms.begin(), e->tab.terms.end(), s);
    if (it == e->tab.terms.end() || *it != s) return -1;
    return (int64_t)(it - e->tab.terms.begin());
} GFT_CATCH_VALUE(-1)


int gft_export_tables(const gft_engine* e, uint8_t* out, uint64_t cap, uint64_t* needed) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    Writer w;
    w.u32(kTablesMagic); w.u32(kTablesVersion); w.u32((uint32_t)sizeof(Scan2Slot)); w.u32(kScan2FptSize); w.u32(e->build_flags);
    const AcTables& a = e->tab;
    w.u64(a.terms.size());
    for (const auto& t : a.terms) { w.u64(t.size()); w.raw(t.data(), t.size()); }
    w.u32(a.n_classes); w.raw(a.byte_class, 256); w.u32(a.n_states); w.u32(a.max_term_len);
    w.vec(a.delta); w.vec(a.out_term); w.vec(a.out_link); w.vec(a.term_len); w.vec(a.depth); w.vec(a.fail);
    w.vec(a.child_begin); w.vec(a.in_class);
    const Scan2Tables& t = e->s2;
    w.u32(t.supported ? 1 : 0); w.u32(t.kp); w.u32(t.pad_class); w.u32(t.hashed ? 1 : 0); w.u32(t.filter_bits); w.u32(t.hash_shift);
    w.vec(t.filter); w.vec(t.short3); w.vec(t.shorts); w.vec(t.short3_big); w.vec(t.shorts_packed); w.u32(t.fpt_lg); w.vec(t.fpt);
    w.u32(t.slot_shift); w.u32(t.slot_seed); w.vec(t.slots); w.vec(t.more);
    w.raw(t.cls, 256); w.raw(t.cls_fold, 256); w.vec(t.term_blob); w.vec(t.term_off); w.u64(t.n_keys);
    const Scan3Tables& u = e->s3;
    w.u32(u.supported ? 1 : 0); w.u32(u.G); w.u32(u.grouped ? 1 : 0); w.raw(u.cls, 256); w.raw(u.cls_fold, 256);
    w.vec(u.filter); w.vec(u.short3); w.vec(u.srec); w.vec(u.short3_big); w.vec(u.srec_big); w.u32(u.bloom_lg); w.vec(u.bloom);
    w.u32(u.slot_shift); w.u32(u.slot_seed); w.vec(u.slots); w.vec(u.more); w.vec(u.term_blob); w.vec(u.term_off);
    w.u64(u.n_keys); w.u64(u.n_anchors);
    uint64_t sum = 1469598103934665603ull;          // FNV-1a over everything before it
    for (uint8_t c : w.b) { sum ^= c; sum *= 1099511628211ull; }
    w.u64(sum);
    if (needed) *needed = w.b.size();
    if (!out || cap < w.b.size()) return GFT_E_INVALID;
    memcpy(out, w.b.data(), w.b.size());
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_import_tables(gft_engine* e, const uint8_t* blob, uint64_t len) try {
    if (!e || !blob) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty() && !e->in_multi) return multi_import_tables(e, blob, len);
    if (len < 28) return fail(e, GFT_E_INVALID, "table blob too short");
    uint64_t sum = 1469598103934665603ull, stored;
    for (uint64_t i = 0; i + 8 < len; i++) { sum ^= blob[i]; sum *= 1099511628211ull; }
    memcpy(&stored, blob + len - 8, 8);
    if (sum != stored) return fail(e, GFT_E_INVALID, "table blob is corrupt (checksum)");
    Reader r{blob, len - 8};
    if (r.u32() != kTablesMagic) return fail(e, GFT_E_INVALID, "not a gft table blob");
    if (r.u32() != kTablesVersion || r.u32() != sizeof(Scan2Slot) || r.u32() != kScan2FptSize)
        return fail(e, GFT_E_UNSUPPORTED, "table blob was written by another library version");
    const uint32_t flags = r.u32();
    AcTables a;
    const uint64_t nt = r.u64();
    if (!r.ok || nt > len) return fail(e, GFT_E_INVALID, "table blob is truncated");
    a.terms.resize((size_t)nt);
    for (auto& t : a.terms) {
        const uint64_t k = r.u64();
        if (!r.ok || k > r.n - r.i) return fail(e, GFT_E_INVALID, "table blob is truncated");
        t.assign((const char*)r.p + r.i, (size_t)k);
        r.i += k;
    }
    a.n_classes = r.u32(); r.raw(a.byte_class, 256); a.n_states = r.u32(); a.max_term_len = r.u32();
    r.vec(a.delta); r.vec(a.out_term); r.vec(a.out_link); r.vec(a.term_len); r.vec(a.depth); r.vec(a.fail);
    r.vec(a.child_begin); r.vec(a.in_class);
    Scan2Tables t;
    t.supported = r.u32() != 0; t.kp = r.u32(); t.pad_class = r.u32(); t.hashed = r.u32() != 0; t.filter_bits = r.u32(); t.hash_shift = r.u32();
    r.vec(t.filter); r.vec(t.short3); r.vec(t.shorts); r.vec(t.short3_big); r.vec(t.shorts_packed); t.fpt_lg = r.u32(); r.vec(t.fpt);
    t.slot_shift = r.u32(); t.slot_seed = r.u32(); r.vec(t.slots); r.vec(t.more);
    r.raw(t.cls, 256); r.raw(t.cls_fold, 256); r.vec(t.term_blob); r.vec(t.term_off); t.n_keys = r.u64();
    Scan3Tables u;
    u.supported = r.u32() != 0; u.G = r.u32(); u.grouped = r.u32() != 0; r.raw(u.cls, 256); r.raw(u.cls_fold, 256);
    r.vec(u.filter); r.vec(u.short3); r.vec(u.srec); r.vec(u.short3_big); r.vec(u.srec_big); u.bloom_lg = r.u32(); r.vec(u.bloom);
    u.slot_shift = r.u32(); u.slot_seed = r.u32(); r.vec(u.slots); r.vec(u.more); r.vec(u.term_blob); r.vec(u.term_off);
    u.n_keys = r.u64(); u.n_anchors = r.u64();
    if (!r.ok || r.i != r.n) return fail(e, GFT_E_INVALID, "table blob is truncated");
    // shape checks first: validate_tables indexes the tables by each other's sizes
    if (a.n_classes == 0 || a.n_classes > 256 || a.delta.size() != (size_t)a.n_states * a.n_classes || a.out_term.size() != a.n_states ||
        a.out_link.size() != a.n_states || a.term_len.size() != a.terms.size() ||
        (t.supported && (t.fpt_lg > 28 || t.fpt.size() != (t.fpt_lg ? (size_t)1 << t.fpt_lg : (size_t)kScan2FptSize) || t.slots.size() != ((size_t)1 << (32 - t.slot_shift)) || t.term_off.size() != a.terms.size() + 1 ||
                         t.filter.size() * 32 != t.filter_bits)))
        return fail(e, GFT_E_INVALID, "table blob is inconsistent");
    if (const char* why = validate_tables(a, t, u)) return fail(e, GFT_E_INVALID, std::string("table blob is inconsistent: ") + why);
    if (!t.supported) t.why_not = "not supported by the suffix-window kernel (imported tables)";
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    e->built = false;
    e->tab = std::move(a);
    e->s2 = std::move(t);
    e->s3 = std::move(u);
    // (a dictionary whose suffix-window set is not serialised as complete -- more than 32 byte classes -- gets its long-term
    // tables from the compiler again: a blob only ever holds what validate_tables checks)
    if (!e->s2.supported) build_scan2_tables(e->tab, e->s2);
    else e->s2.long_ok = true;
    return install_tables(e, flags);
} GFT_CATCH((e ? &e->err : nullptr))

int gft_scan_device(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                    uint32_t flags, gft_matches* out_dev) try {
    if (!e || !out_dev || (n_docs && (!d_text_blob || !d_doc_off))) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    DeviceGuard g(e->device);
    uint64_t nm = 0;
    int rc = scan_pipeline(e, d_text_blob, d_doc_off, n_docs, flags, true, &nm);
    if (rc) return rc;
    if ((flags & GFT_SCAN_UNIQUE) && (rc = unique_pipeline(e, n_docs, &nm))) return rc;
    if ((flags & GFT_POS_RUNES) && !(flags & GFT_SCAN_UNIQUE)) {
        if (e->build_flags & GFT_POS_END) return fail(e, GFT_E_UNSUPPORTED, "GFT_POS_RUNES needs a GFT_POS_START engine (AnknownEngine reports where a match begins)");
        if ((rc = rune_pipeline(e, d_text_blob, d_doc_off, n_docs, nm))) return rc;
    }
    if ((rc = refine_nonascii(e, d_text_blob, d_doc_off, n_docs, flags))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream), "scan pipeline");
    out_dev->n_docs = n_docs; out_dev->n_matches = nm;
    out_dev->match_off = e->d_match_off.as<uint64_t>();
    out_dev->term_id = e->d_term.as<uint32_t>();
    out_dev->pos = e->d_pos.as<uint32_t>();
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

constexpr size_t kPinBuf = 128u << 20;       // bytes per bounce buffer
// what goes through a buffer at once (GFT_HOST_CHUNK_MB, 4 .. 128; timing study)
static size_t pin_chunk() {
    static const size_t n = [] {
        size_t mb = 128;
        if (const char* e = getenv("GFT_HOST_CHUNK_MB")) { const long v = atol(e); if (v >= 4 && v <= 128) mb = (size_t)v; }
        return mb << 20;
    }();
    return n;
}
#define kPinChunk pin_chunk()
// copy threads that fill a bounce buffer: the link (PCIe Gen5 x16, 57 GB/s from pinned memory) is only kept busy when the
// host side copies faster than that -- four threads reach ~58 GB/s, eight 120 (tools/probe_pcie.py); more than eight only add
// wake-ups (12: 11.4-12.0 M documents/s on the 250 000-document batch, 8: 12.2-12.3).  GFT_HOST_THREADS overrides
static unsigned pin_threads() {
    static const unsigned n = [] {
        if (const char* e = getenv("GFT_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)std::min(v, 32); }
        const unsigned hc = std::thread::hardware_concurrency();
        return hc ? std::min(std::max(hc, 2u), 8u) : 4u;
    }();
    return n;
}

// pageable host memory -> device through the pinned bounce buffers
static int h2d_staged(gft_engine* e, void* dst, const void* src, size_t bytes) {
    if (bytes < (8u << 20)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream), "upload");
        return GFT_OK;
    }
    for (int k = 0; k < 2; k++) {
        if (!e->pin[k]) HIP_TRY(hipHostMalloc(&e->pin[k], kPinBuf, hipHostMallocDefault), "pinned alloc");
        if (!e->pin_ev[k]) HIP_TRY(hipEventCreateWithFlags(&e->pin_ev[k], hipEventDisableTiming), "event");
    }
    if (!e->copy_pool) e->copy_pool.reset(new gft::CopyPool(pin_threads() - 1));
    size_t done = 0;
    // (the first chunks are small and double: the link idles while the very first one is filled)
    size_t chunk = std::min<size_t>(kPinChunk, 4u << 20);
    for (int k = 0; done < bytes; k ^= 1, chunk = std::min(kPinChunk, chunk * 2)) {
        const size_t n = std::min(chunk, bytes - done);
        HIP_TRY(hipEventSynchronize(e->pin_ev[k]), "staging");        // the copy out of this buffer has finished
        e->copy_pool->copy(e->pin[k], (const uint8_t*)src + done, n);
        HIP_TRY(hipMemcpyAsync((uint8_t*)dst + done, e->pin[k], n, hipMemcpyHostToDevice, e->stream), "upload");
        HIP_TRY(hipEventRecord(e->pin_ev[k], e->stream), "staging");
        done += n;
    }
    return GFT_OK;
}

// device -> pageable host memory through the same bounce buffers (a copy straight into pageable memory is staged by the
// runtime through one thread); synchronous: returns when dst holds the bytes
static int d2h_staged(gft_engine* e, void* dst, const void* src, size_t bytes) {
    if (bytes < (8u << 20)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream), "download");
        HIP_TRY(hipStreamSynchronize(e->stream), "download");
        return GFT_OK;
    }
    for (int k = 0; k < 2; k++) {
        if (!e->pin[k]) HIP_TRY(hipHostMalloc(&e->pin[k], kPinBuf, hipHostMallocDefault), "pinned alloc");
        if (!e->pin_ev[k]) HIP_TRY(hipEventCreateWithFlags(&e->pin_ev[k], hipEventDisableTiming), "event");
    }
    if (!e->copy_pool) e->copy_pool.reset(new gft::CopyPool(pin_threads() - 1));
    size_t issued = 0, done = 0;
    size_t len[2] = {0, 0};
    int ki = 0, kd = 0;
    // chunk i + 1 is on the wire while chunk i is copied out of its buffer (a result smaller than two buffers goes in
    // quarters, so that there is a chunk i + 1)
    const size_t chunk = std::min(kPinChunk, std::max<size_t>(4u << 20, (bytes / 4 + 4095) & ~(size_t)4095));
    while (done < bytes) {
        while (issued < bytes && len[ki] == 0) {                  // (a buffer is free again once it has been copied out)
            const size_t n = std::min(chunk, bytes - issued);
            HIP_TRY(hipMemcpyAsync(e->pin[ki], (const uint8_t*)src + issued, n, hipMemcpyDeviceToHost, e->stream), "download");
            HIP_TRY(hipEventRecord(e->pin_ev[ki], e->stream), "staging");
            len[ki] = n; issued += n; ki ^= 1;
        }
        HIP_TRY(hipEventSynchronize(e->pin_ev[kd]), "staging");
        const size_t n = len[kd];
        e->copy_pool->copy((uint8_t*)dst + done, e->pin[kd], n);
        len[kd] = 0; done += n; kd ^= 1;
    }
    return GFT_OK;
}

static int stage_docs(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs) {
    const uint64_t bytes = n_docs ? doc_off[n_docs] : 0;
    HIP_TRY(e->d_text.ensure(bytes + 64), "text alloc");
    HIP_TRY(e->d_doc_off.ensure((n_docs + 1) * 8), "doc_off alloc");
    if (bytes) { int rc = h2d_staged(e, e->d_text.p, text_blob, bytes); if (rc) return rc; }
    if (n_docs) HIP_TRY(hipMemcpyAsync(e->d_doc_off.p, doc_off, (n_docs + 1) * 8, hipMemcpyHostToDevice, e->stream), "doc_off upload");
    return GFT_OK;
}

int gft_scan(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
             gft_matches* out) try {
    if (!e || !out || (n_docs && (!doc_off))) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty() && !e->in_multi && n_docs) return multi_scan(e, text_blob, doc_off, n_docs, flags, out);   // (an empty batch -- doc_off may be NULL -- is the first device's)
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    DeviceGuard g(e->device);
    SyncOnExit drained(e);      // host buffers are read by asynchronous copies: drained on every way out
    int rc = stage_docs(e, text_blob, doc_off, n_docs);
    if (rc) return rc;
    uint64_t nm = 0;
    rc = scan_pipeline(e, e->d_text.as<uint8_t>(), e->d_doc_off.as<uint64_t>(), n_docs, flags, true, &nm, doc_off);
    if (rc) return rc;
    if ((flags & GFT_SCAN_UNIQUE) && (rc = unique_pipeline(e, n_docs, &nm))) return rc;
    if ((flags & GFT_POS_RUNES) && !(flags & GFT_SCAN_UNIQUE)) {
        if (e->build_flags & GFT_POS_END) return fail(e, GFT_E_UNSUPPORTED, "GFT_POS_RUNES needs a GFT_POS_START engine (AnknownEngine reports where a match begins)");
        if ((rc = rune_pipeline(e, e->d_text.as<uint8_t>(), e->d_doc_off.as<uint64_t>(), n_docs, nm))) return rc;
    }
    if ((rc = refine_nonascii(e, e->d_text.as<uint8_t>(), e->d_doc_off.as<uint64_t>(), n_docs, flags))) return rc;
    e->h_match_off.assign(n_docs + 1, 0);
    e->h_term.assign(nm, 0);
    e->h_pos.assign(nm, 0);
    HIP_TRY(hipMemcpyAsync(e->h_match_off.data(), e->d_match_off.p, (n_docs + 1) * 8, hipMemcpyDeviceToHost, e->stream), "download");
    if (nm) {
        HIP_TRY(hipMemcpyAsync(e->h_term.data(), e->d_term.p, nm * 4, hipMemcpyDeviceToHost, e->stream), "download");
        HIP_TRY(hipMemcpyAsync(e->h_pos.data(), e->d_pos.p, nm * 4, hipMemcpyDeviceToHost, e->stream), "download");
    }
    HIP_TRY(hipStreamSynchronize(e->stream), "scan pipeline");
    out->n_docs = n_docs; out->n_matches = nm;
    out->match_off = e->h_match_off.data(); out->term_id = e->h_term.data(); out->pos = e->h_pos.data();
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_set_programs(gft_engine* e, const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs,
                     uint32_t n_extra) try {
    if (!e || (n_exprs && (!prog_words || !prog_off))) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty() && !e->in_multi) return multi_set_programs(e, prog_words, prog_off, n_exprs, n_extra);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    const uint32_t n_slots = (uint32_t)e->tab.terms.size() + n_extra;
    // (slot n_slots itself is the solver's never-present slot: it must fit a program word's field too)
    if (n_slots > GFT_SLOT_MASK || n_slots >= (1u << kDwFieldBits)) return fail(e, GFT_E_UNSUPPORTED, "too many slots");
    std::vector<ProgramTraits> traits(n_exprs);
    for (uint32_t i = 0; i < n_exprs; i++) {
        if (prog_off[i + 1] < prog_off[i]) return fail(e, GFT_E_INVALID, "prog_off is not ascending");
        int rc = check_program(e, prog_words + prog_off[i], prog_off[i + 1] - prog_off[i], n_slots, i, &traits[i]);
        if (rc) return rc;
    }
    DeviceGuard g(e->device);
    SyncOnExit drained(e);      // host buffers are read by asynchronous copies: drained on every way out
    std::vector<uint32_t> w(prog_words, prog_words + (n_exprs ? prog_off[n_exprs] : 0));
    std::vector<uint64_t> o(prog_off, prog_off + (n_exprs ? n_exprs + 1 : 0));
    if (o.empty()) o.push_back(0);
    int rc;
    if ((rc = upload(e, e->d_prog, w, "program upload"))) return rc;
    if ((rc = upload(e, e->d_prog_off, o, "program upload"))) return rc;
    std::vector<uint32_t> fw, groups, fdepth;
    std::vector<uint64_t> fo(1, 0);
    for (uint32_t i = 0; i < n_exprs; i++) {
        if (traits[i].over_limit || traits[i].wide_pairs) {
            // beyond a limit of the device solver: the device evaluates a stand-in (one leaf on the never-present slot), the
            // expression itself is solved on the host from the scan's matches (host_solve.hpp) and its bit patched in.
            // An expression with a WIDE INORD group gets the same stand-in in the fused form: the solver's second phase
            // (gft_solve.hip wide_expr_doc) answers it from its public words, a document per wave
            const uint32_t stub = GFT_OP_UNIT << 28 | n_slots;
            fdepth.push_back(fuse_program(&stub, 1, 0, fw, groups));
        } else {
            const size_t fw0 = fw.size(), g0 = groups.size();
            uint32_t depth = fuse_program(prog_words + prog_off[i], prog_off[i + 1] - prog_off[i], prog_off[i], fw, groups);
            if (depth > kMaxBoolDepth) {
                // the fused form still nests deeper than the interpreter's stack (a balanced tree of 2^128 sub-trees would):
                // the host's
                fw.resize(fw0); groups.resize(g0);
                traits[i].over_limit = true;
                const uint32_t stub = GFT_OP_UNIT << 28 | n_slots;
                depth = fuse_program(&stub, 1, 0, fw, groups);
            }
            fdepth.push_back(depth);
        }
        while (fw.size() % 4) fw.push_back((uint32_t)kFopNop << 28);       // the interpreter reads 4-word chunks
        fo.push_back(fw.size());
    }
    // Evaluation order: inside every output tile (kSolveTileWords * 32 expressions) the programs are sorted by the
    // interpreter they need -- 2: nest deeper than its register stack, 1: use the stack, 0: flat (no push / pop at all,
    // half the work per word) -- and by length, and handed to the waves 64 at a time, so that the lanes of a wave run
    // loops of similar length on the cheapest interpreter that serves them all (longest first inside classes 2 and 1,
    // shortest first inside class 0: the block on the border mixes short programs of both).
    // order[i] = expression evaluated at sorted position i; blk_class[b] = the interpreter of block b.
    std::vector<uint32_t> order(n_exprs), blk_class, fprog_t, fblk_off, wave_blk;
    for (uint32_t i = 0; i < n_exprs; i++) order[i] = i;
    auto klass = [&](uint32_t x) { return fdepth[x] > kSolveRegStack ? 2u : fdepth[x] > 0 ? 1u : 0u; };
    auto plen = [&](uint32_t x) { return fo[x + 1] - fo[x]; };
    const uint32_t tile_exprs = kSolveTileWords * 32;
    constexpr uint32_t kWaves = kSolveBlockThreads / 64;
    for (uint32_t t0 = 0; t0 < n_exprs; t0 += tile_exprs) {
        const uint32_t t1 = std::min(n_exprs, t0 + tile_exprs);
        std::stable_sort(order.begin() + t0, order.begin() + t1, [&](uint32_t a, uint32_t b) {
            if (klass(a) != klass(b)) return klass(a) > klass(b);
            return klass(a) ? plen(a) > plen(b) : plen(a) < plen(b);
        });
        std::vector<uint64_t> cost;              // VALU work of a block, for the deal below
        for (uint32_t b0 = t0; b0 < t1; b0 += 64) {
            uint32_t cls = 0;
            uint64_t maxlen = 0;
            for (uint32_t i = b0; i < std::min(t1, b0 + 64); i++) {
                cls = std::max(cls, klass(order[i]));
                maxlen = std::max(maxlen, plen(order[i]));
            }
            blk_class.push_back(cls);
            cost.push_back(maxlen * (cls == 2 ? 40 : cls == 1 ? 26 : 14) + 160);
            // the block's chunks transposed: words 4c..4c+3 of lane l at off + (c * 64 + l) * 4
            if (fprog_t.size() + maxlen * 64 > 0xFFFFFFFFull) return fail(e, GFT_E_UNSUPPORTED, "program set too large");
            fblk_off.push_back((uint32_t)fprog_t.size());
            fprog_t.resize(fprog_t.size() + maxlen * 64, kDwNop);
            for (uint32_t i = b0; i < std::min(t1, b0 + 64); i++) {
                const uint64_t p0 = fo[order[i]], len = fo[order[i] + 1] - p0;
                for (uint64_t pc = 0; pc < len; pc++)
                    fprog_t[fblk_off.back() + ((pc / 4) * 64 + (i - b0)) * 4 + pc % 4] = fused_to_device(fw[p0 + pc]);
            }
        }
        // The deal: the tile's blocks go to the workgroup's waves sixteen at a time.  Wave w runs on SIMD w % 4 and the
        // four waves of a SIMD share its issue slots, so every round's blocks are dealt by cost, the most expensive
        // first, to the SIMD with the least work so far that still has a wave free (its lowest wave: the oldest wave of
        // a SIMD is served first, which suits the block everybody else ends up waiting for).
        // wave_blk[tile's first block + round * 16 + wave] = block (relative to the tile) or ~0.
        // (a full tile is 32 blocks = two rounds, so a tile's entries start at its first block's index)
        const uint32_t nblk = (uint32_t)cost.size();
        std::vector<uint32_t> by_cost(nblk);
        for (uint32_t b = 0; b < nblk; b++) by_cost[b] = b;
        for (uint32_t r0 = 0; r0 < nblk; r0 += kWaves) {
            const uint32_t r1 = std::min(nblk, r0 + kWaves);
            std::stable_sort(by_cost.begin() + r0, by_cost.begin() + r1, [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
            uint64_t load[4] = {0, 0, 0, 0};
            uint32_t used[4] = {0, 0, 0, 0};
            uint32_t deal[kWaves];
            for (uint32_t w = 0; w < kWaves; w++) deal[w] = 0xFFFFFFFFu;
            for (uint32_t k = r0; k < r1; k++) {
                int best = -1;
                for (int q = 0; q < 4; q++)
                    if (used[q] < kWaves / 4 && (best < 0 || load[q] < load[best])) best = q;
                deal[used[best] * 4 + best] = by_cost[k];
                used[best]++;
                load[best] += cost[by_cost[k]];
            }
            for (uint32_t w = 0; w < kWaves; w++) wave_blk.push_back(deal[w]);
        }
    }
    if (order.empty()) order.push_back(0);
    if (blk_class.empty()) blk_class.push_back(0);
    if (wave_blk.empty()) wave_blk.push_back(0xFFFFFFFFu);
    if (fblk_off.empty()) fblk_off.push_back(0);
    if (fprog_t.empty()) fprog_t.push_back(0);
    refresh_options(e);
    if (e->opt_solve_dbg) {
        uint64_t hist[16] = {0}, with_rare = 0, maxlen = 0;
        for (uint32_t i = 0; i < n_exprs; i++) {
            bool rare = false;
            for (uint64_t k = fo[i]; k < fo[i + 1]; k++) { hist[fw[k] >> 28]++; rare |= (fw[k] >> 28) >= kFopAndPop; }
            with_rare += rare;
            maxlen = std::max<uint64_t>(maxlen, fo[i + 1] - fo[i]);
        }
        fprintf(stderr, "[gft solve debug] %u programs, %zu fused words (max %llu); programs with stack/not/inord ops: %llu; ops:",
                n_exprs, fw.size(), (unsigned long long)maxlen, (unsigned long long)with_rare);
        for (int k = 1; k <= 15; k++) fprintf(stderr, " %d:%llu", k, (unsigned long long)hist[k]);
        fprintf(stderr, "\n");
    }
    if (groups.size() / 2 > (1u << kDwFieldBits)) return fail(e, GFT_E_UNSUPPORTED, "too many INORD groups");
    // the kernel reads control bits, not opcodes (gft_kernels.hpp fused_to_device)
    std::vector<uint32_t> dw(fw.size());
    for (size_t i = 0; i < fw.size(); i++) dw[i] = fused_to_device(fw[i]);
    if (dw.empty()) dw.push_back(kDwNop);
    if (groups.empty()) groups.assign(2, 0);
    if ((rc = upload(e, e->d_fprog, dw, "program upload"))) return rc;
    if ((rc = upload(e, e->d_fprog_off, fo, "program upload"))) return rc;
    if ((rc = upload(e, e->d_groups, groups, "program upload"))) return rc;
    if ((rc = upload(e, e->d_order, order, "program upload"))) return rc;
    if ((rc = upload(e, e->d_blk_class, blk_class, "program upload"))) return rc;
    if ((rc = upload(e, e->d_wave_blk, wave_blk, "program upload"))) return rc;
    if ((rc = upload(e, e->d_fprog_t, fprog_t, "program upload"))) return rc;
    if ((rc = upload(e, e->d_fblk_off, fblk_off, "program upload"))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream), "program upload");
    e->n_exprs = n_exprs; e->n_extra = n_extra; e->have_programs = true;
    e->scan_valid_docs = ~0ull;          // positions may not have been written for the old program set
    e->csr_valid = false;
    // what the host may have to solve (host_solve.hpp)
    e->h_prog.swap(w); e->h_prog_off.swap(o);
    e->host_only.clear(); e->inord_exprs.clear();
    e->inord_slot.assign((size_t)n_slots + 1, 0);
    for (uint32_t i = 0; i < n_exprs; i++) {
        if (traits[i].over_limit) e->host_only.push_back(i);
        else if (!traits[i].inord_slots.empty()) {
            e->inord_exprs.push_back(i);
            for (uint32_t sl : traits[i].inord_slots) e->inord_slot[sl] = 1;
        }
    }
    e->wide_pairs = 0;
    std::vector<uint32_t> wide_list;                     // per wide expression: index, offset and length of its public words
    for (uint32_t i = 0; i < n_exprs; i++)
        if (!traits[i].over_limit && traits[i].wide_pairs) {
            e->wide_pairs = std::max(e->wide_pairs, traits[i].wide_pairs);
            if (prog_off[i + 1] > 0xFFFFFFFFull) return fail(e, GFT_E_UNSUPPORTED, "program set too large");
            wide_list.push_back(i); wide_list.push_back((uint32_t)prog_off[i]); wide_list.push_back((uint32_t)(prog_off[i + 1] - prog_off[i]));
        }
    e->n_wide = (uint32_t)(wide_list.size() / 3);
    if (e->n_wide) {
        if ((rc = upload(e, e->d_wide_list, wide_list, "program upload"))) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream), "program upload");        // (wide_list is a local)
    }
    e->traits.swap(traits);
    e->fprog_words = (uint32_t)fw.size();
    e->n_inord_groups = e->n_rare_words = 0;
    for (uint32_t w : fw) {
        e->n_inord_groups += (w >> 28) == kFopInord;
        e->n_rare_words += (w >> 28) == kFopInord || (w >> 28) == kFopNot;
    }
    e->n_inord_groups += e->n_wide;                      // (their groups read positions too: the scan must write them)
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_process_device(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                       uint32_t flags, const gft_extra_matches* d_extra, uint32_t* d_hit_bitmap) try {
    if (!e || (n_docs && (!d_text_blob || !d_doc_off))) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    if (!e->have_programs) return fail(e, GFT_E_NOT_BUILT, "gft_set_programs has not been called");
    if (n_docs && e->n_exprs && !d_hit_bitmap) return fail(e, GFT_E_INVALID, "null bitmap");
    if (e->pend_count) return fail(e, GFT_E_INVALID, "batches of gft_process_device_begin are in flight: gft_process_device_end first");
    DeviceGuard g(e->device);
    // what the host solves (normally nothing): expressions beyond the device solver's limits, and INORD expressions in
    // documents where the caller's matches make a slot's list non-ascending -- for that the caller's (device) arrays are
    // read back, but only when some INORD group could be affected at all
    HostPlan plan;
    std::vector<uint64_t> xo;
    std::vector<uint32_t> xs, xp;
    gft_extra_matches hx{nullptr, nullptr, nullptr};
    const bool want_hx = d_extra && d_extra->off && n_docs && (!e->host_only.empty() || !e->inord_exprs.empty());
    if (want_hx) {
        xo.resize(n_docs + 1);
        HIP_TRY(hipMemcpy(xo.data(), d_extra->off, (n_docs + 1) * 8, hipMemcpyDeviceToHost), "extra read-back");
        // (the host walks these arrays now, not only the kernel: offsets must ascend, slots must exist -- what upload_extra
        // checks for host arrays)
        for (uint64_t d = 0; d < n_docs; d++)
            if (xo[d] > xo[d + 1]) return fail(e, GFT_E_INVALID, "extra offsets are not ascending");
        if (xo[n_docs] > (1ull << 40)) return fail(e, GFT_E_INVALID, "extra offsets are out of range");
        xs.resize(xo[n_docs] + 1); xp.resize(xo[n_docs] + 1);
        if (xo[n_docs]) {
            HIP_TRY(hipMemcpy(xs.data(), d_extra->slot, xo[n_docs] * 4, hipMemcpyDeviceToHost), "extra read-back");
            HIP_TRY(hipMemcpy(xp.data(), d_extra->pos, xo[n_docs] * 4, hipMemcpyDeviceToHost), "extra read-back");
            const uint64_t n_slots = e->tab.terms.size() + e->n_extra;
            for (uint64_t i = xo[0]; i < xo[n_docs]; i++)
                if (xs[i] >= n_slots) return fail(e, GFT_E_INVALID, "extra slot out of range");
        }
        hx.off = xo.data(); hx.slot = xs.data(); hx.pos = xp.data();
    }
    plan_host(e, want_hx ? &hx : nullptr, n_docs, plan);
    // units -> scan -> solve without a host round trip in between; one read-back at the end, and a second pass only when
    // this batch outgrew the unit table or the match pool the previous ones left behind
    for (int pass = 0; pass < 2; pass++) {
        uint64_t nm = 0;
        int rc = scan_pipeline(e, d_text_blob, d_doc_off, n_docs, flags, plan.all_docs, &nm, nullptr, pass == 0);
        if (rc) return rc;
        const bool was_deferred = e->deferred;
        rc = solve_pipeline(e, n_docs, d_extra, d_hit_bitmap);
        if (rc) return rc;
        if (!was_deferred) break;
        bool again = false;
        rc = deferred_check(e, &again);
        if (rc) return rc;
        if (!again) break;
    }
    HIP_TRY(hipStreamSynchronize(e->stream), "process pipeline");
    int rc = refine_nonascii(e, d_text_blob, d_doc_off, n_docs, flags);
    if (rc) return rc;
    return host_eval(e, want_hx ? &hx : nullptr, n_docs, plan, nullptr, d_hit_bitmap);
} GFT_CATCH((e ? &e->err : nullptr))

// ---- two batches in flight (at 125 000 documents -- one GPU's share of 1 M over 8 -- a step is 0.5 ms of
// kernels, and the read-back of the control block plus the launches of the next step are a tenth of it) -----------------------
int gft_process_device_begin(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                             uint32_t flags, const gft_extra_matches* d_extra, uint32_t* d_hit_bitmap) try {
    if (!e || (n_docs && (!d_text_blob || !d_doc_off))) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty()) return fail(e, GFT_E_UNSUPPORTED, "gft_process_device_begin: single-device handles only");
    if (e->pend_count == 2) return fail(e, GFT_E_INVALID, "gft_process_device_begin: two batches are in flight already (gft_process_device_end first)");
    const unsigned k = (e->pend_head + e->pend_count) % 2;
    gft_engine::Pending& pb = e->pend[k];
    pb.done = false; pb.rc = GFT_OK;
    pb.d_text = d_text_blob; pb.d_doc_off = d_doc_off; pb.n_docs = n_docs; pb.flags = flags; pb.d_bitmap = d_hit_bitmap;
    auto sync_path = [&]() {
        // (whatever cannot be deferred -- caller-supplied matches, host-solved expressions, the first batches of an engine,
        // an empty batch -- completes here; _end then only hands its status back.  Batches before it are finished first: the
        // synchronous path reuses the control block they are still to read)
        int rc = GFT_OK;
        while (e->pend_count && rc == GFT_OK) {
            // (statuses of the older batches stay theirs: they are completed, not consumed)
            gft_engine::Pending& o = e->pend[e->pend_head];
            if (!o.done) {
                if (hipEventSynchronize(o.ev) != hipSuccess) { o.rc = fail(e, GFT_E_HIP, "event wait"); o.done = true; break; }
                bool again = false;
                o.rc = deferred_interpret(e, o.rb, DeferredLaunch{o.single, o.n_docs_cap, o.unit_cap, o.static_slabs, o.single ? o.epoch : 0u}, &again);
                if (!o.rc && again) { e->pend_count = 0; o.rc = gft_process_device(e, o.d_text, o.d_doc_off, o.n_docs, o.flags, nullptr, o.d_bitmap); e->pend_count = 1; }
                else if (!o.rc) { DeviceGuard g2(e->device); o.rc = refine_nonascii(e, o.d_text, o.d_doc_off, o.n_docs, o.flags); }
                o.done = true;
            }
            break;                                              // (at most one older batch: k is the second slot then)
        }
        const unsigned keep_head = e->pend_head, keep_count = e->pend_count;
        e->pend_count = 0;                                      // (the synchronous entry point refuses to run beside batches in flight)
        pb.rc = gft_process_device(e, d_text_blob, d_doc_off, n_docs, flags, d_extra, d_hit_bitmap);
        e->pend_head = keep_head; e->pend_count = keep_count;
        pb.done = true;
    };
    const bool simple = n_docs && e->device >= 0 && e->built && e->have_programs && !(d_extra && d_extra->off) && e->host_only.empty() &&
                        (!e->n_exprs || d_hit_bitmap);
    if (!simple) { sync_path(); e->pend_count++; return GFT_OK; }
    DeviceGuard g(e->device);
    uint64_t nm = 0;
    int rc = scan_pipeline(e, d_text_blob, d_doc_off, n_docs, flags, false, &nm, nullptr, true);
    if (rc) { pb.rc = rc; pb.done = true; e->pend_count++; return GFT_OK; }
    const bool was_deferred = e->deferred;
    rc = solve_pipeline(e, n_docs, nullptr, d_hit_bitmap);
    if (rc || !was_deferred) {
        // (sizes were not known yet: this batch ran with its own synchronisations, like gft_process_device's first pass)
        if (!rc) { rc = hipStreamSynchronize(e->stream) == hipSuccess ? GFT_OK : fail(e, GFT_E_HIP, "process pipeline"); }
        if (!rc) rc = refine_nonascii(e, d_text_blob, d_doc_off, n_docs, flags);
        pb.rc = rc; pb.done = true; e->pend_count++;
        return GFT_OK;
    }
    e->deferred = false;
    if (!pb.rb) HIP_TRY(hipHostMalloc((void**)&pb.rb, 64, hipHostMallocDefault), "pinned alloc");
    if (!pb.ev) HIP_TRY(hipEventCreateWithFlags(&pb.ev, hipEventDisableTiming), "event");
    pb.single = e->deferred_single; pb.epoch = e->deferred_epoch; pb.n_docs_cap = e->deferred_n_docs; pb.unit_cap = e->deferred_unit_cap; pb.static_slabs = e->last_static_slabs;
    HIP_TRY(hipMemcpyAsync(pb.rb, e->d_ctl.p, 7 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream), "readback");
    HIP_TRY(hipEventRecord(pb.ev, e->stream), "event");
    e->pend_count++;
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_process_device_end(gft_engine* e) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->pend_count) return fail(e, GFT_E_INVALID, "gft_process_device_end: no batch in flight");
    gft_engine::Pending& pb = e->pend[e->pend_head];
    auto pop = [&]() { e->pend_head = (e->pend_head + 1) % 2; e->pend_count--; };
    if (pb.done) { const int rc = pb.rc; pop(); return rc; }
    DeviceGuard g(e->device);
    if (hipEventSynchronize(pb.ev) != hipSuccess) { pop(); return fail(e, GFT_E_HIP, "event wait"); }
    bool again = false;
    int rc = deferred_interpret(e, pb.rb, DeferredLaunch{pb.single, pb.n_docs_cap, pb.unit_cap, pb.static_slabs, pb.single ? pb.epoch : 0u}, &again);
    if (!rc && again) {
        // this batch outgrew the unit table or the match pool (both have been grown): once more, with its own synchronisations.
        // A younger batch in flight is behind it on the stream; it keeps its own bitmap and its own verdict.
        const unsigned keep_head = e->pend_head, keep_count = e->pend_count;
        gft_engine::Pending* young = keep_count == 2 ? &e->pend[(keep_head + 1) % 2] : nullptr;
        if (young && !young->done) {
            // (its read-back must be taken before the control block is used again)
            if (hipEventSynchronize(young->ev) != hipSuccess) { pop(); return fail(e, GFT_E_HIP, "event wait"); }
        }
        e->pend_count = 0;
        rc = gft_process_device(e, pb.d_text, pb.d_doc_off, pb.n_docs, pb.flags, nullptr, pb.d_bitmap);
        e->pend_head = keep_head; e->pend_count = keep_count;
    } else if (!rc) {
        rc = refine_nonascii(e, pb.d_text, pb.d_doc_off, pb.n_docs, pb.flags);
    }
    pop();
    return rc;
} GFT_CATCH((e ? &e->err : nullptr))

namespace {
// caller-supplied matches (host arrays) -> device copies; pdx stays null when there are none
int upload_extra(gft_engine* e, const gft_extra_matches* extra, uint64_t n_docs, gft_extra_matches& dx, const gft_extra_matches*& pdx) {
    pdx = nullptr;
    if (!(extra && extra->off && n_docs)) return GFT_OK;
    const uint64_t nx = extra->off[n_docs];
    HIP_TRY(e->d_xoff.ensure((n_docs + 1) * 8), "extra alloc");
    HIP_TRY(e->d_xslot.ensure(std::max<uint64_t>(nx, 1) * 4), "extra alloc");
    HIP_TRY(e->d_xpos.ensure(std::max<uint64_t>(nx, 1) * 4), "extra alloc");
    HIP_TRY(hipMemcpyAsync(e->d_xoff.p, extra->off, (n_docs + 1) * 8, hipMemcpyHostToDevice, e->stream), "extra upload");
    if (nx) {
        for (uint64_t i = 0; i < nx; i++)
            if (extra->slot[i] >= e->tab.terms.size() + e->n_extra) return fail(e, GFT_E_INVALID, "extra slot out of range");
        HIP_TRY(hipMemcpyAsync(e->d_xslot.p, extra->slot, nx * 4, hipMemcpyHostToDevice, e->stream), "extra upload");
        HIP_TRY(hipMemcpyAsync(e->d_xpos.p, extra->pos, nx * 4, hipMemcpyHostToDevice, e->stream), "extra upload");
    }
    dx.off = e->d_xoff.as<uint64_t>(); dx.slot = e->d_xslot.as<uint32_t>(); dx.pos = e->d_xpos.as<uint32_t>();
    pdx = &dx;
    return GFT_OK;
}
}  // namespace

int gft_process_again(gft_engine* e, uint64_t n_docs, const gft_extra_matches* extra, uint32_t* hit_bitmap) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty() && !e->in_multi) return multi_process_again(e, n_docs, extra, hit_bitmap);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (!e->built || !e->have_programs) return fail(e, GFT_E_NOT_BUILT, "engine not ready");
    if (e->scan_valid_docs != n_docs || !n_docs) return fail(e, GFT_E_INVALID, "gft_process_again: no scan of these documents to reuse");
    DeviceGuard g(e->device);
    SyncOnExit drained(e);      // host buffers are read by asynchronous copies: drained on every way out
    gft_extra_matches dx;
    const gft_extra_matches* pdx = nullptr;
    int rc = upload_extra(e, extra, n_docs, dx, pdx);
    if (rc) return rc;
    const uint64_t words = (e->n_exprs + 31) / 32;
    HostPlan plan;
    plan_host(e, pdx ? extra : nullptr, n_docs, plan);
    rc = solve_pipeline(e, n_docs, pdx, e->d_bitmap.as<uint32_t>());
    if (rc) return rc;
    if (n_docs * words) {
        if (!hit_bitmap) return fail(e, GFT_E_INVALID, "null bitmap");
        HIP_TRY(hipMemcpyAsync(hit_bitmap, e->d_bitmap.p, n_docs * words * 4, hipMemcpyDeviceToHost, e->stream), "download");
    }
    HIP_TRY(hipStreamSynchronize(e->stream), "solve pipeline");
    return host_eval(e, pdx ? extra : nullptr, n_docs, plan, hit_bitmap, nullptr);
} GFT_CATCH((e ? &e->err : nullptr))

int gft_process(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                const gft_extra_matches* extra, uint32_t* hit_bitmap) try {
    if (!e || (n_docs && !doc_off)) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->peers.empty() && !e->in_multi && n_docs) return multi_process(e, text_blob, doc_off, n_docs, flags, extra, hit_bitmap);
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    if (!e->have_programs) return fail(e, GFT_E_NOT_BUILT, "gft_set_programs has not been called");
    DeviceGuard g(e->device);
    SyncOnExit drained(e);      // host buffers are read by asynchronous copies: drained on every way out
    // GFT_HOST_TIMING=1: where a call from host memory spends its time (stderr; tools/bench_latency.py)
    static const bool timing = getenv("GFT_HOST_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(e->stream);
        fprintf(stderr, "[gft host timing] %s: %.2f ms since the call began\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    int rc = stage_docs(e, text_blob, doc_off, n_docs);
    if (rc) return rc;
    since("text and offsets uploaded");
    gft_extra_matches dx;
    const gft_extra_matches* pdx = nullptr;
    rc = upload_extra(e, extra, n_docs, dx, pdx);
    if (rc) return rc;
    const uint64_t words = (e->n_exprs + 31) / 32;
    HIP_TRY(e->d_bitmap.ensure(std::max<uint64_t>(n_docs * words, 1) * 4), "bitmap alloc");
    HostPlan plan;                        // the (expression, document) pairs the host solves (normally none)
    plan_host(e, pdx ? extra : nullptr, n_docs, plan);
    uint64_t nm = 0;
    // (expressions beyond the device solver's limits need every match with its position: the scan then leaves the CSR too)
    rc = scan_pipeline(e, e->d_text.as<uint8_t>(), e->d_doc_off.as<uint64_t>(), n_docs, flags, plan.all_docs, &nm, doc_off);
    if (rc) return rc;
    e->scan_valid_docs = n_docs;          // gft_process_again may reuse this scan
    rc = solve_pipeline(e, n_docs, pdx, e->d_bitmap.as<uint32_t>());
    if (rc) return rc;
    if ((rc = refine_nonascii(e, e->d_text.as<uint8_t>(), e->d_doc_off.as<uint64_t>(), n_docs, flags))) return rc;
    since("scanned and solved");
    if (n_docs * words) {
        if (!hit_bitmap) return fail(e, GFT_E_INVALID, "null bitmap");
        if ((rc = d2h_staged(e, hit_bitmap, e->d_bitmap.p, n_docs * words * 4))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->stream), "process pipeline");
    since("bitmap downloaded");
    return host_eval(e, pdx ? extra : nullptr, n_docs, plan, hit_bitmap, nullptr);
} GFT_CATCH((e ? &e->err : nullptr))

int gft_debug_emulate_scan(const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, const uint8_t* text,
                           uint32_t len, uint32_t lo, uint32_t flags, uint32_t scan_flags, uint32_t* out_term,
                           uint32_t* out_pos, uint64_t cap, uint64_t* needed) try {
    if ((n_terms && (!terms_blob || !term_off)) || (len && !text) || lo > len || !needed) return GFT_E_INVALID;
    std::vector<std::string> terms;
    for (uint32_t i = 0; i < n_terms; i++) terms.emplace_back((const char*)terms_blob + term_off[i], (size_t)(term_off[i + 1] - term_off[i]));
    AcTables tab;
    build_ac_tables(std::move(terms), tab);
    Scan3Tables t;
    build_scan3_tables(tab, t);
    if (!t.supported) return GFT_E_UNSUPPORTED;
    std::vector<Scan3Hit> hits;
    scan3_emulate(t, text, len, lo, (scan_flags & GFT_FOLD_ASCII) != 0, (flags & GFT_POS_END) != 0, hits);
    *needed = hits.size();
    if (hits.size() > cap || (hits.size() && (!out_term || !out_pos))) return GFT_E_INVALID;
    for (size_t i = 0; i < hits.size(); i++) { out_term[i] = hits[i].term; out_pos[i] = hits[i].pos; }
    return GFT_OK;
} GFT_CATCH(nullptr)

int gft_debug_scan5_filter(const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, const uint8_t* text, uint32_t len,
                           uint32_t lane_start, uint32_t scan_flags, uint32_t groups, uint8_t* out_exact, uint8_t* out_dual,
                           uint32_t* groups_used) try {
    if ((n_terms && (!terms_blob || !term_off)) || (len && (!text || !out_exact || !out_dual)) || lane_start > len) return GFT_E_INVALID;
    std::vector<std::string> terms;
    for (uint32_t i = 0; i < n_terms; i++) terms.emplace_back((const char*)terms_blob + term_off[i], (size_t)(term_off[i + 1] - term_off[i]));
    AcTables tab;
    build_ac_tables(std::move(terms), tab);
    Scan2Tables s2;
    build_scan2_tables(tab, s2);
    if (!s2.long_ok) return GFT_E_UNSUPPORTED;
    Scan5Tables s5;
    // (a filter word has one bit per group: 32 at most, whatever the caller asks for; the kernel's plan stops at kScan5MaxGroups)
    build_scan5_tables(tab, s2, std::min<uint32_t>(groups && groups < s2.kp ? groups : s2.kp, 32u), s5);
    if (groups_used) *groups_used = s5.G;
    const bool fold = (scan_flags & GFT_FOLD_ASCII) != 0;
    const uint8_t* cls = fold ? s2.cls_fold : s2.cls;
    const uint8_t* grp = fold ? s5.grp_fold : s5.grp;
    const uint32_t kp = s2.kp, G = s5.G;
    // the exact filter, from first principles (whatever the alphabet): the 4-window of exact classes that ends at i is the
    // anchor window of a long term (= a key of the bucket table), or a term of length <= 3 ends at i; the pad class stands in
    // front of the document
    auto cl = [&](int64_t i) { return i < 0 ? s2.pad_class : (uint32_t)cls[text[i]]; };
    auto gr = [&](int64_t i) { return i < 0 ? s5.pad_group : (uint32_t)grp[text[i]]; };
    std::unordered_set<uint32_t> keys;
    for (const Scan2Slot& sl : s2.slots) if (sl.key != kScan2EmptyKey) keys.insert(sl.key);
    std::vector<std::vector<uint32_t>> shorts;
    for (const auto& term : tab.terms)
        if (!term.empty() && term.size() < 4) {
            std::vector<uint32_t> v;
            for (unsigned char ch : term) v.push_back(tab.byte_class[ch]);
            shorts.push_back(v);
        }
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t key = (uint32_t)((((uint64_t)cl((int64_t)i - 3) * kp + cl((int64_t)i - 2)) * kp + cl((int64_t)i - 1)) * kp + cl(i));
        bool f = keys.count(key) != 0;
        for (size_t k = 0; k < shorts.size() && !f; k++) {
            const auto& v = shorts[k];
            bool eq = true;
            for (size_t j = 0; j < v.size() && eq; j++) eq = cl((int64_t)i - (int64_t)(v.size() - 1 - j)) == v[j];
            f = eq;
        }
        out_exact[i] = f ? 1 : 0;
        out_dual[i] = 0;
    }
    // gft_scan5.hip: probes at lane_start, lane_start + 2, ...; the probe at j reads entry (g[j-2], g[j-1], g[j]): bit g[j-3] of
    // its low word is the flag of j, bit g[j+1] of its high word the flag of j + 1.  (Positions in front of lane_start belong
    // to the lane before: walked here with the same parity, so that every position is answered once.)
    for (int64_t j = (int64_t)(lane_start & 1u); j < (int64_t)len; j += 2) {
        const uint64_t ent = s5.filter[((size_t)gr(j - 2) * G + gr(j - 1)) * G + gr(j)];
        out_dual[j] = (uint8_t)(ent >> gr(j - 3) & 1);
        if (j + 1 < (int64_t)len) out_dual[j + 1] = (uint8_t)(ent >> (32 + gr(j + 1)) & 1);
    }
    if (lane_start & 1u) {                                   // position 0 is the second half of a probe at -1
        const uint64_t ent = s5.filter[((size_t)gr(-3) * G + gr(-2)) * G + gr(-1)];
        if (len) out_dual[0] = (uint8_t)(ent >> (32 + gr(0)) & 1);
    }
    return GFT_OK;
} GFT_CATCH(nullptr)

int gft_debug_eval_programs(const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs, uint32_t n_slots,
                            const uint8_t* present, uint8_t* out_hit, uint32_t* out_depth) try {
    if (!prog_words || !prog_off || !out_hit || (n_slots && !present) || n_slots > (1u << kDwFieldBits)) return GFT_E_INVALID;
    gft_engine scratch;                              // (only its error string is used, by check_program)
    for (uint32_t i = 0; i < n_exprs; i++) {
        if (prog_off[i + 1] < prog_off[i]) return GFT_E_INVALID;
        const int rc = check_program(&scratch, prog_words + prog_off[i], prog_off[i + 1] - prog_off[i], n_slots, i);
        if (rc) return rc;
        std::vector<uint32_t> fw, groups;
        const uint32_t depth = fuse_program(prog_words + prog_off[i], prog_off[i + 1] - prog_off[i], prog_off[i], fw, groups);
        if (out_depth) out_depth[i] = depth;
        // the device's data flow on one document (gft_kernels.hpp "What the kernel reads", gft_solve.hip run_program)
        bool acc = false;
        std::vector<bool> stack;
        for (uint32_t f : fw) {
            const uint32_t w = fused_to_device(f);
            if (w & kDwRare) {
                if (!(w & kDwNeg)) return GFT_E_UNSUPPORTED;         // an INORD group: needs positions
                acc = !acc;
                continue;
            }
            const bool v = (present[(w & kDwFieldMask) >> kDwFieldShift] != 0) != ((w & kDwNeg) != 0);
            if ((w & kDwPop) && stack.empty()) return GFT_E_INVALID;
            const bool x = (w & kDwPop) ? (bool)stack.back() : v;
            const bool A = (w & kDwSel) ? x : (w & kDwOnes) != 0, B = (w & kDwOr) ? x : false;
            const bool before = acc;
            acc = (acc && A) || B;
            if (w & kDwPop) stack.pop_back();
            if (w & kDwPush) stack.push_back(before);
            if (stack.size() > depth) return GFT_E_INVALID;           // fuse_program's own depth figure must hold
        }
        if (!stack.empty()) return GFT_E_INVALID;
        out_hit[i] = acc ? 1 : 0;
    }
    return GFT_OK;
} GFT_CATCH(nullptr)

int gft_debug_host_solve(const uint32_t* words, uint64_t len, const uint32_t* slots, const uint64_t* list_off,
                         const int64_t* positions, uint32_t n_lists, int* out) try {
    if (!words || !out || (n_lists && (!slots || !list_off))) return GFT_E_INVALID;
    gft_engine scratch;                              // (only its error string is used, by check_program)
    uint32_t n_slots = 0;
    for (uint64_t i = 0; i < len; i++)
        if ((words[i] >> 28) == GFT_OP_UNIT) n_slots = std::max(n_slots, (words[i] & GFT_SLOT_MASK) + 1);
    ProgramTraits tr;
    const int rc = check_program(&scratch, words, len, n_slots, 0, &tr);
    if (rc) return rc;
    SlotLists m;
    for (uint32_t k = 0; k < n_lists; k++) {
        std::vector<int64_t>& v = m[slots[k]];       // (a key may carry an empty list: expression_test.go:29-33)
        for (uint64_t i = list_off[k]; i < list_off[k + 1]; i++) v.push_back(positions[i]);
    }
    *out = host_solve(words, len, m) ? 1 : 0;
    return GFT_OK;
} GFT_CATCH(nullptr)

int gft_profile_enable(gft_engine* e, int on) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    e->profiling = on == 2 ? 2 : on != 0;
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_profile_reset(gft_engine* e) try {
    if (!e) return GFT_E_INVALID;
    GFT_LOCK(e);
    if (e->device < 0) return GFT_OK;
    DeviceGuard g(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& kv : e->prof) {
        for (auto& p : kv.second.ev) { e->prof_pool.push_back(p.first); e->prof_pool.push_back(p.second); }
        kv.second.ev.clear();
    }
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_profile_read(gft_engine* e, const char* name, double* total_ms, uint64_t* launches) try {
    if (!e || !name || !total_ms || !launches) return GFT_E_INVALID;
    GFT_LOCK(e);
    *total_ms = 0; *launches = 0;
    if (e->device < 0) return fail(e, GFT_E_HIP, "no HIP device available");
    DeviceGuard g(e->device);
    HIP_TRY(hipStreamSynchronize(e->stream), "sync");
    auto it = e->prof.find(name);
    if (it == e->prof.end()) return GFT_OK;
    for (auto& p : it->second.ev) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second), "event elapsed");
        *total_ms += ms;
    }
    *launches = it->second.ev.size();
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

}  // extern "C"

// =====================================================================================================================
// Multi-device handles: one process, one host thread + stream per device, tables replicated,
// contiguous document ranges of near-equal text bytes, and -- for device-resident shards -- one RCCL gather of the
// bitmaps to the first device.  The Go side keeps calling finder.NewFinder(&GpuEngine{...}): the
// fan-out lives behind the same gft_engine handle.
// =====================================================================================================================
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

// RCCL is bound at run time (dlopen): libgft.so itself does not depend on it, and a process that already carries a
// copy (PyTorch does) shares that one
struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    void* lib = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString; }
};
RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    });
    return api;
}

std::vector<gft_engine*> all_engines(gft_engine* e) {
    std::vector<gft_engine*> v{e};
    v.insert(v.end(), e->peers.begin(), e->peers.end());
    return v;
}

// contiguous document ranges of near-equal text bytes: device i owns documents [cut[i], cut[i+1])
void split_by_bytes(const uint64_t* doc_off, uint64_t n_docs, size_t n, std::vector<uint64_t>& cut) {
    cut.assign(n + 1, n_docs);
    cut[0] = 0;
    const uint64_t base = n_docs ? doc_off[0] : 0, total = n_docs ? doc_off[n_docs] - base : 0;
    for (size_t i = 1; i < n; i++) {
        const uint64_t target = base + (uint64_t)((unsigned __int128)total * i / n);
        uint64_t c = (uint64_t)(std::lower_bound(doc_off, doc_off + n_docs + 1, target) - doc_off);
        cut[i] = std::min(std::max(c, cut[i - 1]), n_docs);
    }
}

// run f(i, engine_i) for every device, each on its own host thread (the caller's thread takes device 0); the first
// failure's code and message become the handle's
template <class F>
int fan_out(gft_engine* e, F f) {
    const std::vector<gft_engine*> eng = all_engines(e);
    std::vector<int> rc(eng.size(), GFT_OK);
    std::vector<std::thread> th;
    th.reserve(eng.size());
    {
        JoinAll joined(th);                  // (also when a thread could not be started, or device 0's share threw)
        // a thread's body never lets an exception out (that would be std::terminate): it becomes the device's status
        auto guarded = [&](size_t i) noexcept {
            try { rc[i] = f(i, eng[i]); } catch (...) { rc[i] = translate_exception(&eng[i]->err); }
        };
        struct InMulti { gft_engine* e; explicit InMulti(gft_engine* e_) : e(e_) { e->in_multi = true; } ~InMulti() { e->in_multi = false; } };
        for (size_t i = 1; i < eng.size(); i++) th.emplace_back(guarded, i);
        InMulti im(e);
        guarded(0);
    }
    for (size_t i = 0; i < eng.size(); i++)
        if (rc[i]) {
            if (i) e->err = "device " + std::to_string(eng[i]->device) + ": " + eng[i]->err;
            return rc[i];
        }
    return GFT_OK;
}

void destroy_multi(gft_engine* e) {
    if (!e->comms.empty() && rccl_api().ok())
        for (void* c : e->comms) (void)rccl_api().CommDestroy((ncclComm_t)c);
    e->comms.clear();
    for (gft_engine* p : e->peers) gft_engine_destroy(p);
    e->peers.clear();
}

int replicate_tables(gft_engine* e, uint32_t flags) {
    // the compiled tables are copied, not compiled again; every device uploads its own copy
    std::vector<std::thread> th;
    std::vector<int> rc(e->peers.size(), GFT_OK);
    th.reserve(e->peers.size());
    {
        JoinAll joined(th);
        for (size_t i = 0; i < e->peers.size(); i++)
            th.emplace_back([&, i]() noexcept {
                gft_engine* p = e->peers[i];
                try {
                    GFT_LOCK(p);
                    p->built = false;
                    p->tab = e->tab; p->s2 = e->s2; p->s3 = e->s3;
                    rc[i] = install_tables(p, flags);
                } catch (...) { rc[i] = translate_exception(&p->err); }
            });
    }
    for (size_t i = 0; i < rc.size(); i++)
        if (rc[i]) { e->err = "device " + std::to_string(e->peers[i]->device) + ": " + e->peers[i]->err; return rc[i]; }
    return GFT_OK;
}

int multi_build(gft_engine* e, const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, uint32_t flags) {
    e->in_multi = true;
    const int rc = gft_build(e, terms_blob, term_off, n_terms, flags);
    e->in_multi = false;
    return rc ? rc : replicate_tables(e, flags);
}

int multi_import_tables(gft_engine* e, const uint8_t* blob, uint64_t len) {
    e->in_multi = true;
    const int rc = gft_import_tables(e, blob, len);
    e->in_multi = false;
    return rc ? rc : replicate_tables(e, e->build_flags);
}

int multi_set_programs(gft_engine* e, const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs, uint32_t n_extra) {
    return fan_out(e, [&](size_t, gft_engine* g) { return gft_set_programs(g, prog_words, prog_off, n_exprs, n_extra); });
}

// caller-supplied matches of the documents [a, b): the same arrays, offsets rebased
struct ExtraSlice {
    std::vector<uint64_t> off;
    gft_extra_matches x{nullptr, nullptr, nullptr};
    const gft_extra_matches* ptr = nullptr;
    void set(const gft_extra_matches* extra, uint64_t a, uint64_t b) {
        if (!(extra && extra->off)) return;
        off.assign(extra->off + a, extra->off + b + 1);
        const uint64_t base = off[0];
        for (auto& o : off) o -= base;
        x.off = off.data(); x.slot = extra->slot + base; x.pos = extra->pos + base;
        ptr = &x;
    }
};

int multi_process(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                  const gft_extra_matches* extra, uint32_t* hit_bitmap) {
    const size_t n = e->peers.size() + 1;
    // (an empty batch may come without offsets, as on a single device: every shard then is [0, 0) of this one entry)
    static const uint64_t kNoDocs[1] = {0};
    if (n_docs == 0) doc_off = kNoDocs;
    split_by_bytes(doc_off, n_docs, n, e->shard_cut);
    const uint64_t words = (e->n_exprs + 31) / 32;
    e->last_nonascii = false;
    const int rc = fan_out(e, [&](size_t i, gft_engine* g) {
        const uint64_t a = e->shard_cut[i], b = e->shard_cut[i + 1];
        std::vector<uint64_t> off(doc_off + a, doc_off + b + 1);        // this shard's documents, offsets from its first byte
        const uint64_t base = off[0];
        for (auto& o : off) o -= base;
        ExtraSlice xs;
        xs.set(extra, a, b);
        return gft_process(g, text_blob + base, off.data(), b - a, flags, xs.ptr, hit_bitmap ? hit_bitmap + a * words : nullptr);
    });
    for (gft_engine* g : e->peers) e->last_nonascii = e->last_nonascii || g->last_nonascii;
    return rc;
}

int multi_process_again(gft_engine* e, uint64_t n_docs, const gft_extra_matches* extra, uint32_t* hit_bitmap) {
    const size_t n = e->peers.size() + 1;
    if (e->shard_cut.size() != n + 1 || e->shard_cut.back() != n_docs || !n_docs)
        return fail(e, GFT_E_INVALID, "gft_process_again: no scan of these documents to reuse");
    const uint64_t words = (e->n_exprs + 31) / 32;
    return fan_out(e, [&](size_t i, gft_engine* g) {
        const uint64_t a = e->shard_cut[i], b = e->shard_cut[i + 1];
        if (a == b) return (int)GFT_OK;
        ExtraSlice xs;
        xs.set(extra, a, b);
        return gft_process_again(g, b - a, xs.ptr, hit_bitmap ? hit_bitmap + a * words : nullptr);
    });
}

int multi_scan(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags, gft_matches* out) {
    const size_t n = e->peers.size() + 1;
    static const uint64_t kNoDocs[1] = {0};
    if (n_docs == 0) doc_off = kNoDocs;                  // (see multi_process)
    std::vector<uint64_t> cut;
    split_by_bytes(doc_off, n_docs, n, cut);
    std::vector<gft_matches> part(n);
    int rc = fan_out(e, [&](size_t i, gft_engine* g) {
        const uint64_t a = cut[i], b = cut[i + 1];
        std::vector<uint64_t> off(doc_off + a, doc_off + b + 1);
        const uint64_t base = off[0];
        for (auto& o : off) o -= base;
        return gft_scan(g, text_blob + base, off.data(), b - a, flags, &part[i]);
    });
    if (rc) return rc;
    // the shards' CSRs one behind the other (device 0's own result lives in this handle's vectors: copied out first)
    uint64_t total = 0;
    for (const auto& p : part) total += p.n_matches;
    std::vector<uint64_t> mo(n_docs + 1, 0);
    std::vector<uint32_t> ti((size_t)total), po((size_t)total);
    uint64_t at = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t a = cut[i], nd = cut[i + 1] - a;
        for (uint64_t d = 0; d <= nd; d++) mo[a + d] = at + part[i].match_off[d];
        if (part[i].n_matches) {
            memcpy(ti.data() + at, part[i].term_id, part[i].n_matches * 4);
            memcpy(po.data() + at, part[i].pos, part[i].n_matches * 4);
        }
        at += part[i].n_matches;
    }
    e->h_match_off.swap(mo); e->h_term.swap(ti); e->h_pos.swap(po);
    // (device 0's own flag is this handle's: it stays, the peers' are OR-ed in)
    for (gft_engine* g : e->peers) e->last_nonascii = e->last_nonascii || g->last_nonascii;
    out->n_docs = n_docs; out->n_matches = total;
    out->match_off = e->h_match_off.data(); out->term_id = e->h_term.data(); out->pos = e->h_pos.data();
    return GFT_OK;
}

}  // namespace

extern "C" {

int gft_engine_create_multi(gft_engine** out, const int* devices, int n_devices) try {
    if (!out || n_devices < 0 || (n_devices && !devices)) return GFT_E_INVALID;
    *out = nullptr;
    std::vector<int> devs(devices, devices + n_devices);
    if (devs.empty()) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
        for (int d = 0; d < count; d++) devs.push_back(d);
        if (devs.empty()) devs.push_back(0);       // (gft_engine_create reports the missing device)
    }
    gft_engine* e = nullptr;
    int rc = gft_engine_create(&e, devs[0]);
    *out = e;
    if (rc) return rc;
    for (size_t i = 1; i < devs.size(); i++) {
        gft_engine* p = nullptr;
        rc = gft_engine_create(&p, devs[i]);
        if (rc) {
            e->err = "device " + std::to_string(devs[i]) + ": " + (p ? p->err : std::string("cannot create an engine"));
            if (p) gft_engine_destroy(p);
            return rc;
        }
        e->peers.push_back(p);
    }
    // RCCL communicators over xGMI for the device-resident entry point -- only when the devices are distinct (a list
    // that names one device twice is a test configuration: the gather is then plain device-to-device copies)
    std::vector<int> uniq(devs);
    std::sort(uniq.begin(), uniq.end());
    const bool distinct = std::adjacent_find(uniq.begin(), uniq.end()) == uniq.end();
    // GFT_RCCL_SELF=1: a list that names ONE device several times gets a communicator of one rank, and the gather moves every
    // further shard's bitmap with a grouped ncclSend / ncclRecv of that rank to itself -- the same dlopen, the same bound
    // entry points, the same group and stream ordering as the N-device gather, on the one GPU a test box has
    const char* self_env = getenv("GFT_RCCL_SELF");
    e->rccl_self = devs.size() > 1 && uniq.front() == uniq.back() && self_env && self_env[0] == '1';
    if (devs.size() > 1 && (distinct || e->rccl_self)) {
        RcclApi& api = rccl_api();
        if (!api.ok()) {
            e->rccl_self = false;
            e->err = "RCCL (librccl.so) could not be loaded: bitmaps will be gathered by device-to-device copies";
            return GFT_W_NO_RCCL;
        }
        std::vector<ncclComm_t> comms(e->rccl_self ? 1 : devs.size());
        DeviceGuard dg(devs[0]);
        const ncclResult_t r = api.CommInitAll(comms.data(), (int)comms.size(), devs.data());
        if (r != ncclSuccess) {
            // the handle is complete without communicators, but the caller is TOLD that its gathers are not RCCL's
            e->rccl_self = false;
            e->err = std::string("ncclCommInitAll: ") + api.GetErrorString(r) + " (bitmaps will be gathered by device-to-device copies)";
            return GFT_W_NO_RCCL;
        }
        for (ncclComm_t c : comms) e->comms.push_back((void*)c);
    }
    return GFT_OK;
} GFT_CATCH(nullptr)

int gft_n_devices(const gft_engine* e) { return e ? (int)e->peers.size() + 1 : 0; }
const char* gft_gather_mode(const gft_engine* e) { return !e || e->peers.empty() ? "" : e->comms.empty() ? "copy" : "rccl"; }

gft_engine* gft_device_engine(gft_engine* e, int i) {
    if (!e || i < 0 || i > (int)e->peers.size()) return nullptr;
    return i == 0 ? e : e->peers[(size_t)i - 1];
}

int gft_split_docs(const gft_engine* e, const uint64_t* doc_off, uint64_t n_docs, uint64_t* cut) try {
    if (!e || !cut || (n_docs && !doc_off)) return GFT_E_INVALID;
    std::vector<uint64_t> c;
    split_by_bytes(doc_off, n_docs, e->peers.size() + 1, c);
    memcpy(cut, c.data(), c.size() * 8);
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

int gft_process_device_multi(gft_engine* e, const uint8_t* const* d_text, const uint64_t* const* d_doc_off, const uint64_t* n_docs,
                             uint32_t flags, uint32_t* d_bitmap_root) try {
    if (!e || !d_text || !d_doc_off || !n_docs) return e ? fail(e, GFT_E_INVALID, "null argument") : GFT_E_INVALID;
    GFT_LOCK(e);
    if (!e->built) return fail(e, GFT_E_NOT_BUILT, "gft_build has not been called");
    if (!e->have_programs) return fail(e, GFT_E_NOT_BUILT, "gft_set_programs has not been called");
    const std::vector<gft_engine*> eng = all_engines(e);
    const size_t n = eng.size();
    const uint64_t words = (e->n_exprs + 31) / 32;
    std::vector<uint64_t> first(n + 1, 0);
    for (size_t i = 0; i < n; i++) first[i + 1] = first[i] + n_docs[i];
    if (first[n] * words && !d_bitmap_root) return fail(e, GFT_E_INVALID, "null bitmap");
    // every device solves its shard into its own bitmap (device 0 straight into its slice of the result) ...
    int rc = fan_out(e, [&](size_t i, gft_engine* g) {
        uint32_t* dst = d_bitmap_root;
        if (i) {
            GFT_LOCK(g);
            DeviceGuard dg(g->device);
            if (g->d_bitmap.ensure(std::max<uint64_t>(n_docs[i] * words, 1) * 4) != hipSuccess) return fail(g, GFT_E_HIP, "bitmap alloc");
            dst = g->d_bitmap.as<uint32_t>();
        }
        return gft_process_device(g, d_text[i], d_doc_off[i], n_docs[i], flags, nullptr, dst);
    });
    if (rc) return rc;
    // ... then ONE exchange step: the shards' bitmaps to the first device, ncclSend / ncclRecv in one group over xGMI
    // (plain device-to-device copies when there is no communicator)
    if (n > 1 && words) {
        RcclApi& api = rccl_api();
        if (!e->comms.empty() && api.ok()) {
            DeviceGuard dgr(e->device);
            ncclResult_t r = api.GroupStart();
            for (size_t i = 1; i < n && r == ncclSuccess; i++) {
                if (!n_docs[i]) continue;
                // (one rank for all shards under GFT_RCCL_SELF: peer 0 on communicator 0, both halves on the root's stream --
                // the shard's stream was drained when its gft_process_device returned)
                const int from = e->rccl_self ? 0 : (int)i;
                ncclComm_t send_comm = (ncclComm_t)e->comms[e->rccl_self ? 0 : i];
                hipStream_t send_stream = e->rccl_self ? e->stream : eng[i]->stream;
                r = api.Recv(d_bitmap_root + first[i] * words, n_docs[i] * words, ncclUint32, from, (ncclComm_t)e->comms[0], e->stream);
                if (r == ncclSuccess)
                    r = api.Send(eng[i]->d_bitmap.p, n_docs[i] * words, ncclUint32, 0, send_comm, send_stream);
            }
            const ncclResult_t r2 = api.GroupEnd();
            if (r != ncclSuccess || r2 != ncclSuccess)
                return fail(e, GFT_E_HIP, std::string("RCCL gather: ") + api.GetErrorString(r != ncclSuccess ? r : r2));
            for (gft_engine* g : eng) {
                DeviceGuard dg(g->device);
                HIP_TRY(hipStreamSynchronize(g->stream), "RCCL gather");
            }
        } else {
            DeviceGuard dg(e->device);
            for (size_t i = 1; i < n; i++)
                if (n_docs[i]) HIP_TRY(hipMemcpyAsync(d_bitmap_root + first[i] * words, eng[i]->d_bitmap.p, n_docs[i] * words * 4, hipMemcpyDeviceToDevice, e->stream), "bitmap gather");
            HIP_TRY(hipStreamSynchronize(e->stream), "bitmap gather");
        }
    }
    for (gft_engine* g : e->peers) e->last_nonascii = e->last_nonascii || g->last_nonascii;   // (device 0's own flag stays)
    return GFT_OK;
} GFT_CATCH((e ? &e->err : nullptr))

}  // extern "C"
