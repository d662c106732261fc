// gft_scan4.hip -- the suffix-window Aho-Corasick scan for gfx950 in STREAMING form (tables: scan2_tables.hpp, the same as
// gft_scan2.hip).  Replaces (*Matcher).MatchAll behind CloudflareForkEngine.FindSubstrings (finder/substringEngine.go:110-119).
//
// gft_scan2.hip takes one work unit per wave iteration and runs its phases one after the other: every unit pays the
// HBM latency of its first bytes, a candidate-list build, partly filled verification trips (336 flagged positions = 2.6
// trips of 128), a bucket-table trip with two dependent memory round trips for 57 survivors, and a flush -- a fifth of a
// 4 KB document's time is fixed cost, and text that waits for its verification leaves L2 meanwhile (1.5 x the algorithmic
// HBM traffic).  Here a wave takes a CHUNK of up to eight consecutive units -- they are consecutive in the text blob too --
// and streams through it:
//   FILTER   coalesced rounds of 1 KiB: lane k owns bytes [1024 r + 16 k, +16) of round r (one 16-byte load per lane and
//            round, the next round in flight), classes from the 256-byte LDS table, the window key rolled per position,
//            one probe of the LDS bit filter per byte.  The key rolls across unit and document boundaries: a window that
//            spans two documents can only raise a flag that the verification drops (a term must start inside the document
//            its window ends in).
//   QUEUE A  flagged positions (16-bit, chunk relative) are appended to a queue in LDS; whenever 128 are waiting, a
//   STAGE A  trip takes them, two per lane: LDS-only decisions (short terms emitted, fingerprint table) -- always a FULL
//            trip, on text that was streamed a round or two ago (L1 / L2, not HBM); its text loads are issued before the
//            next round is filtered and consumed behind it.
//   QUEUE B  survivors wait with their window key; whenever 64 are waiting, a
//   STAGE B  trip loads both candidate slots of every key and the text around the window at once.
//   OUTPUT   matches go to a fifo in LDS tagged with their unit (3 bits) and are flushed, whenever the fifo fills, into
//            PER-UNIT regions of the wave's slab of the match pool: a region is sized from the match density the previous
//            batch had (bytes x density x 1.6 + 48), so a unit's matches are contiguous (unit_start / unit_count, what
//            the solver and the CSR gather read) without sorting the fifo and without draining the queues per unit.  A
//            unit that outgrows its region is walked again on its own with a region of the size the first walk counted.
//   Only at the end of a chunk the queues are drained with partly filled trips.
// HBM traffic: text once + 8 B per match (4 B in presence-only mode).  No MFMA (byte automaton, not a contraction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gft_kernels.hpp"

namespace gft {

namespace {

#include "gft_scan2_dev.hpp"

constexpr uint32_t kS4Base = 8;                  // a chunk's positions count from (up to) 8 bytes in front of it
// queue A: flagged positions (u16), a ring that holds a whole round's (about 340 on the benchmark) while the round before
// is drained; with positions next to the fifo's terms there is only room for the smaller one
__host__ __device__ constexpr uint32_t s4_qa_cap(bool want_pos) { return want_pos ? 256u : 512u; }
// queue B: survivors (u16 position + u32 window key).  A trip of stage A adds up to 128: stage B runs first until at most
// qb_cap - 128 are waiting
__host__ __device__ constexpr uint32_t s4_qb_cap(bool want_pos) { return want_pos ? 160u : 192u; }
// per-wave bookkeeping in LDS: one record of eight words per unit of the chunk -- {its document's start (position coordinates,
// mod 2^32: only differences are used), the document's length, the unit's end, -, its region's size, its region's offset in
// the chunk's part of the slab, its matches so far, -}: what a verification stage needs of a unit is one 16-byte read, what the
// flush needs another -- and the wave's list of units to walk again
constexpr uint32_t kS4Redo = kScan4ChunkUnits;
constexpr uint32_t kS4MetaWords = 8 * kScan4ChunkUnits + 2 * kS4Redo;
typedef __attribute__((address_space(3))) uint32_t lds32;
typedef __attribute__((address_space(3))) uint16_t lds16;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4u lds128;
enum { kTabDs = 0, kTabDlen = 1, kTabUend = 2, kTabBound = 4, kTabReg = 5, kTabCur = 6 };
__device__ __forceinline__ lds32& tab_at(lds32* meta, int which, uint32_t t) { return meta[8 * t + which]; }

// the unit of every lane's position: tags run from the hint (the unit of an earlier position of the same queue) upwards
struct LaneUnit { uint32_t tag, ds, dlen, uend; };
__device__ __forceinline__ LaneUnit lane_unit(uint32_t p, bool on, lds32* meta, uint32_t nu, uint32_t& hint) {
    uint32_t t = hint, tag = hint;
    bool first = true;
    while (t + 1 < nu) {
        const uint32_t ue = __builtin_amdgcn_readfirstlane(tab_at(meta, kTabUend, t));
        const bool past = on && p >= ue;
        if (first && !__any(on && !past)) hint = t + 1;      // every position is behind unit t: the next trip starts there too
        else first = false;
        if (!__any(past)) break;
        t++;
        if (past) tag = t;
    }
    const v4u r = *reinterpret_cast<lds128*>(meta + 8 * tag);
    return LaneUnit{tag, r.x, r.y, r.z};
}

// the wave's match fifo: term | unit tag << 29 (and the position next to it when positions are wanted)
struct Fifo4 {
    lds32* term;
    lds32* pos;                // nullptr: presence only
    uint32_t cap, n;
};

template <bool HASHED, bool FPT_LDS, bool DBG>
__global__ void __launch_bounds__(kScan2Threads) k_scan4(const Scan2Params P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* cls = smem;
    uint32_t* filt = reinterpret_cast<uint32_t*>(smem + 256);
    uint8_t* short3 = smem + 256 + (size_t)P.filter_words * 4;
    uint8_t* fpt = short3 + P.short3_bytes;
    uint32_t* lrec = reinterpret_cast<uint32_t*>(fpt + (FPT_LDS ? kScan2FptSize : 0));
    uint32_t* wg_next = reinterpret_cast<uint32_t*>(smem + (((size_t)(reinterpret_cast<uint8_t*>(lrec) - smem) + P.shorts_words * 4 + 15) & ~(size_t)15));
    uint8_t* wave_lds_all = reinterpret_cast<uint8_t*>(wg_next) + 16;

    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = P.cls[i];
    for (uint32_t i = threadIdx.x; i < P.filter_words; i += blockDim.x) filt[i] = P.filter[i];
    for (uint32_t i = threadIdx.x; i < P.short3_bytes / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(short3)[i] = reinterpret_cast<const uint32_t*>(P.short3)[i];
    for (uint32_t i = threadIdx.x; FPT_LDS && i < kScan2FptSize / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(fpt)[i] = reinterpret_cast<const uint32_t*>(P.fpt)[i];
    for (uint32_t i = threadIdx.x; i < P.shorts_words; i += blockDim.x) lrec[i] = P.shorts_packed[i];
    if (threadIdx.x == 0) { wg_next[0] = blockDim.x >> 6; wg_next[1] = wg_next[2] = wg_next[3] = 0; }   // [0] work counter, [1] waves done, [2..3] matches
    __syncthreads();

    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wg_waves = blockDim.x >> 6;
    // per-wave LDS: [bookkeeping: kS4MetaWords words][queue A: u16 x s4_qa_cap][queue B positions: u16 x s4_qb_cap]
    //               [queue B keys: u32 x s4_qb_cap][fifo terms: u32 x fifo_cap][fifo positions: u32 x fifo_cap when wanted]
    const uint32_t fcap = P.cand_cap;                                 // (scan4_plan: entries of the match fifo)
    const uint32_t qa_cap = s4_qa_cap(P.want_pos != 0), qb_cap = s4_qb_cap(P.want_pos != 0), qb_lim = qb_cap - 128;
    const uint32_t wave_bytes = kS4MetaWords * 4 + qa_cap * 2 + qb_cap * 6 + fcap * (P.want_pos ? 8u : 4u);
    // (32-bit LDS pointers: one scalar register each, ds_ instructions without an address add)
    lds32* meta = (lds32*)(uintptr_t)((uint32_t)(uintptr_t)(lds_u8*)wave_lds_all + wave * wave_bytes);
    lds32* redo_u = meta + 8 * kScan4ChunkUnits;
    lds32* redo_c = redo_u + kS4Redo;
    lds16* qa = (lds16*)(meta + kS4MetaWords);
    lds16* qbp = qa + qa_cap;
    lds32* qbk = (lds32*)(qbp + qb_cap);
    Fifo4 ff{qbk + qb_cap, P.want_pos ? qbk + qb_cap + fcap : nullptr, fcap, 0};
    const uint32_t kp = __builtin_amdgcn_readfirstlane(P.kp), kp2 = __builtin_amdgcn_readfirstlane(kp * kp);
    lds_u8* lcls = (lds_u8*)0;
    lds_u32* lfilt = (lds_u32*)256;
    if ((uint32_t)(uintptr_t)(lds_u8*)smem != 0) __builtin_trap();   // see lds_u8

    // Match pool: every wave of the grid owns one slab from the start, further slabs come from the cursor behind those
    // (gft_scan2.hip); a chunk reserves the regions of all its units in the wave's current slab
    const uint32_t slab = KARG(slab);
    const uint64_t static_slabs = (uint64_t)gridDim.x * wg_waves * slab;
    uint64_t slab_next = ((uint64_t)blockIdx.x * wg_waves + wave) * slab, wave_matches = 0;   // wave-uniform
    uint32_t slab_left = slab;
    bool told_nonascii = false;
    // phase clocks of the timing studies (GFT_SCAN_DEBUG & 64; the DBG instantiation only): cycles of this wave per phase --
    // 0 chunk set-up, 1 filter, 2 queue A push, 3 stage A issue, 4 stage A decisions, 5 stage B, 6 fifo flush, 7 unit records
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = DBG ? clock64() : 0;
    auto mark = [&](int ph) {
        if (DBG && (P.dbg & 64)) { const unsigned long long now = clock64(); tl[ph] += now - tprev; tprev = now; }
    };

    // Work distribution as in gft_scan2.hip, chunk by chunk: in round k the workgroup owns the chunks
    // k * (grid * waves) + b * waves + [0, waves), its waves take them one by one from a counter in LDS
    const uint32_t cu = __builtin_amdgcn_readfirstlane(P.chunk_units);              // units per chunk (1 .. kScan4ChunkUnits)
    const uint64_t n_chunks = (P.n_units + cu - 1) / cu;
    const uint64_t round_chunks = (uint64_t)gridDim.x * wg_waves, wg_first = (uint64_t)blockIdx.x * wg_waves;
    auto chunk_of = [&](uint32_t item) { return (uint64_t)(item / wg_waves) * round_chunks + wg_first + item % wg_waves; };

    // one unit record per lane (lanes 0 .. cu-1): the next chunk's travel while this one is scanned
    auto fetch_units = [&](uint64_t ch, Unit& un) {
        const uint64_t u = ch * cu + lane;
        un = Unit{0, 0, 0};
        if (lane < cu && u < P.n_units) un = P.units[u];
    };
    auto fetch_docs = [&](uint64_t ch, const Unit& un, uint64_t& dabs, uint32_t& dlen) {
        const uint64_t u = ch * cu + lane;
        dabs = 0; dlen = 0;
        if (lane < cu && u < P.n_units) { dabs = P.doc_off[un.doc]; dlen = (uint32_t)(P.doc_off[un.doc + 1] - dabs); }
    };
    uint64_t ch = wg_first + wave, nch = 0;
    Unit un_n{0, 0, 0};
    uint64_t dabs_n = 0;
    uint32_t dlen_n = 0;
    if (ch < n_chunks) { fetch_units(ch, un_n); fetch_docs(ch, un_n, dabs_n, dlen_n); }
    uint32_t n_redo = 0;                         // units waiting to be walked again (their matches outgrew their regions)

    // Work items: the wave's chunks, and behind a chunk the units of it that have to be walked again, alone, with a region of
    // the size the first walk counted
    for (;;) {
        const bool chunks_left = ch < n_chunks;
        const bool is_redo = n_redo != 0;        // (a chunk adds up to kScan4ChunkUnits = kS4Redo entries: the list is emptied before the next one)
        if (!chunks_left && !is_redo) break;
        Unit un;
        uint64_t dabs;
        uint32_t dlen;
        uint64_t u_first;
        uint32_t nu_all, redo_count = 0;
        bool more_chunks = false;
        if (is_redo) {
            n_redo--;
            u_first = __builtin_amdgcn_readfirstlane(redo_u[n_redo]);
            redo_count = __builtin_amdgcn_readfirstlane(redo_c[n_redo]);
            nu_all = 1;
            un = Unit{0, 0, 0}; dabs = 0; dlen = 0;
            if (lane == 0) { un = P.units[u_first]; dabs = P.doc_off[un.doc]; dlen = (uint32_t)(P.doc_off[un.doc + 1] - dabs); }
        } else {
            un = un_n; dabs = dabs_n; dlen = dlen_n;
            uint32_t item = 0;
            if (lane == 0) item = __hip_atomic_fetch_add(wg_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nch = chunk_of((uint32_t)__builtin_amdgcn_readfirstlane(item));
            more_chunks = nch < n_chunks;
            if (more_chunks) fetch_units(nch, un_n);
            u_first = ch * cu;
            nu_all = (uint32_t)(P.n_units - u_first < cu ? P.n_units - u_first : cu);
        }
        {
            // A chunk is ONE stream: the units of consecutive documents (and the slices of a long one) follow each other in the
            // blob.  Only the empty units behind the real ones of a table that was sized blind do not: they end the chunk (and
            // own nothing).
            const uint64_t a_abs = dabs + un.lo, b_abs = dabs + un.hi;              // this lane's unit in the blob
            const uint64_t a_next = (uint64_t)__shfl((unsigned long long)a_abs, (int)((lane + 1) & 63u), 64);
            const uint64_t brk = __ballot(lane + 1 < nu_all && a_next != b_abs);
            const uint32_t nu = brk ? (uint32_t)__builtin_ctzll(brk) + 1 : nu_all;   // units of the stream (tags 0 .. nu-1)
            if (lane >= nu && lane < nu_all) { KARG(unit_start)[u_first + lane] = 0; KARG(unit_count)[u_first + lane] = 0; }
            const uint64_t s_abs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a_abs >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a_abs);
            const uint64_t e_abs = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(b_abs >> 32), nu - 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)b_abs, nu - 1);
            const uint32_t len = (uint32_t)(e_abs - s_abs);                          // bytes of the stream (< 2^16)
            const uint64_t base = s_abs >= kS4Base ? s_abs - kS4Base : 0;            // positions p = blob offset - base
            const uint32_t s_p = (uint32_t)(s_abs - base), e_p = s_p + len;
            const uint32_t lo0 = __builtin_amdgcn_readfirstlane(un.lo);
            // the unit tables (lane t: unit t) and the regions: one behind the other in the wave's slab (a fresh slab -- of the
            // chunk's size, if that is larger -- when what is left does not hold them all)
            const uint32_t bound_l = lane >= nu ? 0u : is_redo ? redo_count : (uint32_t)(((uint64_t)(un.hi - un.lo) * P.bound_q16) >> 16) + P.bound_add;
            const uint32_t bincl = wave_incl_scan(bound_l);
            const uint32_t btotal = lane_value(bincl, 63);
            if (btotal > slab_left) {
                const uint32_t want = btotal > slab ? btotal : slab;
                uint64_t nb = 0;
                if (lane == 0) nb = atomicAdd(reinterpret_cast<unsigned long long*>(KARG(cursor)), (unsigned long long)want);
                slab_next = static_slabs + __shfl(nb, 0, 64);
                slab_left = want;
            }
            const uint64_t chunk_base = slab_next;
            slab_next += btotal;
            slab_left -= btotal;
            const bool pool_ok = chunk_base + btotal <= KARG(pool_cap);               // (beyond the pool: counted, not written; the host runs the batch again)
            if (lane < kScan4ChunkUnits) {
                lds128* rec = reinterpret_cast<lds128*>(meta + 8 * lane);
                rec[0] = v4u{(uint32_t)(dabs - base), dlen, (uint32_t)(b_abs - base), 0u};
                rec[1] = v4u{bound_l, bincl - bound_l, 0u, 0u};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const Ctx c{P, cls, filt, P.short3_bytes ? short3 : nullptr, fpt, lrec, P.text + base, base, kp2,
                        base < 7, base < 23, s_p, e_p, base + e_p + 4 > P.text_bytes, 0u};

            // ---- the match fifo.  flush() appears at three places only (each stage checks for room ONCE, in front of its
            // appends); the stages themselves appear once each: the job is a loop over one state machine ------------------------
            int phase = 0;                       // (the enclosing phase of a flush, for the phase clocks)
            auto flush = [&]() {
                mark(phase);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t i0 = 0; i0 < ff.n; i0 += 64) {
                    const bool on = i0 + lane < ff.n;
                    const uint32_t e = ff.term[on ? i0 + lane : 0];
                    const uint32_t ps = ff.pos ? ff.pos[on ? i0 + lane : 0] : 0u;
                    const uint32_t tag = e >> 29;
                    uint64_t rem = __ballot(on);
                    while (rem) {
                        const uint32_t t = __builtin_amdgcn_readlane(tag, (uint32_t)__builtin_ctzll(rem));
                        const uint64_t m = __ballot(on && tag == t);
                        rem &= ~m;
                        const v4u rg = *reinterpret_cast<lds128*>(meta + 8 * t + 4);         // {bound, region, matches so far, -}
                        const uint32_t bnd = __builtin_amdgcn_readfirstlane(rg.x), reg = __builtin_amdgcn_readfirstlane(rg.y);
                        const uint32_t cur = __builtin_amdgcn_readfirstlane(rg.z);
                        if (on && tag == t) {
                            const uint32_t idx = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                            if (idx < bnd && pool_ok) {
                                const uint64_t at = chunk_base + reg + idx;
                                __builtin_nontemporal_store(e & 0x1FFFFFFFu, &KARG(pool_term)[at]);
                                if (ff.pos) __builtin_nontemporal_store(ps, &KARG(pool_pos)[at]);
                            }
                        }
                        if (lane == 0) tab_at(meta, kTabCur, t) = cur + (uint32_t)__popcll(m);
                    }
                }
                ff.n = 0;
                __builtin_amdgcn_wave_barrier();
                mark(6);
            };
            // (room for 64 more entries is the caller's business)
            auto append = [&](bool em, uint32_t term, uint32_t tag, uint32_t pos) {
                const uint64_t mask = __ballot(em);
                if (em) {
                    const uint32_t idx = ff.n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                    ff.term[idx] = term | tag << 29;
                    if (ff.pos) ff.pos[idx] = pos;
                }
                ff.n += (uint32_t)__popcll(mask);
            };

            uint32_t qa_n = 0, qa_head = 0, qb_n = 0, qa_hint = 0, qb_hint = 0;
            // the trip of stage A that is in flight: positions and their text (a_n = 0: none)
            bool a_on[kStageAWays] = {false, false};
            uint32_t a_p[kStageAWays] = {0, 0};                                   // position | unit tag << 16
            uint32_t a_pd[kStageAWays] = {0, 0};                                  // position inside its document
            Text8 a_tx[kStageAWays] = {{0, 0}, {0, 0}};
            uint32_t a_n = 0;

            // ---- the positions in front of the stream: a term whose window ends up to kScan2MaxOff bytes before it may end inside
            if (lo0) {
                const uint32_t nb = lo0 < kScan2MaxOff ? lo0 : kScan2MaxOff;
                if (lane < nb) qa[lane] = (uint16_t)(s_p - nb + lane);
                qa_n = nb;
            }

            // ---- the stream: ROUNDS of 64 x C bytes, lane k owns the C consecutive bytes [round + C k, + C) (C = 64, less in the
            // last round so that all lanes have work) and filters them in pieces of 16.  The first piece of a round touches
            // every 128-byte line of the round at once, the later pieces find them in L1 / L2: a round pays the HBM latency
            // once, and the piece that pays it -- the next round's first -- is requested as the YOUNGEST load of the wave, while
            // the last piece of this round is filtered and its flagged positions are queued, work that needs no memory
            // (vector-memory results return in order: whatever is requested behind a load from HBM waits for it).
            const uint32_t c_max = P.round_c;                                        // bytes per lane and round: 16 .. 64
            auto round_c = [&](uint32_t rem) { const uint32_t per = (rem + 63) >> 6; return per >= c_max ? c_max : per <= 16 ? 16u : (per + 15) & ~15u; };
            uint32_t rb = 0;                                                         // the round being filtered: first byte,
            uint32_t C = round_c(len), q = 0;                                        // bytes per lane, the next piece
            bool have_round = len > 0;
            const uint8_t* src = P.text + s_abs;
            U128u nxt{0, 0, 0, 0};
            uint32_t nhist = 0;                  // the four bytes in front of the lane's range (the round's first piece rolls on from them)
            // (the four bytes in front of the stream too, together with its first piece: no round trip of their own.  Only within
            // four bytes of the blob's start lane 0 has to go byte by byte -- the padding class stands for what is not there)
            const bool blob_head = s_abs < 4;
            if (have_round && C * lane < len) {
                nxt = *reinterpret_cast<const U128u*>(src + C * lane);
                if (lane || !blob_head) nhist = load_u32_unaligned(src + C * lane - 4);
            }
            // st_*: the rolling key's state behind this lane's last piece
            uint32_t car_cp = P.pad_class, car_pm1 = 0, car_pm2 = 0, st_cp = 0, st_pm1 = 0, st_pm2 = 0;
            if (blob_head) {
                uint32_t k1 = P.pad_class, k2 = P.pad_class, k3 = P.pad_class;
                if (s_abs >= 1) k1 = lcls[P.text[s_abs - 1]];
                if (s_abs >= 2) k2 = lcls[P.text[s_abs - 2]];
                if (s_abs >= 3) k3 = lcls[P.text[s_abs - 3]];
                car_cp = __builtin_amdgcn_readfirstlane(k1);
                car_pm1 = __builtin_amdgcn_readfirstlane(mad24s(k2, kp, k1));
                car_pm2 = __builtin_amdgcn_readfirstlane(mad24s(k3, kp, k2));
            }
            uint32_t hib = 0;
            mark(0);
            // flags of the round being filtered (bit i of m0:m1 = byte i of the lane's C), and the round whose flagged
            // positions are on their way into queue A: lanes [push_l0, 64) are still to go
            uint32_t m0 = 0, m1 = 0, psh0 = 0, psh1 = 0, fcnt = 0, fincl = 0, push_rb = 0, push_C = 0, push_l0 = 64;
            // the stages as code: each is instantiated twice -- in the stream loop (the hot path: request a trip's text, filter
            // a piece, decide the trip) and in the service loop that runs when queue A is full or the stream is over
            auto stage_b = [&](const uint32_t n) {
                phase = 5;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (P.prio) __builtin_amdgcn_s_setprio(3);
                const bool on = lane < n;
                const uint32_t p = qbp[on ? lane : 0];
                const uint32_t x = qbk[on ? lane : 0];
                // what is left of the queue moves to its front (every lane holds its entry by now)
                const uint32_t left = qb_n - n;
                uint32_t mp[2], mk[2];
#pragma unroll
                for (int h = 0; h < 2; h++) { const uint32_t i = n + 64 * h + lane; mp[h] = qbp[i < qb_n ? i : 0]; mk[h] = qbk[i < qb_n ? i : 0]; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h = 0; h < 2; h++) { const uint32_t i = 64 * h + lane; if (i < left) { qbp[i] = (uint16_t)mp[h]; qbk[i] = mk[h]; } }
                qb_n = left;
                // both candidate slots of the key and the text around the window, all at once (the key is known)
                const Slot s0 = slot_load(&P.slots[scan2_pair_slot(x, 0, P.slot_shift, P.slot_seed)]);
                const Slot s1 = slot_load(&P.slots[scan2_pair_slot(x, 1, P.slot_shift, P.slot_seed)]);
                const Text8 t8 = cand_load(c, p);
                Front t = front_load(c, p, t8.tw);
                const uint32_t tl = tail_load(c, p);
                const LaneUnit lu = lane_unit(p, on, meta, nu, qb_hint);
                const uint32_t pd = p - lu.ds;                                   // position inside the document
                // a term must end inside this stream and inside the document its window ends in
                const uint32_t dend = lu.ds + lu.dlen;
                const uint32_t hi = dend - s_p < e_p - s_p ? dend : e_p;
                Slot e;
                const bool have = slot_pick(x, s0, s1, e) && on;
                if (__any(have)) {
                    // the bucket's entries one after the other -- one for nearly every key; the next entry of a bucket of
                    // several terms is in flight during the compare
                    const bool multi = have && (e.a.y & kScan2Multi);
                    const uint32_t n_ent = have ? (multi ? e.a.z : 1u) : 0u, more_at = e.a.y & ~kScan2Multi;
                    Slot cur = e;
                    if (__any(multi)) { if (multi) cur = slot_load(&P.more[more_at]); }
                    uint32_t folded = 0;
                    for (uint32_t j = 0; __any(j < n_ent); j++) {
                        Slot nx = cur;
                        if (multi && j + 1 < n_ent) nx = slot_load(&P.more[more_at + j + 1]);
                        const bool act = j < n_ent;
                        const uint32_t kmax = wave_kmax(act ? cur.a.z & kScan2LenMask : 0);
                        if (P.fold) front_fold_upto(t, folded, kmax);
                        const bool ok = act && entry_ok_x(c, p, pd, s_p, hi, t, tl, cur, kmax);
                        const uint32_t pe = p + (cur.a.z >> 24);
                        if (ff.n + 64 > ff.cap) flush();
                        append(ok, cur.a.y, lu.tag + (pe >= lu.uend ? 1u : 0u), match_pos(P, pd, cur.a.z));
                        cur = nx;
                    }
                }
                if (P.prio) __builtin_amdgcn_s_setprio(0);
                mark(5);
            };
            auto stage_a_complete = [&]() {
                phase = 4;
                if (P.prio) __builtin_amdgcn_s_setprio(2);
                Cand k[kStageAWays];
#pragma unroll
                for (int q2 = 0; q2 < kStageAWays; q2++) cand_keys(c, a_p[q2] & 0xFFFFu, a_tx[q2], k[q2]);
#pragma unroll
                for (int q2 = 0; q2 < kStageAWays; q2++) cand_decide<FPT_LDS>(c, k[q2]);
#pragma unroll
                for (int q2 = 0; q2 < kStageAWays; q2++) {
                    if (64u * q2 >= a_n) continue;
                    const uint32_t pd = a_pd[q2], ptag = a_p[q2] >> 16;
                    // terms of length <= 3 ending here (not at the positions in front of the stream)
                    uint32_t sid = a_on[q2] && k[q2].p >= s_p ? k[q2].sid : 0;
                    if (DBG && (P.dbg & 8)) sid = 0;                             // timing study: no short terms
                    if (__any(sid != 0)) {
                        uint32_t rec[3] = {0, 0, 0};
                        if (sid) short_record(c, sid, k[q2].x3, rec);
                        uint32_t n_short = 0;
#pragma unroll
                        for (uint32_t j = 0; j < 3; j++) {
                            if (rec[j] && (rec[j] >> 28) > pd + 1) rec[j] = 0;               // (it would start before its document)
                            n_short += (uint32_t)__popcll(__ballot(rec[j] != 0));
                        }
                        if (ff.n + n_short > ff.cap) flush();                                // (n_short <= 192 <= the fifo's capacity)
#pragma unroll
                        for (uint32_t j = 0; j < 3; j++) {
                            if (j && !__any(rec[j] != 0)) break;
                            const uint32_t L = rec[j] >> 28;
                            append(rec[j] != 0, rec[j] & 0x0FFFFFFFu, ptag, P.pos_end ? pd : pd + 1 - L);
                        }
                    }
                    // survivors -> queue B with their window keys (it held fewer than 64: stage B goes first)
                    const bool keep = a_on[q2] && k[q2].go_long && !(DBG && (P.dbg & 4));     // (timing study 4: no stage B)
                    const uint64_t sb = __ballot(keep);
                    if (keep) {
                        const uint32_t idx = qb_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0));
                        qbp[idx] = (uint16_t)k[q2].p;
                        qbk[idx] = k[q2].x;
                    }
                    qb_n += (uint32_t)__popcll(sb);
                }
                a_n = 0;
                if (P.prio) __builtin_amdgcn_s_setprio(0);
                mark(4);
            };
            auto stage_a_issue = [&](const uint32_t n) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int q2 = 0; q2 < kStageAWays; q2++) {
                    const uint32_t i = 64 * q2 + lane;
                    a_on[q2] = i < n;
                    a_p[q2] = qa[(qa_head + (a_on[q2] ? i : 0)) & (qa_cap - 1)];
                }
#pragma unroll
                for (int q2 = 0; q2 < kStageAWays; q2++) a_tx[q2] = cand_load(c, a_p[q2]);
                // the positions' units, here rather than with the decisions: these LDS round trips travel with the text
#pragma unroll
                for (int q2 = 0; q2 < kStageAWays; q2++) {
                    a_pd[q2] = 0;
                    if (64u * q2 >= n) continue;
                    const LaneUnit lu = lane_unit(a_p[q2], a_on[q2], meta, nu, qa_hint);
                    a_pd[q2] = a_p[q2] - lu.ds;                                      // position inside the document
                    a_p[q2] |= lu.tag << 16;
                }
                qa_head = (qa_head + n) & (qa_cap - 1);
                qa_n -= n;
                a_n = n;
                mark(3);
            };
            auto push_some = [&]() {
                const uint32_t before = push_l0 ? lane_value(fincl, push_l0 - 1) : 0;
                const uint32_t room = qa_cap - qa_n;
                const bool fits = lane >= push_l0 && fincl - before <= room;
                const uint64_t fm = __ballot(fits) >> push_l0;
                const uint32_t nl = fm == ~0ull >> push_l0 ? 64 - push_l0 : (uint32_t)__builtin_ctzll(~fm);   // lanes that fit (may be 0)
                if (nl) {
                    const uint32_t l1 = push_l0 + nl;
                    const uint32_t ptotal = lane_value(fincl, l1 - 1) - before;
                    if (lane >= push_l0 && lane < l1) {
                        uint32_t wpos = qa_head + qa_n + fincl - fcnt - before;
                        const uint32_t p0 = s_p + push_rb + push_C * lane;
                        uint32_t mk = psh0;
                        while (mk) {
                            const uint32_t i = __builtin_ctz(mk);
                            mk &= mk - 1;
                            qa[wpos++ & (qa_cap - 1)] = (uint16_t)(p0 + i);
                        }
                        mk = psh1;
                        while (mk) {
                            const uint32_t i = __builtin_ctz(mk);
                            mk &= mk - 1;
                            qa[wpos++ & (qa_cap - 1)] = (uint16_t)(p0 + 32 + i);
                        }
                    }
                    qa_n += ptotal;
                    push_l0 = l1;
                }
                mark(2);
            };
            auto filter_piece = [&]() {
                const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
                const uint32_t off = rb + C * lane + 16 * q;                     // this lane's piece inside the stream
                const bool last_piece = 16 * (q + 1) >= C;
                const uint32_t rb2 = rb + 64 * C;                                // (the next round, if the stream goes on)
                const uint32_t C2 = rb2 < len ? round_c(len - rb2) : 16u;
                if (off < len) hib |= (w[0] | w[1]) | (w[2] | w[3]);            // (up to 15 bytes behind the stream: conservative)
                const uint32_t hist = nhist;
                {
                    const uint32_t noff = last_piece ? rb2 + C2 * lane : off + 16;
                    if (noff < len && (!last_piece || rb2 < len)) {
                        nxt = *reinterpret_cast<const U128u*>(src + noff);
                        if (last_piece) nhist = load_u32_unaligned(src + noff - 4);      // (rb2 >= 1024: never in front of the stream)
                    }
                }
                // the state in front of the piece: this lane's own behind its last piece; for the first piece of a round the
                // classes of the three bytes in front of the lane's range (the previous lane filters them later in this round;
                // lane 0 of the first round: what lies in front of the stream)
                uint32_t cp = st_cp, pm1 = st_pm1, pm2 = st_pm2;
                if (q == 0) {
                    const uint32_t k1 = lcls[hist >> 24], k2 = lcls[(hist >> 16) & 0xFF], k3 = lcls[(hist >> 8) & 0xFF];
                    const bool head = blob_head && rb == 0 && lane == 0;
                    cp = head ? car_cp : k1;
                    pm1 = head ? car_pm1 : mad24s(k2, kp, k1);
                    pm2 = head ? car_pm2 : mad24s(k3, kp, k2);
                }
                // running window key as in gft_scan2.hip: x(i) = pair(i-2) * kp^2 + pair(i), pair(i) = class(i-1) * kp + class(i)
                // (dword by dword: four class lookups in flight together, then four filter probes -- at this kernel's register
                // pressure the compiler's own schedule is one LDS round trip after the other, hence the group barriers)
                uint32_t acc = 0;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    uint32_t cl[4], x[4], fw[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) cl[b] = lcls[(w[d] >> (8 * b)) & 0xFF];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t pair = mad24s(cp, kp, cl[b]);
                        x[b] = mad24s(pm2, kp2, pair);
                        pm2 = pm1; pm1 = pair; cp = cl[b];
                        if (HASHED) x[b] = (x[b] * kGoldDev) >> P.hash_shift;
                    }
#pragma unroll
                    for (int b = 0; b < 4; b++) fw[b] = lfilt[x[b] >> 5];
#pragma unroll
                    for (int b = 0; b < 4; b++) acc = __builtin_amdgcn_alignbit(fw[b] >> (x[b] & 31), acc, 1);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // 4 VALU: the dword's bytes
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // 4 DS reads: classes
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);     // VALU: pairs, keys, probe addresses
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // 4 DS reads: filter words
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // VALU: flags into the mask
                }
                st_cp = cp; st_pm1 = pm1; st_pm2 = pm2;
                const uint32_t nvalid = off < len ? (len - off < 16 ? len - off : 16u) : 0u;
                uint32_t flags = (acc >> 16) & ((1u << nvalid) - 1u);
                if (DBG && (P.dbg & 1)) flags = 0;                               // timing study: the filter alone
                if (q < 2) m0 |= flags << (16 * q); else m1 |= flags << (16 * (q - 2));
                if (last_piece) {
                    // the round is complete: its flagged positions go to queue A, the next round begins
                    psh0 = m0; psh1 = m1; m0 = m1 = 0;
                    fcnt = __popc(psh0) + __popc(psh1);
                    fincl = wave_incl_scan(fcnt);
                    push_rb = rb; push_C = C;
                    push_l0 = lane_value(fincl, 63) ? 0u : 64u;
                    rb = rb2; C = C2; q = 0;
                    have_round = rb < len;
                } else q++;
                mark(1);
            };
            for (;;) {
                // ---- one round, piece by piece: a trip of stage A is requested in front of a piece and decided behind it
                if (have_round) do {
                    if (!a_n && qa_n >= 128) stage_a_issue(128);
                    filter_piece();
                    if (a_n) {
                        while (qb_n > qb_lim) stage_b(qb_n < 64 ? qb_n : 64);        // (queue B has room for the up to 128 that stage A adds)
                        stage_a_complete();
                    }
                } while (q != 0);
                // ---- the round's flagged positions -> queue A; trips in between while they do not fit, and to the last
                // position when the stream is over
                while (push_l0 < 64 || (!have_round && (qa_n || a_n || qb_n))) {
                    if (push_l0 < 64) { push_some(); if (push_l0 >= 64 && have_round) break; }
                    if (qb_n > qb_lim || (qb_n && !a_n && !qa_n)) stage_b(qb_n < 64 ? qb_n : 64);
                    else if (a_n) stage_a_complete();
                    else if (qa_n) stage_a_issue(qa_n < 128 ? qa_n : 128);
                }
                if (!have_round) break;
            }

            phase = 7;
            flush();


            // ASCII folding is not strings.ToLower once the text leaves ASCII (finder.go:140-142): tell the host
            if (P.fold && P.nonascii && !told_nonascii && __any((hib & 0x80808080u) != 0)) {
                told_nonascii = true;
                if (lane == 0 && !(__hip_atomic_fetch_or(wg_next + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 31))
                    atomicOr(P.nonascii, 1u);
            }

            // ---- the units' records: region and count.  A unit that outgrew its region has nothing valid there: it goes on the
            // wave's list and is walked again, alone, with a region of the size counted
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t cur_l = lane < nu ? tab_at(meta, kTabCur, lane) : 0u;
            const bool over = lane < nu && cur_l > bound_l;
            if (lane < nu && !over) {
                const uint64_t u = u_first + lane;
                KARG(unit_start)[u] = chunk_base + (bincl - bound_l);
                KARG(unit_count)[u] = pool_ok ? cur_l : 0u;
            }
            {
                const uint64_t om = __ballot(over);                                  // (never on a second walk: its region is the count)
                if (over) {
                    const uint32_t at = n_redo + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0));
                    redo_u[at] = (uint32_t)(u_first + lane);
                    redo_c[at] = cur_l;
                }
                n_redo += (uint32_t)__popcll(om);
            }
            {
                uint32_t mine = over ? 0u : cur_l;
#pragma unroll
                for (int sh = 32; sh; sh >>= 1) mine += __shfl_xor(mine, sh, 64);
                wave_matches += mine;
            }
            mark(7);
        }
        if (!is_redo) {
            if (more_chunks) fetch_docs(nch, un_n, dabs_n, dlen_n);
            ch = nch;
        }
    }
    // the match count: summed in LDS, one global atomic per workgroup by the wave that finishes last
    if (lane == 0) {
        if (wave_matches)
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(wg_next + 2), (unsigned long long)wave_matches, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t done = __hip_atomic_fetch_add(wg_next + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) & 0x7FFFFFFFu;
        if (done + 1 == (blockDim.x >> 6)) {
            const unsigned long long all = __hip_atomic_load(reinterpret_cast<unsigned long long*>(wg_next + 2), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP);
            if (all) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(n_matches)), all);
        }
    }
    if (DBG && (P.dbg & 64) && lane == 0 && KARG(dbg_counters)) {
        unsigned long long all = 0;
        for (int ph = 0; ph < 8; ph++) { atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 4 + ph), tl[ph]); all += tl[ph]; }
        atomicMax(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 12), all);      // the wave that finishes last
        atomicAdd(reinterpret_cast<unsigned long long*>(KARG(dbg_counters) + 13), 1ull);
    }
}

}  // namespace

static size_t scan4_fixed_lds(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes) {
    return ((256 + (size_t)filter_words * 4 + short3_bytes + fpt_lds_bytes + (size_t)shorts_words * 4 + 15) & ~(size_t)15) + 16;
}

// waves per workgroup and the match fifo's capacity (entries) that fit lds_max; false if nothing fits
bool scan4_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max, bool want_pos,
                uint32_t* waves, uint32_t* fifo_cap) {
    const size_t fixed = scan4_fixed_lds(filter_words, short3_bytes, shorts_words, fpt_lds_bytes);
    const size_t queues = kS4MetaWords * 4 + s4_qa_cap(want_pos) * 2 + s4_qb_cap(want_pos) * 6, per_entry = want_pos ? 8 : 4;
    for (uint32_t w : {16u, 12u, 8u, 4u}) {
        if (fixed + (size_t)w * (queues + 192 * per_entry) > lds_max) continue;    // (a way of stage A appends up to 192 entries)
        const size_t per = ((lds_max - fixed) / w) & ~(size_t)15;
        size_t cap = (per - queues) / per_entry;
        cap = cap > 1024 ? 1024 : cap & ~(size_t)63;
        *waves = w;
        *fifo_cap = (uint32_t)cap;
        return true;
    }
    return false;
}

hipError_t launch_scan4(const Scan2Params& P, uint32_t waves, unsigned n_cus, hipStream_t st) {
    if (!P.n_units) return hipSuccess;
    const bool fl = P.fpt_lg == 0;
    const size_t lds = scan4_fixed_lds(P.filter_words, P.short3_bytes, P.shorts_words, fl ? kScan2FptSize : 0) +
                       (size_t)waves * (kS4MetaWords * 4 + s4_qa_cap(P.want_pos != 0) * 2 + s4_qb_cap(P.want_pos != 0) * 6 + (size_t)P.cand_cap * (P.want_pos ? 8 : 4));
    using Kern = void (*)(const Scan2Params);
    // timing studies (GFT_SCAN_DEBUG): the benchmark's shape only (direct filter, fingerprint table in LDS)
    const Kern fn = P.dbg && !P.hashed && fl ? k_scan4<false, true, true>
                    : P.hashed ? (fl ? k_scan4<true, true, false> : k_scan4<true, false, false>) : (fl ? k_scan4<false, true, false> : k_scan4<false, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t n_chunks = (P.n_units + P.chunk_units - 1) / P.chunk_units;
    uint64_t g = (n_chunks + waves - 1) / waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    fn<<<dim3(grid), dim3(waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
