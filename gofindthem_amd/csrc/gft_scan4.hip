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
constexpr uint32_t kS4QaCap = 256;               // queue A: flagged positions (u16)
constexpr uint32_t kS4QbCap = 128;               // queue B: survivors (u16 position + u32 window key)

// the unit of every lane's position: tags run from the hint (the unit of an earlier position of the same queue) upwards
struct LaneUnit { uint32_t tag, ds, dlen, uend; };
__device__ __forceinline__ LaneUnit lane_unit(uint32_t p, bool on, uint32_t ds_v, uint32_t dlen_v, uint32_t uend_v, uint32_t nu, uint32_t& hint) {
    uint32_t t = hint;
    LaneUnit u{t, (uint32_t)__builtin_amdgcn_readlane(ds_v, t), (uint32_t)__builtin_amdgcn_readlane(dlen_v, t),
               (uint32_t)__builtin_amdgcn_readlane(uend_v, t)};
    bool first = true;
    while (t + 1 < nu) {
        const uint32_t ue = __builtin_amdgcn_readlane(uend_v, t);
        const bool past = on && p >= ue;
        if (first && !__any(on && !past)) hint = t + 1;      // every position is behind unit t: the next trip starts there too
        else first = false;
        if (!__any(past)) break;
        t++;
        const uint32_t ds = __builtin_amdgcn_readlane(ds_v, t), dl = __builtin_amdgcn_readlane(dlen_v, t), un = __builtin_amdgcn_readlane(uend_v, t);
        if (past) { u.tag = t; u.ds = ds; u.dlen = dl; u.uend = un; }
    }
    return u;
}

// the wave's match fifo: term | unit tag << 29 (and the position next to it when positions are wanted)
struct Fifo4 {
    uint32_t* term;
    uint32_t* pos;             // nullptr: presence only
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
    // per-wave LDS: [queue A: u16 x kS4QaCap][queue B positions: u16 x kS4QbCap][queue B keys: u32 x kS4QbCap]
    //               [fifo terms: u32 x fifo_cap][fifo positions: u32 x fifo_cap when positions are wanted]
    const uint32_t fcap = P.cand_cap;                                 // (scan4_plan: entries of the match fifo)
    const uint32_t wave_bytes = kS4QaCap * 2 + kS4QbCap * 6 + fcap * (P.want_pos ? 8u : 4u);
    uint8_t* wave_lds = wave_lds_all + (size_t)wave * wave_bytes;
    uint16_t* qa = reinterpret_cast<uint16_t*>(wave_lds);
    uint16_t* qbp = qa + kS4QaCap;
    uint32_t* qbk = reinterpret_cast<uint32_t*>(qbp + kS4QbCap);
    Fifo4 ff{qbk + kS4QbCap, P.want_pos ? qbk + kS4QbCap + fcap : nullptr, fcap, 0};
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

    for (; ch < n_chunks; ch = nch) {
        const Unit un = un_n;
        const uint64_t dabs = dabs_n;
        const uint32_t dlen = dlen_n;
        {
            uint32_t item = 0;
            if (lane == 0) item = __hip_atomic_fetch_add(wg_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nch = chunk_of((uint32_t)__builtin_amdgcn_readfirstlane(item));
        }
        const bool more_chunks = nch < n_chunks;
        if (more_chunks) fetch_units(nch, un_n);
        const uint64_t u_first = ch * cu;
        const uint32_t nu_all = (uint32_t)(P.n_units - u_first < cu ? P.n_units - u_first : cu);

        // A chunk is scanned as one stream if its units follow each other in the blob (units of consecutive documents do;
        // the empty units behind the real ones of a table that was sized blind do not): jobs = maximal runs of such units.
        // A unit that outgrew its region is walked again as a job of its own (`redo`).
        const uint64_t a_abs = dabs + un.lo, b_abs = dabs + un.hi;                  // this lane's unit in the blob
        uint32_t t0 = 0;
        uint32_t redo = 0;                       // bit t: unit t must be walked again; redo_n: with this many entries
        uint32_t redo_n = 0;                     // (lane t holds its unit's count)
        bool docs_fetched = false;
        while (t0 < nu_all || redo) {
            uint32_t j0, j1;
            bool is_redo = false;
            if (t0 < nu_all) {
                j0 = t0;
                // the run [j0, j1): a[t + 1] == b[t]
                const uint64_t a_next = (uint64_t)__shfl((unsigned long long)a_abs, (int)((lane + 1) & 63u), 64);
                const uint64_t brk = __ballot(lane >= j0 && lane + 1 < nu_all && a_next != b_abs);
                j1 = brk ? (uint32_t)__builtin_ctzll(brk) + 1 : nu_all;
                t0 = j1;
            } else {
                j0 = (uint32_t)__builtin_ctz(redo);
                j1 = j0 + 1;
                redo &= redo - 1;
                is_redo = true;
            }
            const uint32_t nu = j1 - j0;                                             // units of this job (tags 0 .. nu-1)
            const uint64_t s_abs = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(a_abs >> 32), j0) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)a_abs, j0);
            const uint64_t e_abs = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(b_abs >> 32), j1 - 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)b_abs, j1 - 1);
            const uint32_t len = (uint32_t)(e_abs - s_abs);                          // bytes of the stream (< 2^16)
            const uint64_t base = s_abs >= kS4Base ? s_abs - kS4Base : 0;            // positions p = blob offset - base
            const uint32_t s_p = (uint32_t)(s_abs - base), e_p = s_p + len;
            // per-unit values, unit j0 + t in lane t: start of its document (mod 2^32: only differences are used), the
            // document's length, the unit's end, the size of its region
            const uint32_t src_l = (lane + j0) & 63u;
            const uint32_t ds_v = (uint32_t)(__shfl((unsigned long long)dabs, (int)src_l, 64) - base);
            const uint32_t dlen_v = __shfl(dlen, (int)src_l, 64);
            const uint32_t uend_v = (uint32_t)(__shfl((unsigned long long)b_abs, (int)src_l, 64) - base);
            const uint32_t lo0 = __builtin_amdgcn_readlane(un.lo, j0);
            const uint32_t ubytes = __shfl(un.hi - un.lo, (int)src_l, 64);
            uint32_t bound_v = lane < nu ? (uint32_t)(((uint64_t)ubytes * P.bound_q16) >> 16) + P.bound_add : 0u;
            if (is_redo) bound_v = lane == 0 ? (uint32_t)__builtin_amdgcn_readlane(redo_n, j0) : 0u;
            // regions: one behind the other in the wave's slab (a fresh slab -- of the chunk's size, if that is larger -- when
            // what is left does not hold them all)
            const uint32_t bincl = wave_incl_scan(bound_v);
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
            const uint32_t reg_v = bincl - bound_v;                                  // region of unit t: chunk_base + reg_v[t]
            uint32_t cur_v = 0;                                                      // lane t: matches of unit t so far

            const Ctx c{P, cls, filt, P.short3_bytes ? short3 : nullptr, fpt, lrec, P.text + base, base, kp2,
                        base < 7, base < 23, s_p, e_p, base + e_p + 4 > P.text_bytes, 0u};

            // ---- the match fifo.  flush() appears at three places only (each stage checks for room ONCE, in front of its
            // appends); the stages themselves appear once each: the job is a loop over one state machine ------------------------
            auto flush = [&]() {
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
                        const uint32_t cur = __builtin_amdgcn_readlane(cur_v, t), bnd = __builtin_amdgcn_readlane(bound_v, t);
                        const uint32_t reg = __builtin_amdgcn_readlane(reg_v, t);
                        if (on && tag == t) {
                            const uint32_t idx = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                            if (idx < bnd && pool_ok) {
                                const uint64_t at = chunk_base + reg + idx;
                                __builtin_nontemporal_store(e & 0x1FFFFFFFu, &KARG(pool_term)[at]);
                                if (ff.pos) __builtin_nontemporal_store(ps, &KARG(pool_pos)[at]);
                            }
                        }
                        // (one scalar operand per instruction on gfx9: the lane select goes through m0)
                        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(cur_v) : "s"(cur + (uint32_t)__popcll(m)), "s"(t));
                    }
                }
                ff.n = 0;
                __builtin_amdgcn_wave_barrier();
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

            uint32_t qa_n = 0, qb_n = 0, qa_hint = 0, qb_hint = 0;
            // the trip of stage A that is in flight: positions and their text (a_n = 0: none)
            bool a_on[kStageAWays] = {false, false};
            uint32_t a_p[kStageAWays] = {0, 0};
            Text8 a_tx[kStageAWays] = {{0, 0}, {0, 0}};
            uint32_t a_n = 0;

            // ---- the positions in front of the stream: a term whose window ends up to kScan2MaxOff bytes before it may end inside
            if (lo0) {
                const uint32_t nb = lo0 < kScan2MaxOff ? lo0 : kScan2MaxOff;
                if (lane < nb) qa[lane] = (uint16_t)(s_p - nb + lane);
                qa_n = nb;
            }

            // ---- the stream -------------------------------------------------------------------------------------------
            const uint32_t nr = (len + 1023) >> 10;
            const uint8_t* src = P.text + s_abs + lane * 16;
            U128u nxt{0, 0, 0, 0};
            if (lane * 16 < len) nxt = *reinterpret_cast<const U128u*>(src);
            // the rolling key's state in front of the stream: the classes of the three bytes before it (the padding class
            // where the blob starts)
            uint32_t car_cp, car_pm1, car_pm2;
            {
                uint32_t k1 = P.pad_class, k2 = P.pad_class, k3 = P.pad_class;
                if (s_abs >= 1) k1 = lcls[P.text[s_abs - 1]];
                if (s_abs >= 2) k2 = lcls[P.text[s_abs - 2]];
                if (s_abs >= 3) k3 = lcls[P.text[s_abs - 3]];
                car_cp = __builtin_amdgcn_readfirstlane(k1);
                car_pm1 = __builtin_amdgcn_readfirstlane(mad24s(k2, kp, k1));
                car_pm2 = __builtin_amdgcn_readfirstlane(mad24s(k3, kp, k2));
            }
            uint32_t hib = 0;
            uint32_t r = 0;                      // the next round to filter
            // flagged positions of the last filtered round that are not in queue A yet: lanes [push_l0, 64)
            uint32_t flags = 0, fcnt = 0, fincl = 0, foff = 0, push_l0 = 64;
            for (;;) {
                const bool pushing = push_l0 < 64;
                const bool final = r == nr && !pushing;                              // nothing more will enter queue A
                // ---- (a) STAGE B: 64 survivors wait (or what is left, at the end) -- and always before stage A adds up to
                // 128 more, so that queue B never overflows
                if (qb_n >= 64 || (qb_n && final && !qa_n && !a_n)) {
                    const uint32_t n = qb_n < 64 ? qb_n : 64;
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
                    const Slot s0 = slot_load(&P.slots[scan2_slot_hash(x, 0, P.slot_shift, P.slot_seed)]);
                    const Slot s1 = slot_load(&P.slots[scan2_slot_hash(x, 1, P.slot_shift, P.slot_seed)]);
                    const Text8 t8 = cand_load(c, p);
                    Front t = front_load(c, p, t8.tw);
                    const uint32_t tl = tail_load(c, p);
                    const LaneUnit lu = lane_unit(p, on, ds_v, dlen_v, uend_v, nu, qb_hint);
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
                    continue;
                }
                // ---- (b) STAGE A, second half: the decisions of the trip whose text was requested before the last round
                if (a_n) {
                    if (P.prio) __builtin_amdgcn_s_setprio(2);
                    Cand k[kStageAWays];
                    LaneUnit lu[kStageAWays];
                    uint32_t rec[kStageAWays][3];
                    uint32_t pd[kStageAWays];
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) cand_keys(c, a_p[q], a_tx[q], k[q]);
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) cand_decide<FPT_LDS>(c, k[q]);
                    uint32_t n_short = 0;
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        rec[q][0] = rec[q][1] = rec[q][2] = 0;
                        pd[q] = 0;
                        lu[q] = LaneUnit{0, 0, 0, 0};
                        if (64u * q >= a_n) continue;
                        lu[q] = lane_unit(a_p[q], a_on[q], ds_v, dlen_v, uend_v, nu, qa_hint);
                        pd[q] = a_p[q] - lu[q].ds;
                        // terms of length <= 3 ending here (not at the positions in front of the stream)
                        const uint32_t sid = a_on[q] && a_p[q] >= s_p ? k[q].sid : 0;
                        if (__any(sid != 0)) {
                            if (sid) short_record(c, sid, k[q].x3, rec[q]);
#pragma unroll
                            for (uint32_t j = 0; j < 3; j++) {
                                if (rec[q][j] && (rec[q][j] >> 28) > pd[q] + 1) rec[q][j] = 0;       // (it would start before its document)
                                n_short += (uint32_t)__popcll(__ballot(rec[q][j] != 0));
                            }
                        }
                    }
                    if (ff.n + n_short > ff.cap) flush();
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
#pragma unroll
                        for (uint32_t j = 0; j < 3; j++) {
                            if (!__any(rec[q][j] != 0)) continue;
                            const uint32_t L = rec[q][j] >> 28;
                            append(rec[q][j] != 0, rec[q][j] & 0x0FFFFFFFu, lu[q].tag, P.pos_end ? pd[q] : pd[q] + 1 - L);
                        }
                    }
                    // survivors -> queue B with their window keys (it holds fewer than 64: stage B goes first)
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        if (64u * q >= a_n) continue;
                        const bool keep = a_on[q] && k[q].go_long;
                        const uint64_t sb = __ballot(keep);
                        if (keep) {
                            const uint32_t idx = qb_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0));
                            qbp[idx] = (uint16_t)a_p[q];
                            qbk[idx] = k[q].x;
                        }
                        qb_n += (uint32_t)__popcll(sb);
                    }
                    a_n = 0;
                    if (P.prio) __builtin_amdgcn_s_setprio(0);
                }
                // ---- (c) flagged positions of the last round -> queue A, as many lanes as fit
                if (pushing) {
                    const uint32_t before = push_l0 ? lane_value(fincl, push_l0 - 1) : 0;
                    const uint32_t room = kS4QaCap - qa_n;
                    const bool fits = lane >= push_l0 && fincl - before <= room;
                    const uint64_t fm = __ballot(fits) >> push_l0;
                    const uint32_t nl = fm == ~0ull >> push_l0 ? 64 - push_l0 : (uint32_t)__builtin_ctzll(~fm);   // lanes that fit (may be 0)
                    if (nl) {
                        const uint32_t l1 = push_l0 + nl;
                        const uint32_t ptotal = lane_value(fincl, l1 - 1) - before;
                        if (lane >= push_l0 && lane < l1) {
                            uint32_t wpos = qa_n + fincl - fcnt - before;
                            const uint32_t p0 = s_p + foff;
                            uint32_t mk = flags;
                            while (mk) {
                                const uint32_t i = __builtin_ctz(mk);
                                mk &= mk - 1;
                                qa[wpos++] = (uint16_t)(p0 + i);
                            }
                        }
                        qa_n += ptotal;
                        push_l0 = l1;
                    }
                }
                const bool still_pushing = push_l0 < 64;
                // ---- (d) STAGE A, first half: 128 positions wait (or the queue is full, or the stream is over): their text is
                // requested here and consumed at (b), behind the next round's filter
                if (!a_n && (qa_n >= 128 || (qa_n && (still_pushing || r == nr)))) {
                    const uint32_t n = qa_n < 128 ? qa_n : 128;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) {
                        const uint32_t i = 64 * q + lane;
                        a_on[q] = i < n;
                        a_p[q] = qa[a_on[q] ? i : 0];
                    }
#pragma unroll
                    for (int q = 0; q < kStageAWays; q++) a_tx[q] = cand_load(c, a_p[q]);
                    // what is left of the queue moves to its front
                    const uint32_t left = qa_n - n;
                    uint32_t mv[4];
#pragma unroll
                    for (int h = 0; h < 4; h++) { const uint32_t i = n + 64 * h + lane; mv[h] = qa[i < qa_n ? i : 0]; }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int h = 0; h < 4; h++) { const uint32_t i = 64 * h + lane; if (i < left) qa[i] = (uint16_t)mv[h]; }
                    qa_n = left;
                    a_n = n;
                }
                // ---- (e) FILTER: the next round
                if (!still_pushing && r < nr) {
                    const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
                    const uint32_t off = r * 1024 + lane * 16;                       // this lane's piece inside the stream
                    if (off < len) hib |= (w[0] | w[1]) | (w[2] | w[3]);            // (up to 15 bytes behind the stream: conservative)
                    if (r + 1 < nr && off + 1024 < len) nxt = *reinterpret_cast<const U128u*>(src + (size_t)(r + 1) * 1024);
                    // classes, pairs pair(i) = class(i-1) * kp + class(i), keys x(i) = pair(i-2) * kp^2 + pair(i)
                    uint32_t cl[16], pr[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) cl[i] = lcls[(w[i >> 2] >> (8 * (i & 3))) & 0xFF];
#pragma unroll
                    for (int i = 1; i < 16; i++) pr[i] = mad24s(cl[i - 1], kp, cl[i]);
                    // the state behind the previous lane's piece (lane 0: behind the previous round)
                    const uint32_t p_cp = (uint32_t)__builtin_amdgcn_update_dpp((int)car_cp, (int)cl[15], 0x138, 0xF, 0xF, false);     // wave_shr:1
                    const uint32_t p_pm1 = (uint32_t)__builtin_amdgcn_update_dpp((int)car_pm1, (int)pr[15], 0x138, 0xF, 0xF, false);
                    const uint32_t p_pm2 = (uint32_t)__builtin_amdgcn_update_dpp((int)car_pm2, (int)pr[14], 0x138, 0xF, 0xF, false);
                    car_cp = __builtin_amdgcn_readlane(cl[15], 63);
                    car_pm1 = __builtin_amdgcn_readlane(pr[15], 63);
                    car_pm2 = __builtin_amdgcn_readlane(pr[14], 63);
                    pr[0] = mad24s(p_cp, kp, cl[0]);
                    uint32_t acc = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t before = i == 0 ? p_pm2 : i == 1 ? p_pm1 : pr[i - 2];
                        const uint32_t x = mad24s(before, kp2, pr[i]);
                        const uint32_t fi = HASHED ? (x * kGoldDev) >> P.hash_shift : x;
                        const uint32_t fw = lfilt[fi >> 5];
                        acc = __builtin_amdgcn_alignbit(fw >> (fi & 31), acc, 1);
                    }
                    const uint32_t nvalid = off < len ? (len - off < 16 ? len - off : 16u) : 0u;
                    flags = (acc >> 16) & ((1u << nvalid) - 1u);
                    fcnt = __popc(flags);
                    fincl = wave_incl_scan(fcnt);
                    foff = off;
                    push_l0 = lane_value(fincl, 63) ? 0u : 64u;
                    r++;
                    continue;
                }
                if (r == nr && !still_pushing && !a_n && !qa_n && !qb_n) break;
            }
            flush();


            // ASCII folding is not strings.ToLower once the text leaves ASCII (finder.go:140-142): tell the host
            if (P.fold && P.nonascii && !told_nonascii && __any((hib & 0x80808080u) != 0)) {
                told_nonascii = true;
                if (lane == 0 && !(__hip_atomic_fetch_or(wg_next + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 31))
                    atomicOr(P.nonascii, 1u);
            }

            // ---- the units' records: region and count.  A unit that outgrew its region has nothing valid there: it is walked
            // again, alone, with a region of the size counted
            const bool over = lane < nu && cur_v > bound_v;
            if (lane < nu && !over) {
                const uint64_t u = u_first + j0 + lane;
                KARG(unit_start)[u] = chunk_base + reg_v;
                KARG(unit_count)[u] = pool_ok ? cur_v : 0u;
            }
            if (!is_redo) {
                const uint64_t om = __ballot(over);
                if (om) {
                    redo |= (uint32_t)om << j0;
                    // (unit j0 + t's count into lane j0 + t)
                    const uint32_t moved = __shfl(cur_v, (int)((lane - j0) & 63u), 64);
                    if (lane >= j0 && lane < j1 && ((om >> (lane - j0)) & 1)) redo_n = moved;
                }
            }
            {
                uint32_t mine = lane < nu && !over ? cur_v : 0u;
#pragma unroll
                for (int s = 32; s; s >>= 1) mine += __shfl_xor(mine, s, 64);
                wave_matches += mine;
            }
            if (more_chunks && !docs_fetched) { fetch_docs(nch, un_n, dabs_n, dlen_n); docs_fetched = true; }
        }
        if (more_chunks && !docs_fetched) fetch_docs(nch, un_n, dabs_n, dlen_n);
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
}

}  // namespace

static size_t scan4_fixed_lds(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes) {
    return ((256 + (size_t)filter_words * 4 + short3_bytes + fpt_lds_bytes + (size_t)shorts_words * 4 + 15) & ~(size_t)15) + 16;
}

// waves per workgroup and the match fifo's capacity (entries) that fit lds_max; false if nothing fits
bool scan4_plan(uint32_t filter_words, uint32_t short3_bytes, uint32_t shorts_words, uint32_t fpt_lds_bytes, size_t lds_max, bool want_pos,
                uint32_t* waves, uint32_t* fifo_cap) {
    const size_t fixed = scan4_fixed_lds(filter_words, short3_bytes, shorts_words, fpt_lds_bytes);
    const size_t queues = kS4QaCap * 2 + kS4QbCap * 6, per_entry = want_pos ? 8 : 4;
    for (uint32_t w : {16u, 12u, 8u, 4u}) {
        if (fixed + (size_t)w * (queues + 128 * per_entry) > lds_max) continue;
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
                       (size_t)waves * (kS4QaCap * 2 + kS4QbCap * 6 + (size_t)P.cand_cap * (P.want_pos ? 8 : 4));
    using Kern = void (*)(const Scan2Params);
    const Kern fn = P.hashed ? (fl ? k_scan4<true, true, false> : k_scan4<true, false, false>) : (fl ? k_scan4<false, true, false> : k_scan4<false, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t n_chunks = (P.n_units + P.chunk_units - 1) / P.chunk_units;
    uint64_t g = (n_chunks + waves - 1) / waves;
    const unsigned grid = (unsigned)(g < n_cus ? (g ? g : 1) : n_cus);
    fn<<<dim3(grid), dim3(waves * 64), lds, st>>>(P);
    return hipGetLastError();
}

}  // namespace gft
