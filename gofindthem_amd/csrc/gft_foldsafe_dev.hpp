// gft_foldsafe_dev.hpp -- "is ASCII case folding the whole of strings.ToLower for this text?" (finder/finder.go:140-142
// lower-cases with strings.ToLower, the scan kernels fold A-Z only), decided INSIDE the scan kernels for the pieces of text
// that hold a byte >= 0x80.  The rule of k_fold_safe (gft_kernels.hip): accepted are ASCII and the two-byte sequences
// C2 80..BF (Latin-1 signs: no case) and C3 9F..BF / C3 97 (Latin-1 lower-case letters, the multiplication sign); anything
// else -- upper-case Latin-1, every other lead byte, a continuation byte out of place, a lead byte that nothing continues
// (Go rewrites invalid UTF-8 to U+FFFD) -- makes the batch unsafe: the finder then lower-cases it on the host.
//
// The filter loops note which 16-byte pieces hold a high byte (a handful per 4 KB of real text) in the wave's candidate list,
// which is idle until the filter is done; fold_jobs_begin / fold_jobs_finish then take them 64 at a time, one piece per lane.  Documents are
// judged one by one, as strings.ToLower sees them: a continuation byte at a document's start has no lead, a lead byte at a
// document's end has no continuation (a unit in the middle of a document looks at the bytes of its neighbours).
// Included inside namespace gft { namespace { ... } } of a scan kernel's .hip file.
#pragma once

struct __attribute__((packed, aligned(1))) FoldPiece { uint32_t x, y, z, w; };

// the high bytes of text[a, a + 16) that lie in [ulo, uhi) (blob offsets; [ulo, uhi) is one unit, doc_start: ulo begins a document)
// One noted piece in flight: its 16 bytes, the byte in front of them and the byte behind them -- three independent loads,
// issued together and looked at later (fold_piece_unsafe), so that the caller's next stage hides them.  (Plain scalars
// throughout: a struct handed on by pointer, or words indexed by a lane's own k, end up in scratch memory.)
#define FOLD_JOB_ARGS uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t jprev, uint32_t jnext, uint32_t jlim
#define FOLD_JOB_REFS uint32_t &w0, uint32_t &w1, uint32_t &w2, uint32_t &w3, uint32_t &jprev, uint32_t &jnext, uint32_t &jlim
#define FOLD_JOB_VARS(p) uint32_t p##w0 = 0, p##w1 = 0, p##w2 = 0, p##w3 = 0, p##prev = 0, p##next = 0, p##lim = 0
#define FOLD_JOB_PASS(p) p##w0, p##w1, p##w2, p##w3, p##prev, p##next, p##lim

// the high bytes of one piece.  jlim = bytes of the piece that lie in its unit (0: no piece) | what is left of the DOCUMENT
// from the piece's first byte on, at most 255, << 8; jprev = the byte in front of the piece (0 in front of a document)
__device__ __forceinline__ bool fold_piece_unsafe(FOLD_JOB_ARGS) {
    auto byte_at = [&](uint32_t k) {
        const uint32_t lo = k & 8 ? w2 : w0, hi = k & 8 ? w3 : w1;
        return ((k & 4 ? hi : lo) >> (8 * (k & 3))) & 0xFFu;
    };
    const uint32_t k_hi = jlim & 0xFFu, doc_left = jlim >> 8;
    auto nib = [](uint32_t w) { return (((w & 0x80808080u) >> 7) * 0x00204081u) >> 21 & 0xFu; };     // a dword's four high bits
    uint32_t m = nib(w0) | nib(w1) << 4 | nib(w2) << 8 | nib(w3) << 12;
    m &= 0xFFFFu >> (16 - k_hi);                                    // (k_hi = 0: 0xFFFF >> 16 = 0)
    bool bad = false;
    while (m) {
        const uint32_t k = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        const uint32_t b = byte_at(k), p = k ? byte_at(k - 1) : jprev;
        const bool p_lead = p == 0xC2u || p == 0xC3u;
        if (b == 0xC2u || b == 0xC3u) {
            const uint32_t nb = k + 1 < 16 ? byte_at(k + 1) : jnext;
            bad |= p_lead || (nb & 0xC0u) != 0x80u || k + 1 >= doc_left;     // (a lead byte that ends its document)
        } else if ((b & 0xC0u) == 0x80u) {
            bad |= !p_lead || (p == 0xC3u && !(b >= 0x9Fu || b == 0x97u));          // C3 80..9E: upper-case Latin-1
        } else {
            bad = true;
        }
    }
    return bad;
}

// lanes with `high` note piece `rel` (its offset from the unit's first byte); n = jobs so far (wave-uniform).  Jobs beyond
// `cap` are dropped and counted: the caller then reports the text as unchecked
__device__ __forceinline__ void fold_job_push(bool high, uint32_t rel, uint16_t* jobs, uint32_t cap, uint32_t& n) {
    const uint64_t m = __ballot(high);
    if (!m) return;
    if (high) {
        const uint32_t idx = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (idx < cap) jobs[idx] = (uint16_t)rel;
    }
    n += (uint32_t)__popcll(m);
}

// The noted pieces, 64 per trip: all but the last trip are judged here, the last one's loads are left in flight (the job
// variables, which need nothing else) for fold_jobs_finish.  unit_abs = blob offset of the unit's first byte, own = its
// length, doc_start: the unit begins its document, doc_end = blob offset of the first byte behind that document.
// Returns: a piece broke the rule
__device__ __forceinline__ bool fold_jobs_begin(const uint8_t* text, uint64_t doc_end, uint64_t unit_abs, uint32_t own, bool doc_start,
                                                const uint16_t* jobs, uint32_t n, FOLD_JOB_REFS) {
    const uint32_t lane = threadIdx.x & 63u;
    bool bad = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        if (i0) bad |= fold_piece_unsafe(w0, w1, w2, w3, jprev, jnext, jlim);
        const bool on = i0 + lane < n;
        const uint32_t rel = jobs[on ? i0 + lane : 0];
        const uint64_t a = unit_abs + rel;
        jlim = 0;
        if (on) {
            const FoldPiece v = *reinterpret_cast<const FoldPiece*>(text + a);         // (the blob is readable 64 bytes past its end: gft.h)
            w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
            jprev = rel || !doc_start ? text[a - 1] : 0u;
            jnext = text[a + 16];
            const uint64_t left = doc_end - a;
            jlim = (own - rel < 16 ? own - rel : 16u) | (left < 255 ? (uint32_t)left : 255u) << 8;
        }
    }
    __builtin_amdgcn_wave_barrier();                                // (the list may be overwritten now)
    return __any(bad);
}
__device__ __forceinline__ bool fold_jobs_finish(FOLD_JOB_ARGS) { return __any(fold_piece_unsafe(w0, w1, w2, w3, jprev, jnext, jlim)); }
