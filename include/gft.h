/* gft.h -- C ABI of libgft.so: MI355X (gfx950) implementation of gofindthem's ProcessText hot path.
 *
 * Drop-in boundary for the reference's Go interface finder.SubstringEngine
 * (finder/substringEngine.go:11-18) and for the per-document solve loop of finder.Finder.ProcessText
 * (finder/finder.go:139-215).  Plain pointers and sizes only: a cgo shim binds these directly (see
 * INTEGRATION.md).  Paths cited below are relative to the reference repository.
 *
 * Conventions
 *   - every function returns GFT_OK (0) or a negative gft_status; gft_last_error() gives the message.
 *   - the caller owns its input buffers; the library never retains them after a call returns (cgo rule).
 *   - "blob + offsets": n byte strings are passed as one contiguous blob and n+1 uint64 offsets.
 *   - term ids index the engine's own dictionary order: unique terms sorted bytewise (the reference's
 *     DictIndex is Go-map-iteration order, i.e. meaningless across runs: substringEngine.go:99-104).
 *   - positions are byte offsets into the (already case-folded) text, like Match.Position
 *     (finder/finder.go:11-14), uint32 per document.
 *   - match order inside one document is the reference engine's emission order: end offset ascending,
 *     then term length descending (node first, then its dictionary-suffix chain).
 *   - handles are single-caller: do not use one gft_engine from two threads at once.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with GFT_E_HIP.
 */
#ifndef GFT_H
#define GFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gft_status {
    GFT_OK = 0,
    GFT_E_INVALID = -1,     /* bad argument */
    GFT_E_NOT_BUILT = -2,   /* gft_build / gft_set_programs has not been called */
    GFT_E_HIP = -3,         /* HIP runtime error (message has the HIP error string) */
    GFT_E_UNSUPPORTED = -4, /* input exceeds a documented limit of the device path */
    GFT_E_PARSE = -5,       /* DSL error; message is the reference parser's error text */
    GFT_E_ENGINE = -6       /* an injected engine (host mirror) reported an error */
} gft_status;

/* gft_build flags */
#define GFT_POS_START 0u /* Position = offset of the first byte of the match (default) */
#define GFT_POS_END 1u   /* Position = offset of the last byte of the match */
/* gft_scan / gft_process flags */
#define GFT_FOLD_ASCII 1u /* lower-case A-Z while reading the text (finder.go:140-142 for ASCII input) */

typedef struct gft_engine gft_engine;

/* ---- lifetime ------------------------------------------------------------------------------------- */
/* device = HIP device ordinal, or -1 for the calling thread's current device. */
int gft_engine_create(gft_engine** out, int device);
void gft_engine_destroy(gft_engine* e);
const char* gft_last_error(const gft_engine* e);
/* Run all work of this engine on an existing HIP stream (hipStream_t passed as void*); NULL = own stream. */
int gft_set_stream(gft_engine* e, void* hip_stream);

/* ---- SubstringEngine.BuildEngine (finder/substringEngine.go:98-106) ----------------------------------- */
/* Receives the full keyword set (already lower-cased by the DSL parser when case-insensitive,
 * dsl/parser.go:79-81).  Copies, sorts, de-duplicates, compiles the automaton and uploads it. */
int gft_build(gft_engine* e, const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, uint32_t flags);
uint32_t gft_n_terms(const gft_engine* e);  /* unique terms */
uint32_t gft_n_states(const gft_engine* e); /* automaton states incl. root */
/* term_id -> bytes of the term (pointer valid until the next gft_build / destroy) */
int gft_term(const gft_engine* e, uint32_t term_id, const uint8_t** ptr, uint32_t* len);
/* bytes of a term -> term_id, or -1 if it is not in the dictionary */
int64_t gft_term_id(const gft_engine* e, const uint8_t* term, uint32_t len);

/* ---- SubstringEngine.FindSubstrings (finder/substringEngine.go:110-119), batched ------------------------ */
/* CSR result: matches of document d are [match_off[d], match_off[d+1]).  Buffers are library-owned and stay
 * valid until the next call on the same engine. */
typedef struct gft_matches {
    uint64_t n_docs;
    uint64_t n_matches;
    const uint64_t* match_off; /* n_docs + 1 */
    const uint32_t* term_id;   /* n_matches */
    const uint32_t* pos;       /* n_matches */
} gft_matches;

/* Host buffers in, host buffers out (the cgo path).  One document == one FindSubstrings call. */
int gft_scan(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
             gft_matches* out);
/* Device-resident variant: text_blob / doc_off already live in HBM on the engine's device; the returned
 * pointers are DEVICE pointers (n_docs / n_matches are host values).  text_blob must be readable for 64 bytes
 * past doc_off[n_docs] (vector loads). */
int gft_scan_device(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                    uint32_t flags, gft_matches* out_dev);

/* ---- Expression programs: the solver half (dsl/expression.go:60-142, finder/finder.go:199-215) ---------- */
/* An expression is a postfix program of uint32 words, produced from the dsl.Expression tree by the host
 * side (gft_finder_* below, or the Go shim).  Word = opcode << 28 | operand.
 *   GFT_OP_UNIT  operand = slot.  Slots [0, n_terms) are dictionary term ids; slots [n_terms, n_terms+n_extra)
 *                are "extra" literals whose matches the caller supplies (regex terms, finder/regexEngine.go).
 *   GFT_OP_AND / GFT_OP_OR   binary;  GFT_OP_NOT unary;  GFT_OP_INORD unary (closes an INORD(...) group).
 *   operand bit 0 of AND/OR/UNIT-less ops is unused; UNIT/AND/OR words inside an INORD group carry
 *   GFT_INORD_FLAG (bit 27) == dsl.Expression.Inord. */
#define GFT_OP_UNIT 1u
#define GFT_OP_AND 2u
#define GFT_OP_OR 3u
#define GFT_OP_NOT 4u
#define GFT_OP_INORD 5u
#define GFT_INORD_FLAG (1u << 27)
#define GFT_SLOT_MASK ((1u << 27) - 1u)

int gft_set_programs(gft_engine* e, const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs,
                     uint32_t n_extra);
uint32_t gft_n_exprs(const gft_engine* e);

/* matches of the extra slots (regex engine output), CSR per document; slot is relative to n_terms */
typedef struct gft_extra_matches {
    const uint64_t* off; /* n_docs + 1 */
    const uint32_t* slot;
    const uint32_t* pos;
} gft_extra_matches;

/* Finder.ProcessText over a batch: hit_bitmap[d * words + (i >> 5)] bit (i & 31) == expression i is true for
 * document d, words = ceil(n_exprs / 32).  `extra` may be NULL. */
int gft_process(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                const gft_extra_matches* extra, uint32_t* hit_bitmap);
/* Device-resident variant (all pointers are device pointers, bitmap written in HBM). */
int gft_process_device(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                       uint32_t flags, const gft_extra_matches* d_extra, uint32_t* d_hit_bitmap);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by HIP events on the engine's stream. */
int gft_profile_enable(gft_engine* e, int on);
/* Sums since the last reset.  names: "scan", "solve", "aux".  Synchronises the stream. */
int gft_profile_read(gft_engine* e, const char* name, double* total_ms, uint64_t* launches);
int gft_profile_reset(gft_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* GFT_H */
