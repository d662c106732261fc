/* gft.h -- C ABI of libgft.so: MI355X (gfx950) implementation of gofindthem's ProcessText hot path.
 *
 * Drop-in boundary for the reference's Go interface finder.SubstringEngine
 * (finder/substringEngine.go:11-18) and for the per-document solve loop of finder.Finder.ProcessText
 * (finder/finder.go:139-215).  Plain pointers and sizes only: a cgo shim binds these directly (see
 * INTEGRATION.md).  Paths cited below are relative to the reference repository.
 *
 * Conventions
 *   - every function returns GFT_OK (0) or a negative gft_status; gft_last_error() gives the message.  (One positive
 *     value exists, GFT_W_NO_RCCL, a warning of gft_engine_create_multi.)  No C++ exception ever crosses this boundary:
 *     every entry point translates whatever its host code throws into GFT_E_NOMEM / GFT_E_INTERNAL.
 *   - the caller owns its input buffers; the library never retains them after a call returns (cgo rule).
 *   - "blob + offsets": n byte strings are passed as one contiguous blob and n+1 uint64 offsets.
 *   - term ids index the engine's own dictionary order: unique terms sorted bytewise (the reference's
 *     DictIndex is Go-map-iteration order, i.e. meaningless across runs: substringEngine.go:99-104).
 *   - positions are byte offsets into the (already case-folded) text, like Match.Position
 *     (finder/finder.go:11-14), uint32 per document.
 *   - match order inside one document is the reference engine's emission order: end offset ascending,
 *     then term length descending (node first, then its dictionary-suffix chain).
 *   - every entry point takes the handle's own mutex: a built engine / finder may be shared between threads (goroutines),
 *     calls on one handle are serialised; result buffers the library owns (gft_scan, gft_finder_expression ...) stay valid
 *     until the NEXT call on the same handle, so a caller that shares a handle copies them before releasing its own lock.
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
    GFT_E_ENGINE = -6,      /* an injected engine (host mirror) reported an error */
    GFT_E_NOMEM = -7,       /* the host side ran out of memory (std::bad_alloc / a container beyond max_size) */
    GFT_E_INTERNAL = -8,    /* any other C++ exception on the host side: caught at the ABI, never propagated (the reference
                             * returns errors, it does not panic: finder/finder.go:149-158) */
    GFT_W_NO_RCCL = 1       /* gft_engine_create_multi only, a WARNING: the handle is valid and complete, but RCCL could not
                             * be loaded or ncclCommInitAll failed (gft_last_error says why) -- gft_process_device_multi
                             * gathers the bitmaps with device-to-device copies instead of ncclSend / ncclRecv */
} gft_status;

/* gft_build flags */
#define GFT_POS_START 0u /* Position = offset of the first byte of the match (default) */
#define GFT_POS_END 1u   /* Position = offset of the last byte of the match */
/* gft_scan / gft_process flags */
#define GFT_FOLD_ASCII 1u /* lower-case A-Z while reading the text (finder.go:140-142 for ASCII input).  This IS
                           * strings.ToLower only while the text is ASCII: the kernels notice bytes >= 0x80 on their way
                           * (gft_last_nonascii), and the finder entry points then fold such a batch on the host instead */

#define GFT_SCAN_UNIQUE 2u /* gft_scan / gft_scan_device: CloudflareEngine's output instead of CloudflareForkEngine's
                            * (finder/substringEngine.go:77-86): every term that occurs in a document once, in the order
                            * of its first occurrence, every pos 0 -- enough for expressions without INORD */

#define GFT_POS_RUNES 4u   /* gft_scan / gft_scan_device: AnknownEngine's positions (finder/substringEngine.go:44-53 searches
                            * []rune(text) and reports m.Pos): every pos is the number of RUNES in front of the match instead of
                            * the number of bytes, under Go's decoder (an invalid byte is one U+FFFD of width 1).  The match SET
                            * is the byte-level one; GFT_POS_START engines only; ignored together with GFT_SCAN_UNIQUE (pos 0) */

typedef struct gft_engine gft_engine;

/* ---- lifetime ------------------------------------------------------------------------------------- */
/* device = HIP device ordinal, or -1 for the calling thread's current device. */
int gft_engine_create(gft_engine** out, int device);
void gft_engine_destroy(gft_engine* e);
const char* gft_last_error(const gft_engine* e);
/* Run all work of this engine on an existing HIP stream (hipStream_t passed as void*).  NULL = a stream of the
 * engine's own, created blocking, i.e. ordered with the legacy default stream: device buffers produced there (torch's
 * default stream, plain hipMemcpy) can be handed to the *_device entry points without an explicit synchronisation.
 * Every entry point returns after its work has completed. */
int gft_set_stream(gft_engine* e, void* hip_stream);
/* Leave `margin` compute units free of this engine's kernels (default 0; environment: GFT_CU_MARGIN).  The scan and solver
 * kernels are persistent -- one workgroup per CU that holds the CU's whole LDS for the length of the launch --, so a kernel
 * of somebody else's (RCCL's send / receive kernels when the bitmap gather of batch i travels beside batch i + 1, bench.py
 * N > 1) finds no CU until they exit and then keeps the next launch's workgroups waiting.  With a margin those kernels
 * always find room; the batch's kernels run on n_cus - margin CUs.  Not while batches are in flight. */
int gft_set_cu_margin(gft_engine* e, uint32_t margin);

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

/* Compiled tables of the current dictionary as one blob (compiling a 100 k-term dictionary costs
 * ~0.6 s of host time; a blob is installed with a copy and an upload).  gft_export_tables writes into out (cap bytes) and
 * the size into *needed (GFT_E_INVALID when cap is too small); gft_import_tables is equivalent to the gft_build call that
 * produced the blob (same terms, ids and flags).  Blobs are tied to the library version that wrote them. */
int gft_export_tables(const gft_engine* e, uint8_t* out, uint64_t cap, uint64_t* needed);
int gft_import_tables(gft_engine* e, const uint8_t* blob, uint64_t len);

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

/* Limits of the DEVICE solver that the reference does not have: an INORD group with more than 64 leaves-with-thresholds
 * alive at once or nested deeper than 32, an operand stack deeper than 128 (left-deep chains of any length are fine: they
 * need no stack).  An expression beyond them is accepted all the same: gft_process* solve it on the host from the scan's
 * matches (csrc/host_solve.cpp, the reference's recursion restated -- dsl/expression.go:66-142 has no such limits) and put
 * its bit into the bitmap; every other expression of the set still runs on the device.
 * Refused with GFT_E_UNSUPPORTED: 2^25 slots or more (terms + extra literals; the solver's program words carry a 25-bit
 * slot), keywords longer than 7 424 bytes (gft_build). */
int gft_set_programs(gft_engine* e, const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs,
                     uint32_t n_extra);
uint32_t gft_n_exprs(const gft_engine* e);
/* how many of them the HOST solves for every document (host_solve: the reference's recursion, dsl/expression.go:66-142): the
 * ones beyond the device solver's limits -- an INORD group of more than 8 192 (slot, threshold) pairs alive at once or a
 * pair stack deeper than 64, a fused form that nests deeper than 128.  0 for any rule set a person would write. */
uint32_t gft_n_host_exprs(const gft_engine* e);
/* 1 when the last scan / process call on this engine ran with GFT_FOLD_ASCII over text for which lower-casing A-Z is
 * not provably the whole of strings.ToLower (finder/finder.go:140-142): it holds bytes >= 0x80 other than the two-byte
 * sequences C2 80..BF and C3 9F..BF / C3 97 (Latin-1 signs and LOWER-case letters) -- i.e. possibly an upper-case
 * non-ASCII letter, a rune whose lower-case form has another length, or invalid UTF-8.  The scan kernels notice high
 * bytes on their way; only such a batch pays one more pass over its text for this answer.
 * gft_finder_process_device checks it and repeats an unsafe batch through the host's ToLower. */
int gft_last_nonascii(const gft_engine* e);
/* which scan kernel the built dictionary runs on: "scan5" (suffix-window kernel, one filter probe per two bytes: the
 * default wherever the long-term tables exist), "scan3" (stride-2 suffix-window kernel, any alphabet: the fallback) or
 * "dfa" (general two-tier DFA kernel: only when forced with GFT_SCAN_KERNEL); "scan2" / "scan4" (the earlier
 * suffix-window kernels) only in a library built with GFT_EXTRA_KERNELS (tools/, the opt-in cross-check job) */
const char* gft_scan_kernel(const gft_engine* e);
/* how this library was built: "gfx950 extra_kernels=0|1" (1: the cross-check kernels scan2 / scan4 are compiled in) */
const char* gft_build_info(void);

/* Caller-supplied matches (regex engine output, or the output of a foreign SubstringEngine), CSR per document.
 * `slot` is ABSOLUTE: n_terms + j for extra literal j, or a dictionary term id when a regex literal has the same
 * text as a keyword (both feed one map key in the reference, finder/finder.go:181-196).  Positions of one slot
 * must be ascending within a document (README.md:155). */
typedef struct gft_extra_matches {
    const uint64_t* off; /* n_docs + 1 */
    const uint32_t* slot;
    const uint32_t* pos;
} gft_extra_matches;

/* Finder.ProcessText over a batch: hit_bitmap[d * words + (i >> 5)] bit (i & 31) == expression i is true for
 * document d, words = ceil(n_exprs / 32).  `extra` may be NULL. */
int gft_process(gft_engine* e, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                const gft_extra_matches* extra, uint32_t* hit_bitmap);
/* Solve again over the documents of the LAST gft_process call on this engine (same n_docs), with other caller-supplied
 * matches: the scan results are still in the engine, only the solver kernel runs.  The finder's regex prefilter uses it:
 * first pass without regex hits, host regex engine on the candidate documents only, second pass with their hits. */
int gft_process_again(gft_engine* e, uint64_t n_docs, const gft_extra_matches* extra, uint32_t* hit_bitmap);
/* Device-resident variant of gft_process (all pointers are device pointers, bitmap written in HBM).  As for
 * gft_scan_device, d_text_blob must be readable for 64 bytes past doc_off[n_docs] (vector loads); documents of 4 GiB
 * and more, and offsets that descend, are refused with GFT_E_INVALID. */
int gft_process_device(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                       uint32_t flags, const gft_extra_matches* d_extra, uint32_t* d_hit_bitmap);
/* The same, pipelined: _begin enqueues the batch (units -> scan -> solve -> the read-back of its control block) on the
 * engine's stream and returns WITHOUT waiting; _end completes the oldest batch begun and returns ITS status (and sets
 * gft_last_nonascii for it).  At most two batches are in flight, so a caller that begins batch i + 1 before it ends
 * batch i keeps the device busy while the host reads batch i's verdict and launches the next one -- what a step of 0.5 ms
 * (125 000 documents: one GPU's share of 1 M over 8) needs.  Inputs and the bitmap of a batch must stay untouched until its
 * _end has returned: a batch that outgrew the engine's unit table or match pool is run again there.  A batch that cannot be
 * deferred (caller-supplied matches, host-solved expressions, an engine's first batches) completes inside _begin; _end
 * then only hands its status back.  Single-device handles; no other entry point of the handle between _begin and _end. */
int gft_process_device_begin(gft_engine* e, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                             uint32_t flags, const gft_extra_matches* d_extra, uint32_t* d_hit_bitmap);
int gft_process_device_end(gft_engine* e);

/* ---- finder.Finder mirror (finder/finder.go:32-240) ---------------------------------------------------------
 * Host-side orchestration with the reference's semantics: expression registry, keyword / regex sets, lazy engine
 * build with the same dirty flags (incl. the ForceBuild quirk, finder.go:218-235), error propagation, results in
 * registration order.  The DSL parser inside is dsl/parser.go + dsl/scanner.go restated (same trees, same error
 * strings).  All solving happens on the GPU (gft_process); text case folding follows strings.ToLower.
 * The substring engine defaults to the built-in GPU engine; foreign engines (any SubstringEngine / RegexEngine
 * implementation, e.g. Go's regexp behind RegexpEngine, or test mocks) are injected as callbacks. */
typedef struct gft_finder gft_finder;

/* emit one Match{Position, Term} (finder/finder.go:11-14) */
typedef void (*gft_emit_fn)(void* sink, const uint8_t* term, uint32_t term_len, int64_t position);
/* BuildEngine(keywords|regexes, caseSensitive): return 0, or non-zero with a NUL-terminated message in err */
typedef int (*gft_engine_build_fn)(void* user, const uint8_t* blob, const uint64_t* off, uint32_t n,
                                   int case_sensitive, char* err, uint32_t err_cap);
/* FindSubstrings / FindRegexes(text): call emit(sink, ...) per match; return 0, or non-zero with message */
typedef int (*gft_engine_find_fn)(void* user, const uint8_t* text, uint64_t text_len, gft_emit_fn emit, void* sink,
                                  char* err, uint32_t err_cap);

int gft_finder_create(gft_finder** out, int case_sensitive, int device); /* NewFinder(GpuEngine, EmptyRgxEngine, cs) */
/* the same finder over several devices (gft_engine_create_multi): finder.NewFinder(&GpuEngine{Devices: ...}, ...) */
int gft_finder_create_multi(gft_finder** out, int case_sensitive, const int* devices, int n_devices);
void gft_finder_destroy(gft_finder* f);
const char* gft_finder_last_error(const gft_finder* f);
gft_engine* gft_finder_engine(gft_finder* f); /* the GPU engine handle used for scanning/solving */
int gft_finder_set_substring_engine(gft_finder* f, gft_engine_build_fn build, gft_engine_find_fn find, void* user);
int gft_finder_set_regex_engine(gft_finder* f, gft_engine_build_fn build, gft_engine_find_fn find, void* user);
/* AddExpressionWithTag (finder.go:115-134).  GFT_E_PARSE + the reference's error text on malformed input. */
int gft_finder_add_expression(gft_finder* f, const uint8_t* expr, uint64_t expr_len, const uint8_t* tag,
                              uint64_t tag_len);
uint32_t gft_finder_n_expressions(const gft_finder* f);
/* which: 0 = keywords, 1 = regexes.  Returns the set size; item i via gft_finder_literal. */
uint32_t gft_finder_n_literals(const gft_finder* f, int which);
int gft_finder_literal(const gft_finder* f, int which, uint32_t i, const uint8_t** ptr, uint32_t* len);
/* expression i: its source string, tag and parsed tree as JSON ({"Type":..,"LExpr":..}); pointers valid until
 * the next call on f */
int gft_finder_expression(const gft_finder* f, uint32_t i, const uint8_t** str, uint32_t* str_len,
                          const uint8_t** tag, uint32_t* tag_len, const uint8_t** tree_json, uint32_t* json_len);
int gft_finder_force_build(gft_finder* f);                       /* ForceBuild (finder.go:218-235) */
/* ProcessText (finder.go:139-179): indices of the expressions that are true, in registration order. */
int gft_finder_process_text(gft_finder* f, const uint8_t* text, uint64_t text_len, uint32_t* out_idx, uint32_t cap,
                            uint32_t* n_true);
/* Batch extension: one bitmap row per document (layout as gft_process). */
int gft_finder_process_texts(gft_finder* f, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs,
                             uint32_t* hit_bitmap);
/* documents the host regex engine was called on by the last gft_finder_process_texts (the regex prefilter, SURVEY.md
 * 8(f) #3, sends it only documents that contain every required literal of some regex; GFT_REGEX_PREFILTER=0 disables) */
uint64_t gft_finder_last_regex_docs(const gft_finder* f);
/* Same with the corpus resident in HBM (GPU substring engine, no regex terms). */
int gft_finder_process_device(gft_finder* f, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                              uint32_t* d_hit_bitmap);
/* ... pipelined (gft_process_device_begin / _end): _end also repeats a batch that left ASCII through the host's ToLower */
int gft_finder_process_device_begin(gft_finder* f, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                                    uint32_t* d_hit_bitmap);
int gft_finder_process_device_end(gft_finder* f);
/* test hooks mirroring what finder_test.go does by poking struct fields (finder/finder_test.go:205-217) */
int gft_finder_debug_add_literal(gft_finder* f, int which, const uint8_t* lit, uint32_t len);
int gft_finder_debug_set_updated(gft_finder* f, int updated_sub, int updated_rgx);
int gft_finder_debug_get_updated(const gft_finder* f, int* updated_sub, int* updated_rgx);

/* ---- Group finder: group/finder/finder.go + group/dsl on top of a finder --------------------
 * The tag-rule DSL and the object walk run on the host; every string leaf of every JSON document of a call becomes one
 * document of ONE batch through the finder (the reference runs one ProcessText per leaf, group/finder/internal.go:28-31).
 * Results are JSON documents written like the gft_dsl_* calls below (out/cap/needed). */
typedef struct gft_group gft_group;
int gft_group_create(gft_group** out, gft_finder* finder);      /* NewFinder(findthem) (finder.go:28-35); finder is borrowed */
void gft_group_destroy(gft_group* g);
const char* gft_group_last_error(const gft_group* g);
/* AddRule(ruleName, []string{expr}) (finder.go:45-66): GFT_E_PARSE + the reference's error text on malformed input */
int gft_group_add_rule(gft_group* g, const uint8_t* name, uint64_t name_len, const uint8_t* expr, uint64_t expr_len);
/* {"rules":{name:[{"ExpressionString":..,"Expression":{tree}}]},"fields":[..],"tags":[..]} (the struct finder_test.go compares) */
int gft_group_state(const gft_group* g, char* out, uint64_t cap, uint64_t* needed);
/* ProcessJson (what = 0, finder.go:160-172) / TagJson (what = 1, finder.go:80-92) over a batch of raw JSON documents
 * (blob + offsets).  include/exclude: JSON arrays of path prefixes, or NULL.  Output: one element per document,
 * {"rules":{rule:[expressions]}} / {"tags":{tag:{field:[expressions]}}} or {"error":"..."} */
int gft_group_process_jsons(gft_group* g, const uint8_t* json_blob, const uint64_t* doc_off, uint64_t n_docs,
                            const uint8_t* include_json, uint64_t include_len, const uint8_t* exclude_json,
                            uint64_t exclude_len, int what, char* out, uint64_t cap, uint64_t* needed);
/* the document of the last gft_group_process_jsons call again (it is kept, so a too-small buffer costs no second run) */
int gft_group_last_result(const gft_group* g, char* out, uint64_t cap, uint64_t* needed);
/* EvaluateRules (finder.go:118-137) on a caller-supplied {tag:{field:[expressions]}} map -> {rule:[expressions]} */
int gft_group_evaluate(gft_group* g, const uint8_t* tagmap_json, uint64_t len, char* out, uint64_t cap, uint64_t* needed);
/* string leaves and text bytes the last gft_group_process_jsons call sent through the finder (measurement) */
int gft_group_last_batch(const gft_group* g, uint64_t* leaves, uint64_t* bytes);
/* group DSL alone (host only): {"tree":..,"tags":[..],"fields":[..]} or {"error":..}; token list as gft_dsl_tokens */
int gft_group_dsl_parse(const uint8_t* expr, uint64_t len, char* out, uint64_t cap, uint64_t* needed);
int gft_group_dsl_tokens(const uint8_t* expr, uint64_t len, char* out, uint64_t cap, uint64_t* needed);

/* ---- DSL front-end alone (host only, no device needed) ------------------------------------------------------
 * Each writes a NUL-terminated JSON document into out (cap bytes) and the size it needs into *needed; returns
 * GFT_OK, or GFT_E_INVALID when cap is too small (call again with *needed bytes).
 *   gft_dsl_parse : {"tree":{...},"keywords":[..],"regexes":[..],"program":[..]} or {"error":"<reference text>"};
 *                   "program" uses slots = index into keywords ++ regexes (first-seen order).
 *   gft_dsl_tokens: [{"Tok":"AND","Lit":"and","Err":null}, ...] up to and including EOF or the first error
 *                   (dsl/scanner.go:79-106).
 *   gft_to_lower  : strings.ToLower of the input (raw bytes out, not JSON). */
int gft_dsl_parse(const uint8_t* expr, uint64_t len, int case_sensitive, char* out, uint64_t cap, uint64_t* needed);
int gft_dsl_tokens(const uint8_t* expr, uint64_t len, char* out, uint64_t cap, uint64_t* needed);
/* the literal runs every match of an RE2-syntax pattern must contain, as a JSON array (empty: the pattern cannot be
 * prefiltered) -- what the finder's regex prefilter adds to the device dictionary */
int gft_regex_required_literals(const uint8_t* pattern, uint64_t len, char* out, uint64_t cap, uint64_t* needed);
int gft_to_lower(const uint8_t* in, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* needed);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------------- */
/* on = 1: every kernel launch is bracketed by HIP events on the engine's stream (categories "scan", "solve", "aux");
 * on = 2: the scan kernel's launches only -- an event record is a node of its own on the stream, a few microseconds
 * between two kernels: a run that is being timed as a whole brackets the one kernel it prices; 0: off. */
int gft_profile_enable(gft_engine* e, int on);
/* Sums since the last reset.  names: "scan", "solve", "aux".  Synchronises the stream. */
int gft_profile_read(gft_engine* e, const char* name, double* total_ms, uint64_t* launches);
int gft_profile_reset(gft_engine* e);

/* ---- several devices behind one handle ------------------------------------------------------ */
/* A handle over n_devices HIP devices (devices == NULL / n_devices == 0: every visible device).  It is used exactly like
 * a single-device handle -- gft_build, gft_set_programs, gft_scan, gft_process, gft_process_again and the gft_finder_*
 * functions on top of it --: tables and programs are replicated, a batch is cut into contiguous document ranges of
 * near-equal text bytes, every device has its own host thread and stream for the duration of a call, results land in the
 * caller's buffers in document order.  The *_device entry points of such a handle run on its first device; shards that
 * are already resident on their devices go through gft_process_device_multi.  A device may be named twice (tests). */
int gft_engine_create_multi(gft_engine** out, const int* devices, int n_devices);
int gft_n_devices(const gft_engine* e);
/* how gft_process_device_multi moves the shards' bitmaps to the first device: "rccl" (ncclSend / ncclRecv over xGMI),
 * "copy" (device-to-device copies: RCCL unavailable, or a device named twice) or "" (single-device handle).
 * GFT_RCCL_SELF=1 in the environment when the handle is created: a list that names ONE device several times gets one
 * communicator of one rank and the gather is that rank's grouped ncclSend / ncclRecv to itself ("rccl") -- how the RCCL
 * branch is exercised on a one-GPU box (tests/test_gpu_multi.py). */
const char* gft_gather_mode(const gft_engine* e);
gft_engine* gft_device_engine(gft_engine* e, int i);     /* the per-device engine (its stream, its profile counters) */
/* the document cuts gft_process would use: device i gets documents [cut[i], cut[i+1]); cut has n_devices + 1 entries */
int gft_split_docs(const gft_engine* e, const uint64_t* doc_off, uint64_t n_docs, uint64_t* cut);
/* Device-resident shards: d_text[i] / d_doc_off[i] / n_docs[i] live on device i (64 bytes of readable slack behind every
 * blob).  Every device scans and solves its shard; then the path's ONE exchange step gathers the bitmaps into
 * d_bitmap_root on the first device, shard after shard (sum(n_docs) x ceil(n_exprs / 32) words): ncclSend / ncclRecv in
 * one group over xGMI on communicators from ncclCommInitAll (device-to-device copies if RCCL is not available). */
int gft_process_device_multi(gft_engine* e, const uint8_t* const* d_text, const uint64_t* const* d_doc_off, const uint64_t* n_docs,
                             uint32_t flags, uint32_t* d_bitmap_root);

/* ---- test hook: the table compiler without a device ------------------------------------------------------- */
/* Compiles `terms` into the scan kernel's tables on the host and walks them over ONE document the way the kernel does
 * (host emulation of the per-probe logic, csrc/scan3_tables.cpp): the matches of the unit [lo, len) of the document, in no
 * particular order, as (term id in sorted-unique order, position) pairs.  No HIP device is needed; tests use it to check
 * the table compiler against the oracle.  *needed = number of matches (GFT_E_INVALID when cap is too small). */
int gft_debug_emulate_scan(const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, const uint8_t* text,
                           uint32_t len, uint32_t lo, uint32_t flags, uint32_t scan_flags, uint32_t* out_term,
                           uint32_t* out_pos, uint64_t cap, uint64_t* needed);

/* The two-positions-per-probe filter of gft_scan5.hip alone, on the host: compiles `terms` (suffix-window tables, then the
 * 3-gram filter over `groups` merged byte classes; 0 = as many as the dictionary has) and walks ONE document the way the
 * kernel's lanes do -- a probe at every even offset from `lane_start` answers that position from the low word and the next
 * one from the high word.  out_exact[i] / out_dual[i] = 1 when the one-probe-per-byte filter of gft_scan2.hip / this filter
 * flags a window ending at byte i.  The second must flag whatever the first flags (and is equal to it when no classes are
 * merged); *groups_used = the number of groups.  The first is computed from the bucket table's keys and the short terms
 * themselves, so any alphabet is served.  GFT_E_UNSUPPORTED when the dictionary has no suffix-window tables at all.
 * No HIP device is needed. */
int gft_debug_scan5_filter(const uint8_t* terms_blob, const uint64_t* term_off, uint32_t n_terms, const uint8_t* text, uint32_t len,
                           uint32_t lane_start, uint32_t scan_flags, uint32_t groups, uint8_t* out_exact, uint8_t* out_dual,
                           uint32_t* groups_used);

/* The solver's program compiler alone, on the host: every program goes through the same steps as in gft_set_programs
 * (check, fusion with Sethi-Ullman operand order, control-bit device words) and its device words are then interpreted for
 * ONE document whose presence set is `present` (one byte per slot, non-zero = the slot's term occurs).  out_hit[i] = the
 * expression's truth value, out_depth[i] (nullable) = the accumulator-stack depth its fused form needs.  Programs with an
 * INORD group of more than one leaf need positions: GFT_E_UNSUPPORTED.  No HIP device is needed; tests use it to check the
 * compiler against the oracle's tree evaluation. */
int gft_debug_eval_programs(const uint32_t* prog_words, const uint64_t* prog_off, uint32_t n_exprs, uint32_t n_slots,
                            const uint8_t* present, uint8_t* out_hit, uint32_t* out_depth);

/* The host solver alone (csrc/host_solve.cpp: dsl/expression.go:66-142 restated over the postfix words, lists materialised):
 * Solve of ONE program over ONE document's map, given as n_lists keys -- slots[k] with the positions
 * positions[list_off[k] .. list_off[k+1]) in the order addMatchesToSolverMap appended them (a key may have no position at
 * all: it is present all the same, dsl/expression_test.go:29-33).  *out = 1 / 0.  This is what gft_process* runs for the
 * (expression, document) pairs the device does not answer itself: expressions beyond the device solver's limits, and INORD
 * expressions over a slot whose list is not ascending (a keyword and a regex with the same literal).  No device needed. */
int gft_debug_host_solve(const uint32_t* words, uint64_t len, const uint32_t* slots, const uint64_t* list_off,
                         const int64_t* positions, uint32_t n_lists, int* out);

#ifdef __cplusplus
}
#endif
#endif /* GFT_H */
