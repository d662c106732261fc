#include "finder_host.hpp"

#include <hip/hip_runtime.h>

#include <mutex>
#include <algorithm>
#include <cstring>
#include <thread>

#include "gft_guard.hpp"

namespace gft {

namespace {
void pack(const std::vector<std::string>& v, std::vector<uint8_t>& blob, std::vector<uint64_t>& off) {
    blob.clear(); off.assign(1, 0);
    for (const auto& s : v) {
        blob.insert(blob.end(), s.begin(), s.end());
        off.push_back(blob.size());
    }
    if (blob.empty()) blob.push_back(0);
}
struct Sink { std::vector<Match>* out; };
void emit_cb(void* sink, const uint8_t* term, uint32_t len, int64_t pos) {
    static_cast<Sink*>(sink)->out->push_back(Match{pos, std::string((const char*)term, len)});
}
}  // namespace

// ---- GpuEngine -------------------------------------------------------------------------------------------
GpuEngine::GpuEngine(int device) {
    int rc = gft_engine_create(&h_, device);
    if (rc != GFT_OK) {
        create_err_ = h_ ? gft_last_error(h_) : "gft_engine_create failed";
        if (h_) { gft_engine_destroy(h_); h_ = nullptr; }
    }
}
GpuEngine::GpuEngine(const int* devices, int n_devices) {
    int rc = gft_engine_create_multi(&h_, devices, n_devices);
    // (GFT_W_NO_RCCL: a complete handle, its gathers are device-to-device copies -- gft_gather_mode tells the caller)
    if (rc != GFT_OK && rc != GFT_W_NO_RCCL) {
        create_err_ = h_ ? gft_last_error(h_) : "gft_engine_create_multi failed";
        if (h_) { gft_engine_destroy(h_); h_ = nullptr; }
    }
}
GpuEngine::~GpuEngine() { if (h_) gft_engine_destroy(h_); }

Error GpuEngine::BuildEngine(const std::vector<std::string>& keywords, bool) {
    if (!h_) return create_err_;
    std::vector<uint8_t> blob; std::vector<uint64_t> off;
    pack(keywords, blob, off);
    if (gft_build(h_, blob.data(), off.data(), (uint32_t)keywords.size(), GFT_POS_START) != GFT_OK) return gft_last_error(h_);
    builds_++;
    return "";
}

Error GpuEngine::FindSubstrings(const std::string& text, std::vector<Match>& matches) {
    if (!h_) return create_err_;
    const uint64_t off[2] = {0, text.size()};
    gft_matches m;
    if (gft_scan(h_, (const uint8_t*)text.data(), off, 1, 0, &m) != GFT_OK) return gft_last_error(h_);
    for (uint64_t i = 0; i < m.n_matches; i++) {
        const uint8_t* p; uint32_t n;
        gft_term(h_, m.term_id[i], &p, &n);
        matches.push_back(Match{(int64_t)m.pos[i], std::string((const char*)p, n)});
    }
    return "";
}

// ---- callback engines ------------------------------------------------------------------------------------
static Error cb_build(gft_engine_build_fn b, void* u, const std::vector<std::string>& v, bool cs) {
    if (!b) return "";
    std::vector<uint8_t> blob; std::vector<uint64_t> off;
    pack(v, blob, off);
    char err[512]; err[0] = 0;
    if (b(u, blob.data(), off.data(), (uint32_t)v.size(), cs ? 1 : 0, err, sizeof err) != 0) {
        err[sizeof err - 1] = 0;
        return err[0] ? Error(err) : Error("engine build failed");
    }
    return "";
}
static Error cb_find(gft_engine_find_fn f, void* u, const std::string& text, std::vector<Match>& out) {
    if (!f) return "";
    Sink s{&out};
    char err[512]; err[0] = 0;
    if (f(u, (const uint8_t*)text.data(), text.size(), emit_cb, &s, err, sizeof err) != 0) {
        err[sizeof err - 1] = 0;
        return err[0] ? Error(err) : Error("engine find failed");
    }
    return "";
}
Error CallbackSubEngine::BuildEngine(const std::vector<std::string>& k, bool cs) { return cb_build(b_, u_, k, cs); }
Error CallbackSubEngine::FindSubstrings(const std::string& t, std::vector<Match>& m) { return cb_find(f_, u_, t, m); }
Error CallbackRgxEngine::BuildEngine(const std::vector<std::string>& k, bool cs) { return cb_build(b_, u_, k, cs); }
Error CallbackRgxEngine::FindRegexes(const std::string& t, std::vector<Match>& m) { return cb_find(f_, u_, t, m); }

// ---- Finder ----------------------------------------------------------------------------------------------
Finder::Finder(SubstringEngine* subEng, RegexEngine* rgxEng, bool caseSensitive, GpuEngine* gpu)
    : subEng_(subEng), rgxEng_(rgxEng), caseSensitive_(caseSensitive), gpu_(gpu),
      gpu_sub_(static_cast<SubstringEngine*>(gpu) == subEng) {}

void Finder::debug_add_literal(int which, const std::string& lit) {
    if ((which ? rgx_set_ : kw_set_).insert(lit).second) (which ? regexes_ : keywords_).push_back(lit);
    programs_dirty_ = true;
}

Error Finder::AddExpressions(const std::vector<std::string>& expressions) { return AddExpressionsWithTag(expressions, ""); }

Error Finder::AddExpressionsWithTag(const std::vector<std::string>& expressions, const std::string& tag) {
    for (const auto& e : expressions) {
        Error err = AddExpressionWithTag(e, tag);
        if (!err.empty()) return err;
    }
    return "";
}

// finder/finder.go:115-134
Error Finder::AddExpressionWithTag(const std::string& expression, const std::string& tag) {
    dsl::ParseResult r = dsl::Parse(expression, caseSensitive_);
    if (!r.err.empty()) { last_code_ = GFT_E_PARSE; return r.err; }
    if (solve_error_.empty()) solve_error_ = dsl::SolveError(*r.expr);   // first failing expression wins (finder.go:201-205)
    expressions_.push_back(ExprWrapper{expression, std::move(r.expr), tag});
    // the reference clears the dirty flag for every key of the parser's set, new or not (finder.go:123-131)
    for (const auto& k : r.keywords) {
        if (kw_set_.insert(k).second) keywords_.push_back(k);
        updatedSubMachine = false;
    }
    for (const auto& g : r.regexes) {
        if (rgx_set_.insert(g).second) {
            regexes_.push_back(g);
            rgx_required_.push_back(dsl::RegexRequiredLiterals(g));
            if (gpu_sub_) updatedSubMachine = false;        // the prefilter's literals join the device dictionary
        }
        updatedRgxMachine = false;
    }
    programs_dirty_ = true;
    return "";
}

bool Finder::prefilter_active() const {
    if (!gpu_sub_ || regexes_.empty()) return false;
    if (const char* e = getenv("GFT_REGEX_PREFILTER")) if (e[0] == '0') return false;
    for (const auto& r : rgx_required_) if (r.empty()) return false;   // one unfilterable regex makes every document a candidate
    return rgx_required_.size() == regexes_.size();
}

const std::vector<std::string>& Finder::device_dictionary() const {
    // rebuilt only when a literal was added or the prefilter switched (both lists only ever grow): ProcessText asks
    // for it on every call
    const bool active = prefilter_active();
    if (dev_dict_ok_ && dev_dict_kw_ == keywords_.size() && dev_dict_rx_ == regexes_.size() && dev_dict_active_ == active) return dev_dict_;
    dev_dict_ = keywords_;
    if (active)
        for (const auto& r : rgx_required_)
            for (const auto& lit : r)
                if (!kw_set_.count(lit) && std::find(dev_dict_.begin() + (long)keywords_.size(), dev_dict_.end(), lit) == dev_dict_.end())
                    dev_dict_.push_back(lit);
    dev_dict_ok_ = true; dev_dict_kw_ = keywords_.size(); dev_dict_rx_ = regexes_.size(); dev_dict_active_ = active;
    return dev_dict_;
}

Error Finder::fail_gft(int rc) {
    last_code_ = rc;
    return gft_last_error(gpu_->handle());
}

// (re)build what the solver kernel needs: a dictionary on the device and the postfix programs over its slots
Error Finder::sync_device() {
    if (!gpu_ || !gpu_->handle()) { last_code_ = GFT_E_HIP; return gpu_ ? gpu_->create_error() : "no GPU engine"; }
    gft_engine* h = gpu_->handle();
    // With a foreign substring engine (or no keywords at all) every literal is a caller-supplied slot and the
    // device dictionary is empty; otherwise the dictionary was built by GpuEngine::BuildEngine in collect().
    const bool want_empty = !gpu_sub_ || device_dictionary().empty();
    if (want_empty && (!empty_ready_ || gpu_->builds() != seen_builds_)) {
        const uint64_t off0[1] = {0};
        const uint8_t z = 0;
        int rc = gft_build(h, &z, off0, 0, GFT_POS_START);
        if (rc) return fail_gft(rc);
        empty_ready_ = true;
        programs_dirty_ = true;
    }
    if (!want_empty) empty_ready_ = false;
    if (gpu_->builds() != seen_builds_) programs_dirty_ = true;
    if (!programs_dirty_) return "";
    const uint32_t n_terms = gft_n_terms(h);
    std::unordered_map<std::string, uint32_t> slots;
    uint32_t n_extra = 0;
    auto slot_of = [&](const std::string& lit) -> uint32_t {
        auto it = slots.find(lit);
        if (it != slots.end()) return it->second;
        int64_t t = n_terms ? gft_term_id(h, (const uint8_t*)lit.data(), (uint32_t)lit.size()) : -1;
        uint32_t s = t >= 0 ? (uint32_t)t : n_terms + n_extra++;
        slots.emplace(lit, s);
        return s;
    };
    for (const auto& k : keywords_) slot_of(k);
    for (const auto& g : regexes_) slot_of(g);
    std::vector<uint32_t> words;
    std::vector<uint64_t> poff(1, 0);
    for (const auto& w : expressions_) {
        dsl::CompileProgram(*w.expression, slot_of, words);
        poff.push_back(words.size());
    }
    if (prefilter_active())
        for (const auto& req : rgx_required_) {               // hidden program j: every required literal is present
            for (size_t i = 0; i < req.size(); i++) {
                words.push_back(GFT_OP_UNIT << 28 | slot_of(req[i]));
                if (i) words.push_back(GFT_OP_AND << 28);
            }
            poff.push_back(words.size());
        }
    if (words.empty()) words.push_back(0);
    int rc = gft_set_programs(h, words.data(), poff.data(), (uint32_t)(poff.size() - 1), n_extra);
    if (rc) return fail_gft(rc);
    slot_of_ = std::move(slots);   // literal -> slot, for addMatchesToSolverMap
    seen_builds_ = gpu_->builds();
    programs_dirty_ = false;
    return "";
}

// finder/finder.go:181-196: group by (lower-cased) term; terms that no expression mentions cannot influence
// any Solve and are dropped here
void Finder::add_matches(const std::vector<Match>& ms, std::vector<Record>& out) {
    for (const auto& m : ms) {
        auto it = slot_of_.find(caseSensitive_ ? m.Term : dsl::ToLower(m.Term));
        if (it == slot_of_.end()) continue;
        out.push_back(Record{it->second, (uint32_t)m.Position});
    }
}

// engine calls of ProcessText for one (already lower-cased) document, in the reference's order
// (finder/finder.go:146-176).  Matches are appended to `pending` and mapped to slots after sync_device().
Error Finder::collect(const std::string& text, bool run_sub, std::vector<Record>& out, bool run_rgx) {
    std::vector<Match> all;
    const std::vector<std::string>& dict = gpu_sub_ ? device_dictionary() : keywords_;
    if (!dict.empty()) {
        if (!updatedSubMachine) {
            Error err = subEng_->BuildEngine(dict, caseSensitive_);
            if (!err.empty()) { last_code_ = GFT_E_ENGINE; return err; }
            updatedSubMachine = true;
        }
        if (run_sub && !keywords_.empty()) {
            Error err = subEng_->FindSubstrings(text, all);
            if (!err.empty()) { last_code_ = GFT_E_ENGINE; return err; }
        }
    }
    if (!regexes_.empty()) {
        if (!updatedRgxMachine) {
            Error err = rgxEng_->BuildEngine(regexes_, caseSensitive_);
            if (!err.empty()) { last_code_ = GFT_E_ENGINE; return err; }
            updatedRgxMachine = true;
        }
        if (run_rgx) {
            Error err = rgxEng_->FindRegexes(text, all);
            if (!err.empty()) { last_code_ = GFT_E_ENGINE; return err; }
        }
    }
    // solveExpressions (finder.go:199-215) aborts on the first Solve error; such errors are properties of the
    // expression tree alone (see dsl::SolveError), so they are raised here, after the engine calls, like the
    // reference does
    if (!solve_error_.empty()) { last_code_ = GFT_E_INVALID; return solve_error_; }
    Error err = sync_device();
    if (!err.empty()) return err;
    add_matches(all, out);
    return "";
}

// no byte of [p, p + n) has its top bit set; eight bytes per step, a few threads for large batches
static bool all_ascii(const uint8_t* p, uint64_t n) {
    auto range = [](const uint8_t* q, uint64_t m) {
        uint64_t acc = 0, i = 0;
        for (; i + 8 <= m; i += 8) { uint64_t w; memcpy(&w, q + i, 8); acc |= w; }
        for (; i < m; i++) acc |= q[i];
        return (acc & 0x8080808080808080ull) == 0;
    };
    if (n < (16u << 20)) return range(p, n);
    constexpr unsigned kT = 4;
    bool ok[kT];
    const uint64_t part = (n + kT - 1) / kT;
    {
        std::vector<std::thread> th;
        th.reserve(kT);
        gft::JoinAll joined(th);             // (a thread that could not be started: the others are joined, the error travels up)
        for (unsigned t = 0; t < kT; t++) {
            const uint64_t a = std::min(n, t * part), b = std::min(n, (t + 1) * part);
            th.emplace_back([&ok, t, p, a, b, range]() noexcept { ok[t] = range(p + a, b - a); });
        }
    }
    bool all = true;
    for (unsigned t = 0; t < kT; t++) all = all && ok[t];
    return all;
}

// finder/finder.go:139-179
Error Finder::ProcessText(const std::string& text_in, std::vector<ExpressionResult>& expRes) {
    expRes.clear();
    last_code_ = 0;
    // strings.ToLower (finder.go:140-142).  ASCII text that only the device reads (GPU substring engine, no regex terms)
    // is folded by the scan kernel while it is read; anything else is lower-cased here first (it may change byte lengths)
    const bool host_engines = !gpu_sub_ || !regexes_.empty();
    const bool fold_on_device = !caseSensitive_ && !host_engines && all_ascii((const uint8_t*)text_in.data(), text_in.size());
    std::string lowered;
    if (!caseSensitive_ && !fold_on_device) lowered = dsl::ToLower(text_in);
    const std::string& text = (caseSensitive_ || fold_on_device) ? text_in : lowered;
    std::vector<Record> recs;
    Error err = collect(text, !gpu_sub_, recs);
    if (!err.empty()) return err;
    const uint64_t doff[2] = {0, text.size()};
    const uint64_t xoff[2] = {0, recs.size()};
    std::vector<uint32_t> xs(recs.size() + 1), xp(recs.size() + 1);
    for (size_t i = 0; i < recs.size(); i++) { xs[i] = recs[i].slot; xp[i] = recs[i].pos; }
    gft_extra_matches x{xoff, xs.data(), xp.data()};
    const size_t words = (total_programs() + 31) / 32;        // hidden prefilter programs sit behind the user's
    std::vector<uint32_t> bm(std::max<size_t>(words, 1), 0);
    int rc = gft_process(gpu_->handle(), (const uint8_t*)text.data(), doff, 1, fold_on_device ? GFT_FOLD_ASCII : 0, &x, bm.data());
    if (rc) return fail_gft(rc);
    for (size_t i = 0; i < expressions_.size(); i++)
        if (bm[i >> 5] >> (i & 31) & 1)
            expRes.push_back(ExpressionResult{(int)i, expressions_[i].exprString, expressions_[i].tag});
    return "";
}

Error Finder::ProcessTexts(const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t* bitmap) {
    last_code_ = 0;
    // case folding: ASCII-only batches fold on the device while the text is read; anything else goes through
    // strings.ToLower on the host first (it may change byte lengths, finder.go:140-142)
    uint32_t flags = 0;
    std::vector<uint8_t> lowered;
    std::vector<uint64_t> loff;
    const uint64_t total = n_docs ? doc_off[n_docs] : 0;
    bool ascii = true;
    // A large batch that only the device reads is not scanned for high bytes here first (a pass over the whole text on
    // host cores, a third of the call's time at PCIe rates): it goes up as it is, the scan kernels notice what ASCII
    // folding does not cover (gft_last_nonascii), and only such a batch comes back for the host's ToLower.
    const bool optimistic = !caseSensitive_ && !force_host_lower_ && gpu_sub_ && regexes_.empty() && !prefilter_active() &&
                            total - (n_docs ? doc_off[0] : 0) >= (16u << 20);
    if (!caseSensitive_) {
        if (force_host_lower_) ascii = false;
        else if (!optimistic) ascii = all_ascii(blob + (n_docs ? doc_off[0] : 0), total - (n_docs ? doc_off[0] : 0));
        if (ascii) flags = GFT_FOLD_ASCII;
    }
    const bool need_host_text = !caseSensitive_ && !ascii;
    if (prefilter_active() && n_docs) return process_texts_prefiltered(blob, doc_off, n_docs, bitmap, flags, need_host_text);
    const bool per_doc_engines = !gpu_sub_ || !regexes_.empty();
    std::vector<Record> recs;
    std::vector<uint64_t> xoff(1, 0);
    if (need_host_text || per_doc_engines) {
        if (need_host_text) loff.assign(1, 0);
        for (uint64_t d = 0; d < n_docs; d++) {
            std::string t((const char*)blob + doc_off[d], (size_t)(doc_off[d + 1] - doc_off[d]));
            if (!caseSensitive_) t = dsl::ToLower(t);
            if (need_host_text) { lowered.insert(lowered.end(), t.begin(), t.end()); loff.push_back(lowered.size()); }
            Error err = collect(t, !gpu_sub_, recs);
            if (!err.empty()) return err;
            xoff.push_back(recs.size());
        }
        if (need_host_text) { lowered.push_back(0); blob = lowered.data(); doc_off = loff.data(); flags = 0; }
    }
    if (n_docs == 0 || !(need_host_text || per_doc_engines)) {
        // the build the reference would do on its first ProcessText
        std::vector<Record> none;
        Error err = collect(std::string(), false, none);
        if (!err.empty()) return err;
    }
    std::vector<uint32_t> xs(recs.size() + 1), xp(recs.size() + 1);
    for (size_t i = 0; i < recs.size(); i++) { xs[i] = recs[i].slot; xp[i] = recs[i].pos; }
    gft_extra_matches x{xoff.data(), xs.data(), xp.data()};
    // (a disabled prefilter still leaves its programs out: total_programs() == expressions_.size() then)
    int rc = gft_process(gpu_->handle(), blob, doc_off, n_docs, flags, per_doc_engines && n_docs ? &x : nullptr, bitmap);
    if (rc) return fail_gft(rc);
    if (optimistic && gft_last_nonascii(gpu_->handle())) {
        // text that ASCII folding does not lower-case the way strings.ToLower does (finder.go:140-142): once more, through it
        struct Reset { bool& f; ~Reset() { f = false; } } reset{force_host_lower_};
        force_host_lower_ = true;
        return ProcessTexts(blob, doc_off, n_docs, bitmap);
    }
    return "";
}

// ProcessTexts with regex terms and the GPU substring engine: solve once without regex hits, run the host regex engine
// only where a regex's required literals are all present, solve again with those hits (the scan is reused)
Error Finder::process_texts_prefiltered(const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t* bitmap,
                                        uint32_t flags, bool need_host_text) {
    std::vector<uint8_t> lowered;
    std::vector<uint64_t> loff;
    if (need_host_text) {
        loff.assign(1, 0);
        for (uint64_t d = 0; d < n_docs; d++) {
            const std::string t = dsl::ToLower(std::string((const char*)blob + doc_off[d], (size_t)(doc_off[d + 1] - doc_off[d])));
            lowered.insert(lowered.end(), t.begin(), t.end());
            loff.push_back(lowered.size());
        }
        lowered.push_back(0);
        blob = lowered.data(); doc_off = loff.data(); flags = 0;
    }
    std::vector<Record> none;
    Error err = collect(std::string(), false, none, false);       // engine builds + programs, no Find* call
    if (!err.empty()) return err;
    const size_t n_user = expressions_.size(), n_all = total_programs();
    const size_t uw = (n_user + 31) / 32, aw = (n_all + 31) / 32;
    std::vector<uint32_t> full((size_t)n_docs * aw + 1, 0);
    int rc = gft_process(gpu_->handle(), blob, doc_off, n_docs, flags, nullptr, full.data());
    if (rc) return fail_gft(rc);
    // candidates: a hidden program fired
    std::vector<Record> recs;
    std::vector<uint64_t> xoff(1, 0);
    last_regex_docs = 0;
    for (uint64_t d = 0; d < n_docs; d++) {
        const uint32_t* row = full.data() + d * aw;
        bool cand = false;
        for (size_t e = n_user; e < n_all && !cand; e++) cand = row[e >> 5] >> (e & 31) & 1;
        if (cand) {
            last_regex_docs++;
            std::string t((const char*)blob + doc_off[d], (size_t)(doc_off[d + 1] - doc_off[d]));
            if (!caseSensitive_ && !need_host_text) t = dsl::ToLower(t);     // ASCII batch: fold here for the host engine
            std::vector<Match> ms;
            Error e2 = rgxEng_->FindRegexes(t, ms);
            if (!e2.empty()) { last_code_ = GFT_E_ENGINE; return e2; }
            add_matches(ms, recs);
        }
        xoff.push_back(recs.size());
    }
    if (!recs.empty()) {
        std::vector<uint32_t> xs(recs.size()), xp(recs.size());
        for (size_t i = 0; i < recs.size(); i++) { xs[i] = recs[i].slot; xp[i] = recs[i].pos; }
        gft_extra_matches x{xoff.data(), xs.data(), xp.data()};
        rc = gft_process_again(gpu_->handle(), n_docs, &x, full.data());
        if (rc) return fail_gft(rc);
    }
    // the user's expressions are the first n_user bits of every row
    const uint32_t tail = (n_user & 31) ? (1u << (n_user & 31)) - 1 : 0xFFFFFFFFu;
    for (uint64_t d = 0; d < n_docs; d++)
        for (size_t w = 0; w < uw; w++)
            bitmap[d * uw + w] = full[d * aw + w] & (w + 1 == uw ? tail : 0xFFFFFFFFu);
    return "";
}

Error Finder::ProcessDevice(const uint8_t* d_blob, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_bitmap) {
    last_code_ = 0;
    if (!gpu_sub_ || !regexes_.empty()) {
        last_code_ = GFT_E_UNSUPPORTED;
        return "device-resident processing needs the GPU substring engine and no regex terms";
    }
    std::vector<Record> none;
    Error err = collect(std::string(), false, none);
    if (!err.empty()) return err;
    int rc = gft_process_device(gpu_->handle(), d_blob, d_doc_off, n_docs, caseSensitive_ ? 0 : GFT_FOLD_ASCII, nullptr,
                                d_bitmap);
    if (rc) return fail_gft(rc);
    return repeat_if_not_ascii(d_blob, d_doc_off, n_docs, d_bitmap);
}

Error Finder::ProcessDeviceBegin(const uint8_t* d_blob, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_bitmap) {
    last_code_ = 0;
    if (!gpu_sub_ || !regexes_.empty()) {
        last_code_ = GFT_E_UNSUPPORTED;
        return "device-resident processing needs the GPU substring engine and no regex terms";
    }
    if (n_begun_ == 2) { last_code_ = GFT_E_INVALID; return "two batches are in flight already"; }
    std::vector<Record> none;
    Error err = collect(std::string(), false, none);
    if (!err.empty()) return err;
    int rc = gft_process_device_begin(gpu_->handle(), d_blob, d_doc_off, n_docs, caseSensitive_ ? 0 : GFT_FOLD_ASCII, nullptr, d_bitmap);
    if (rc) return fail_gft(rc);
    begun_[(first_begun_ + n_begun_) % 2] = Begun{d_blob, d_doc_off, n_docs, d_bitmap};
    n_begun_++;
    return "";
}

Error Finder::ProcessDeviceEnd() {
    last_code_ = 0;
    if (!n_begun_) { last_code_ = GFT_E_INVALID; return "no batch in flight"; }
    const Begun b = begun_[first_begun_];
    first_begun_ = (first_begun_ + 1) % 2;
    n_begun_--;
    int rc = gft_process_device_end(gpu_->handle());
    if (rc) return fail_gft(rc);
    if (n_begun_ && !caseSensitive_ && b.n_docs && gft_last_nonascii(gpu_->handle())) {
        // (the host repeat below uses the handle's synchronous entry points: the younger batch is completed first -- its
        // verdict is kept for its own End)
        last_code_ = GFT_E_UNSUPPORTED;
        return "a batch that leaves ASCII cannot be repeated on the host while another batch is in flight: end that one first (or use ProcessDevice)";
    }
    return repeat_if_not_ascii(b.d_blob, b.d_doc_off, b.n_docs, b.d_bitmap);
}

Error Finder::repeat_if_not_ascii(const uint8_t* d_blob, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_bitmap) {
    if (!caseSensitive_ && n_docs && gft_last_nonascii(gpu_->handle())) {
        // The kernels lower-case A-Z only; the reference runs strings.ToLower (finder.go:140-142), which also maps
        // non-ASCII upper-case letters, rewrites invalid UTF-8 and may change byte lengths.  A batch that holds bytes
        // >= 0x80 is therefore repeated through the host path: text back to the host, ToLower per document, bitmap up.
        std::vector<uint64_t> off(n_docs + 1);
        if (hipMemcpy(off.data(), d_doc_off, (n_docs + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) { last_code_ = GFT_E_HIP; return "device-to-host copy failed"; }
        std::vector<uint8_t> text((size_t)(off[n_docs] - off[0]) + 1);
        if (off[n_docs] > off[0] &&
            hipMemcpy(text.data(), d_blob + off[0], (size_t)(off[n_docs] - off[0]), hipMemcpyDeviceToHost) != hipSuccess) { last_code_ = GFT_E_HIP; return "device-to-host copy failed"; }
        const uint64_t base = off[0];
        for (auto& o : off) o -= base;
        const size_t words = (expressions_.size() + 31) / 32;
        std::vector<uint32_t> bm((size_t)n_docs * words + 1);
        Error err = ProcessTexts(text.data(), off.data(), n_docs, bm.data());
        if (!err.empty()) return err;
        if (words && hipMemcpy(d_bitmap, bm.data(), (size_t)n_docs * words * 4, hipMemcpyHostToDevice) != hipSuccess) { last_code_ = GFT_E_HIP; return "host-to-device copy failed"; }
    }
    return "";
}

// finder/finder.go:218-235, including its quirk: the second branch sets updatedSubMachine
Error Finder::ForceBuild() {
    last_code_ = 0;
    if (!updatedSubMachine) {
        Error err = subEng_->BuildEngine(keywords_, caseSensitive_);
        if (!err.empty()) { last_code_ = GFT_E_ENGINE; return err; }
        updatedSubMachine = true;
    }
    if (!updatedRgxMachine) {
        Error err = rgxEng_->BuildEngine(regexes_, caseSensitive_);
        if (!err.empty()) { last_code_ = GFT_E_ENGINE; return err; }
        updatedSubMachine = true;
    }
    return "";
}

}  // namespace gft

// ---- C ABI ---------------------------------------------------------------------------------------------------
using namespace gft;

struct gft_finder {
    std::unique_ptr<GpuEngine> gpu;
    std::unique_ptr<SubstringEngine> sub_cb;
    std::unique_ptr<RegexEngine> rgx;
    std::unique_ptr<Finder> finder;
    bool case_sensitive = true;
    std::string err, json;
    // one caller at a time per finder (ProcessText mutates the lazy-build flags, finder.go:152,168): Go callers share a
    // built Finder between goroutines, so every entry point takes this
    mutable std::recursive_mutex mu;
};
#define GFT_FLOCK(f) std::lock_guard<std::recursive_mutex> _gft_flock((f)->mu)

// for group_host.cpp: the C++ object behind the handle
gft::Finder* gft_finder_impl(gft_finder* f) { return f ? f->finder.get() : nullptr; }

extern "C" {

int gft_finder_create(gft_finder** out, int case_sensitive, int device) try {
    if (!out) return GFT_E_INVALID;
    gft_finder* f = new gft_finder();
    f->case_sensitive = case_sensitive != 0;
    f->gpu.reset(new GpuEngine(device));
    f->rgx.reset(new EmptyRgxEngine());
    f->finder.reset(new Finder(f->gpu.get(), f->rgx.get(), f->case_sensitive, f->gpu.get()));
    *out = f;
    if (!f->gpu->handle()) { f->err = f->gpu->create_error(); return GFT_E_HIP; }
    return GFT_OK;
} GFT_CATCH(nullptr)

int gft_finder_create_multi(gft_finder** out, int case_sensitive, const int* devices, int n_devices) try {
    if (!out || n_devices < 0 || (n_devices && !devices)) return GFT_E_INVALID;
    gft_finder* f = new gft_finder();
    f->case_sensitive = case_sensitive != 0;
    f->gpu.reset(new GpuEngine(devices, n_devices));
    f->rgx.reset(new EmptyRgxEngine());
    f->finder.reset(new Finder(f->gpu.get(), f->rgx.get(), f->case_sensitive, f->gpu.get()));
    *out = f;
    if (!f->gpu->handle()) { f->err = f->gpu->create_error(); return GFT_E_HIP; }
    return GFT_OK;
} GFT_CATCH(nullptr)

void gft_finder_destroy(gft_finder* f) { delete f; }
const char* gft_finder_last_error(const gft_finder* f) { return f ? f->err.c_str() : "null finder"; }
gft_engine* gft_finder_engine(gft_finder* f) { return f && f->gpu ? f->gpu->handle() : nullptr; }

static int finder_ret(gft_finder* f, const Error& e, int dflt) {
    if (e.empty()) return GFT_OK;
    f->err = e;
    int c = f->finder->last_code();
    return c ? c : dflt;
}

// engines can only be swapped before the first expression is added (like passing them to NewFinder)
int gft_finder_set_substring_engine(gft_finder* f, gft_engine_build_fn build, gft_engine_find_fn find, void* user) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    if (f->finder->expressions().size() || f->finder->GetKeywords().size()) { f->err = "engines must be set before expressions are added"; return GFT_E_INVALID; }
    f->sub_cb.reset(new CallbackSubEngine(build, find, user));
    f->finder.reset(new Finder(f->sub_cb.get(), f->rgx.get(), f->case_sensitive, f->gpu.get()));
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_set_regex_engine(gft_finder* f, gft_engine_build_fn build, gft_engine_find_fn find, void* user) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    if (f->finder->expressions().size() || f->finder->GetRegexes().size()) { f->err = "engines must be set before expressions are added"; return GFT_E_INVALID; }
    f->rgx.reset(new CallbackRgxEngine(build, find, user));
    SubstringEngine* sub = f->sub_cb ? f->sub_cb.get() : static_cast<SubstringEngine*>(f->gpu.get());
    f->finder.reset(new Finder(sub, f->rgx.get(), f->case_sensitive, f->gpu.get()));
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_add_expression(gft_finder* f, const uint8_t* expr, uint64_t expr_len, const uint8_t* tag,
                              uint64_t tag_len) try {
    if (!f || (!expr && expr_len)) return GFT_E_INVALID;
    GFT_FLOCK(f);
    Error e = f->finder->AddExpressionWithTag(std::string((const char*)expr, (size_t)expr_len),
                                              std::string(tag ? (const char*)tag : "", (size_t)(tag ? tag_len : 0)));
    return finder_ret(f, e, GFT_E_PARSE);
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

uint32_t gft_finder_n_expressions(const gft_finder* f) { return f ? (uint32_t)f->finder->expressions().size() : 0; }

uint32_t gft_finder_n_literals(const gft_finder* f, int which) try {
    if (!f) return 0;
    return (uint32_t)(which ? f->finder->GetRegexes() : f->finder->GetKeywords()).size();
} GFT_CATCH_VALUE(0)

int gft_finder_literal(const gft_finder* f, int which, uint32_t i, const uint8_t** ptr, uint32_t* len) try {
    if (!f || !ptr || !len) return GFT_E_INVALID;
    GFT_FLOCK(f);
    const auto& v = which ? f->finder->GetRegexes() : f->finder->GetKeywords();
    if (i >= v.size()) return GFT_E_INVALID;
    *ptr = (const uint8_t*)v[i].data(); *len = (uint32_t)v[i].size();
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_expression(const gft_finder* f, uint32_t i, const uint8_t** str, uint32_t* str_len,
                          const uint8_t** tag, uint32_t* tag_len, const uint8_t** tree_json, uint32_t* json_len) try {
    if (!f || i >= f->finder->expressions().size()) return GFT_E_INVALID;
    GFT_FLOCK(f);
    const auto& w = f->finder->expressions()[i];
    if (str) { *str = (const uint8_t*)w.exprString.data(); *str_len = (uint32_t)w.exprString.size(); }
    if (tag) { *tag = (const uint8_t*)w.tag.data(); *tag_len = (uint32_t)w.tag.size(); }
    if (tree_json) {
        const_cast<gft_finder*>(f)->json = dsl::ToJson(*w.expression);
        *tree_json = (const uint8_t*)f->json.data(); *json_len = (uint32_t)f->json.size();
    }
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

uint64_t gft_finder_last_regex_docs(const gft_finder* f) { return f && f->finder ? f->finder->last_regex_docs : 0; }

int gft_finder_force_build(gft_finder* f) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    return finder_ret(f, f->finder->ForceBuild(), GFT_E_ENGINE);
} GFT_CATCH((f ? &f->err : nullptr))

int gft_finder_process_text(gft_finder* f, const uint8_t* text, uint64_t text_len, uint32_t* out_idx, uint32_t cap,
                            uint32_t* n_true) try {
    if (!f || !n_true || (!text && text_len)) return GFT_E_INVALID;
    GFT_FLOCK(f);
    std::vector<ExpressionResult> res;
    Error e = f->finder->ProcessText(std::string((const char*)text, (size_t)text_len), res);
    if (!e.empty()) return finder_ret(f, e, GFT_E_ENGINE);
    *n_true = (uint32_t)res.size();
    for (uint32_t i = 0; i < res.size() && i < cap; i++) out_idx[i] = (uint32_t)res[i].ExpresionIndex;
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_process_texts(gft_finder* f, const uint8_t* text_blob, const uint64_t* doc_off, uint64_t n_docs,
                             uint32_t* hit_bitmap) try {
    if (!f || (n_docs && !doc_off)) return GFT_E_INVALID;
    GFT_FLOCK(f);
    return finder_ret(f, f->finder->ProcessTexts(text_blob, doc_off, n_docs, hit_bitmap), GFT_E_ENGINE);
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_process_device(gft_finder* f, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                              uint32_t* d_hit_bitmap) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    return finder_ret(f, f->finder->ProcessDevice(d_text_blob, d_doc_off, n_docs, d_hit_bitmap), GFT_E_ENGINE);
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_process_device_begin(gft_finder* f, const uint8_t* d_text_blob, const uint64_t* d_doc_off, uint64_t n_docs,
                                    uint32_t* d_hit_bitmap) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    return finder_ret(f, f->finder->ProcessDeviceBegin(d_text_blob, d_doc_off, n_docs, d_hit_bitmap), GFT_E_ENGINE);
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_process_device_end(gft_finder* f) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    return finder_ret(f, f->finder->ProcessDeviceEnd(), GFT_E_ENGINE);
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_debug_add_literal(gft_finder* f, int which, const uint8_t* lit, uint32_t len) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    f->finder->debug_add_literal(which, std::string((const char*)lit, len));
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_debug_set_updated(gft_finder* f, int updated_sub, int updated_rgx) try {
    if (!f) return GFT_E_INVALID;
    GFT_FLOCK(f);
    f->finder->updatedSubMachine = updated_sub != 0;
    f->finder->updatedRgxMachine = updated_rgx != 0;
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

int gft_finder_debug_get_updated(const gft_finder* f, int* updated_sub, int* updated_rgx) try {
    if (!f || !updated_sub || !updated_rgx) return GFT_E_INVALID;
    GFT_FLOCK(f);
    *updated_sub = f->finder->updatedSubMachine; *updated_rgx = f->finder->updatedRgxMachine;
    return GFT_OK;
} GFT_CATCH((f ? &const_cast<gft_finder*>(f)->err : nullptr))

}  // extern "C"
