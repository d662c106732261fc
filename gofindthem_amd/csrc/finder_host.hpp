// C++ mirror of the reference's finder package API (finder/finder.go, finder/substringEngine.go,
// finder/regexEngine.go) on top of the GPU engine.  Same names, argument meaning and error behaviour; the Go
// `error` value is a std::string here ("" == nil).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/gft.h"
#include "dsl_compile.hpp"

namespace gft {

using Error = std::string;

// finder.Match (finder/finder.go:11-14)
struct Match {
    int64_t Position;
    std::string Term;
};

// finder.ExpressionResult (finder/finder.go:25-29; field names sic)
struct ExpressionResult {
    int ExpresionIndex;
    std::string ExpresionStr;
    std::string Tag;
};

// finder.SubstringEngine (finder/substringEngine.go:11-18)
class SubstringEngine {
public:
    virtual ~SubstringEngine() {}
    virtual Error BuildEngine(const std::vector<std::string>& keywords, bool caseSensitive) = 0;
    virtual Error FindSubstrings(const std::string& text, std::vector<Match>& matches) = 0;
};

// finder.RegexEngine (finder/regexEngine.go:8-15)
class RegexEngine {
public:
    virtual ~RegexEngine() {}
    virtual Error BuildEngine(const std::vector<std::string>& regexes, bool caseSensitive) = 0;
    virtual Error FindRegexes(const std::string& text, std::vector<Match>& matches) = 0;
};

// finder.EmptyEngine / finder.EmptyRgxEngine (substringEngine.go:121-133, regexEngine.go:49-60)
class EmptyEngine : public SubstringEngine {
public:
    Error BuildEngine(const std::vector<std::string>&, bool) override { return ""; }
    Error FindSubstrings(const std::string&, std::vector<Match>&) override { return ""; }
};
class EmptyRgxEngine : public RegexEngine {
public:
    Error BuildEngine(const std::vector<std::string>&, bool) override { return ""; }
    Error FindRegexes(const std::string&, std::vector<Match>&) override { return ""; }
};

// The drop-in for finder.CloudflareForkEngine: same two methods, backed by libgft's HIP kernels.
class GpuEngine : public SubstringEngine {
public:
    explicit GpuEngine(int device = -1);
    // one engine over several devices (gft_engine_create_multi): batches are sharded across them by the library
    GpuEngine(const int* devices, int n_devices);
    ~GpuEngine() override;
    Error BuildEngine(const std::vector<std::string>& keywords, bool caseSensitive) override;
    Error FindSubstrings(const std::string& text, std::vector<Match>& matches) override;
    gft_engine* handle() const { return h_; }
    const Error& create_error() const { return create_err_; }
    uint64_t builds() const { return builds_; }
private:
    gft_engine* h_ = nullptr;
    Error create_err_;
    uint64_t builds_ = 0;
};

// engines supplied through the C ABI as callbacks (foreign implementations, test mocks)
class CallbackSubEngine : public SubstringEngine {
public:
    CallbackSubEngine(gft_engine_build_fn b, gft_engine_find_fn f, void* u) : b_(b), f_(f), u_(u) {}
    Error BuildEngine(const std::vector<std::string>& keywords, bool caseSensitive) override;
    Error FindSubstrings(const std::string& text, std::vector<Match>& matches) override;
private:
    gft_engine_build_fn b_; gft_engine_find_fn f_; void* u_;
};
class CallbackRgxEngine : public RegexEngine {
public:
    CallbackRgxEngine(gft_engine_build_fn b, gft_engine_find_fn f, void* u) : b_(b), f_(f), u_(u) {}
    Error BuildEngine(const std::vector<std::string>& regexes, bool caseSensitive) override;
    Error FindRegexes(const std::string& text, std::vector<Match>& matches) override;
private:
    gft_engine_build_fn b_; gft_engine_find_fn f_; void* u_;
};

// finder.Finder (finder/finder.go:32-240)
class Finder {
public:
    // gpu: the engine whose kernels scan (when subEng is that same GpuEngine) and solve.
    Finder(SubstringEngine* subEng, RegexEngine* rgxEng, bool caseSensitive, GpuEngine* gpu);

    Error AddExpression(const std::string& expression) { return AddExpressionWithTag(expression, ""); }
    Error AddExpressions(const std::vector<std::string>& expressions);
    Error AddExpressionsWithTag(const std::vector<std::string>& expressions, const std::string& tag);
    Error AddExpressionWithTag(const std::string& expression, const std::string& tag);
    // returns the expressions that evaluated true, in registration order (never "nil": empty vector)
    Error ProcessText(const std::string& text, std::vector<ExpressionResult>& expRes);
    // batch extension: bitmap[d * words + (i >> 5)] bit (i & 31); words = ceil(n_expressions / 32)
    Error ProcessTexts(const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t* bitmap);
    Error ProcessDevice(const uint8_t* d_blob, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_bitmap);
    // pipelined: Begin enqueues (gft_process_device_begin), End completes the oldest batch begun, host ToLower repeat included
    Error ProcessDeviceBegin(const uint8_t* d_blob, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_bitmap);
    Error ProcessDeviceEnd();
    Error ForceBuild();
    const std::vector<std::string>& GetKeywords() const { return keywords_; }
    const std::vector<std::string>& GetRegexes() const { return regexes_; }

    struct ExprWrapper {
        std::string exprString;
        std::unique_ptr<dsl::Expression> expression;
        std::string tag;
    };
    const std::vector<ExprWrapper>& expressions() const { return expressions_; }
    int last_code() const { return last_code_; }
    uint64_t last_regex_docs = 0;        // documents the host regex engine saw in the last prefiltered ProcessTexts
    bool force_host_lower_ = false;      // ProcessTexts is repeating a batch whose text the device cannot fold (finder_host.cpp)

    // test hooks (finder_test.go pokes the struct fields directly)
    void debug_add_literal(int which, const std::string& lit);
    bool updatedSubMachine = false, updatedRgxMachine = false;

private:
    struct Record { uint32_t slot, pos; };
    Error sync_device();
    Error fail_gft(int rc);
    void add_matches(const std::vector<Match>& ms, std::vector<Record>& out);
    Error collect(const std::string& lowered, bool run_sub, std::vector<Record>& out, bool run_rgx = true);
    Error process_texts_prefiltered(const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, uint32_t* bitmap,
                                    uint32_t flags, bool need_host_text);

    // ---- regex prefilter.  With the GPU substring engine every regex that has required literals
    // (dsl::RegexRequiredLiterals) gets a hidden AND-of-literals program behind the user's expressions; the literals
    // join the device dictionary.  A batch is then solved once without regex hits, the host regex engine runs only on
    // the documents whose hidden programs fired, and the solver runs again (scan reused) if any of them matched.
    bool prefilter_active() const;
    const std::vector<std::string>& device_dictionary() const;   // keywords (+ hidden literals when the prefilter is active)
    mutable std::vector<std::string> dev_dict_;
    mutable bool dev_dict_ok_ = false, dev_dict_active_ = false;
    mutable size_t dev_dict_kw_ = 0, dev_dict_rx_ = 0;
    size_t total_programs() const { return expressions_.size() + (prefilter_active() ? regexes_.size() : 0); }
    std::vector<std::vector<std::string>> rgx_required_;    // per regex: literal runs every match contains

    std::vector<ExprWrapper> expressions_;
    std::vector<std::string> keywords_, regexes_;
    SubstringEngine* subEng_;
    RegexEngine* rgxEng_;
    bool caseSensitive_;
    GpuEngine* gpu_;
    bool gpu_sub_;                       // subEng_ is the GPU engine itself: scanning is fused into gft_process
    bool programs_dirty_ = true;
    bool empty_ready_ = false;           // we uploaded an empty dictionary ourselves (foreign engine / no keywords)
    uint64_t seen_builds_ = ~0ull;
    std::unordered_set<std::string> kw_set_, rgx_set_;
    std::unordered_map<std::string, uint32_t> slot_of_;
    Error solve_error_;                  // what Expression.Solve would return for every document, if anything
    int last_code_ = 0;
    struct Begun { const uint8_t* d_blob; const uint64_t* d_doc_off; uint64_t n_docs; uint32_t* d_bitmap; };
    Begun begun_[2] = {};                // batches of ProcessDeviceBegin that ProcessDeviceEnd has not completed yet
    unsigned first_begun_ = 0, n_begun_ = 0;
    Error repeat_if_not_ascii(const uint8_t* d_blob, const uint64_t* d_doc_off, uint64_t n_docs, uint32_t* d_bitmap);
};

}  // namespace gft
