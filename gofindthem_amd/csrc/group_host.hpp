// C++ mirror of the reference's group package (group/dsl/*.go, group/finder/finder.go, group/finder/internal.go) on
// top of gft::Finder -- SURVEY.md 8(f) row 2.  Same names, argument meaning and error behaviour; the one structural
// change is the point of the exercise: every string leaf of an object (or of a whole batch of JSON documents) becomes
// one document of ONE Finder::ProcessTexts call, instead of one ProcessText per leaf (internal.go:28-31).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "finder_host.hpp"
#include "json_mini.hpp"

namespace gft {
namespace gdsl {

// group/dsl/scanner.go:12-35
enum Token { ILLEGAL = 0, END_OF_INPUT, WS, TAG, FIELD_PATH, QUOTATION, OPPAR, CLPAR, AND, OR, NOT };
const char* token_name(Token t);

// group/dsl/expression.go:11-17
enum ExprType { UNSET_EXPR = 0, AND_EXPR, OR_EXPR, NOT_EXPR, UNIT_EXPR };
const char* expr_type_name(ExprType t);

struct TagInfo { std::string Name, FieldPath; };            // expression.go:39-42

struct Expression {                                         // expression.go:46-51
    std::unique_ptr<Expression> LExpr, RExpr;
    ExprType Type = UNSET_EXPR;
    TagInfo Tag;
    mutable int32_t tag_id = -1;                            // batch evaluation: index of Tag.Name among the finder's tags
};

struct ScanResult { Token tok = ILLEGAL; std::string lit; std::string err; };

class Scanner {                                             // scanner.go:67-263
public:
    explicit Scanner(const std::string& src) : s_(src) {}
    ScanResult Scan();
private:
    int32_t read();
    void unread();
    ScanResult scan_whitespace();
    ScanResult scan_operators();
    ScanResult scan_tag();
    ScanResult scan_field_path();
    const std::string& s_;
    size_t i_ = 0, last_ = 0;
};

struct ParseResult {
    std::unique_ptr<Expression> expr;      // null on error
    std::vector<std::string> tags, fields; // unique, first-seen order (GetTags / GetFields, parser.go:281-297)
    std::string err;
};
ParseResult Parse(const std::string& src);                  // parser.go:35-175

// tag -> field path -> set of expression strings (the reference's map[string]map[string]map[string]struct{})
using TagMap = std::map<std::string, std::map<std::string, std::set<std::string>>>;

// Expression.Solve (expression.go:61-125); err = "" when fine
bool Solve(const Expression& e, const TagMap& m, std::string& err);

// the same recursion over a caller-supplied UNIT predicate (batch evaluation keeps tags as ids, not map keys)
template <class UnitPred>
bool SolveWith(const Expression& e, UnitPred&& unit, std::string& err) {
    switch (e.Type) {
    case UNIT_EXPR:
        return unit(e);
    case AND_EXPR:
    case OR_EXPR: {
        if (!e.LExpr || !e.RExpr) {
            err = std::string(e.Type == AND_EXPR ? "AND" : "OR") + " statement do not have right or left expression";
            return false;
        }
        const bool l = SolveWith(*e.LExpr, unit, err);
        if (!err.empty()) return false;
        const bool r = SolveWith(*e.RExpr, unit, err);
        if (!err.empty()) return false;
        return e.Type == AND_EXPR ? (l && r) : (l || r);
    }
    case NOT_EXPR: {
        if (!e.RExpr) { err = "NOT statement do not have expression"; return false; }
        const bool r = SolveWith(*e.RExpr, unit, err);
        if (!err.empty()) return false;
        return !r;
    }
    default:
        err = "unable to process expression type " + std::to_string((int)e.Type);
        return false;
    }
}

std::string ToJson(const Expression& e);                    // {"Type":"AND","LExpr":..,"RExpr":..} / {"Type":"UNIT","Tag":{..}}

}  // namespace gdsl

// group/finder/finder.go:12-17
class GroupFinder {
public:
    struct ExpressionWrapper { std::string ExpressionString; std::unique_ptr<gdsl::Expression> Expression; };
    using RuleResult = std::map<std::string, std::vector<std::string>>;   // expressionsByRule

    explicit GroupFinder(Finder* findthem) : findthem_(findthem) {}
    Error AddRule(const std::string& ruleName, const std::vector<std::string>& expressions);
    std::vector<std::string> GetFieldNames() const { return {fields_.begin(), fields_.end()}; }

    // One entry per input document: err (json.Unmarshal's, or the finder's), else the document's tag map
    // (want_tags) or its rule hits.
    struct DocResult { Error err; gdsl::TagMap tags; RuleResult rules; };
    // ProcessJson / TagJson over a batch (finder.go:80-103,160-172): all string leaves of all documents go through
    // ONE Finder::ProcessTexts; decoding, the walk, rule evaluation run on host threads, a document each
    Error ProcessJsons(const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, const std::vector<std::string>& includePaths,
                       const std::vector<std::string>& excludePaths, bool want_tags, std::vector<DocResult>& out);
    // EvaluateRules (finder.go:118-137)
    Error EvaluateRules(const gdsl::TagMap& m, RuleResult& out) const;

    const std::map<std::string, std::vector<ExpressionWrapper>>& rules() const { return rules_; }
    const std::set<std::string>& fields() const { return fields_; }
    const std::set<std::string>& tags() const { return tags_; }
    // leaves and bytes of the last TagJsons call (measurement)
    uint64_t last_leaves = 0, last_bytes = 0;

private:
    Finder* findthem_;
    std::map<std::string, std::vector<ExpressionWrapper>> rules_;
    std::set<std::string> fields_, tags_;
};

// isValidateFieldPath (internal.go:99-119)
bool IsValidFieldPath(const std::string& fieldPath, const std::vector<std::string>& includePaths,
                      const std::vector<std::string>& excludePaths);

}  // namespace gft
