// Host-side DSL front-end: expression string -> AST -> postfix program for the solver kernel.
// Mirrors the observable behaviour of dsl.NewParser(...).Parse() (dsl/parser.go:28-315, dsl/scanner.go:79-250):
// same tree shapes (precedence-free, left to right), same keyword / regex sets, same error strings.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace gft {
namespace dsl {

enum Token { ILLEGAL = 0, END_OF_INPUT, WS, KEYWORD, QUOTATION, OPPAR, CLPAR, AND, OR, NOT, INORD, REGEX };
const char* token_name(Token t);

enum ExprType { UNSET_EXPR = 0, AND_EXPR, OR_EXPR, NOT_EXPR, UNIT_EXPR, INORD_EXPR };
const char* expr_type_name(ExprType t);

// dsl.Expression (dsl/expression.go:42-48)
struct Expression {
    std::unique_ptr<Expression> LExpr, RExpr;
    ExprType Type = UNSET_EXPR;
    std::string Literal;
    bool Inord = false;
};

struct ScanResult { Token tok = ILLEGAL; std::string lit; std::string err; };

class Scanner {
public:
    explicit Scanner(const std::string& src) : s_(src) {}
    ScanResult Scan();
private:
    int32_t read();       // next rune, 0 at end of input (a NUL rune also reads as end of input, scanner.go:250)
    void unread();
    ScanResult scan_whitespace();
    ScanResult scan_operators();
    ScanResult scan_keyword(bool is_regex);
    const std::string& s_;
    size_t i_ = 0, last_ = 0;
};

struct ParseResult {
    std::unique_ptr<Expression> expr;      // null on error
    std::vector<std::string> keywords;     // first-seen order, unique
    std::vector<std::string> regexes;
    std::string err;                       // empty == nil error
};

ParseResult Parse(const std::string& src, bool case_sensitive);

// UTF-8 as Go reads it: an invalid byte decodes to U+FFFD (advance 1) and is re-encoded as such
int32_t DecodeRune(const std::string& s, size_t i, size_t* adv);
void EncodeRune(int32_t cp, std::string& out);

// Literal runs that every match of an RE2-syntax pattern must contain, in pattern order (the
// regex prefilter).  Conservative: anything not understood yields no literal from that part, or no literals at all
// (alternation at the top level, flag groups, \x / \p / \Q escapes, malformed repeats).  Runs shorter than min_len bytes
// are dropped.  An empty result means the pattern cannot be prefiltered.
std::vector<std::string> RegexRequiredLiterals(const std::string& pattern, size_t min_len = 2);

// strings.ToLower for the parser's literals and for document text (Unicode simple case mapping)
std::string ToLower(const std::string& s);
bool IsAscii(const std::string& s);

// tree -> postfix words of include/gft.h; slot_of maps a literal to its slot
void CompileProgram(const Expression& e, const std::function<uint32_t(const std::string&)>& slot_of,
                    std::vector<uint32_t>& out);

// "" if Expression.Solve can never fail on this tree, else the error text Solve returns for EVERY document
std::string SolveError(const Expression& e);

// debug/fixture form: {"Type":"AND","LExpr":{...},...}
std::string ToJson(const Expression& e);
void json_str(const std::string& s, std::string& out);   // append s as a JSON string literal

}  // namespace dsl
}  // namespace gft
