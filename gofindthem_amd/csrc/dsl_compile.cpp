#include "dsl_compile.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "../../include/gft.h"

#include "gft_guard.hpp"

namespace gft {
namespace dsl {

const char* token_name(Token t) {
    static const char* n[] = {"ILLEGAL", "EOF", "WS", "KEYWORD", "QUOTATION", "OPPAR", "CLPAR",
                              "AND", "OR", "NOT", "INORD", "REGEX"};
    return (int)t >= 0 && (int)t < 12 ? n[t] : "UNEXPECTED";
}
const char* expr_type_name(ExprType t) {
    static const char* n[] = {"UNSET", "AND", "OR", "NOT", "UNIT", "INORD"};
    return (int)t >= 0 && (int)t < 6 ? n[t] : "UNEXPECTED";
}

// ---- UTF-8 helpers (Go reads runes; an invalid byte decodes to U+FFFD and is re-encoded as such) -------------
namespace {
constexpr int32_t kRuneError = 0xFFFD;

int32_t decode_rune(const std::string& s, size_t i, size_t* adv) {
    const unsigned char c0 = (unsigned char)s[i];
    *adv = 1;
    if (c0 < 0x80) return c0;
    int need; int32_t cp, minv;
    if (c0 >= 0xC2 && c0 <= 0xDF) { need = 1; cp = c0 & 0x1F; minv = 0x80; }
    else if (c0 >= 0xE0 && c0 <= 0xEF) { need = 2; cp = c0 & 0x0F; minv = 0x800; }
    else if (c0 >= 0xF0 && c0 <= 0xF4) { need = 3; cp = c0 & 0x07; minv = 0x10000; }
    else return kRuneError;
    for (int k = 1; k <= need; k++) {
        if (i + k >= s.size()) return kRuneError;
        const unsigned char c = (unsigned char)s[i + k];
        if ((c & 0xC0) != 0x80) return kRuneError;
        cp = (cp << 6) | (c & 0x3F);
    }
    if (cp < minv || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return kRuneError;
    *adv = (size_t)need + 1;
    return cp;
}

void encode_rune(int32_t cp, std::string& out) {
    if (cp < 0 || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) cp = kRuneError;
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

struct LowerPair { int32_t from, to; };
const LowerPair kLower[] = {
#include "unicode_lower.inc"
};

int32_t lower_rune(int32_t cp) {
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    const LowerPair* b = kLower;
    const LowerPair* e = kLower + sizeof(kLower) / sizeof(kLower[0]);
    const LowerPair* it = std::lower_bound(b, e, cp, [](const LowerPair& p, int32_t v) { return p.from < v; });
    return (it != e && it->from == cp) ? it->to : cp;
}

std::string rune_str(int32_t cp) { std::string s; encode_rune(cp, s); return s; }
}  // namespace

int32_t DecodeRune(const std::string& s, size_t i, size_t* adv) { return decode_rune(s, i, adv); }
void EncodeRune(int32_t cp, std::string& out) { encode_rune(cp, out); }

// ---- regex prefilter literals ---------------------------------------------------------------------------------------
std::vector<std::string> RegexRequiredLiterals(const std::string& p, size_t min_len) {
    std::vector<std::string> out;
    std::string cur;
    std::vector<size_t> starts;          // byte offsets in cur where each character (rune) of the run begins
    auto flush = [&]() {
        if (cur.size() >= min_len) out.push_back(cur);
        cur.clear(); starts.clear();
    };
    auto drop_last_char = [&]() {       // the last character carries a quantifier that allows zero repeats
        if (starts.empty()) return;
        cur.resize(starts.back());
        starts.pop_back();
    };
    auto push_bytes = [&](const char* b, size_t n) { starts.push_back(cur.size()); cur.append(b, n); };
    const size_t n = p.size();
    size_t i = 0;
    bool last_is_char = false;           // the previous atom is a literal character sitting at the end of cur
    while (i < n) {
        const unsigned char c = (unsigned char)p[i];
        bool atom_is_char = false;
        if (c == '\\') {
            if (i + 1 >= n) return {};
            const unsigned char e = (unsigned char)p[i + 1];
            if (e == 'd' || e == 'w' || e == 's' || e == 'D' || e == 'W' || e == 'S' || e == 'b' || e == 'B' || e == 'A' || e == 'z') {
                flush();
                i += 2;
            } else if ((e >= '0' && e <= '9') || (e >= 'a' && e <= 'z') || (e >= 'A' && e <= 'Z') || e >= 0x80) {
                return {};               // \x41, \pL, \Q..\E, \n, \t, octal ...: not modelled
            } else {
                push_bytes((const char*)&p[i + 1], 1);     // escaped punctuation is itself
                atom_is_char = true;
                i += 2;
            }
        } else if (c == '[') {
            // character class: skip to its closing bracket
            size_t j = i + 1;
            if (j < n && p[j] == '^') j++;
            if (j < n && p[j] == ']') j++;
            for (;;) {
                if (j >= n) return {};
                if (p[j] == '\\') { j += 2; continue; }
                if (p[j] == '[' && j + 1 < n && p[j + 1] == ':') {
                    const size_t k = p.find(":]", j + 2);
                    if (k == std::string::npos) return {};
                    j = k + 2;
                    continue;
                }
                if (p[j] == ']') break;
                j++;
            }
            flush();
            i = j + 1;
        } else if (c == '(') {
            if (i + 1 < n && p[i + 1] == '?') {
                // (?:...) and (?P<name>...) are plain groups; anything else sets flags
                if (!(i + 2 < n && (p[i + 2] == ':' || p[i + 2] == 'P'))) return {};
            }
            // a group is an unknown atom: skip to the matching parenthesis
            int depth = 0;
            size_t j = i;
            for (;;) {
                if (j >= n) return {};
                if (p[j] == '\\') { j += 2; continue; }
                if (p[j] == '[') {
                    size_t k = j + 1;
                    if (k < n && p[k] == '^') k++;
                    if (k < n && p[k] == ']') k++;
                    while (k < n && p[k] != ']') k += p[k] == '\\' ? 2 : 1;
                    if (k >= n) return {};
                    j = k + 1;
                    continue;
                }
                if (p[j] == '(') { if (j + 1 < n && p[j + 1] == '?' && !(j + 2 < n && (p[j + 2] == ':' || p[j + 2] == 'P'))) return {}; depth++; }
                if (p[j] == ')') { depth--; if (depth == 0) break; }
                j++;
            }
            flush();
            i = j + 1;
        } else if (c == ')') {
            return {};                   // unbalanced
        } else if (c == '|') {
            return {};                   // top-level alternation: no literal is required by every branch in general
        } else if (c == '.' || c == '^' || c == '$') {
            flush();
            i++;
        } else if (c == '*' || c == '?' || c == '+' || c == '{') {
            size_t j = i + 1;
            bool zero_ok = c == '*' || c == '?';
            if (c == '{') {
                // {m}, {m,}, {m,n}; anything else is a literal brace in RE2 -- not modelled
                size_t k = i + 1;
                uint64_t m = 0;
                bool digits = false;
                while (k < n && p[k] >= '0' && p[k] <= '9') { m = m * 10 + (uint64_t)(p[k] - '0'); k++; digits = true; if (m > 1000000) return {}; }
                if (!digits) return {};
                if (k < n && p[k] == ',') { k++; while (k < n && p[k] >= '0' && p[k] <= '9') k++; }
                if (k >= n || p[k] != '}') return {};
                zero_ok = m == 0;
                j = k + 1;
            }
            if (j < n && p[j] == '?') j++;       // lazy form
            if (last_is_char) {
                if (zero_ok) drop_last_char();
                flush();                          // the run ends at the repeated character either way
            }
            i = j;
        } else {
            // an ordinary character (all bytes of a UTF-8 sequence belong to one character)
            size_t adv = 1;
            if (c >= 0x80) { (void)decode_rune(p, i, &adv); }
            push_bytes(&p[i], adv);
            atom_is_char = true;
            i += adv;
        }
        last_is_char = atom_is_char;
    }
    flush();
    return out;
}

bool IsAscii(const std::string& s) {
    unsigned char acc = 0;
    for (unsigned char c : s) acc |= c;
    return acc < 0x80;
}

std::string ToLower(const std::string& s) {
    std::string out;
    if (IsAscii(s)) {
        out.resize(s.size());
        const char* in = s.data();
        char* o = &out[0];
        for (size_t i = 0; i < s.size(); i++) {                  // branch-free: vectorises
            const unsigned char c = (unsigned char)in[i];
            o[i] = (char)(c + (((unsigned char)(c - 'A') < 26) ? 32 : 0));
        }
        return out;
    }
    out.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        size_t adv;
        int32_t cp = decode_rune(s, i, &adv);
        encode_rune(lower_rune(cp), out);
        i += adv;
    }
    return out;
}

// ---- scanner (dsl/scanner.go:79-250) ---------------------------------------------------------------------
int32_t Scanner::read() {
    if (i_ >= s_.size()) { last_ = 0; return 0; }
    size_t adv;
    int32_t cp = decode_rune(s_, i_, &adv);
    i_ += adv;
    last_ = adv;
    return cp;
}
void Scanner::unread() { i_ -= last_; last_ = 0; }

static bool is_ws(int32_t c) { return c == ' ' || c == '\t' || c == '\n'; }
static bool is_letter(int32_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

ScanResult Scanner::Scan() {
    ScanResult r;
    const int32_t ch = read();
    if (is_ws(ch)) { unread(); return scan_whitespace(); }
    if (ch == '"') { unread(); return scan_keyword(false); }
    if (is_letter(ch)) { unread(); return scan_operators(); }
    if (ch == '(') { r.tok = OPPAR; r.lit = "("; return r; }
    if (ch == ')') { r.tok = CLPAR; r.lit = ")"; return r; }
    if (ch == 0) { r.tok = END_OF_INPUT; return r; }
    r.err = "illegal char was found " + rune_str(ch);
    return r;
}

ScanResult Scanner::scan_whitespace() {
    ScanResult r;
    encode_rune(read(), r.lit);
    for (;;) {
        const int32_t ch = read();
        if (ch == 0) break;
        if (!is_ws(ch)) { unread(); break; }
        encode_rune(ch, r.lit);
    }
    r.tok = WS;
    return r;
}

ScanResult Scanner::scan_operators() {
    ScanResult r;
    int32_t ch = read();
    if (!is_letter(ch)) { r.err = "fail to scan operator: expected letter but found " + rune_str(ch); return r; }
    std::string lit(1, (char)ch);
    for (;;) {
        ch = read();
        if (ch == 0) break;
        if (!is_letter(ch)) { unread(); break; }
        lit.push_back((char)ch);
    }
    std::string up = lit;
    for (char& c : up) if (c >= 'a' && c <= 'z') c -= 32;
    r.lit = lit;
    if (up == "AND") r.tok = AND;
    else if (up == "OR") r.tok = OR;
    else if (up == "NOT") r.tok = NOT;
    else if (up == "INORD") r.tok = INORD;
    else if (up == "R") return scan_keyword(true);
    else { r.lit.clear(); r.err = "failed to scan operator: unexpected operator '" + lit + "' found"; }
    return r;
}

ScanResult Scanner::scan_keyword(bool is_regex) {
    ScanResult r;
    const std::string kind = is_regex ? "regex" : "keyword";
    int32_t ch = read();
    if (ch != '"') { r.err = "fail to scan " + kind + ": expected \" but found " + rune_str(ch); return r; }
    std::string buf;
    for (;;) {
        ch = read();
        if (ch == 0) { r.err = "fail to scan " + kind + ": expected \" but found EOF"; return r; }
        if (ch == '\\') {
            const int32_t sc = read();
            switch (sc) {
            case '\\': buf.push_back('\\'); break;
            case 'n': buf.push_back('\n'); break;
            case 'r': buf.push_back('\r'); break;
            case 't': buf.push_back('\t'); break;
            case '"': buf.push_back('"'); break;
            default: r.err = "fail to scan " + kind + ": invalid escaped char " + rune_str(sc); return r;
            }
        } else if (ch == '"') {
            break;
        } else {
            encode_rune(ch, buf);
        }
    }
    r.tok = is_regex ? REGEX : KEYWORD;
    r.lit = buf;
    return r;
}

// ---- parser (dsl/parser.go:58-315) -------------------------------------------------------------------------
namespace {
class Parser {
public:
    Parser(const std::string& src, bool cs) : sc_(src), cs_(cs) {}
    std::unique_ptr<Expression> parse(std::string& err);
    std::vector<std::string> keywords, regexes;

private:
    // one token of look-back, like Parser.buf / unscan (parser.go:254-276)
    bool scan(Token& tok, std::string& lit, std::string& err) {
        if (unscanned_) { unscanned_ = false; tok = btok_; lit = blit_; return true; }
        ScanResult r = sc_.Scan();
        if (!r.err.empty()) { err = r.err; return false; }
        btok_ = tok = r.tok; blit_ = lit = r.lit;
        return true;
    }
    bool scan_skip_ws(Token& tok, std::string& lit, std::string& err) {
        if (!scan(tok, lit, err)) return false;
        if (tok == WS) return scan(tok, lit, err);
        return true;
    }
    void unscan() { unscanned_ = true; }
    std::unique_ptr<Expression> open_par(std::string& err);
    bool dual_op(std::unique_ptr<Expression>& exp, ExprType t, std::string& err);
    void add_literal(Token tok, const std::string& lit) {
        auto& v = tok == REGEX ? regexes : keywords;
        if (std::find(v.begin(), v.end(), lit) == v.end()) v.push_back(lit);
    }
    static void attach(Expression& exp, std::unique_ptr<Expression> n) {
        if (!exp.LExpr) exp.LExpr = std::move(n); else exp.RExpr = std::move(n);
    }

    Scanner sc_;
    bool cs_;
    bool inord_ = false, unscanned_ = false;
    int par_count_ = 0;
    Token btok_ = ILLEGAL;
    std::string blit_;
};

std::unique_ptr<Expression> Parser::open_par(std::string& err) {
    const int lvl = par_count_;
    par_count_++;
    auto e = parse(err);
    if (!err.empty()) return nullptr;
    if (par_count_ != lvl) { err = "invalid expression: Unexpected '('"; return nullptr; }
    return e;
}

bool Parser::dual_op(std::unique_ptr<Expression>& exp, ExprType t, std::string& err) {
    if (!exp->LExpr) {
        err = std::string("invalid expression: no left expression was found for ") + expr_type_name(t);
        return false;
    }
    if (!exp->RExpr) { exp->Type = t; return true; }
    auto n = std::make_unique<Expression>();
    n->Type = t; n->Inord = inord_;
    n->LExpr = std::move(exp);
    exp = std::move(n);
    Token tok; std::string lit;
    if (!scan_skip_ws(tok, lit, err)) return false;
    if (tok == OPPAR) {
        auto g = open_par(err);
        if (!err.empty()) return false;
        exp->RExpr = std::move(g);
    } else {
        unscan();
    }
    return true;
}

std::unique_ptr<Expression> Parser::parse(std::string& err) {
    auto exp = std::make_unique<Expression>();
    exp->Inord = inord_;
    for (;;) {
        Token tok; std::string lit;
        if (!scan_skip_ws(tok, lit, err)) return nullptr;
        switch (tok) {
        case OPPAR: {
            auto g = open_par(err);
            if (!err.empty()) return nullptr;
            attach(*exp, std::move(g));
            break;
        }
        case KEYWORD:
        case REGEX: {
            if (!cs_) lit = ToLower(lit);
            auto u = std::make_unique<Expression>();
            u->Type = UNIT_EXPR; u->Literal = lit; u->Inord = inord_;
            attach(*exp, std::move(u));
            add_literal(tok, lit);
            break;
        }
        case AND:
            if (!dual_op(exp, AND_EXPR, err)) return nullptr;
            break;
        case OR:
            if (!dual_op(exp, OR_EXPR, err)) return nullptr;
            break;
        case NOT: {
            if (inord_) { err = "invalid expression: INORD operator must not contain NOT operator"; return nullptr; }
            Token nt; std::string nl;
            if (!scan_skip_ws(nt, nl, err)) return nullptr;
            auto ne = std::make_unique<Expression>();
            ne->Type = NOT_EXPR;
            if (nt == KEYWORD || nt == REGEX) {
                if (!cs_) nl = ToLower(nl);
                ne->RExpr = std::make_unique<Expression>();
                ne->RExpr->Type = UNIT_EXPR; ne->RExpr->Literal = nl;
                add_literal(nt, nl);
            } else if (nt == OPPAR) {
                auto g = open_par(err);
                if (!err.empty()) return nullptr;
                ne->RExpr = std::move(g);
            } else {
                err = std::string("invalid expression: Unexpected token '") + token_name(nt) + "' after NOT";
                return nullptr;
            }
            attach(*exp, std::move(ne));
            break;
        }
        case INORD: {
            if (inord_) { err = "invalid expression: INORD operator must not contain INORD operator"; return nullptr; }
            Token nt; std::string nl;
            if (!scan_skip_ws(nt, nl, err)) return nullptr;
            if (nt != OPPAR) {
                err = std::string("invalid expression: Unexpected token '") + token_name(nt) + "' after INORD";
                return nullptr;
            }
            auto ie = std::make_unique<Expression>();
            ie->Type = INORD_EXPR;
            inord_ = true;
            auto g = open_par(err);
            if (!err.empty()) return nullptr;
            inord_ = false;
            ie->RExpr = std::move(g);
            attach(*exp, std::move(ie));
            break;
        }
        case CLPAR:
            par_count_--;
            /* fallthrough */
        case END_OF_INPUT: {
            if (par_count_ < 0) {
                err = "invalid expression: unexpected EOF found. Extra closing parentheses: " + std::to_string(-par_count_);
                return nullptr;
            }
            std::unique_ptr<Expression> fin;
            if (exp->Type == UNSET_EXPR) {
                if (exp->RExpr) fin = std::move(exp->RExpr);
                else if (exp->LExpr) fin = std::move(exp->LExpr);
                else { err = "invalid expression: unexpected EOF found"; return nullptr; }
            } else {
                fin = std::move(exp);
            }
            if ((fin->Type == AND_EXPR || fin->Type == OR_EXPR) && !fin->RExpr) {
                err = std::string("invalid expression: incomplete expression ") + expr_type_name(fin->Type);
                return nullptr;
            }
            return fin;
        }
        default:
            err = "invalid expression: Unexpected operator was found (" + std::to_string((int)tok) + " = '" + lit + "')";
            return nullptr;
        }
    }
}
}  // namespace

ParseResult Parse(const std::string& src, bool case_sensitive) {
    ParseResult r;
    Parser p(src, case_sensitive);
    r.expr = p.parse(r.err);
    if (!r.err.empty()) r.expr.reset();
    r.keywords = std::move(p.keywords);
    r.regexes = std::move(p.regexes);
    return r;
}

// Expression.solve evaluates every node (no short-circuit, dsl/expression.go:74-137), so whether Solve returns
// an error does not depend on the document: it happens iff the tree holds a node the switch cannot handle.  The
// parser can build one: `"a" "b" and "c"` leaves an UNSET node holding both literals (parser.go:220-233).
std::string SolveError(const Expression& root) {
    std::vector<const Expression*> st{&root};
    while (!st.empty()) {
        const Expression* e = st.back();
        st.pop_back();
        switch (e->Type) {
        case UNIT_EXPR: break;
        case AND_EXPR:
        case OR_EXPR:
            if (!e->LExpr || !e->RExpr)
                return std::string(expr_type_name(e->Type)) + " statment do not have rigth or left expression";
            st.push_back(e->RExpr.get()); st.push_back(e->LExpr.get());   // left is evaluated first
            break;
        case NOT_EXPR:
        case INORD_EXPR:
            if (!e->RExpr) return std::string(expr_type_name(e->Type)) + " statement do not have expression";
            st.push_back(e->RExpr.get());
            break;
        default:
            return "unable to process expression type " + std::to_string((int)e->Type);
        }
    }
    return "";
}

void CompileProgram(const Expression& root, const std::function<uint32_t(const std::string&)>& slot_of,
                    std::vector<uint32_t>& out) {
    // iterative post-order: expressions such as the reference benchmark's exp10000 are 10 000 levels deep
    struct Frame { const Expression* e; int stage; };
    std::vector<Frame> st;
    st.push_back({&root, 0});
    while (!st.empty()) {
        Frame& f = st.back();
        const Expression* e = f.e;
        const uint32_t fl = e->Inord ? GFT_INORD_FLAG : 0u;
        switch (e->Type) {
        case UNIT_EXPR:
            out.push_back(GFT_OP_UNIT << 28 | fl | (slot_of(e->Literal) & GFT_SLOT_MASK));
            st.pop_back();
            break;
        case AND_EXPR:
        case OR_EXPR:
            if (f.stage == 0) { f.stage = 1; st.push_back({e->LExpr.get(), 0}); }
            else if (f.stage == 1) { f.stage = 2; st.push_back({e->RExpr.get(), 0}); }
            else { out.push_back((e->Type == AND_EXPR ? GFT_OP_AND : GFT_OP_OR) << 28 | fl); st.pop_back(); }
            break;
        case NOT_EXPR:
        case INORD_EXPR:
            if (f.stage == 0) { f.stage = 1; st.push_back({e->RExpr.get(), 0}); }
            else { out.push_back((e->Type == NOT_EXPR ? GFT_OP_NOT : GFT_OP_INORD) << 28); st.pop_back(); }
            break;
        default:
            st.pop_back();   // UNSET cannot be produced by Parse
            break;
        }
    }
}

void json_str(const std::string& s, std::string& o) {
    o.push_back('"');
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
        else if (c == '\n') o += "\\n";
        else if (c == '\r') o += "\\r";
        else if (c == '\t') o += "\\t";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
    o.push_back('"');
}

static void to_json(const Expression& e, std::string& o) {
    o += "{\"Type\":\""; o += expr_type_name(e.Type); o += "\"";
    if (!e.Literal.empty()) { o += ",\"Literal\":"; json_str(e.Literal, o); }
    if (e.Inord) o += ",\"Inord\":true";
    if (e.LExpr) { o += ",\"LExpr\":"; to_json(*e.LExpr, o); }
    if (e.RExpr) { o += ",\"RExpr\":"; to_json(*e.RExpr, o); }
    o += "}";
}

std::string ToJson(const Expression& e) { std::string o; to_json(e, o); return o; }

}  // namespace dsl
}  // namespace gft

// ---- host-only C ABI (include/gft.h "DSL front-end alone") ------------------------------------------------------
namespace {
int emit_out(const std::string& doc, char* out, uint64_t cap, uint64_t* needed) {
    if (needed) *needed = doc.size() + 1;
    if (!out || cap < doc.size() + 1) return GFT_E_INVALID;
    memcpy(out, doc.data(), doc.size());
    out[doc.size()] = 0;
    return GFT_OK;
}
void json_list(const std::vector<std::string>& v, std::string& o) {
    o.push_back('[');
    for (size_t i = 0; i < v.size(); i++) {
        if (i) o.push_back(',');
        gft::dsl::json_str(v[i], o);
    }
    o.push_back(']');
}
}  // namespace

extern "C" {

int gft_regex_required_literals(const uint8_t* pattern, uint64_t len, char* out, uint64_t cap, uint64_t* needed) try {
    std::string doc;
    json_list(gft::dsl::RegexRequiredLiterals(std::string((const char*)pattern, (size_t)len)), doc);
    return emit_out(doc, out, cap, needed);
} GFT_CATCH(nullptr)

int gft_dsl_parse(const uint8_t* expr, uint64_t len, int case_sensitive, char* out, uint64_t cap, uint64_t* needed) try {
    using namespace gft::dsl;
    ParseResult r = Parse(std::string((const char*)expr, (size_t)len), case_sensitive != 0);
    std::string doc;
    if (!r.err.empty()) {
        doc = "{\"error\":";
        json_str(r.err, doc);
        doc += "}";
        return emit_out(doc, out, cap, needed);
    }
    std::vector<std::string> lits = r.keywords;
    for (const auto& g : r.regexes)
        if (std::find(lits.begin(), lits.end(), g) == lits.end()) lits.push_back(g);
    std::vector<uint32_t> prog;
    if (SolveError(*r.expr).empty())
        CompileProgram(*r.expr, [&](const std::string& l) {
            return (uint32_t)(std::find(lits.begin(), lits.end(), l) - lits.begin());
        }, prog);
    doc = "{\"tree\":" + ToJson(*r.expr) + ",\"keywords\":";
    json_list(r.keywords, doc);
    doc += ",\"regexes\":";
    json_list(r.regexes, doc);
    doc += ",\"solve_error\":";
    { std::string se = SolveError(*r.expr); if (se.empty()) doc += "null"; else json_str(se, doc); }
    doc += ",\"program\":[";
    for (size_t i = 0; i < prog.size(); i++) { if (i) doc.push_back(','); doc += std::to_string(prog[i]); }
    doc += "]}";
    return emit_out(doc, out, cap, needed);
} GFT_CATCH(nullptr)

int gft_dsl_tokens(const uint8_t* expr, uint64_t len, char* out, uint64_t cap, uint64_t* needed) try {
    using namespace gft::dsl;
    const std::string src((const char*)expr, (size_t)len);
    Scanner sc(src);
    std::string doc = "[";
    for (int n = 0;; n++) {
        ScanResult r = sc.Scan();
        if (n) doc.push_back(',');
        doc += "{\"Tok\":\""; doc += token_name(r.tok); doc += "\",\"Lit\":";
        json_str(r.lit, doc);
        doc += ",\"Err\":";
        if (r.err.empty()) doc += "null"; else json_str(r.err, doc);
        doc += "}";
        if (!r.err.empty() || r.tok == END_OF_INPUT) break;
    }
    doc += "]";
    return emit_out(doc, out, cap, needed);
} GFT_CATCH(nullptr)

int gft_to_lower(const uint8_t* in, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* needed) try {
    const std::string r = gft::dsl::ToLower(std::string((const char*)in, (size_t)len));
    if (needed) *needed = r.size();
    if (cap < r.size() || (!out && r.size())) return GFT_E_INVALID;
    if (r.size()) memcpy(out, r.data(), r.size());
    return GFT_OK;
} GFT_CATCH(nullptr)

}  // extern "C"
