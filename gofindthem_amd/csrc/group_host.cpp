#include "group_host.hpp"

#include <mutex>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

#include "gft_guard.hpp"

namespace gft {
namespace gdsl {

const char* token_name(Token t) {
    static const char* n[] = {"ILLEGAL", "EOF", "WS", "TAG", "FIELD_PATH", "QUOTATION", "OPPAR", "CLPAR", "AND", "OR", "NOT"};
    return (int)t >= 0 && (int)t < 11 ? n[t] : "UNEXPECTED";
}
const char* expr_type_name(ExprType t) {
    static const char* n[] = {"UNSET", "AND", "OR", "NOT", "UNIT"};
    return (int)t >= 0 && (int)t < 5 ? n[t] : "UNEXPECTED";
}

namespace {
std::string rune_str(int32_t cp) { std::string s; dsl::EncodeRune(cp, s); return s; }
bool is_ws(int32_t c) { return c == ' ' || c == '\t' || c == '\n'; }
bool is_letter(int32_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
std::string trim_spaces(const std::string& s) {          // strings.Trim(s, " ")
    size_t a = 0, b = s.size();
    while (a < b && s[a] == ' ') a++;
    while (b > a && s[b - 1] == ' ') b--;
    return s.substr(a, b - a);
}
}  // namespace

// ---- scanner (group/dsl/scanner.go) -----------------------------------------------------------------------
int32_t Scanner::read() {                 // rune 0 doubles as the end marker (scanner.go:263)
    if (i_ >= s_.size()) { last_ = 0; return 0; }
    size_t adv;
    const int32_t cp = dsl::DecodeRune(s_, i_, &adv);
    i_ += adv;
    last_ = adv;
    return cp;
}
void Scanner::unread() { i_ -= last_; last_ = 0; }

ScanResult Scanner::Scan() {              // scanner.go:78-109
    ScanResult r;
    const int32_t ch = read();
    if (is_ws(ch)) { unread(); return scan_whitespace(); }
    if (ch == '"') { unread(); return scan_tag(); }
    if (ch == ':') { unread(); return scan_field_path(); }
    if (is_letter(ch)) { unread(); return scan_operators(); }
    if (ch == '(') { r.tok = OPPAR; r.lit = "("; return r; }
    if (ch == ')') { r.tok = CLPAR; r.lit = ")"; return r; }
    if (ch == 0) { r.tok = END_OF_INPUT; return r; }
    r.err = "illegal char was found " + rune_str(ch);
    return r;
}

ScanResult Scanner::scan_whitespace() {   // scanner.go:112-131
    ScanResult r;
    dsl::EncodeRune(read(), r.lit);
    for (;;) {
        const int32_t ch = read();
        if (ch == 0) break;
        if (!is_ws(ch)) { unread(); break; }
        dsl::EncodeRune(ch, r.lit);
    }
    r.tok = WS;
    return r;
}

ScanResult Scanner::scan_operators() {    // scanner.go:134-172
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
    for (char& c : up) if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    if (up == "AND") r.tok = AND;
    else if (up == "OR") r.tok = OR;
    else if (up == "NOT") r.tok = NOT;
    else { r.err = "failed to scan operator: unexpected operator '" + lit + "' found"; return r; }
    r.lit = lit;
    return r;
}

ScanResult Scanner::scan_tag() {          // scanner.go:177-210
    ScanResult r;
    int32_t ch = read();
    if (ch != '"') { r.err = "fail to scan tag: expected \" but found " + rune_str(ch); return r; }
    std::string buf;
    for (;;) {
        ch = read();
        if (ch == 0) { r.err = "fail to scan tag: expected ':' but found EOF"; return r; }
        if (ch == '\\') {
            const int32_t esc = read();
            if (esc == '\\' || esc == '"' || esc == ':') dsl::EncodeRune(esc, buf);
            else { r.err = "fail to scan tag: invalid escaped char " + rune_str(esc); return r; }
            continue;
        }
        if (ch == ':') { unread(); break; }      // the field path is the next token
        if (ch == '"') break;
        dsl::EncodeRune(ch, buf);
    }
    r.lit = trim_spaces(buf);
    r.tok = TAG;
    return r;
}

ScanResult Scanner::scan_field_path() {   // scanner.go:215-244
    ScanResult r;
    int32_t ch = read();
    if (ch != ':') { r.err = "fail to scan field: expected ':' but found " + rune_str(ch); return r; }
    std::string buf;
    for (;;) {
        ch = read();
        if (ch == 0) { r.err = "fail to scan field: expected '\"' but found EOF"; return r; }
        if (ch == '\\') {
            const int32_t esc = read();
            if (esc == '\\' || esc == '"') dsl::EncodeRune(esc, buf);
            else { r.err = "fail to scan field: invalid escaped char " + rune_str(esc); return r; }
            continue;
        }
        if (ch == '"') break;
        dsl::EncodeRune(ch, buf);
    }
    r.lit = trim_spaces(buf);
    r.tok = FIELD_PATH;
    return r;
}

// ---- parser (group/dsl/parser.go) --------------------------------------------------------------------------
namespace {

struct Parser {
    Scanner s;
    struct { Token tok = ILLEGAL; std::string lit; bool unscanned = false; } buf;
    int parCount = 0;
    std::vector<std::string> tags, fields;

    explicit Parser(const std::string& src) : s(src) {}

    static void add_unique(std::vector<std::string>& v, const std::string& x) {
        for (const auto& y : v) if (y == x) return;
        v.push_back(x);
    }

    ScanResult scan() {                                   // parser.go:204-219
        if (buf.unscanned) { buf.unscanned = false; ScanResult r; r.tok = buf.tok; r.lit = buf.lit; return r; }
        ScanResult r = s.Scan();
        if (!r.err.empty()) return r;
        buf.tok = r.tok; buf.lit = r.lit;
        return r;
    }
    void unscan() { buf.unscanned = true; }
    ScanResult scan_ignore_ws() {                         // parser.go:226-235
        ScanResult r = scan();
        if (!r.err.empty()) return r;
        if (r.tok == WS) r = scan();
        return r;
    }

    std::string parse_tag_info(TagInfo& tag) {            // parser.go:252-278
        ScanResult r = scan_ignore_ws();
        if (!r.err.empty()) return r.err;
        if (r.tok != TAG) return std::string("invalid expression: Expecting TAG but found ") + token_name(r.tok);
        if (r.lit.empty()) return "invalid expression: Found empty TAG";
        tag.Name = r.lit;
        ScanResult n = scan_ignore_ws();
        if (!n.err.empty()) return n.err;
        if (n.tok != FIELD_PATH) { unscan(); return ""; }
        tag.FieldPath = n.lit;
        return "";
    }

    void note(const TagInfo& tag) {
        add_unique(tags, tag.Name);
        if (!tag.FieldPath.empty()) add_unique(fields, tag.FieldPath);
    }

    std::string handle_open_par(std::unique_ptr<Expression>& out) {   // parser.go:238-249
        const int parlvl = parCount;
        parCount++;
        std::string err = parse(out);
        if (!err.empty()) return err;
        if (parCount != parlvl) return "invalid expression: Unexpected '('";
        return "";
    }

    // parser.go:178-201; exp is replaced by the node the caller continues with
    std::string handle_dual_op(std::unique_ptr<Expression>& exp, ExprType type) {
        if (!exp->LExpr) return std::string("invalid expression: no left expression was found for ") + expr_type_name(type);
        if (!exp->RExpr) { exp->Type = type; return ""; }
        std::unique_ptr<Expression> up(new Expression());
        up->Type = type;
        up->LExpr = std::move(exp);
        exp = std::move(up);
        ScanResult n = scan_ignore_ws();
        if (!n.err.empty()) return n.err;
        if (n.tok == OPPAR) {
            std::unique_ptr<Expression> sub;
            std::string err = handle_open_par(sub);
            if (!err.empty()) return err;
            exp->RExpr = std::move(sub);
        } else {
            unscan();
        }
        return "";
    }

    static void attach(Expression& exp, std::unique_ptr<Expression> child) {
        if (!exp.LExpr) exp.LExpr = std::move(child); else exp.RExpr = std::move(child);
    }

    std::string parse(std::unique_ptr<Expression>& out) {  // parser.go:41-175
        std::unique_ptr<Expression> exp(new Expression());
        for (;;) {
            ScanResult r = scan_ignore_ws();
            if (!r.err.empty()) return r.err;
            switch (r.tok) {
            case OPPAR: {
                std::unique_ptr<Expression> sub;
                std::string err = handle_open_par(sub);
                if (!err.empty()) return err;
                attach(*exp, std::move(sub));
                break;
            }
            case TAG: {
                unscan();
                TagInfo tag;
                std::string err = parse_tag_info(tag);
                if (!err.empty()) return err;
                std::unique_ptr<Expression> unit(new Expression());
                unit->Type = UNIT_EXPR;
                unit->Tag = tag;
                attach(*exp, std::move(unit));
                note(tag);
                break;
            }
            case AND:
            case OR: {
                std::string err = handle_dual_op(exp, r.tok == AND ? AND_EXPR : OR_EXPR);
                if (!err.empty()) return err;
                break;
            }
            case NOT: {
                ScanResult n = scan_ignore_ws();
                if (!n.err.empty()) return n.err;
                std::unique_ptr<Expression> neg(new Expression());
                neg->Type = NOT_EXPR;
                if (n.tok == TAG) {
                    unscan();
                    TagInfo tag;
                    std::string err = parse_tag_info(tag);
                    if (!err.empty()) return err;
                    neg->RExpr.reset(new Expression());
                    neg->RExpr->Type = UNIT_EXPR;
                    neg->RExpr->Tag = tag;
                    note(tag);
                } else if (n.tok == OPPAR) {
                    std::unique_ptr<Expression> sub;
                    std::string err = handle_open_par(sub);
                    if (!err.empty()) return err;
                    neg->RExpr = std::move(sub);
                } else {
                    return std::string("invalid expression: Unexpected token '") + token_name(n.tok) + "' after NOT";
                }
                attach(*exp, std::move(neg));
                break;
            }
            case CLPAR:
                parCount--;
                // fall through
            case END_OF_INPUT: {
                if (parCount < 0)
                    return "invalid expression: unexpected EOF found. Extra closing parentheses: " + std::to_string(-parCount);
                std::unique_ptr<Expression> fin;
                if (exp->Type == UNSET_EXPR) {
                    if (exp->RExpr) fin = std::move(exp->RExpr);
                    else if (exp->LExpr) fin = std::move(exp->LExpr);
                    else return "invalid expression: unexpected EOF found";
                } else {
                    fin = std::move(exp);
                }
                if ((fin->Type == AND_EXPR || fin->Type == OR_EXPR) && !fin->RExpr)
                    return std::string("invalid expression: incomplete expression ") + expr_type_name(fin->Type);
                out = std::move(fin);
                return "";
            }
            default:
                return "invalid expression: Unexpected operator was found (" + std::to_string((int)r.tok) + " = '" + r.lit + "')";
            }
        }
    }
};

}  // namespace

ParseResult Parse(const std::string& src) {
    ParseResult res;
    Parser p(src);
    res.err = p.parse(res.expr);
    if (!res.err.empty()) res.expr.reset();
    res.tags = p.tags;
    res.fields = p.fields;
    return res;
}

bool Solve(const Expression& e, const TagMap& m, std::string& err) {
    return SolveWith(e, [&](const Expression& u) {
        auto it = m.find(u.Tag.Name);
        if (it == m.end()) return false;
        if (u.Tag.FieldPath.empty()) return true;
        for (const auto& fp : it->second)
            if (fp.first.compare(0, u.Tag.FieldPath.size(), u.Tag.FieldPath) == 0) return true;
        return false;
    }, err);
}

std::string ToJson(const Expression& e) {
    std::string o = "{\"Type\":\"";
    o += expr_type_name(e.Type);
    o += "\"";
    if (e.Type == UNIT_EXPR) {
        o += ",\"Tag\":{\"Name\":";
        dsl::json_str(e.Tag.Name, o);
        o += ",\"FieldPath\":";
        dsl::json_str(e.Tag.FieldPath, o);
        o += "}";
    }
    if (e.LExpr) { o += ",\"LExpr\":"; o += ToJson(*e.LExpr); }
    if (e.RExpr) { o += ",\"RExpr\":"; o += ToJson(*e.RExpr); }
    o += "}";
    return o;
}

}  // namespace gdsl

// ---- GroupFinder (group/finder/finder.go, internal.go) ----------------------------------------------------------
bool IsValidFieldPath(const std::string& fieldPath, const std::vector<std::string>& includePaths,
                      const std::vector<std::string>& excludePaths) {
    for (const auto& x : excludePaths)
        if (fieldPath.compare(0, x.size(), x) == 0) return false;
    if (!includePaths.empty()) {
        for (const auto& x : includePaths)
            if (fieldPath.compare(0, x.size(), x) == 0) return true;
        return false;
    }
    return true;
}

Error GroupFinder::AddRule(const std::string& ruleName, const std::vector<std::string>& expressions) {   // finder.go:45-66
    for (const auto& raw : expressions) {
        gdsl::ParseResult pr = gdsl::Parse(raw);
        if (!pr.err.empty()) return pr.err;
        ExpressionWrapper w;
        w.ExpressionString = raw;
        w.Expression = std::move(pr.expr);
        rules_[ruleName].push_back(std::move(w));
        for (const auto& t : pr.tags) tags_.insert(t);
        for (const auto& f : pr.fields) fields_.insert(f);
    }
    return "";
}

namespace {
struct Leaf { uint32_t doc; std::string path; const std::string* text; };

// getRulesInfo (internal.go:9-97) over a decoded JSON value: strings are leaves, objects extend the path with
// ".key", arrays with ".index(i)"; numbers, booleans and null are not taggable
void walk(const json::Value& root, const std::string& root_path, uint32_t doc, const std::vector<std::string>& inc,
          const std::vector<std::string>& exc, std::vector<Leaf>& out) {
    // depth first, children in document order, with an explicit stack (documents nest up to 10 000 levels)
    struct Item { const json::Value* v; std::string path; };
    std::vector<Item> todo;
    todo.push_back(Item{&root, root_path});
    while (!todo.empty()) {
        Item it = std::move(todo.back());
        todo.pop_back();
        const json::Value& v = *it.v;
        const std::string& path = it.path;
        switch (v.kind) {
        case json::Value::String:
            if (IsValidFieldPath(path, inc, exc)) out.push_back(Leaf{doc, path, &v.str});
            break;
        case json::Value::Object:
            for (size_t i = v.obj.size(); i-- > 0;) {          // pushed in reverse: popped in document order
                if (!v.last_wins(i)) continue;                   // a Go map keeps the last duplicate
                todo.push_back(Item{&v.obj[i].second, path.empty() ? v.obj[i].first : path + "." + v.obj[i].first});
            }
            break;
        case json::Value::Array:
            for (size_t i = v.arr.size(); i-- > 0;) {
                const std::string fn = "index(" + std::to_string(i) + ")";
                todo.push_back(Item{&v.arr[i], path.empty() ? fn : path + "." + fn});
            }
            break;
        default:
            break;
        }
    }
}

unsigned host_threads() {
    if (const char* e = getenv("GFT_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? std::min(hc, 16u) : 4u;
}
template <class F>
void parallel_for(uint64_t n, F&& body) {            // body(index, worker)
    const unsigned nt = (unsigned)std::min<uint64_t>(host_threads(), (n + 63) / 64);
    if (nt <= 1) { for (uint64_t i = 0; i < n; i++) body(i, 0u); return; }
    std::atomic<uint64_t> next(0);
    std::vector<std::thread> pool;
    pool.reserve(nt);
    // an exception inside a worker would be std::terminate: the first one is carried to the calling thread and thrown
    // again there (the entry point's barrier turns it into a status), the other workers stop taking work
    std::exception_ptr first;
    std::mutex first_mu;
    {
        gft::JoinAll joined(pool);           // (also when a worker could not be started)
        for (unsigned t = 0; t < nt; t++)
            pool.emplace_back([&, t]() noexcept {
                try {
                    for (;;) {
                        const uint64_t b = next.fetch_add(64);
                        if (b >= n) return;
                        for (uint64_t i = b; i < std::min(n, b + 64); i++) body(i, t);
                    }
                } catch (...) {
                    next.store(n);
                    std::lock_guard<std::mutex> g(first_mu);
                    if (!first) first = std::current_exception();
                }
            });
    }
    if (first) std::rethrow_exception(first);
}
void resolve_tags(const gdsl::Expression& e, const std::unordered_map<std::string, uint32_t>& ids) {
    if (e.Type == gdsl::UNIT_EXPR) { auto it = ids.find(e.Tag.Name); e.tag_id = it == ids.end() ? -1 : (int32_t)it->second; }
    if (e.LExpr) resolve_tags(*e.LExpr, ids);
    if (e.RExpr) resolve_tags(*e.RExpr, ids);
}
}  // namespace

Error GroupFinder::ProcessJsons(const uint8_t* jblob, const uint64_t* doc_off, uint64_t n_docs,
                                const std::vector<std::string>& includePaths, const std::vector<std::string>& excludePaths,
                                bool want_tags, std::vector<DocResult>& out) {
    out.assign(n_docs, DocResult());
    // 1. decode + walk, a document per task
    std::vector<json::Value> docs(n_docs);
    std::vector<std::vector<Leaf>> doc_leaves(n_docs);
    parallel_for(n_docs, [&](uint64_t d, unsigned) {
        out[d].err = json::Parse((const char*)jblob + doc_off[d], (size_t)(doc_off[d + 1] - doc_off[d]), docs[d]);
        if (out[d].err.empty()) walk(docs[d], "", (uint32_t)d, includePaths, excludePaths, doc_leaves[d]);
    });
    // 2. every leaf is one document of one batch
    std::vector<uint64_t> first(n_docs + 1, 0);
    for (uint64_t d = 0; d < n_docs; d++) first[d + 1] = first[d] + doc_leaves[d].size();
    const uint64_t n_leaves = first[n_docs];
    std::vector<uint64_t> off(n_leaves + 1, 0);
    for (uint64_t d = 0; d < n_docs; d++)
        for (size_t k = 0; k < doc_leaves[d].size(); k++) off[first[d] + k + 1] = doc_leaves[d][k].text->size();
    for (uint64_t i = 0; i < n_leaves; i++) off[i + 1] += off[i];
    std::vector<uint8_t> blob(off[n_leaves] + 64, 0);
    parallel_for(n_docs, [&](uint64_t d, unsigned) {
        for (size_t k = 0; k < doc_leaves[d].size(); k++)
            memcpy(blob.data() + off[first[d] + k], doc_leaves[d][k].text->data(), doc_leaves[d][k].text->size());
    });
    last_leaves = n_leaves;
    last_bytes = off[n_leaves];
    const auto& exprs = findthem_->expressions();
    const uint32_t words = (uint32_t)((exprs.size() + 31) / 32);
    std::vector<uint32_t> bitmap((size_t)n_leaves * words + 1, 0);
    if (n_leaves) {
        Error err = findthem_->ProcessTexts(blob.data(), off.data(), n_leaves, bitmap.data());
        if (!err.empty()) {
            // the reference aborts the walk of a document at its first failing ProcessText (internal.go:29-31);
            // the finder's errors (engine build / find, unsolvable expression) do not depend on the text
            for (uint64_t d = 0; d < n_docs; d++)
                if (!doc_leaves[d].empty()) out[d].err = err;
        }
    }
    // 3. bitmap rows -> tags -> rules, a document per task.  Rules only ask "was tag T matched at a path with prefix
    // P" (group/dsl/expression.go:70-82): unless the expression strings are wanted (TagJson) a leaf contributes each of
    // its tags once, and tags are ids rather than map keys
    std::unordered_map<std::string, uint32_t> tag_ids;
    std::vector<uint32_t> tag_of(exprs.size());
    for (size_t e = 0; e < exprs.size(); e++) tag_of[e] = tag_ids.emplace(exprs[e].tag, (uint32_t)tag_ids.size()).first->second;
    const uint32_t n_tags = (uint32_t)tag_ids.size();
    for (const auto& kv : rules_) for (const auto& ew : kv.second) resolve_tags(*ew.Expression, tag_ids);
    struct Scratch { std::vector<std::vector<uint32_t>> leaves_of_tag; std::vector<uint32_t> touched; std::vector<uint8_t> seen; };
    std::vector<Scratch> scratch(host_threads());
    parallel_for(n_docs, [&](uint64_t d, unsigned t) {
        if (!out[d].err.empty()) return;
        const auto& lv = doc_leaves[d];
        if (want_tags) {
            for (size_t k = 0; k < lv.size(); k++) {
                const uint32_t* row = bitmap.data() + (first[d] + k) * words;
                for (uint32_t w = 0; w < words; w++)
                    for (uint32_t bits = row[w]; bits; bits &= bits - 1) {
                        const uint32_t e = w * 32 + (uint32_t)__builtin_ctz(bits);
                        out[d].tags[exprs[e].tag][lv[k].path].insert(exprs[e].exprString);
                    }
            }
            return;
        }
        Scratch& sc = scratch[t];
        if (sc.leaves_of_tag.size() != n_tags) { sc.leaves_of_tag.assign(n_tags, {}); sc.seen.assign(n_tags, 0); }
        for (uint32_t tg : sc.touched) sc.leaves_of_tag[tg].clear();
        sc.touched.clear();
        for (size_t k = 0; k < lv.size(); k++) {
            const uint32_t* row = bitmap.data() + (first[d] + k) * words;
            std::fill(sc.seen.begin(), sc.seen.end(), 0);
            for (uint32_t w = 0; w < words; w++)
                for (uint32_t bits = row[w]; bits; bits &= bits - 1) {
                    const uint32_t tg = tag_of[w * 32 + (uint32_t)__builtin_ctz(bits)];
                    if (sc.seen[tg]) continue;
                    sc.seen[tg] = 1;
                    if (sc.leaves_of_tag[tg].empty()) sc.touched.push_back(tg);
                    sc.leaves_of_tag[tg].push_back((uint32_t)k);
                }
        }
        auto unit = [&](const gdsl::Expression& u) {
            if (u.tag_id < 0) return false;
            const auto& where = sc.leaves_of_tag[u.tag_id];
            if (where.empty()) return false;
            if (u.Tag.FieldPath.empty()) return true;
            for (uint32_t k : where)
                if (lv[k].path.compare(0, u.Tag.FieldPath.size(), u.Tag.FieldPath) == 0) return true;
            return false;
        };
        for (const auto& kv : rules_)
            for (const auto& ew : kv.second) {
                std::string err;
                const bool v = gdsl::SolveWith(*ew.Expression, unit, err);
                if (!err.empty()) { out[d].err = err; out[d].rules.clear(); return; }
                if (v) out[d].rules[kv.first].push_back(ew.ExpressionString);
            }
    });
    return "";
}

Error GroupFinder::EvaluateRules(const gdsl::TagMap& m, RuleResult& out) const {    // finder.go:118-137
    out.clear();
    for (const auto& kv : rules_)
        for (const auto& ew : kv.second) {
            std::string err;
            const bool v = gdsl::Solve(*ew.Expression, m, err);
            if (!err.empty()) { out.clear(); return err; }
            if (v) out[kv.first].push_back(ew.ExpressionString);
        }
    return "";
}

}  // namespace gft

// ---- C ABI (include/gft.h) --------------------------------------------------------------------------------------
using namespace gft;

struct gft_group {
    std::unique_ptr<GroupFinder> g;
    std::string err;
    std::string result;      // the last gft_group_process_jsons document (gft_group_last_result)
    mutable std::recursive_mutex mu;   // one caller at a time per handle
};
#define GFT_GLOCK(g) std::lock_guard<std::recursive_mutex> _gft_glock((g)->mu)

// finder_host.cpp
Finder* gft_finder_impl(gft_finder* f);

namespace {

int put(const std::string& s, char* out, uint64_t cap, uint64_t* needed) {
    if (needed) *needed = s.size() + 1;
    if (!out || cap < s.size() + 1) return GFT_E_INVALID;
    memcpy(out, s.c_str(), s.size() + 1);
    return GFT_OK;
}

void str_array(const std::vector<std::string>& v, std::string& o) {
    o += "[";
    for (size_t i = 0; i < v.size(); i++) { if (i) o += ","; dsl::json_str(v[i], o); }
    o += "]";
}

void tagmap_json(const gdsl::TagMap& m, std::string& o) {
    o += "{";
    bool f1 = true;
    for (const auto& t : m) {
        if (!f1) o += ",";
        f1 = false;
        dsl::json_str(t.first, o);
        o += ":{";
        bool f2 = true;
        for (const auto& fp : t.second) {
            if (!f2) o += ",";
            f2 = false;
            dsl::json_str(fp.first, o);
            o += ":";
            str_array({fp.second.begin(), fp.second.end()}, o);
        }
        o += "}";
    }
    o += "}";
}

void rules_json(const GroupFinder::RuleResult& r, std::string& o) {
    o += "{";
    bool first = true;
    for (const auto& kv : r) {
        if (!first) o += ",";
        first = false;
        dsl::json_str(kv.first, o);
        o += ":";
        str_array(kv.second, o);
    }
    o += "}";
}

bool string_list(const uint8_t* p, uint64_t n, std::vector<std::string>& out, std::string& err) {
    out.clear();
    if (!p || !n) return true;
    json::Value v;
    err = json::Parse((const char*)p, n, v);
    if (!err.empty()) return false;
    if (v.kind == json::Value::Null) return true;
    if (v.kind != json::Value::Array) { err = "expected a JSON array of strings"; return false; }
    for (const auto& x : v.arr) {
        if (x.kind != json::Value::String) { err = "expected a JSON array of strings"; return false; }
        out.push_back(x.str);
    }
    return true;
}

bool tagmap_from_json(const json::Value& v, gdsl::TagMap& m, std::string& err) {
    if (v.kind != json::Value::Object) { err = "expected {tag: {field: [expressions]}}"; return false; }
    for (const auto& t : v.obj) {
        auto& fields = m[t.first];
        if (t.second.kind == json::Value::Null) continue;
        if (t.second.kind != json::Value::Object) { err = "expected {tag: {field: [expressions]}}"; return false; }
        for (const auto& fp : t.second.obj) {
            auto& set = fields[fp.first];
            if (fp.second.kind == json::Value::Array)
                for (const auto& x : fp.second.arr) if (x.kind == json::Value::String) set.insert(x.str);
        }
    }
    return true;
}

}  // namespace

extern "C" {

int gft_group_create(gft_group** out, gft_finder* finder) try {
    if (!out || !finder) return GFT_E_INVALID;
    gft_group* g = new gft_group();
    g->g.reset(new GroupFinder(gft_finder_impl(finder)));
    *out = g;
    return GFT_OK;
} GFT_CATCH(nullptr)
void gft_group_destroy(gft_group* g) { delete g; }
const char* gft_group_last_error(const gft_group* g) { return g ? g->err.c_str() : "null group finder"; }

int gft_group_add_rule(gft_group* g, const uint8_t* name, uint64_t name_len, const uint8_t* expr, uint64_t expr_len) try {
    if (!g) return GFT_E_INVALID;
    GFT_GLOCK(g);
    g->err = g->g->AddRule(std::string((const char*)name, name_len), {std::string((const char*)expr, expr_len)});
    return g->err.empty() ? GFT_OK : GFT_E_PARSE;
} GFT_CATCH((g ? &const_cast<gft_group*>(g)->err : nullptr))

int gft_group_state(const gft_group* g, char* out, uint64_t cap, uint64_t* needed) try {
    if (!g) return GFT_E_INVALID;
    GFT_GLOCK(g);
    std::string o = "{\"rules\":{";
    bool first = true;
    for (const auto& kv : g->g->rules()) {
        if (!first) o += ",";
        first = false;
        dsl::json_str(kv.first, o);
        o += ":[";
        for (size_t i = 0; i < kv.second.size(); i++) {
            if (i) o += ",";
            o += "{\"ExpressionString\":";
            dsl::json_str(kv.second[i].ExpressionString, o);
            o += ",\"Expression\":" + gdsl::ToJson(*kv.second[i].Expression) + "}";
        }
        o += "]";
    }
    o += "},\"fields\":";
    str_array({g->g->fields().begin(), g->g->fields().end()}, o);
    o += ",\"tags\":";
    str_array({g->g->tags().begin(), g->g->tags().end()}, o);
    o += "}";
    return put(o, out, cap, needed);
} GFT_CATCH((g ? &const_cast<gft_group*>(g)->err : nullptr))

int gft_group_process_jsons(gft_group* g, const uint8_t* json_blob, const uint64_t* doc_off, uint64_t n_docs,
                            const uint8_t* include_json, uint64_t include_len, const uint8_t* exclude_json,
                            uint64_t exclude_len, int what, char* out, uint64_t cap, uint64_t* needed) try {
    if (!g || (n_docs && (!json_blob || !doc_off))) return GFT_E_INVALID;
    GFT_GLOCK(g);
    std::vector<std::string> inc, exc;
    if (!string_list(include_json, include_len, inc, g->err) || !string_list(exclude_json, exclude_len, exc, g->err))
        return GFT_E_INVALID;
    std::vector<GroupFinder::DocResult> res;
    g->err = g->g->ProcessJsons(json_blob, doc_off, n_docs, inc, exc, what != 0, res);
    if (!g->err.empty()) return GFT_E_ENGINE;
    std::vector<std::string> parts(res.size());
    parallel_for(res.size(), [&](uint64_t d, unsigned) {
        std::string& o = parts[d];
        if (!res[d].err.empty()) { o = "{\"error\":"; dsl::json_str(res[d].err, o); o += "}"; return; }
        o = what == 0 ? "{\"rules\":" : "{\"tags\":";
        if (what == 0) rules_json(res[d].rules, o); else tagmap_json(res[d].tags, o);
        o += "}";
    });
    size_t total = 2;
    for (const auto& p : parts) total += p.size() + 1;
    std::string& o = g->result;
    o.clear();
    o.reserve(total);
    o = "[";
    for (size_t d = 0; d < parts.size(); d++) { if (d) o += ","; o += parts[d]; }
    o += "]";
    return put(o, out, cap, needed);
} GFT_CATCH((g ? &const_cast<gft_group*>(g)->err : nullptr))

int gft_group_last_result(const gft_group* g, char* out, uint64_t cap, uint64_t* needed) try {
    if (!g) return GFT_E_INVALID;
    GFT_GLOCK(g);
    return put(g->result, out, cap, needed);
} GFT_CATCH((g ? &const_cast<gft_group*>(g)->err : nullptr))

int gft_group_evaluate(gft_group* g, const uint8_t* tagmap, uint64_t len, char* out, uint64_t cap, uint64_t* needed) try {
    if (!g || !tagmap) return GFT_E_INVALID;
    GFT_GLOCK(g);
    json::Value v;
    g->err = json::Parse((const char*)tagmap, len, v);
    gdsl::TagMap m;
    if (!g->err.empty() || !tagmap_from_json(v, m, g->err)) return GFT_E_INVALID;
    GroupFinder::RuleResult rr;
    g->err = g->g->EvaluateRules(m, rr);
    if (!g->err.empty()) return GFT_E_ENGINE;
    std::string o;
    rules_json(rr, o);
    return put(o, out, cap, needed);
} GFT_CATCH((g ? &const_cast<gft_group*>(g)->err : nullptr))

int gft_group_last_batch(const gft_group* g, uint64_t* leaves, uint64_t* bytes) try {
    if (!g) return GFT_E_INVALID;
    GFT_GLOCK(g);
    if (leaves) *leaves = g->g->last_leaves;
    if (bytes) *bytes = g->g->last_bytes;
    return GFT_OK;
} GFT_CATCH((g ? &const_cast<gft_group*>(g)->err : nullptr))

int gft_group_dsl_parse(const uint8_t* expr, uint64_t len, char* out, uint64_t cap, uint64_t* needed) try {
    gdsl::ParseResult pr = gdsl::Parse(std::string((const char*)expr, len));
    std::string o;
    if (!pr.err.empty()) { o = "{\"error\":"; dsl::json_str(pr.err, o); o += "}"; }
    else {
        o = "{\"tree\":" + gdsl::ToJson(*pr.expr) + ",\"tags\":";
        str_array(pr.tags, o);
        o += ",\"fields\":";
        str_array(pr.fields, o);
        o += "}";
    }
    return put(o, out, cap, needed);
} GFT_CATCH(nullptr)

int gft_group_dsl_tokens(const uint8_t* expr, uint64_t len, char* out, uint64_t cap, uint64_t* needed) try {
    const std::string src((const char*)expr, len);
    gdsl::Scanner sc(src);
    std::string o = "[";
    for (int i = 0;; i++) {
        gdsl::ScanResult r = sc.Scan();
        if (i) o += ",";
        o += "{\"Tok\":\"";
        o += gdsl::token_name(r.tok);
        o += "\",\"Lit\":";
        dsl::json_str(r.lit, o);
        o += ",\"Err\":";
        if (r.err.empty()) o += "null"; else dsl::json_str(r.err, o);
        o += "}";
        if (!r.err.empty() || r.tok == gdsl::END_OF_INPUT) break;
    }
    o += "]";
    return put(o, out, cap, needed);
} GFT_CATCH(nullptr)

}  // extern "C"
