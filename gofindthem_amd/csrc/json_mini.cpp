#include "json_mini.hpp"

#include <cstdio>

#include "dsl_compile.hpp"

namespace gft {
namespace json {

namespace {

constexpr int kMaxDepth = 10000;       // encoding/json: "exceeded max depth"

struct Reader {
    const char* p;
    size_t n, i = 0;
    std::string err;

    static std::string show(unsigned char c) {          // quoteChar of encoding/json
        if (c == '\'') return "'\\''";
        if (c == '"') return "'\"'";
        char b[16];
        if (c < 0x20 || c == 0x7F) { snprintf(b, sizeof b, "'\\x%02x'", c); return b; }
        // (Go prints the whole rune for multi-byte characters; a single byte is what we have here)
        snprintf(b, sizeof b, "'%c'", c);
        return b;
    }
    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
    bool fail_char(const char* ctx) { return fail("invalid character " + show((unsigned char)p[i]) + " " + ctx); }
    bool eof() { return fail("unexpected end of JSON input"); }
    void ws() { while (i < n && (p[i] == ' ' || p[i] == '\t' || p[i] == '\n' || p[i] == '\r')) i++; }

    static int hex(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    bool u4(int32_t& v) {
        v = 0;
        for (int k = 0; k < 4; k++) {
            if (i >= n) return eof();
            const int h = hex(p[i]);
            if (h < 0) return fail_char("in \\u hexadecimal character escape");
            v = v * 16 + h;
            i++;
        }
        return true;
    }

    bool string(std::string& out) {                    // p[i] == '"'
        i++;
        std::string raw;
        for (;;) {
            if (i >= n) return eof();
            const unsigned char c = (unsigned char)p[i];
            if (c == '"') { i++; break; }
            if (c < 0x20) return fail_char("in string literal");
            if (c != '\\') { raw.push_back((char)c); i++; continue; }
            i++;
            if (i >= n) return eof();
            const char e = p[i];
            switch (e) {
            case '"': raw.push_back('"'); i++; break;
            case '\\': raw.push_back('\\'); i++; break;
            case '/': raw.push_back('/'); i++; break;
            case 'b': raw.push_back('\b'); i++; break;
            case 'f': raw.push_back('\f'); i++; break;
            case 'n': raw.push_back('\n'); i++; break;
            case 'r': raw.push_back('\r'); i++; break;
            case 't': raw.push_back('\t'); i++; break;
            case 'u': {
                i++;
                int32_t cp;
                if (!u4(cp)) return false;
                if (cp >= 0xD800 && cp <= 0xDBFF) {    // high surrogate: needs \uDC00..\uDFFF right behind it
                    int32_t lo = -1;
                    if (i + 1 < n && p[i] == '\\' && p[i + 1] == 'u') {
                        const size_t save = i;
                        i += 2;
                        if (!u4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) { lo = -1; i = save; }
                    }
                    cp = lo >= 0 ? 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00) : 0xFFFD;
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                    cp = 0xFFFD;
                }
                dsl::EncodeRune(cp, raw);
                break;
            }
            default:
                return fail_char("in string escape code");
            }
        }
        // invalid UTF-8 -> U+FFFD per offending byte, as encoding/json's unquote does
        out.clear();
        out.reserve(raw.size());
        bool ascii = true;
        for (unsigned char c : raw) if (c >= 0x80) { ascii = false; break; }
        if (ascii) { out = raw; return true; }
        for (size_t k = 0; k < raw.size();) {
            size_t adv;
            dsl::EncodeRune(dsl::DecodeRune(raw, k, &adv), out);
            k += adv;
        }
        return true;
    }

    bool number(std::string& out) {
        const size_t s = i;
        if (p[i] == '-') { i++; if (i >= n) return eof(); }
        if (p[i] == '0') i++;
        else if (p[i] >= '1' && p[i] <= '9') { while (i < n && p[i] >= '0' && p[i] <= '9') i++; }
        else return fail_char("in numeric literal");
        if (i < n && p[i] == '.') {
            i++;
            if (i >= n) return eof();
            if (p[i] < '0' || p[i] > '9') return fail_char("after decimal point in numeric literal");
            while (i < n && p[i] >= '0' && p[i] <= '9') i++;
        }
        if (i < n && (p[i] == 'e' || p[i] == 'E')) {
            i++;
            if (i < n && (p[i] == '+' || p[i] == '-')) i++;
            if (i >= n) return eof();
            if (p[i] < '0' || p[i] > '9') return fail_char("in exponent of numeric literal");
            while (i < n && p[i] >= '0' && p[i] <= '9') i++;
        }
        out.assign(p + s, i - s);
        return true;
    }

    bool literal(const char* word, const char* ctx) {
        for (size_t k = 0; word[k]; k++) {
            if (i >= n) return eof();
            if (p[i] != word[k]) return fail_char(ctx);
            i++;
        }
        return true;
    }

    // Containers are walked with an explicit stack (no recursion per nesting level: documents come from outside and a
    // caller's thread may have a small stack); depth is capped like encoding/json caps it
    bool scalar(Value& v) {
        const char c = p[i];
        if (c == '"') { v.kind = Value::String; return string(v.str); }
        if (c == 't') { v.kind = Value::Bool; v.b = true; return literal("true", "in literal true (expecting 'r')"); }
        if (c == 'f') { v.kind = Value::Bool; v.b = false; return literal("false", "in literal false (expecting 'a')"); }
        if (c == 'n') { v.kind = Value::Null; return literal("null", "in literal null (expecting 'u')"); }
        if (c == '-' || (c >= '0' && c <= '9')) { v.kind = Value::Number; return number(v.str); }
        return fail_char("looking for beginning of value");
    }

    bool value(Value& root) {
        std::vector<Value*> open;                      // the containers being filled, outermost first
        Value* v = &root;                              // where the next value goes
        for (;;) {
            ws();
            if (i >= n) return eof();
            const char c = p[i];
            bool opened = false;
            if (c == '{' || c == '[') {
                if ((int)open.size() >= kMaxDepth) return fail("exceeded max depth");
                v->kind = c == '{' ? Value::Object : Value::Array;
                i++;
                ws();
                if (i >= n) return eof();
                if (p[i] == (c == '{' ? '}' : ']')) i++;               // empty container: a finished value
                else { open.push_back(v); opened = true; }
            } else if (!scalar(*v)) {
                return false;
            }
            // after a finished value: close containers / step to the next member; after an opening: its first member
            for (;;) {
                if (open.empty()) return true;
                Value* top = open.back();
                if (!opened) {
                    ws();
                    if (i >= n) return eof();
                    if (top->kind == Value::Object) {
                        if (p[i] == '}') { i++; open.pop_back(); continue; }
                        if (p[i] != ',') return fail_char("after object key:value pair");
                    } else {
                        if (p[i] == ']') { i++; open.pop_back(); continue; }
                        if (p[i] != ',') return fail_char("after array element");
                    }
                    i++;
                }
                if (top->kind == Value::Object) {
                    ws();
                    if (i >= n) return eof();
                    if (p[i] != '"') return fail_char("looking for beginning of object key string");
                    std::string key;
                    if (!string(key)) return false;
                    ws();
                    if (i >= n) return eof();
                    if (p[i] != ':') return fail_char("after object key");
                    i++;
                    top->obj.emplace_back(std::move(key), Value());
                    v = &top->obj.back().second;
                } else {
                    top->arr.emplace_back();
                    v = &top->arr.back();
                }
                break;
            }
        }
    }
};

}  // namespace

std::string Parse(const char* p, size_t n, Value& out) {
    Reader r{p, n};
    out = Value();
    if (!r.value(out)) return r.err;
    r.ws();
    if (r.i < n) { r.fail_char("after top-level value"); return r.err; }
    return "";
}

// teardown without recursion: the children of every container are moved onto one work list before their parent dies
Value::~Value() {
    if (arr.empty() && obj.empty()) return;
    std::vector<Value> work;
    auto take = [&](Value& v) {
        for (auto& c : v.arr) if (!c.arr.empty() || !c.obj.empty()) work.push_back(std::move(c));
        for (auto& kv : v.obj) if (!kv.second.arr.empty() || !kv.second.obj.empty()) work.push_back(std::move(kv.second));
        v.arr.clear();
        v.obj.clear();
    };
    take(*this);
    while (!work.empty()) {
        Value v = std::move(work.back());
        work.pop_back();
        take(v);
    }      // (v and the emptied children die here: nothing below them is left)
}

void Quote(const std::string& s, std::string& out) { dsl::json_str(s, out); }

}  // namespace json
}  // namespace gft
