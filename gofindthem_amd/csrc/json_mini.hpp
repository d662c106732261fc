// A small JSON reader for the host side of the group finder (group/finder/finder.go:80-92 hands raw JSON to
// encoding/json).  Accepts what encoding/json accepts for `interface{}` targets; string values are decoded the way Go
// does it (escapes, surrogate pairs, U+FFFD for lone surrogates and invalid UTF-8).  Error texts follow encoding/json's
// SyntaxError messages for the common cases; the reference has no test that pins them.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace gft {
namespace json {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string str;                                   // String: decoded text; Number: its literal
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;    // in document order; duplicate keys are kept (last one wins
                                                       // for a Go map: see last_wins())
    Value() = default;
    Value(Value&&) = default;
    Value& operator=(Value&&) = default;
    Value(const Value&) = default;
    Value& operator=(const Value&) = default;
    // documents nest up to 10 000 levels (encoding/json's limit): the teardown, like the parser and the walk over a
    // value, must not recurse once per level -- a caller's thread may have a small stack
    ~Value();
    // index of the member that a Go map would hold for key obj[i].first (the last duplicate)
    bool last_wins(size_t i) const {
        for (size_t j = i + 1; j < obj.size(); j++)
            if (obj[j].first == obj[i].first) return false;
        return true;
    }
};

// "" on success, else the error text
std::string Parse(const char* p, size_t n, Value& out);

// append s as a JSON string literal
void Quote(const std::string& s, std::string& out);

}  // namespace json
}  // namespace gft
