// json_min.hpp -- a small strict JSON reader for the host scene loader.
//
// The reference parses scenes with nlohmann/json (its vendored json.hpp); that header is not
// copied here.  Numbers follow the same rule nlohmann applies: a literal without fraction or
// exponent is an integer, anything else is read with strtod into a double, and a float is
// obtained by converting that double (so `float x = doc["k"]` yields the same bits).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace jsonmin {

struct Value {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    int64_t i = 0;
    double d = 0.0;
    std::string s;
    std::vector<Value> a;
    std::vector<std::pair<std::string, Value>> o;

    bool contains(const char *key) const {
        if (kind != Object) return false;
        for (const auto &kv : o)
            if (kv.first == key) return true;
        return false;
    }
    const Value &at(const char *key) const {
        if (kind == Object)
            for (const auto &kv : o)
                if (kv.first == key) return kv.second;
        throw std::runtime_error(std::string("missing key \"") + key + "\"");
    }
    const Value &idx(size_t k) const {
        if (kind != Array || k >= a.size()) throw std::runtime_error("array index out of range");
        return a[k];
    }
    size_t size() const { return kind == Array ? a.size() : (kind == Object ? o.size() : 0); }
    bool is_number() const { return kind == Int || kind == Float; }
    double as_double() const {
        if (kind == Int) return (double)i;
        if (kind == Float) return d;
        throw std::runtime_error("expected a number");
    }
    float as_float() const { return (float)as_double(); }
    uint64_t as_index() const {
        if (kind == Int && i >= 0) return (uint64_t)i;
        if (kind == Float && d >= 0) return (uint64_t)d;
        throw std::runtime_error("expected a non-negative integer");
    }
    const std::string &as_string() const {
        if (kind != String) throw std::runtime_error("expected a string");
        return s;
    }
};

class Parser {
  public:
    explicit Parser(const std::string &text) : p_(text.data()), end_(text.data() + text.size()) {}
    Value parse() {
        Value v = value();
        ws();
        if (p_ != end_) fail("trailing characters");
        return v;
    }

  private:
    const char *p_, *end_;
    [[noreturn]] void fail(const char *what) { throw std::runtime_error(std::string("JSON: ") + what); }
    void ws() {
        while (p_ != end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool lit(const char *w) {
        size_t n = std::strlen(w);
        if ((size_t)(end_ - p_) >= n && std::memcmp(p_, w, n) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }
    static void utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }
    unsigned hex4() {
        if (end_ - p_ < 4) fail("bad \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        std::string out;
        ++p_;  // opening quote
        for (;;) {
            if (p_ == end_) fail("unterminated string");
            char c = *p_++;
            if (c == '"') break;
            if (c == '\\') {
                if (p_ == end_) fail("unterminated escape");
                char e = *p_++;
                switch (e) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        unsigned cp = hex4();
                        if (cp >= 0xD800 && cp <= 0xDBFF && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                            p_ += 2;
                            unsigned lo = hex4();
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(out, cp);
                        break;
                    }
                    default: fail("bad escape");
                }
            } else {
                out += c;
            }
        }
        return out;
    }
    Value number() {
        const char *start = p_;
        bool integral = true;
        if (p_ != end_ && *p_ == '-') ++p_;
        while (p_ != end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        if (p_ != end_ && *p_ == '.') {
            integral = false;
            ++p_;
            while (p_ != end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ != end_ && (*p_ == 'e' || *p_ == 'E')) {
            integral = false;
            ++p_;
            if (p_ != end_ && (*p_ == '+' || *p_ == '-')) ++p_;
            while (p_ != end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ == start || (p_ == start + 1 && *start == '-')) fail("bad number");
        std::string tok(start, p_);
        Value v;
        if (integral && tok.size() <= 18) {
            v.kind = Value::Int;
            v.i = std::strtoll(tok.c_str(), nullptr, 10);
            v.d = (double)v.i;
        } else {
            v.kind = Value::Float;
            v.d = std::strtod(tok.c_str(), nullptr);
        }
        return v;
    }
    Value value() {
        ws();
        if (p_ == end_) fail("unexpected end");
        Value v;
        char c = *p_;
        if (c == '{') {
            v.kind = Value::Object;
            ++p_;
            ws();
            if (p_ != end_ && *p_ == '}') {
                ++p_;
                return v;
            }
            for (;;) {
                ws();
                if (p_ == end_ || *p_ != '"') fail("expected a key");
                std::string k = string();
                ws();
                if (p_ == end_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v.o.emplace_back(std::move(k), value());
                ws();
                if (p_ != end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ != end_ && *p_ == '}') {
                    ++p_;
                    break;
                }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = Value::Array;
            ++p_;
            ws();
            if (p_ != end_ && *p_ == ']') {
                ++p_;
                return v;
            }
            for (;;) {
                v.a.push_back(value());
                ws();
                if (p_ != end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ != end_ && *p_ == ']') {
                    ++p_;
                    break;
                }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Value::String;
            v.s = string();
        } else if (lit("true")) {
            v.kind = Value::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = Value::Bool;
        } else if (lit("null")) {
        } else {
            v = number();
        }
        return v;
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace jsonmin
