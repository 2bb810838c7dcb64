// value.hpp -- the subset of cozo-core's DataValue the fixed-rule / HNSW operator surface touches, in C++.
//
// Mirrors cozo-core/src/data/value.rs: `enum DataValue` (:146-174, derive(Ord) over the variants in declaration
// order: Null < Bool < Num < Str < Bytes < ... < List), `enum Num {Int(i64), Float(f64)}` with the mixed
// Int/Float order of `impl Ord for Num` (:575-598: numeric order, an Int sorts before the Float it equals,
// floats by total_cmp), and `Tuple = Vec<DataValue>` (data/tuple.rs).  Variants the hot path never sees as a
// node key (Uuid, Regex, Set, Json, Validity, Bot) are not modelled; `Vec` (the f32 vector of an HNSW row) is.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>
#include <variant>
#include <vector>

namespace cozo {

struct DataValue;
using Tuple = std::vector<DataValue>;

struct Null {};
struct Bytes {
    std::vector<uint8_t> b;
};
struct List {
    std::vector<DataValue> items;
};
// Vector::F32 (data/value.rs:207-213); ordered after List like the enum declares it
struct F32Vec {
    std::vector<float> v;
};

struct DataValue {
    // index order == DataValue's variant order restricted to the modelled variants
    using Repr = std::variant<Null, bool, int64_t, double, std::string, Bytes, List, F32Vec>;
    Repr r;

    DataValue() : r(Null{}) {}
    DataValue(Null) : r(Null{}) {}
    DataValue(bool b) : r(b) {}
    DataValue(int v) : r((int64_t)v) {}
    DataValue(int64_t v) : r(v) {}
    DataValue(uint32_t v) : r((int64_t)v) {}
    DataValue(double v) : r(v) {}
    DataValue(const char *s) : r(std::string(s)) {}
    DataValue(std::string s) : r(std::move(s)) {}
    DataValue(Bytes b) : r(std::move(b)) {}
    DataValue(List l) : r(std::move(l)) {}
    DataValue(F32Vec v) : r(std::move(v)) {}
    static DataValue list(std::vector<DataValue> items) { return DataValue(List{std::move(items)}); }

    bool is_null() const { return std::holds_alternative<Null>(r); }
    bool is_bool() const { return std::holds_alternative<bool>(r); }
    bool is_int() const { return std::holds_alternative<int64_t>(r); }
    bool is_float() const { return std::holds_alternative<double>(r); }
    bool is_num() const { return is_int() || is_float(); }
    bool is_str() const { return std::holds_alternative<std::string>(r); }
    bool is_list() const { return std::holds_alternative<List>(r); }
    bool is_vec() const { return std::holds_alternative<F32Vec>(r); }

    // DataValue::get_bool / get_int / get_float / get_str (data/value.rs): Num only, an integral Float is an int
    bool get_bool(bool *out) const {
        if (!is_bool()) return false;
        *out = std::get<bool>(r);
        return true;
    }
    bool get_int(int64_t *out) const {
        if (is_int()) {
            *out = std::get<int64_t>(r);
            return true;
        }
        if (is_float()) {
            const double f = std::get<double>(r);
            if (std::isfinite(f) && f == std::floor(f)) {
                *out = (int64_t)f;
                return true;
            }
        }
        return false;
    }
    bool get_float(double *out) const {
        if (is_int()) {
            *out = (double)std::get<int64_t>(r);
            return true;
        }
        if (is_float()) {
            *out = std::get<double>(r);
            return true;
        }
        return false;
    }
    const std::string *get_str() const { return std::get_if<std::string>(&r); }
    const std::vector<DataValue> *get_slice() const {
        auto *l = std::get_if<List>(&r);
        return l ? &l->items : nullptr;
    }
    const std::vector<float> *get_vec() const {
        auto *v = std::get_if<F32Vec>(&r);
        return v ? &v->v : nullptr;
    }

    static int rank(const Repr &r) {
        switch (r.index()) {
            case 0: return 0;           // Null
            case 1: return 1;           // Bool
            case 2: case 3: return 2;   // Num
            case 4: return 3;           // Str
            case 5: return 4;           // Bytes
            case 6: return 7;           // List
            default: return 9;          // Vec
        }
    }
    // f64::total_cmp
    static int total_cmp(double a, double b) {
        int64_t x, y;
        std::memcpy(&x, &a, 8);
        std::memcpy(&y, &b, 8);
        x ^= (int64_t)((uint64_t)(x >> 63) >> 1);
        y ^= (int64_t)((uint64_t)(y >> 63) >> 1);
        return x < y ? -1 : (x > y ? 1 : 0);
    }
    static int cmp_num(const DataValue &a, const DataValue &b) {
        if (a.is_int() && b.is_int()) {
            const int64_t x = std::get<int64_t>(a.r), y = std::get<int64_t>(b.r);
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        if (a.is_float() && b.is_float()) return total_cmp(std::get<double>(a.r), std::get<double>(b.r));
        if (a.is_int()) {  // (Int, Float): Equal => Less
            const int c = total_cmp((double)std::get<int64_t>(a.r), std::get<double>(b.r));
            return c == 0 ? -1 : c;
        }
        const int c = total_cmp(std::get<double>(a.r), (double)std::get<int64_t>(b.r));
        return c == 0 ? 1 : c;
    }
    static int compare(const DataValue &a, const DataValue &b) {
        const int ra = rank(a.r), rb = rank(b.r);
        if (ra != rb) return ra < rb ? -1 : 1;
        switch (ra) {
            case 0: return 0;
            case 1: return (int)std::get<bool>(a.r) - (int)std::get<bool>(b.r);
            case 2: return cmp_num(a, b);
            case 3: {
                const int c = std::get<std::string>(a.r).compare(std::get<std::string>(b.r));
                return c < 0 ? -1 : (c > 0 ? 1 : 0);
            }
            case 4: {
                const auto &x = std::get<Bytes>(a.r).b, &y = std::get<Bytes>(b.r).b;
                if (x < y) return -1;
                return y < x ? 1 : 0;
            }
            case 7: {
                const auto &x = std::get<List>(a.r).items, &y = std::get<List>(b.r).items;
                const size_t n = std::min(x.size(), y.size());
                for (size_t i = 0; i < n; i++) {
                    const int c = compare(x[i], y[i]);
                    if (c) return c;
                }
                return x.size() < y.size() ? -1 : (x.size() > y.size() ? 1 : 0);
            }
            default: {
                // impl Ord for Vector (data/value.rs:389-404): length first, then elements as OrderedFloat (numeric
                // order, -0.0 == 0.0, NaN greatest and equal to itself)
                const auto &x = std::get<F32Vec>(a.r).v, &y = std::get<F32Vec>(b.r).v;
                if (x.size() != y.size()) return x.size() < y.size() ? -1 : 1;
                for (size_t i = 0; i < x.size(); i++) {
                    const bool nx = std::isnan(x[i]), ny = std::isnan(y[i]);
                    if (nx || ny) {
                        if (nx != ny) return nx ? 1 : -1;
                        continue;
                    }
                    if (x[i] < y[i]) return -1;
                    if (x[i] > y[i]) return 1;
                }
                return 0;
            }
        }
    }
    friend bool operator<(const DataValue &a, const DataValue &b) { return compare(a, b) < 0; }
    friend bool operator==(const DataValue &a, const DataValue &b) { return compare(a, b) == 0; }
    friend bool operator!=(const DataValue &a, const DataValue &b) { return compare(a, b) != 0; }

    // consistent with operator== (Int 1 and Float 1.0 are different values, like DataValue's derived Hash)
    size_t hash() const {
        auto mix = [](size_t h, size_t v) { return (h ^ (v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2))); };
        size_t h = (size_t)r.index() * 0x100000001b3ull;
        switch (r.index()) {
            case 0: return h;
            case 1: return mix(h, (size_t)std::get<bool>(r));
            case 2: return mix(h, std::hash<int64_t>()(std::get<int64_t>(r)));
            case 3: {
                uint64_t b;
                const double d = std::get<double>(r);
                std::memcpy(&b, &d, 8);
                return mix(h, std::hash<uint64_t>()(b));
            }
            case 4: return mix(h, std::hash<std::string>()(std::get<std::string>(r)));
            case 5: {
                const auto &b = std::get<Bytes>(r).b;
                return mix(h, std::hash<std::string_view>()(std::string_view((const char *)b.data(), b.size())));
            }
            case 6: {
                for (const auto &x : std::get<List>(r).items) h = mix(h, x.hash());
                return h;
            }
            default: {
                // OrderedFloat's Hash (value.rs:424-437): -0.0 hashes like 0.0 and every NaN alike, as they compare equal
                for (float x : std::get<F32Vec>(r).v) {
                    if (x == 0.0f) x = 0.0f;
                    if (std::isnan(x)) x = std::numeric_limits<float>::quiet_NaN();
                    uint32_t b;
                    std::memcpy(&b, &x, 4);
                    h = mix(h, b);
                }
                return h;
            }
        }
    }

    // Display (data/value.rs:606-...): enough for diagnostics
    std::string to_string() const {
        std::ostringstream o;
        print(o);
        return o.str();
    }
    void print(std::ostream &o) const {
        switch (r.index()) {
            case 0: o << "null"; break;
            case 1: o << (std::get<bool>(r) ? "true" : "false"); break;
            case 2: o << std::get<int64_t>(r); break;
            case 3: {
                const double d = std::get<double>(r);
                if (std::isfinite(d) && d == std::floor(d) && std::fabs(d) < 1e15) o << (int64_t)d << ".0";
                else o << d;
                break;
            }
            case 4: o << '"' << std::get<std::string>(r) << '"'; break;
            case 5: o << "bytes(" << std::get<Bytes>(r).b.size() << ")"; break;
            case 6: {
                o << '[';
                bool first = true;
                for (const auto &x : std::get<List>(r).items) {
                    if (!first) o << ", ";
                    first = false;
                    x.print(o);
                }
                o << ']';
                break;
            }
            default: o << "vec(" << std::get<F32Vec>(r).v.size() << ")"; break;
        }
    }
};

struct DataValueHash {
    size_t operator()(const DataValue &v) const { return v.hash(); }
};

inline bool tuple_less(const Tuple &a, const Tuple &b) {
    const size_t n = std::min(a.size(), b.size());
    for (size_t i = 0; i < n; i++) {
        const int c = DataValue::compare(a[i], b[i]);
        if (c) return c < 0;
    }
    return a.size() < b.size();
}
struct TupleLess {
    bool operator()(const Tuple &a, const Tuple &b) const { return tuple_less(a, b); }
};

}  // namespace cozo
