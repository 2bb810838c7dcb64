// codec.cpp -- memcmp keys and rmp-serde values of stored rows (see codec.hpp)
#include "cozo_host/codec.hpp"

#include <algorithm>
#include <cstring>
#include <map>

namespace cozo {

namespace {
enum : uint8_t {
    INIT_TAG = 0x00, NULL_TAG = 0x01, FALSE_TAG = 0x02, TRUE_TAG = 0x03, VEC_TAG = 0x04, NUM_TAG = 0x05, STR_TAG = 0x06,
    BYTES_TAG = 0x07, LIST_TAG = 0x0A
};
enum : uint8_t { VEC_F32 = 0x01 };
enum : uint8_t { IS_FLOAT = 0x10, IS_APPROX_INT = 0x04, IS_EXACT_INT = 0x00 };
constexpr uint64_t SIGN_MARK = 0x8000000000000000ull;
constexpr int64_t EXACT_INT_BOUND = 0x20000000000000ll;

void put_be64(std::vector<uint8_t> &o, uint64_t v) {
    for (int i = 7; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i)));
}
uint64_t get_be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = v << 8 | p[i];
    return v;
}
uint64_t order_encode_f64(double f) {  // memcmp.rs:204-211
    uint64_t u;
    std::memcpy(&u, &f, 8);
    return (u >> 63) ? ~u : (u | SIGN_MARK);
}
double order_decode_f64(uint64_t u) {
    u = (u & SIGN_MARK) ? (u & ~SIGN_MARK) : ~u;
    double f;
    std::memcpy(&f, &u, 8);
    return f;
}
void need(const uint8_t *p, const uint8_t *end, size_t n) {
    if ((size_t)(end - p) < n) throw CodecError("truncated stored bytes");
}
}  // namespace

void encode_bytes(std::vector<uint8_t> &out, const uint8_t *key, size_t len) {
    size_t index = 0;
    while (index <= len) {
        const size_t remain = len - index;
        if (remain > 8) {
            out.insert(out.end(), key + index, key + index + 8);
            out.push_back(0xFF);
        } else {
            const size_t pad = 8 - remain;
            out.insert(out.end(), key + index, key + len);
            out.insert(out.end(), pad, 0);
            out.push_back((uint8_t)(0xFF - pad));
        }
        index += 8;
    }
}

std::vector<uint8_t> decode_bytes(const uint8_t *&p, const uint8_t *end) {
    std::vector<uint8_t> key;
    for (;;) {
        need(p, end, 9);
        const uint8_t marker = p[8];
        const size_t pad = 0xFF - marker;
        if (pad > 8) throw CodecError("bad byte-string group marker");
        key.insert(key.end(), p, p + 8 - pad);
        p += 9;
        if (pad) return key;
    }
}

void encode_datavalue(std::vector<uint8_t> &out, const DataValue &v) {  // memcmp.rs:47-145
    switch (v.r.index()) {
        case 0: out.push_back(NULL_TAG); return;
        case 1: out.push_back(std::get<bool>(v.r) ? TRUE_TAG : FALSE_TAG); return;
        case 2: {
            const int64_t i = std::get<int64_t>(v.r);
            out.push_back(NUM_TAG);
            put_be64(out, order_encode_f64((double)i));
            if (i > -EXACT_INT_BOUND && i < EXACT_INT_BOUND) {
                out.push_back(IS_EXACT_INT);
            } else {
                out.push_back(IS_APPROX_INT);
                put_be64(out, (uint64_t)i ^ SIGN_MARK);
            }
            return;
        }
        case 3:
            out.push_back(NUM_TAG);
            put_be64(out, order_encode_f64(std::get<double>(v.r)));
            out.push_back(IS_FLOAT);
            return;
        case 4: {
            const std::string &s = std::get<std::string>(v.r);
            out.push_back(STR_TAG);
            encode_bytes(out, (const uint8_t *)s.data(), s.size());
            return;
        }
        case 5: {
            const auto &b = std::get<Bytes>(v.r).b;
            out.push_back(BYTES_TAG);
            encode_bytes(out, b.data(), b.size());
            return;
        }
        case 6:
            out.push_back(LIST_TAG);
            for (const DataValue &el : std::get<List>(v.r).items) encode_datavalue(out, el);
            out.push_back(INIT_TAG);
            return;
        default: {
            const auto &a = std::get<F32Vec>(v.r).v;
            out.push_back(VEC_TAG);
            out.push_back(VEC_F32);
            put_be64(out, a.size());
            for (float f : a) {
                uint32_t u;
                std::memcpy(&u, &f, 4);
                for (int i = 3; i >= 0; i--) out.push_back((uint8_t)(u >> (8 * i)));
            }
        }
    }
}

DataValue decode_datavalue(const uint8_t *&p, const uint8_t *end) {  // memcmp.rs:258-365
    need(p, end, 1);
    const uint8_t tag = *p++;
    switch (tag) {
        case NULL_TAG: return DataValue();
        case FALSE_TAG: return DataValue(false);
        case TRUE_TAG: return DataValue(true);
        case NUM_TAG: {
            need(p, end, 9);
            const double f = order_decode_f64(get_be64(p));
            const uint8_t kind = p[8];
            p += 9;
            if (kind == IS_FLOAT) return DataValue(f);
            if (kind == IS_EXACT_INT) return DataValue((int64_t)f);
            if (kind != IS_APPROX_INT) throw CodecError("bad number kind in a key");
            need(p, end, 8);
            const int64_t i = (int64_t)(get_be64(p) ^ SIGN_MARK);
            p += 8;
            return DataValue(i);
        }
        case STR_TAG: {
            const std::vector<uint8_t> b = decode_bytes(p, end);
            return DataValue(std::string(b.begin(), b.end()));
        }
        case BYTES_TAG: return DataValue(Bytes{decode_bytes(p, end)});
        case LIST_TAG: {
            std::vector<DataValue> items;
            for (;;) {
                need(p, end, 1);
                if (*p == INIT_TAG) {
                    p++;
                    return DataValue::list(std::move(items));
                }
                items.push_back(decode_datavalue(p, end));
            }
        }
        case VEC_TAG: {
            need(p, end, 9);
            if (p[0] != VEC_F32) throw CodecError("only F32 vectors are modelled");
            const uint64_t n = get_be64(p + 1);
            p += 9;
            if (n > (uint64_t)(end - p) / 4) throw CodecError("truncated vector in a key");
            F32Vec v;
            v.v.resize(n);
            for (uint64_t i = 0; i < n; i++) {
                const uint32_t u = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
                std::memcpy(&v.v[i], &u, 4);
                p += 4;
            }
            return DataValue(std::move(v));
        }
        default: throw CodecError("key tag " + std::to_string(tag) + " is not a modelled variant");
    }
}

// ---- msgpack (rmp-serde 1.2.0 shape of the derived enums) -----------------------------------------------------
namespace {
void mp_str(std::vector<uint8_t> &o, const char *s, size_t n) {
    if (n < 32) o.push_back((uint8_t)(0xa0 | n));
    else if (n < 256) { o.push_back(0xd9); o.push_back((uint8_t)n); }
    else if (n < 65536) { o.push_back(0xda); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n); }
    else { o.push_back(0xdb); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(n >> (8 * i))); }
    o.insert(o.end(), s, s + n);
}
void mp_variant(std::vector<uint8_t> &o, const char *name) {
    o.push_back(0x81);
    mp_str(o, name, std::strlen(name));
}
void mp_bin(std::vector<uint8_t> &o, const uint8_t *p, size_t n) {
    if (n < 256) { o.push_back(0xc4); o.push_back((uint8_t)n); }
    else if (n < 65536) { o.push_back(0xc5); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n); }
    else { o.push_back(0xc6); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(n >> (8 * i))); }
    o.insert(o.end(), p, p + n);
}
void mp_array(std::vector<uint8_t> &o, size_t n) {
    if (n < 16) o.push_back((uint8_t)(0x90 | n));
    else if (n < 65536) { o.push_back(0xdc); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n); }
    else { o.push_back(0xdd); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(n >> (8 * i))); }
}
void mp_int(std::vector<uint8_t> &o, int64_t v) {  // the most compact form (rmp::encode::write_sint)
    auto be = [&](uint8_t tag, int bytes) {
        o.push_back(tag);
        for (int i = bytes - 1; i >= 0; i--) o.push_back((uint8_t)((uint64_t)v >> (8 * i)));
    };
    if (v >= 0) {
        if (v < 128) o.push_back((uint8_t)v);
        else if (v < 256) be(0xcc, 1);
        else if (v < 65536) be(0xcd, 2);
        else if (v < 4294967296ll) be(0xce, 4);
        else be(0xcf, 8);
    } else {
        if (v >= -32) o.push_back((uint8_t)v);
        else if (v >= -128) be(0xd0, 1);
        else if (v >= -32768) be(0xd1, 2);
        else if (v >= -2147483648ll) be(0xd2, 4);
        else be(0xd3, 8);
    }
}
void mp_value(std::vector<uint8_t> &o, const DataValue &v) {
    switch (v.r.index()) {
        case 0: mp_str(o, "Null", 4); return;
        case 1: mp_variant(o, "Bool"); o.push_back(std::get<bool>(v.r) ? 0xc3 : 0xc2); return;
        case 2: mp_variant(o, "Num"); mp_variant(o, "Int"); mp_int(o, std::get<int64_t>(v.r)); return;
        case 3: {
            mp_variant(o, "Num");
            mp_variant(o, "Float");
            uint64_t u;
            const double f = std::get<double>(v.r);
            std::memcpy(&u, &f, 8);
            o.push_back(0xcb);
            put_be64(o, u);
            return;
        }
        case 4: {
            const std::string &s = std::get<std::string>(v.r);
            mp_variant(o, "Str");
            mp_str(o, s.data(), s.size());
            return;
        }
        case 5: {
            const auto &b = std::get<Bytes>(v.r).b;
            mp_variant(o, "Bytes");
            mp_bin(o, b.data(), b.size());
            return;
        }
        case 6: {
            const auto &items = std::get<List>(v.r).items;
            mp_variant(o, "List");
            mp_array(o, items.size());
            for (const DataValue &el : items) mp_value(o, el);
            return;
        }
        default: {  // Vector: (0u8, bytes of the f32s in native order), data/value.rs:226-240
            const auto &a = std::get<F32Vec>(v.r).v;
            mp_variant(o, "Vec");
            mp_array(o, 2);
            o.push_back(0x00);
            mp_bin(o, (const uint8_t *)a.data(), a.size() * 4);
        }
    }
}

struct MpReader {
    const uint8_t *p, *end;
    uint8_t peek() {
        need(p, end, 1);
        return *p;
    }
    uint32_t be(int bytes) {
        need(p, end, (size_t)bytes);
        uint32_t v = 0;
        for (int i = 0; i < bytes; i++) v = v << 8 | *p++;
        return v;
    }
    uint32_t array() {
        const uint8_t t = peek();
        p++;
        if (t >= 0x90 && t <= 0x9f) return t & 0x0f;
        if (t == 0xdc) return be(2);
        if (t == 0xdd) return be(4);
        throw CodecError("msgpack: expected an array");
    }
    std::string str() {
        const uint8_t t = peek();
        p++;
        uint32_t n;
        if (t >= 0xa0 && t <= 0xbf) n = t & 0x1f;
        else if (t == 0xd9) n = be(1);
        else if (t == 0xda) n = be(2);
        else if (t == 0xdb) n = be(4);
        else throw CodecError("msgpack: expected a string");
        need(p, end, n);
        std::string s((const char *)p, n);
        p += n;
        return s;
    }
    std::vector<uint8_t> bin() {
        const uint8_t t = peek();
        p++;
        uint32_t n;
        if (t == 0xc4) n = be(1);
        else if (t == 0xc5) n = be(2);
        else if (t == 0xc6) n = be(4);
        else throw CodecError("msgpack: expected bin");
        need(p, end, n);
        std::vector<uint8_t> b(p, p + n);
        p += n;
        return b;
    }
    int64_t integer() {
        const uint8_t t = peek();
        p++;
        if (t <= 0x7f) return t;
        if (t >= 0xe0) return (int8_t)t;
        switch (t) {
            case 0xcc: return be(1);
            case 0xcd: return be(2);
            case 0xce: return be(4);
            case 0xcf: { const uint64_t hi = be(4); return (int64_t)(hi << 32 | be(4)); }
            case 0xd0: return (int8_t)be(1);
            case 0xd1: return (int16_t)be(2);
            case 0xd2: return (int32_t)be(4);
            case 0xd3: { const uint64_t hi = be(4); return (int64_t)(hi << 32 | be(4)); }
        }
        throw CodecError("msgpack: expected an integer");
    }
    std::string variant() {  // the key of a one-entry map
        if (peek() != 0x81) throw CodecError("msgpack: expected a one-entry map");
        p++;
        return str();
    }
    DataValue value() {
        const uint8_t t = peek();
        if ((t >= 0xa0 && t <= 0xbf) || t == 0xd9) {
            const std::string unit = str();
            if (unit == "Null") return DataValue();
            throw CodecError("msgpack: unit variant " + unit + " is not modelled");
        }
        const std::string name = variant();
        if (name == "Bool") {
            const uint8_t b = peek();
            p++;
            if (b != 0xc2 && b != 0xc3) throw CodecError("msgpack: expected a bool");
            return DataValue(b == 0xc3);
        }
        if (name == "Num") {
            const std::string kind = variant();
            if (kind == "Int") return DataValue(integer());
            if (kind != "Float" || peek() != 0xcb) throw CodecError("msgpack: bad number");
            p++;
            need(p, end, 8);
            const uint64_t u = get_be64(p);
            p += 8;
            double f;
            std::memcpy(&f, &u, 8);
            return DataValue(f);
        }
        if (name == "Str") return DataValue(str());
        if (name == "Bytes") return DataValue(Bytes{bin()});
        if (name == "List") {
            const uint32_t n = array();
            std::vector<DataValue> items;
            for (uint32_t i = 0; i < n; i++) items.push_back(value());
            return DataValue::list(std::move(items));
        }
        if (name == "Vec") {
            if (array() != 2 || integer() != 0) throw CodecError("msgpack: only F32 vectors are modelled");
            const std::vector<uint8_t> b = bin();
            F32Vec v;
            v.v.resize(b.size() / 4);
            if (!v.v.empty()) std::memcpy(v.v.data(), b.data(), v.v.size() * 4);
            return DataValue(std::move(v));
        }
        throw CodecError("msgpack: variant " + name + " is not modelled");
    }
};
}  // namespace

std::vector<uint8_t> encode_key_for_store(uint64_t relation_id, const Tuple &t, size_t n_key_cols) {
    std::vector<uint8_t> out;
    put_be64(out, relation_id);
    for (size_t i = 0; i < n_key_cols && i < t.size(); i++) encode_datavalue(out, t[i]);
    return out;
}

std::vector<uint8_t> encode_val_for_store(uint64_t relation_id, const Tuple &t, size_t n_key_cols) {
    std::vector<uint8_t> out;
    put_be64(out, relation_id);
    const size_t from = std::min(n_key_cols, t.size());
    mp_array(out, t.size() - from);
    for (size_t i = from; i < t.size(); i++) mp_value(out, t[i]);
    return out;
}

Tuple decode_tuple_from_key(const std::vector<uint8_t> &key) {
    if (key.size() < 8) throw CodecError("a stored key is at least the 8-byte relation id");
    const uint8_t *p = key.data() + 8, *end = key.data() + key.size();
    Tuple t;
    while (p < end) t.push_back(decode_datavalue(p, end));
    return t;
}

Tuple decode_tuple_from_kv(const uint8_t *key, size_t key_len, const uint8_t *val, size_t val_len) {
    if (key_len < 8) throw CodecError("a stored key is at least the 8-byte relation id");
    const uint8_t *p = key + 8, *end = key + key_len;
    Tuple t;
    while (p < end) t.push_back(decode_datavalue(p, end));
    if (val_len) {  // extend_tuple_from_v: an empty value adds nothing
        if (val_len < 8) throw CodecError("a stored value starts with the 8-byte prefix");
        MpReader m{val + 8, val + val_len};
        const uint32_t n = m.array();
        for (uint32_t i = 0; i < n; i++) t.push_back(m.value());
    }
    return t;
}

StoredRows StoredRows::from_tuples(uint64_t relation_id, const std::vector<Tuple> &tuples, uint32_t n_key_cols) {
    std::map<std::vector<uint8_t>, std::vector<uint8_t>> kv;  // ordered by key BYTES, like the store
    for (const Tuple &t : tuples) kv[encode_key_for_store(relation_id, t, n_key_cols)] = encode_val_for_store(relation_id, t, n_key_cols);
    StoredRows r;
    r.n_key_cols = n_key_cols;
    for (const auto &e : kv) {
        r.keys.insert(r.keys.end(), e.first.begin(), e.first.end());
        r.key_off.push_back(r.keys.size());
        r.vals.insert(r.vals.end(), e.second.begin(), e.second.end());
        r.val_off.push_back(r.vals.size());
    }
    return r;
}

void stored_rows_delta(const StoredRows &a, const StoredRows &b, StoredRows *puts, std::vector<std::vector<uint8_t>> *dels) {
    *puts = StoredRows();
    puts->n_key_cols = b.n_key_cols;
    dels->clear();
    auto key = [](const StoredRows &r, size_t i) { return std::make_pair(r.keys.data() + r.key_off[i], (size_t)(r.key_off[i + 1] - r.key_off[i])); };
    auto val = [](const StoredRows &r, size_t i) { return std::make_pair(r.vals.data() + r.val_off[i], (size_t)(r.val_off[i + 1] - r.val_off[i])); };
    auto cmp = [](std::pair<const uint8_t *, size_t> x, std::pair<const uint8_t *, size_t> y) {
        const int c = std::memcmp(x.first, y.first, std::min(x.second, y.second));
        return c ? c : (x.second < y.second ? -1 : x.second > y.second ? 1 : 0);
    };
    auto put = [&](size_t j) {
        const auto k = key(b, j), v = val(b, j);
        puts->keys.insert(puts->keys.end(), k.first, k.first + k.second);
        puts->key_off.push_back(puts->keys.size());
        puts->vals.insert(puts->vals.end(), v.first, v.first + v.second);
        puts->val_off.push_back(puts->vals.size());
    };
    size_t i = 0, j = 0;
    const size_t na = a.size(), nb = b.size();
    while (i < na || j < nb) {
        if (j == nb) {
            const auto k = key(a, i++);
            dels->emplace_back(k.first, k.first + k.second);
        } else if (i == na) {
            put(j++);
        } else {
            const int c = cmp(key(a, i), key(b, j));
            if (c == 0) {
                if (cmp(val(a, i), val(b, j)) != 0) put(j);
                i++;
                j++;
            } else if (c < 0) {
                const auto k = key(a, i++);
                dels->emplace_back(k.first, k.first + k.second);
            } else {
                put(j++);
            }
        }
    }
}

}  // namespace cozo
