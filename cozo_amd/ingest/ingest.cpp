// ingest.cpp -- libcozo_ingest.so: stored rows (key bytes + value bytes) -> the flat arrays libcozo_gpu.so takes.
// Host code only.  See include/cozo_ingest.h for the contract; the byte formats restated here are
//   memcmp keys      data/memcmp.rs:22-163 (encode), :165-365 (decode)
//   stored key/value data/tuple.rs:27-52, runtime/relation.rs:169-296, 520-531
//   msgpack values   rmp-serde 1.2.0 over the derives of data/value.rs:143-175 (+ Vector's own impl, :226-252)
#include "cozo_ingest.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <array>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

struct Error {
    int code;
};
[[noreturn]] void raise(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    throw Error{code};
}

inline uint64_t be64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}
inline uint32_t be32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return __builtin_bswap32(v);
}
inline uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

// ------------------------------------------------------------------------------------------------ memcmp keys
enum : uint8_t {
    INIT_TAG = 0x00, NULL_TAG = 0x01, FALSE_TAG = 0x02, TRUE_TAG = 0x03, VEC_TAG = 0x04, NUM_TAG = 0x05, STR_TAG = 0x06,
    BYTES_TAG = 0x07, UUID_TAG = 0x08, REGEX_TAG = 0x09, LIST_TAG = 0x0A, SET_TAG = 0x0B, VLD_TAG = 0x0C, JSON_TAG = 0x0D,
    BOT_TAG = 0xFF
};
enum : uint8_t { VEC_F32 = 0x01, VEC_F64 = 0x02 };
enum : uint8_t { IS_FLOAT = 0x10, IS_APPROX_INT = 0x04, IS_EXACT_INT = 0x00 };
constexpr uint64_t SIGN_MARK = 0x8000000000000000ull;
constexpr int64_t EXACT_INT_BOUND = 0x20000000000000ll;
constexpr int kMaxDepth = 64;

// encode_bytes groups (memcmp.rs:147-163): 8 payload bytes + a marker 0xFF - pad; the group with pad > 0 ends the string
const uint8_t *mc_skip_groups(const uint8_t *p, const uint8_t *end) {
    for (;;) {
        if (end - p < 9) raise(CZI_E_CORRUPT, "truncated byte-string group in a key");
        const uint8_t marker = p[8];
        p += 9;
        if (marker == 0xFF) continue;
        if (marker < 0xF7) raise(CZI_E_CORRUPT, "bad byte-string group marker 0x%02x", marker);
        return p;
    }
}

// one encoded DataValue -> pointer past it
const uint8_t *mc_skip(const uint8_t *p, const uint8_t *end, int depth = 0) {
    if (p >= end) raise(CZI_E_CORRUPT, "key ends where a column should start");
    if (depth > kMaxDepth) raise(CZI_E_CORRUPT, "key nests deeper than %d lists", kMaxDepth);
    const uint8_t tag = *p++;
    switch (tag) {
    case NULL_TAG: case FALSE_TAG: case TRUE_TAG: case BOT_TAG:
        return p;
    case NUM_TAG: {
        if (end - p < 9) raise(CZI_E_CORRUPT, "truncated number in a key");
        const uint8_t kind = p[8];
        p += 9;
        if (kind == IS_APPROX_INT) {
            if (end - p < 8) raise(CZI_E_CORRUPT, "truncated number in a key");
            p += 8;
        } else if (kind != IS_FLOAT && kind != IS_EXACT_INT) {
            raise(CZI_E_CORRUPT, "bad number kind 0x%02x in a key", kind);
        }
        return p;
    }
    case STR_TAG: case BYTES_TAG: case REGEX_TAG: case JSON_TAG:
        return mc_skip_groups(p, end);
    case UUID_TAG:
        if (end - p < 16) raise(CZI_E_CORRUPT, "truncated uuid in a key");
        return p + 16;
    case VLD_TAG:
        if (end - p < 9) raise(CZI_E_CORRUPT, "truncated validity in a key");
        return p + 9;
    case VEC_TAG: {
        if (end - p < 9) raise(CZI_E_CORRUPT, "truncated vector in a key");
        const uint8_t t = p[0];
        const uint64_t len = be64(p + 1);
        p += 9;
        const uint64_t w = t == VEC_F32 ? 4 : t == VEC_F64 ? 8 : 0;
        if (!w) raise(CZI_E_CORRUPT, "bad vector element tag 0x%02x in a key", t);
        if (len > (uint64_t)(end - p) / w) raise(CZI_E_CORRUPT, "truncated vector in a key");
        return p + len * w;
    }
    case LIST_TAG: case SET_TAG:
        for (;;) {
            if (p >= end) raise(CZI_E_CORRUPT, "unterminated list in a key");
            if (*p == INIT_TAG) return p + 1;
            p = mc_skip(p, end, depth + 1);
        }
    default:
        raise(CZI_E_CORRUPT, "unknown key tag 0x%02x", tag);
    }
}

inline double order_decode_f64(uint64_t u) {
    u = (u & SIGN_MARK) ? (u & ~SIGN_MARK) : ~u;
    double f;
    memcpy(&f, &u, 8);
    return f;
}
inline uint64_t order_encode_f64(double v) {
    uint64_t u;
    memcpy(&u, &v, 8);
    return (u >> 63) ? ~u : (u | SIGN_MARK);  // is_sign_positive <=> sign bit clear (also for NaN)
}

struct NumVal {
    bool is_int;
    int64_t i;
    double f;  // get_float(): the int widened
};
// Num::decode_from_key (memcmp.rs:227-245); p points past NUM_TAG
NumVal mc_num(const uint8_t *p, const uint8_t *end) {
    if (end - p < 9) raise(CZI_E_CORRUPT, "truncated number in a key");
    const double f = order_decode_f64(be64(p));
    const uint8_t kind = p[8];
    if (kind == IS_FLOAT) return {false, 0, f};
    if (kind == IS_EXACT_INT) return {true, (int64_t)f, f};
    if (kind != IS_APPROX_INT || end - p < 17) raise(CZI_E_CORRUPT, "bad number in a key");
    const int64_t i = (int64_t)(be64(p + 9) ^ SIGN_MARK);
    return {true, i, (double)i};
}

struct Buf {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u64be(uint64_t v) {
        v = __builtin_bswap64(v);
        const uint8_t *p = (const uint8_t *)&v;
        b.insert(b.end(), p, p + 8);
    }
    void raw(const uint8_t *p, size_t n) { b.insert(b.end(), p, p + n); }
    // encode_bytes (memcmp.rs:147-163)
    void groups(const uint8_t *key, size_t len) {
        size_t index = 0;
        while (index <= len) {
            const size_t remain = len - index;
            if (remain > 8) {
                raw(key + index, 8);
                u8(0xFF);
            } else {
                const size_t pad = 8 - remain;
                raw(key + index, remain);
                for (size_t i = 0; i < pad; i++) u8(0);
                u8((uint8_t)(0xFF - pad));
            }
            index += 8;
        }
    }
    void num_int(int64_t i) {  // encode_num (memcmp.rs:127-145)
        u8(NUM_TAG);
        u64be(order_encode_f64((double)i));
        if (i > -EXACT_INT_BOUND && i < EXACT_INT_BOUND) {
            u8(IS_EXACT_INT);
        } else {
            u8(IS_APPROX_INT);
            u64be((uint64_t)i ^ SIGN_MARK);
        }
    }
    void num_float(double f) {
        u8(NUM_TAG);
        u64be(order_encode_f64(f));
        u8(IS_FLOAT);
    }
};

// ------------------------------------------------------------------------------------------------ msgpack values
struct Mp {
    const uint8_t *p, *end;
    void need(size_t n) const {
        if ((size_t)(end - p) < n) raise(CZI_E_CORRUPT, "truncated msgpack value");
    }
    uint8_t peek() const {
        need(1);
        return *p;
    }
    bool is_str() const {
        const uint8_t t = peek();
        return (t >= 0xa0 && t <= 0xbf) || t == 0xd9 || t == 0xda || t == 0xdb;
    }
    bool is_map() const {
        const uint8_t t = peek();
        return (t >= 0x80 && t <= 0x8f) || t == 0xde || t == 0xdf;
    }
    bool is_int() const {
        const uint8_t t = peek();
        return t <= 0x7f || t >= 0xe0 || (t >= 0xcc && t <= 0xd3);
    }
    uint32_t array() {
        const uint8_t t = peek();
        if (t >= 0x90 && t <= 0x9f) { p++; return t & 0x0f; }
        if (t == 0xdc) { need(3); const uint32_t n = be16(p + 1); p += 3; return n; }
        if (t == 0xdd) { need(5); const uint32_t n = be32(p + 1); p += 5; return n; }
        raise(CZI_E_CORRUPT, "msgpack: expected an array, found 0x%02x", t);
    }
    uint32_t map() {
        const uint8_t t = peek();
        if (t >= 0x80 && t <= 0x8f) { p++; return t & 0x0f; }
        if (t == 0xde) { need(3); const uint32_t n = be16(p + 1); p += 3; return n; }
        if (t == 0xdf) { need(5); const uint32_t n = be32(p + 1); p += 5; return n; }
        raise(CZI_E_CORRUPT, "msgpack: expected a map, found 0x%02x", t);
    }
    void str(const uint8_t *&s, uint32_t &n) {
        const uint8_t t = peek();
        if (t >= 0xa0 && t <= 0xbf) { n = t & 0x1f; p++; }
        else if (t == 0xd9) { need(2); n = p[1]; p += 2; }
        else if (t == 0xda) { need(3); n = be16(p + 1); p += 3; }
        else if (t == 0xdb) { need(5); n = be32(p + 1); p += 5; }
        else raise(CZI_E_CORRUPT, "msgpack: expected a string, found 0x%02x", t);
        need(n);
        s = p;
        p += n;
    }
    // serde_bytes writes bin; a Vec<u8> without it would be an array of ints -- accepted too
    void bin(const uint8_t *&s, uint32_t &n, std::vector<uint8_t> &scratch) {
        const uint8_t t = peek();
        if (t == 0xc4) { need(2); n = p[1]; p += 2; }
        else if (t == 0xc5) { need(3); n = be16(p + 1); p += 3; }
        else if (t == 0xc6) { need(5); n = be32(p + 1); p += 5; }
        else if (is_str()) { str(s, n); return; }
        else {
            const uint32_t k = array();
            scratch.resize(k);
            for (uint32_t i = 0; i < k; i++) scratch[i] = (uint8_t)integer();
            s = scratch.data();
            n = k;
            return;
        }
        need(n);
        s = p;
        p += n;
    }
    int64_t integer() {
        const uint8_t t = peek();
        if (t <= 0x7f) { p++; return t; }
        if (t >= 0xe0) { p++; return (int8_t)t; }
        switch (t) {
        case 0xcc: need(2); p += 2; return p[-1];
        case 0xcd: need(3); p += 3; return be16(p - 2);
        case 0xce: need(5); p += 5; return be32(p - 4);
        case 0xcf: need(9); p += 9; return (int64_t)be64(p - 8);
        case 0xd0: need(2); p += 2; return (int8_t)p[-1];
        case 0xd1: need(3); p += 3; return (int16_t)be16(p - 2);
        case 0xd2: need(5); p += 5; return (int32_t)be32(p - 4);
        case 0xd3: need(9); p += 9; return (int64_t)be64(p - 8);
        }
        raise(CZI_E_CORRUPT, "msgpack: expected an integer, found 0x%02x", t);
    }
    double real() {
        const uint8_t t = peek();
        if (t == 0xcb) {
            need(9);
            const uint64_t u = be64(p + 1);
            p += 9;
            double f;
            memcpy(&f, &u, 8);
            return f;
        }
        if (t == 0xca) {
            need(5);
            const uint32_t u = be32(p + 1);
            p += 5;
            float f;
            memcpy(&f, &u, 4);
            return f;
        }
        return (double)integer();
    }
    bool boolean() {
        const uint8_t t = peek();
        if (t != 0xc2 && t != 0xc3) raise(CZI_E_CORRUPT, "msgpack: expected a bool, found 0x%02x", t);
        p++;
        return t == 0xc3;
    }
    void skip(int depth = 0) {
        if (depth > kMaxDepth) raise(CZI_E_CORRUPT, "msgpack value nests too deep");
        const uint8_t t = peek();
        size_t n = 0;
        if (t <= 0x7f || t >= 0xe0 || t == 0xc0 || t == 0xc2 || t == 0xc3) { p++; return; }
        if (t >= 0xa0 && t <= 0xbf) n = 1 + (t & 0x1f);
        else if (t >= 0x90 && t <= 0x9f) { uint32_t k = array(); while (k--) skip(depth + 1); return; }
        else if (t >= 0x80 && t <= 0x8f) { uint32_t k = map(); while (k--) { skip(depth + 1); skip(depth + 1); } return; }
        else switch (t) {
        case 0xc4: case 0xd9: need(2); n = 2 + p[1]; break;
        case 0xc5: case 0xda: need(3); n = 3 + be16(p + 1); break;
        case 0xc6: case 0xdb: need(5); n = 5 + (size_t)be32(p + 1); break;
        case 0xca: case 0xce: case 0xd2: n = 5; break;
        case 0xcb: case 0xcf: case 0xd3: n = 9; break;
        case 0xcc: case 0xd0: n = 2; break;
        case 0xcd: case 0xd1: n = 3; break;
        case 0xd4: n = 3; break;
        case 0xd5: n = 4; break;
        case 0xd6: n = 6; break;
        case 0xd7: n = 10; break;
        case 0xd8: n = 18; break;
        case 0xc7: need(2); n = 3 + p[1]; break;
        case 0xc8: need(3); n = 4 + be16(p + 1); break;
        case 0xc9: need(5); n = 6 + (size_t)be32(p + 1); break;
        case 0xdc: case 0xdd: { uint32_t k = array(); while (k--) skip(depth + 1); return; }
        case 0xde: case 0xdf: { uint32_t k = map(); while (k--) { skip(depth + 1); skip(depth + 1); } return; }
        default: raise(CZI_E_CORRUPT, "msgpack: reserved byte 0x%02x", t);
        }
        need(n);
        p += n;
    }
};

// DataValue's variants in declaration order (data/value.rs:146-175) -- the index form some serde encoders use
enum Variant { V_NULL, V_BOOL, V_NUM, V_STR, V_BYTES, V_UUID, V_REGEX, V_LIST, V_SET, V_VEC, V_JSON, V_VALIDITY, V_BOT, V_COUNT };
const char *const kVariantName[V_COUNT] = {"Null", "Bool", "Num", "Str", "Bytes", "Uuid", "Regex", "List", "Set", "Vec", "Json",
                                           "Validity", "Bot"};

int variant_of(Mp &m, const char *const *names, int count, const char *what) {
    if (m.is_str()) {
        const uint8_t *s;
        uint32_t n;
        m.str(s, n);
        for (int v = 0; v < count; v++)
            if (strlen(names[v]) == n && memcmp(names[v], s, n) == 0) return v;
        raise(CZI_E_CORRUPT, "msgpack: unknown %s variant '%.*s'", what, (int)std::min<uint32_t>(n, 32), (const char *)s);
    }
    if (m.is_int()) {
        const int64_t v = m.integer();
        if (v < 0 || v >= count) raise(CZI_E_CORRUPT, "msgpack: %s variant index %lld out of range", what, (long long)v);
        return (int)v;
    }
    raise(CZI_E_CORRUPT, "msgpack: expected a %s variant, found 0x%02x", what, m.peek());
}

// reads the head of one DataValue: unit variants are bare, the others a one-entry map whose value follows
Variant mp_value_head(Mp &m) {
    if (m.is_map()) {
        if (m.map() != 1) raise(CZI_E_CORRUPT, "msgpack: a value must be a one-entry map");
        return (Variant)variant_of(m, kVariantName, V_COUNT, "DataValue");
    }
    const Variant v = (Variant)variant_of(m, kVariantName, V_COUNT, "DataValue");
    if (v != V_NULL && v != V_BOT) raise(CZI_E_CORRUPT, "msgpack: variant %s needs a payload", kVariantName[v]);
    return v;
}

NumVal mp_num(Mp &m) {  // enum Num { Int(i64), Float(f64) }, data/value.rs:493-499
    static const char *const names[2] = {"Int", "Float"};
    if (m.map() != 1) raise(CZI_E_CORRUPT, "msgpack: a number must be a one-entry map");
    if (variant_of(m, names, 2, "Num") == 0) {
        const int64_t i = m.integer();
        return {true, i, (double)i};
    }
    return {false, 0, m.real()};
}

// Vector (data/value.rs:226-252): tuple (0u8 | 1u8, bytes of the elements in NATIVE = little-endian order)
void mp_vec(Mp &m, int &el, const uint8_t *&bytes, uint32_t &n, std::vector<uint8_t> &scratch) {
    if (m.array() != 2) raise(CZI_E_CORRUPT, "msgpack: a vector must be a 2-tuple");
    el = (int)m.integer();
    if (el != 0 && el != 1) raise(CZI_E_CORRUPT, "msgpack: bad vector element type %d", el);
    m.bin(bytes, n, scratch);
    if (n % (el == 0 ? 4 : 8)) raise(CZI_E_CORRUPT, "msgpack: vector payload of %u bytes", n);
}

// one msgpack DataValue -> its memcmp encoding appended to `out` (so that a node value stored in the value part of a
// row gets the same identity as the same value stored in a key column)
void mp_to_memcmp(Mp &m, Buf &out, std::vector<uint8_t> &scratch, int depth = 0) {
    if (depth > kMaxDepth) raise(CZI_E_CORRUPT, "msgpack value nests too deep");
    const Variant v = mp_value_head(m);
    const uint8_t *s;
    uint32_t n;
    switch (v) {
    case V_NULL: out.u8(NULL_TAG); return;
    case V_BOT: out.u8(BOT_TAG); return;
    case V_BOOL: out.u8(m.boolean() ? TRUE_TAG : FALSE_TAG); return;
    case V_NUM: {
        const NumVal x = mp_num(m);
        if (x.is_int) out.num_int(x.i); else out.num_float(x.f);
        return;
    }
    case V_STR: m.str(s, n); out.u8(STR_TAG); out.groups(s, n); return;
    case V_BYTES: m.bin(s, n, scratch); out.u8(BYTES_TAG); out.groups(s, n); return;
    case V_UUID: {  // uuid's binary form is its 16 bytes; memcmp.rs:86-93 writes d3, d2, d1, rest
        m.bin(s, n, scratch);
        if (n != 16) raise(CZI_E_CORRUPT, "msgpack: uuid of %u bytes", n);
        out.u8(UUID_TAG);
        out.raw(s + 6, 2); out.raw(s + 4, 2); out.raw(s, 4); out.raw(s + 8, 8);
        return;
    }
    case V_LIST: case V_SET: {
        uint32_t k = m.array();
        out.u8(v == V_LIST ? LIST_TAG : SET_TAG);
        while (k--) mp_to_memcmp(m, out, scratch, depth + 1);
        out.u8(INIT_TAG);
        return;
    }
    case V_VEC: {
        int el;
        mp_vec(m, el, s, n, scratch);
        out.u8(VEC_TAG);
        out.u8(el == 0 ? VEC_F32 : VEC_F64);
        const uint32_t w = el == 0 ? 4 : 8;
        out.u64be(n / w);
        for (uint32_t i = 0; i < n; i += w)  // little-endian payload -> big-endian key bytes
            for (uint32_t b = 0; b < w; b++) out.u8(s[i + w - 1 - b]);
        return;
    }
    case V_VALIDITY: {  // struct Validity { timestamp: ValidityTs(Reverse<i64>), is_assert: Reverse<bool> } as an array
        if (m.array() != 2) raise(CZI_E_CORRUPT, "msgpack: a validity must be a 2-tuple");
        const int64_t ts = m.integer();
        const bool is_assert = m.boolean();
        out.u8(VLD_TAG);
        out.u64be(~((uint64_t)ts ^ SIGN_MARK));
        out.u8(is_assert ? 0 : 1);
        return;
    }
    default:
        raise(CZI_E_UNSUPPORTED, "a %s node value in the value part of a row", kVariantName[v]);
    }
}

// ------------------------------------------------------------------------------------------------ byte-string table
inline uint64_t mix(uint64_t a, uint64_t b) {
    const __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
uint64_t hash_bytes(const uint8_t *p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = mix(h ^ w, 0xA0761D6478BD642Full);
        p += 8;
        n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = mix(h ^ w, 0xE7037ED1A0B428DBull);
    }
    return mix(h, 0x8EBC6AF09C88C6E3ull);
}

// byte strings -> dense ids in insertion order; the strings are kept (concatenated) for the way back.
// Lookups are latency bound (a random slot, then the candidate's bytes): a slot therefore carries where the candidate's
// bytes are (no second indirection), and callers that know their next keys call hint() on them a few dozen lookups
// ahead so that both cache lines are in flight before the authoritative probe.
struct ByteTable {
    struct Slot {
        uint64_t at;   // offset of the entry in `arena`: u32 length, then the bytes
        uint32_t id;   // CZ_NONE = empty
        uint32_t tag;  // high half of the hash
    };
    std::vector<Slot> slots;
    std::vector<uint8_t> arena;
    std::vector<uint8_t> bytes;  // the same strings back to back, in id order (the way back)
    std::vector<uint64_t> off{0};
    uint64_t mask = 0;

    ByteTable() { rehash(1024); }
    uint32_t size() const { return (uint32_t)(off.size() - 1); }
    void rehash(size_t cap) {
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(cap, Slot{0, CZ_NONE, 0});
        mask = cap - 1;
        for (const Slot &s : old)
            if (s.id != CZ_NONE) {
                uint32_t len;
                memcpy(&len, arena.data() + s.at, 4);
                uint64_t i = hash_bytes(arena.data() + s.at + 4, len) & mask;
                while (slots[i].id != CZ_NONE) i = (i + 1) & mask;
                slots[i] = s;
            }
    }
    bool same(const Slot &s, const uint8_t *k, size_t len) const {
        uint32_t l;
        memcpy(&l, arena.data() + s.at, 4);
        return l == len && memcmp(arena.data() + s.at + 4, k, len) == 0;
    }
    void hint_slot(uint64_t h) const { __builtin_prefetch(&slots[h & mask]); }
    void hint_bytes(uint64_t h) const {
        const Slot &s = slots[h & mask];
        if (s.id != CZ_NONE && s.tag == (uint32_t)(h >> 32)) __builtin_prefetch(arena.data() + s.at);
    }
    uint32_t find_h(const uint8_t *k, size_t len, uint64_t h) const {
        const uint32_t tag = (uint32_t)(h >> 32);
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
            const Slot &s = slots[i];
            if (s.id == CZ_NONE) return CZ_NONE;
            if (s.tag == tag && same(s, k, len)) return s.id;
        }
    }
    uint32_t find(const uint8_t *k, size_t len) const { return find_h(k, len, hash_bytes(k, len)); }
    uint32_t find_or_insert_h(const uint8_t *k, size_t len, uint64_t h) {
        const uint32_t tag = (uint32_t)(h >> 32);
        uint64_t i = h & mask;
        for (;; i = (i + 1) & mask) {
            const Slot &s = slots[i];
            if (s.id == CZ_NONE) break;
            if (s.tag == tag && same(s, k, len)) return s.id;
        }
        const uint32_t id = size();
        if (id >= 0xFFFFFFFEu) raise(CZI_E_TOO_LARGE, "more than 2^32 - 2 distinct nodes");
        if (len > 0xFFFFFFFFull) raise(CZI_E_TOO_LARGE, "a node value of %zu bytes", len);
        const uint64_t at = arena.size();
        const uint32_t l = (uint32_t)len;
        arena.insert(arena.end(), (const uint8_t *)&l, (const uint8_t *)&l + 4);
        arena.insert(arena.end(), k, k + len);
        bytes.insert(bytes.end(), k, k + len);
        off.push_back(bytes.size());
        slots[i] = Slot{at, id, tag};
        if ((uint64_t)(id + 1) * 2 > mask + 1) rehash((mask + 1) * 2);
        return id;
    }
    uint32_t find_or_insert(const uint8_t *k, size_t len) { return find_or_insert_h(k, len, hash_bytes(k, len)); }

    // renumber: id = rank of the string in byte order (memcmp keys: the DataValue order); a, b are relabelled with it
    std::vector<uint32_t> relabel_by_rank(std::vector<uint32_t> &a, std::vector<uint32_t> &b) {
        const uint32_t n = size();
        std::vector<uint32_t> order(n), rank(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const size_t lx = off[x + 1] - off[x], ly = off[y + 1] - off[y];
            const int c = memcmp(bytes.data() + off[x], bytes.data() + off[y], std::min(lx, ly));
            return c ? c < 0 : lx < ly;
        });
        for (uint32_t r = 0; r < n; r++) rank[order[r]] = r;
        std::vector<uint8_t> nb;
        nb.reserve(bytes.size());
        std::vector<uint64_t> noff{0};
        noff.reserve(n + 1);
        for (uint32_t r = 0; r < n; r++) {
            nb.insert(nb.end(), bytes.begin() + off[order[r]], bytes.begin() + off[order[r] + 1]);
            noff.push_back(nb.size());
        }
        bytes.swap(nb);
        off.swap(noff);
        for (Slot &s : slots)
            if (s.id != CZ_NONE) s.id = rank[s.id];
        for (uint32_t &x : a) x = rank[x];
        for (uint32_t &x : b) x = rank[x];
        return rank;
    }
};

// ------------------------------------------------------------------------------------------------ rows
struct Row {
    const uint8_t *k, *kend, *v, *vend;
};
Row row_at(const czi_rows *r, uint64_t i) {
    Row x;
    x.k = r->keys + r->key_off[i];
    x.kend = r->keys + r->key_off[i + 1];
    if (x.kend - x.k < 8) raise(CZI_E_CORRUPT, "row %llu: a stored key is at least the 8-byte relation id", (unsigned long long)i);
    x.k += 8;  // ENCODED_KEY_MIN_LEN, data/tuple.rs:86
    x.v = x.vend = nullptr;
    if (r->vals && r->val_off) {
        x.v = r->vals + r->val_off[i];
        x.vend = r->vals + r->val_off[i + 1];
        if (x.vend - x.v >= 8) x.v += 8; else x.v = x.vend;  // extend_tuple_from_v: empty value = no columns
    }
    return x;
}
void check_rows(const czi_rows *r, const char *what) {
    if (!r) raise(CZI_E_INVALID, "%s: null rows", what);
    if (r->n_rows && (!r->keys || !r->key_off)) raise(CZI_E_INVALID, "%s: null key buffers", what);
    if (r->n_rows && (r->vals == nullptr) != (r->val_off == nullptr)) raise(CZI_E_INVALID, "%s: vals and val_off go together", what);
}

// walks the columns of one stored row in tuple order: key columns first, then the msgpack array of the value
struct ColumnCursor {
    const uint8_t *kp, *kend;
    uint32_t key_left;
    Mp m;
    uint32_t val_left;
    bool val_open;
    ColumnCursor(const Row &r, uint32_t n_key_cols) : kp(r.k), kend(r.kend), key_left(n_key_cols), m{r.v, r.vend}, val_left(0), val_open(false) {}
    // 0 = no more columns, 1 = a key column [a, b), 2 = a value column (m.p stands on it; the caller consumes it)
    int next(const uint8_t *&a, const uint8_t *&b) {
        if (key_left) {
            key_left--;
            a = kp;
            b = kp = mc_skip(kp, kend);
            return 1;
        }
        if (!val_open) {
            val_open = true;
            val_left = (m.p && m.p < m.end) ? m.array() : 0;
        }
        if (!val_left) return 0;
        val_left--;
        return 2;
    }
};

template <class F>
int guarded(F &&f) {
    try {
        f();
        return CZI_OK;
    } catch (const Error &e) {
        return e.code;
    } catch (const std::bad_alloc &) {
        return fail(CZI_E_OOM, "out of host memory");
    } catch (const std::exception &e) {  // nothing may unwind through the C ABI
        return fail(CZI_E_INVALID, "%s", e.what());
    }
}

}  // namespace

// ==================================================================================================== graph
struct czi_graph {
    ByteTable nodes;  // one-thread path: the table; threaded path: only its bytes / off (node keys in id order) are filled
    // threaded path: the endpoints are partitioned by hash, one table per partition (local ids) + local -> global id
    std::vector<ByteTable> parts;
    std::vector<std::vector<uint32_t>> gid;
    std::vector<uint32_t> src, dst;
    std::vector<float> w;
    bool weighted = false, undirected = false;
};

extern "C" const char *czi_last_error(void) { return g_err.c_str(); }
extern "C" const char *czi_version(void) { return "cozo_ingest 0.1 (memcmp keys + rmp-serde 1.2 values of cozo 0.7.6)"; }

namespace {

struct End {
    const uint8_t *a;
    size_t len;
    uint64_t h;
};

// columns 0 and 1 of row i as memcmp byte slices (+ hash), column 2 as the weight when asked for (fixed_rule/mod.rs:146-262)
inline void parse_edge_row(const czi_rows *rel, uint64_t i, End ends[2], Buf tmp[2], std::vector<uint8_t> &scratch, bool weighted,
                           bool allow_negative_weights, float *w) {
    ColumnCursor cur(row_at(rel, i), rel->n_key_cols);
    for (int c = 0; c < 2; c++) {
        const uint8_t *a, *b;
        const int where = cur.next(a, b);
        if (where == 0) raise(CZI_E_NOT_AN_EDGE, "The relation cannot be interpreted as an edge");  // mod.rs:846-850
        if (where == 2) {  // the endpoint lives in the value part: its memcmp form is its identity
            tmp[c].b.clear();
            mp_to_memcmp(cur.m, tmp[c], scratch);
            a = tmp[c].b.data();
            b = a + tmp[c].b.size();
        }
        ends[c].a = a;
        ends[c].len = (size_t)(b - a);
        ends[c].h = hash_bytes(a, ends[c].len);
    }
    if (!weighted) return;
    const uint8_t *a, *b;
    const int where = cur.next(a, b);
    double f = 1.0;
    if (where == 1) {
        if (*a != NUM_TAG) raise(CZI_E_BAD_WEIGHT, "row %llu: the value cannot be interpreted as an edge weight", (unsigned long long)i);
        f = mc_num(a + 1, b).f;
    } else if (where == 2) {
        if (mp_value_head(cur.m) != V_NUM) raise(CZI_E_BAD_WEIGHT, "row %llu: the value cannot be interpreted as an edge weight", (unsigned long long)i);
        f = mp_num(cur.m).f;
    }
    if (where != 0 && (!std::isfinite(f) || (f < 0.0 && !allow_negative_weights)))
        raise(CZI_E_BAD_WEIGHT, "row %llu: the value %g cannot be interpreted as an edge weight", (unsigned long long)i, f);
    *w = (float)f;
}

// One thread.  Rows are handled in blocks: parse a block (endpoint slices + weights), hash every endpoint and start its two
// cache lines moving (slot, then candidate bytes), and only then resolve the endpoints IN ROW ORDER -- the
// first-appearance numbering is untouched, the lookups of a block overlap instead of queueing behind each other.
void assign_ids_serial(const czi_rows *rel, czi_graph &g, bool allow_negative_weights) {
    const uint64_t E = rel->n_rows;
    constexpr uint32_t kBlock = 32;
    End ends[2 * kBlock];
    Buf tmp[2 * kBlock];
    std::vector<uint8_t> scratch;
    float unused = 0;
    for (uint64_t i0 = 0; i0 < E; i0 += kBlock) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(kBlock, E - i0);
        for (uint32_t r = 0; r < nb; r++) {
            parse_edge_row(rel, i0 + r, ends + 2 * r, tmp + 2 * r, scratch, g.weighted, allow_negative_weights,
                           g.weighted ? &g.w[i0 + r] : &unused);
            g.nodes.hint_slot(ends[2 * r].h);
            g.nodes.hint_slot(ends[2 * r + 1].h);
        }
        for (uint32_t x = 0; x < 2 * nb; x++) g.nodes.hint_bytes(ends[x].h);
        for (uint32_t r = 0; r < nb; r++) {
            g.src[i0 + r] = g.nodes.find_or_insert_h(ends[2 * r].a, ends[2 * r].len, ends[2 * r].h);
            g.dst[i0 + r] = g.nodes.find_or_insert_h(ends[2 * r + 1].a, ends[2 * r + 1].len, ends[2 * r + 1].h);
        }
    }
}

// ---- threads ----
struct WorkerError {
    std::mutex m;
    bool set = false;
    uint64_t order = 0;  // the failure of the earliest row wins, like a scan that stops at the first bad row
    int code = 0;
    std::string msg;
    void report(uint64_t ord, int c, const std::string &s) {
        std::lock_guard<std::mutex> lock(m);
        if (!set || ord < order) {
            set = true;
            order = ord;
            code = c;
            msg = s;
        }
    }
};

template <class F>
void parallel_for(uint32_t threads, F &&f) {
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (uint32_t t = 1; t < threads; t++) pool.emplace_back([&f, t] { f(t); });
    f(0);
    for (std::thread &th : pool) th.join();
}

inline uint32_t part_of(uint64_t h, uint32_t P) { return (uint32_t)(((h >> 56) * P) >> 8); }

// T threads.  The first-appearance numbering looks sequential but is not: the id of a node is the number of FIRST
// appearances before its own, and whether an endpoint is a first appearance only depends on the endpoints with the same
// hash partition before it.  So: (A) rows in parallel: parse, hash, weights; (B) partitions in parallel: every thread
// walks the hash array in order, resolves the endpoints of ITS partition in its own table (local ids in order of first
// appearance) and marks first appearances in a bitmap over endpoint positions; (C) global id = rank of the first position
// in that bitmap (a prefix popcount); (D) rows in parallel: local -> global.  Same ids as the one-thread path, bit for bit.
void assign_ids_threaded(const czi_rows *rel, czi_graph &g, bool allow_negative_weights, uint32_t T) {
    const uint64_t E = rel->n_rows;
    const uint32_t P = T;
    WorkerError err;
    // per-endpoint scratch, deliberately NOT zero-filled: the threads that write it take the page faults, in parallel
    std::unique_ptr<uint64_t[]> hashes(new uint64_t[2 * E + 1]);
    std::unique_ptr<const uint8_t *[]> eptr(new const uint8_t *[2 * E + 1]);  // the endpoint's memcmp bytes: inside the caller's key
    std::unique_ptr<uint32_t[]> elen(new uint32_t[2 * E + 1]);  // buffer, or (value-part endpoints) in the parsing thread's arena
    std::unique_ptr<uint32_t[]> loc(new uint32_t[2 * E + 1]);
    struct Arena {
        std::vector<std::unique_ptr<uint8_t[]>> chunks;
        size_t left = 0;
        uint8_t *at = nullptr;
        const uint8_t *keep(const uint8_t *p, size_t n) {
            if (n > left) {
                const size_t sz = std::max<size_t>(n, 1u << 20);
                chunks.emplace_back(new uint8_t[sz]);
                at = chunks.back().get();
                left = sz;
            }
            memcpy(at, p, n);
            const uint8_t *r = at;
            at += n;
            left -= n;
            return r;
        }
    };
    std::vector<Arena> arenas(T);
    // (A)
    parallel_for(T, [&](uint32_t t) {
        const uint64_t lo = E * t / T, hi = E * (t + 1) / T;
        End ends[2];
        Buf tmp[2];
        std::vector<uint8_t> scratch;
        float unused = 0;
        uint64_t i = lo;
        try {
            for (; i < hi; i++) {
                parse_edge_row(rel, i, ends, tmp, scratch, g.weighted, allow_negative_weights, g.weighted ? &g.w[i] : &unused);
                for (int c = 0; c < 2; c++) {
                    if (ends[c].len > 0xFFFFFFFFull) raise(CZI_E_TOO_LARGE, "a node value of %zu bytes", ends[c].len);
                    hashes[2 * i + c] = ends[c].h;
                    elen[2 * i + c] = (uint32_t)ends[c].len;
                    eptr[2 * i + c] = ends[c].a == tmp[c].b.data() ? arenas[t].keep(ends[c].a, ends[c].len) : ends[c].a;
                }
            }
        } catch (const Error &e) {
            err.report(i, e.code, g_err);
        } catch (const std::exception &e) {
            err.report(i, CZI_E_INVALID, e.what());
        }
    });
    if (err.set) {
        g_err = err.msg;
        throw Error{err.code};
    }
    // (B)
    g.parts.resize(P);
    g.gid.resize(P);
    std::vector<uint64_t> first_bits((2 * E + 63) / 64 + 1, 0);
    std::vector<std::vector<uint64_t>> first_pos(P);
    parallel_for(T, [&](uint32_t p) {
        ByteTable &tab = g.parts[p];
        constexpr uint32_t kBlock = 64;
        uint64_t pend[kBlock];
        uint32_t np = 0;
        try {
            auto flush = [&] {
                for (uint32_t x = 0; x < np; x++) tab.hint_bytes(hashes[pend[x]]);
                for (uint32_t x = 0; x < np; x++) {
                    const uint64_t j = pend[x];
                    const uint32_t before = tab.size();
                    const uint32_t id = tab.find_or_insert_h(eptr[j], elen[j], hashes[j]);
                    loc[j] = id;
                    if (id == before) {  // a first appearance
                        first_pos[p].push_back(j);
                        __atomic_fetch_or(&first_bits[j >> 6], 1ull << (j & 63), __ATOMIC_RELAXED);
                    }
                }
                np = 0;
            };
            for (uint64_t j = 0; j < 2 * E; j++) {
                if (part_of(hashes[j], P) != p) continue;
                tab.hint_slot(hashes[j]);
                __builtin_prefetch(eptr[j]);
                pend[np++] = j;
                if (np == kBlock) flush();
            }
            flush();
        } catch (const Error &e) {
            err.report(0, e.code, g_err);
        } catch (const std::exception &e) {
            err.report(0, CZI_E_INVALID, e.what());
        }
    });
    if (err.set) {
        g_err = err.msg;
        throw Error{err.code};
    }
    // (C) rank of every first position
    std::vector<uint32_t> word_rank(first_bits.size());
    uint64_t total = 0;
    for (size_t wd = 0; wd < first_bits.size(); wd++) {
        word_rank[wd] = (uint32_t)total;
        total += (uint64_t)__builtin_popcountll(first_bits[wd]);
    }
    if (total >= 0xFFFFFFFEull) raise(CZI_E_TOO_LARGE, "more than 2^32 - 2 distinct nodes");
    const uint32_t n = (uint32_t)total;
    std::vector<uint64_t> &off = g.nodes.off;
    off.assign((size_t)n + 1, 0);
    parallel_for(T, [&](uint32_t p) {
        const ByteTable &tab = g.parts[p];
        g.gid[p].resize(tab.size());
        for (uint32_t l = 0; l < tab.size(); l++) {
            const uint64_t pos = first_pos[p][l];
            const uint32_t id = word_rank[pos >> 6] + (uint32_t)__builtin_popcountll(first_bits[pos >> 6] & ((1ull << (pos & 63)) - 1));
            g.gid[p][l] = id;
            off[(size_t)id + 1] = tab.off[l + 1] - tab.off[l];  // lengths first, prefix sum below
        }
    });
    for (uint32_t v = 0; v < n; v++) off[v + 1] += off[v];
    g.nodes.bytes.resize(off[n]);
    parallel_for(T, [&](uint32_t p) {
        const ByteTable &tab = g.parts[p];
        for (uint32_t l = 0; l < tab.size(); l++)
            memcpy(g.nodes.bytes.data() + off[g.gid[p][l]], tab.bytes.data() + tab.off[l], tab.off[l + 1] - tab.off[l]);
    });
    // (D)
    parallel_for(T, [&](uint32_t t) {
        for (uint64_t i = E * t / T; i < E * (t + 1) / T; i++) {
            g.src[i] = g.gid[part_of(hashes[2 * i], P)][loc[2 * i]];
            g.dst[i] = g.gid[part_of(hashes[2 * i + 1], P)][loc[2 * i + 1]];
        }
    });
}

uint32_t ingest_threads(uint64_t rows) {
    const char *env = getenv("CZI_THREADS");
    long t = env ? atol(env) : (long)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const char *min_env = getenv("CZI_THREADED_MIN_ROWS");
    const uint64_t min_rows = min_env ? (uint64_t)atoll(min_env) : (1ull << 18);
    if (t < 2 || rows < min_rows) return 1;
    return (uint32_t)std::min(t, 64l);
}

// CsrLayout::Sorted: lists ascending by target, parallel edges kept, ties in input order.  The entries are sorted as
// 64-bit (source << 32 | target) keys by a stable LSD radix sort over just the bits two node ids need: every pass
// streams the arrays through 2^11 write cursors, where a counting sort over N buckets would pay a cache miss per entry.
// Threads split every pass by position: per-thread histograms, one prefix over (digit, thread), per-thread scatter --
// stable, so the result does not depend on the thread count.
// `undirected` feeds every row a second time with the ends swapped, right after it (fixed_rule/mod.rs:187-191).
void build_csr(const czi_graph &g, bool inverse, uint32_t *offsets, uint32_t *targets, float *weights) {
    const uint32_t n = g.nodes.size();
    const uint64_t rows = g.src.size();
    const uint64_t e = g.undirected ? rows * 2 : rows;
    const std::vector<uint32_t> &A = inverse ? g.dst : g.src, &B = inverse ? g.src : g.dst;
    const bool carry = weights != nullptr && g.weighted;
    const uint32_t T = ingest_threads(e);
    std::unique_ptr<uint64_t[]> key(new uint64_t[e + 1]), key2(new uint64_t[e + 1]);
    std::unique_ptr<uint32_t[]> pay, pay2;  // the row a key came from (weights follow their edges through the sort)
    if (carry) {
        pay.reset(new uint32_t[e + 1]);
        pay2.reset(new uint32_t[e + 1]);
    }
    int id_bits = 1;
    while (id_bits < 32 && (1ull << id_bits) < (uint64_t)n) id_bits++;
    constexpr int kRadix = 11;
    struct Digit {
        int shift, bits;
    };
    std::vector<Digit> digits;  // target bits first (least significant), then source bits
    for (int half = 0; half < 2; half++)
        for (int done = 0; done < id_bits; done += kRadix) digits.push_back({32 * half + done, std::min(kRadix, id_bits - done)});
    const size_t D = digits.size();
    // cnt[t][d][x]: how many keys of thread t's slice have value x in digit d
    std::vector<std::vector<uint64_t>> cnt(T, std::vector<uint64_t>(D << kRadix, 0));
    parallel_for(T, [&](uint32_t t) {
        uint64_t *c = cnt[t].data();
        // digit 0 always; with one thread the slice is the whole array in every pass, so all digits can be counted now
        const size_t Dnow = T == 1 ? D : 1;
        auto tally = [&](uint64_t k) {
            for (size_t d = 0; d < Dnow; d++) c[(d << kRadix) + ((k >> digits[d].shift) & ((1ull << digits[d].bits) - 1))]++;
        };
        for (uint64_t r = rows * t / T; r < rows * (t + 1) / T; r++) {
            const uint64_t a = A[r], b = B[r];
            if (g.undirected) {
                key[2 * r] = a << 32 | b;
                key[2 * r + 1] = b << 32 | a;
                tally(key[2 * r]);
                tally(key[2 * r + 1]);
                if (carry) pay[2 * r] = pay[2 * r + 1] = (uint32_t)r;
            } else {
                key[r] = a << 32 | b;
                tally(key[r]);
                if (carry) pay[r] = (uint32_t)r;
            }
        }
    });
    // a thread's slice of the ENTRIES is the image of its slice of the rows, in every pass (positions, not values)
    auto lo_of = [&](uint32_t t) { return (rows * t / T) * (g.undirected ? 2 : 1); };
    for (size_t d = 0; d < D; d++) {
        const int shift = digits[d].shift;
        const uint64_t m = (1ull << digits[d].bits) - 1;
        if (d > 0 && T > 1) {  // digit 0 was counted while the keys were built; the slices of later passes are known only now
            parallel_for(T, [&](uint32_t t) {
                uint64_t *c = cnt[t].data() + (d << kRadix);
                std::fill(c, c + m + 1, 0);
                for (uint64_t j = lo_of(t); j < lo_of(t + 1); j++) c[(key[j] >> shift) & m]++;
            });
        }
        uint64_t sum = 0;
        for (uint64_t x = 0; x <= m; x++)
            for (uint32_t t = 0; t < T; t++) {
                uint64_t &c = cnt[t][(d << kRadix) + x];
                const uint64_t v = c;
                c = sum;
                sum += v;
            }
        parallel_for(T, [&](uint32_t t) {
            uint64_t *c = cnt[t].data() + (d << kRadix);
            if (carry) {
                for (uint64_t j = lo_of(t); j < lo_of(t + 1); j++) {
                    const uint64_t at = c[(key[j] >> shift) & m]++;
                    key2[at] = key[j];
                    pay2[at] = pay[j];
                }
            } else {
                for (uint64_t j = lo_of(t); j < lo_of(t + 1); j++) key2[c[(key[j] >> shift) & m]++] = key[j];
            }
        });
        key.swap(key2);
        if (carry) pay.swap(pay2);
    }
    // sorted by (source, target): targets are the low halves; offsets[v] = first position whose source is >= v
    parallel_for(T, [&](uint32_t t) {
        const uint64_t lo = e * t / T, hi = e * (t + 1) / T;
        for (uint64_t j = lo; j < hi; j++) {
            targets[j] = (uint32_t)key[j];
            if (weights) weights[j] = carry ? g.w[pay[j]] : 1.0f;
            const uint32_t sj = (uint32_t)(key[j] >> 32);
            const uint32_t prev = j ? (uint32_t)(key[j - 1] >> 32) + 1 : 0;
            for (uint32_t v = prev; v <= sj; v++) offsets[v] = (uint32_t)j;  // empty when the source repeats
        }
    });
    const uint32_t last = e ? (uint32_t)(key[e - 1] >> 32) + 1 : 0;
    for (uint64_t v = last; v <= n; v++) offsets[v] = (uint32_t)e;
}

}  // namespace

extern "C" int czi_graph_ingest(const czi_rows *rel, uint32_t flags, czi_graph **out) {
    if (!out) return fail(CZI_E_INVALID, "null out");
    *out = nullptr;
    std::unique_ptr<czi_graph> g(new (std::nothrow) czi_graph);
    if (!g) return fail(CZI_E_OOM, "out of host memory");
    const int rc = guarded([&] {
        check_rows(rel, "czi_graph_ingest");
        g->weighted = (flags & CZI_WEIGHTED) != 0;
        g->undirected = (flags & CZI_UNDIRECTED) != 0;
        const bool allow_negative_weights = (flags & CZI_ALLOW_NEGATIVE_WEIGHTS) != 0;
        const uint64_t E = rel->n_rows;
        if ((g->undirected ? E * 2 : E) >= 0xFFFFFFFFull) raise(CZI_E_TOO_LARGE, "%llu rows do not fit u32 CSR offsets", (unsigned long long)E);
        g->src.resize(E);
        g->dst.resize(E);
        if (g->weighted) g->w.resize(E);
        const uint32_t T = ingest_threads(E);
        if (T > 1) assign_ids_threaded(rel, *g, allow_negative_weights, T);
        else assign_ids_serial(rel, *g, allow_negative_weights);
        if (flags & CZI_ORDERED_IDS) {
            const std::vector<uint32_t> rank = g->nodes.relabel_by_rank(g->src, g->dst);
            for (std::vector<uint32_t> &m : g->gid)
                for (uint32_t &x : m) x = rank[x];
        }
    });
    if (rc) return rc;
    *out = g.release();
    return CZI_OK;
}

extern "C" void czi_graph_free(czi_graph *g) { delete g; }
extern "C" uint32_t czi_graph_node_count(const czi_graph *g) { return g ? g->nodes.size() : 0; }
extern "C" uint64_t czi_graph_edge_count(const czi_graph *g) { return g ? (uint64_t)g->src.size() * (g->undirected ? 2 : 1) : 0; }

extern "C" int czi_graph_csr(const czi_graph *g, int inverse, uint32_t *offsets, uint32_t *targets, float *weights) {
    if (!g || !offsets || (!targets && !g->src.empty())) return fail(CZI_E_INVALID, "null argument");
    return guarded([&] { build_csr(*g, inverse != 0, offsets, targets, weights); });
}

extern "C" int czi_graph_node_keys(const czi_graph *g, const uint8_t **bytes, const uint64_t **off) {
    if (!g || !bytes || !off) return fail(CZI_E_INVALID, "null argument");
    *bytes = g->nodes.bytes.data();
    *off = g->nodes.off.data();
    return CZI_OK;
}

extern "C" uint32_t czi_graph_lookup(const czi_graph *g, const uint8_t *key, uint64_t len) {
    if (!g || (!key && len)) return CZ_NONE;
    if (g->parts.empty()) return g->nodes.find(key, (size_t)len);
    const uint64_t h = hash_bytes(key, (size_t)len);
    const uint32_t p = part_of(h, (uint32_t)g->parts.size());
    const uint32_t l = g->parts[p].find_h(key, (size_t)len, h);
    return l == CZ_NONE ? CZ_NONE : g->gid[p][l];
}

// ==================================================================================================== hnsw
struct czi_hnsw {
    uint32_t n = 0, dim = 0;
    int32_t metric = 0, n_levels = 0;
    uint32_t entry = 0;
    std::vector<float> vectors;
    std::vector<uint32_t> level_size;
    std::vector<int32_t> level_width;
    std::vector<std::vector<uint32_t>> level_nodes, level_nbrs;
    std::vector<const uint32_t *> level_nodes_p, level_nbrs_p;
    std::vector<uint64_t> base_row;
    std::vector<uint32_t> field;
    std::vector<int32_t> sub;
    uint64_t n_rows = 0, n_self = 0, n_live = 0, n_ignored = 0;
};

namespace {

struct IdxRow {
    int64_t layer;
    const uint8_t *fr, *fr_end, *to, *to_end;  // [key x K, field, sub] of either end, as raw key bytes
    const uint8_t *fr_key_end, *to_key_end;    // end of the K row-key columns inside each
    uint64_t to_hash;
    bool ignore;
};

int64_t key_int(const uint8_t *a, const uint8_t *b, const char *what) {
    if (a >= b || *a != NUM_TAG) raise(CZI_E_CORRUPT, "index row: %s is not a number", what);
    const NumVal x = mc_num(a + 1, b);
    if (!x.is_int) raise(CZI_E_CORRUPT, "index row: %s is not an integer", what);
    return x.i;
}

// one row of `tbl:idx`; returns false for the canary row (layer 1, all-Null ends; hnsw.rs:641-669)
bool parse_idx_row(const czi_rows *idx, uint64_t i, uint32_t K, IdxRow &r) {
    const Row row = row_at(idx, i);
    const uint8_t *p = row.k, *q = mc_skip(p, row.kend);
    r.layer = key_int(p, q, "layer");
    if (r.layer > 0) return false;
    p = q;
    for (int side = 0; side < 2; side++) {
        const uint8_t *start = p;
        for (uint32_t c = 0; c < K; c++) p = mc_skip(p, row.kend);
        const uint8_t *key_end = p;
        p = mc_skip(p, row.kend);
        p = mc_skip(p, row.kend);
        if (side == 0) { r.fr = start; r.fr_key_end = key_end; r.fr_end = p; }
        else { r.to = start; r.to_key_end = key_end; r.to_end = p; }
    }
    if (p != row.kend) raise(CZI_E_CORRUPT, "index row %llu: %zu bytes after the last key column", (unsigned long long)i, (size_t)(row.kend - p));
    // value [dist, hash, ignore_link]: only ignore_link matters for the link tables
    r.ignore = false;
    if (row.v && row.v < row.vend) {
        Mp m{row.v, row.vend};
        if (m.array() != 3) raise(CZI_E_CORRUPT, "index row %llu: the value is not [dist, hash, ignore_link]", (unsigned long long)i);
        m.skip();
        m.skip();
        if (mp_value_head(m) != V_BOOL) raise(CZI_E_CORRUPT, "index row %llu: ignore_link is not a bool", (unsigned long long)i);
        r.ignore = m.boolean();
    } else {
        raise(CZI_E_CORRUPT, "index row %llu has no value", (unsigned long long)i);
    }
    return true;
}

// the vector a node names, copied as f32[dim] into dst; col = the field's column of the base row
void copy_vector(const czi_rows *base, uint64_t brow, uint32_t fld, int32_t sub_idx, uint32_t dim, float *dst,
                 std::vector<uint8_t> &scratch) {
    ColumnCursor cur(row_at(base, brow), base->n_key_cols);
    const uint8_t *a = nullptr, *b = nullptr;
    int where = 0;
    for (uint32_t c = 0; c <= fld; c++) {
        where = cur.next(a, b);
        if (where == 0) raise(CZI_E_MISSING_ROW, "base row %llu has no column %u", (unsigned long long)brow, fld);
        if (where == 2 && c < fld) cur.m.skip();
    }
    if (where == 1) {  // a key column: [LIST_TAG elements.. INIT] or VEC_TAG
        if (sub_idx >= 0) {
            if (*a != LIST_TAG) raise(CZI_E_MISSING_ROW, "base row %llu column %u is not a list", (unsigned long long)brow, fld);
            a++;
            for (int32_t s = 0; s < sub_idx; s++) {
                if (a >= b || *a == INIT_TAG) raise(CZI_E_MISSING_ROW, "base row %llu column %u has no element %d", (unsigned long long)brow, fld, sub_idx);
                a = mc_skip(a, b);
            }
        }
        if (a >= b || *a != VEC_TAG) raise(CZI_E_MISSING_ROW, "base row %llu column %u: not a vector", (unsigned long long)brow, fld);
        if (a[1] != VEC_F32) raise(CZI_E_UNSUPPORTED, "F64 vectors are not supported on the GPU path");
        if (be64(a + 2) != dim) raise(CZI_E_CORRUPT, "base row %llu: vector of length %llu, index dimension %u", (unsigned long long)brow, (unsigned long long)be64(a + 2), dim);
        a += 10;
        for (uint32_t d = 0; d < dim; d++) {
            const uint32_t u = be32(a + 4 * d);
            memcpy(dst + d, &u, 4);
        }
        return;
    }
    Mp &m = cur.m;
    Variant v = mp_value_head(m);
    if (sub_idx >= 0) {
        if (v != V_LIST) raise(CZI_E_MISSING_ROW, "base row %llu column %u is not a list", (unsigned long long)brow, fld);
        const uint32_t k = m.array();
        if ((uint32_t)sub_idx >= k) raise(CZI_E_MISSING_ROW, "base row %llu column %u has no element %d", (unsigned long long)brow, fld, sub_idx);
        for (int32_t s = 0; s < sub_idx; s++) m.skip();
        v = mp_value_head(m);
    }
    if (v != V_VEC) raise(CZI_E_MISSING_ROW, "base row %llu column %u: not a vector", (unsigned long long)brow, fld);
    int el;
    const uint8_t *s;
    uint32_t nb;
    mp_vec(m, el, s, nb, scratch);
    if (el != 0) raise(CZI_E_UNSUPPORTED, "F64 vectors are not supported on the GPU path");
    if (nb != dim * 4) raise(CZI_E_CORRUPT, "base row %llu: vector of length %u, index dimension %u", (unsigned long long)brow, nb / 4, dim);
    memcpy(dst, s, nb);
}

void ingest_hnsw(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields, uint32_t dim,
                 int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw &h) {
    check_rows(idx, "czi_hnsw_ingest(idx)");
    check_rows(base, "czi_hnsw_ingest(base)");
    if (!dim || !m_max || !m_max0) raise(CZI_E_INVALID, "dim, m_max and m_max0 must be > 0");
    if (n_fields && !vec_fields) raise(CZI_E_INVALID, "null vec_fields");
    const uint32_t K = base->n_key_cols;
    if (idx->n_rows && idx->n_key_cols != 2 * K + 5)
        raise(CZI_E_INVALID, "the index relation of a %u-key base relation has %u key columns, not %u", K, 2 * K + 5, idx->n_key_cols);
    h.dim = dim;
    h.metric = metric;
    h.n_rows = idx->n_rows;

    const uint32_t T = ingest_threads(idx->n_rows);
    WorkerError err;
    auto rethrow = [&] {
        if (err.set) {
            g_err = err.msg;
            throw Error{err.code};
        }
    };
    auto guarded_worker = [&](uint64_t order, auto &&body) {
        try {
            body();
        } catch (const Error &e) {
            err.report(order, e.code, g_err);
        } catch (const std::exception &e) {
            err.report(order, CZI_E_INVALID, e.what());
        }
    };

    // pass 1 (threads over rows): parse every row once.  Rows are in key order, so the layers never decrease and the
    // canary rows (layer 1) are a suffix.
    const uint64_t R = idx->n_rows;
    std::unique_ptr<IdxRow[]> rows(new IdxRow[R + 1]);
    parallel_for(T, [&](uint32_t t) {
        uint64_t i = R * t / T;
        guarded_worker(i, [&] {
            for (; i < R * (t + 1) / T; i++)
                if (parse_idx_row(idx, i, K, rows[i])) rows[i].to_hash = hash_bytes(rows[i].to, (size_t)(rows[i].to_end - rows[i].to));
        });
    });
    rethrow();
    uint64_t nv = 0;  // rows that are not the canary
    for (uint64_t i = 0; i < R; i++) {
        if (i && rows[i].layer < rows[i - 1].layer) raise(CZI_E_CORRUPT, "index rows are not in key order (row %llu)", (unsigned long long)i);
        if (rows[i].layer <= 0) nv = i + 1;
    }
    if (nv == 0) return;  // only the canary, or nothing: an empty index (hnsw.rs:903-909)
    const int64_t min_layer = rows[0].layer;
    if (min_layer < -62) raise(CZI_E_CORRUPT, "index has layer %lld", (long long)min_layer);
    h.n_levels = (int32_t)(-min_layer) + 1;

    // group starts: a (layer, fr) group is one node's rows on one layer
    std::unique_ptr<uint8_t[]> group_start(new uint8_t[nv + 1]);
    parallel_for(T, [&](uint32_t t) {
        for (uint64_t j = nv * t / T; j < nv * (t + 1) / T; j++) {
            const size_t len = (size_t)(rows[j].fr_end - rows[j].fr);
            group_start[j] = j == 0 || rows[j].layer != rows[j - 1].layer || (size_t)(rows[j - 1].fr_end - rows[j - 1].fr) != len ||
                             memcmp(rows[j - 1].fr, rows[j].fr, len) != 0;
        }
    });

    // pass 2: node ids = order of the `fr` groups of layer 0 (every node has its self-loop row there, hnsw.rs:630-678)
    ByteTable nodes;
    for (uint64_t j = 0; j < nv; j++) {
        if (!group_start[j] || rows[j].layer != 0) continue;
        const uint32_t before = nodes.size();
        if (nodes.find_or_insert(rows[j].fr, (size_t)(rows[j].fr_end - rows[j].fr)) != before)
            raise(CZI_E_CORRUPT, "index rows are not in key order (layer 0)");
    }
    h.n = nodes.size();
    if (!h.n) raise(CZI_E_CORRUPT, "the index has upper layers but no layer 0");

    // node -> CompoundKey -> base row -> vector (threads over nodes)
    ByteTable base_keys;
    for (uint64_t i = 0; i < base->n_rows; i++) {
        const Row r = row_at(base, i);
        const uint8_t *p = r.k;
        for (uint32_t c = 0; c < K; c++) p = mc_skip(p, r.kend);
        if (base_keys.find_or_insert(r.k, (size_t)(p - r.k)) != (uint32_t)i) raise(CZI_E_CORRUPT, "base row %llu repeats a key", (unsigned long long)i);
    }
    h.base_row.resize(h.n);
    h.field.resize(h.n);
    h.sub.resize(h.n);
    h.vectors.resize((size_t)h.n * dim);
    parallel_for(T, [&](uint32_t t) {
        std::vector<uint8_t> scratch;
        uint32_t v = (uint32_t)((uint64_t)h.n * t / T);
        guarded_worker(v, [&] {
            for (; v < (uint32_t)((uint64_t)h.n * (t + 1) / T); v++) {
                const uint8_t *a = nodes.bytes.data() + nodes.off[v], *b = nodes.bytes.data() + nodes.off[v + 1];
                const uint8_t *p = a;
                for (uint32_t c = 0; c < K; c++) p = mc_skip(p, b);
                const uint8_t *q = mc_skip(p, b);
                const int64_t fld = key_int(p, q, "fr__field");
                const int64_t sb = key_int(q, b, "fr__sub_idx");
                const uint32_t br = base_keys.find(a, (size_t)(p - a));
                if (br == CZ_NONE) raise(CZI_E_MISSING_ROW, "node %u of the index has no base row (corrupted index, hnsw.rs:131-140)", v);
                bool known = n_fields == 0;
                for (uint32_t f = 0; f < n_fields; f++) known |= vec_fields[f] == (uint32_t)fld;
                if (fld < 0 || !known) raise(CZI_E_CORRUPT, "node %u: field %lld is not one of the index' vec_fields", v, (long long)fld);
                h.base_row[v] = br;
                h.field[v] = (uint32_t)fld;
                h.sub[v] = (int32_t)sb;
                copy_vector(base, br, (uint32_t)fld, (int32_t)sb, dim, h.vectors.data() + (size_t)v * dim, scratch);
            }
        });
    });
    rethrow();

    // pass 3a (threads over rows): what every row is -- dropped exactly as hnsw_get_neighbours drops it -- and the id of `to`
    enum : uint8_t { LIVE = 0, SELF = 1, SAME_ROW = 2, IGNORED = 3 };
    std::unique_ptr<uint8_t[]> kind(new uint8_t[nv + 1]);
    std::unique_ptr<uint32_t[]> to_id(new uint32_t[nv + 1]);
    parallel_for(T, [&](uint32_t t) {
        uint64_t j = nv * t / T;
        const uint64_t hi = nv * (t + 1) / T;
        guarded_worker(j, [&] {
            for (; j < hi; j++) {
                const IdxRow &r = rows[j];
                // the `to` lookups are random probes of the node table: keep two stages of them in flight
                if (j + 32 < hi) nodes.hint_slot(rows[j + 32].to_hash);
                if (j + 16 < hi) nodes.hint_bytes(rows[j + 16].to_hash);
                const size_t len = (size_t)(r.fr_end - r.fr), klen = (size_t)(r.fr_key_end - r.fr);
                const bool same_row = (size_t)(r.to_key_end - r.to) == klen && memcmp(r.to, r.fr, klen) == 0;
                if (same_row) {  // hnsw.rs:609-610: the self-loop row and links between vectors of one base row
                    kind[j] = ((size_t)(r.to_end - r.to) == len && memcmp(r.to, r.fr, len) == 0) ? SELF : SAME_ROW;
                } else if (r.ignore) {  // :616-619
                    kind[j] = IGNORED;
                } else {
                    kind[j] = LIVE;
                    to_id[j] = nodes.find_h(r.to, (size_t)(r.to_end - r.to), r.to_hash);
                    if (to_id[j] == CZ_NONE) raise(CZI_E_CORRUPT, "a link points at a node with no layer-0 row");
                }
            }
        });
    });
    rethrow();

    // pass 3b: per level, the nodes present (ascending id = key order) and their live rows
    const int L = h.n_levels;
    h.level_size.assign(L, 0);
    h.level_width.assign(L, 0);
    h.level_nodes.assign(L, {});
    h.level_nbrs.assign(L, {});
    std::vector<std::vector<uint64_t>> row_at_flat(L);  // per present node: where its live links start in flat[lv]
    std::vector<std::vector<uint32_t>> flat(L);         // concatenated live links per level
    for (uint64_t i = 0; i < nv;) {
        const int lv = (int)(-rows[i].layer);
        const uint32_t fr = nodes.find(rows[i].fr, (size_t)(rows[i].fr_end - rows[i].fr));
        if (fr == CZ_NONE) raise(CZI_E_CORRUPT, "a layer %lld row starts at a node with no layer-0 row", (long long)rows[i].layer);
        if (!h.level_nodes[lv].empty() && h.level_nodes[lv].back() >= fr) raise(CZI_E_CORRUPT, "index rows are not in key order");
        row_at_flat[lv].push_back(flat[lv].size());
        bool self = false;
        uint64_t j = i;
        do {
            switch (kind[j]) {
            case LIVE: flat[lv].push_back(to_id[j]); h.n_live++; break;
            case SELF: self = true; h.n_self++; break;
            case IGNORED: h.n_ignored++; break;
            default: break;
            }
            j++;
        } while (j < nv && !group_start[j]);
        if (!self) raise(CZI_E_CORRUPT, "node %u has rows on layer %lld but no self-loop row there", fr, (long long)rows[i].layer);
        h.level_nodes[lv].push_back(fr);
        i = j;
    }
    if (h.level_nodes[0].size() != h.n) raise(CZI_E_CORRUPT, "layer 0 holds %zu of %u nodes", h.level_nodes[0].size(), h.n);
    // row widths: m_max0 on level 0, m_max above -- one width for ALL upper levels (cz_hnsw_index_create keeps them in one
    // table); a live row longer than that (never written by hnsw_put_vector, which shrinks to m_max) widens its group
    uint32_t width0 = m_max0, width_up = m_max;
    for (int lv = 0; lv < L; lv++) {
        if (h.level_nodes[lv].empty()) raise(CZI_E_CORRUPT, "layer %d is empty", -lv);
        row_at_flat[lv].push_back(flat[lv].size());
        for (size_t r = 0; r < h.level_nodes[lv].size(); r++) {
            const uint32_t len = (uint32_t)(row_at_flat[lv][r + 1] - row_at_flat[lv][r]);
            if (lv == 0) width0 = std::max(width0, len); else width_up = std::max(width_up, len);
        }
    }
    for (int lv = 0; lv < L; lv++) {
        const size_t sz = h.level_nodes[lv].size();
        const uint32_t width = lv == 0 ? width0 : width_up;
        h.level_size[lv] = (uint32_t)sz;
        h.level_width[lv] = (int32_t)width;
        std::vector<uint32_t> &tab = h.level_nbrs[lv];
        tab.assign(sz * width, CZ_NONE);
        parallel_for(T, [&](uint32_t t) {
            for (size_t r = sz * t / T; r < sz * (t + 1) / T; r++) {
                // the scan yields `to` ends in key order = ascending id already; sort defensively (ids are what the kernels need)
                const uint64_t a0 = row_at_flat[lv][r], a1 = row_at_flat[lv][r + 1];
                std::copy(flat[lv].begin() + a0, flat[lv].begin() + a1, tab.begin() + r * width);
                std::sort(tab.begin() + r * width, tab.begin() + r * width + (a1 - a0));
            }
        });
    }
    h.entry = nodes.find(rows[0].fr, (size_t)(rows[0].fr_end - rows[0].fr));  // hnsw.rs:891-915
    for (int lv = 0; lv < L; lv++) {
        h.level_nodes_p.push_back(h.level_nodes[lv].data());
        h.level_nbrs_p.push_back(h.level_nbrs[lv].data());
    }
}

}  // namespace

extern "C" int czi_hnsw_ingest(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields,
                               uint32_t dim, int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw **out) {
    if (!out) return fail(CZI_E_INVALID, "null out");
    *out = nullptr;
    std::unique_ptr<czi_hnsw> h(new (std::nothrow) czi_hnsw);
    if (!h) return fail(CZI_E_OOM, "out of host memory");
    const int rc = guarded([&] { ingest_hnsw(idx, base, vec_fields, n_fields, dim, metric, m_max, m_max0, *h); });
    if (rc) return rc;
    *out = h.release();
    return CZI_OK;
}

extern "C" void czi_hnsw_free(czi_hnsw *h) { delete h; }

extern "C" int czi_hnsw_desc(const czi_hnsw *h, cz_hnsw_desc *desc, const float **vectors) {
    if (!h || !desc || !vectors) return fail(CZI_E_INVALID, "null argument");
    desc->n = h->n;
    desc->dim = h->dim;
    desc->metric = h->metric;
    desc->n_levels = h->n_levels;
    desc->entry = h->entry;
    desc->level_size = h->level_size.data();
    desc->level_width = h->level_width.data();
    desc->level_nodes = h->level_nodes_p.data();
    desc->level_nbrs = h->level_nbrs_p.data();
    *vectors = h->vectors.data();
    return CZI_OK;
}

extern "C" int czi_hnsw_nodes(const czi_hnsw *h, const uint64_t **base_row, const uint32_t **field, const int32_t **sub) {
    if (!h) return fail(CZI_E_INVALID, "null handle");
    if (base_row) *base_row = h->base_row.data();
    if (field) *field = h->field.data();
    if (sub) *sub = h->sub.data();
    return CZI_OK;
}

extern "C" int czi_hnsw_row_counts(const czi_hnsw *h, uint64_t *n_rows, uint64_t *n_self, uint64_t *n_live_links, uint64_t *n_ignored) {
    if (!h) return fail(CZI_E_INVALID, "null handle");
    if (n_rows) *n_rows = h->n_rows;
    if (n_self) *n_self = h->n_self;
    if (n_live_links) *n_live_links = h->n_live;
    if (n_ignored) *n_ignored = h->n_ignored;
    return CZI_OK;
}

// ==================================================================================================== write-back
struct czi_row_buf {
    std::vector<uint8_t> keys, vals;
    std::vector<uint64_t> key_off{0}, val_off{0};
};

namespace {

// FIPS 180-4 SHA-256 (Vector::get_hash, data/value.rs:333-348, hashes the little-endian element bytes)
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t block[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
            0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
            0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
            0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
            0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = be32(p + 4 * i);
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t *p, size_t n) {
        total += n;
        while (n) {
            const size_t take = std::min(n, 64 - fill);
            memcpy(block + fill, p, take);
            fill += take;
            p += take;
            n -= take;
            if (fill == 64) {
                compress(block);
                fill = 0;
            }
        }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)(h[i] >> 24);
            out[4 * i + 1] = (uint8_t)(h[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h[i] >> 8);
            out[4 * i + 3] = (uint8_t)h[i];
        }
    }
};

// msgpack writers for the three value columns, in rmp-serde 1.2.0's shape of the derived enums
void mp_put_str(std::vector<uint8_t> &o, const char *s) {
    const size_t n = strlen(s);  // variant names are < 32 bytes: fixstr
    o.push_back((uint8_t)(0xa0 | n));
    o.insert(o.end(), s, s + n);
}
void mp_put_variant(std::vector<uint8_t> &o, const char *name) {
    o.push_back(0x81);  // a one-entry map
    mp_put_str(o, name);
}
void mp_put_f64(std::vector<uint8_t> &o, double f) {
    mp_put_variant(o, "Num");
    mp_put_variant(o, "Float");
    uint64_t u;
    memcpy(&u, &f, 8);
    o.push_back(0xcb);
    for (int i = 7; i >= 0; i--) o.push_back((uint8_t)(u >> (8 * i)));
}
void mp_put_int(std::vector<uint8_t> &o, int64_t v) {  // the most compact form, as rmp's write_sint picks it
    mp_put_variant(o, "Num");
    mp_put_variant(o, "Int");
    if (v >= 0) {
        if (v < 128) o.push_back((uint8_t)v);
        else if (v < 256) { o.push_back(0xcc); o.push_back((uint8_t)v); }
        else if (v < 65536) { o.push_back(0xcd); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
        else if (v < 4294967296ll) { o.push_back(0xce); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
        else { o.push_back(0xcf); for (int i = 7; i >= 0; i--) o.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
    } else {
        if (v >= -32) o.push_back((uint8_t)v);
        else if (v >= -128) { o.push_back(0xd0); o.push_back((uint8_t)v); }
        else if (v >= -32768) { o.push_back(0xd1); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
        else if (v >= -2147483648ll) { o.push_back(0xd2); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
        else { o.push_back(0xd3); for (int i = 7; i >= 0; i--) o.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
    }
}
void mp_put_bytes(std::vector<uint8_t> &o, const uint8_t *p, size_t n) {
    mp_put_variant(o, "Bytes");
    if (n < 256) { o.push_back(0xc4); o.push_back((uint8_t)n); }
    else if (n < 65536) { o.push_back(0xc5); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n); }
    else { o.push_back(0xc6); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(n >> (8 * i))); }
    o.insert(o.end(), p, p + n);
}
void mp_put_bool(std::vector<uint8_t> &o, bool b) {
    mp_put_variant(o, "Bool");
    o.push_back(b ? 0xc3 : 0xc2);
}
void put_prefix(std::vector<uint8_t> &o, uint64_t relation_id) {
    for (int i = 7; i >= 0; i--) o.push_back((uint8_t)(relation_id >> (8 * i)));
}

void encode_index_rows(const cz_hnsw_desc *d, const float *vectors, const uint8_t *nk, const uint64_t *nko,
                       const double *const *level_dist, uint64_t rid, czi_row_buf &out) {
    if (d->n_levels <= 0 || d->n == 0) return;
    const uint32_t n = d->n;
    // key order of the nodes: rows of one layer are ordered by the `fr` bytes, then by the `to` bytes
    std::vector<uint32_t> order(n), rank(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    auto less = [&](uint32_t x, uint32_t y) {
        const size_t lx = nko[x + 1] - nko[x], ly = nko[y + 1] - nko[y];
        const int c = memcmp(nk + nko[x], nk + nko[y], std::min(lx, ly));
        return c ? c < 0 : lx < ly;
    };
    std::sort(order.begin(), order.end(), less);
    for (uint32_t r = 0; r < n; r++) rank[order[r]] = r;
    for (uint32_t r = 1; r < n; r++)
        if (!less(order[r - 1], order[r])) raise(CZI_E_INVALID, "nodes %u and %u have the same key", order[r - 1], order[r]);
    if (d->entry >= n) raise(CZI_E_INVALID, "entry %u of %u nodes", d->entry, n);

    // the two value shapes are constant up to the number / the hash: build them once, patch per row
    std::vector<uint8_t> link_val, self_val;
    put_prefix(link_val, rid);
    link_val.push_back(0x93);  // [dist, hash, ignore_link]
    mp_put_f64(link_val, 0.0);
    const size_t link_num_at = link_val.size() - 8;
    mp_put_str(link_val, "Null");
    mp_put_bool(link_val, false);
    const uint8_t zero_hash[32] = {0};
    put_prefix(self_val, rid);
    self_val.push_back(0x93);
    mp_put_f64(self_val, 0.0);
    const size_t self_num_at = self_val.size() - 8;
    mp_put_bytes(self_val, zero_hash, 32);
    const size_t self_hash_at = self_val.size() - 32;
    mp_put_bool(self_val, false);
    auto patch_f64 = [](uint8_t *at, double f) {
        uint64_t u;
        memcpy(&u, &f, 8);
        u = __builtin_bswap64(u);
        memcpy(at, &u, 8);
    };

    // pass 1: validate, and lay the output out exactly.  An item = one node on one level (its self-loop row + its link
    // rows); items in output order: top layer first (most negative layer = smallest key), nodes by key order.
    struct Item {
        int lv;
        uint32_t r, fr;            // row in the level's tables, node id
        uint64_t row0, key0, val0;  // where its rows start in the output
    };
    std::vector<Item> items;
    uint64_t n_rows = 0, key_bytes = 0, val_bytes = 0;
    std::vector<std::pair<uint32_t, uint32_t>> present;  // (rank of node, row in the level's tables)
    for (int lv = d->n_levels - 1; lv >= 0; lv--) {
        const uint32_t sz = d->level_size[lv], width = (uint32_t)d->level_width[lv];
        const uint32_t *ids = d->level_nodes[lv], *tab = d->level_nbrs[lv];
        if (!tab || (lv > 0 && !ids)) raise(CZI_E_INVALID, "level %d: null table", lv);
        present.clear();
        for (uint32_t r = 0; r < sz; r++) {
            const uint32_t fr = ids ? ids[r] : r;
            if (fr >= n) raise(CZI_E_INVALID, "level %d names node %u of %u", lv, fr, n);
            present.push_back({rank[fr], r});
        }
        std::sort(present.begin(), present.end());
        for (const auto &pr : present) {
            const uint32_t r = pr.second, fr = ids ? ids[r] : r;
            const uint64_t flen = nko[fr + 1] - nko[fr];
            items.push_back({lv, r, fr, n_rows, key_bytes, val_bytes});
            n_rows++;
            key_bytes += 18 + 2 * flen;
            val_bytes += self_val.size();
            for (uint32_t s = 0; s < width; s++) {
                const uint32_t t = tab[(size_t)r * width + s];
                if (t == CZ_NONE) continue;
                if (t >= n) raise(CZI_E_INVALID, "level %d: link to node %u of %u", lv, t, n);
                if (t == fr) raise(CZI_E_INVALID, "level %d: node %u links to itself", lv, fr);
                n_rows++;
                key_bytes += 18 + flen + (nko[t + 1] - nko[t]);
                val_bytes += link_val.size();
            }
        }
    }
    // the canary row: layer 1, every other key column Null; the number of those = the columns of two CompoundKeys
    size_t cols = 0;
    for (const uint8_t *p = nk + nko[d->entry], *e = nk + nko[d->entry + 1]; p < e; p = mc_skip(p, e)) cols++;
    Buf target;  // the entry's self-row key with a Null layer (hnsw.rs:641-656)
    put_prefix(target.b, rid);
    target.u8(NULL_TAG);
    for (int side = 0; side < 2; side++) target.raw(nk + nko[d->entry], (size_t)(nko[d->entry + 1] - nko[d->entry]));
    std::vector<uint8_t> canary_val;
    put_prefix(canary_val, rid);
    canary_val.push_back(0x93);
    mp_put_int(canary_val, -(int64_t)(d->n_levels - 1));
    mp_put_bytes(canary_val, target.b.data(), target.b.size());
    mp_put_bool(canary_val, false);
    const uint64_t canary_row = n_rows, canary_key = key_bytes, canary_valat = val_bytes;
    n_rows++;
    key_bytes += 18 + 2 * cols;
    val_bytes += canary_val.size();

    out.keys.resize(key_bytes);
    out.vals.resize(val_bytes);
    out.key_off.resize(n_rows + 1);
    out.val_off.resize(n_rows + 1);
    out.key_off[n_rows] = key_bytes;
    out.val_off[n_rows] = val_bytes;

    const uint32_t T = ingest_threads(n_rows);
    // SHA-256 of every vector (Vector::get_hash): threads over nodes
    std::vector<uint8_t> hashes((size_t)n * 32);
    parallel_for(T, [&](uint32_t t) {
        for (uint32_t v = (uint32_t)((uint64_t)n * t / T); v < (uint32_t)((uint64_t)n * (t + 1) / T); v++) {
            Sha256 sha;
            sha.update((const uint8_t *)(vectors + (size_t)v * d->dim), (size_t)d->dim * 4);  // host is little-endian
            sha.finish(hashes.data() + (size_t)v * 32);
        }
    });
    std::vector<std::array<uint8_t, 10>> layer_keys(d->n_levels);
    for (int lv = 0; lv < d->n_levels; lv++) {
        Buf k;
        k.num_int(-(int64_t)lv);  // |layer| < 2^53: 10 bytes
        memcpy(layer_keys[lv].data(), k.b.data(), 10);
    }
    // pass 2: write (threads over items; every item knows where its rows go)
    parallel_for(T, [&](uint32_t t) {
        struct Link {
            uint32_t rank, to;
            double dist;
        };
        std::vector<Link> links;
        uint8_t head[18];  // relation id + the layer column
        for (int i = 0; i < 8; i++) head[i] = (uint8_t)(rid >> (56 - 8 * i));
        for (size_t it = items.size() * t / T; it < items.size() * (t + 1) / T; it++) {
            const Item &item = items[it];
            const uint32_t width = (uint32_t)d->level_width[item.lv], fr = item.fr;
            const uint32_t *tab = d->level_nbrs[item.lv];
            const double *dist = level_dist ? level_dist[item.lv] : nullptr;
            memcpy(head + 8, layer_keys[item.lv].data(), 10);
            uint8_t *kw = out.keys.data() + item.key0, *vw = out.vals.data() + item.val0;
            uint64_t row = item.row0;
            auto begin_row = [&] {
                out.key_off[row] = (uint64_t)(kw - out.keys.data());
                out.val_off[row] = (uint64_t)(vw - out.vals.data());
                row++;
            };
            auto put_key = [&](uint32_t to) {
                memcpy(kw, head, 18);
                kw += 18;
                memcpy(kw, nk + nko[fr], nko[fr + 1] - nko[fr]);
                kw += nko[fr + 1] - nko[fr];
                memcpy(kw, nk + nko[to], nko[to + 1] - nko[to]);
                kw += nko[to + 1] - nko[to];
            };
            links.clear();
            for (uint32_t s = 0; s < width; s++) {
                const uint32_t to = tab[(size_t)item.r * width + s];
                if (to != CZ_NONE) links.push_back({rank[to], to, dist ? dist[(size_t)item.r * width + s] : 0.0});
            }
            std::sort(links.begin(), links.end(), [](const Link &a, const Link &b) { return a.rank < b.rank; });
            bool self_done = false;
            auto put_self = [&] {
                begin_row();
                put_key(fr);
                memcpy(vw, self_val.data(), self_val.size());
                patch_f64(vw + self_num_at, (double)links.size());
                memcpy(vw + self_hash_at, hashes.data() + (size_t)fr * 32, 32);
                vw += self_val.size();
                self_done = true;
            };
            for (const Link &l : links) {
                if (!self_done && rank[fr] < l.rank) put_self();
                begin_row();
                put_key(l.to);
                memcpy(vw, link_val.data(), link_val.size());
                patch_f64(vw + link_num_at, l.dist);
                vw += link_val.size();
            }
            if (!self_done) put_self();
        }
    });
    {
        uint8_t *kw = out.keys.data() + canary_key;
        for (int i = 0; i < 8; i++) kw[i] = (uint8_t)(rid >> (56 - 8 * i));
        Buf k;
        k.num_int(1);
        memcpy(kw + 8, k.b.data(), 10);
        memset(kw + 18, NULL_TAG, 2 * cols);
        memcpy(out.vals.data() + canary_valat, canary_val.data(), canary_val.size());
        out.key_off[canary_row] = canary_key;
        out.val_off[canary_row] = canary_valat;
    }
}

}  // namespace

extern "C" int czi_hnsw_encode_rows(const cz_hnsw_desc *desc, const float *vectors, const uint8_t *node_keys,
                                    const uint64_t *node_key_off, const double *const *level_dist, uint64_t relation_id,
                                    czi_row_buf **out) {
    if (!out) return fail(CZI_E_INVALID, "null out");
    *out = nullptr;
    if (!desc || (desc->n && (!vectors || !node_keys || !node_key_off))) return fail(CZI_E_INVALID, "null argument");
    std::unique_ptr<czi_row_buf> b(new (std::nothrow) czi_row_buf);
    if (!b) return fail(CZI_E_OOM, "out of host memory");
    const int rc = guarded([&] { encode_index_rows(desc, vectors, node_keys, node_key_off, level_dist, relation_id, *b); });
    if (rc) return rc;
    *out = b.release();
    return CZI_OK;
}

extern "C" int czi_row_buf_rows(const czi_row_buf *b, czi_rows *rows) {
    if (!b || !rows) return fail(CZI_E_INVALID, "null argument");
    rows->keys = b->keys.data();
    rows->key_off = b->key_off.data();
    rows->vals = b->vals.data();
    rows->val_off = b->val_off.data();
    rows->n_rows = b->key_off.size() - 1;
    rows->n_key_cols = 0;
    return CZI_OK;
}

extern "C" void czi_row_buf_free(czi_row_buf *b) { delete b; }
