// common.hpp -- shared by the translation units of libcozo_ingest.so (graph.cpp, hnsw.cpp, writeback.cpp): errors, the byte
// formats, the byte-string table, row access, threads.  Host code only.  See include/cozo_ingest.h for the contract; the
// byte formats restated here are
//   memcmp keys      data/memcmp.rs:22-163 (encode), :165-365 (decode)
//   stored key/value data/tuple.rs:27-52, runtime/relation.rs:169-296, 520-531
//   msgpack values   rmp-serde 1.2.0 over the derives of data/value.rs:143-175 (+ Vector's own impl, :226-252)
#pragma once
// the library is built with -fvisibility=hidden: only what include/cozo_ingest.h declares is exported
#pragma GCC visibility push(default)
#include "cozo_ingest.h"
#pragma GCC visibility pop

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace czi {

inline thread_local std::string g_err;  // one per thread for the whole library (czi_last_error)

inline int fail(int code, const char *fmt, ...) {
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
[[noreturn]] inline void raise(int code, const char *fmt, ...) {
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
inline const uint8_t *mc_skip_groups(const uint8_t *p, const uint8_t *end) {
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
inline const uint8_t *mc_skip(const uint8_t *p, const uint8_t *end, int depth = 0) {
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
inline NumVal mc_num(const uint8_t *p, const uint8_t *end) {
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
            if ((size_t)(end - p) < k) raise(CZI_E_CORRUPT, "truncated msgpack array");  // every element takes >= 1 byte
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
inline const char *const kVariantName[V_COUNT] = {"Null", "Bool", "Num", "Str", "Bytes", "Uuid", "Regex", "List", "Set", "Vec", "Json",
                                           "Validity", "Bot"};

inline int variant_of(Mp &m, const char *const *names, int count, const char *what) {
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
inline Variant mp_value_head(Mp &m) {
    if (m.is_map()) {
        if (m.map() != 1) raise(CZI_E_CORRUPT, "msgpack: a value must be a one-entry map");
        return (Variant)variant_of(m, kVariantName, V_COUNT, "DataValue");
    }
    const Variant v = (Variant)variant_of(m, kVariantName, V_COUNT, "DataValue");
    if (v != V_NULL && v != V_BOT) raise(CZI_E_CORRUPT, "msgpack: variant %s needs a payload", kVariantName[v]);
    return v;
}

inline NumVal mp_num(Mp &m) {  // enum Num { Int(i64), Float(f64) }, data/value.rs:493-499
    static const char *const names[2] = {"Int", "Float"};
    if (m.map() != 1) raise(CZI_E_CORRUPT, "msgpack: a number must be a one-entry map");
    if (variant_of(m, names, 2, "Num") == 0) {
        const int64_t i = m.integer();
        return {true, i, (double)i};
    }
    return {false, 0, m.real()};
}

// Vector (data/value.rs:226-252): tuple (0u8 | 1u8, bytes of the elements in NATIVE = little-endian order)
inline void mp_vec(Mp &m, int &el, const uint8_t *&bytes, uint32_t &n, std::vector<uint8_t> &scratch) {
    if (m.array() != 2) raise(CZI_E_CORRUPT, "msgpack: a vector must be a 2-tuple");
    el = (int)m.integer();
    if (el != 0 && el != 1) raise(CZI_E_CORRUPT, "msgpack: bad vector element type %d", el);
    m.bin(bytes, n, scratch);
    if (n % (el == 0 ? 4 : 8)) raise(CZI_E_CORRUPT, "msgpack: vector payload of %u bytes", n);
}

// one msgpack DataValue -> its memcmp encoding appended to `out` (so that a node value stored in the value part of a
// row gets the same identity as the same value stored in a key column)
inline void mp_to_memcmp(Mp &m, Buf &out, std::vector<uint8_t> &scratch, int depth = 0) {
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
inline uint64_t hash_bytes(const uint8_t *p, size_t n) {
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
inline Row row_at(const czi_rows *r, uint64_t i) {
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
inline void check_rows(const czi_rows *r, const char *what) {
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

inline uint32_t ingest_threads(uint64_t rows) {
    const char *env = getenv("CZI_THREADS");
    long t = env ? atol(env) : (long)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const char *min_env = getenv("CZI_THREADED_MIN_ROWS");
    const uint64_t min_rows = min_env ? (uint64_t)atoll(min_env) : (1ull << 18);
    if (t < 2 || rows < min_rows) return 1;
    return (uint32_t)std::min(t, 64l);
}


}  // namespace czi
