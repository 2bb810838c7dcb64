// distance.h -- wave64 distance arithmetic for gfx950 (device code only).
//
// Arithmetic contract (VectorCache::dist, cozo-core/src/runtime/hnsw.rs:66-109, F32 arms):
//   L2     = dot(a-b, a-b)           f32, widened to f64 at the end      (squared, no sqrt)
//   Cosine = 1 - dot(a,b)/sqrt(dot(a,a)*dot(b,b))   each dot f32, widened BEFORE the 1-, /, sqrt
//   IP     = 1 - dot(a,b)            dot f32, widened before the 1-
//
// Summation tree (restated bit-for-bit by oracle/cozo_oracle.c orc_dot_gpu / orc_l2_gpu):
//   a row is cut into 16-byte chunks; a GROUP of LPV lanes (16/32/64 = smallest power of two >=
//   #chunks, capped at 64) owns one vector; lane g of the group owns chunks g, g+LPV, g+2*LPV ...
//   and runs ONE fma chain over its elements in address order; the group is then combined by an
//   xor butterfly with offsets LPV/2 ... 1.  Rows are zero-padded to a multiple of 4 floats in HBM.
//
// Memory: one chunk-load is a fully coalesced global_load_dwordx4 (16 B/lane, 1 KiB per wave
// instruction at LPV = 64); U vectors per group are kept in flight to cover HBM latency.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace czd {

constexpr int kWave = 64;

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// v[lane] + v[lane ^ OFF] without touching the LDS crossbar (ds_bpermute costs an LDS round trip per step and
// serialises the butterfly: measured, the 12 dependent round trips per cosine row were most of a wave's time
// between two row fetches).  Additions are commutative, so every lane gets exactly the value of the textbook
// `v + __shfl_xor(v, OFF)` butterfly; scratch/dpp_butterfly_test.hip checks each step bit for bit on the device.
//   32, 16: gfx950 v_permlane32_swap / v_permlane16_swap (halves / odd-even 16-lane rows exchanged between two copies)
//   8:      DPP row_ror:8 (a rotation by half a 16-lane row is the xor)
//   4:      two bank-masked DPP moves (row_shl:4 into banks 0,2; row_shr:4 into banks 1,3)
//   2, 1:   DPP quad_perm
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(v), CTRL, 0xF, BANK, false));
}
template <int OFF>
__device__ __forceinline__ float xor_add(float v) {
    static_assert(OFF == 32 || OFF == 16 || OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "butterfly offset");
    if constexpr (OFF == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (OFF == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (OFF == 8) {
        return v + dpp_mov<0x128, 0xF>(v, v);
    } else if constexpr (OFF == 4) {
        float t = dpp_mov<0x104, 0x5>(v, v);
        t = dpp_mov<0x114, 0xA>(t, v);
        return v + t;
    } else if constexpr (OFF == 2) {
        return v + dpp_mov<0x4E, 0xF>(v, v);
    } else {
        return v + dpp_mov<0xB1, 0xF>(v, v);
    }
}

// xor butterfly over a group of LPV lanes (offsets LPV/2 ... 1) for K independent values, step by step so that
// the K chains interleave
template <int LPV, int K>
__device__ __forceinline__ void group_reduce_many(float (&v)[K]) {
    if constexpr (LPV >= 64) {
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = xor_add<32>(v[k]);
    }
    if constexpr (LPV >= 32) {
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = xor_add<16>(v[k]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = xor_add<8>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = xor_add<4>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = xor_add<2>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = xor_add<1>(v[k]);
}

template <int LPV>
__device__ __forceinline__ float group_reduce(float v) {
    float a[1] = {v};
    group_reduce_many<LPV, 1>(a);
    return a[0];
}

// order-preserving u64 key of an f64 distance; NaN sorts greatest (ordered-float semantics)
__device__ __forceinline__ uint64_t dist_key(double d) {
    if (d != d) return ~0ull;
    uint64_t b = (uint64_t)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_dist(uint64_t k) {
    if (k == ~0ull) return __longlong_as_double(0x7FF8000000000000ll);
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double finish_distance(int metric, float acc_main, float acc_bn, float qnorm) {
    if (metric == CZ_L2) return (double)acc_main;
    if (metric == CZ_COSINE) return 1.0 - (double)acc_main / sqrt((double)qnorm * (double)acc_bn);
    return 1.0 - (double)acc_main;
}

__device__ __forceinline__ float4 ld_chunk(const float4 *row, int c, int chunks) {
    return c < chunks ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// accumulate one chunk of one vector (METRIC is a compile-time cz_metric)
template <int METRIC>
__device__ __forceinline__ void acc_chunk_m(const float4 &q, const float4 &v, float &a0, float &a1) {
    if constexpr (METRIC == CZ_L2) {
        float d;
        d = q.x - v.x; a0 = fma_(d, d, a0);
        d = q.y - v.y; a0 = fma_(d, d, a0);
        d = q.z - v.z; a0 = fma_(d, d, a0);
        d = q.w - v.w; a0 = fma_(d, d, a0);
    } else if constexpr (METRIC == CZ_COSINE) {
        a0 = fma_(q.x, v.x, a0); a1 = fma_(v.x, v.x, a1);
        a0 = fma_(q.y, v.y, a0); a1 = fma_(v.y, v.y, a1);
        a0 = fma_(q.z, v.z, a0); a1 = fma_(v.z, v.z, a1);
        a0 = fma_(q.w, v.w, a0); a1 = fma_(v.w, v.w, a1);
    } else {
        a0 = fma_(q.x, v.x, a0);
        a0 = fma_(q.y, v.y, a0);
        a0 = fma_(q.z, v.z, a0);
        a0 = fma_(q.w, v.w, a0);
    }
}
__device__ __forceinline__ void acc_chunk(int metric, const float4 &q, const float4 &v, float &a0, float &a1) {
    if (metric == CZ_L2) acc_chunk_m<CZ_L2>(q, v, a0, a1);
    else if (metric == CZ_COSINE) acc_chunk_m<CZ_COSINE>(q, v, a0, a1);
    else acc_chunk_m<CZ_IP>(q, v, a0, a1);
}

// Register-resident batch of U base rows of one lane group (ITERS chunks per lane each): the unit the traversal
// kernels keep in flight.  `full` (uniform) = the row has exactly LPV * ITERS chunks, so no lane is ever out of range
// and the loads carry no predicate.
template <int ITERS, int U>
struct RowRegs {
    float4 v[U][ITERS];
};
// one 16-byte chunk of a base row.  NT = non-temporal hint (global_load_dwordx4 ... nt): for a pure stream of
// rows that are read once it keeps them from evicting what IS reused out of the 4 MiB L2 -- the pairs kernel's
// query rows (measured 4.95 -> 5.28 TB/s) and the search kernel's link rows / visited words (3.45 -> 3.27 ms per
// 1024 queries).  Index construction does not use it: its selection heuristic re-reads rows through L2 / Infinity
// Cache (NT cost it 40 %).
template <bool NT>
__device__ __forceinline__ float4 ld_row_chunk(const float4 *p) {
    if constexpr (NT) {
        typedef float f4_ __attribute__((ext_vector_type(4)));
        const f4_ t = __builtin_nontemporal_load((const f4_ *)p);
        return make_float4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}
template <int LPV, int ITERS, int U, bool NT = false>
__device__ __forceinline__ void load_rows(RowRegs<ITERS, U> &r, const float4 *const (&rows)[U], int glane, int chunks,
                                          bool full) {
    if (full) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < ITERS; j++) r.v[u][j] = ld_row_chunk<NT>(rows[u] + glane + LPV * j);
    } else {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < ITERS; j++) {
                const int c = glane + LPV * j;
                r.v[u][j] = c < chunks ? ld_row_chunk<NT>(rows[u] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
}
// raw accumulators of the U rows against the register-resident query, reduced over the lane group:
// main[u] = dot / squared distance, bn[u] = self dot of the row (cosine only)
template <int METRIC, int LPV, int ITERS, int U>
__device__ __forceinline__ void dot_rows(const float4 (&q)[ITERS], const RowRegs<ITERS, U> &r, float (&main)[U],
                                         float (&bn)[U]) {
    if constexpr (METRIC == CZ_COSINE) {
        float acc[2 * U];
#pragma unroll
        for (int u = 0; u < 2 * U; u++) acc[u] = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < ITERS; j++) acc_chunk_m<METRIC>(q[j], r.v[u][j], acc[u], acc[U + u]);
        group_reduce_many<LPV, 2 * U>(acc);
#pragma unroll
        for (int u = 0; u < U; u++) {
            main[u] = acc[u];
            bn[u] = acc[U + u];
        }
    } else {
        float dummy = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++) main[u] = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < ITERS; j++) acc_chunk_m<METRIC>(q[j], r.v[u][j], main[u], dummy);
        group_reduce_many<LPV, U>(main);
#pragma unroll
        for (int u = 0; u < U; u++) bn[u] = 0.f;
    }
}

// self dot of the (register / LDS resident) query: the `a_norm` of the cosine arm
template <int LPV, int ITERS>
__device__ __forceinline__ float query_norm(const float4 (&q)[ITERS > 0 ? ITERS : 1], const float4 *q_lds, int glane,
                                            int chunks) {
    float a = 0.f;
    if constexpr (ITERS > 0) {
#pragma unroll
        for (int j = 0; j < ITERS; j++) {
            a = fma_(q[j].x, q[j].x, a);
            a = fma_(q[j].y, q[j].y, a);
            a = fma_(q[j].z, q[j].z, a);
            a = fma_(q[j].w, q[j].w, a);
        }
    } else {
        for (int c = glane; c < chunks; c += LPV) {
            float4 x = q_lds[c];
            a = fma_(x.x, x.x, a);
            a = fma_(x.y, x.y, a);
            a = fma_(x.z, x.z, a);
            a = fma_(x.w, x.w, a);
        }
    }
    return group_reduce<LPV>(a);
}

// Distances from the query to U base rows handled by one lane group.
//   ITERS > 0: query chunks in registers q[ITERS]; ITERS == 0: generic dims, query read from LDS.
//   rows[u] = pointer to the base row (nullptr => slot unused; result undefined).
template <int LPV, int ITERS, int U>
__device__ __forceinline__ void group_distances(int metric, const float4 (&q)[ITERS > 0 ? ITERS : 1],
                                                const float4 *q_lds, int glane, int chunks, float qnorm,
                                                const float4 *(&rows)[U], double (&out)[U]) {
    float a0[U], a1[U];
#pragma unroll
    for (int u = 0; u < U; u++) a0[u] = a1[u] = 0.f;
    if constexpr (ITERS > 0) {
        float4 v[U][ITERS];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < ITERS; j++) {
                int c = glane + LPV * j;
                v[u][j] = (rows[u] != nullptr && c < chunks) ? rows[u][c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < ITERS; j++) acc_chunk(metric, q[j], v[u][j], a0[u], a1[u]);
        }
    } else {
        for (int c = glane; c < chunks; c += LPV) {
            float4 qq = q_lds[c];
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = rows[u] != nullptr ? rows[u][c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; u++) acc_chunk(metric, qq, v[u], a0[u], a1[u]);
        }
    }
    group_reduce_many<LPV, U>(a0);
    if (metric == CZ_COSINE) group_reduce_many<LPV, U>(a1);
#pragma unroll
    for (int u = 0; u < U; u++) out[u] = finish_distance(metric, a0[u], a1[u], qnorm);
}

// compile-time shape of a dimension: lanes per vector and chunk iterations per lane
struct Shape {
    int lpv, iters, chunks;
};
inline Shape shape_of(uint32_t dim) {
    int chunks = (int)((dim + 3) / 4);
    int lpv = 16;
    while (lpv < chunks && lpv < 64) lpv <<= 1;
    int iters = (chunks + lpv - 1) / lpv;
    return {lpv, iters, chunks};
}

}  // namespace czd
