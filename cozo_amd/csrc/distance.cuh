// distance.cuh -- wave64 distance arithmetic for gfx950 (device code only).
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

template <int LPV>
__device__ __forceinline__ float group_reduce(float v) {
#pragma unroll
    for (int off = LPV / 2; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, kWave);
    return v;
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

// accumulate one chunk of one vector
__device__ __forceinline__ void acc_chunk(int metric, const float4 &q, const float4 &v, float &a0, float &a1) {
    if (metric == CZ_L2) {
        float d;
        d = q.x - v.x; a0 = fma_(d, d, a0);
        d = q.y - v.y; a0 = fma_(d, d, a0);
        d = q.z - v.z; a0 = fma_(d, d, a0);
        d = q.w - v.w; a0 = fma_(d, d, a0);
    } else if (metric == CZ_COSINE) {
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
#pragma unroll
    for (int u = 0; u < U; u++) {
        float m = group_reduce<LPV>(a0[u]);
        float bn = metric == CZ_COSINE ? group_reduce<LPV>(a1[u]) : 0.f;
        out[u] = finish_distance(metric, m, bn, qnorm);
    }
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
