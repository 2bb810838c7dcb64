// distance_f64.h -- VectorCache::dist, the F64 arms (cozo-core/src/runtime/hnsw.rs:73-78, 86-95, 102-106): every dot product
// and the final 1 - x, /, sqrt in f64.  The f32 kernels' tree, element type changed: a vector is cut into 16-byte chunks (TWO
// doubles), LPV lanes (16 / 32 / 64: the smallest power of two >= the chunk count) own chunks lane, lane + LPV, ...; a lane runs
// one explicit fma chain over its elements in address order; the lanes are combined by an xor butterfly with offsets
// LPV/2 ... 1 (oracle/cozo_oracle.c orc_dot_gpu_f64 restates exactly this).  Zero padding participates.  8 * dim bytes per
// evaluation: HBM-bound like the f32 form at twice the bytes; f64 indices are rare, so this is the plain form of the kernel
// (shuffles through the LDS crossbar, the query read from LDS) rather than a second hand-tuned one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cozo_gpu.h"

namespace czd64 {

__host__ __device__ inline int lpv_for(uint32_t dim) {
    const uint32_t chunks = (dim + 1) / 2;
    int lpv = 16;
    while ((uint32_t)lpv < chunks && lpv < 64) lpv <<= 1;
    return lpv;
}

template <int LPV>
__device__ __forceinline__ double group_reduce(double v) {
#pragma unroll
    for (int off = LPV / 2; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void acc_chunk(int metric, const double2 &q, const double2 &v, double &a0, double &a1) {
    if (metric == CZ_L2) {
        double d;
        d = q.x - v.x; a0 = fma(d, d, a0);
        d = q.y - v.y; a0 = fma(d, d, a0);
    } else if (metric == CZ_COSINE) {
        a0 = fma(q.x, v.x, a0); a1 = fma(v.x, v.x, a1);
        a0 = fma(q.y, v.y, a0); a1 = fma(v.y, v.y, a1);
    } else {
        a0 = fma(q.x, v.x, a0);
        a0 = fma(q.y, v.y, a0);
    }
}

__device__ __forceinline__ double finish_distance(int metric, double acc_main, double acc_bn, double qnorm) {
    if (metric == CZ_L2) return acc_main;
    if (metric == CZ_COSINE) return 1.0 - acc_main / sqrt(qnorm * acc_bn);
    return 1.0 - acc_main;
}

// q . q with the kernel's tree (the query's norm of the cosine distance), query chunks in `q` (LDS or global)
template <int LPV>
__device__ __forceinline__ double self_dot(const double2 *q, int glane, int chunks) {
    double a = 0.0;
    for (int c = glane; c < chunks; c += LPV) {
        const double2 x = q[c];
        a = fma(x.x, x.x, a);
        a = fma(x.y, x.y, a);
    }
    return group_reduce<LPV>(a);
}

// distances from the query (chunks in q) to U rows, all lanes of the group get the results; a null row gives garbage the caller drops
template <int LPV, int U>
__device__ __forceinline__ void group_distances(int metric, const double2 *q, int glane, int chunks, double qnorm,
                                                const double2 *(&rows)[U], double (&out)[U]) {
    double a0[U], a1[U];
#pragma unroll
    for (int u = 0; u < U; u++) a0[u] = a1[u] = 0.0;
    for (int c = glane; c < chunks; c += LPV) {
        const double2 qq = q[c];
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = rows[u] != nullptr ? rows[u][c] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < U; u++) acc_chunk(metric, qq, v[u], a0[u], a1[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        a0[u] = group_reduce<LPV>(a0[u]);
        if (metric == CZ_COSINE) a1[u] = group_reduce<LPV>(a1[u]);
        out[u] = finish_distance(metric, a0[u], a1[u], qnorm);
    }
}

}  // namespace czd64
