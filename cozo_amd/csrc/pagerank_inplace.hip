// pagerank_inplace.hip -- PageRank under the OTHER reading of graph 0.3.1's loop (VERDICT r3 weak #1b).
//
// fixed_rule/algos/pagerank.rs:47-50 calls graph::page_rank, a crate whose source is not in the reference tree.  Its loop either
// refreshes every node's contribution in a pass of its own after the sweep (Jacobi: csrc/pagerank.hip, oracle orc_pagerank) or
// writes `out_scores[u] = new_score / out_degree(u)` INSIDE the per-node loop, so that nodes later in the same sweep already pull
// the updated value (oracle orc_pagerank_mode(ORC_PR_INPLACE)).  After the default 10 sweeps the two differ by ~1e-2 relative:
// far outside north_star's 1e-5, so until a box with cargo runs oracle/ref_fixtures (tests/test_ref_fixtures.py::check_pagerank
// decides between the readings) the device offers BOTH.  Under the in-place reading the reference is deterministic on one rayon
// thread only -- an ascending Gauss-Seidel sweep -- and THAT execution is what this file reproduces, bit for bit.
//
// A Gauss-Seidel sweep in ascending node order is not a parallel sweep, but its dependences are sparse: node u needs the NEW
// contribution of its in-neighbours v < u and the OLD one of those with v >= u.  So
//     level(u) = 0 if u has no in-neighbour below itself, else 1 + max level(v) over in-neighbours v < u,
// and all nodes of one level are independent: a sweep is one launch per level (a few dozen on the 10M / 100M graphs), every node
// reading `new[v]` for v < u and `old[v]` for v >= u -- the in-lists ascend, so a row is a "new" prefix and an "old" suffix, and the
// choice rides in the top bit of the stored source id.  Two contribution arrays take turns as old / new (every node is written
// once per sweep).  Rows are laid out level by level at set-up so that a level's rows are contiguous and their ids stream coalesced;
// the row sum is the reference's sequential f32 sum (one lane per row through an LDS tile; rows of >= 1 024 terms by a wave through
// exact_sum.h, tile after tile), the epilogue the same two roundings as pagerank.hip.
//
// Set-up (levels by relaxation to the fixed point on the device, the level-major layout by a counting sort on the host) is paid per
// call: this is the parity path of a reading that may turn out not to be the reference's, not the tuned one.  Roofline: gather-bound
// like pr_step_kernel (one random 4-byte read per edge), ~6x the blocked Jacobi sweep's time.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"
#include "exact_sum.h"

namespace {

constexpr int kT = 256;
constexpr uint32_t kTile = 8192;       // f32 values per LDS tile (32 KiB)
constexpr uint32_t kLongRow = 1024;    // rows of at least this many terms: one workgroup each, summed by a wave
constexpr uint32_t kOldBit = 0x80000000u;

struct Block {
    uint32_t row0, row1;  // rows [row0, row1) of the level-major layout
    uint32_t e0, e1;      // their edge slots
};

inline int grid_for(uint64_t n, int per_block = kT) {
    return (int)std::max<uint64_t>(1, std::min<uint64_t>((n + per_block - 1) / per_block, 256 * 32));
}

// what the level schedule relies on: every in-list ascends (strictly or with repeats: parallel edges are kept) and every source is a
// node.  bad[0] counts lists out of order, bad[1] sources out of range (an unsorted list gave wrong levels -- a sweep
// then read a contribution of the same or a later level, a race; an id out of range was read out of bounds)
__global__ void __launch_bounds__(kT) validate_in_lists_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ src, uint32_t N,
                                                               uint64_t E, uint32_t *__restrict__ bad) {
    for (uint64_t e = (uint64_t)blockIdx.x * kT + threadIdx.x; e < E; e += (uint64_t)gridDim.x * kT) {
        const uint32_t v = src[e];
        if (v >= N) atomicAdd(&bad[1], 1u);
    }
    for (uint64_t u = (uint64_t)blockIdx.x * kT + threadIdx.x; u < N; u += (uint64_t)gridDim.x * kT) {
        const uint32_t a = off[u], z = off[u + 1];
        if (z < a || z > E) {
            atomicAdd(&bad[0], 1u);
            continue;
        }
        for (uint32_t e = a + 1; e < z; e++)
            if (src[e] < src[e - 1]) {
                atomicAdd(&bad[0], 1u);
                break;
            }
    }
}

// level(u) = 1 + max level(v) over in-neighbours v < u: relaxed until nothing moves (levels only grow; reading a value another
// workgroup has just raised only gets there sooner).  A 16-lane group per node; the in-list ascends, so the lanes stop at u.
__global__ void __launch_bounds__(kT) level_relax_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ src, uint32_t N,
                                                         uint32_t *__restrict__ level, uint32_t *__restrict__ changed) {
    const uint32_t gl = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * kT + threadIdx.x) >> 4, ngroups = ((uint64_t)gridDim.x * kT) >> 4;
    bool any = false;
    for (uint64_t u = group; u < N; u += ngroups) {
        const uint32_t e1 = off[u + 1];
        uint32_t best = 0;
        for (uint32_t e = off[u] + gl; e < e1; e += 16) {
            const uint32_t v = src[e];
            if (v >= (uint32_t)u) break;  // ascending: everything from here on is an "old" read
            best = max(best, __hip_atomic_load(&level[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u);
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 16));
        if (gl == 0 && best > level[u]) {
            __hip_atomic_store(&level[u], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            any = true;
        }
    }
    if (any) *changed = 1u;
}

// rows into the level-major layout: row i of the new layout is node order[i]; a source below the node reads `new`, the rest `old`
__global__ void __launch_bounds__(kT) relayout_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ src,
                                                      const uint32_t *__restrict__ order, const uint32_t *__restrict__ off2, uint32_t N,
                                                      uint32_t *__restrict__ src2) {
    const uint32_t gl = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * kT + threadIdx.x) >> 4, ngroups = ((uint64_t)gridDim.x * kT) >> 4;
    for (uint64_t i = group; i < N; i += ngroups) {
        const uint32_t u = order[i], a = off[u], n = off[u + 1] - a, b = off2[i];
        for (uint32_t k = gl; k < n; k += 16) {
            const uint32_t v = src[a + k];
            src2[b + k] = v | (v >= u ? kOldBit : 0u);
        }
    }
}

__global__ void __launch_bounds__(kT) init_kernel(const uint32_t *__restrict__ out_deg, uint32_t N, float init, float *__restrict__ scores,
                                                  float *__restrict__ contrib) {
    for (uint32_t v = blockIdx.x * kT + threadIdx.x; v < N; v += gridDim.x * kT) {
        scores[v] = init;
        contrib[v] = init / (float)out_deg[v];
    }
}

__device__ __forceinline__ double block_sum(double x, double *red) {  // fixed order: lanes by xor butterfly, waves 0..3
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ double finish(float s, uint32_t u, const uint32_t *__restrict__ out_deg, float *__restrict__ scores,
                                         float *__restrict__ cnew, float base, float damping, int f64_diff) {
    const float old = scores[u];
    const float nw = base + damping * s;  // two roundings (-ffp-contract=off), like the reference
    scores[u] = nw;
    cnew[u] = nw / (float)out_deg[u];     // written inside the sweep: the reading this file exists for
    return f64_diff ? fabs((double)nw - (double)old) : fabs((double)(nw - old));
}

// the short rows of one level: a block's ids are read coalesced, the contributions gathered into an LDS tile, one lane per row adds
// its stretch in order
__global__ void __launch_bounds__(kT) gs_rows_kernel(const Block *__restrict__ blocks, const uint32_t *__restrict__ off2,
                                                     const uint32_t *__restrict__ src2, const uint32_t *__restrict__ order,
                                                     const uint32_t *__restrict__ out_deg, const float *__restrict__ cold,
                                                     float *__restrict__ cnew, float *__restrict__ scores, float base, float damping,
                                                     int f64_diff, double *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float tile[kTile];
    __shared__ double red[kT / 64];
    const Block b = blocks[blockIdx.x];
    const uint32_t nnz = b.e1 - b.e0;
    for (uint32_t i = threadIdx.x; i < nnz; i += kT) {
        const uint32_t v = src2[b.e0 + i];
        tile[i] = (v & kOldBit) ? cold[v & ~kOldBit] : cnew[v];
    }
    __syncthreads();
    double err = 0.0;
    const uint32_t r = b.row0 + threadIdx.x;
    if (r < b.row1) {
        const uint32_t a = off2[r] - b.e0, z = off2[r + 1] - b.e0;
        float s = 0.0f;
        for (uint32_t e = a; e < z; e++) s = s + tile[e];
        err = finish(s, order[r], out_deg, scores, cnew, base, damping, f64_diff);
    }
    const double tot = block_sum(err, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// the long rows of one level, one workgroup each: all threads gather a tile, wave 0 adds it to the running sum with the wave form
// of the sequential f32 sum (exact_sum.h), tile after tile
__global__ void __launch_bounds__(kT) gs_long_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ off2,
                                                     const uint32_t *__restrict__ src2, const uint32_t *__restrict__ order,
                                                     const uint32_t *__restrict__ out_deg, const float *__restrict__ cold,
                                                     float *__restrict__ cnew, float *__restrict__ scores, float base, float damping,
                                                     int f64_diff, double *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float tile[kTile + 4];
    const uint32_t r = rows[blockIdx.x];
    const uint32_t e0 = off2[r], e1 = off2[r + 1];
    float s = 0.0f;
    for (uint32_t t0 = e0; t0 < e1; t0 += kTile) {
        const uint32_t n = min(kTile, e1 - t0);
        __syncthreads();  // wave 0 is done with the previous tile
        for (uint32_t i = threadIdx.x; i < n; i += kT) {
            const uint32_t v = src2[t0 + i];
            tile[i] = (v & kOldBit) ? cold[v & ~kOldBit] : cnew[v];
        }
        __syncthreads();
        if (threadIdx.x < 64) s = cz_exact::wave_seq_sum<16>(tile, n, s);
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = finish(s, order[r], out_deg, scores, cnew, base, damping, f64_diff);
}

// the sweep's error: the partials added up in index order by ONE workgroup (a fixed tree: the same bits on every run)
__global__ void __launch_bounds__(kT) sum_partials_kernel(const double *__restrict__ partial, uint32_t n, double *__restrict__ out) {
    __shared__ double red[kT / 64];
    double x = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kT) x += partial[i];
    const double tot = block_sum(x, red);
    if (threadIdx.x == 0) *out = tot;
}

}  // namespace

extern "C" int cz_pagerank_inplace(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N, uint64_t E,
                                   float damping, double tolerance, uint32_t max_iter, uint32_t flags, float *scores, uint32_t *iters_run,
                                   double *final_err, uint32_t *n_levels, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (n_levels) *n_levels = 0;
    if (N == 0) return CZ_OK;  // pagerank.rs:43-45
    if (!in_offsets || !out_degree || !scores || (E > 0 && !in_sources)) return cz::set_error(CZ_E_INVALID, "null buffer");
    if (in_offsets[0] != 0 || in_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "offsets[0] must be 0 and offsets[N] == E");
    if (N >= kOldBit || E >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "node ids must stay below 2^31 and E below 2^32 - 1");
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "max_iter must be positive");
    const int f64_diff = (flags & CZ_PR_ERR_F64_DIFF) ? 1 : 0;
    cz::DevBuf<uint32_t> d_off, d_src, d_od, d_level, d_changed, d_order, d_off2, d_src2, d_long;
    cz::DevBuf<float> d_scores, d_ca, d_cb;
    cz::DevBuf<double> d_partial, d_err;
    cz::DevBuf<Block> d_blocks;
    CZ_HIP(d_off.alloc((size_t)N + 1));
    CZ_HIP(d_src.alloc(E));
    CZ_HIP(d_od.alloc(N));
    CZ_HIP(d_level.alloc(N));
    CZ_HIP(d_changed.alloc(1));
    CZ_HIP(hipMemcpy(d_off.p, in_offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) CZ_HIP(hipMemcpy(d_src.p, in_sources, E * 4, hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(d_od.p, out_degree, (size_t)N * 4, hipMemcpyHostToDevice));
    {   // the lists as_directed_graph builds are sorted (CsrLayout::Sorted); anything else is refused, not mis-scheduled
        cz::DevBuf<uint32_t> d_bad;
        CZ_HIP(d_bad.alloc(2));
        CZ_HIP(hipMemset(d_bad.p, 0, 8));
        hipLaunchKernelGGL(validate_in_lists_kernel, dim3(grid_for(std::max<uint64_t>(E, N))), dim3(kT), 0, nullptr, d_off.p, d_src.p, N, E, d_bad.p);
        uint32_t bad[2] = {0, 0};
        CZ_HIP(hipMemcpy(bad, d_bad.p, 8, hipMemcpyDeviceToHost));
        if (bad[1]) return cz::set_error(CZ_E_INVALID, "%u in_sources are not node ids (>= N = %u)", bad[1], N);
        if (bad[0]) return cz::set_error(CZ_E_INVALID, "%u in-lists are not in ascending order (the level schedule needs CsrLayout::Sorted rows)", bad[0]);
    }
    // ---- levels.  A chain-like graph has about N of them: every level is a launch per sweep and a round trip of the relaxation,
    // so beyond kMaxLevels the call is refused (CZ_PR_INPLACE_MAX_LEVELS overrides) rather than left to issue millions of launches
    const char *ml_env = getenv("CZ_PR_INPLACE_MAX_LEVELS");
    const uint32_t max_levels = ml_env && atoi(ml_env) > 0 ? (uint32_t)atoi(ml_env) : 4096u;
    CZ_HIP(hipMemset(d_level.p, 0, (size_t)N * 4));
    for (uint32_t round = 0;; round++) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        if (round > max_levels)
            return cz::set_error(CZ_E_UNSUPPORTED, "the graph has more than %u dependence levels (a chain-like graph): the level-scheduled sweep "
                                                   "would be one launch per level; use cz_pagerank or the CPU path", max_levels);
        CZ_HIP(hipMemsetAsync(d_changed.p, 0, 4, nullptr));
        hipLaunchKernelGGL(level_relax_kernel, dim3(grid_for((uint64_t)N * 16)), dim3(kT), 0, nullptr, d_off.p, d_src.p, N, d_level.p,
                           d_changed.p);
        uint32_t ch = 0;
        CZ_HIP(hipMemcpy(&ch, d_changed.p, 4, hipMemcpyDeviceToHost));
        if (!ch) break;
    }
    std::vector<uint32_t> level(N);
    CZ_HIP(hipMemcpy(level.data(), d_level.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    // ---- level-major layout (host: a counting sort by level, ids ascending inside a level) and the blocks of every level
    uint32_t L = 0;
    for (uint32_t u = 0; u < N; u++) L = std::max(L, level[u] + 1);
    if (L > max_levels)
        return cz::set_error(CZ_E_UNSUPPORTED, "the graph has %u dependence levels (more than %u: a chain-like graph): the level-scheduled "
                                               "sweep would be one launch per level; use cz_pagerank or the CPU path", L, max_levels);
    std::vector<uint32_t> first(L + 1, 0);
    for (uint32_t u = 0; u < N; u++) first[level[u] + 1]++;
    for (uint32_t l = 0; l < L; l++) first[l + 1] += first[l];
    std::vector<uint32_t> order(N), off2((size_t)N + 1);
    {
        std::vector<uint32_t> cur(first.begin(), first.end() - 1);
        for (uint32_t u = 0; u < N; u++) order[cur[level[u]]++] = u;
    }
    off2[0] = 0;
    for (uint32_t i = 0; i < N; i++) off2[i + 1] = off2[i] + (in_offsets[order[i] + 1] - in_offsets[order[i]]);
    std::vector<Block> blocks;
    std::vector<uint32_t> long_rows;
    std::vector<uint32_t> blk_first(L + 1, 0), long_first(L + 1, 0);
    for (uint32_t l = 0; l < L; l++) {
        blk_first[l] = (uint32_t)blocks.size();
        long_first[l] = (uint32_t)long_rows.size();
        uint32_t i = first[l];
        while (i < first[l + 1]) {
            if (off2[i + 1] - off2[i] >= kLongRow) {
                long_rows.push_back(i++);
                continue;
            }
            Block b{i, i, off2[i], off2[i]};
            while (b.row1 < first[l + 1] && b.row1 - b.row0 < (uint32_t)kT && off2[b.row1 + 1] - off2[b.row1] < kLongRow &&
                   off2[b.row1 + 1] - b.e0 <= kTile)
                b.row1++;
            b.e1 = off2[b.row1];
            blocks.push_back(b);
            i = b.row1;
        }
    }
    blk_first[L] = (uint32_t)blocks.size();
    long_first[L] = (uint32_t)long_rows.size();
    if (n_levels) *n_levels = L;
    CZ_HIP(d_order.alloc(N));
    CZ_HIP(d_off2.alloc((size_t)N + 1));
    CZ_HIP(d_src2.alloc(E));
    CZ_HIP(d_blocks.alloc(blocks.size()));
    CZ_HIP(d_long.alloc(long_rows.size()));
    CZ_HIP(hipMemcpy(d_order.p, order.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(d_off2.p, off2.data(), ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (!blocks.empty()) CZ_HIP(hipMemcpy(d_blocks.p, blocks.data(), blocks.size() * sizeof(Block), hipMemcpyHostToDevice));
    if (!long_rows.empty()) CZ_HIP(hipMemcpy(d_long.p, long_rows.data(), long_rows.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for((uint64_t)N * 16)), dim3(kT), 0, nullptr, d_off.p, d_src.p, d_order.p, d_off2.p, N, d_src2.p);
    // ---- the loop of graph::page_rank
    const uint32_t n_partial = (uint32_t)(blocks.size() + long_rows.size());
    CZ_HIP(d_scores.alloc(N));
    CZ_HIP(d_ca.alloc(N));
    CZ_HIP(d_cb.alloc(N));
    CZ_HIP(d_partial.alloc(n_partial));
    CZ_HIP(d_err.alloc(1));
    const float init = 1.0f / (float)N, base = (1.0f - damping) / (float)N;
    hipLaunchKernelGGL(init_kernel, dim3(grid_for(N)), dim3(kT), 0, nullptr, d_od.p, N, init, d_scores.p, d_ca.p);
    float *cold = d_ca.p, *cnew = d_cb.p;
    uint32_t it = 0;
    double err = 0.0;
    for (;;) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        for (uint32_t l = 0; l < L; l++) {
            const uint32_t nb = blk_first[l + 1] - blk_first[l], nl = long_first[l + 1] - long_first[l];
            if (nb)
                hipLaunchKernelGGL(gs_rows_kernel, dim3(nb), dim3(kT), 0, nullptr, d_blocks.p + blk_first[l], d_off2.p, d_src2.p, d_order.p, d_od.p,
                                   cold, cnew, d_scores.p, base, damping, f64_diff, d_partial.p + blk_first[l]);
            if (nl)
                hipLaunchKernelGGL(gs_long_kernel, dim3(nl), dim3(kT), 0, nullptr, d_long.p + long_first[l], d_off2.p, d_src2.p, d_order.p, d_od.p,
                                   cold, cnew, d_scores.p, base, damping, f64_diff, d_partial.p + blocks.size() + long_first[l]);
        }
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kT), 0, nullptr, d_partial.p, n_partial, d_err.p);
        CZ_HIP(hipMemcpy(&err, d_err.p, 8, hipMemcpyDeviceToHost));
        std::swap(cold, cnew);
        it++;
        if (err < tolerance || it == max_iter) break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank_inplace launch: %s", hipGetErrorString(e));
    CZ_HIP(hipMemcpy(scores, d_scores.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return CZ_OK;
}
