// pagerank_inplace.hip -- PageRank under the OTHER reading of graph 0.3.1's loop, resident and level-scheduled.
//
// fixed_rule/algos/pagerank.rs:47-50 calls graph::page_rank, a crate whose source is not in the reference tree.  Its loop either
// refreshes every node's contribution in a pass of its own after the sweep (Jacobi: csrc/pagerank.hip, oracle orc_pagerank) or
// writes `out_scores[u] = new_score / out_degree(u)` INSIDE the per-node loop, so that nodes later in the same sweep already pull
// the updated value (oracle orc_pagerank_mode(ORC_PR_INPLACE)).  After the default 10 sweeps the two differ by ~1e-2 relative, far
// outside north_star's 1e-5, and nothing in the reference tree decides between them -- so the device offers BOTH with equal
// standing: a resident plan, a sweep launched without host round trips, event-timed in bench.py.  Under the in-place reading the
// reference is deterministic on one rayon thread only -- an ascending Gauss-Seidel sweep -- and THAT execution is what this file
// reproduces, bit for bit.
//
// The sweep (layout and its reasons: csrc/inplace_plan.hpp).  Nodes are numbered level-major; ONE launch per level, gi_level_kernel,
// whose workgroups play one of two roles:
//   phase B  a row block of the level: its stream elements -- X values of this sweep, Y values of the previous one -- go from HBM
//            into their CSR places of an LDS tile (runs of consecutive stream positions, one per slice and class), the few "urgent"
//            ones are gathered from the contribution vector the launch before wrote, then a lane per row adds its stretch in order
//            (rows of >= 192 terms: a wave, csrc/exact_sum.h) and the epilogue writes score, |delta| and the new contribution;
//   phase A  a work item of the level `urgent_gap` below: a slice of that level's new contributions staged in LDS, the values of its
//            out-edges written as a coalesced stream -- X for the levels above in this sweep, Y for the next sweep (two Y streams
//            take turns).  Nothing the same launch's phase B reads: it fills the chip beside it, without a second stream.
// gi_long_kernel: rows longer than a tile, one workgroup each, gathered tile by tile (hubs of a skewed graph).
// The whole sweep -- L + 2 launches in one chain -- is one hipGraph per sweep parity, replayed.
// Every row's sum is the reference's sequential f32 sum in ascending source order; the epilogue the same two roundings as
// pagerank.hip (base + damping * s with -ffp-contract=off).
//
// Roofline model: SURVEY 8d's compulsory bytes (6.4 B/edge at 10M / 100M) like the Jacobi sweep; what this formulation moves is
// 16 B per streamed edge (2 + 4 in phase A, 4 + 2 + 4 in phase B) + 6 B per urgent edge + 24 B per node.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

#include "common.h"
#include "exact_sum.h"
#include "inplace_plan.hpp"

namespace {

using czgs::Block;
using czgs::Item;

constexpr int kT = 256;                // long rows / helpers
constexpr int kLT = 1024;              // the level kernel: phase B row blocks and phase A items in one launch
constexpr uint32_t kTileDefault = 16384;   // f32 values of a phase-B tile (64 KiB of LDS)
constexpr uint32_t kSliceDefault = 16384;  // nodes of a phase-A slice (64 KiB of LDS)
constexpr uint32_t kPartDefault = 16384;   // stream positions of a phase-A work item
constexpr uint32_t kRowsPerLane = 2;       // rows of a row block per lane of its workgroup
constexpr uint32_t kLongTile = 8192;   // long rows: values gathered per round
constexpr uint32_t kWaveRow = 192;     // rows of at least this many terms are added by a wave
constexpr uint32_t kMaxWaveRows = 32768 / kWaveRow + 2;
#ifndef CZ_GI_F
#define CZ_GI_F 4  // stream groups (four values each) in flight per lane of a row block: a 16384-value tile at once
#endif
constexpr uint32_t kOldBit = czgs::kOldBit;
constexpr uint32_t kYBit = czgs::kYBit;

inline int grid_for(uint64_t n, int per_block = kT) {
    return (int)std::max<uint64_t>(1, std::min<uint64_t>((n + per_block - 1) / per_block, 256 * 32));
}

__global__ void __launch_bounds__(kT) gi_init_kernel(const uint32_t *__restrict__ od, uint32_t N, float init, float *__restrict__ scores,
                                                     float *__restrict__ contrib) {
    for (uint32_t v = blockIdx.x * kT + threadIdx.x; v < N; v += gridDim.x * kT) {
        scores[v] = init;
        contrib[v] = init / (float)od[v];
    }
}

// scores (level-major) -> the caller's numbering
__global__ void __launch_bounds__(kT) gi_unpermute_kernel(const float *__restrict__ s, const uint32_t *__restrict__ order, uint32_t N,
                                                          float *__restrict__ out) {
    for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < N; i += gridDim.x * kT) out[order[i]] = s[i];
}

template <int THREADS>
__device__ __forceinline__ double block_sum(double x, double *red) {  // fixed order: lanes by xor butterfly, then the waves in order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) t += red[w];
    return t;
}

__device__ __forceinline__ double finish(float s, float old, uint32_t od, uint32_t r, float *__restrict__ scores, float *cnew, float base,
                                         float damping, int f64_diff) {
    const float nw = base + damping * s;  // two roundings (-ffp-contract=off), like the reference
    scores[r] = nw;
    cnew[r] = nw / (float)od;             // written inside the sweep: the reading this file exists for
    return f64_diff ? fabs((double)nw - (double)old) : fabs((double)(nw - old));
}

struct LevelArgs {
    const Block *blocks;     // the level's row blocks ...
    uint32_t nb;
    const Item *items;       // ... and the phase-A items that ride in the same launch (an earlier level's)
    const uint32_t *off2, *gpos, *usrc, *od;
    const uint16_t *gperm, *upos, *asrc_x, *asrc_y;
    const float *ycur;       // the Y stream this sweep reads
    float *x, *ynext;        // the X stream (read by phase B, written by phase A), the Y stream of the next sweep
    float *cnew;             // this sweep's contributions (urgent reads, epilogue writes, phase A's slices)
    float *scores;
    double *partial;         // [nb]
    float base, damping;
    int f64_diff;
};

// ---- phase A: val[i] = slice[asrc[i]] for the positions of one work item --------------------------------------------------
// A workgroup step covers THREADS * 4 positions: lane t loads 4 local ids (8 bytes) and stores 4 values (16 bytes); the ids of
// the first steps are requested before the slice is staged.
template <int THREADS>
__device__ __forceinline__ void expand_role(const Item it, float *sl, const LevelArgs &a) {
    constexpr int PF = 4;
    constexpr uint32_t STEP = THREADS * 4;
    const uint16_t *as = it.cls ? a.asrc_y : a.asrc_x;
    float *out = it.cls ? a.ynext : a.x;
    const uint32_t a0 = it.begin + threadIdx.x * 4;
    const uint32_t last4 = it.end - 4u;  // the item's last id vector: loads past the end read it instead (no branch around a load)
    uint2 k[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) k[j] = *(const uint2 *)(as + min(a0 + j * STEP, last4));
    // the slice as aligned 16-byte vectors: LDS word 0 = node (node0 & ~3) -- the local ids carry the misalignment
    const float4 *c4 = (const float4 *)(a.cnew + (it.node0 & ~3u));
    const uint32_t nvec = ((it.node0 & 3u) + it.n + 3u) >> 2;
    for (uint32_t i = threadIdx.x; i < nvec; i += THREADS) ((float4 *)sl)[i] = c4[i];
    __syncthreads();
    for (uint32_t i0 = a0; i0 < it.end; i0 += PF * STEP) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const uint32_t i = i0 + j * STEP;
            if (i < it.end) {  // (begin and end are multiples of four: whole vectors)
                float4 v;
                v.x = sl[k[j].x & 0xffff];
                v.y = sl[k[j].x >> 16];
                v.z = sl[k[j].y & 0xffff];
                v.w = sl[k[j].y >> 16];
                *(float4 *)(out + i) = v;
            }
            k[j] = *(const uint2 *)(as + min(i + PF * STEP, last4));
        }
    }
}

// ---- phase B: one row block ---------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void reduce_role(const Block b, float *tile, const LevelArgs &a, double *red, uint32_t *wrow, uint32_t *n_wrow) {
    constexpr int RPL = kRowsPerLane;
    if (threadIdx.x == 0) *n_wrow = 0;
    // Two rounds of loads, each issued as a whole before anything waits: (1) the rows' own data, the stream elements' positions and
    // tile places, the urgent elements' sources and places; (2) the values -- from the streams, and for the urgent ones straight
    // from the contribution vector (written by the launch before this one).  Every load is UNCONDITIONAL, its index clamped into
    // the block (the arrays carry four spare words), and masked at use: a branch around a load makes the wave wait for everything
    // it has in flight at the end of the branch (the first form of this kernel waited five times in a row this way).
    uint32_t ra[RPL], rz[RPL], odv[RPL];
    float old[RPL];
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const uint32_t r = min(b.row0 + threadIdx.x + j * THREADS, b.row1 - 1u);
        ra[j] = a.off2[r] - b.e0;
        rz[j] = a.off2[r + 1] - b.e0;
        old[j] = a.scores[r];
        odv[j] = a.od[r];
    }
    constexpr int F = CZ_GI_F, U = 4;
    const uint32_t ng = b.g1 - b.g0, nu = b.u1 - b.u0;
    const uint32_t ng1 = max(ng, 1u) - 1u, nu1 = max(nu, 1u) - 1u;
    const uint32_t *gp = a.gpos + b.g0;
    const uint2 *gq = (const uint2 *)a.gperm + b.g0;
    const uint32_t *us = a.usrc + b.u0;
    const uint16_t *up = a.upos + b.u0;
    uint32_t usv[U], upv[U], p[F];
    uint2 q[F];
#pragma unroll
    for (int i = 0; i < U; i++) {
        const uint32_t j = min(threadIdx.x + i * THREADS, nu1);
        usv[i] = us[j];
        upv[i] = up[j];
    }
#pragma unroll
    for (int i = 0; i < F; i++) {
        const uint32_t g = min(threadIdx.x + i * THREADS, ng1);
        p[i] = gp[g];
        q[i] = gq[g];
    }
    auto place = [&](uint2 qq, float4 v) {  // (padding goes to the tile's spare words)
        tile[qq.x & 0xffff] = v.x;
        tile[qq.x >> 16] = v.y;
        tile[qq.y & 0xffff] = v.z;
        tile[qq.y >> 16] = v.w;
    };
    {
        float4 v[F];
        float uv[U];
#pragma unroll
        for (int i = 0; i < U; i++) uv[i] = a.cnew[usv[i]];
#pragma unroll
        for (int i = 0; i < F; i++) v[i] = *(const float4 *)(((p[i] & kYBit) ? a.ycur : (const float *)a.x) + (p[i] & ~kYBit));
#pragma unroll
        for (int i = 0; i < U; i++)
            if (threadIdx.x + i * THREADS < nu) tile[upv[i]] = uv[i];
#pragma unroll
        for (int i = 0; i < F; i++)
            if (threadIdx.x + i * THREADS < ng) place(q[i], v[i]);
    }
    for (uint32_t j = threadIdx.x + U * THREADS; j < nu; j += THREADS) tile[up[j]] = a.cnew[us[j]];  // (more than U urgent per lane: rare)
    for (uint32_t i0 = threadIdx.x + F * THREADS; i0 < ng; i0 += THREADS * F) {  // (more than F groups per lane: a 32768-value tile)
        float4 v[F];
#pragma unroll
        for (int i = 0; i < F; i++) {
            const uint32_t g = min(i0 + i * THREADS, ng1);
            p[i] = gp[g];
            q[i] = gq[g];
        }
#pragma unroll
        for (int i = 0; i < F; i++) v[i] = *(const float4 *)(((p[i] & kYBit) ? a.ycur : (const float *)a.x) + (p[i] & ~kYBit));
#pragma unroll
        for (int i = 0; i < F; i++)
            if (i0 + i * THREADS < ng) place(q[i], v[i]);
    }
    __syncthreads();
    double err = 0.0;
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const uint32_t r = b.row0 + threadIdx.x + j * THREADS;
        if (r < b.row1) {
            if (rz[j] - ra[j] >= kWaveRow) {
                wrow[atomicAdd(n_wrow, 1u)] = threadIdx.x + j * THREADS;
            } else {
                float s = 0.0f;
                for (uint32_t e = ra[j]; e < rz[j]; e++) s = s + tile[e];
                err += finish(s, old[j], odv[j], r, a.scores, a.cnew, a.base, a.damping, a.f64_diff);
            }
        }
    }
    __syncthreads();
    const uint32_t nw = *n_wrow;
    if (nw) {  // (uniform) rows worth a whole wave: exact_sum.h gives the value of the sequential f32 loop
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (uint32_t i = wave; i < nw; i += THREADS / 64) {
            const uint32_t rr = b.row0 + wrow[i];
            const uint32_t e0 = a.off2[rr] - b.e0, e1 = a.off2[rr + 1] - b.e0;
            const float s = cz_exact::wave_seq_sum<16>(tile + e0, e1 - e0, 0.0f);
            if (lane == 0) err += finish(s, a.scores[rr], a.od[rr], rr, a.scores, a.cnew, a.base, a.damping, a.f64_diff);
        }
    }
    const double tot = block_sum<THREADS>(err, red);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = tot;
}

// One launch per level: workgroups [0, nb) are the level's row blocks (phase B), the rest the phase-A items of the level
// `urgent_gap` below it -- the two are independent (csrc/inplace_plan.hpp), so phase A fills the chip beside phase B without a
// second stream or an event (a dependency between two branches of a hipGraph measured ~8 us; a kernel boundary ~1.5).
#ifndef CZ_GI_WAVES
#define CZ_GI_WAVES 8  // waves per SIMD the level kernel is compiled for: 8 = two 1024-thread workgroups per CU (64 VGPRs)
#endif
__global__ void __launch_bounds__(kLT) __attribute__((amdgpu_waves_per_eu(CZ_GI_WAVES, CZ_GI_WAVES))) gi_level_kernel(const LevelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[kLT / 64];
    __shared__ uint32_t wrow[kMaxWaveRows], n_wrow;
    if (blockIdx.x < a.nb) reduce_role<kLT>(a.blocks[blockIdx.x], lds, a, red, wrow, &n_wrow);
    else expand_role<kLT>(a.items[blockIdx.x - a.nb], lds, a);
}

// ---- rows longer than a tile, one workgroup each: all threads gather a tile, wave 0 adds it to the running sum with the wave
// form of the sequential f32 sum, tile after tile
__global__ void __launch_bounds__(kT) gi_long_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ long_off,
                                                     const uint32_t *__restrict__ long_src, uint32_t k0, const uint32_t *__restrict__ od,
                                                     const float *__restrict__ cold, float *cnew, float *__restrict__ scores, float base,
                                                     float damping, int f64_diff, double *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float tile[kLongTile + 4];
    const uint32_t k = k0 + blockIdx.x;
    const uint32_t r = rows[k];
    const uint32_t e0 = long_off[k], e1 = long_off[k + 1];
    float s = 0.0f;
    for (uint32_t t0 = e0; t0 < e1; t0 += kLongTile) {
        const uint32_t n = min(kLongTile, e1 - t0);
        __syncthreads();  // wave 0 is done with the previous tile
        for (uint32_t i = threadIdx.x; i < n; i += kT) {
            const uint32_t v = long_src[t0 + i];
            tile[i] = (v & kOldBit) ? cold[v & ~kOldBit] : cnew[v];
        }
        __syncthreads();
        if (threadIdx.x < 64) s = cz_exact::wave_seq_sum<16>(tile, n, s);
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = finish(s, scores[r], od[r], r, scores, cnew, base, damping, f64_diff);
}

// the sweep's error: the partials added up in a fixed order by ONE workgroup (lane t takes partial[t], partial[t + 1024], ...;
// the same tree, hence the same bits, on every run)
__global__ void __launch_bounds__(kLT) gi_sum_partials_kernel(const double *__restrict__ partial, uint32_t n, double *__restrict__ out) {
    __shared__ double red[kLT / 64];
    double x = 0.0;
    uint32_t i = threadIdx.x;
    for (; i + 7 * kLT < n; i += 8 * kLT) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = partial[i + j * kLT];
#pragma unroll
        for (int j = 0; j < 8; j++) x += v[j];
    }
    for (; i < n; i += kLT) x += partial[i];
    const double tot = block_sum<kLT>(x, red);
    if (threadIdx.x == 0) *out = tot;
}

template <typename T, typename Vec>
hipError_t upload(cz::DevBuf<T> &d, const Vec &h, size_t extra = 0) {
    static_assert(std::is_same<T, typename Vec::value_type>::value, "element types differ");
    hipError_t e = d.alloc(h.size() + extra);
    if (e != hipSuccess) return e;
    if (!h.empty()) e = hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    // the spare words are READ (clamped, unconditional loads of a block without elements) and used as indices: they must be zero
    if (e == hipSuccess) e = hipMemset(d.p + h.size(), 0, std::max<size_t>(extra, h.empty() ? 1 : 0) * sizeof(T));
    return e;
}

}  // namespace

struct cz_pagerank_inplace_plan {
    uint32_t N = 0, L = 0, gap = 1;
    uint64_t E = 0;
    float damping = 0.85f;
    int f64_diff = 0;
    // static layout
    cz::DevBuf<uint32_t> d_order, d_off2, d_od, d_long_rows, d_long_off, d_long_src, d_gpos, d_usrc;
    cz::DevBuf<uint16_t> d_asrc[2], d_gperm, d_upos;
    cz::DevBuf<Block> d_blocks;
    cz::DevBuf<Item> d_items;
    std::vector<uint32_t> blk_first, item_first, long_first;
    uint32_t n_blocks = 0, n_items = 0, n_long = 0;
    uint64_t n_pos[2] = {0, 0}, n_edges[3] = {0, 0, 0}, n_long_edges = 0;
    uint32_t slice = kSliceDefault, tile = kTileDefault;
    uint32_t lds_bytes = 0;  // dynamic LDS of the level kernel: a tile or a slice, whichever is larger
    // state
    cz::DevBuf<float> d_c[2], d_X, d_Y[2], d_scores;
    cz::DevBuf<double> d_partial, d_err;
    uint32_t sweeps_done = 0;  // parity of the next sweep = sweeps_done & 1
    // launch machinery
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    bool use_graph = true;
    uint32_t launches = 0;
    double build_ms = 0.0, h2d_ms = 0.0;

    ~cz_pagerank_inplace_plan() {
        for (hipGraphExec_t g : exec)
            if (g) (void)hipGraphExecDestroy(g);
    }
};

namespace {

LevelArgs level_args(const cz_pagerank_inplace_plan *p, uint32_t par) {
    LevelArgs a{};
    a.off2 = p->d_off2.p;
    a.gpos = p->d_gpos.p;
    a.usrc = p->d_usrc.p;
    a.od = p->d_od.p;
    a.gperm = p->d_gperm.p;
    a.upos = p->d_upos.p;
    a.asrc_x = p->d_asrc[0].p;
    a.asrc_y = p->d_asrc[1].p;
    a.ycur = p->d_Y[par].p;
    a.x = p->d_X.p;
    a.ynext = p->d_Y[par ^ 1].p;
    a.cnew = p->d_c[par].p;
    a.scores = p->d_scores.p;
    a.base = (1.0f - p->damping) / (float)p->N;
    a.damping = p->damping;
    a.f64_diff = p->f64_diff;
    return a;
}

// one launch: row blocks [b0, b1) and items [i0, i1)
void launch_level(const cz_pagerank_inplace_plan *p, LevelArgs a, uint32_t b0, uint32_t b1, uint32_t i0, uint32_t i1, hipStream_t s) {
    if (b1 <= b0 && i1 <= i0) return;
    a.blocks = p->d_blocks.p + b0;
    a.nb = b1 - b0;
    a.items = p->d_items.p + i0;
    a.partial = p->d_partial.p + b0;
    hipLaunchKernelGGL(gi_level_kernel, dim3((b1 - b0) + (i1 - i0)), dim3(kLT), p->lds_bytes, s, a);
}

// the launches of one sweep of parity `par` (it reads Y[par], writes the contributions c[par] and Y[par ^ 1]), all on one stream:
// launch l = phase B of level l + phase A of level l - gap; the last `gap` levels' phase A in one launch at the end.
// Called once per parity under stream capture, or directly (CZ_PR_INPLACE_GRAPH=0).  Returns the number of launches through *n.
int enqueue_sweep(cz_pagerank_inplace_plan *p, uint32_t par, hipStream_t s, uint32_t *n) {
    const LevelArgs a = level_args(p, par);
    const float *co = p->d_c[par ^ 1].p;
    uint32_t count = 0;
    for (uint32_t l = 0; l < p->L; l++) {
        uint32_t i0 = 0, i1 = 0;
        if (p->gap > 0 && l >= p->gap) {
            i0 = p->item_first[l - p->gap];
            i1 = p->item_first[l - p->gap + 1];
        }
        const uint32_t b0 = p->blk_first[l], b1 = p->blk_first[l + 1];
        if (b1 > b0 || i1 > i0) {
            launch_level(p, a, b0, b1, i0, i1, s);
            count++;
        }
        const uint32_t nl = p->long_first[l + 1] - p->long_first[l];
        if (nl) {
            hipLaunchKernelGGL(gi_long_kernel, dim3(nl), dim3(kT), 0, s, p->d_long_rows.p, p->d_long_off.p, p->d_long_src.p, p->long_first[l],
                               p->d_od.p, co, a.cnew, p->d_scores.p, a.base, a.damping, a.f64_diff, p->d_partial.p + p->n_blocks + p->long_first[l]);
            count++;
        }
        if (p->gap == 0 && p->item_first[l + 1] > p->item_first[l]) {  // no urgent class: phase A between the levels
            launch_level(p, a, 0, 0, p->item_first[l], p->item_first[l + 1], s);
            count++;
        }
    }
    if (p->gap > 0) {  // phase A of the last levels (their Y values are the next sweep's)
        const uint32_t i0 = p->item_first[p->L > p->gap ? p->L - p->gap : 0], i1 = p->item_first[p->L];
        if (i1 > i0) {
            launch_level(p, a, 0, 0, i0, i1, s);
            count++;
        }
    }
    hipLaunchKernelGGL(gi_sum_partials_kernel, dim3(1), dim3(kLT), 0, s, p->d_partial.p, p->n_blocks + p->n_long, p->d_err.p);
    count++;
    if (n) *n = count;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank_inplace launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

int capture_sweep(cz_pagerank_inplace_plan *p, uint32_t par) {
    hipStream_t cap = nullptr;
    CZ_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    struct Guard {
        hipStream_t a;
        ~Guard() { (void)hipStreamDestroy(a); }
    } guard{cap};
    CZ_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_sweep(p, par, cap, &p->launches);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cap, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) return cz::set_error(CZ_E_HIP, "pagerank_inplace: stream capture failed: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&p->exec[par], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        p->exec[par] = nullptr;
        return cz::set_error(CZ_E_HIP, "pagerank_inplace: hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    return CZ_OK;
}

// one sweep on `stream`, no host round trip
int launch_sweep(cz_pagerank_inplace_plan *p, hipStream_t stream) {
    const uint32_t par = p->sweeps_done & 1u;
    if (p->use_graph) {
        if (!p->exec[par]) {
            if (capture_sweep(p, par) != CZ_OK) {  // capture is an optimisation: without it the launches go out one by one
                (void)hipGetLastError();
                if (getenv("CZ_PR_INPLACE_TRACE")) fprintf(stderr, "[pagerank_inplace] no graph replay: %s\n", cz_last_error());
                p->use_graph = false;
            }
        }
        if (p->exec[par]) {
            CZ_HIP(hipGraphLaunch(p->exec[par], stream));
            p->sweeps_done++;
            return CZ_OK;
        }
    }
    int rc = enqueue_sweep(p, par, stream, &p->launches);
    if (rc) return rc;
    p->sweeps_done++;
    return CZ_OK;
}

int env_u32(const char *name, uint32_t dflt, uint32_t lo, uint32_t hi, uint32_t *out) {
    *out = dflt;
    if (const char *e = getenv(name)) {
        const long v = atol(e);
        if (v < (long)lo || v > (long)hi) return cz::set_error(CZ_E_INVALID, "%s=%s out of range [%u, %u]", name, e, lo, hi);
        *out = (uint32_t)v;
    }
    return CZ_OK;
}

}  // namespace

extern "C" int cz_pagerank_inplace_plan_create(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                                               uint64_t E, float damping, uint32_t flags, cz_pagerank_inplace_plan **out) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (!in_offsets || (N && !out_degree) || (E > 0 && !in_sources)) return cz::set_error(CZ_E_INVALID, "null buffer");
    if (N >= kOldBit || E >= 0x7FFFFFF0ull) return cz::set_error(CZ_E_UNSUPPORTED, "node ids must stay below 2^31 and E below 2^31 - 16");
    const auto t0 = std::chrono::steady_clock::now();
    // the layout is built on the host: device arrays come back first
    czgs::PodVec<uint32_t> h_off, h_src, h_od;  // (not zero-filled first: 400 MB on the 10M / 100M graph)
    if (flags & CZ_DEVICE_PTRS) {
        h_off.resize_uninit((size_t)N + 1);
        h_src.resize_uninit(E);
        h_od.resize_uninit(N);
        CZ_HIP(hipMemcpy(h_off.data(), in_offsets, ((size_t)N + 1) * 4, hipMemcpyDeviceToHost));
        if (E) CZ_HIP(hipMemcpy(h_src.data(), in_sources, E * 4, hipMemcpyDeviceToHost));
        if (N) CZ_HIP(hipMemcpy(h_od.data(), out_degree, (size_t)N * 4, hipMemcpyDeviceToHost));
        in_offsets = h_off.data();
        in_sources = h_src.data();
        out_degree = h_od.data();
    }
    if (in_offsets[0] != 0 || in_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "offsets[0] must be 0 and offsets[N] == E");
    {   // what the level schedule relies on: every in-list ascends (strictly or with repeats: parallel edges are kept) and every
        // source is a node; the lists as_directed_graph builds are sorted (CsrLayout::Sorted), anything else is refused
        const uint32_t T = E < (1u << 20) ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<uint64_t> bad((size_t)T * 2, 0);
        auto check = [&](uint32_t t) {  // nodes [N t / T, N (t + 1) / T)
            uint64_t order = 0, id = 0;
            for (uint32_t u = (uint32_t)((uint64_t)N * t / T), u1 = (uint32_t)((uint64_t)N * (t + 1) / T); u < u1; u++) {
                const uint32_t a = in_offsets[u], z = in_offsets[u + 1];
                if (z < a || z > E) {
                    order++;
                    continue;
                }
                bool asc = true;
                for (uint32_t e = a; e < z; e++) {
                    if (in_sources[e] >= N) id++;
                    if (e > a && in_sources[e] < in_sources[e - 1]) asc = false;
                }
                if (!asc) order++;
            }
            bad[2 * (size_t)t] = order;
            bad[2 * (size_t)t + 1] = id;
        };
        if (T == 1) check(0);
        else {
            std::vector<std::thread> th;
            for (uint32_t t = 0; t < T; t++) th.emplace_back(check, t);
            for (auto &x : th) x.join();
        }
        uint64_t bad_order = 0, bad_id = 0;
        for (uint32_t t = 0; t < T; t++) {
            bad_order += bad[2 * (size_t)t];
            bad_id += bad[2 * (size_t)t + 1];
        }
        if (bad_id) return cz::set_error(CZ_E_INVALID, "%llu in_sources are not node ids (>= N = %u)", (unsigned long long)bad_id, N);
        if (bad_order)
            return cz::set_error(CZ_E_INVALID, "%llu in-lists are not in ascending order (the level schedule needs CsrLayout::Sorted rows)",
                                 (unsigned long long)bad_order);
    }
    czgs::Params prm;
    if ((rc = env_u32("CZ_PR_INPLACE_TILE", kTileDefault, 256, 32768, &prm.tile))) return rc;
    prm.rows_per_block = kLT * kRowsPerLane;
    if ((rc = env_u32("CZ_PR_INPLACE_SLICE", kSliceDefault, 64, 40444, &prm.slice))) return rc;
    if ((rc = env_u32("CZ_PR_INPLACE_PART", kPartDefault, 64, 1u << 20, &prm.part))) return rc;
    prm.part &= ~3u;
    if ((rc = env_u32("CZ_PR_INPLACE_GAP", 1, 0, 64, &prm.urgent_gap))) return rc;
    if ((rc = env_u32("CZ_PR_INPLACE_MAX_LEVELS", 4096, 1, 1u << 24, &prm.max_levels))) return rc;
    if (flags & CZ_PR_INPLACE_AS_JACOBI) prm.jacobi = true;
    czgs::Plan h;
    if (!czgs::build_plan(in_offsets, in_sources, out_degree, N, prm, h))
        return cz::set_error(h.error.find("dependence levels") != std::string::npos ? CZ_E_UNSUPPORTED : CZ_E_INVALID, "%s", h.error.c_str());
    const auto t1 = std::chrono::steady_clock::now();
    std::unique_ptr<cz_pagerank_inplace_plan> p(new cz_pagerank_inplace_plan());
    p->N = N;
    p->E = E;
    p->L = h.L;
    p->gap = prm.urgent_gap;
    p->slice = prm.slice;
    p->tile = prm.tile;
    p->lds_bytes = std::max(prm.tile + 4u, prm.slice + 4u) * 4u;
    p->damping = damping;
    p->f64_diff = (flags & CZ_PR_ERR_F64_DIFF) ? 1 : 0;
    p->blk_first = h.blk_first;
    p->item_first = h.item_first;
    p->long_first = h.long_first;
    p->n_blocks = (uint32_t)h.blocks.size();
    p->n_items = (uint32_t)h.items.size();
    p->n_long = (uint32_t)h.long_rows.size();
    for (int c = 0; c < 2; c++) p->n_pos[c] = h.n_pos[c];
    for (int c = 0; c < 3; c++) p->n_edges[c] = h.n_edges[c];
    p->n_long_edges = h.n_long_edges;
    if (N) {
        CZ_HIP(upload(p->d_order, h.order));
        CZ_HIP(upload(p->d_off2, h.off2));
        CZ_HIP(upload(p->d_od, h.od));
        CZ_HIP(upload(p->d_blocks, h.blocks));
        CZ_HIP(upload(p->d_items, h.items));
        CZ_HIP(upload(p->d_long_rows, h.long_rows));
        CZ_HIP(upload(p->d_long_off, h.long_off));
        CZ_HIP(upload(p->d_long_src, h.long_src));
        CZ_HIP(upload(p->d_asrc[0], h.asrc[0]));
        CZ_HIP(upload(p->d_asrc[1], h.asrc[1]));
        CZ_HIP(upload(p->d_gpos, h.gpos, 4));
        CZ_HIP(upload(p->d_gperm, h.gperm, 16));
        CZ_HIP(upload(p->d_upos, h.upos, 4));
        CZ_HIP(upload(p->d_usrc, h.usrc, 4));
        CZ_HIP(p->d_c[0].alloc((size_t)N + 8));  // (phase A reads whole aligned vectors around a slice)
        CZ_HIP(p->d_c[1].alloc((size_t)N + 8));
        CZ_HIP(hipMemset(p->d_c[0].p, 0, ((size_t)N + 8) * 4));
        CZ_HIP(hipMemset(p->d_c[1].p, 0, ((size_t)N + 8) * 4));
        CZ_HIP(p->d_scores.alloc(N));
        CZ_HIP(p->d_X.alloc(h.n_pos[0] + 4));
        CZ_HIP(p->d_Y[0].alloc(h.n_pos[1] + 4));
        CZ_HIP(p->d_Y[1].alloc(h.n_pos[1] + 4));
        CZ_HIP(hipMemset(p->d_X.p, 0, (h.n_pos[0] + 4) * 4));
        CZ_HIP(hipMemset(p->d_Y[0].p, 0, (h.n_pos[1] + 4) * 4));
        CZ_HIP(hipMemset(p->d_Y[1].p, 0, (h.n_pos[1] + 4) * 4));
        CZ_HIP(p->d_partial.alloc((size_t)p->n_blocks + p->n_long + 1));
        CZ_HIP(p->d_err.alloc(1));
        // more than 64 KiB of dynamic LDS has to be asked for, on THIS device (the attribute is per device)
        CZ_HIP(hipFuncSetAttribute((const void *)gi_level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
        CZ_HIP(hipDeviceSynchronize());
    }
    if (const char *g = getenv("CZ_PR_INPLACE_GRAPH")) p->use_graph = atoi(g) != 0;
    const auto t2 = std::chrono::steady_clock::now();
    p->build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    p->h2d_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    *out = p.release();
    return CZ_OK;
}

extern "C" void cz_pagerank_inplace_plan_destroy(cz_pagerank_inplace_plan *p) {
    if (!p) return;
    (void)cz::ensure_device();
    (void)hipDeviceSynchronize();
    delete p;
}

// graph::page_rank's initial state: scores 1/N, contributions (1/N) / out_degree, and the Y stream the first sweep reads
extern "C" int cz_pagerank_inplace_plan_init(cz_pagerank_inplace_plan *p, void *stream) {
    if (!p) return cz::set_error(CZ_E_INVALID, "null plan");
    int rc = cz::ensure_device();
    if (rc) return rc;
    p->sweeps_done = 0;
    if (p->N == 0) return CZ_OK;
    hipStream_t s = (hipStream_t)stream;
    const float init = 1.0f / (float)p->N;
    // "the sweep before the first": its contributions live in c[1], its Y values in Y[0]
    hipLaunchKernelGGL(gi_init_kernel, dim3(grid_for(p->N)), dim3(kT), 0, s, p->d_od.p, p->N, init, p->d_scores.p, p->d_c[1].p);
    if (p->n_items) {  // every item once: the X values it writes are overwritten before any row reads them
        LevelArgs a = level_args(p, 1);  // stages c[1], writes X and Y[0]
        launch_level(p, a, 0, 0, 0, p->n_items, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank_inplace init: %s", hipGetErrorString(e));
    return CZ_OK;
}

// n more sweeps on `stream`, nothing read back (bench.py brackets this with events)
extern "C" int cz_pagerank_inplace_plan_sweeps(cz_pagerank_inplace_plan *p, uint32_t n, void *stream) {
    if (!p) return cz::set_error(CZ_E_INVALID, "null plan");
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (p->N == 0) return CZ_OK;
    for (uint32_t i = 0; i < n; i++)
        if ((rc = launch_sweep(p, (hipStream_t)stream))) return rc;
    return CZ_OK;
}

// the loop of graph::page_rank from the initial state: sweeps until err < tolerance or max_iter
extern "C" int cz_pagerank_inplace_plan_run(cz_pagerank_inplace_plan *p, double tolerance, uint32_t max_iter, uint32_t *iters_run,
                                            double *final_err, const volatile uint8_t *poison, void *stream) {
    if (!p) return cz::set_error(CZ_E_INVALID, "null plan");
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (p->N == 0) return CZ_OK;  // pagerank.rs:43-45
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "max_iter must be positive");
    int rc = cz_pagerank_inplace_plan_init(p, stream);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    uint32_t it = 0;
    double err = 0.0;
    for (;;) {
        if (cz::poisoned(poison)) {
            (void)hipStreamSynchronize(s);
            return cz::set_error(CZ_E_CANCELLED, "cancelled");
        }
        if ((rc = launch_sweep(p, s))) return rc;
        CZ_HIP(hipMemcpyAsync(&err, p->d_err.p, 8, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        it++;
        if (err < tolerance || it == max_iter) break;
    }
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return CZ_OK;
}

// scores [N] in the caller's numbering: host memory, or device memory with CZ_DEVICE_PTRS
extern "C" int cz_pagerank_inplace_plan_read_scores(cz_pagerank_inplace_plan *p, float *scores, uint32_t flags, void *stream) {
    if (!p || (p->N && !scores)) return cz::set_error(CZ_E_INVALID, "null argument");
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (p->N == 0) return CZ_OK;
    hipStream_t s = (hipStream_t)stream;
    if (flags & CZ_DEVICE_PTRS) {
        hipLaunchKernelGGL(gi_unpermute_kernel, dim3(grid_for(p->N)), dim3(kT), 0, s, p->d_scores.p, p->d_order.p, p->N, scores);
        CZ_HIP(hipStreamSynchronize(s));
        return CZ_OK;
    }
    cz::DevBuf<float> tmp;
    CZ_HIP(tmp.alloc(p->N));
    hipLaunchKernelGGL(gi_unpermute_kernel, dim3(grid_for(p->N)), dim3(kT), 0, s, p->d_scores.p, p->d_order.p, p->N, tmp.p);
    CZ_HIP(hipMemcpyAsync(scores, tmp.p, (size_t)p->N * 4, hipMemcpyDeviceToHost, s));
    CZ_HIP(hipStreamSynchronize(s));
    return CZ_OK;
}

// shape [16] (u64 each): levels, row blocks, phase-A items, long rows, urgent gap, slice width, launches per sweep, graph replay
// (1/0), X edges, Y edges, urgent edges, long-row edges, X stream positions, Y stream positions; build_ms / h2d_ms: host layout, upload
extern "C" int cz_pagerank_inplace_plan_info(const cz_pagerank_inplace_plan *p, uint64_t *shape, double *build_ms, double *h2d_ms) {
    if (!p) return cz::set_error(CZ_E_INVALID, "null plan");
    if (shape) {
        uint32_t launches = p->launches;
        if (!launches) {  // nothing launched yet: count what a sweep will launch
            launches = 1 + (p->gap > 0 ? 1 : 0);
            for (uint32_t l = 0; l < p->L; l++)
                launches += 1 + (p->long_first[l + 1] > p->long_first[l]) + (p->gap == 0 && p->item_first[l + 1] > p->item_first[l]);
        }
        const uint64_t v[14] = {p->L, p->n_blocks, p->n_items, p->n_long, p->gap, p->slice, launches, (uint64_t)(p->use_graph ? 1 : 0),
                                p->n_edges[0], p->n_edges[1], p->n_edges[2], p->n_long_edges, p->n_pos[0], p->n_pos[1]};
        for (int i = 0; i < 14; i++) shape[i] = v[i];
        shape[14] = p->tile;
        shape[15] = 0;
    }
    if (build_ms) *build_ms = p->build_ms;
    if (h2d_ms) *h2d_ms = p->h2d_ms;
    return CZ_OK;
}

// the one-shot form (host pointers): plan, loop, scores, plan dropped -- what an `impl FixedRule` calls with `in_place: true`
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
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "max_iter must be positive");
    if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    cz_pagerank_inplace_plan *raw = nullptr;
    rc = cz_pagerank_inplace_plan_create(in_offsets, in_sources, out_degree, N, E, damping, flags & CZ_PR_ERR_F64_DIFF, &raw);
    if (rc) return rc;
    struct Guard {
        cz_pagerank_inplace_plan *p;
        ~Guard() { cz_pagerank_inplace_plan_destroy(p); }
    } guard{raw};
    if (n_levels) *n_levels = raw->L;
    rc = cz_pagerank_inplace_plan_run(raw, tolerance, max_iter, iters_run, final_err, poison, nullptr);
    if (rc) return rc;
    return cz_pagerank_inplace_plan_read_scores(raw, scores, 0, nullptr);
}
