// pagerank.hip -- PageRank fixed rule on gfx950 (C ABI: cz_pagerank, cz_pagerank_plan_*).
//
// Reference: PageRank::run (cozo-core/src/fixed_rule/algos/pagerank.rs:29-56) -> graph 0.3.1 `page_rank`:
//   init = 1/N, base = (1-d)/N, contrib[v] = score[v]/out_degree(v); per iteration every node u
//   new = base + d * sum_{v in in(u)} contrib[v]   (f32, in-neighbours summed SEQUENTIALLY in sorted order)
//   err += |new - old| (f64); contrib refreshed after the sweep (Jacobi); stop at err < tol or max_iter.
//
// Two device formulations of the same sweep, both bit-identical to the oracle's restatement of it (each row's
// in-neighbour contributions are added one by one in ascending source order: short rows by ONE lane each, rows of
// >= kWaveRow terms by a whole wave through exact_sum.h -- the same value as the sequential f32 loop, bit for bit):
//
//  * "blocked" (source-blocked two-phase sweep; the fast path).  A 4-byte gather from a 40 MB contribution
//    vector moves a whole 128-byte line through the L2->L1 path: measured <= 215 G gathers/s even when the
//    vector is L2-resident, 58 G/s from the Infinity Cache (scratch/gather_bench.hip).  So the gather is
//    done in LDS instead:
//      phase A `pb_expand_kernel`: the source range is cut into slices of W = 32768 nodes; a workgroup stages
//        one slice of contrib[] in LDS (128 KiB), streams the slice's edges (u16 local source ids, stored in
//        (slice, row-block, row, source) order), and writes val[i] = contrib[src_i] as one coalesced stream;
//      phase B `pb_reduce_kernel`: a workgroup owns a row block (consecutive rows with <= 16384 in-edges),
//        reads the block's pieces of every slice's value stream (contiguous runs), drops each value at its
//        CSR position inside an LDS tile (u16 permutation), then each row is summed in order by one lane with
//        the fused epilogue (new score, |delta| in f64, next contribution).
//    HBM bytes per edge: 2 (local ids) + 4 (val out) + 4 (val in) + 2 (permutation) = 12 instead of the
//    compulsory 4, all of it streaming.  The static layout (sort by slice, segment table, permutation) is
//    built once on the device at plan creation.
//  * "gather" (CSR-stream pull SpMV): phase 1 streams the row block's source ids and gathers contrib[src]
//    from global memory into the LDS tile; phase 2 as above.  Used for small graphs, for shards whose slices
//    would be too sparse.  Rows longer than a tile (hubs) of either formulation: `pr_hub_kernel`, one workgroup per
//    row, gathered tile by tile while one wave sums the tile before it, still in order.
//
// Algorithmic HBM bytes per iteration (the roofline model of SURVEY.md section 8d, cache-perfect gathers):
// 4E (ids) + 4(N+1) (offsets) + 20N (contrib in/out, score in/out, out-degree) = 6.4 B/edge at N = 10M, E = 100M.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "sort_scan.h"

#include "common.h"
#include "exact_sum.h"

namespace {

constexpr int kGThreads = 256;      // gather kernel
constexpr int kGTileNnz = 4096;     // f32 values per LDS tile of the gather kernel (16 KiB)
constexpr int kBThreads = 1024;     // blocked path, phase B
constexpr int kBTileNnz = 16384;    // 64 KiB tile: two workgroups (32 waves) per CU
constexpr int kAThreads = 1024;     // blocked path, phase A
constexpr int kMaxSliceLog2 = 15;   // 32768 sources = 128 KiB of LDS
constexpr uint32_t kPartEdges = 98304;  // phase-A work item: at most this many edges of one slice
constexpr int kMaxRowsPerBlock = 2048;  // = 2 rows per lane of phase B
constexpr uint32_t kMinWaveRow = 128;   // rows of at least `wave_row` (>= this) terms are summed by a wave (exact_sum.h)
constexpr uint32_t kWaveRowDefault = 2048;
constexpr uint32_t kHeavyRowDefault = 128;  // rows of at least this many terms are moved behind the others, longest first
constexpr int kHThreads = 1024;         // hub rows: one workgroup per row
constexpr int kHTileNnz = 8192;         // two of these in LDS (64 KiB)

struct RowBlock {
    uint32_t row0, row1;  // local rows [row0, row1)
    uint32_t e0, e1;      // their in-edges [e0, e1) = off[row0], off[row1]
};
struct AItem {
    uint32_t begin, end, slice, pad;  // positions [begin, end) of the slice-ordered edge stream
};

__global__ void __launch_bounds__(256)
pr_init_kernel(float *__restrict__ contrib, const uint32_t *__restrict__ out_deg, uint32_t N, float init) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x)
        contrib[v] = init / (float)out_deg[v];
}

__global__ void __launch_bounds__(256) pr_fill_kernel(float *__restrict__ p, uint32_t n, float v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

template <int THREADS>
__device__ __forceinline__ double block_sum_f64(double v, double *red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < THREADS / 64; w++) s += red[w];
    return s;  // valid on thread 0
}

// ---- rows of a tile -> scores -------------------------------------------------------------------------------------
// the fused epilogue of a row whose f32 sum is s; returns |new - old| for the f64 error
// (r = the row as the CALLER numbers it: plans that moved their long rows pass row_id[plan row])
__device__ __forceinline__ double finish_row(float s, uint32_t r, float old, uint32_t od, uint32_t row_begin,
                                             float *__restrict__ contrib_out, float *__restrict__ scores, float base,
                                             float damping) {
    const float nw = base + damping * s;  // two roundings, like the reference (no fma: -ffp-contract=off)
    scores[r] = nw;
    contrib_out[row_begin + r] = nw / (float)od;
    return fabs((double)(nw - old));
}
__device__ __forceinline__ uint32_t caller_row(const uint32_t *__restrict__ row_id, uint32_t r) { return row_id ? row_id[r] : r; }

// one lane adds tile[e .. z) in order.  A long row is a serial chain of dependent v_add_f32: its LDS reads are kept 16
// values ahead of the adds in two register sets that take turns (a read waited for in place costs ~100 cycles per add).
__device__ __forceinline__ float lane_row_sum(const float *tile, uint32_t e, uint32_t z) {
    float s = 0.0f;
    if (z - e >= 32) {
        float a[16], b[16];
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = tile[e + i];
        while (e + 48 <= z) {
#pragma unroll
            for (int i = 0; i < 16; i++) b[i] = tile[e + 16 + i];
#pragma unroll
            for (int i = 0; i < 16; i++) s = s + a[i];
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = tile[e + 32 + i];
#pragma unroll
            for (int i = 0; i < 16; i++) s = s + b[i];
            e += 32;
        }
        // a holds [e, e + 16); between 16 and 47 values are left
        if (e + 32 <= z) {
#pragma unroll
            for (int i = 0; i < 16; i++) b[i] = tile[e + 16 + i];
#pragma unroll
            for (int i = 0; i < 16; i++) s = s + a[i];
#pragma unroll
            for (int i = 0; i < 16; i++) s = s + b[i];
            e += 32;
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) s = s + a[i];
            e += 16;
        }
    }
    for (; e < z; e++) s = s + tile[e];
    return s;
}

// Rows of >= wave_row terms are pushed to `raw` in whatever order the lanes get there; the list is then put in row order
// so that which wave sums which row -- and with it the order of the f64 error terms -- does not depend on timing.
// Ends with a barrier; every thread gets the list length.
constexpr int kMaxWaveRows = kBTileNnz / (int)kMinWaveRow;
struct WaveRowList {
    uint32_t raw[kMaxWaveRows], sorted[kMaxWaveRows];
    uint32_t n;
};
__device__ __forceinline__ uint32_t order_wave_rows(WaveRowList &l) {
    __syncthreads();
    const uint32_t nl = l.n;
    if (nl == 0) return 0;
    if (threadIdx.x < nl) {
        const uint32_t mine = l.raw[threadIdx.x];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < nl; i++) rank += l.raw[i] < mine ? 1u : 0u;
        l.sorted[rank] = mine;
    }
    __syncthreads();
    return nl;
}

// the listed rows, one wave each: exact_sum.h gives the value of the sequential f32 loop
template <int THREADS>
__device__ __forceinline__ double wave_rows(const WaveRowList &l, uint32_t nl, const RowBlock rb,
                                            const uint32_t *__restrict__ off, uint32_t e0, const float *tile,
                                            const uint32_t *__restrict__ out_deg, uint32_t row_begin,
                                            float *__restrict__ contrib_out, float *__restrict__ scores, float base,
                                            float damping, const uint32_t *__restrict__ row_id) {
    double err = 0.0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t i = wave; i < nl; i += THREADS / 64) {
        const uint32_t r = rb.row0 + l.sorted[i];
        const uint32_t a0 = off[r] - e0, z0 = off[r + 1] - e0;
        const float s = cz_exact::wave_seq_sum<16>(tile + a0, z0 - a0, 0.0f);
        if (lane == 0) {
            const uint32_t cr = caller_row(row_id, r);
            err += finish_row(s, cr, scores[cr], out_deg[row_begin + cr], row_begin, contrib_out, scores, base, damping);
        }
    }
    return err;
}

// ---- "gather" formulation -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(kGThreads)
pr_step_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ off /* local, [rows+1] */,
               const uint32_t *__restrict__ src, const uint32_t *__restrict__ out_deg /* global ids */,
               uint32_t row_begin, const float *__restrict__ contrib_in, float *__restrict__ contrib_out,
               float *__restrict__ scores /* local */, float base, float damping, double *__restrict__ partial,
               const uint32_t *__restrict__ row_id, uint32_t wave_row) {
    __shared__ __attribute__((aligned(16))) float tile[kGTileNnz];
    __shared__ double red[kGThreads / 64];
    __shared__ WaveRowList wl;
    const RowBlock rb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t e0 = rb.e0, e1 = rb.e1;
    if (tid == 0) wl.n = 0;
    // phase 1: coalesced id stream + gather (a block holds at most one tile; longer rows: pr_hub_kernel)
    const uint32_t nnz = e1 - e0;
    uint32_t i = tid;
    for (; i + 3 * kGThreads < nnz; i += 4 * kGThreads) {
        uint32_t s0 = src[e0 + i], s1 = src[e0 + i + kGThreads], s2 = src[e0 + i + 2 * kGThreads],
                 s3 = src[e0 + i + 3 * kGThreads];
        float c0 = contrib_in[s0], c1 = contrib_in[s1], c2 = contrib_in[s2], c3 = contrib_in[s3];
        tile[i] = c0;
        tile[i + kGThreads] = c1;
        tile[i + 2 * kGThreads] = c2;
        tile[i + 3 * kGThreads] = c3;
    }
    for (; i < nnz; i += kGThreads) tile[i] = contrib_in[src[e0 + i]];
    __syncthreads();
    // phase 2: one lane per row adds its LDS segment in order; rows of >= wave_row terms are left to the waves
    double err = 0.0;
    for (uint32_t r = rb.row0 + tid; r < rb.row1; r += kGThreads) {
        const uint32_t a = off[r] - e0, b = off[r + 1] - e0;
        if (b - a >= wave_row) {
            wl.raw[atomicAdd(&wl.n, 1u)] = r - rb.row0;
            continue;
        }
        const float s = lane_row_sum(tile, a, b);
        const uint32_t cr = caller_row(row_id, r);
        err += finish_row(s, cr, scores[cr], out_deg[row_begin + cr], row_begin, contrib_out, scores, base, damping);
    }
    const uint32_t nl = order_wave_rows(wl);
    if (nl) err += wave_rows<kGThreads>(wl, nl, rb, off, e0, tile, out_deg, row_begin, contrib_out, scores, base, damping, row_id);
    const double total = block_sum_f64<kGThreads>(err, red);
    if (tid == 0) partial[blockIdx.x] = total;
}

// ---- hub rows: a row longer than a tile, one workgroup per row -------------------------------------------------------
// The row is gathered tile by tile (two LDS tiles: waves 1..15 gather the next one while wave 0 adds the current one
// to the running sum with exact_sum.h, ~1 cycle per term instead of the ~14 of a one-lane chain).
__global__ void __launch_bounds__(kHThreads)
pr_hub_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ src, const uint32_t *__restrict__ out_deg,
              uint32_t row_begin, const float *__restrict__ contrib_in, float *__restrict__ contrib_out,
              float *__restrict__ scores, float base, float damping, double *__restrict__ partial,
              const uint32_t *__restrict__ row_id) {
    __shared__ __attribute__((aligned(16))) float tiles[2][kHTileNnz];
    const RowBlock rb = blocks[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t e0 = rb.e0, e1 = rb.e1, r = rb.row0;
    {
        const uint32_t nnz = min((uint32_t)kHTileNnz, e1 - e0);
        for (uint32_t i = tid; i < nnz; i += kHThreads) tiles[0][i] = contrib_in[src[e0 + i]];
    }
    __syncthreads();
    float s = 0.0f;
    int cur = 0;
    for (uint32_t t0 = e0; t0 < e1; t0 += kHTileNnz, cur ^= 1) {
        if (tid < 64) {
            s = cz_exact::wave_seq_sum<16>(tiles[cur], min((uint32_t)kHTileNnz, e1 - t0), s);
        } else if (t0 + kHTileNnz < e1) {
            const uint32_t n0 = t0 + kHTileNnz;
            const uint32_t nn = min((uint32_t)kHTileNnz, e1 - n0);
            float *nx = tiles[cur ^ 1];
            constexpr uint32_t G = kHThreads - 64;
            uint32_t i = tid - 64;
            for (; i + 3 * G < nn; i += 4 * G) {
                const uint32_t s0 = src[n0 + i], s1 = src[n0 + i + G], s2 = src[n0 + i + 2 * G], s3 = src[n0 + i + 3 * G];
                const float c0 = contrib_in[s0], c1 = contrib_in[s1], c2 = contrib_in[s2], c3 = contrib_in[s3];
                nx[i] = c0;
                nx[i + G] = c1;
                nx[i + 2 * G] = c2;
                nx[i + 3 * G] = c3;
            }
            for (; i < nn; i += G) nx[i] = contrib_in[src[n0 + i]];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t cr = caller_row(row_id, r);
        partial[blockIdx.x] = finish_row(s, cr, scores[cr], out_deg[row_begin + cr], row_begin, contrib_out, scores, base, damping);
    }
}

// ---- "blocked" formulation ----------------------------------------------------------------------------------
// phase A: val[i] = contrib[slice * W + asrc[i]] for the positions of one work item, slice staged in LDS (W * 4 bytes of
// dynamic LDS: a power of two <= 32768 in the blocked formulation, any multiple of 4 up to kMaxAccSlice in the accumulate one).
// A workgroup step covers 4096 positions: lane t loads 4 local ids (8 bytes) and stores 4 values (16 bytes), so
// that every wave store is one contiguous KiB; the first id vectors are requested before the slice is staged.
__global__ void __launch_bounds__(kAThreads)
pb_expand_kernel(const AItem *__restrict__ items, const uint16_t *__restrict__ asrc,
                 const float *__restrict__ contrib, uint32_t N, uint32_t W, float *__restrict__ val) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    constexpr int PF = 4;
    constexpr uint32_t STEP = kAThreads * 4;
    const AItem it = items[blockIdx.x];
    const uint32_t a0 = (it.begin & ~3u) + threadIdx.x * 4;
    uint2 k[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) {
        const uint32_t i = a0 + j * STEP;
        k[j] = i < it.end ? *(const uint2 *)(asrc + i) : make_uint2(0, 0);
    }
    const uint32_t base = it.slice * W;
    const uint32_t n = min(W, N - base);
    const float *c = contrib + base;
    if ((reinterpret_cast<uintptr_t>(c) & 15) == 0) {
        const uint32_t n4 = n & ~3u;
        for (uint32_t i = threadIdx.x * 4; i < n4; i += kAThreads * 4) *(float4 *)(sl + i) = *(const float4 *)(c + i);
        for (uint32_t i = n4 + threadIdx.x; i < n; i += kAThreads) sl[i] = c[i];
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += kAThreads) sl[i] = c[i];
    }
    __syncthreads();
    for (uint32_t i0 = a0; i0 < it.end; i0 += PF * STEP) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const uint32_t i = i0 + j * STEP;
            if (i < it.end) {
                float4 v;
                v.x = sl[k[j].x & 0xffff];
                v.y = sl[k[j].x >> 16];
                v.z = sl[k[j].y & 0xffff];
                v.w = sl[k[j].y >> 16];
                if (i >= it.begin && i + 4 <= it.end) {
                    *(float4 *)(val + i) = v;
                } else {
                    if (i >= it.begin) val[i] = v.x;
                    if (i + 1 >= it.begin && i + 1 < it.end) val[i + 1] = v.y;
                    if (i + 2 >= it.begin && i + 2 < it.end) val[i + 2] = v.z;
                    if (i + 3 >= it.begin && i + 3 < it.end) val[i + 3] = v.w;
                }
            }
            const uint32_t i2 = i + PF * STEP;
            k[j] = i2 < it.end ? *(const uint2 *)(asrc + i2) : make_uint2(0, 0);
        }
    }
}

// Build-time knobs of phase B (defaults = the measured best; scratch/build_variant.sh builds the other settings for A/B runs)
// (round 3, one box, uniform 10M / 100M sweep: U = 10 early rows unbatched 0.334 ms; U = 20 + late rows 0.348; short rows
// batched 0.349; U = 16 + batched 0.361; U = 20 + late rows + batched 0.384-0.388 -- profiles/r03_pagerank_phase_b_knobs.txt)
#ifndef CZ_PR_U
#define CZ_PR_U 10  // runs requested per wave before the first value is placed
#endif
#ifndef CZ_PR_LATE_ROWS
#define CZ_PR_LATE_ROWS 0  // 1: the rows' own data is requested after the runs (frees its registers while the runs are in flight)
#endif
#ifndef CZ_PR_ROWS_BATCHED
#define CZ_PR_ROWS_BATCHED 0  // 1: short rows: both rows of a lane advance together, eight LDS reads in flight
#endif

// Phase timing (profiling builds only: scratch/build_variant.sh prphase -DCZ_PR_PHASE_TIMING): thread 0 of every phase-B
// workgroup adds the cycles between its stamps to g_pr_phase[]; cz_pagerank_phase_cycles reads / clears the counters.
#ifdef CZ_PR_PHASE_TIMING
__device__ unsigned long long g_pr_phase[8];
#define PR_STAMP(k)                                             \
    do {                                                        \
        if (threadIdx.x == 0) {                                 \
            const unsigned long long now_ = clock64();          \
            atomicAdd(&g_pr_phase[k], now_ - t_prev_);          \
            t_prev_ = now_;                                     \
        }                                                       \
    } while (0)
#define PR_STAMP_INIT unsigned long long t_prev_ = clock64()
#else
#define PR_STAMP(k)
#define PR_STAMP_INIT
#endif

// phase B: row block b gathers its run of every slice's value stream into CSR order inside LDS, then sums rows.
// A run is short (tile / #slices entries), so the kernel lives on loads in flight: 32 waves per CU, 8 runs
// requested per wave before the first value is placed, and the rows' own data (offsets, old score, out-degree)
// requested before the runs so that nothing is fetched after the barrier.
// FLAT: the (row block, slice) runs are too short to be walked one wave-instruction each (a shard of a wide graph:
// 7 values per run at 8 ranks, 57 of 64 lanes idle and the kernel instruction bound).  The plan then also holds, for
// every element of the block in slice-major order, its position in the value stream (`vpos`, 4 more bytes per edge),
// and the tile is filled element by element with every lane busy: tile[perm[e]] = val[vpos[e]].
template <bool FLAT>
__global__ void __launch_bounds__(kBThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
pb_reduce_kernel(const RowBlock *__restrict__ blocks, uint32_t blk0, const uint32_t *__restrict__ off,
                 const uint2 *__restrict__ seg /* [blocks][S+1]: (stream position, block-local prefix) */, uint32_t S,
                 const uint16_t *__restrict__ perm, const uint32_t *__restrict__ vpos, const float *__restrict__ val,
                 const uint32_t *__restrict__ out_deg, uint32_t row_begin, float *__restrict__ contrib_out,
                 float *__restrict__ scores, float base, float damping, double *__restrict__ partial, int xcd_remap,
                 const uint32_t *__restrict__ row_id, uint32_t wave_row) {
    __shared__ __attribute__((aligned(16))) float tile[kBTileNnz];
    __shared__ double red[kBThreads / 64];
    __shared__ WaveRowList wl;
    // runs longer than one wave instruction (a skewed graph: most of a row block's edges come from the few slices that
    // hold the hubs): the first 64 values are placed by the wave that owns the run, the rest is queued here in pieces of
    // <= 64 values and placed afterwards by ALL waves, one piece per wave instruction -- not 64 values at a time by the
    // one wave that owns the slice.  A tile holds 16384 values: at most 256 full pieces plus one partial piece per run
    // longer than 64 (< 256 of those).
    __shared__ uint32_t tail_st[512], tail_p0[512], tail_cnt[512];
    __shared__ uint32_t n_tail;
    PR_STAMP_INIT;
    if (threadIdx.x == 0) {
        n_tail = 0;
        wl.n = 0;
    }
    __syncthreads();
    constexpr int NW = kBThreads / 64;
    constexpr int RPL = kMaxRowsPerBlock / kBThreads;  // rows per lane
    constexpr int U = CZ_PR_U;
    // Workgroup i is dispatched to XCD i % 8, and every XCD has its own L2.  The runs of ADJACENT row blocks are
    // adjacent in each slice's value stream and share their boundary cache lines, so adjacent row blocks are given
    // to the same XCD (workgroups 8j + x, j = 0, 1, ... take a contiguous range of blocks): the shared line is then
    // fetched from memory once instead of once per L2.
    uint32_t local = blockIdx.x;
    if (xcd_remap) {
        const uint32_t nb = gridDim.x, x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        local = x * (nb >> 3) + min(x, nb & 7u) + j;
    }
    const uint32_t b = blk0 + local;
    const RowBlock rb = blocks[b];
    const uint32_t e0 = rb.e0;
    const uint2 *sg = seg + (size_t)b * (S + 1);
    const uint16_t *pm = perm + e0;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t ra[RPL], rz[RPL], od[RPL];
    float old[RPL];
    auto load_rows = [&]() {
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const uint32_t r = rb.row0 + threadIdx.x + j * kBThreads;
            ra[j] = rz[j] = 0;
            if (r < rb.row1) {
                ra[j] = off[r] - e0;
                rz[j] = off[r + 1] - e0;
                const uint32_t cr = caller_row(row_id, r);
                old[j] = scores[cr];
                od[j] = out_deg[row_begin + cr];
            }
        }
    };
    if (FLAT || !CZ_PR_LATE_ROWS) load_rows();
    if constexpr (FLAT) {
        const uint32_t nnz = rb.e1 - e0;
        const uint32_t *vp = vpos + e0;
        constexpr int F = 8;
        for (uint32_t eb = threadIdx.x; eb < nnz; eb += kBThreads * F) {
            uint32_t src[F], q[F];
            float v[F];
#pragma unroll
            for (int i = 0; i < F; i++) {
                const uint32_t e = eb + i * kBThreads;
                src[i] = e < nnz ? vp[e] : 0;
                q[i] = e < nnz ? pm[e] : 0;
            }
#pragma unroll
            for (int i = 0; i < F; i++)
                if (eb + i * kBThreads < nnz) v[i] = val[src[i]];
#pragma unroll
            for (int i = 0; i < F; i++)
                if (eb + i * kBThreads < nnz) tile[q[i]] = v[i];
        }
    } else {
    // first descriptors of this wave: lane l of round g holds the run of slice s = (g*64 + l)*NW + wave
    uint2 d = make_uint2(0, 0);
    uint32_t cnt = 0;
    {
        const uint32_t s = lane * NW + wave;
        if (s < S) {
            d = sg[s];
            cnt = sg[s + 1].y - d.y;
        }
    }
    for (uint32_t g = 0; (g * 64) * NW + wave < S; g++) {
        if (g > 0) {
            const uint32_t s = (g * 64 + lane) * NW + wave;
            d = make_uint2(0, 0);
            cnt = 0;
            if (s < S) {
                d = sg[s];
                cnt = sg[s + 1].y - d.y;
            }
        }
        const uint32_t live = min(64u, (S - (g * 64 * NW + wave) + NW - 1) / NW);  // descriptors held this round
        for (uint32_t t0 = 0; t0 < live; t0 += U) {
            // a run's descriptor is read out of the holding lane where it is needed (three times) rather than kept:
            // U = 20 runs' worth of scalars do not fit the scalar registers
            uint32_t q[U];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t t = min(t0 + u, 63u);
                const uint32_t st = __builtin_amdgcn_readlane(d.x, t), p0 = __builtin_amdgcn_readlane(d.y, t);
                const uint32_t c = t0 + u < live ? __builtin_amdgcn_readlane(cnt, t) : 0;
                if (lane < c) {
                    v[u] = val[st + lane];
                    q[u] = pm[p0 + lane];
                }
            }
            asm volatile("" : "+v"(cnt));  // (keeps the compiler from holding the scalars of the loop above)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = t0 + u < live ? __builtin_amdgcn_readlane(cnt, min(t0 + u, 63u)) : 0;
                if (lane < c) tile[q[u]] = v[u];
            }
            asm volatile("" : "+v"(cnt));
            if (__ballot(cnt > 64) != 0ull) {  // some run of this wave is longer than one wave instruction (rare on a uniform graph)
#pragma unroll 1
                for (uint32_t u = 0; u < (uint32_t)U && t0 + u < live; u++) {
                    const uint32_t t = t0 + u;
                    const uint32_t c = __builtin_amdgcn_readlane(cnt, t);
                    if (c > 64) {  // uniform over the wave
                        const uint32_t st = __builtin_amdgcn_readlane(d.x, t), p0 = __builtin_amdgcn_readlane(d.y, t);
                        const uint32_t rest = c - 64, pieces = (rest + 63) / 64;
                        uint32_t first = 0;
                        if (lane == 0) first = atomicAdd(&n_tail, pieces);
                        first = __builtin_amdgcn_readfirstlane(first);
                        for (uint32_t pc = lane; pc < pieces; pc += 64) {
                            tail_st[first + pc] = st + 64 + pc * 64;
                            tail_p0[first + pc] = p0 + 64 + pc * 64;
                            tail_cnt[first + pc] = min(64u, rest - pc * 64);
                        }
                    }
                }
            }
        }
    }
    if (CZ_PR_LATE_ROWS) load_rows();  // in flight through the barrier and the queued pieces; first used by the row sums
    __syncthreads();
    PR_STAMP(0);  // descriptors + the first 64 values of every run
    // the queued pieces: TU of them in flight per wave (one piece after the other waited a memory round trip each --
    // most of a skewed graph's row block arrives this way)
    const uint32_t nt = n_tail;
    constexpr int TU = 8;
    for (uint32_t i0 = wave; i0 < nt; i0 += NW * TU) {
        float v[TU];
        uint32_t q[TU];
        bool on[TU];
#pragma unroll
        for (int u = 0; u < TU; u++) {
            const uint32_t i = i0 + u * NW;
            on[u] = i < nt && lane < tail_cnt[min(i, 511u)];
            if (on[u]) {
                v[u] = val[tail_st[i] + lane];
                q[u] = pm[tail_p0[i] + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < TU; u++)
            if (on[u]) tile[q[u]] = v[u];
    }
    }
    __syncthreads();
    PR_STAMP(1);  // queued pieces (FLAT: the whole fill)
    double err = 0.0;
    // Rows are summed in order (the reference's sequential f32 sum): one lane per row, and the rows of >= wave_row terms
    // afterwards by a wave each (exact_sum.h: the same bits, without the serial chain a skewed graph's sweep waited for).
#if CZ_PR_ROWS_BATCHED
    {
        // Rows of < 32 terms (all of a uniform graph's): the lane's two rows advance together, eight values each per step,
        // requested before the first is added -- padding reads return +0.0, which adds nothing to a sum that is never -0.0.
        // One LDS read at a time per term (waited for in place) made this phase a quarter of the workgroup's time.
        static_assert(RPL == 2, "two rows per lane");
        const uint32_t len0 = rz[0] - ra[0], len1 = rz[1] - ra[1];
        const bool short0 = len0 < 32, short1 = len1 < 32;
        float s0 = 0.0f, s1 = 0.0f;
        uint32_t steps = max(short0 ? len0 : 0u, short1 ? len1 : 0u);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) steps = max(steps, (uint32_t)__shfl_xor((int)steps, o, 64));
        for (uint32_t k = 0; k < steps; k += 8) {
            float a[8], c[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                a[i] = (short0 && k + i < len0) ? tile[ra[0] + k + i] : 0.0f;
                c[i] = (short1 && k + i < len1) ? tile[ra[1] + k + i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                s0 = s0 + a[i];
                s1 = s1 + c[i];
            }
        }
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const uint32_t r = rb.row0 + threadIdx.x + j * kBThreads;
            if (r < rb.row1) {
                const uint32_t len = rz[j] - ra[j];
                if (len >= wave_row) {
                    wl.raw[atomicAdd(&wl.n, 1u)] = threadIdx.x + j * kBThreads;
                    continue;
                }
                const float s = len < 32 ? (j == 0 ? s0 : s1) : lane_row_sum(tile, ra[j], rz[j]);
                // (the caller's row number is fetched again rather than kept in a register through the tile fill)
                err += finish_row(s, caller_row(row_id, r), old[j], od[j], row_begin, contrib_out, scores, base, damping);
            }
        }
    }
#else
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const uint32_t r = rb.row0 + threadIdx.x + j * kBThreads;
        if (r < rb.row1) {
            if (rz[j] - ra[j] >= wave_row) {
                wl.raw[atomicAdd(&wl.n, 1u)] = threadIdx.x + j * kBThreads;
                continue;
            }
            const float s = lane_row_sum(tile, ra[j], rz[j]);
            // (the caller's row number is fetched again rather than kept in a register through the tile fill)
            err += finish_row(s, caller_row(row_id, r), old[j], od[j], row_begin, contrib_out, scores, base, damping);
        }
    }
#endif
    const uint32_t nl = order_wave_rows(wl);
    PR_STAMP(2);  // one lane per row (until the slowest wave is through)
    if (nl) err += wave_rows<kBThreads>(wl, nl, rb, off, e0, tile, out_deg, row_begin, contrib_out, scores, base, damping, row_id);
    const double total = block_sum_f64<kBThreads>(err, red);
    PR_STAMP(3);  // rows summed by whole waves
    if (threadIdx.x == 0) partial[b] = total;
#ifdef CZ_PR_PHASE_TIMING
    if (threadIdx.x == 0) {
        atomicAdd(&g_pr_phase[4], 1ull);
        atomicAdd(&g_pr_phase[5], (unsigned long long)nl);
        atomicAdd(&g_pr_phase[6], (unsigned long long)n_tail);
    }
#endif
}

// ---- "accumulate" formulation: phase B without a tile -------------------------------------------------------------
// The slices of phase A ascend in source id, and inside a slice the value stream is in CSR order -- (row, source)
// ascending.  So the reference's sum of a row (s = s + contrib[src], sources ascending) is "the row's values of slice 0 in
// stream order, then those of slice 1, ...": the values can be ADDED AS THEY ARRIVE into one f32 accumulator per row and
// nothing has to be put back into CSR order first.  A wave owns a GROUP of consecutive rows (<= 8192 of them, one LDS word
// each) and walks the group's part of every slice's stream (a CELL) in slice order.
//
// What bounds such a walk is not bytes but REQUESTS: the chip retires ~22 G wave-level load instructions per second
// whatever they carry (scratch/cellread_bench.hip: 256-byte requests 5.6 TB/s, 128-byte requests 2.9 TB/s, same rate), and
// a uniform 10M / 100M graph has about a million (group, slice) cells of some hundred values.  The tile formulation's
// phase B and the first two forms of this kernel (a value per lane: two 64-lane requests per 48 values on average) all
// sat on that ceiling at 205-235 us (profiles/r05_pagerank_accumulate.txt).  Hence:
//   * a PIECE is up to 256 stream positions, FOUR consecutive ones per lane: one 16-byte load per lane for the values (1 KiB
//     per request), one 8-byte load for the four u16 that carry the rows' indices inside the group; cells start on
//     multiples of four positions (the stream is padded: < 1 % on the bench graphs) so that both loads are aligned;
//   * few, long-lived waves: 8 per CU (one workgroup, all of the CU's LDS as accumulators), eight pieces in flight each;
//   * a cell is sorted by row, so the values of one row are neighbours in it -- a STRETCH.  A lane adds its four values
//     in order, so a stretch inside one lane needs nothing; a stretch that begins in a lower lane is finished by a DPP
//     wave_shr:1 chain, one step per lane it crosses, lowest lane first (the reference's order).  Where a value stands is
//     static and worked out when the plan is built: the u16 carries the row (13 bits), "last of its stretch inside the
//     piece" (it stores the row's word) and two bits of the number of lanes the stretch of the lane's FIRST value has
//     crossed (six bits over the first three u16 of the lane); the piece's descriptor carries the largest such number, so
//     the kernel runs exactly that many chain steps and detects nothing at run time;
//   * LDS operations of one wave execute in order, so a row that continues in the next piece or the next cell reads
//     what the previous one stored: no barrier anywhere; the waves of a workgroup share nothing but the LDS allocation;
//   * the pieces of a group are ONE list (cells cut at 256, empty cells gone, padded to batches of eight plus one empty
//     batch), read eight descriptors at a time with scalar loads two batches ahead: a counted loop, no cursor.
// The epilogue (new score, |delta|, next contribution) runs over the group's rows, coalesced.
#ifndef CZ_PR_ACC_NT
#define CZ_PR_ACC_NT 1  // the value / row streams of phase B are read once: non-temporal loads
#endif
constexpr int kAccBatch = 8;               // pieces per descriptor batch = pieces in flight per wave of phase B
constexpr int kAccV = 4;                   // stream positions per lane of a piece
constexpr uint32_t kAccPiece = 64 * kAccV;
constexpr int kMaxAccSlice = 40448;        // floats of LDS a phase-A workgroup may stage (158 KiB; one workgroup per CU)
constexpr uint32_t kAccLdsBytes = 161792;  // phase B: accumulators of one CU (158 KiB), split over its workgroups and waves
constexpr uint32_t kAccMaxRows = 8192;     // rows of a group: 13 bits of the u16
constexpr uint32_t kAccTail = 0x2000u;     // u16 bit 13: last value of its stretch inside the piece
constexpr uint32_t kAccHopShift = 14;      // u16 bits 14-15 of a lane's first three values: lanes crossed by the first value's stretch

typedef float acc_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t acc_u2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T acc_stream_load(const T *p) {
#if CZ_PR_ACC_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ float wave_shr1_f32(float v) {  // lane i <- lane i - 1, lane 0 <- 0
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x138, 0xF, 0xF, false));
}

// adds one piece into acc[]: the lane's four values v and annotated rows (rr: four u16); dy = count | most lanes crossed << 16
__device__ __forceinline__ void acc_piece(float *acc, acc_f4 v, acc_u2 rr, uint32_t dy, uint32_t lane) {
    const uint32_t n = dy & 0xffffu, mr = dy >> 16;
    const uint32_t a[4] = {rr.x & 0xffffu, rr.x >> 16, rr.y & 0xffffu, rr.y >> 16};
    uint32_t row[4];
    bool on[4], cont[4];  // cont: the same row as the value before it IN THIS LANE
#pragma unroll
    for (int j = 0; j < 4; j++) {
        row[j] = a[j] & (kAccMaxRows - 1u);
        on[j] = lane * 4 + j < n;
        cont[j] = j > 0 && row[j] == row[j - 1];
    }
    const uint32_t hops = ((a[0] >> kAccHopShift) & 3u) | (((a[1] >> kAccHopShift) & 3u) << 2) | (((a[2] >> kAccHopShift) & 3u) << 4);
    // the stretches that begin in this lane start from the rows' words
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool head = j == 0 ? hops == 0 : !cont[j];
        w[j] = (on[j] && head) ? acc[row[j]] : 0.0f;
    }
    const float vv[4] = {v.x, v.y, v.z, v.w};
    float x[4];
    x[0] = w[0] + vv[0];
#pragma unroll
    for (int j = 1; j < 4; j++) x[j] = (cont[j] ? x[j - 1] : w[j]) + vv[j];
    if (mr != 0) {  // (uniform) some stretch of this piece crosses lanes: lanes at distance k from its first lane join at step k
        const bool c1 = cont[1], c2 = c1 && cont[2], c3 = c2 && cont[3];  // values of the lane's first stretch
        for (uint32_t k = 1; k <= mr; k++) {
            const float y = wave_shr1_f32(x[3]);
            if (hops == k) {
                x[0] = y + vv[0];
                if (c1) x[1] = x[0] + vv[1];
                if (c2) x[2] = x[1] + vv[2];
                if (c3) x[3] = x[2] + vv[3];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (on[j] && (a[j] & kAccTail)) acc[row[j]] = x[j];
}

template <int NW>
__global__ void __launch_bounds__(NW * 64)
pa_reduce_kernel(const uint32_t *__restrict__ grow /* [G + 1]: first plan row of every group */, uint32_t G,
                 const uint32_t *__restrict__ pbase /* [G + 1]: the groups' piece lists inside `piece`, multiples of 8 */,
                 const uint2 *__restrict__ piece /* (stream position, count | most lanes crossed << 16) */,
                 const uint16_t *__restrict__ arow, const float *__restrict__ val, const uint32_t *__restrict__ out_deg,
                 uint32_t row_begin, float *__restrict__ contrib_out, float *__restrict__ scores, float base, float damping,
                 double *__restrict__ partial, const uint32_t *__restrict__ row_id, uint32_t rw) {
    extern __shared__ __attribute__((aligned(16))) float acc_all[];
    __shared__ double red[NW];
    constexpr int U = kAccBatch;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * NW + wave;
    float *acc = acc_all + (size_t)wave * rw;
    double err = 0.0;
#ifdef CZ_PR_PHASE_TIMING
    unsigned long long t_prev_ = clock64();
#define PA_STAMP(k)                                                 \
    do {                                                            \
        if (lane == 0) {                                            \
            const unsigned long long now_ = clock64();              \
            atomicAdd(&g_pr_phase[k], now_ - t_prev_);              \
            t_prev_ = now_;                                         \
        }                                                           \
    } while (0)
#else
#define PA_STAMP(k)
#endif
    if (g < G) {
        const uint32_t r0 = grow[g], nr = grow[g + 1] - r0;
        const uint32_t p0 = pbase[g];
        const uint32_t nb = (pbase[g + 1] - p0) / U - 1;  // batches that hold pieces (the list ends with an empty batch)
        const uint2 *pd = piece + p0;
        // (positions are multiples of 4: the value vector of a lane is 16-byte aligned, its four u16 8-byte aligned)
        const acc_f4 *val4 = reinterpret_cast<const acc_f4 *>(val);
        const acc_u2 *arow4 = reinterpret_cast<const acc_u2 *>(arow);
        // (lanes beyond a piece's end repeat its last lane's address: what follows the piece belongs to another wave, most
        //  often on another XCD, and reading it here as well made the kernel fetch 24 % more than its streams hold)
        auto lane_of = [&](uint32_t dy) { return min(lane, (((dy & 0xffffu) + 3u) >> 2) - ((dy & 0xffffu) != 0u ? 1u : 0u)); };
        // Two batches of eight pieces are in flight (the kernel is bound by the memory round trip, ~4 us under load, times
        // the pieces a wave has to walk, over the pieces it keeps in flight): batch A and batch B take turns, the
        // descriptors of the batch to be requested next (dn) have arrived, those of the one after it (dm) are on their way.
        uint32_t ya[U], yb[U];  // count | lanes crossed of the pieces whose loads are in flight
        uint2 dn[U], dm[U];
        acc_f4 va[U], vb[U];
        acc_u2 ra[U], rb[U];
        const uint32_t last = nb;  // (the empty batch at the end of the list: requests beyond the list read it)
#pragma unroll
        for (int j = 0; j < U; j++) dn[j] = pd[j];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t at = (dn[j].x >> 2) + lane_of(dn[j].y);
            va[j] = acc_stream_load(val4 + at);
            ra[j] = acc_stream_load(arow4 + at);
            ya[j] = dn[j].y;
        }
        {
            const uint2 *p1 = pd + (size_t)min(1u, last) * U;
#pragma unroll
            for (int j = 0; j < U; j++) dn[j] = p1[j];
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t at = (dn[j].x >> 2) + lane_of(dn[j].y);
            vb[j] = acc_stream_load(val4 + at);
            rb[j] = acc_stream_load(arow4 + at);
            yb[j] = dn[j].y;
        }
        {
            const uint2 *p2 = pd + (size_t)min(2u, last) * U;
#pragma unroll
            for (int j = 0; j < U; j++) dn[j] = p2[j];
        }
        for (uint32_t i = lane; i < nr; i += 64) acc[i] = 0.0f;
        PA_STAMP(0);  // descriptors, first requests, accumulators cleared
        for (uint32_t b = 0; b < nb; b += 2) {
            {
                const uint2 *pm = pd + (size_t)min(b + 3, last) * U;
#pragma unroll
                for (int j = 0; j < U; j++) dm[j] = pm[j];
            }
#pragma unroll
            for (int j = 0; j < U; j++) {  // batch b; its registers then take batch b + 2
                acc_piece(acc, va[j], ra[j], ya[j], lane);
                const uint32_t at = (dn[j].x >> 2) + lane_of(dn[j].y);
                va[j] = acc_stream_load(val4 + at);
                ra[j] = acc_stream_load(arow4 + at);
                ya[j] = dn[j].y;
            }
#pragma unroll
            for (int j = 0; j < U; j++) dn[j] = dm[j];
            {
                const uint2 *pm = pd + (size_t)min(b + 4, last) * U;
#pragma unroll
                for (int j = 0; j < U; j++) dm[j] = pm[j];
            }
#pragma unroll
            for (int j = 0; j < U; j++) {  // batch b + 1 (empty when the list holds an odd number of batches); then batch b + 3
                acc_piece(acc, vb[j], rb[j], yb[j], lane);
                const uint32_t at = (dn[j].x >> 2) + lane_of(dn[j].y);
                vb[j] = acc_stream_load(val4 + at);
                rb[j] = acc_stream_load(arow4 + at);
                yb[j] = dn[j].y;
            }
#pragma unroll
            for (int j = 0; j < U; j++) dn[j] = dm[j];
        }
        PA_STAMP(1);  // the stream
        // epilogue: the group's rows, sixteen per lane at a time, the next sixteen requested before these are worked on (a
        // wave has some thousand rows and the CU few waves: four rows per lane at a time, waited for in place, left this
        // part latency-bound -- 60 of the kernel's 207 us).  The caller's row numbers: the test for "rows were moved" stays
        // outside the loop (inside it, it put a branch and a wait for everything in flight in front of every row's loads).
        constexpr int ER = 16;
        auto rows_out = [&](auto moved) {
            uint32_t cr0[ER], od0[ER], cr1[ER], od1[ER];
            float old0[ER], s0[ER], old1[ER], s1[ER];
            auto request = [&](uint32_t ib, uint32_t (&cr)[ER], uint32_t (&od)[ER], float (&old)[ER], float (&sv)[ER]) {
#pragma unroll
                for (int j = 0; j < ER; j++) {
                    const uint32_t i = min(ib + lane + j * 64, nr - 1);  // (rows beyond the end: the last row again, results dropped)
                    cr[j] = decltype(moved)::value ? row_id[r0 + i] : r0 + i;
                    sv[j] = acc[i];
                }
#pragma unroll
                for (int j = 0; j < ER; j++) {
                    old[j] = scores[cr[j]];
                    od[j] = out_deg[row_begin + cr[j]];
                }
            };
            auto finish = [&](uint32_t ib, uint32_t (&cr)[ER], uint32_t (&od)[ER], float (&old)[ER], float (&sv)[ER]) {
#pragma unroll
                for (int j = 0; j < ER; j++)
                    if (ib + lane + j * 64 < nr) err += finish_row(sv[j], cr[j], old[j], od[j], row_begin, contrib_out, scores, base, damping);
            };
            if (nr == 0) return;
            request(0, cr0, od0, old0, s0);
            for (uint32_t ib = 0; ib < nr; ib += 128 * ER) {  // (ib is the same on every lane: the branches are uniform)
                if (ib + 64 * ER < nr) request(ib + 64 * ER, cr1, od1, old1, s1);
                finish(ib, cr0, od0, old0, s0);
                if (ib + 128 * ER < nr) request(ib + 128 * ER, cr0, od0, old0, s0);
                if (ib + 64 * ER < nr) finish(ib + 64 * ER, cr1, od1, old1, s1);
            }
        };
        if (row_id) rows_out(std::true_type{});
        else rows_out(std::false_type{});
        PA_STAMP(2);  // the rows
#ifdef CZ_PR_PHASE_TIMING
        if (lane == 0) atomicAdd(&g_pr_phase[4], 1ull);
#endif
    }
    const double total = block_sum_f64<NW * 64>(err, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// ---- plan construction of the accumulate formulation ---------------------------------------------------------------------
// the row's index inside its group for every in-edge of the groups' rows (CSR order)
__global__ void __launch_bounds__(256)
pa_rowlocal_kernel(const uint32_t *__restrict__ grow, const uint32_t *__restrict__ off, uint16_t *__restrict__ rl) {
    const uint32_t r0 = grow[blockIdx.x], r1 = grow[blockIdx.x + 1];
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const uint32_t a = off[r], z = off[r + 1];
        for (uint32_t e = a; e < z; e++) rl[e] = (uint16_t)(r - r0);
    }
}
// per slice: the stream positions its groups' cells take, every cell rounded up to a multiple of four, and their edges
__global__ void __launch_bounds__(256)
pa_slice_sums_kernel(const uint2 *__restrict__ cell, uint32_t G, uint32_t S, uint32_t *__restrict__ padded_len, uint32_t *__restrict__ edges) {
    __shared__ uint32_t red[2][4];
    const uint32_t s = blockIdx.x;
    uint32_t pl = 0, ed = 0;
    for (uint32_t g = threadIdx.x; g < G; g += 256) {
        const uint32_t c = cell[(size_t)g * (S + 1) + s].y;
        pl += (c + 3u) & ~3u;
        ed += c;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        pl += __shfl_xor(pl, o, 64);
        ed += __shfl_xor(ed, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = pl;
        red[1][threadIdx.x >> 6] = ed;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        padded_len[s] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        edges[s] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
// per slice: the cells' padded positions (an exclusive scan over the groups, from the slice's own start)
__global__ void __launch_bounds__(256)
pa_cell_pos_kernel(const uint2 *__restrict__ cell, uint32_t G, uint32_t S, const uint32_t *__restrict__ slice_start,
                   uint32_t *__restrict__ cellp /* [G][S + 1] */) {
    __shared__ uint32_t wsum[4];
    const uint32_t s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = slice_start[s];
    for (uint32_t g0 = 0; g0 < G; g0 += 256) {
        const uint32_t g = g0 + threadIdx.x;
        const uint32_t c = g < G ? (cell[(size_t)g * (S + 1) + s].y + 3u) & ~3u : 0u;
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += wsum[w];
        const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (g < G) cellp[(size_t)g * (S + 1) + s] = run + before + x - c;
        run += total;
        __syncthreads();
    }
}
// the streams at their padded positions: a wave per cell (local source ids and rows' indices; the padding reads source 0
// of the slice and is never added anywhere), a workgroup per slice for the tile blocks' part behind the cells
__global__ void __launch_bounds__(256)
pa_cell_streams_kernel(const uint2 *__restrict__ cell, const uint32_t *__restrict__ cellp, uint32_t G, uint32_t S,
                       const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ src, const uint16_t *__restrict__ rl,
                       uint32_t W, uint16_t *__restrict__ asrc, uint16_t *__restrict__ arow) {
    const uint64_t ci = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= (uint64_t)G * S) return;
    const uint32_t g = (uint32_t)(ci / S), s = (uint32_t)(ci % S), lane = threadIdx.x & 63;
    const uint2 d = cell[(size_t)g * (S + 1) + s];
    const uint32_t pp = cellp[(size_t)g * (S + 1) + s];
    const uint32_t c4 = (d.y + 3u) & ~3u;
    for (uint32_t k = lane; k < c4; k += 64) {
        uint16_t ls = 0, lr = 0;
        if (k < d.y) {
            const uint32_t e = sidx[d.x + k];
            ls = (uint16_t)(src[e] - s * W);
            lr = rl[e];
        }
        asrc[pp + k] = ls;
        arow[pp + k] = lr;
    }
}
__global__ void __launch_bounds__(256)
pa_tile_streams_kernel(const uint32_t *__restrict__ key_ptr, const uint32_t *__restrict__ slice_edges, const uint32_t *__restrict__ tile_start,
                       const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ src, uint32_t W, uint16_t *__restrict__ asrc) {
    const uint32_t s = blockIdx.x;
    const uint32_t i0 = key_ptr[s] + slice_edges[s], i1 = key_ptr[s + 1], to = tile_start[s];
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256) asrc[to + (i - i0)] = (uint16_t)(src[sidx[i]] - s * W);
}
// the tile blocks' segment table was built in sorted-order positions: move every run to where its slice's tile part went
__global__ void __launch_bounds__(256)
pa_seg_shift_kernel(uint2 *__restrict__ seg, uint32_t n_blocks, uint32_t S, const uint32_t *__restrict__ key_ptr,
                    const uint32_t *__restrict__ slice_edges, const uint32_t *__restrict__ tile_start) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)n_blocks * S) return;
    const uint32_t b = (uint32_t)(t / S), s = (uint32_t)(t % S);
    uint2 *e = seg + (size_t)b * (S + 1) + s;
    e->x = e->x - (key_ptr[s] + slice_edges[s]) + tile_start[s];
}
// how many of the first `e_front` in-edges fall into every slice (the choice of the formulation: a skewed source distribution
// leaves most (group, slice) cells nearly empty, and a piece costs the same whatever it holds)
__global__ void __launch_bounds__(256)
pa_slice_hist_kernel(const uint32_t *__restrict__ src, uint32_t e_front, uint32_t W, uint32_t S, uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t h[];
    for (uint32_t i = threadIdx.x; i < S; i += 256) h[i] = 0;
    __syncthreads();
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < e_front; e += gridDim.x * 256) atomicAdd(&h[src[e] / W], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < S; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}
// pieces per group = sum over its cells of ceil(count / 256), padded to batches of eight + one empty batch
__global__ void __launch_bounds__(64)
pa_piece_count_kernel(const uint2 *__restrict__ cell, uint32_t S, uint32_t *__restrict__ padded) {
    const uint2 *cg = cell + (size_t)blockIdx.x * (S + 1);
    uint32_t n = 0;
    for (uint32_t s = threadIdx.x; s < S; s += 64) n += (cg[s].y + kAccPiece - 1u) / kAccPiece;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
    if (threadIdx.x == 0) padded[blockIdx.x] = ((n + kAccBatch - 1) / kAccBatch + 1) * kAccBatch;
    if (blockIdx.x == 0 && threadIdx.x == 0) padded[gridDim.x] = 0;
}
// one wave per group: the cells' pieces in slice order, then empty pieces up to the padded length
__global__ void __launch_bounds__(64)
pa_piece_list_kernel(const uint2 *__restrict__ cell, const uint32_t *__restrict__ cellp, uint32_t S, const uint32_t *__restrict__ pbase,
                     uint2 *__restrict__ piece) {
    const uint2 *cg = cell + (size_t)blockIdx.x * (S + 1);
    const uint32_t *cp = cellp + (size_t)blockIdx.x * (S + 1);
    const uint32_t p0 = pbase[blockIdx.x], p1 = pbase[blockIdx.x + 1];
    const int lane = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t sl = s0 + lane;
        const uint32_t cnt = sl < S ? cg[sl].y : 0u, pos = sl < S ? cp[sl] : 0u;
        const uint32_t c = (cnt + kAccPiece - 1u) / kAccPiece;
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        const uint32_t at = p0 + run + x - c;
        for (uint32_t k = 0; k < c; k++) piece[at + k] = make_uint2(pos + kAccPiece * k, min(kAccPiece, cnt - kAccPiece * k));
        run += __shfl(x, 63, 64);
    }
    for (uint32_t i = p0 + run + lane; i < p1; i += 64) piece[i] = make_uint2(0, 0);
}
// one wave per piece: where every value stands in its stretch of equal rows inside the piece goes into the u16 beside the
// row index (last of its stretch; lanes crossed by the stretch of the lane's first value), the largest number of lanes
// crossed into the descriptor
__global__ void __launch_bounds__(256)
pa_piece_flags_kernel(uint2 *__restrict__ piece, uint32_t n_pieces, uint16_t *__restrict__ arow) {
    const uint32_t pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= n_pieces) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint2 d = piece[pi];
    const uint32_t n = d.y;
    if (n == 0) return;
    uint32_t row[4];
    bool on[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        on[j] = lane * 4 + j < n;
        row[j] = on[j] ? arow[d.x + lane * 4 + j] : 0xFFFFFFFFu - j;  // (values beyond the end: rows of their own)
    }
    const uint32_t below = __shfl_up(row[3], 1, 64), above = __shfl_down(row[0], 1, 64);
    const bool from_below = lane > 0 && on[0] && row[0] == below;  // the lane's first value continues the lane below
    // a lane the stretch passes THROUGH: it comes from below and all four values are its row
    const bool through = from_below && row[1] == row[0] && row[2] == row[0] && row[3] == row[0];
    const unsigned long long stops = __ballot(!through);
    const unsigned long long lower = stops & ((1ull << lane) - 1ull);  // lanes below this one that the stretch does not pass through
    const uint32_t hops = from_below ? lane - (63u - (uint32_t)__builtin_clzll(lower | 1ull)) : 0u;
    uint32_t mr = hops;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mr = max(mr, (uint32_t)__shfl_xor((int)mr, o, 64));
    const uint32_t n4 = (n + 3u) & ~3u;  // (a cell's padding positions belong to its last piece: they carry hop bits only)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = lane * 4 + j;
        if (idx >= n4) continue;
        const uint32_t hb = j < 3 ? ((hops >> (2 * j)) & 3u) : 0u;
        if (!on[j]) {
            arow[d.x + idx] = (uint16_t)(hb << kAccHopShift);
            continue;
        }
        const uint32_t next = j < 3 ? row[j + 1] : (lane < 63 ? above : 0xFFFFFFF0u);
        const bool tail = idx + 1 == n || next != row[j];
        arow[d.x + idx] = (uint16_t)(row[j] | (tail ? kAccTail : 0u) | (hb << kAccHopShift));
    }
    if (lane == 0) piece[pi].y = n | (mr << 16);
}

// the accumulate formulation adds a row's values slice by slice, which is the CSR order only when every in-list ASCENDS
// (as_directed_graph's CsrLayout::Sorted does); counts the plan rows whose list does not
__global__ void __launch_bounds__(256)
pr_unsorted_rows_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ src, uint32_t rows, uint32_t *__restrict__ bad) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        const uint32_t a = off[r], z = off[r + 1];
        for (uint32_t e = a + 1; e < z; e++)
            if (src[e] < src[e - 1]) {
                atomicAdd(bad, 1u);
                break;
            }
    }
}

// ---- plan construction kernels (run once) -------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pb_keys_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ blk_chunk,
               const uint32_t *__restrict__ off, const uint32_t *__restrict__ src, uint32_t W, uint32_t S,
               uint32_t skip_key, uint32_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const RowBlock rb = blocks[blockIdx.x];
    const uint32_t ch = blk_chunk[blockIdx.x];
    const uint32_t e0 = rb.e0, e1 = rb.e1;
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) {
        keys[e] = ch == CZ_NONE ? skip_key : ch * S + src[e] / W;
        idx[e] = e;
    }
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ a, uint32_t lo, uint32_t hi, uint32_t x) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
pb_keyptr_kernel(const uint32_t *__restrict__ skeys, uint32_t E, uint32_t n_keys, uint32_t *__restrict__ key_ptr) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k <= n_keys) key_ptr[k] = lower_bound_u32(skeys, 0, E, k);
}

__global__ void __launch_bounds__(256)
pb_asrc_kernel(const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ src, uint32_t n, uint32_t W,
               uint16_t *__restrict__ asrc) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t sv = src[sidx[i]];
        asrc[i] = (uint16_t)(sv - (sv / W) * W);
    }
}

// stability of the sort (edge indices ascending inside every key bucket) is what the layout relies on: verify it
__global__ void __launch_bounds__(256)
pb_check_sorted_kernel(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ sidx, uint32_t n,
                       uint32_t *__restrict__ bad) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x + 1; i < n; i += gridDim.x * 256)
        if (skeys[i] == skeys[i - 1] && sidx[i] <= sidx[i - 1]) atomicAdd(bad, 1u);
}

__global__ void __launch_bounds__(256)
pb_seg_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ blk_chunk, uint32_t n_blocks,
              const uint32_t *__restrict__ off, const uint32_t *__restrict__ key_ptr,
              const uint32_t *__restrict__ sidx, uint32_t S, uint2 *__restrict__ seg) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)n_blocks * S) return;
    const uint32_t b = (uint32_t)(t / S), s = (uint32_t)(t % S);
    const RowBlock rb = blocks[b];
    const uint32_t key = blk_chunk[b] * S + s;
    const uint32_t lo0 = key_ptr[key], hi0 = key_ptr[key + 1];
    const uint32_t start = lower_bound_u32(sidx, lo0, hi0, rb.e0);
    const uint32_t end = lower_bound_u32(sidx, start, hi0, rb.e1);
    seg[(size_t)b * (S + 1) + s] = make_uint2(start, end - start);
}

// one wave per block: counts -> exclusive prefix; entry S holds the block's total
__global__ void __launch_bounds__(64)
pb_segscan_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ off, uint32_t S,
                  uint2 *__restrict__ seg, uint32_t *__restrict__ bad) {
    uint2 *sg = seg + (size_t)blockIdx.x * (S + 1);
    const int lane = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t s = s0 + lane;
        const uint32_t c = s < S ? sg[s].y : 0;
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (s < S) sg[s].y = run + x - c;
        run += __shfl(x, 63, 64);
    }
    if (lane == 0) {
        sg[S] = make_uint2(0, run);
        const RowBlock rb = blocks[blockIdx.x];
        if (run != rb.e1 - rb.e0) atomicAdd(bad, 1u);
    }
}

__global__ void __launch_bounds__(256)
pb_perm_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ off, const uint2 *__restrict__ seg,
               uint32_t S, const uint32_t *__restrict__ sidx, uint16_t *__restrict__ perm) {
    const RowBlock rb = blocks[blockIdx.x];
    const uint32_t e0 = rb.e0;
    const uint2 *sg = seg + (size_t)blockIdx.x * (S + 1);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t s = wave; s < S; s += 4) {
        const uint2 d = sg[s];
        const uint32_t cnt = sg[s + 1].y - d.y;
        for (uint32_t k = lane; k < cnt; k += 64) perm[e0 + d.y + k] = (uint16_t)(sidx[d.x + k] - e0);
    }
}

__global__ void __launch_bounds__(256)
pb_vpos_kernel(const RowBlock *__restrict__ blocks, const uint2 *__restrict__ seg, uint32_t S, uint32_t *__restrict__ vpos) {
    const RowBlock rb = blocks[blockIdx.x];
    const uint2 *sg = seg + (size_t)blockIdx.x * (S + 1);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t s = wave; s < S; s += 4) {
        const uint2 d = sg[s];
        const uint32_t cnt = sg[s + 1].y - d.y;
        for (uint32_t k = lane; k < cnt; k += 64) vpos[rb.e0 + d.y + k] = d.x + k;
    }
}

// rows without in-edges (a skewed graph has many): new score = base, no tile, no runs -- they are kept out of the row
// blocks (whose cost per block is fixed: one descriptor per slice) and handled here, 2048 rows per workgroup
constexpr int kERowsPerBlock = 2048;
__global__ void __launch_bounds__(256)
pr_empty_rows_kernel(const uint32_t *__restrict__ row_id, uint32_t r0, uint32_t r1, const uint32_t *__restrict__ out_deg,
                     uint32_t row_begin, float *__restrict__ contrib_out, float *__restrict__ scores, float base, float damping,
                     double *__restrict__ partial) {
    __shared__ double red[256 / 64];
    double err = 0.0;
    const uint32_t b0 = r0 + blockIdx.x * kERowsPerBlock, b1 = min(r1, b0 + kERowsPerBlock);
    for (uint32_t r = b0 + threadIdx.x; r < b1; r += 256) {
        const uint32_t cr = caller_row(row_id, r);
        err += finish_row(0.0f, cr, scores[cr], out_deg[row_begin + cr], row_begin, contrib_out, scores, base, damping);
    }
    const double total = block_sum_f64<256>(err, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// plan construction, skewed graphs: the row order is built on the device (on the host it was 0.1 s of loops over 10M rows
// for a sweep that takes 0.5 ms).  Classes: 0 = stays in the natural order, 1 = heavy, 2 = without in-edges (when those
// are moved); the class flags are scanned into positions, the heavy rows sorted by length (descending, stable).
__global__ void __launch_bounds__(256)
pr_row_class_kernel(const uint32_t *__restrict__ off, uint32_t rows, uint32_t heavy, int drop_empty, uint32_t *__restrict__ is_light,
                    uint32_t *__restrict__ is_heavy, uint32_t *__restrict__ is_empty, uint32_t *__restrict__ counts /* [2] heavy, empty */) {
    uint32_t nh = 0, ne = 0;
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        const uint32_t len = off[r + 1] - off[r];
        const bool h = heavy > 0 && len >= heavy, e = len == 0;
        if (is_light) {
            const bool moved_empty = e && drop_empty;
            is_heavy[r] = h ? 1u : 0u;
            is_empty[r] = moved_empty ? 1u : 0u;
            is_light[r] = (h || moved_empty) ? 0u : 1u;
        }
        nh += h ? 1u : 0u;
        ne += e ? 1u : 0u;
    }
    if (counts) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            nh += __shfl_xor(nh, o, 64);
            ne += __shfl_xor(ne, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            if (nh) atomicAdd(&counts[0], nh);
            if (ne) atomicAdd(&counts[1], ne);
        }
    }
}
// positions of the three classes -> row_id for the light and the empty rows; (inverted length, row) pairs of the heavy ones
__global__ void __launch_bounds__(256)
pr_row_place_kernel(const uint32_t *__restrict__ off, uint32_t rows, const uint32_t *__restrict__ is_heavy,
                    const uint32_t *__restrict__ is_empty, const uint32_t *__restrict__ pos_light, const uint32_t *__restrict__ pos_heavy,
                    const uint32_t *__restrict__ pos_empty, uint32_t n_light, uint32_t n_heavy, uint32_t *__restrict__ row_id,
                    uint32_t *__restrict__ hkey, uint32_t *__restrict__ hrow) {
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        if (is_heavy[r]) {
            hkey[pos_heavy[r]] = ~(off[r + 1] - off[r]);  // ascending key = descending length; the sort is stable: rows ascending
            hrow[pos_heavy[r]] = r;
        } else if (is_empty[r]) {
            row_id[n_light + n_heavy + pos_empty[r]] = r;
        } else {
            row_id[pos_light[r]] = r;
        }
    }
}
__global__ void __launch_bounds__(256)
pr_row_len_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ row_id, uint32_t rows, uint32_t *__restrict__ len) {
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        const uint32_t o = row_id[r];
        len[r] = off[o + 1] - off[o];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) len[rows] = 0;
}

// plan construction, skewed graphs: in-edges of the rows in their new order.  A workgroup owns 256 consecutive new rows
// (their edges are one contiguous stretch of the new array): new offsets of those rows in LDS, every edge finds its row by
// bisection there and copies src[old offset of that row + position in the row].
__global__ void __launch_bounds__(256)
pr_permute_src_kernel(const uint32_t *__restrict__ old_off, const uint32_t *__restrict__ new_off,
                      const uint32_t *__restrict__ row_id, uint32_t rows, const uint32_t *__restrict__ src_in,
                      uint32_t *__restrict__ src_out) {
    __shared__ uint32_t noff[257], ooff[256];
    const uint32_t r0 = blockIdx.x * 256u;
    const uint32_t nr = min(256u, rows - r0);
    if (threadIdx.x < nr) ooff[threadIdx.x] = old_off[row_id[r0 + threadIdx.x]];
    if (threadIdx.x <= nr) noff[threadIdx.x] = new_off[r0 + threadIdx.x];
    if (threadIdx.x == 0 && nr == 256) noff[256] = new_off[r0 + 256];
    __syncthreads();
    const uint32_t e0 = noff[0], e1 = noff[nr];
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) {
        uint32_t lo = 0, hi = nr;  // the last row whose new offset is <= e
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (noff[mid] <= e) lo = mid;
            else hi = mid;
        }
        src_out[e] = src_in[ooff[lo] + (e - noff[lo])];
    }
}

// fixed-order reduction of the per-block partial errors; accumulates into *err_out
__global__ void __launch_bounds__(1024) pr_err_reduce_kernel(const double *__restrict__ partial, uint32_t n,
                                                              double *__restrict__ err_out) {
    __shared__ double red[16];
    double v = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) v += partial[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < 16; w++) s += red[w];
        *err_out += s;
    }
}

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Temporaries of the plan build and of a run come from the device's stream-ordered pool, which keeps what is freed (release
// threshold = never): the build allocates and frees ~2 GB of scratch, and handing that to the driver and back cost 20-50 ms
// per stage whenever the driver actually mapped / unmapped it (CZ_PR_PLAN_TRACE: the same stage took 2.8 or 56 ms).
// Everything here runs on the null stream and is waited for before the buffers go.
void pool_keep_freed_memory() {
    static std::mutex mu;
    static std::vector<int> done;  // devices whose pool has been told (one pool per device; cz_pagerank_multi drives several)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(done.begin(), done.end(), dev) != done.end()) return;
    done.push_back(dev);
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
        uint64_t keep = ~0ull;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
}
template <typename T>
struct PoolBuf {
    T *p = nullptr;
    PoolBuf() = default;
    PoolBuf(const PoolBuf &) = delete;
    PoolBuf &operator=(const PoolBuf &) = delete;
    ~PoolBuf() { reset(); }
    void reset() {
        if (p) (void)hipFreeAsync(p, nullptr);
        p = nullptr;
    }
    hipError_t alloc(size_t count) {
        reset();
        pool_keep_freed_memory();
        hipError_t e = hipMallocAsync((void **)&p, std::max<size_t>(1, count) * sizeof(T), nullptr);
        if (e != hipSuccess) p = nullptr;
        return e;
    }
    T *release() {
        T *r = p;
        p = nullptr;
        return r;
    }
};

// the plan's own arrays come from the same pool (a plan of the 10M / 100M graph holds 1.5 GB: creating and destroying one
// through hipMalloc / hipFree cost tens of milliseconds of driver time)
hipError_t plan_alloc(void **p, size_t bytes) {
    pool_keep_freed_memory();
    return hipMallocAsync(p, std::max<size_t>(1, bytes), nullptr);
}
void plan_free(void *p) {
    if (p) (void)hipFreeAsync(p, nullptr);
}

// CZ_PR_PLAN_TRACE=1: where the plan build's time goes, stage by stage, on stderr (scratch/ experiments)
struct StageTimer {
    bool on = getenv("CZ_PR_PLAN_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[plan] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

}  // namespace

struct cz_pagerank_plan {
    uint32_t N = 0, row_begin = 0, rows = 0;
    uint64_t E = 0;
    float damping = 0, base = 0, init = 0;
    bool blocked = false;
    int xcd_remap = 1;
    // gather formulation: the row blocks of at most one tile; either formulation: the rows longer than a tile (hubs)
    uint32_t n_gblocks = 0, n_hblocks = 0;
    RowBlock *d_gblocks = nullptr, *d_hblocks = nullptr;
    // blocked formulation
    uint32_t slice_w = 0, S = 0, n_chunks = 0, n_bblocks = 0;
    // accumulate formulation (pa_reduce_kernel): plan rows [0, grow[n_groups]) in groups, one wave each
    bool accum = false;
    uint32_t n_groups = 0, e_groups = 0, acc_nw = 16, acc_rw = 0, n_awgs = 0;
    uint32_t *d_grow = nullptr, *d_pbase = nullptr;
    uint2 *d_piece = nullptr;
    uint64_t n_pieces = 0, stream_len = 0;
    uint16_t *d_arow = nullptr;
    RowBlock *d_bblocks = nullptr;
    AItem *d_items = nullptr;
    std::vector<uint32_t> item_ptr, blk_ptr;  // per chunk
    std::vector<uint32_t> val_shift;          // per chunk: stream position that maps to d_val[0] (multiple of 4)
    uint16_t *d_asrc = nullptr, *d_perm = nullptr;
    uint32_t *d_vpos = nullptr;  // flat phase B only (short runs)
    uint2 *d_seg = nullptr;
    float *d_val = nullptr;
    uint64_t E_blocked = 0;
    double h2d_ms = 0, build_ms = 0;  // what creating the plan cost: CSR upload / static layout
    // rows longer than a tile are a handful of workgroups (pr_hub_kernel): they run on a stream of their own beside the
    // sweep of the other rows (both read contrib_in, write disjoint rows), joined before the error sum
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // skewed graphs: rows of >= kHeavyRow in-edges are moved behind the others, longest first, so that the rows of one
    // row block -- and with them the lanes of one wave -- carry similar lengths; row_id[plan row] = the caller's row
    uint32_t *d_rowid = nullptr;
    uint32_t wave_row = kWaveRowDefault;
    uint32_t n_empty = 0, n_eblocks = 0;  // rows without in-edges, moved to the very end (pr_empty_rows_kernel)
    // shared (d_off / d_src: in plan row order)
    uint32_t *d_off = nullptr, *d_src = nullptr, *d_outdeg = nullptr;
    float *d_scores = nullptr;
    double *d_partial = nullptr;
    ~cz_pagerank_plan() {
        void *ps[] = {d_gblocks, d_hblocks, d_bblocks, d_items, d_asrc, d_perm, d_vpos, d_seg, d_val, d_off, d_src, d_outdeg, d_scores,
                      d_partial, d_rowid, d_grow, d_pbase, d_piece, d_arow};
        (void)hipDeviceSynchronize();  // (what hipFree did implicitly: nothing of this plan is in flight any more)
        for (void *p : ps) plan_free(p);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
};

namespace {

// cut [r_begin, rows) into row blocks: consecutive rows whose in-edges fit one tile; a row longer than a tile is alone
int cut_row_blocks(const uint32_t *in_offsets, uint32_t r_begin, uint32_t rows, uint32_t tile, std::vector<RowBlock> &blocks) {
    blocks.clear();
    blocks.reserve((size_t)((in_offsets[rows] - in_offsets[r_begin]) / tile) + (rows - r_begin) / kMaxRowsPerBlock + 16);
    uint32_t r = r_begin;
    while (r < rows) {
        uint32_t r1 = r + 1;
        if (in_offsets[r1] - in_offsets[r] <= tile) {
            const uint32_t lim = std::min<uint32_t>(rows, r + kMaxRowsPerBlock);
            while (r1 < lim && in_offsets[r1 + 1] - in_offsets[r] <= tile) r1++;
        }
        blocks.push_back({r, r1, in_offsets[r], in_offsets[r1]});
        r = r1;
    }
    return CZ_OK;
}

// The accumulate formulation's shape for a shard: NW waves per phase-B workgroup, `per_cu` workgroups per CU, groups of at
// most rw rows; phase A: S slices of W sources (W a multiple of 4, W * 4 bytes of LDS)
struct AccShape {
    uint32_t W = 0, S = 0, NW = 16, per_cu = 1, rw = 0, a_per_cu = 1;
};
int device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}
AccShape acc_shape(uint32_t N) {
    AccShape a;
    const uint32_t cus = (uint32_t)device_cus();
    a.NW = env_int("CZ_PR_ACC_WAVES", 8) == 16 ? 16u : 8u;
    a.per_cu = (uint32_t)std::min(2, std::max(1, env_int("CZ_PR_ACC_PER_CU", 1)));
    a.rw = std::min<uint32_t>(kAccMaxRows, (kAccLdsBytes / a.per_cu / 4 / a.NW) & ~3u);
    a.a_per_cu = (uint32_t)std::min(2, std::max(1, env_int("CZ_PR_ACC_A_PER_CU", 1)));
    const uint32_t wmax = ((uint32_t)kMaxAccSlice / a.a_per_cu) & ~3u;
    // slices: a multiple of the workgroup slots of phase A (one round of workgroups on a balanced graph)
    const uint32_t slots = cus * a.a_per_cu;
    uint32_t k = 1;
    const int s_env = env_int("CZ_PR_ACC_SLICES", 0);
    if (s_env > 0) {
        a.W = std::min<uint32_t>(wmax, std::max<uint32_t>(4, (uint32_t)((((uint64_t)N + s_env - 1) / s_env + 3) & ~3ull)));
    } else {
        while ((((uint64_t)N + (uint64_t)slots * k - 1) / ((uint64_t)slots * k)) > wmax) k++;
        a.W = (uint32_t)(((((uint64_t)N + (uint64_t)slots * k - 1) / ((uint64_t)slots * k)) + 3) & ~3ull);
        a.W = std::max<uint32_t>(a.W, 1024);
    }
    a.S = (uint32_t)(((uint64_t)N + a.W - 1) / a.W);
    return a;
}

// groups of the accumulate formulation: plan rows [0, n_rows) cut at equal shares of their edges, no group above rw rows
void cut_groups(const uint32_t *h_off, uint32_t n_rows, const AccShape &a, std::vector<uint32_t> &grow) {
    grow.assign(1, 0);
    if (n_rows == 0) return;
    const uint32_t cus = (uint32_t)device_cus();
    const uint64_t E = h_off[n_rows];
    const uint64_t per_round = (uint64_t)cus * a.per_cu * a.NW;
    // whole rounds of workgroups; 3 % of head room in rows for groups cut by edges
    uint64_t G = per_round;
    while ((n_rows + G - 1) / G > (uint64_t)a.rw * 97 / 100) G += per_round;
    if (const int g_env = env_int("CZ_PR_ACC_GROUPS", 0)) G = std::max<uint64_t>((uint64_t)g_env, (n_rows + a.rw - 1) / a.rw);
    G = std::min<uint64_t>(G, n_rows);
    uint32_t r = 0;
    for (uint64_t k = 1; k <= G && r < n_rows; k++) {
        uint32_t r1 = k == G ? n_rows : (uint32_t)(std::lower_bound(h_off + r, h_off + n_rows, (uint32_t)(E * k / G)) - h_off);
        r1 = std::min(std::max(r1, r + 1), n_rows);
        while (r1 - r > a.rw) {  // (a stretch of rows with few edges)
            r += a.rw;
            grow.push_back(r);
        }
        r = r1;
        grow.push_back(r);
    }
    if (grow.back() != n_rows) {
        while (n_rows - grow.back() > a.rw) grow.push_back(grow.back() + a.rw);
        grow.push_back(n_rows);
    }
}

// builds the static layout of the streamed formulations on the device; p->d_off / d_src / d_outdeg are resident.
//   blocked:    every row in a tile block (n_group_rows = 0), slices of W = 2^k sources
//   accumulate: plan rows [0, n_group_rows) in groups (pa_reduce_kernel), the rows behind them -- the heavy ones of a skewed
//               graph, longest first -- in tile blocks (pb_reduce_kernel), both fed by one value stream
int build_streamed(cz_pagerank_plan *p, const uint32_t *h_off, uint32_t W, uint32_t n_chunks, uint32_t n_group_rows,
                   const AccShape *shape) {
    const uint32_t rows = p->rows - p->n_empty;  // the rows without in-edges sit behind the others: pr_empty_rows_kernel
    const uint64_t E = p->E;
    StageTimer st;
    n_group_rows = std::min(n_group_rows, rows);
    std::vector<uint32_t> grow;
    if (n_group_rows) cut_groups(h_off, n_group_rows, *shape, grow);
    const uint32_t G = grow.empty() ? 0 : (uint32_t)grow.size() - 1;
    const uint32_t e_groups = n_group_rows ? h_off[n_group_rows] : 0;
    std::vector<RowBlock> all;
    cut_row_blocks(h_off, n_group_rows, rows, kBTileNnz, all);
    st.lap("cut row blocks (host)");
    std::vector<RowBlock> bb, gb;  // blocked / long-row
    uint64_t e_blocked = e_groups;
    for (const RowBlock &rb : all) {
        const uint32_t nnz = rb.e1 - rb.e0;
        if (nnz > (uint32_t)kBTileNnz) gb.push_back(rb);
        else {
            bb.push_back(rb);
            e_blocked += nnz;
        }
    }
    const uint32_t S = (uint32_t)(((uint64_t)p->N + W - 1) / W);
    if (G) n_chunks = 1;
    n_chunks = std::max(1u, std::min<uint32_t>(n_chunks, (uint32_t)std::max<size_t>(1, bb.size())));
    if ((uint64_t)n_chunks * S + 1 >= (1ull << 31)) return cz::set_error(CZ_E_UNSUPPORTED, "too many slices");
    // chunks: consecutive blocked row blocks holding about e_blocked / n_chunks edges each
    std::vector<uint32_t> blk_ptr(1, 0), chunk_of(bb.size());
    {
        uint64_t acc = 0;
        uint32_t c = 0;
        for (size_t i = 0; i < bb.size(); i++) {
            if (c + 1 < n_chunks && i > blk_ptr.back() && acc >= (e_blocked * (c + 1)) / n_chunks) {
                blk_ptr.push_back((uint32_t)i);
                c++;
            }
            chunk_of[i] = c;
            acc += bb[i].e1 - bb[i].e0;
        }
        blk_ptr.push_back((uint32_t)bb.size());
        n_chunks = (uint32_t)blk_ptr.size() - 1;
    }
    p->slice_w = W;
    p->S = S;
    p->n_chunks = n_chunks;
    p->n_bblocks = (uint32_t)bb.size();
    p->n_gblocks = 0;
    p->n_hblocks = (uint32_t)gb.size();
    p->blk_ptr = blk_ptr;
    p->E_blocked = e_blocked;
    p->n_groups = G;
    p->e_groups = e_groups;
    if (G) {
        p->acc_nw = shape->NW;
        p->acc_rw = 0;
        for (uint32_t g = 0; g < G; g++) p->acc_rw = std::max(p->acc_rw, grow[g + 1] - grow[g]);
        p->acc_rw = (p->acc_rw + 3) & ~3u;
        p->n_awgs = (G + shape->NW - 1) / shape->NW;
    }
    const uint32_t n_keys = n_chunks * S;  // key n_keys = "not in the streamed layout" (long rows)
    // key pass works on the block list [groups..., blocked..., long...]; the groups are `key blocks` of chunk 0
    std::vector<RowBlock> key_blocks;
    key_blocks.reserve((size_t)G + bb.size() + gb.size());
    for (uint32_t g = 0; g < G; g++) key_blocks.push_back({grow[g], grow[g + 1], h_off[grow[g]], h_off[grow[g + 1]]});
    key_blocks.insert(key_blocks.end(), bb.begin(), bb.end());
    key_blocks.insert(key_blocks.end(), gb.begin(), gb.end());
    std::vector<uint32_t> key_chunk(G, 0u);
    key_chunk.insert(key_chunk.end(), chunk_of.begin(), chunk_of.end());
    key_chunk.resize(key_blocks.size(), CZ_NONE);

    CZ_HIP(plan_alloc((void **)&p->d_bblocks, std::max<size_t>(1, bb.size()) * sizeof(RowBlock)));
    if (!bb.empty()) CZ_HIP(hipMemcpy(p->d_bblocks, bb.data(), bb.size() * sizeof(RowBlock), hipMemcpyHostToDevice));
    CZ_HIP(plan_alloc((void **)&p->d_hblocks, std::max<size_t>(1, gb.size()) * sizeof(RowBlock)));
    if (!gb.empty()) CZ_HIP(hipMemcpy(p->d_hblocks, gb.data(), gb.size() * sizeof(RowBlock), hipMemcpyHostToDevice));
    if (G) {
        CZ_HIP(plan_alloc((void **)&p->d_grow, ((size_t)G + 1) * 4));
        CZ_HIP(hipMemcpy(p->d_grow, grow.data(), ((size_t)G + 1) * 4, hipMemcpyHostToDevice));
    }

    PoolBuf<RowBlock> d_kblocks;
    PoolBuf<uint32_t> d_kchunk, keys_in, keys_out, idx_in, idx_out, d_keyptr, d_bad;
    CZ_HIP(d_kblocks.alloc(key_blocks.size()));
    CZ_HIP(d_kchunk.alloc(key_blocks.size()));
    CZ_HIP(hipMemcpy(d_kblocks.p, key_blocks.data(), key_blocks.size() * sizeof(RowBlock), hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(d_kchunk.p, key_chunk.data(), key_chunk.size() * 4, hipMemcpyHostToDevice));
    CZ_HIP(keys_in.alloc(E));
    CZ_HIP(keys_out.alloc(E));
    CZ_HIP(idx_in.alloc(E));
    CZ_HIP(idx_out.alloc(E));
    CZ_HIP(d_keyptr.alloc((size_t)n_keys + 2));
    CZ_HIP(d_bad.alloc(1));
    CZ_HIP(hipMemset(d_bad.p, 0, 4));
    hipLaunchKernelGGL(pb_keys_kernel, dim3((uint32_t)key_blocks.size()), dim3(256), 0, nullptr, d_kblocks.p, d_kchunk.p,
                       p->d_off, p->d_src, W, S, n_keys, keys_in.p, idx_in.p);
    unsigned bits = 1;
    while ((1ull << bits) <= n_keys) bits++;
    {   // stable sort of the (key, edge index) pairs by key: csrc/sort_scan.h (own kernels since round 4)
        PoolBuf<uint32_t> d_sort;
        CZ_HIP(d_sort.alloc(czsort::sort_scratch_words(E)));
        bool in_a = true;
        if (int src_rc = czsort::radix_sort_pairs_u32(keys_in.p, idx_in.p, keys_out.p, idx_out.p, E, bits, d_sort.p, nullptr, &in_a)) return src_rc;
        if (in_a) {  // an even number of passes: the result sits in the input arrays
            std::swap(keys_in.p, keys_out.p);
            std::swap(idx_in.p, idx_out.p);
        }
        CZ_HIP(hipStreamSynchronize(nullptr));  // d_sort dies with this scope
    }
    st.lap("keys + radix sort");
    keys_in.reset();
    idx_in.reset();
    hipLaunchKernelGGL(pb_keyptr_kernel, dim3((n_keys + 256) / 256), dim3(256), 0, nullptr, keys_out.p, (uint32_t)E, n_keys,
                       d_keyptr.p);
    std::vector<uint32_t> key_ptr((size_t)n_keys + 1);
    CZ_HIP(hipMemcpy(key_ptr.data(), d_keyptr.p, key_ptr.size() * 4, hipMemcpyDeviceToHost));
    if (key_ptr[n_keys] != e_blocked)
        return cz::set_error(CZ_E_HIP, "streamed PageRank layout: %u edges sorted into slices, expected %llu", key_ptr[n_keys],
                             (unsigned long long)e_blocked);
    const uint32_t EB = (uint32_t)e_blocked;
    hipLaunchKernelGGL(pb_check_sorted_kernel, dim3(2048), dim3(256), 0, nullptr, keys_out.p, idx_out.p, EB, d_bad.p);
    // Stream positions.  Blocked: a slice's edges sit where the sort put them.  Accumulate: every (group, slice) cell is
    // rounded up to a multiple of four positions (aligned 16-byte value loads in phase B), the tile blocks' part of a slice
    // follows its cells, a slice starts on a multiple of eight: pos_ptr[s] = where slice s starts, tile_start[s] = where its
    // tile part starts.
    std::vector<uint32_t> pos_ptr(key_ptr), tile_start, slice_edges;
    PoolBuf<uint2> cell;       // [G][S + 1]: (sorted-order position, count) of every (group, slice) cell
    PoolBuf<uint32_t> cellp;   // [G][S + 1]: its stream position
    PoolBuf<uint32_t> d_slice_edges, d_tile_start, d_slice_start;
    uint64_t stream_len = EB;
    if (G) {
        if (p->acc_rw > kAccMaxRows) return cz::set_error(CZ_E_HIP, "internal: a group of %u rows", p->acc_rw);
        PoolBuf<uint32_t> d_plen;
        CZ_HIP(cell.alloc((size_t)G * ((size_t)S + 1)));
        CZ_HIP(cellp.alloc((size_t)G * ((size_t)S + 1)));
        CZ_HIP(d_plen.alloc(S));
        CZ_HIP(d_slice_edges.alloc(S));
        CZ_HIP(d_tile_start.alloc(S));
        CZ_HIP(d_slice_start.alloc(S));
        const uint64_t pairs = (uint64_t)G * S;
        hipLaunchKernelGGL(pb_seg_kernel, dim3((uint32_t)((pairs + 255) / 256)), dim3(256), 0, nullptr, d_kblocks.p, d_kchunk.p, G,
                           p->d_off, d_keyptr.p, idx_out.p, S, cell.p);
        hipLaunchKernelGGL(pa_slice_sums_kernel, dim3(S), dim3(256), 0, nullptr, cell.p, G, S, d_plen.p, d_slice_edges.p);
        std::vector<uint32_t> plen(S);
        slice_edges.resize(S);
        tile_start.resize(S);
        CZ_HIP(hipMemcpy(plen.data(), d_plen.p, (size_t)S * 4, hipMemcpyDeviceToHost));
        CZ_HIP(hipMemcpy(slice_edges.data(), d_slice_edges.p, (size_t)S * 4, hipMemcpyDeviceToHost));
        uint64_t at = 0;
        for (uint32_t sl = 0; sl < S; sl++) {
            pos_ptr[sl] = (uint32_t)at;
            tile_start[sl] = (uint32_t)(at + plen[sl]);
            at += (uint64_t)plen[sl] + ((key_ptr[sl + 1] - key_ptr[sl]) - slice_edges[sl]);
            at = (at + 7) & ~7ull;
            if (at >= 0xFFFFFF00ull) return cz::set_error(CZ_E_UNSUPPORTED, "the padded value stream exceeds 2^32 positions");
        }
        pos_ptr[S] = (uint32_t)at;
        stream_len = at;
        p->stream_len = at;
        CZ_HIP(hipMemcpy(d_slice_start.p, pos_ptr.data(), (size_t)S * 4, hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(d_tile_start.p, tile_start.data(), (size_t)S * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pa_cell_pos_kernel, dim3(S), dim3(256), 0, nullptr, cell.p, G, S, d_slice_start.p, cellp.p);
    }
    // phase-A streams (padded to a multiple of 8 entries plus one vector so that aligned 16-byte loads stay inside; + a
    // piece's length: phase B of the accumulate formulation reads whole wave instructions at a piece's start)
    const size_t padded = (((size_t)stream_len + 7) & ~(size_t)7) + 8 + kAccPiece;
    CZ_HIP(plan_alloc((void **)&p->d_asrc, padded * 2));
    CZ_HIP(hipMemset(p->d_asrc, 0, padded * 2));
    // value stream: with more than one chunk the chunks can share ONE buffer (each chunk's expand output is
    // consumed by its reduce before the next chunk starts), which keeps it resident in the Infinity Cache
    {
        const bool reuse = n_chunks > 1 && env_int("CZ_PR_VAL_REUSE", 1) != 0;
        p->val_shift.assign(n_chunks, 0);
        size_t need = padded;
        if (reuse) {
            need = 0;
            for (uint32_t c = 0; c < n_chunks; c++) {
                const uint32_t lo = key_ptr[(size_t)c * S] & ~3u, hi = key_ptr[(size_t)(c + 1) * S];
                p->val_shift[c] = lo;
                need = std::max<size_t>(need, (size_t)(hi - lo));
            }
            need = ((need + 7) & ~(size_t)7) + 8;
        }
        CZ_HIP(plan_alloc((void **)&p->d_val, need * 4));
        if (G) CZ_HIP(hipMemsetAsync(p->d_val, 0, need * 4, nullptr));  // (the padding positions behind the last slice are read, never written)
    }
    // the tile blocks' permutation is addressed by CSR position: theirs start behind the groups' edges
    const uint64_t e_tiles = E - e_groups;
    CZ_HIP(plan_alloc((void **)&p->d_perm, std::max<uint64_t>(1, e_tiles) * 2));
    CZ_HIP(plan_alloc((void **)&p->d_seg, std::max<size_t>(1, bb.size()) * ((size_t)S + 1) * sizeof(uint2)));
    if (G) {
        PoolBuf<uint16_t> rl;
        PoolBuf<uint32_t> pcount, d_ss;
        CZ_HIP(plan_alloc((void **)&p->d_arow, padded * 2));
        CZ_HIP(hipMemsetAsync(p->d_arow, 0, padded * 2, nullptr));
        CZ_HIP(rl.alloc(std::max<uint32_t>(1, e_groups)));
        hipLaunchKernelGGL(pa_rowlocal_kernel, dim3(G), dim3(256), 0, nullptr, p->d_grow, p->d_off, rl.p);
        const uint64_t pairs = (uint64_t)G * S;
        hipLaunchKernelGGL(pa_cell_streams_kernel, dim3((uint32_t)((pairs + 3) / 4)), dim3(256), 0, nullptr, cell.p, cellp.p, G, S,
                           idx_out.p, p->d_src, rl.p, W, p->d_asrc, p->d_arow);
        if (!bb.empty())
            hipLaunchKernelGGL(pa_tile_streams_kernel, dim3(S), dim3(256), 0, nullptr, d_keyptr.p, d_slice_edges.p, d_tile_start.p, idx_out.p,
                               p->d_src, W, p->d_asrc);
        // the groups' piece lists
        CZ_HIP(pcount.alloc((size_t)G + 1));
        CZ_HIP(d_ss.alloc(czsort::scan_scratch_words((uint64_t)G + 1)));
        CZ_HIP(plan_alloc((void **)&p->d_pbase, ((size_t)G + 1) * 4));
        hipLaunchKernelGGL(pa_piece_count_kernel, dim3(G), dim3(64), 0, nullptr, cell.p, S, pcount.p);
        if (int src_rc = czsort::exclusive_scan_u32(pcount.p, p->d_pbase, G + 1, d_ss.p, nullptr)) return src_rc;
        uint32_t total = 0;
        CZ_HIP(hipMemcpy(&total, p->d_pbase + G, 4, hipMemcpyDeviceToHost));
        if ((uint64_t)total > (uint64_t)EB + (uint64_t)G * (S + 2 * kAccBatch))
            return cz::set_error(CZ_E_HIP, "internal: %u pieces for %u stream positions", total, EB);
        p->n_pieces = total;
        CZ_HIP(plan_alloc((void **)&p->d_piece, std::max<size_t>(1, total) * sizeof(uint2)));
        hipLaunchKernelGGL(pa_piece_list_kernel, dim3(G), dim3(64), 0, nullptr, cell.p, cellp.p, S, p->d_pbase, p->d_piece);
        if (total) hipLaunchKernelGGL(pa_piece_flags_kernel, dim3((total + 3) / 4), dim3(256), 0, nullptr, p->d_piece, total, p->d_arow);
        CZ_HIP(hipStreamSynchronize(nullptr));  // the temporaries die with this scope
    } else if (EB) {
        hipLaunchKernelGGL(pb_asrc_kernel, dim3(4096), dim3(256), 0, nullptr, idx_out.p, p->d_src, EB, W, p->d_asrc);
    }
    if (!bb.empty()) {
        const uint64_t pairs = (uint64_t)bb.size() * S;
        hipLaunchKernelGGL(pb_seg_kernel, dim3((uint32_t)((pairs + 255) / 256)), dim3(256), 0, nullptr, d_kblocks.p + G, d_kchunk.p + G,
                           (uint32_t)bb.size(), p->d_off, d_keyptr.p, idx_out.p, S, p->d_seg);
        hipLaunchKernelGGL(pb_segscan_kernel, dim3((uint32_t)bb.size()), dim3(64), 0, nullptr, p->d_bblocks, p->d_off, S, p->d_seg,
                           d_bad.p);
        hipLaunchKernelGGL(pb_perm_kernel, dim3((uint32_t)bb.size()), dim3(256), 0, nullptr, p->d_bblocks, p->d_off, p->d_seg, S,
                           idx_out.p, p->d_perm - e_groups);
        // short runs (average tile / #slices below 20 values): phase B fills its tile element by element
        const int flat_env = env_int("CZ_PR_FLAT", -1);
        const bool flat = flat_env >= 0 ? flat_env != 0 : (uint64_t)kBTileNnz < 20ull * S;
        if (flat && n_chunks == 1 && G == 0) {
            CZ_HIP(plan_alloc((void **)&p->d_vpos, std::max<uint64_t>(1, E) * 4));
            hipLaunchKernelGGL(pb_vpos_kernel, dim3((uint32_t)bb.size()), dim3(256), 0, nullptr, p->d_bblocks, p->d_seg, S, p->d_vpos);
        }
        if (G)  // (after the permutation, which reads the sorted order through the table)
            hipLaunchKernelGGL(pa_seg_shift_kernel, dim3((uint32_t)((pairs + 255) / 256)), dim3(256), 0, nullptr, p->d_seg, (uint32_t)bb.size(), S,
                               d_keyptr.p, d_slice_edges.p, d_tile_start.p);
    }
    st.lap("asrc / seg / perm kernels");
    uint32_t bad = 0;
    CZ_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return cz::set_error(CZ_E_HIP, "streamed PageRank layout failed its self-check (%u violations)", bad);
    // phase-A work items: each key bucket cut into parts (cuts on multiples of 8).  Blocked: parts of <= kPartEdges positions.
    // Accumulate: a balanced graph's slices are one item each -- S is a multiple of the workgroup slots, so a sweep stages
    // every contribution once; slices that hold more than their share (a skewed graph's first ones) are cut to a quarter of it.
    uint32_t part_edges = kPartEdges;
    if (G) {
        uint32_t longest = 0;
        for (uint32_t s = 0; s < S; s++) longest = std::max(longest, pos_ptr[s + 1] - pos_ptr[s]);
        const uint64_t share = (stream_len + S - 1) / S;
        part_edges = longest <= share + share / 32 + 64 ? std::max<uint32_t>(longest, 8) : (uint32_t)std::max<uint64_t>(kPartEdges / 4, share / 4);
    }
    std::vector<AItem> items;
    p->item_ptr.assign(1, 0);
    for (uint32_t c = 0; c < n_chunks; c++) {
        for (uint32_t s = 0; s < S; s++) {
            const uint32_t lo = pos_ptr[c * S + s], hi = pos_ptr[c * S + s + 1];
            if (hi == lo) continue;
            const uint32_t parts = (hi - lo + part_edges - 1) / part_edges;
            const uint32_t step = (((hi - lo + parts - 1) / parts) + 7) & ~7u;
            for (uint32_t a = lo; a < hi;) {
                uint32_t b = std::min<uint64_t>(hi, ((uint64_t)a + step) & ~7ull);
                if (b <= a) b = std::min<uint64_t>(hi, (uint64_t)a + step);
                if (hi - b < 64) b = hi;  // (no item of a handful of positions: it would stage a whole slice for them)
                items.push_back({a, b, s, 0});
                a = b;
            }
        }
        p->item_ptr.push_back((uint32_t)items.size());
    }
    CZ_HIP(plan_alloc((void **)&p->d_items, std::max<size_t>(1, items.size()) * sizeof(AItem)));
    if (!items.empty()) CZ_HIP(hipMemcpy(p->d_items, items.data(), items.size() * sizeof(AItem), hipMemcpyHostToDevice));
    CZ_HIP(hipDeviceSynchronize());
    st.lap("phase-A items (host)");
    if (gb.empty()) {  // the global ids are only needed by the hub rows' gather
        plan_free(p->d_src);
        p->d_src = nullptr;
    }
    if (G) {  // the rows' offsets are only needed by the tile blocks
        if (bb.empty() && gb.empty()) {
            plan_free(p->d_off);
            p->d_off = nullptr;
        }
        // more than 64 KiB of dynamic LDS has to be asked for -- per DEVICE (the attribute belongs to the kernel's code object on the
        // current device: cz_pagerank_multi builds one plan per device, each on its own thread), so at every plan build, and checked
        CZ_HIP(hipFuncSetAttribute((const void *)pa_reduce_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAccLdsBytes));
        CZ_HIP(hipFuncSetAttribute((const void *)pa_reduce_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAccLdsBytes));
    }
    CZ_HIP(hipFuncSetAttribute((const void *)pb_expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxAccSlice * 4));
    p->blocked = G == 0;
    p->accum = G != 0;
    // Adjacent row blocks on one XCD share the boundary lines of their runs (1.82 -> 1.58 GB per sweep on the uniform graph).
    // A plan that moved its heavy rows has blocks of very different cost in different parts of the list -- light rows
    // first, then the heavy ones longest first -- and a contiguous range per XCD then leaves some XCDs with nothing but
    // blocks whose long row sums hold their LDS tile while memory idles: R-MAT 10M / 100M 0.614 ms per sweep with ranges
    // of equal count, 0.587 with ranges of equal cost, 0.527 dealt round-robin (profiles/r03_pagerank_rmat.txt).
    p->xcd_remap = env_int("CZ_PR_XCD", p->d_rowid ? 0 : 1);
    return CZ_OK;
}

}  // namespace

extern "C" int cz_pagerank_plan_create(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree,
                                       uint32_t N, uint32_t row_begin, uint32_t row_end, float damping,
                                       cz_pagerank_plan **out, uint32_t flags) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    int rc = cz::ensure_device();
    if (rc) return rc;
    const bool dev = flags & CZ_DEVICE_PTRS;
    std::vector<uint32_t> host_off;
    const uint32_t *dev_off = in_offsets;
    if (dev && in_offsets && row_end >= row_begin && row_end <= N) {  // the row blocks are cut on the host
        host_off.resize((size_t)(row_end - row_begin) + 1);
        CZ_HIP(hipMemcpy(host_off.data(), in_offsets, host_off.size() * 4, hipMemcpyDeviceToHost));
        in_offsets = host_off.data();
    }
    const hipMemcpyKind up = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (row_begin > row_end || row_end > N) return cz::set_error(CZ_E_INVALID, "bad row range [%u,%u) of %u", row_begin, row_end, N);
    if (N > 0 && (!in_offsets || !out_degree)) return cz::set_error(CZ_E_INVALID, "null CSR array");
    const uint32_t rows = row_end - row_begin;
    const uint64_t E = rows ? in_offsets[rows] : 0;
    if (rows && in_offsets[0] != 0) return cz::set_error(CZ_E_INVALID, "in_offsets must be relative to the shard (in_offsets[0] == 0)");
    if (E > 0 && !in_sources) return cz::set_error(CZ_E_INVALID, "null in_sources");
    if (E >= 0xFFFFFFF0ull) return cz::set_error(CZ_E_UNSUPPORTED, "a shard holds at most 2^32-17 edges");
    for (uint32_t q = 0; q < rows; q++)
        if (in_offsets[q + 1] < in_offsets[q]) return cz::set_error(CZ_E_INVALID, "in_offsets not monotone at row %u", q);
    std::unique_ptr<cz_pagerank_plan> p(new cz_pagerank_plan());
    p->N = N;
    p->row_begin = row_begin;
    p->rows = rows;
    p->E = E;
    p->damping = damping;
    p->init = N ? 1.0f / (float)N : 0.f;
    p->base = N ? (1.0f - damping) / (float)N : 0.f;
    const auto t_h2d = std::chrono::steady_clock::now();
    CZ_HIP(plan_alloc((void **)&p->d_off, ((size_t)rows + 1) * 4));
    CZ_HIP(plan_alloc((void **)&p->d_src, std::max<uint64_t>(1, E) * 4));
    CZ_HIP(plan_alloc((void **)&p->d_outdeg, std::max<size_t>(1, N) * 4));
    CZ_HIP(plan_alloc((void **)&p->d_scores, std::max<size_t>(1, rows) * 4));
    if (rows) CZ_HIP(hipMemcpy(p->d_off, dev ? dev_off : in_offsets, ((size_t)rows + 1) * 4, up));
    else {
        uint32_t z = 0;
        CZ_HIP(hipMemcpy(p->d_off, &z, 4, hipMemcpyHostToDevice));
    }
    if (E) CZ_HIP(hipMemcpy(p->d_src, in_sources, E * 4, up));
    if (N) CZ_HIP(hipMemcpy(p->d_outdeg, out_degree, (size_t)N * 4, up));
    CZ_HIP(hipDeviceSynchronize());
    const auto t_build = std::chrono::steady_clock::now();
    p->h2d_ms = std::chrono::duration<double, std::milli>(t_build - t_h2d).count();

    p->wave_row = (uint32_t)std::max<int>((int)kMinWaveRow, env_int("CZ_PR_WAVE_ROW", (int)kWaveRowDefault));
    // Skewed graphs: one lane adds one row, so a wave takes as long as its longest row, and a row block as long as its
    // longest wave.  Rows of >= `heavy` in-edges are taken out of the natural order and appended longest first: the blocks
    // cut from that tail hold rows of similar length (and the few rows worth a whole wave sit together).  The kernels
    // write every result through row_id, so the caller sees its own numbering.
    std::vector<uint32_t> perm_off;  // offsets in plan row order (when rows were moved)
    StageTimer st_plan;
    uint32_t n_front_rows = rows;  // plan rows in their natural order at the front: all of them unless some were moved
    {
        const uint32_t heavy = (uint32_t)std::max(0, env_int("CZ_PR_HEAVY", (int)kHeavyRowDefault));
        PoolBuf<uint32_t> counts;
        uint32_t h_counts[2] = {0, 0};
        if (heavy > 0 && rows > 0) {
            CZ_HIP(counts.alloc(2));
            CZ_HIP(hipMemset(counts.p, 0, 8));
            hipLaunchKernelGGL(pr_row_class_kernel, dim3(2048), dim3(256), 0, nullptr, p->d_off, rows, heavy, 0, (uint32_t *)nullptr,
                               (uint32_t *)nullptr, (uint32_t *)nullptr, counts.p);
            CZ_HIP(hipMemcpy(h_counts, counts.p, 8, hipMemcpyDeviceToHost));
        }
        const uint32_t n_heavy = h_counts[0];
        const bool drop_empty = heavy > 0 && h_counts[1] >= rows / 8 && h_counts[1] > 0;  // a few empty rows stay where they are
        const uint32_t n_empty = drop_empty ? h_counts[1] : 0;
        if ((n_heavy > 0 || n_empty > 0) && n_heavy + n_empty < rows) {
            const uint32_t n_light = rows - n_heavy - n_empty;
            PoolBuf<uint32_t> f_light, f_heavy, f_empty, p_light, p_heavy, p_empty, hkey_in, hkey_out, hrow_in, new_len;
            PoolBuf<uint32_t> new_off, new_src;  // these two become the plan's CSR
            PoolBuf<char> tmp;
            for (PoolBuf<uint32_t> *b3 : {&f_light, &f_heavy, &f_empty, &p_light, &p_heavy, &p_empty}) CZ_HIP(b3->alloc(rows));
            CZ_HIP(hkey_in.alloc(n_heavy));
            CZ_HIP(hkey_out.alloc(n_heavy));
            CZ_HIP(hrow_in.alloc(n_heavy));
            CZ_HIP(new_len.alloc((size_t)rows + 1));
            CZ_HIP(new_off.alloc((size_t)rows + 1));
            CZ_HIP(new_src.alloc(std::max<uint64_t>(1, E)));
            CZ_HIP(plan_alloc((void **)&p->d_rowid, (size_t)rows * 4));
            hipLaunchKernelGGL(pr_row_class_kernel, dim3(2048), dim3(256), 0, nullptr, p->d_off, rows, heavy, drop_empty ? 1 : 0,
                               f_light.p, f_heavy.p, f_empty.p, (uint32_t *)nullptr);
            // scans and the sort of the heavy rows by (inverted length): csrc/sort_scan.h (own kernels since round 4)
            PoolBuf<uint32_t> d_ss, hrow_out;
            CZ_HIP(d_ss.alloc(std::max(czsort::scan_scratch_words((uint64_t)rows + 1), czsort::sort_scratch_words(n_heavy))));
            CZ_HIP(hrow_out.alloc(n_heavy));
            for (auto pr : {std::make_pair(&f_light, &p_light), std::make_pair(&f_heavy, &p_heavy), std::make_pair(&f_empty, &p_empty)})
                if (int src_rc = czsort::exclusive_scan_u32(pr.first->p, pr.second->p, rows, d_ss.p, nullptr)) return src_rc;
            hipLaunchKernelGGL(pr_row_place_kernel, dim3(2048), dim3(256), 0, nullptr, p->d_off, rows, f_heavy.p, f_empty.p, p_light.p,
                               p_heavy.p, p_empty.p, n_light, n_heavy, p->d_rowid, hkey_in.p, hrow_in.p);
            if (n_heavy) {
                bool in_a = true;
                if (int src_rc = czsort::radix_sort_pairs_u32(hkey_in.p, hrow_in.p, hkey_out.p, hrow_out.p, n_heavy, 32u, d_ss.p, nullptr, &in_a)) return src_rc;
                CZ_HIP(hipMemcpyAsync(p->d_rowid + n_light, in_a ? hrow_in.p : hrow_out.p, (size_t)n_heavy * 4, hipMemcpyDeviceToDevice, nullptr));
            }
            hipLaunchKernelGGL(pr_row_len_kernel, dim3(2048), dim3(256), 0, nullptr, p->d_off, p->d_rowid, rows, new_len.p);
            if (int src_rc = czsort::exclusive_scan_u32(new_len.p, new_off.p, rows + 1, d_ss.p, nullptr)) return src_rc;
            hipLaunchKernelGGL(pr_permute_src_kernel, dim3((rows + 255) / 256), dim3(256), 0, nullptr, p->d_off, new_off.p, p->d_rowid,
                               rows, p->d_src, new_src.p);
            perm_off.resize((size_t)rows + 1);
            CZ_HIP(hipMemcpy(perm_off.data(), new_off.p, ((size_t)rows + 1) * 4, hipMemcpyDeviceToHost));  // the row blocks are cut on the host
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank row order: %s", hipGetErrorString(le));
            if (perm_off[rows] != E) return cz::set_error(CZ_E_HIP, "internal: the reordered rows hold %u edges, expected %llu", perm_off[rows], (unsigned long long)E);
            p->n_empty = n_empty;
            p->n_eblocks = (n_empty + kERowsPerBlock - 1) / kERowsPerBlock;
            n_front_rows = n_light;
            plan_free(p->d_off);
            plan_free(p->d_src);
            p->d_off = new_off.release();
            p->d_src = new_src.release();
            in_offsets = perm_off.data();
        }
    }

    st_plan.lap("row order");
    // formulation: explicit flag > CZ_PR_MODE (gather | blocked | accumulate) > heuristic
    int mode = 0;  // 0 auto, 1 gather, 2 blocked, 3 accumulate
    if (flags & CZ_PR_GATHER) mode = 1;
    else if (flags & CZ_PR_BLOCKED) mode = 2;
    else if (flags & CZ_PR_ACCUMULATE) mode = 3;
    else if (const char *m = getenv("CZ_PR_MODE")) mode = !strcmp(m, "gather") ? 1 : !strcmp(m, "blocked") ? 2 : !strcmp(m, "accumulate") ? 3 : 0;
    uint32_t wlog = (uint32_t)std::min(kMaxSliceLog2, std::max(4, env_int("CZ_PR_SLICE_LOG2", kMaxSliceLog2)));
    const uint32_t n_chunks = (uint32_t)std::max(1, env_int("CZ_PR_CHUNKS", 1));
    const AccShape shape = acc_shape(N);
    if (mode == 0) {
        // The blocked layout wins whenever there is enough work to stream, short runs included: measured on one
        // rank's shard of a row-sharded graph (10M rows, 100M in-edges, sources over N = 10M * world nodes), runs of
        // 53 / 27 / 13 / 7 values at world = 1 / 2 / 4 / 8: blocked 0.34 / 0.37 / 0.44 / 0.50 ms per sweep, gather
        // 1.7 / 1.9 / 2.0 / 2.0 ms (adjacent row blocks run on the same XCD, so a short run's cache line is fetched
        // once and shared through that L2).  The segment table (row blocks x slices x 8 bytes) bounds it.
        const uint64_t S = ((uint64_t)N + (1u << wlog) - 1) >> wlog;
        const uint64_t seg_bytes = (E / kBTileNnz + 1) * (S + 1) * 8;
        mode = (E >= (4u << 20) && seg_bytes <= (2ull << 30)) ? 2 : 1;
        // The accumulate formulation walks a (group, slice) cell per wave instruction: it needs cells of some length.
        // One round of workgroups holds cus * per_cu * NW groups; a cell then averages E_front / (S * groups) values.
        // The accumulate formulation streams fewer bytes and no tile, but walks a (group, slice) cell in pieces of 256 positions
        // that cost the same full or empty: it wins when the pieces are well filled (uniform 10M / 100M: 190 of 256, 0.27 ms
        // against 0.33) and loses when the sources are skewed and most cells hold a handful of values (R-MAT: 0.59 against
        // 0.51).  The fill is estimated from the slices' edge counts: a cell of slice s holds about count_s / groups values.
        if (mode == 2 && n_front_rows > 0 && n_chunks == 1 && env_int("CZ_PR_ACC_AUTO", 1) != 0 && shape.S <= 8192) {
            const uint64_t per_round = (uint64_t)device_cus() * shape.per_cu * shape.NW;
            uint64_t G = per_round;
            while ((n_front_rows + G - 1) / G > (uint64_t)shape.rw * 97 / 100) G += per_round;
            const uint32_t e_front = in_offsets[n_front_rows];
            PoolBuf<uint32_t> d_hist;
            CZ_HIP(d_hist.alloc(shape.S));
            CZ_HIP(hipMemsetAsync(d_hist.p, 0, (size_t)shape.S * 4, nullptr));
            hipLaunchKernelGGL(pa_slice_hist_kernel, dim3(1024), dim3(256), shape.S * 4, nullptr, p->d_src, e_front, shape.W, shape.S, d_hist.p);
            std::vector<uint32_t> hist(shape.S);
            CZ_HIP(hipMemcpy(hist.data(), d_hist.p, (size_t)shape.S * 4, hipMemcpyDeviceToHost));
            double pieces = 0;
            for (uint32_t c : hist)
                if (c) pieces += (double)G * std::ceil((double)c / (double)G / (double)kAccPiece);
            const double fill = pieces > 0 ? (double)e_front / (pieces * kAccPiece) : 0.0;
            const char *mf = getenv("CZ_PR_ACC_MIN_FILL");
            if (fill >= (mf ? atof(mf) : 0.55)) mode = 3;
            st_plan.lap("formulation choice (slice histogram)");
        }
    }
    if (mode == 3 && rows > 0 && E > 0 && n_front_rows > 0) {
        //  unsorted in-lists: gather and blocked add in CSR order whatever it is, accumulate in slice order -- the same
        // only for ascending lists.  An explicit request is refused, the heuristic's choice falls back to the blocked formulation.
        PoolBuf<uint32_t> d_bad;
        CZ_HIP(d_bad.alloc(1));
        CZ_HIP(hipMemsetAsync(d_bad.p, 0, 4, nullptr));
        hipLaunchKernelGGL(pr_unsorted_rows_kernel, dim3(2048), dim3(256), 0, nullptr, p->d_off, p->d_src, n_front_rows, d_bad.p);
        uint32_t bad = 0;
        CZ_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
        if (bad) {
            if ((flags & CZ_PR_ACCUMULATE) || getenv("CZ_PR_MODE"))
                return cz::set_error(CZ_E_INVALID, "CZ_PR_ACCUMULATE needs ascending in-lists (CsrLayout::Sorted): %u rows are not", bad);
            mode = 2;
        }
        st_plan.lap("in-list order check");
    }
    if (mode == 3 && rows > 0 && E > 0) {
        rc = build_streamed(p.get(), in_offsets, shape.W, 1, n_front_rows, &shape);
        if (rc) return rc;
    } else if (mode == 2 && rows > 0 && E > 0) {
        rc = build_streamed(p.get(), in_offsets, 1u << wlog, n_chunks, 0, nullptr);
        if (rc) return rc;
    } else {
        std::vector<RowBlock> all, blocks, hubs;
        if (rows > p->n_empty) cut_row_blocks(in_offsets, 0, rows - p->n_empty, kGTileNnz, all);
        for (const RowBlock &rb : all) (rb.e1 - rb.e0 > (uint32_t)kGTileNnz ? hubs : blocks).push_back(rb);
        p->n_gblocks = (uint32_t)blocks.size();
        p->n_hblocks = (uint32_t)hubs.size();
        CZ_HIP(plan_alloc((void **)&p->d_gblocks, std::max<size_t>(1, blocks.size()) * sizeof(RowBlock)));
        if (!blocks.empty()) CZ_HIP(hipMemcpy(p->d_gblocks, blocks.data(), blocks.size() * sizeof(RowBlock), hipMemcpyHostToDevice));
        CZ_HIP(plan_alloc((void **)&p->d_hblocks, std::max<size_t>(1, hubs.size()) * sizeof(RowBlock)));
        if (!hubs.empty()) CZ_HIP(hipMemcpy(p->d_hblocks, hubs.data(), hubs.size() * sizeof(RowBlock), hipMemcpyHostToDevice));
    }
    CZ_HIP(plan_alloc((void **)&p->d_partial, std::max<size_t>(1, (size_t)p->n_awgs + p->n_bblocks + p->n_gblocks + p->n_hblocks + p->n_eblocks) * 8));
    CZ_HIP(hipDeviceSynchronize());
    p->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
    *out = p.release();
    return CZ_OK;
}

extern "C" void cz_pagerank_plan_destroy(cz_pagerank_plan *p) {
    if (!p) return;
    (void)cz::ensure_device();
    delete p;
}

extern "C" int cz_pagerank_plan_init(cz_pagerank_plan *p, float *contrib_dev, void *stream_) {
    if (!p || !contrib_dev) return cz::set_error(CZ_E_INVALID, "null argument");
    int rc = cz::ensure_device();
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (p->N == 0) return CZ_OK;
    hipLaunchKernelGGL(pr_init_kernel, dim3(2048), dim3(256), 0, stream, contrib_dev, p->d_outdeg, p->N, p->init);
    if (p->rows) hipLaunchKernelGGL(pr_fill_kernel, dim3(2048), dim3(256), 0, stream, p->d_scores, p->rows, p->init);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank init launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

extern "C" int cz_pagerank_plan_step(cz_pagerank_plan *p, const float *contrib_in_dev, float *contrib_out_dev,
                                     double *err_out_dev, void *stream_) {
    if (!p || !contrib_in_dev || !contrib_out_dev || !err_out_dev) return cz::set_error(CZ_E_INVALID, "null argument");
    if (contrib_in_dev == contrib_out_dev) return cz::set_error(CZ_E_INVALID, "contrib_in and contrib_out must differ (Jacobi sweep)");
    int rc = cz::ensure_device();
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    // partial errors: [accumulate workgroups | blocked | gather | hub | empty]
    const uint32_t n_partial = p->n_awgs + p->n_bblocks + p->n_gblocks + p->n_hblocks + p->n_eblocks;
    if (n_partial == 0) return CZ_OK;
    const uint32_t n_main = p->n_awgs + p->n_bblocks + p->n_gblocks;
    const bool side_work = p->n_hblocks > 0 || p->n_eblocks > 0;
    const bool fork = side_work && n_main > 0;
    hipStream_t hs = stream;
    if (fork) {  // hub rows and rows without in-edges run beside the sweep of the others, joined before the error sum
        if (!p->side) {
            CZ_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
            CZ_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            CZ_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
        }
        CZ_HIP(hipEventRecord(p->ev_fork, stream));  // contrib_in is ready where the caller's stream stands
        CZ_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
        hs = p->side;
    }
    if (p->n_hblocks)
        hipLaunchKernelGGL(pr_hub_kernel, dim3(p->n_hblocks), dim3(kHThreads), 0, hs, p->d_hblocks, p->d_src, p->d_outdeg,
                           p->row_begin, contrib_in_dev, contrib_out_dev, p->d_scores, p->base, p->damping, p->d_partial + n_main,
                           p->d_rowid);
    if (p->n_eblocks)  // rows without in-edges: nothing to read but their own score
        hipLaunchKernelGGL(pr_empty_rows_kernel, dim3(p->n_eblocks), dim3(256), 0, hs, p->d_rowid, p->rows - p->n_empty, p->rows,
                           p->d_outdeg, p->row_begin, contrib_out_dev, p->d_scores, p->base, p->damping,
                           p->d_partial + n_main + p->n_hblocks);
    if (fork) CZ_HIP(hipEventRecord(p->ev_join, p->side));
    if (p->blocked || p->accum) {
        const uint32_t a_lds = (p->slice_w * 4 + 15) & ~15u;
        double *const bpartial = p->d_partial + p->n_awgs;
        const uint16_t *const tperm = p->d_perm - p->e_groups;  // (the tile blocks' permutation is addressed by CSR position)
        for (uint32_t c = 0; c < p->n_chunks; c++) {
            const uint32_t i0 = p->item_ptr[c], i1 = p->item_ptr[c + 1];
            const uint32_t b0 = p->blk_ptr[c], b1 = p->blk_ptr[c + 1];
            float *val = p->d_val - p->val_shift[c];  // stream position i of this chunk lives at val[i]
            if (i1 > i0)
                hipLaunchKernelGGL(pb_expand_kernel, dim3(i1 - i0), dim3(kAThreads), a_lds, stream, p->d_items + i0, p->d_asrc,
                                   contrib_in_dev, p->N, p->slice_w, val);
            if (p->n_awgs) {
                const uint32_t lds = p->acc_nw * p->acc_rw * 4;
                if (p->acc_nw == 8)
                    hipLaunchKernelGGL((pa_reduce_kernel<8>), dim3(p->n_awgs), dim3(8 * 64), lds, stream, p->d_grow, p->n_groups,
                                       p->d_pbase, p->d_piece, p->d_arow, val, p->d_outdeg, p->row_begin, contrib_out_dev, p->d_scores,
                                       p->base, p->damping, p->d_partial, p->d_rowid, p->acc_rw);
                else
                    hipLaunchKernelGGL((pa_reduce_kernel<16>), dim3(p->n_awgs), dim3(16 * 64), lds, stream, p->d_grow, p->n_groups,
                                       p->d_pbase, p->d_piece, p->d_arow, val, p->d_outdeg, p->row_begin, contrib_out_dev, p->d_scores,
                                       p->base, p->damping, p->d_partial, p->d_rowid, p->acc_rw);
            }
            if (b1 > b0) {
                if (p->d_vpos)
                    hipLaunchKernelGGL(pb_reduce_kernel<true>, dim3(b1 - b0), dim3(kBThreads), 0, stream, p->d_bblocks, b0,
                                       p->d_off, p->d_seg, p->S, tperm, p->d_vpos, val, p->d_outdeg, p->row_begin,
                                       contrib_out_dev, p->d_scores, p->base, p->damping, bpartial, p->xcd_remap, p->d_rowid,
                                       p->wave_row);
                else
                    hipLaunchKernelGGL(pb_reduce_kernel<false>, dim3(b1 - b0), dim3(kBThreads), 0, stream, p->d_bblocks, b0,
                                       p->d_off, p->d_seg, p->S, tperm, p->d_vpos, val, p->d_outdeg, p->row_begin,
                                       contrib_out_dev, p->d_scores, p->base, p->damping, bpartial, p->xcd_remap, p->d_rowid,
                                       p->wave_row);
            }
        }
    } else if (p->n_gblocks) {
        hipLaunchKernelGGL(pr_step_kernel, dim3(p->n_gblocks), dim3(kGThreads), 0, stream, p->d_gblocks, p->d_off, p->d_src,
                           p->d_outdeg, p->row_begin, contrib_in_dev, contrib_out_dev, p->d_scores, p->base, p->damping,
                           p->d_partial, p->d_rowid, p->wave_row);
    }
    if (fork) CZ_HIP(hipStreamWaitEvent(stream, p->ev_join, 0));
    hipLaunchKernelGGL(pr_err_reduce_kernel, dim3(1), dim3(1024), 0, stream, p->d_partial, n_partial, err_out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank step launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

extern "C" float *cz_pagerank_plan_scores(cz_pagerank_plan *p) { return p ? p->d_scores : nullptr; }
extern "C" uint64_t cz_pagerank_plan_edges(const cz_pagerank_plan *p) { return p ? p->E : 0; }
extern "C" uint32_t cz_pagerank_plan_nodes(const cz_pagerank_plan *p) { return p ? p->N : 0; }
extern "C" int cz_pagerank_plan_is_blocked(const cz_pagerank_plan *p) { return p && p->blocked ? 1 : 0; }
extern "C" int cz_pagerank_plan_formulation(const cz_pagerank_plan *p) { return !p ? 0 : p->accum ? 3 : p->blocked ? 2 : 1; }
extern "C" int cz_pagerank_plan_shape(const cz_pagerank_plan *p, uint32_t *out12) {
    if (!p || !out12) return cz::set_error(CZ_E_INVALID, "null argument");
    const uint32_t v[12] = {p->S, p->slice_w, p->n_groups, p->acc_nw, p->acc_rw, p->n_awgs, p->n_bblocks, p->n_hblocks,
                            (uint32_t)p->n_pieces, p->e_groups, (uint32_t)p->stream_len, 0};
    memcpy(out12, v, sizeof(v));
    return CZ_OK;
}

extern "C" int cz_pagerank_plan_read_scores(cz_pagerank_plan *p, float *out, uint32_t flags, void *stream_) {
    if (!p || !out) return cz::set_error(CZ_E_INVALID, "null argument");
    int rc = cz::ensure_device();
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (p->rows == 0) return CZ_OK;
    if (flags & CZ_DEVICE_PTRS) {
        CZ_HIP(hipMemcpyAsync(out, p->d_scores, (size_t)p->rows * 4, hipMemcpyDeviceToDevice, stream));
        return CZ_OK;
    }
    CZ_HIP(hipMemcpyAsync(out, p->d_scores, (size_t)p->rows * 4, hipMemcpyDeviceToHost, stream));
    CZ_HIP(hipStreamSynchronize(stream));
    return CZ_OK;
}

extern "C" int cz_pagerank_plan_timing(const cz_pagerank_plan *p, double *h2d_ms, double *build_ms) {
    if (!p) return cz::set_error(CZ_E_INVALID, "null plan");
    if (h2d_ms) *h2d_ms = p->h2d_ms;
    if (build_ms) *build_ms = p->build_ms;
    return CZ_OK;
}

namespace {

// graph::page_rank's loop on a resident plan: iterate with the reference's stopping rule, scores back to the host
int run_plan(cz_pagerank_plan *plan, double tolerance, uint32_t max_iter, float *scores, uint32_t *iters_run,
             double *final_err, const volatile uint8_t *poison, cz_pagerank_timing *tm) {
    const uint32_t N = plan->N;
    PoolBuf<float> c0, c1;  // (freed in stream order behind the last kernel that uses them)
    PoolBuf<double> derr;
    CZ_HIP(c0.alloc(N));
    CZ_HIP(c1.alloc(N));
    CZ_HIP(derr.alloc(1));
    const auto t0 = std::chrono::steady_clock::now();
    int rc = cz_pagerank_plan_init(plan, c0.p, nullptr);
    if (rc) return rc;
    float *cin = c0.p, *cout = c1.p;
    uint32_t it = 0;
    double err = 0.0;
    for (;;) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        CZ_HIP(hipMemsetAsync(derr.p, 0, 8, nullptr));
        rc = cz_pagerank_plan_step(plan, cin, cout, derr.p, nullptr);
        if (rc) return rc;
        CZ_HIP(hipMemcpy(&err, derr.p, 8, hipMemcpyDeviceToHost));
        std::swap(cin, cout);
        it++;
        if (err < tolerance || it == max_iter) break;
    }
    const auto t1 = std::chrono::steady_clock::now();
    CZ_HIP(hipMemcpy(scores, plan->d_scores, (size_t)N * 4, hipMemcpyDeviceToHost));
    const auto t2 = std::chrono::steady_clock::now();
    if (tm) {
        tm->iterate_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        tm->d2h_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    }
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return CZ_OK;
}

// Plans of relations that are queried again (the same stored relation at the same snapshot): keyed by the caller's
// 128-bit identity of (relation, snapshot) -- the library never looks at the arrays again on a hit, so the key is a
// promise that they are the same.  A plan is handed to ONE caller at a time (it owns scores / value-stream scratch).
struct CacheEntry {
    uint64_t hi, lo;
    uint32_t N;
    uint64_t E;
    float damping;
    uint32_t flags;
    std::unique_ptr<cz_pagerank_plan> plan;
};
std::mutex g_cache_mu;
std::list<CacheEntry> g_cache;  // most recently used first; entries in use are taken out
size_t cache_capacity() { return (size_t)std::max(0, env_int("CZ_PR_CACHE_PLANS", 4)); }

}  // namespace

extern "C" void cz_pagerank_cache_clear(void) {
    (void)cz::ensure_device();
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.clear();
}

extern "C" int cz_pagerank_cached(uint64_t key_hi, uint64_t key_lo, const uint32_t *in_offsets, const uint32_t *in_sources,
                                  const uint32_t *out_degree, uint32_t N, uint64_t E, float damping, double tolerance,
                                  uint32_t max_iter, uint32_t flags, float *scores, uint32_t *iters_run, double *final_err,
                                  const volatile uint8_t *poison, cz_pagerank_timing *timing) {
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (timing) memset(timing, 0, sizeof(*timing));
    if (N == 0) return CZ_OK;  // pagerank.rs:43-45 empty input -> empty output
    if (!scores) return cz::set_error(CZ_E_INVALID, "null scores");
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "iterations must be positive");
    if (flags & CZ_DEVICE_PTRS) return cz::set_error(CZ_E_INVALID, "cz_pagerank takes host arrays (device-resident CSR: the plan API)");
    int rc = cz::ensure_device();
    if (rc) return rc;
    const bool keyed = (key_hi | key_lo) != 0 && cache_capacity() > 0;
    std::unique_ptr<cz_pagerank_plan> plan;
    if (keyed) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
            if (it->hi == key_hi && it->lo == key_lo && it->N == N && it->E == E && it->damping == damping && it->flags == flags) {
                plan = std::move(it->plan);
                g_cache.erase(it);
                break;
            }
    }
    if (plan) {
        if (timing) timing->cache_hit = 1;
    } else {
        if (!in_offsets || !out_degree || (E && !in_sources)) return cz::set_error(CZ_E_INVALID, "null CSR array");
        if (in_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "in_offsets[N] (%u) != E (%llu)", in_offsets[N], (unsigned long long)E);
        cz_pagerank_plan *raw = nullptr;
        rc = cz_pagerank_plan_create(in_offsets, in_sources, out_degree, N, 0, N, damping, &raw, flags);
        if (rc) return rc;
        plan.reset(raw);
        if (timing) {
            timing->h2d_ms = plan->h2d_ms;
            timing->plan_build_ms = plan->build_ms;
        }
    }
    rc = run_plan(plan.get(), tolerance, max_iter, scores, iters_run, final_err, poison, timing);
    if (keyed && (rc == CZ_OK || rc == CZ_E_CANCELLED)) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_cache.push_front(CacheEntry{key_hi, key_lo, N, E, damping, flags, std::move(plan)});
        while (g_cache.size() > cache_capacity()) g_cache.pop_back();
    }
    return rc;
}

// one-shot form: upload, iterate with the reference's stopping rule, download
extern "C" int cz_pagerank(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                           uint64_t E, float damping, double tolerance, uint32_t max_iter, float *scores,
                           uint32_t *iters_run, double *final_err, const volatile uint8_t *poison) {
    return cz_pagerank_cached(0, 0, in_offsets, in_sources, out_degree, N, E, damping, tolerance, max_iter, 0, scores, iters_run,
                              final_err, poison, nullptr);
}

#ifdef CZ_PR_PHASE_TIMING
// profiling builds only: {fill, queued pieces, lane rows, wave rows} cycles of thread 0 summed over the workgroups,
// then workgroups, wave rows, queued pieces counted
extern "C" int cz_pagerank_phase_cycles(unsigned long long *out8, int reset) {
    if (out8) CZ_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pr_phase), 64));
    if (reset) {
        unsigned long long z[8] = {0};
        CZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pr_phase), z, 64));
    }
    return CZ_OK;
}
#endif
