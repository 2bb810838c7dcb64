// sort_scan.h -- the two device primitives the PageRank plan build needs, hand-written for gfx950 (round 4: they replace the
// rocprim::exclusive_scan / rocprim::radix_sort_pairs calls the plan build carried since round 1; the sweep never used a library).
//
//   exclusive_scan_u32   tiles of 1 024, recursive on the tile sums (the same scheme as graph.hip's)
//   radix_sort_pairs_u32 STABLE least-significant-digit radix sort of (u32 key, u32 value) pairs, 8 bits per pass.  Per pass:
//       histogram   a workgroup counts the digits of its tile of 4 096 pairs in LDS and writes them digit-major;
//       scan        one exclusive scan over the [256][tiles] counts gives every (digit, tile) its place;
//       scatter     the workgroup walks its tile 256 pairs at a time IN ORDER: the lanes of a wave that hold the same digit find
//                   each other with eight ballots (rank inside the wave = earlier lanes of the same digit), the waves' counts
//                   meet in LDS (rank inside the round = earlier waves), a running count per digit carries over the rounds --
//                   so equal digits keep their order, which is what makes the passes compose (and what the plan build relies
//                   on: inside a (chunk, slice) key the edges must stay in (row, source) order).
//     Writes of one digit from one tile land next to each other (16 pairs per digit and tile on average) and are merged by the
//     L2 before they reach HBM.  100M pairs, 16 key bits: see profiles/ (plan build).
// Everything is stream-ordered on `s`; scratch comes from the caller.
#pragma once
#include "common.h"

namespace czsort {
namespace {

constexpr int kT = 256;
constexpr int kScanTile = 1024;
constexpr int kSortTile = 4096;  // pairs per workgroup and pass

__global__ void __launch_bounds__(kT)
scan_tiles_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t wsum[kT / 64];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = base + j < n ? in[base + j] : 0u;
    const uint32_t t = v[0] + v[1] + v[2] + v[3];
    uint32_t x = t;  // inclusive scan across the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if ((int)(threadIdx.x & 63) >= off) x += y;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    uint32_t excl = woff + x - t;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) out[base + j] = excl;
        excl += v[j];
    }
    if (threadIdx.x == kT - 1) sums[blockIdx.x] = woff + x;
}

__global__ void __launch_bounds__(kT) scan_add_kernel(uint32_t *__restrict__ out, uint32_t n, const uint32_t *__restrict__ tile_off) {
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    const uint32_t o = tile_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < n) out[base + j] += o;
}

// digit-major counts: hist[d * tiles + tile]
__global__ void __launch_bounds__(kT)
sort_hist_kernel(const uint32_t *__restrict__ keys, uint64_t n, uint32_t shift, uint32_t tiles, uint32_t *__restrict__ hist) {
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int r = 0; r < kSortTile / kT; r++) {
        const uint64_t i = base + (uint64_t)r * kT + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(kT)
sort_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t n, uint32_t shift, uint32_t tiles,
                    const uint32_t *__restrict__ place /* scanned hist */, uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t run[256];                // where the next pair of digit d from this tile goes
    __shared__ uint32_t wcnt[kT / 64][256];      // this round's count per wave and digit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    run[threadIdx.x] = place[(size_t)threadIdx.x * tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kT / 64; w++) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / kT; r++) {
        const uint64_t i = base + (uint64_t)r * kT + threadIdx.x;
        if (base + (uint64_t)r * kT >= n) break;  // uniform
        const bool live = i < n;
        const uint32_t k = live ? keys_in[i] : 0u, v = live ? vals_in[i] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        // the lanes of this wave that hold digit d (dead lanes match nobody)
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot(live && ((d >> b) & 1u));
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t below = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (live && below == 0) wcnt[wave][d] = (uint32_t)__popcll(same);  // the first lane of a digit speaks for all of them
        __syncthreads();
        if (live) {
            uint32_t pos = run[d] + below;
            for (int w = 0; w < wave; w++) pos += wcnt[w][d];
            keys_out[pos] = k;
            vals_out[pos] = v;
        }
        __syncthreads();
        {   // thread d: the round is done for digit d
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < kT / 64; w++) {
                t += wcnt[w][threadIdx.x];
                wcnt[w][threadIdx.x] = 0;
            }
            run[threadIdx.x] += t;
        }
        __syncthreads();
    }
}

}  // namespace

inline size_t scan_scratch_words(uint64_t n) {
    size_t w = 0;
    while (n > 1) {
        n = (n + kScanTile - 1) / kScanTile;
        w += 2 * n + 2;
        if (n == 1) break;
    }
    return w + 8;
}

// out[i] = in[0] + ... + in[i - 1]; in == out is fine.  scratch: scan_scratch_words(n) words.
inline int exclusive_scan_u32(const uint32_t *d_in, uint32_t *d_out, uint32_t n, uint32_t *scratch, hipStream_t s) {
    if (n == 0) return CZ_OK;
    const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
    uint32_t *sums = scratch, *sums_scan = scratch + tiles + 1;
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(kT), 0, s, d_in, d_out, n, sums);
    if (tiles == 1) return CZ_OK;
    const int rc = exclusive_scan_u32(sums, sums_scan, tiles, scratch + 2 * (tiles + 1), s);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_add_kernel, dim3(tiles), dim3(kT), 0, s, d_out, n, sums_scan);
    return CZ_OK;
}

inline size_t sort_scratch_words(uint64_t n) {
    const uint64_t tiles = (n + kSortTile - 1) / kSortTile;
    return (size_t)(256 * tiles) + scan_scratch_words(256 * tiles);
}

// Stable sort of n pairs by the low `bits` bits of the key (keys must be < 2^bits).  The pairs ping-pong between (keys_a, vals_a)
// and (keys_b, vals_b); returns in *in_a whether the result ended up in the a arrays.  scratch: sort_scratch_words(n) words.
inline int radix_sort_pairs_u32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint64_t n, unsigned bits,
                                uint32_t *scratch, hipStream_t s, bool *in_a) {
    *in_a = true;
    if (n == 0) return CZ_OK;
    const uint64_t tiles64 = (n + kSortTile - 1) / kSortTile;
    if (256 * tiles64 >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "too many pairs for one sort");
    const uint32_t tiles = (uint32_t)tiles64;
    uint32_t *hist = scratch, *scan_scratch = scratch + (size_t)256 * tiles;
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    for (unsigned shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(tiles), dim3(kT), 0, s, ki, n, shift, tiles, hist);
        const int rc = exclusive_scan_u32(hist, hist, 256u * tiles, scan_scratch, s);
        if (rc) return rc;
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(tiles), dim3(kT), 0, s, ki, vi, n, shift, tiles, hist, ko, vo);
        std::swap(ki, ko);
        std::swap(vi, vo);
        *in_a = !*in_a;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "radix sort launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

}  // namespace czsort
