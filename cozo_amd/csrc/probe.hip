// probe.hip -- what THIS box's HBM delivers, measured with the access patterns of the path ("report both nominal
// and measured ceilings").  Two hand-written gfx950 kernels, no arithmetic worth the name:
//   stream   every lane reads 16 bytes of a contiguous buffer per instruction, grid-stride, non-temporal (what PageRank's
//            phase A / a brute-force scan does);
//   rows     a wave fetches whole `row_bytes`-byte rows at pseudo-random row numbers of a table, 4 rows in flight, non-temporal
//            (what hnsw_knn_kernel / distance_pairs_kernel do with 3 KiB vectors) -- over the CALLER's table, i.e. with its
//            size, its allocation and its translation footprint.
//   words    every lane has 8 independent accesses in flight to pseudo-random words of a per-node array, as plain loads and as
//            atomicMin without a returned value (what BFS / SSSP / LabelPropagation do to their per-node words).
// The roofline fractions in bench.py stay priced against the nominal 8 TB/s; these numbers say how much of a box-to-box
// difference is the box (the same binary measured 0.64-0.76 of the nominal peak on different GPUs).
#include "common.h"
#include "exact_sum.h"
#include "sort_scan.h"
#include "hnsw_index.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) probe_stream_kernel(const f4 *__restrict__ p, uint64_t n16, float *__restrict__ sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {  // four independent 16-byte loads per lane in flight
        const f4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        const f4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += (a + b) + (c + d);
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;  // keeps the loads alive, never true in practice
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <int U>
__global__ void __launch_bounds__(256) probe_rows_kernel(const char *__restrict__ base, uint64_t rows, uint32_t row_bytes, uint64_t n_fetch,
                                                         uint64_t seed, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * 256) >> 6;
    const uint32_t pieces = (row_bytes + 1023) / 1024;  // a wave instruction moves 1 KiB
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t r0 = wave * U; r0 < n_fetch; r0 += n_waves * U) {
        const f4 *p[U];
#pragma unroll
        for (int u = 0; u < U; u++) p[u] = (const f4 *)(base + (mix64(seed + r0 + u) % rows) * (uint64_t)row_bytes);
        for (uint32_t j = 0; j < pieces; j++) {
            const uint32_t c = j * 64 + lane;
            if (c * 16 < row_bytes) {
                f4 v[U];
#pragma unroll
                for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p[u] + c);
#pragma unroll
                for (int u = 0; u < U; u++) acc += v[u];
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// random single words of a per-node array (what the traversal rules do to depth / claim / label / (cost, parent) words): every lane
// U independent accesses in flight at pseudo-random indices; plain loads, or atomicMin without a returned value
template <typename W, int U, bool ATOMIC>
__global__ void __launch_bounds__(256) probe_words_kernel(W *__restrict__ words, uint64_t n_words, uint64_t n_access, uint64_t seed,
                                                          unsigned long long *__restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, n_threads = (uint64_t)gridDim.x * 256;
    unsigned long long acc = 0;
    for (uint64_t a0 = t * U; a0 < n_access; a0 += n_threads * U) {
        uint64_t idx[U];
#pragma unroll
        for (int u = 0; u < U; u++) idx[u] = mix64(seed + a0 + u) % n_words;
        if constexpr (ATOMIC) {
#pragma unroll
            for (int u = 0; u < U; u++) atomicMin(words + idx[u], (W)(mix64(idx[u] + a0) | 1u));
        } else {
            W v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = words[idx[u]];
#pragma unroll
            for (int u = 0; u < U; u++) acc += (unsigned long long)v[u];
        }
    }
    if (acc == 0x123456789abcdefull) sink[0] = acc;
}

}  // namespace

// What this box sustains on the traversal rules' access pattern: G accesses / s of independent random loads and of random
// atomicMin (no return) over a per-node array of n_words words of 4 or 8 bytes (allocated here, filled with 0xFF).
extern "C" int cz_random_access_probe(uint64_t n_words, uint32_t word_bytes, uint64_t n_access, uint32_t reps, double *loads_g_per_s,
                                      double *atomic_min_g_per_s) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (loads_g_per_s) *loads_g_per_s = 0.0;
    if (atomic_min_g_per_s) *atomic_min_g_per_s = 0.0;
    if (word_bytes != 4 && word_bytes != 8) return cz::set_error(CZ_E_INVALID, "word_bytes must be 4 or 8");
    if (n_words == 0) return cz::set_error(CZ_E_INVALID, "empty array");
    if (n_access == 0) n_access = 256u << 20;
    if (reps == 0) reps = 3;
    cz::DevBuf<char> arr;
    cz::DevBuf<unsigned long long> sink;
    CZ_HIP(arr.alloc(n_words * word_bytes));
    CZ_HIP(sink.alloc(1));
    CZ_HIP(hipMemset(arr.p, 0xFF, n_words * word_bytes));
    hipEvent_t e0, e1;
    CZ_HIP(hipEventCreate(&e0));
    CZ_HIP(hipEventCreate(&e1));
    const dim3 grid(256 * 32), block(256);
    for (int pass = 0; pass < 2; pass++) {  // 0: loads, 1: atomicMin
        float ms = 0.f;
        for (uint32_t i = 0; i <= reps; i++) {  // (the first launch warms up)
            if (i == 1) CZ_HIP(hipEventRecord(e0, nullptr));
            const uint64_t seed = 4242ull + (uint64_t)i * n_access;
            if (word_bytes == 4) {
                if (pass == 0) hipLaunchKernelGGL((probe_words_kernel<uint32_t, 8, false>), grid, block, 0, nullptr, (uint32_t *)arr.p, n_words, n_access, seed, sink.p);
                else hipLaunchKernelGGL((probe_words_kernel<uint32_t, 8, true>), grid, block, 0, nullptr, (uint32_t *)arr.p, n_words, n_access, seed, sink.p);
            } else {
                if (pass == 0) hipLaunchKernelGGL((probe_words_kernel<unsigned long long, 8, false>), grid, block, 0, nullptr, (unsigned long long *)arr.p, n_words, n_access, seed, sink.p);
                else hipLaunchKernelGGL((probe_words_kernel<unsigned long long, 8, true>), grid, block, 0, nullptr, (unsigned long long *)arr.p, n_words, n_access, seed, sink.p);
            }
        }
        CZ_HIP(hipEventRecord(e1, nullptr));
        CZ_HIP(hipEventSynchronize(e1));
        CZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        double *dst = pass == 0 ? loads_g_per_s : atomic_min_g_per_s;
        if (dst && ms > 0.f) *dst = (double)n_access * reps / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "probe launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

extern "C" int cz_hbm_probe(const void *table, uint64_t rows, uint32_t row_bytes, uint64_t n_fetch, uint32_t reps, double *stream_gbs,
                            double *row_fetch_gbs) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (stream_gbs) *stream_gbs = 0.0;
    if (row_fetch_gbs) *row_fetch_gbs = 0.0;
    if (row_bytes == 0 || row_bytes % 16 != 0) return cz::set_error(CZ_E_INVALID, "row_bytes must be a positive multiple of 16");
    if (reps == 0) reps = 5;
    cz::DevBuf<char> own;
    cz::DevBuf<float> sink;
    CZ_HIP(sink.alloc(4));
    const char *base = (const char *)table;
    if (!base) {  // no table of the caller's: 4 GiB of our own (contents irrelevant, but written once so that pages exist)
        rows = (4ull << 30) / row_bytes;
        CZ_HIP(own.alloc(rows * row_bytes));
        CZ_HIP(hipMemset(own.p, 1, rows * row_bytes));
        base = own.p;
    }
    if (rows == 0) return cz::set_error(CZ_E_INVALID, "empty table");
    if (n_fetch == 0) n_fetch = 4u << 20;
    hipEvent_t e0, e1;
    CZ_HIP(hipEventCreate(&e0));
    CZ_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    // (a) stream: at most 8 GiB of the table, contiguous
    const uint64_t sbytes = std::min<uint64_t>(rows * (uint64_t)row_bytes, 8ull << 30) & ~15ull;
    for (uint32_t i = 0; i < 2; i++) hipLaunchKernelGGL(probe_stream_kernel, dim3(256 * 16), dim3(256), 0, nullptr, (const f4 *)base, sbytes / 16, sink.p);
    CZ_HIP(hipEventRecord(e0, nullptr));
    for (uint32_t i = 0; i < reps; i++) hipLaunchKernelGGL(probe_stream_kernel, dim3(256 * 16), dim3(256), 0, nullptr, (const f4 *)base, sbytes / 16, sink.p);
    CZ_HIP(hipEventRecord(e1, nullptr));
    CZ_HIP(hipEventSynchronize(e1));
    CZ_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (stream_gbs && ms > 0.f) *stream_gbs = (double)sbytes * reps / (ms * 1e-3) / 1e9;
    // (b) random whole rows
    for (uint32_t i = 0; i < 2; i++) hipLaunchKernelGGL(probe_rows_kernel<4>, dim3(2048), dim3(256), 0, nullptr, base, rows, row_bytes, n_fetch, 77ull + i, sink.p);
    CZ_HIP(hipEventRecord(e0, nullptr));
    for (uint32_t i = 0; i < reps; i++) hipLaunchKernelGGL(probe_rows_kernel<4>, dim3(2048), dim3(256), 0, nullptr, base, rows, row_bytes, n_fetch, 1000ull + i * n_fetch, sink.p);
    CZ_HIP(hipEventRecord(e1, nullptr));
    CZ_HIP(hipEventSynchronize(e1));
    CZ_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (row_fetch_gbs && ms > 0.f) *row_fetch_gbs = (double)n_fetch * row_bytes * reps / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "probe launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

// ---- test hook: exact_sum.h's wave procedure on arbitrary rows (tests/test_gpu_graph.py) -------------------------------------
// PageRank only ever feeds it non-negative finite terms; the paths it maps "past the binade" (negative terms, inf / nan, denormal
// running sums, a term above the sum's exponent, a negative or non-finite start) are reached through this entry alone.
namespace {

template <int LANES, int T>
__global__ void __launch_bounds__(64) debug_seq_sum_kernel(const float *__restrict__ terms, const unsigned long long *__restrict__ off,
                                                           const float *__restrict__ init, uint32_t n_rows, float *__restrict__ out) {
    constexpr uint32_t per_wave = 64 / LANES;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = blockIdx.x * per_wave + lane / LANES;
    const bool have = row < n_rows;
    const unsigned long long b = have ? off[row] : 0ull, e = have ? off[row + 1] : 0ull;
    const float s = cz_exact::group_seq_sum<LANES, T>(terms + b, (uint32_t)(e - b), have ? init[row] : 0.f);
    if (have && (lane & (LANES - 1)) == 0) out[row] = s;
}

}  // namespace

extern "C" int cz_debug_seq_sum(const float *terms, const uint64_t *row_off, const float *init, uint32_t n_rows, int lanes, int per_lane,
                                float *out) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (n_rows == 0) return CZ_OK;
    if (!row_off || !init || !out) return cz::set_error(CZ_E_INVALID, "null buffer");
    const uint64_t total = row_off[n_rows];
    cz::DevBuf<float> d_t, d_i, d_o;
    cz::DevBuf<unsigned long long> d_off;
    CZ_HIP(d_t.alloc(total + 4));  // (a row's last 16-byte vector may reach past its end: exact_sum.h masks the lanes, the bytes must exist)
    CZ_HIP(d_i.alloc(n_rows));
    CZ_HIP(d_o.alloc(n_rows));
    CZ_HIP(d_off.alloc((size_t)n_rows + 1));
    CZ_HIP(hipMemset(d_t.p, 0, (total + 4) * 4));
    if (total) CZ_HIP(hipMemcpy(d_t.p, terms, total * 4, hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(d_i.p, init, (size_t)n_rows * 4, hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(d_off.p, row_off, ((size_t)n_rows + 1) * 8, hipMemcpyHostToDevice));
#define CZ_DBG(L, T_)                                                                                                              \
    hipLaunchKernelGGL((debug_seq_sum_kernel<L, T_>), dim3((n_rows + 64 / L - 1) / (64 / L)), dim3(64), 0, nullptr, d_t.p, d_off.p, d_i.p, \
                       n_rows, d_o.p)
    if (lanes == 64 && per_lane == 16) CZ_DBG(64, 16);
    else if (lanes == 64 && per_lane == 8) CZ_DBG(64, 8);
    else if (lanes == 64 && per_lane == 4) CZ_DBG(64, 4);
    else if (lanes == 16 && per_lane == 16) CZ_DBG(16, 16);
    else if (lanes == 16 && per_lane == 4) CZ_DBG(16, 4);
    else return cz::set_error(CZ_E_INVALID, "lanes must be 16 or 64, per_lane 4 / 8 / 16");
#undef CZ_DBG
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "debug_seq_sum launch: %s", hipGetErrorString(e));
    CZ_HIP(hipMemcpy(out, d_o.p, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
}

// ---- test hook: the plan build's own stable radix sort and scan (csrc/sort_scan.h) on arbitrary pairs ------------------------
extern "C" int cz_debug_sort_pairs(const uint32_t *keys, const uint32_t *vals, uint64_t n, uint32_t bits, uint32_t *out_keys, uint32_t *out_vals,
                                   uint32_t *out_scan /* [n] exclusive scan of vals, or NULL */) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (n == 0) return CZ_OK;
    if (!keys || !vals || !out_keys || !out_vals) return cz::set_error(CZ_E_INVALID, "null buffer");
    if (bits == 0 || bits > 32) return cz::set_error(CZ_E_INVALID, "bits must be in 1..32");
    cz::DevBuf<uint32_t> ka, va, kb, vb, scratch;
    CZ_HIP(ka.alloc(n));
    CZ_HIP(va.alloc(n));
    CZ_HIP(kb.alloc(n));
    CZ_HIP(vb.alloc(n));
    CZ_HIP(scratch.alloc(std::max(czsort::sort_scratch_words(n), czsort::scan_scratch_words(n))));
    CZ_HIP(hipMemcpy(ka.p, keys, n * 4, hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(va.p, vals, n * 4, hipMemcpyHostToDevice));
    if (out_scan) {
        if (n >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "scan of at most 2^32 - 2 entries");
        if ((rc = czsort::exclusive_scan_u32(va.p, vb.p, (uint32_t)n, scratch.p, nullptr))) return rc;
        CZ_HIP(hipMemcpy(out_scan, vb.p, n * 4, hipMemcpyDeviceToHost));
    }
    bool in_a = true;
    if ((rc = czsort::radix_sort_pairs_u32(ka.p, va.p, kb.p, vb.p, n, bits, scratch.p, nullptr, &in_a))) return rc;
    CZ_HIP(hipMemcpy(out_keys, in_a ? ka.p : kb.p, n * 4, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(out_vals, in_a ? va.p : vb.p, n * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
}

// where an index' vector table sits in the process's address space (placement experiments: scratch/r5_landing2.py)
extern "C" uint64_t cz_debug_index_table_address(const cz_hnsw_index *h) {
    if (!h) return 0;
    auto *ix = reinterpret_cast<const cz::HnswIndex *>(h);
    return (uint64_t)(uintptr_t)(ix->vec ? (const void *)ix->vec : (const void *)ix->vec64);
}

// Placement experiments (scratch/r5_landing3.py): give ONE of an index' arrays a new place in device memory, contents kept.  The
// new array is allocated while the old one is still held, so it cannot land where the old one is.
//   what = 0: the vector table   1: the level-0 link table   2: the upper-level tables   3: drop the pooled visited workspaces
//   contiguous != 0: ask for a physically contiguous range (plain hipMalloc otherwise)
extern "C" int cz_debug_index_rehome(cz_hnsw_index *h, int what, int contiguous) {
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    auto *ix = reinterpret_cast<cz::HnswIndex *>(h);
    CZ_HIP(hipDeviceSynchronize());
    auto move = [&](void **slot, size_t bytes) -> int {
        if (!*slot || !bytes) return CZ_OK;
        void *q = nullptr;
        if (!(contiguous && hipExtMallocWithFlags(&q, bytes, hipDeviceMallocContiguous) == hipSuccess)) {
            (void)hipGetLastError();
            CZ_HIP(hipMalloc(&q, bytes));
        }
        CZ_HIP(hipMemcpy(q, *slot, bytes, hipMemcpyDeviceToDevice));
        (void)hipFree(*slot);
        *slot = q;
        return CZ_OK;
    };
    if (what == 0) return ix->vec ? move((void **)&ix->vec, (size_t)ix->n * ix->ld * 4) : move((void **)&ix->vec64, (size_t)ix->n * ix->ld * 8);
    if (what == 1) return move((void **)&ix->nbr0, (size_t)ix->n * ix->w0 * 4);
    if (what == 2) {
        int rc = move((void **)&ix->up_base, (size_t)ix->n * 4);
        return rc ? rc : move((void **)&ix->up_nbrs, (size_t)std::max<uint64_t>(ix->up_rows, 1) * ix->wu * 4);
    }
    if (what == 3) {
        std::lock_guard<std::mutex> lk(ix->mu);
        for (auto &w : ix->pool) cz::HnswIndex::destroy(w);
        ix->pool.clear();
        return CZ_OK;
    }
    return cz::set_error(CZ_E_INVALID, "what = %d", what);
}
