// scratch microbenchmark (not product): what bounds a random fetch of C-byte rows from HBM on gfx950?
//   hipcc --offload-arch=gfx950 -O3 scratch/rowfetch_bench.hip -o scratch/rowfetch_bench
// Every wave fetches whole rows (C bytes read, S bytes row stride) with global_load_dwordx4, 1 KiB per wave
// instruction, U rows in flight; the row list is sequential or a random permutation sample.  Reports GB/s of
// bytes actually requested.  Variables: row bytes C, stride S, footprint, rows in flight, waves per CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT>
__global__ void __launch_bounds__(256) fetch_rows(const char *__restrict__ base, const uint32_t *__restrict__ rows,
                                                  uint64_t n_rows, uint32_t row_bytes, uint64_t stride, float *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * 256) >> 6;
    const uint32_t iters = (row_bytes + 1023) / 1024;
    float acc = 0.f;
    for (uint64_t r0 = wave * U; r0 < n_rows; r0 += n_waves * U) {
        const float4 *p[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t r = r0 + u < n_rows ? r0 + u : r0;
            p[u] = (const float4 *)(base + (uint64_t)rows[r] * stride);
        }
        for (uint32_t j = 0; j < iters; j++) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = j * 64 + lane;
                if (c * 16 < row_bytes) {
                    if (NT) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        f4 t = __builtin_nontemporal_load((const f4 *)(p[u] + c));
                        v[u] = make_float4(t.x, t.y, t.z, t.w);
                    } else v[u] = p[u][c];
                } else v[u] = make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// variant: 3 KiB rows, cosine-style accumulation against a register-resident query (MODE 1), or against a query ROW
// fetched per pair from a small L2-resident table like distance_pairs_kernel does (MODE 2); DPP butterfly per row.
template <int OFF> __device__ __forceinline__ float xadd(float v);
template <> __device__ __forceinline__ float xadd<32>(float v) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
template <> __device__ __forceinline__ float xadd<16>(float v) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
template <> __device__ __forceinline__ float xadd<8>(float v) { return v + __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x128, 0xF, 0xF, false)); }
template <> __device__ __forceinline__ float xadd<4>(float v) { float t = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x104, 0xF, 0x5, false)); t = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(t), __float_as_uint(v), 0x114, 0xF, 0xA, false)); return v + t; }
template <> __device__ __forceinline__ float xadd<2>(float v) { return v + __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x4E, 0xF, 0xF, false)); }
template <> __device__ __forceinline__ float xadd<1>(float v) { return v + __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ float wsum(float v) { return xadd<1>(xadd<2>(xadd<4>(xadd<8>(xadd<16>(xadd<32>(v)))))); }

template <int U, int MODE, bool DB>
__global__ void __launch_bounds__(256) fetch_dot(const char *__restrict__ base, const uint32_t *__restrict__ rows,
                                                 uint64_t n_rows, const float4 *__restrict__ qtab, float *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * 256) >> 6;
    float4 q[3];
    for (int j = 0; j < 3; j++) q[j] = qtab[j * 64 + lane];
    float acc = 0.f;
    float4 v[2][U][3], qq[U][3];
    auto issue = [&](int buf, uint64_t r0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t r = r0 + u < n_rows ? r0 + u : n_rows - 1;
            const uint32_t id = rows[r];
            const float4 *p = (const float4 *)(base + (uint64_t)id * 3072);
#pragma unroll
            for (int j = 0; j < 3; j++) v[buf][u][j] = p[j * 64 + lane];
            if (MODE == 2) {
                const float4 *pq = qtab + (uint64_t)(id & 1023) * 192;
#pragma unroll
                for (int j = 0; j < 3; j++) qq[u][j] = pq[j * 64 + lane];
            }
        }
    };
    auto retire = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float4 x = v[buf][u][j];
                const float4 y = MODE == 2 ? qq[u][j] : q[j];
                a0 = fmaf(y.x, x.x, a0); a1 = fmaf(x.x, x.x, a1);
                a0 = fmaf(y.y, x.y, a0); a1 = fmaf(x.y, x.y, a1);
                a0 = fmaf(y.z, x.z, a0); a1 = fmaf(x.z, x.z, a1);
                a0 = fmaf(y.w, x.w, a0); a1 = fmaf(x.w, x.w, a1);
            }
            acc += wsum(a0) + wsum(a1);
        }
    };
    if (DB) {
        uint64_t r0 = wave * U;
        if (r0 < n_rows) issue(0, r0);
        int buf = 0;
        for (; r0 < n_rows; r0 += n_waves * U) {
            const uint64_t nx = r0 + n_waves * U;
            if (nx < n_rows) issue(buf ^ 1, nx);
            retire(buf);
            buf ^= 1;
        }
    } else {
        for (uint64_t r0 = wave * U; r0 < n_rows; r0 += n_waves * U) {
            issue(0, r0);
            retire(0);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int U, int MODE, bool DB>
static float run_dot(const char *d_base, const uint32_t *d_rows, uint64_t n_rows, const float4 *qtab, int blocks, float *d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((fetch_dot<U, MODE, DB>), dim3(blocks), dim3(256), 0, 0, d_base, d_rows, n_rows, qtab, d_out);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((fetch_dot<U, MODE, DB>), dim3(blocks), dim3(256), 0, 0, d_base, d_rows, n_rows, qtab, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

static uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <int U, bool NT>
static float run(const char *d_base, const uint32_t *d_rows, uint64_t n_rows, uint32_t row_bytes, uint64_t stride,
                 int blocks, float *d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++)
        hipLaunchKernelGGL((fetch_rows<U, NT>), dim3(blocks), dim3(256), 0, 0, d_base, d_rows, n_rows, row_bytes, stride, d_out);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((fetch_rows<U, NT>), dim3(blocks), dim3(256), 0, 0, d_base, d_rows, n_rows, row_bytes, stride, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const uint64_t foot_gb_max = argc > 1 ? (uint64_t)atoll(argv[1]) : 12;
    char *d_base;
    const uint64_t bytes = foot_gb_max << 30;
    CK(hipMalloc((void **)&d_base, bytes));
    CK(hipMemset(d_base, 1, bytes));
    float *d_out;
    CK(hipMalloc((void **)&d_out, 4));
    const uint64_t n_fetch = 4u << 20;  // rows fetched per launch (x row bytes)
    std::vector<uint32_t> h(n_fetch);
    uint32_t *d_rows;
    CK(hipMalloc((void **)&d_rows, n_fetch * 4));
    printf("%-10s %8s %8s %8s %4s %3s %6s %10s %8s\n", "pattern", "rowB", "stride", "foot_MB", "U", "nt", "blocks", "ms", "GB/s");
    struct Cfg { uint32_t row_bytes; uint64_t stride; uint64_t foot_mb; int random; };
    std::vector<Cfg> cfgs;
    for (uint64_t foot : {3072ull, 768ull, 12288ull}) {
        if ((foot << 20) > bytes) continue;
        cfgs.push_back({3072, 3072, foot, 1});
    }
    cfgs.push_back({3072, 3072, 3072, 0});
    cfgs.push_back({3072, 4096, 4096, 1});   // rows padded to a 4 KiB stride
    for (uint32_t rb : {256u, 512u, 1024u, 2048u, 4096u, 8192u, 16384u, 65536u}) cfgs.push_back({rb, rb, 3072, 1});
    if (argc > 2) {  // compute variants on random 3 KiB rows over a 3 GB footprint
        const uint64_t n = n_fetch / 2;
        uint64_t sd = 42;
        for (uint64_t i = 0; i < n; i++) h[i] = (uint32_t)(sm64(sd) % (3072ull << 20) / 3072);
        CK(hipMemcpy(d_rows, h.data(), n * 4, hipMemcpyHostToDevice));
        float4 *qtab;
        CK(hipMalloc((void **)&qtab, 1024 * 3072));
        CK(hipMemset(qtab, 0, 1024 * 3072));
        printf("%-44s %10s %8s\n", "variant (random 3 KiB rows, 3 GB)", "ms", "GB/s");
#define RUN(U, MODE, DB, BL, NAME) { float ms = run_dot<U, MODE, DB>(d_base, d_rows, n, qtab, BL, d_out); printf("%-44s %10.3f %8.0f\n", NAME, ms, (double)n * 3072 / ms / 1e6); fflush(stdout); }
        RUN(4, 1, false, 2048, "dot, reg query, U=4, 2048 blocks");
        RUN(2, 1, false, 2048, "dot, reg query, U=2, 2048 blocks");
        RUN(2, 1, true, 2048, "dot, reg query, U=2 double-buffered");
        RUN(4, 1, true, 2048, "dot, reg query, U=4 double-buffered");
        RUN(2, 1, true, 1024, "dot, reg query, U=2 dbuf, 1024 blocks");
        RUN(4, 2, false, 2048, "dot, query ROW per pair (L2), U=4");
        RUN(2, 2, true, 2048, "dot, query ROW per pair (L2), U=2 dbuf");
        return 0;
    }
    for (const Cfg &c : cfgs) {
        const uint64_t n_slots = (c.foot_mb << 20) / c.stride;
        const uint64_t n = std::min<uint64_t>(n_fetch, (uint64_t)(6ull << 30) / c.row_bytes);  // <= 6 GB per launch
        uint64_t s = 42;
        for (uint64_t i = 0; i < n; i++) h[i] = c.random ? (uint32_t)(sm64(s) % n_slots) : (uint32_t)(i % n_slots);
        CK(hipMemcpy(d_rows, h.data(), n * 4, hipMemcpyHostToDevice));
        for (int blocks : {2048, 4096}) {
            for (int variant = 0; variant < 4; variant++) {
                if (c.row_bytes != 3072 && (variant == 0 || variant == 2 || blocks == 4096)) continue;
                float ms;
                int U;
                bool nt = false;
                if (variant == 0) { U = 2; ms = run<2, false>(d_base, d_rows, n, c.row_bytes, c.stride, blocks, d_out); }
                else if (variant == 1) { U = 4; ms = run<4, false>(d_base, d_rows, n, c.row_bytes, c.stride, blocks, d_out); }
                else if (variant == 2) { U = 8; ms = run<8, false>(d_base, d_rows, n, c.row_bytes, c.stride, blocks, d_out); }
                else { U = 4; nt = true; ms = run<4, true>(d_base, d_rows, n, c.row_bytes, c.stride, blocks, d_out); }
                printf("%-10s %8u %8llu %8llu %4d %3d %6d %10.3f %8.0f\n", c.random ? "random" : "sequential", c.row_bytes,
                       (unsigned long long)c.stride, (unsigned long long)c.foot_mb, U, (int)nt, blocks, ms,
                       (double)n * c.row_bytes / ms / 1e6);
                fflush(stdout);
            }
        }
    }
    return 0;
}
