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
