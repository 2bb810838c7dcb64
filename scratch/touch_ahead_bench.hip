// scratch microbenchmark (not product), round 6: does TOUCHING the rows of the next step ahead of time shorten the step?
//   hipcc --offload-arch=gfx950 -O3 scratch/touch_ahead_bench.hip -o scratch/touch_ahead_bench
// One workgroup of 256 threads on an otherwise idle chip -- the B = 1 search (hnsw_knn_spec_kernel) -- walks STEPS steps.  A step
// fetches R random 3 KiB rows of a TABLE_GB table (4 waves x U = 8 rows in flight: one round for R <= 32) and then "works" for
// WORK_NS nanoseconds without touching memory (the step's finish / merge / select).  Variants:
//   0  as described (what the search does today)
//   1  before the work phase, wave 1 issues ONE 4-byte load per row of the NEXT step (lane i -> row i): page walk and the row's
//      first line ride under the work phase
//   2  as 1 but one 64-byte-apart load per 128-byte line of the whole row (24 lanes x ... = the whole row into L2)
// Reports nanoseconds per step (shader clock at 100 MHz constant clock -> wall via hipEvents).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VARIANT>
__global__ void __launch_bounds__(256) walk(const char *__restrict__ base, const uint32_t *__restrict__ rows, uint32_t steps, uint32_t R,
                                            uint32_t work_cycles, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    uint32_t sink = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const uint32_t *rs = rows + (size_t)s * R;
        // the step's rows: wave w takes rows w, w + 4, ... (8 in flight)
        float4 v[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = wave + 4 * u;
            const uint32_t id = rs[j < R ? j : 0];
            const float4 *p = (const float4 *)(base + (uint64_t)id * 3072);
#pragma unroll
            for (int c = 0; c < 3; c++) v[u][c] = p[c * 64 + lane];
        }
        uint32_t t = 0;
        if (VARIANT == 1) {  // branch-free: a branch around a load makes the wave wait for everything in flight at its end
            const uint32_t *rn = rows + (size_t)min(s + 1, steps - 1) * R;
            const bool mine = wave == 1 && lane < (int)R;
            const uint32_t id = mine ? rn[lane] : rs[0];
            t = *(const uint32_t *)(base + (uint64_t)id * 3072);
        } else if (VARIANT == 2) {
            const uint32_t *rn = rows + (size_t)min(s + 1, steps - 1) * R;
#pragma unroll
            for (int i = 0; i < 3; i++) {  // 4 waves x 64 lanes x 3 = 768 = 32 rows x 24 lines
                const uint32_t k = threadIdx.x + i * 256;
                const bool mine = k < R * 24;
                const uint32_t id = mine ? rn[k / 24] : rs[0];
                t ^= *(const uint32_t *)(base + (uint64_t)id * 3072 + (mine ? (k % 24) * 128 : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < 3; c++) acc += v[u][c].x + v[u][c].y + v[u][c].z + v[u][c].w;
        // (a raw barrier: __syncthreads() carries a fence that waits for every outstanding load -- the touches included)
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < work_cycles) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_s_barrier();
        sink ^= t;
    }
    if (acc == 12345.678f || sink == 0x12345u) out[0] = acc + sink;
}

int main(int argc, char **argv) {
    const double table_gb = argc > 1 ? atof(argv[1]) : 30.0;
    const uint32_t R = argc > 2 ? atoi(argv[2]) : 32, steps = 2000;
    const uint64_t n_rows = (uint64_t)(table_gb * 1e9 / 3072);
    char *base;
    CK(hipMalloc(&base, n_rows * 3072));
    CK(hipMemset(base, 1, n_rows * 3072));
    std::mt19937_64 rng(7);
    std::vector<uint32_t> h((size_t)steps * 32);
    uint32_t *d_rows;
    float *d_out;
    CK(hipMalloc(&d_rows, h.size() * 4));
    CK(hipMalloc(&d_out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("table %.1f GB (%llu rows), R = %u rows per step, %u steps; clock64 ~ %d kHz\n", table_gb, (unsigned long long)n_rows, R, steps, clk_khz);
    for (uint32_t work_ns : {0u, 1000u, 2000u, 3000u}) {
        for (int variant = 0; variant < 3; variant++) {
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                for (auto &x : h) x = (uint32_t)(rng() % n_rows);
                CK(hipMemcpy(d_rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
                const uint32_t wc = (uint32_t)((double)work_ns * 0.1);  // s_memtime / clock64 tick at 100 MHz
                CK(hipEventRecord(e0));
                if (variant == 0) hipLaunchKernelGGL(walk<0>, dim3(1), dim3(256), 0, 0, base, d_rows, steps, R, wc, d_out);
                else if (variant == 1) hipLaunchKernelGGL(walk<1>, dim3(1), dim3(256), 0, 0, base, d_rows, steps, R, wc, d_out);
                else hipLaunchKernelGGL(walk<2>, dim3(1), dim3(256), 0, 0, base, d_rows, steps, R, wc, d_out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, (double)ms);
            }
            printf("work %4u ns  variant %d: %8.1f ns per step (fetch part %8.1f ns)\n", work_ns, variant, best * 1e6 / steps, best * 1e6 / steps - work_ns);
        }
    }
    return 0;
}
