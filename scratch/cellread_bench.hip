// scratch microbenchmark (not product): the access pattern of PageRank's accumulate phase B -- every wave walks S slices and
// reads its own C contiguous bytes of each (wave w's cell of slice s sits at s * slice_bytes + w * C), D requests in flight
// per wave, 64-lane dword loads (256 B per instruction) or dwordx4 loads (1 KiB per instruction).
//   hipcc --offload-arch=gfx950 -O3 scratch/cellread_bench.hip -o scratch/cellread_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int D, int VEC, int ACTIVE>  // VEC = dwords per lane per request; lanes >= ACTIVE repeat lane 0 (a short piece)
__global__ void cell_read(const float *__restrict__ base, uint32_t S, uint64_t slice_words, uint32_t cell_words, uint32_t n_waves, float *out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= n_waves) return;
    const uint32_t per_req = ACTIVE * VEC;
    const uint32_t reqs_per_cell = (cell_words + per_req - 1) / per_req;  // (the last request of a cell runs into the next cell)
    const uint32_t total = S * reqs_per_cell;
    float acc = 0.f;
    float v[D][VEC];
    auto addr = [&](uint32_t q) { const uint32_t s = q / reqs_per_cell, k = q % reqs_per_cell; return base + (uint64_t)s * slice_words + (uint64_t)wave * cell_words + k * per_req + (lane < ACTIVE ? lane : 0) * VEC; };
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float *p = addr(d < (int)total ? d : 0);
#pragma unroll
        for (int j = 0; j < VEC; j++) v[d][j] = __builtin_nontemporal_load(p + j);
    }
    for (uint32_t q0 = 0; q0 < total; q0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
#pragma unroll
            for (int j = 0; j < VEC; j++) acc += v[d][j];
            const uint32_t qn = q0 + D + d;
            const float *p = addr(qn < total ? qn : 0);
#pragma unroll
            for (int j = 0; j < VEC; j++) v[d][j] = __builtin_nontemporal_load(p + j);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint64_t total_bytes = 1600ull << 20;  // (4 x the sweep's value stream: beyond the 256 MB Infinity Cache)
    float *buf, *out;
    CK(hipMalloc(&buf, total_bytes + (1 << 20)));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0, total_bytes + (1 << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%8s %6s %9s %6s %4s %4s %9s %8s\n", "cell_B", "S", "waves", "w/CU", "D", "vec", "ms", "GB/s");
    const uint32_t S = 1024;
    for (uint32_t waves_per_cu : {4u, 8u, 16u, 32u}) {
        const uint32_t n_waves = waves_per_cu * 256;
        const uint64_t slice_words = total_bytes / 4 / S;
        const uint32_t cell_words = (uint32_t)(slice_words / n_waves) & ~63u;
        for (int cfg = 0; cfg < 5; cfg++) {
            // (dword, 64 lanes) (dword, 32 lanes = 128 B per request) (dword, 48 lanes) (dwordx2) (dwordx4); 8 requests in flight
            const int VEC = cfg == 3 ? 2 : cfg == 4 ? 4 : 1, ACT = cfg == 1 ? 32 : cfg == 2 ? 48 : 64;
            if (cell_words < (uint32_t)(ACT * VEC)) continue;
            const uint32_t threads = waves_per_cu >= 16 ? 1024 : waves_per_cu * 64;
            const uint32_t blocks = n_waves * 64 / threads;
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
#define LAUNCH(DD, VV, AA) hipLaunchKernelGGL((cell_read<DD, VV, AA>), dim3(blocks), dim3(threads), 0, 0, buf, S, slice_words, cell_words, n_waves, out)
                if (cfg == 0) LAUNCH(8, 1, 64); else if (cfg == 1) LAUNCH(8, 1, 32); else if (cfg == 2) LAUNCH(8, 1, 48); else if (cfg == 3) LAUNCH(8, 2, 64); else LAUNCH(8, 4, 64);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            const uint32_t per_req = ACT * VEC, reqs = (cell_words + per_req - 1) / per_req;
            const double bytes = (double)S * n_waves * reqs * per_req * 4, nreq = (double)S * n_waves * reqs;
            printf("%8u %6u %9u %6u %4d %4d %9.4f %8.0f   %3d lanes, %5.1f G requests/s\n", cell_words * 4, S, n_waves, waves_per_cu, 8, VEC, best, bytes / best / 1e6, ACT, nreq / best / 1e6);
        }
    }
    return 0;
}
