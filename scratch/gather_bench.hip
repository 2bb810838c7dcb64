// scratch microbenchmark (not product): what bounds a 4-byte gather on gfx950 as a function of the footprint?
//   hipcc --offload-arch=gfx950 -O3 scratch/gather_bench.hip -o scratch/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// idx[] streamed coalesced (4 B/edge), value gathered from table[base + (idx & mask)], summed
__global__ void __launch_bounds__(256) gather_global(const uint32_t *__restrict__ idx, const float *__restrict__ table,
                                                     uint32_t mask, uint64_t n, float *__restrict__ out) {
    float s = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
        const uint4 k = *(const uint4 *)(idx + i);
        s += table[k.x & mask] + table[k.y & mask] + table[k.z & mask] + table[k.w & mask];
    }
    if (s == 12345.678f) out[0] = s;
}

// the same, but every block sweeps the table in lock-step slices of `slice` entries: block handles entries
// i with slice id = (i / per_slice) ... emulates a source-blocked sweep where concurrently running blocks share a slice
__global__ void __launch_bounds__(256) gather_sliced(const uint32_t *__restrict__ idx, const float *__restrict__ table,
                                                     uint32_t slice_mask, uint32_t n_slices, uint32_t per_tile,
                                                     float *__restrict__ out) {
    // block b owns tiles (b, s) for s = 0..n_slices-1; each tile has per_tile entries, stored at ((b*n_slices)+s)*per_tile
    float s = 0.f;
    for (uint32_t sl = 0; sl < n_slices; sl++) {
        const uint32_t *p = idx + ((uint64_t)blockIdx.x * n_slices + sl) * per_tile;
        const float *t = table + (uint64_t)sl * (slice_mask + 1);
        for (uint32_t i = threadIdx.x * 4; i + 3 < per_tile; i += 1024) {
            const uint4 k = *(const uint4 *)(p + i);
            s += t[k.x & slice_mask] + t[k.y & slice_mask] + t[k.z & slice_mask] + t[k.w & slice_mask];
        }
    }
    if (s == 12345.678f) out[0] = s;
}

// PB phase A proxy: slice staged in LDS (W entries), edges streamed (u32 local src), LDS gather, coalesced store
template <int W>
__global__ void __launch_bounds__(1024) lds_gather_store(const uint32_t *__restrict__ idx, const float *__restrict__ table,
                                                         uint32_t per_block, float *__restrict__ vals) {
    extern __shared__ float sl[];
    const float *t = table + (uint64_t)blockIdx.x * W;
    for (int i = threadIdx.x * 4; i < W; i += 4096) *(float4 *)(sl + i) = *(const float4 *)(t + i);
    __syncthreads();
    const uint32_t *p = idx + (uint64_t)blockIdx.x * per_block;
    float *o = vals + (uint64_t)blockIdx.x * per_block;
    for (uint32_t i = threadIdx.x * 4; i + 3 < per_block; i += 4096) {
        const uint4 k = *(const uint4 *)(p + i);
        float4 v;
        v.x = sl[k.x & (W - 1)]; v.y = sl[k.y & (W - 1)]; v.z = sl[k.z & (W - 1)]; v.w = sl[k.w & (W - 1)];
        *(float4 *)(o + i) = v;
    }
}

// PB phase B proxy: stream vals + u16 rows, accumulate into LDS acc (R rows) with ds_add (unordered; rate only)
template <int R>
__global__ void __launch_bounds__(256) stream_accumulate(const float *__restrict__ vals, const uint32_t *__restrict__ rows,
                                                         uint32_t per_block, float *__restrict__ out) {
    __shared__ float acc[R];
    for (int i = threadIdx.x; i < R; i += 256) acc[i] = 0.f;
    __syncthreads();
    const float *v = vals + (uint64_t)blockIdx.x * per_block;
    const uint32_t *r = rows + (uint64_t)blockIdx.x * per_block;
    for (uint32_t i = threadIdx.x * 4; i + 3 < per_block; i += 1024) {
        const float4 x = *(const float4 *)(v + i);
        const uint4 k = *(const uint4 *)(r + i);
        atomicAdd(&acc[k.x & (R - 1)], x.x);
        atomicAdd(&acc[k.y & (R - 1)], x.y);
        atomicAdd(&acc[k.z & (R - 1)], x.z);
        atomicAdd(&acc[k.w & (R - 1)], x.w);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += 256) out[(uint64_t)blockIdx.x * R + i] = acc[i];
}


template <int MODE>
__device__ __forceinline__ float ldg(const float *p) {
    float v;
    if (MODE == 0) return *p;
    else if (MODE == 1) asm volatile("global_load_dword %0, %1, off sc1\n" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 2) asm volatile("global_load_dword %0, %1, off nt\n" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0\n" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__global__ void __launch_bounds__(256) gather_mode(const uint32_t *__restrict__ idx, const float *__restrict__ table,
                                                   uint32_t mask, uint64_t n, float *__restrict__ out) {
    float s = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
        const uint4 k = *(const uint4 *)(idx + i);
        float a = ldg<MODE>(table + (k.x & mask)), b = ldg<MODE>(table + (k.y & mask)), c = ldg<MODE>(table + (k.z & mask)),
              d = ldg<MODE>(table + (k.w & mask));
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s += a + b + c + d;
    }
    if (s == 12345.678f) out[0] = s;
}
// u16 gathers: two 16-bit halves? no -- 8-byte gather (pairs) to see whether the limit is per request or per byte
__global__ void __launch_bounds__(256) gather8(const uint32_t *__restrict__ idx, const float2 *__restrict__ table,
                                               uint32_t mask, uint64_t n, float *__restrict__ out) {
    float s = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
        const uint4 k = *(const uint4 *)(idx + i);
        float2 a = table[k.x & mask], b = table[k.y & mask], c = table[k.z & mask], d = table[k.w & mask];
        s += a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y;
    }
    if (s == 12345.678f) out[0] = s;
}
// PB phase B proxy 2: stream vals + u16 perm, scatter into LDS, barrier, one lane per row sums ~deg entries in order
template <int L>
__global__ void __launch_bounds__(512) scatter_rowsum(const float *__restrict__ vals, const uint16_t *__restrict__ perm,
                                                      uint32_t deg, float *__restrict__ out) {
    __shared__ float buf[L];
    const float *v = vals + (uint64_t)blockIdx.x * L;
    const uint16_t *p = perm + (uint64_t)blockIdx.x * L;
    for (uint32_t i = threadIdx.x * 4; i < L; i += 2048) {
        const float4 x = *(const float4 *)(v + i);
        const uint2 k = *(const uint2 *)(p + i);
        buf[(k.x & 0xffff) % L] = x.x; buf[(k.x >> 16) % L] = x.y; buf[(k.y & 0xffff) % L] = x.z; buf[(k.y >> 16) % L] = x.w;
    }
    __syncthreads();
    const uint32_t rows = L / deg;
    for (uint32_t r = threadIdx.x; r < rows; r += 512) {
        float s = 0.f;
        for (uint32_t e = 0; e < deg; e++) s = s + buf[r * deg + e];
        out[(uint64_t)blockIdx.x * rows + r] = s;
    }
}

__global__ void fill_idx(uint32_t *idx, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        idx[i] = mix((uint32_t)i * 2654435761u + seed);
}
__global__ void fill_f(float *p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}
__global__ void copy4(const float4 *a, float4 *b, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <typename F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const uint64_t E = 100ull << 20;  // 104.9M "edges"
    const uint64_t NT = 16ull << 20;  // table of 16M floats (64 MB)
    uint32_t *idx; float *table, *out, *vals;
    CK(hipMalloc(&idx, E * 4)); CK(hipMalloc(&table, NT * 4)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&vals, E * 4));
    fill_idx<<<4096, 256>>>(idx, E, 17);
    fill_f<<<4096, 256>>>(table, NT);
    CK(hipDeviceSynchronize());
    {
        float ms = time_ms([&] { copy4<<<8192, 256>>>((const float4 *)idx, (float4 *)vals, E / 4); });
        printf("copy 400MB->400MB: %.3f ms  %.2f TB/s\n", ms, 2 * E * 4 / ms / 1e9);
    }
    printf("-- gather_global: footprint sweep (all blocks share one footprint), 4 B idx stream + 4 B gather\n");
    for (uint32_t fp_log = 13; fp_log <= 24; fp_log++) {  // entries: 8K (32KB) .. 16M (64MB)
        const uint32_t mask = (1u << fp_log) - 1;
        for (int blocks : {2048, 8192}) {
            float ms = time_ms([&] { gather_global<<<blocks, 256>>>(idx, table, mask, E, out); });
            printf("footprint %8.2f MB blocks %5d: %.3f ms  %.1f Ggather/s\n", (mask + 1) * 4.0 / 1048576, blocks, ms, E / ms / 1e6);
        }
    }
    printf("-- gather_sliced: block b sweeps slices 0..S-1 (tile = per_tile entries), table 40MB\n");
    for (uint32_t sl_log : {16u, 17u, 18u, 19u, 20u}) {  // slice entries 64K(256KB) .. 1M (4MB)
        const uint32_t n_slices = (10u << 20) >> sl_log;
        for (int blocks : {512, 1024, 2048}) {
            uint32_t per_tile = (uint32_t)(E / blocks / n_slices) & ~1023u;
            if (per_tile < 1024) continue;
            float ms = time_ms([&] { gather_sliced<<<blocks, 256>>>(idx, table, (1u << sl_log) - 1, n_slices, per_tile, out); });
            const double n = (double)per_tile * n_slices * blocks;
            printf("slice %6.2f MB x %3u slices, blocks %4d, per_tile %6u: %.3f ms  %.1f Ggather/s\n", (4 << sl_log) / 1048576.0,
                   n_slices, blocks, per_tile, ms, n / ms / 1e6);
        }
    }
    printf("-- lds_gather_store (PB phase A proxy)\n");
    {
        constexpr int W = 32768;
        const int blocks = 320;
        const uint32_t per_block = (uint32_t)(E / blocks) & ~4095u;
        hipFuncSetAttribute((const void *)lds_gather_store<W>, hipFuncAttributeMaxDynamicSharedMemorySize, W * 4);
        float ms = time_ms([&] { lds_gather_store<W><<<blocks, 1024, W * 4>>>(idx, table, per_block, vals); });
        printf("W=32K blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)per_block * blocks / ms / 1e6);
    }
    {
        constexpr int W = 16384;
        const int blocks = 640;
        const uint32_t per_block = (uint32_t)(E / blocks) & ~4095u;
        hipFuncSetAttribute((const void *)lds_gather_store<W>, hipFuncAttributeMaxDynamicSharedMemorySize, W * 4);
        float ms = time_ms([&] { lds_gather_store<W><<<blocks, 1024, W * 4>>>(idx, table, per_block, vals); });
        printf("W=16K blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)per_block * blocks / ms / 1e6);
    }
    {
        constexpr int W = 8192;
        const int blocks = 1280;
        const uint32_t per_block = (uint32_t)(E / blocks) & ~4095u;
        float ms = time_ms([&] { lds_gather_store<W><<<blocks, 1024, W * 4>>>(idx, table, per_block, vals); });
        printf("W=8K blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)per_block * blocks / ms / 1e6);
    }
    printf("-- stream_accumulate (PB phase B proxy, LDS atomic add)\n");
    {
        constexpr int R = 8192;
        const int blocks = 1280;
        const uint32_t per_block = (uint32_t)(E / blocks) & ~1023u;
        float ms = time_ms([&] { stream_accumulate<R><<<blocks, 256>>>(vals, idx, per_block, out); });
        printf("R=8K blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)per_block * blocks / ms / 1e6);
    }
    {
        constexpr int R = 4096;
        const int blocks = 2560;
        const uint32_t per_block = (uint32_t)(E / blocks) & ~1023u;
        float ms = time_ms([&] { stream_accumulate<R><<<blocks, 256>>>(vals, idx, per_block, out); });
        printf("R=4K blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)per_block * blocks / ms / 1e6);
    }

    printf("-- load policy variants, footprint 1MB and 2MB and 40MB(mask 8M entries=32MB)\n");
    for (uint32_t fp_log : {18u, 19u, 23u}) {
        const uint32_t mask = (1u << fp_log) - 1;
        float ms;
        ms = time_ms([&] { gather_mode<0><<<8192, 256>>>(idx, table, mask, E, out); }); printf("fp %5.1f MB plain  : %.3f ms %.1f G/s\n", (mask+1)*4.0/1048576, ms, E / ms / 1e6);
        ms = time_ms([&] { gather_mode<1><<<8192, 256>>>(idx, table, mask, E, out); }); printf("fp %5.1f MB sc1    : %.3f ms %.1f G/s\n", (mask+1)*4.0/1048576, ms, E / ms / 1e6);
        ms = time_ms([&] { gather_mode<2><<<8192, 256>>>(idx, table, mask, E, out); }); printf("fp %5.1f MB nt     : %.3f ms %.1f G/s\n", (mask+1)*4.0/1048576, ms, E / ms / 1e6);
        ms = time_ms([&] { gather_mode<3><<<8192, 256>>>(idx, table, mask, E, out); }); printf("fp %5.1f MB sc0sc1 : %.3f ms %.1f G/s\n", (mask+1)*4.0/1048576, ms, E / ms / 1e6);
        ms = time_ms([&] { gather_mode<4><<<8192, 256>>>(idx, table, mask, E, out); }); printf("fp %5.1f MB sc0    : %.3f ms %.1f G/s\n", (mask+1)*4.0/1048576, ms, E / ms / 1e6);
        ms = time_ms([&] { gather8<<<8192, 256>>>(idx, (const float2 *)table, mask >> 1, E, out); }); printf("fp %5.1f MB 8B plain: %.3f ms %.1f G/s\n", (mask+1)*4.0/1048576, ms, E / ms / 1e6);
    }
    printf("-- scatter_rowsum (PB phase B proxy 2: LDS scatter + per-row in-order sum)\n");
    {
        constexpr int L = 20480;
        const int blocks = (int)(E / L);
        float ms = time_ms([&] { scatter_rowsum<L><<<blocks, 512>>>(vals, (const uint16_t *)idx, 10, out); });
        printf("L=20480 blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)L * blocks / ms / 1e6);
    }
    {
        constexpr int L = 10240;
        const int blocks = (int)(E / L);
        float ms = time_ms([&] { scatter_rowsum<L><<<blocks, 512>>>(vals, (const uint16_t *)idx, 10, out); });
        printf("L=10240 blocks %d: %.3f ms  %.1f Gedge/s\n", blocks, ms, (double)L * blocks / ms / 1e6);
    }
    return 0;
}
