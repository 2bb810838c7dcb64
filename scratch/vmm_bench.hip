// scratch microbenchmark (not product): does the PLACEMENT of the 30 GB vector table decide what the random-row kernels reach?
//   hipcc --offload-arch=gfx950 -O3 scratch/vmm_bench.hip -Iinclude -Lcozo_amd/lib -lcozo_gpu -Wl,-rpath,'$ORIGIN/../cozo_amd/lib' -o scratch/vmm_bench
// The product kernel itself is timed (`cz_distance_batch` with device pointers = distance_pairs_kernel over 4M random pairs), over a
// base table allocated in different ways:
//   malloc            plain hipMalloc (what hnsw_api.hip did through round 3)
//   vmm:C:A           hipMemAddressReserve with alignment A MiB, physical chunks of C MiB from hipMemCreate, hipMemMap + hipMemSetAccess
//   contig            hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   vmm1:A            ONE physical allocation for the whole table, VA aligned to A MiB
//   vmma:C            chunks of C MiB mapped at a VA aligned to C MiB BY US (the reservation is C MiB larger and the mapping starts at the
//                     next multiple): if what decides a landing is whether virtual and physical addresses agree modulo a large power of
//                     two (so that the page tables can use large fragments), this form agrees by construction
// optionally after fragmenting VRAM (`frag:G:K` = allocate G GiB in K-MiB hipMallocs, free every other one, then take the
// largest contiguous remainder away with one more hipMalloc).  Usage:
//   vmm_bench <rows> <reps> mode [mode ...]       (mode "frag:G:K" applies to the modes after it; "window:R" restricts pairs to R rows)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cozo_gpu.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)

__global__ void fill_kernel(float *p, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int32_t)(h & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
    }
}

struct Table {
    float *p = nullptr;
    void *va_base = nullptr;      // what hipMemAddressReserve returned (vmma: the mapped range starts above it)
    size_t bytes = 0, va_bytes = 0, va_reserved = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool vmm = false;
    void free_() {
        if (!p) return;
        if (!vmm) { CK(hipFree(p)); }
        else {
            CK(hipMemUnmap(p, va_bytes));
            for (auto h : handles) CK(hipMemRelease(h));
            CK(hipMemAddressFree(va_base ? va_base : (void *)p, va_reserved ? va_reserved : va_bytes));
            handles.clear();
            va_base = nullptr;
            va_reserved = 0;
        }
        p = nullptr;
    }
};

static size_t g_gran_min = 0, g_gran_rec = 0;

static void alloc_table(Table &t, size_t bytes, const std::string &mode) {
    t.bytes = bytes;
    if (mode == "malloc") {
        t.vmm = false;
        CK(hipMalloc((void **)&t.p, bytes));
        return;
    }
    if (mode == "contig") {  // hipExtMallocWithFlags(hipDeviceMallocContiguous): physically contiguous VRAM, if the driver finds it
        t.vmm = false;
        hipError_t e = hipExtMallocWithFlags((void **)&t.p, bytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            printf("contig: %s -- falling back to hipMalloc\n", hipGetErrorString(e));
            (void)hipGetLastError();
            CK(hipMalloc((void **)&t.p, bytes));
        }
        return;
    }
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t chunk = 0, align = 0;
    bool self_align = false;
    if (mode.rfind("vmma:", 0) == 0) { chunk = align = (size_t)atoll(mode.c_str() + 5) << 20; self_align = true; }
    else if (mode.rfind("vmm1:", 0) == 0) { align = (size_t)atoll(mode.c_str() + 5) << 20; chunk = 0; }
    else if (mode.rfind("vmm:", 0) == 0) {
        const char *s = mode.c_str() + 4;
        chunk = (size_t)atoll(s) << 20;
        const char *c = strchr(s, ':');
        align = c ? (size_t)atoll(c + 1) << 20 : chunk;
    } else { printf("unknown mode %s\n", mode.c_str()); exit(2); }
    if (chunk == 0) chunk = (bytes + g_gran_rec - 1) / g_gran_rec * g_gran_rec;
    chunk = (chunk + g_gran_min - 1) / g_gran_min * g_gran_min;
    const size_t n_chunks = (bytes + chunk - 1) / chunk;
    t.va_bytes = n_chunks * chunk;
    t.vmm = true;
    void *va = nullptr;
    if (self_align) {  // the alignment argument is ignored by this runtime: reserve `align` more and start at the next multiple
        t.va_reserved = t.va_bytes + align;
        CK(hipMemAddressReserve(&va, t.va_reserved, 0, nullptr, 0));
        t.va_base = va;
        va = (void *)(((uintptr_t)va + align - 1) / align * align);
    } else {
        CK(hipMemAddressReserve(&va, t.va_bytes, align, nullptr, 0));
    }
    t.p = (float *)va;
    for (size_t i = 0; i < n_chunks; i++) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)va + i * chunk, chunk, 0, h, 0));
        t.handles.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, t.va_bytes, &acc, 1));
}

static uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    if (argc < 4) { printf("usage: vmm_bench <rows> <reps> mode...\n"); return 2; }
    const uint64_t n = (uint64_t)atoll(argv[1]);
    const int reps = atoi(argv[2]);
    const uint32_t dim = 768, nq = 1024;
    const uint64_t P = 4u << 20;
    if (cz_init(0) != 0) { printf("cz_init: %s\n", cz_last_error()); return 1; }
    int dev = 0;
    CK(hipGetDevice(&dev));
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        CK(hipMemGetAllocationGranularity(&g_gran_min, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&g_gran_rec, &prop, hipMemAllocationGranularityRecommended));
        size_t fr = 0, tot = 0;
        CK(hipMemGetInfo(&fr, &tot));
        printf("granularity: minimum %zu, recommended %zu bytes; free %.1f of %.1f GiB\n", g_gran_min, g_gran_rec, fr / 1073741824.0, tot / 1073741824.0);
    }
    float *d_q;
    uint32_t *d_pairs;
    double *d_out;
    CK(hipMalloc((void **)&d_q, (size_t)nq * dim * 4));
    CK(hipMalloc((void **)&d_pairs, P * 8));
    CK(hipMalloc((void **)&d_out, P * 8));
    fill_kernel<<<1024, 256>>>(d_q, (uint64_t)nq * dim, 7u);
    std::vector<uint32_t> h_pairs(P * 2);
    std::vector<void *> frag_keep;
    uint64_t window = n;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int a = 3; a < argc; a++) {
        std::string mode = argv[a];
        if (mode.rfind("window:", 0) == 0) { window = (uint64_t)atoll(mode.c_str() + 7); if (window > n) window = n; continue; }
        if (mode.rfind("frag:", 0) == 0) {
            const size_t G = (size_t)atoll(mode.c_str() + 5);
            const char *c = strchr(mode.c_str() + 5, ':');
            const size_t K = c ? (size_t)atoll(c + 1) : 8;
            const size_t cnt = (G << 30) / (K << 20);
            std::vector<void *> all(cnt, nullptr);
            auto t0 = std::chrono::steady_clock::now();
            for (size_t i = 0; i < cnt; i++) CK(hipMalloc(&all[i], K << 20));
            for (size_t i = 0; i < cnt; i++) { if (i & 1) { CK(hipFree(all[i])); } else frag_keep.push_back(all[i]); }
            size_t fr = 0, tot = 0;
            CK(hipMemGetInfo(&fr, &tot));
            // take the untouched remainder (everything that was never handed out) away, so the next table has to come from the holes
            const size_t holes = (cnt / 2) * (K << 20);
            if (fr > holes + (4ull << 30)) {
                void *blk = nullptr;
                const size_t take = fr - holes - (2ull << 30);
                if (hipMalloc(&blk, take) == hipSuccess) frag_keep.push_back(blk);
                else printf("  (could not take the %.1f GiB remainder in one piece)\n", take / 1073741824.0);
            }
            CK(hipMemGetInfo(&fr, &tot));
            printf("frag: %zu x %zu MiB allocated, every other one freed (%.1f s); free now %.1f GiB\n", cnt, K,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), fr / 1073741824.0);
            fflush(stdout);
            continue;
        }
        if (mode == "unfrag") { for (void *p : frag_keep) CK(hipFree(p)); frag_keep.clear(); continue; }
        if (mode.rfind("hold:", 0) == 0) {
            // K tables alive AT ONCE (hipMalloc), each measured in turn, three rounds: is a landing good or bad for as long as it lives
            // (then index creation can draw K and keep the best), or does the same table move between measurements (then it is noise)?
            const int K = atoi(mode.c_str() + 5);
            std::vector<Table> tabs(K);
            for (int k = 0; k < K; k++) {
                alloc_table(tabs[k], n * dim * 4, "malloc");
                fill_kernel<<<4096, 256>>>(tabs[k].p, n * dim, 1u);
            }
            CK(hipDeviceSynchronize());
            uint64_t s = 42;
            for (uint64_t i = 0; i < P; i++) { h_pairs[2 * i] = (uint32_t)(sm64(s) % nq); h_pairs[2 * i + 1] = (uint32_t)(sm64(s) % window); }
            CK(hipMemcpy(d_pairs, h_pairs.data(), P * 8, hipMemcpyHostToDevice));
            for (int round = 0; round < 3; round++)
                for (int k = 0; k < K; k++) {
                    for (int i = 0; i < 2; i++) cz_distance_batch(CZ_COSINE, tabs[k].p, (uint32_t)n, dim, d_q, nq, d_pairs, P, d_out, CZ_DEVICE_PTRS, nullptr);
                    CK(hipDeviceSynchronize());
                    float sum = 0.f;
                    for (int i = 0; i < reps; i++) {
                        CK(hipEventRecord(e0, nullptr));
                        cz_distance_batch(CZ_COSINE, tabs[k].p, (uint32_t)n, dim, d_q, nq, d_pairs, P, d_out, CZ_DEVICE_PTRS, nullptr);
                        CK(hipEventRecord(e1, nullptr));
                        CK(hipEventSynchronize(e1));
                        float ms = 0;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        sum += ms;
                    }
                    printf("hold round %d table %d  va %p  avg %.3f ms  frac %.3f\n", round, k, (void *)tabs[k].p, sum / reps,
                           (double)P * 3072 / (sum / reps) / 1e9 / 8.0);
                    fflush(stdout);
                }
            for (auto &t : tabs) t.free_();
            continue;
        }
        Table t;
        auto t0 = std::chrono::steady_clock::now();
        alloc_table(t, n * dim * 4, mode);
        const double t_alloc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fill_kernel<<<4096, 256>>>(t.p, n * dim, 1u);
        CK(hipDeviceSynchronize());
        uint64_t s = 42;
        for (uint64_t i = 0; i < P; i++) { h_pairs[2 * i] = (uint32_t)(sm64(s) % nq); h_pairs[2 * i + 1] = (uint32_t)(sm64(s) % window); }
        CK(hipMemcpy(d_pairs, h_pairs.data(), P * 8, hipMemcpyHostToDevice));
        for (int i = 0; i < 3; i++)
            if (cz_distance_batch(CZ_COSINE, t.p, (uint32_t)n, dim, d_q, nq, d_pairs, P, d_out, CZ_DEVICE_PTRS, nullptr) != 0) { printf("cz_distance_batch: %s\n", cz_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0.f;
        for (int i = 0; i < reps; i++) {
            CK(hipEventRecord(e0, nullptr));
            cz_distance_batch(CZ_COSINE, t.p, (uint32_t)n, dim, d_q, nq, d_pairs, P, d_out, CZ_DEVICE_PTRS, nullptr);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        double chk = 0;
        std::vector<double> h_out(16);
        CK(hipMemcpy(h_out.data(), d_out, 16 * 8, hipMemcpyDeviceToHost));
        for (double v : h_out) chk += v;
        printf("%-16s window %9llu rows  va %p (low bits %#llx)  alloc %.2f s  avg %.3f ms  best %.3f ms  = %.2f / %.2f TB/s  frac %.3f / %.3f  chk %.6f\n",
               mode.c_str(), (unsigned long long)window, (void *)t.p, (unsigned long long)((uintptr_t)t.p & ((1ull << 30) - 1)), t_alloc,
               sum / reps, best, (double)P * 3072 / (sum / reps) / 1e9, (double)P * 3072 / best / 1e9,
               (double)P * 3072 / (sum / reps) / 1e9 / 8.0, (double)P * 3072 / best / 1e9 / 8.0, chk);
        fflush(stdout);
        t.free_();
    }
    return 0;
}
