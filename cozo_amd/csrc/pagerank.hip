// pagerank.hip -- PageRank fixed rule on gfx950 (C ABI: cz_pagerank, cz_pagerank_plan_*).
//
// Reference: PageRank::run (cozo-core/src/fixed_rule/algos/pagerank.rs:29-56) -> graph 0.3.1 `page_rank`:
//   init = 1/N, base = (1-d)/N, contrib[v] = score[v]/out_degree(v); per iteration every node u
//   new = base + d * sum_{v in in(u)} contrib[v]   (f32, in-neighbours summed SEQUENTIALLY in sorted order)
//   err += |new - old| (f64); contrib refreshed after the sweep (Jacobi); stop at err < tol or max_iter.
//
// Kernel: CSR-stream pull SpMV.  A workgroup owns a run of consecutive rows whose in-edges fit one LDS
// tile (kTileNnz entries).  Phase 1 streams the tile's source ids with coalesced loads and gathers
// contrib[src] into LDS (every lane busy, many gathers in flight); phase 2 gives each row to one lane,
// which adds its LDS segment in order -- the same sequential f32 order as the reference, so the scores
// are bit-identical to it -- and runs the fused epilogue (new score, |delta| in f64, next contribution).
// Rows longer than a tile are streamed tile by tile and summed by one lane, still in order.
// Algorithmic HBM bytes per iteration: 4E (ids) + 4(N+1) (offsets) + 20N (contrib in/out, score
// in/out, out-degree)  = 6.4 B/edge at N = 10M, E = 100M  (SURVEY.md section 8d).
#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileNnz = 4096;   // f32 values per LDS tile (16 KiB)
constexpr int kMaxRowsPerBlock = 1024;

struct RowBlock {
    uint32_t row0, row1;  // local rows [row0, row1)
};

__global__ void __launch_bounds__(kThreads)
pr_init_kernel(float *__restrict__ contrib, const uint32_t *__restrict__ out_deg, uint32_t N, float init) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x)
        contrib[v] = init / (float)out_deg[v];
}

__global__ void __launch_bounds__(kThreads) pr_fill_kernel(float *__restrict__ p, uint32_t n, float v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

__device__ __forceinline__ double block_sum_f64(double v, double *red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kThreads / 64; w++) s += red[w];
    return s;  // valid on thread 0
}

__global__ void __launch_bounds__(kThreads)
pr_step_kernel(const RowBlock *__restrict__ blocks, const uint32_t *__restrict__ off /* local, [rows+1] */,
               const uint32_t *__restrict__ src, const uint32_t *__restrict__ out_deg /* global ids */,
               uint32_t row_begin, const float *__restrict__ contrib_in, float *__restrict__ contrib_out,
               float *__restrict__ scores /* local */, float base, float damping, double *__restrict__ partial) {
    __shared__ float tile[kTileNnz];
    __shared__ double red[kThreads / 64];
    const RowBlock rb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t e0 = off[rb.row0], e1 = off[rb.row1];
    double err = 0.0;
    if (e1 - e0 <= (uint32_t)kTileNnz) {
        // phase 1: coalesced id stream + gather
        const uint32_t nnz = e1 - e0;
        uint32_t i = tid;
        for (; i + 3 * kThreads < nnz; i += 4 * kThreads) {
            uint32_t s0 = src[e0 + i], s1 = src[e0 + i + kThreads], s2 = src[e0 + i + 2 * kThreads],
                     s3 = src[e0 + i + 3 * kThreads];
            float c0 = contrib_in[s0], c1 = contrib_in[s1], c2 = contrib_in[s2], c3 = contrib_in[s3];
            tile[i] = c0;
            tile[i + kThreads] = c1;
            tile[i + 2 * kThreads] = c2;
            tile[i + 3 * kThreads] = c3;
        }
        for (; i < nnz; i += kThreads) tile[i] = contrib_in[src[e0 + i]];
        __syncthreads();
        // phase 2: one lane per row, sequential sum, fused epilogue
        for (uint32_t r = rb.row0 + tid; r < rb.row1; r += kThreads) {
            const uint32_t a = off[r] - e0, b = off[r + 1] - e0;
            float s = 0.0f;
            for (uint32_t e = a; e < b; e++) s = s + tile[e];
            const float old = scores[r];
            const float nw = base + damping * s;  // two roundings, like the reference (no fma: -ffp-contract=off)
            scores[r] = nw;
            contrib_out[row_begin + r] = nw / (float)out_deg[row_begin + r];
            err += fabs((double)(nw - old));
        }
    } else {
        // a single long row: stream it tile by tile, lane 0 adds in order
        const uint32_t r = rb.row0;
        float s = 0.0f;
        for (uint32_t t0 = e0; t0 < e1; t0 += kTileNnz) {
            const uint32_t nnz = min((uint32_t)kTileNnz, e1 - t0);
            for (uint32_t i = tid; i < nnz; i += kThreads) tile[i] = contrib_in[src[t0 + i]];
            __syncthreads();
            if (tid == 0)
                for (uint32_t e = 0; e < nnz; e++) s = s + tile[e];
            __syncthreads();
        }
        if (tid == 0) {
            const float old = scores[r];
            const float nw = base + damping * s;
            scores[r] = nw;
            contrib_out[row_begin + r] = nw / (float)out_deg[row_begin + r];
            err = fabs((double)(nw - old));
        }
    }
    const double total = block_sum_f64(err, red);
    if (tid == 0) partial[blockIdx.x] = total;
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

}  // namespace

struct cz_pagerank_plan {
    uint32_t N = 0, row_begin = 0, rows = 0;
    uint64_t E = 0;
    float damping = 0, base = 0, init = 0;
    uint32_t n_blocks = 0;
    RowBlock *d_blocks = nullptr;
    uint32_t *d_off = nullptr, *d_src = nullptr, *d_outdeg = nullptr;
    float *d_scores = nullptr;
    double *d_partial = nullptr;
    ~cz_pagerank_plan() {
        if (d_blocks) (void)hipFree(d_blocks);
        if (d_off) (void)hipFree(d_off);
        if (d_src) (void)hipFree(d_src);
        if (d_outdeg) (void)hipFree(d_outdeg);
        if (d_scores) (void)hipFree(d_scores);
        if (d_partial) (void)hipFree(d_partial);
    }
};

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
    if (E >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "a shard holds at most 2^32-2 edges");
    std::unique_ptr<cz_pagerank_plan> p(new cz_pagerank_plan());
    p->N = N;
    p->row_begin = row_begin;
    p->rows = rows;
    p->E = E;
    p->damping = damping;
    p->init = N ? 1.0f / (float)N : 0.f;
    p->base = N ? (1.0f - damping) / (float)N : 0.f;
    // row blocks: consecutive rows whose edges fit one LDS tile
    std::vector<RowBlock> blocks;
    blocks.reserve((size_t)(E / kTileNnz) + rows / kMaxRowsPerBlock + 16);
    uint32_t r = 0;
    while (r < rows) {
        uint32_t r1 = r + 1;
        if (in_offsets[r1] - in_offsets[r] <= (uint32_t)kTileNnz) {
            const uint32_t lim = std::min<uint32_t>(rows, r + kMaxRowsPerBlock);
            while (r1 < lim && in_offsets[r1 + 1] - in_offsets[r] <= (uint32_t)kTileNnz) r1++;
        }
        for (uint32_t q = r; q < r1; q++)
            if (in_offsets[q + 1] < in_offsets[q]) return cz::set_error(CZ_E_INVALID, "in_offsets not monotone at row %u", q);
        blocks.push_back({r, r1});
        r = r1;
    }
    p->n_blocks = (uint32_t)blocks.size();
    CZ_HIP(hipMalloc((void **)&p->d_blocks, std::max<size_t>(1, blocks.size()) * sizeof(RowBlock)));
    CZ_HIP(hipMalloc((void **)&p->d_off, ((size_t)rows + 1) * 4));
    CZ_HIP(hipMalloc((void **)&p->d_src, std::max<uint64_t>(1, E) * 4));
    CZ_HIP(hipMalloc((void **)&p->d_outdeg, std::max<size_t>(1, N) * 4));
    CZ_HIP(hipMalloc((void **)&p->d_scores, std::max<size_t>(1, rows) * 4));
    CZ_HIP(hipMalloc((void **)&p->d_partial, std::max<size_t>(1, blocks.size()) * 8));
    if (!blocks.empty()) CZ_HIP(hipMemcpy(p->d_blocks, blocks.data(), blocks.size() * sizeof(RowBlock), hipMemcpyHostToDevice));
    if (rows) CZ_HIP(hipMemcpy(p->d_off, dev ? dev_off : in_offsets, ((size_t)rows + 1) * 4, up));
    else {
        uint32_t z = 0;
        CZ_HIP(hipMemcpy(p->d_off, &z, 4, hipMemcpyHostToDevice));
    }
    if (E) CZ_HIP(hipMemcpy(p->d_src, in_sources, E * 4, up));
    if (N) CZ_HIP(hipMemcpy(p->d_outdeg, out_degree, (size_t)N * 4, up));
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
    hipLaunchKernelGGL(pr_init_kernel, dim3(2048), dim3(kThreads), 0, stream, contrib_dev, p->d_outdeg, p->N, p->init);
    if (p->rows) hipLaunchKernelGGL(pr_fill_kernel, dim3(2048), dim3(kThreads), 0, stream, p->d_scores, p->rows, p->init);
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
    if (p->n_blocks == 0) return CZ_OK;
    hipLaunchKernelGGL(pr_step_kernel, dim3(p->n_blocks), dim3(kThreads), 0, stream, p->d_blocks, p->d_off, p->d_src,
                       p->d_outdeg, p->row_begin, contrib_in_dev, contrib_out_dev, p->d_scores, p->base, p->damping,
                       p->d_partial);
    hipLaunchKernelGGL(pr_err_reduce_kernel, dim3(1), dim3(1024), 0, stream, p->d_partial, p->n_blocks, err_out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "pagerank step launch: %s", hipGetErrorString(e));
    return CZ_OK;
}

extern "C" float *cz_pagerank_plan_scores(cz_pagerank_plan *p) { return p ? p->d_scores : nullptr; }
extern "C" uint64_t cz_pagerank_plan_edges(const cz_pagerank_plan *p) { return p ? p->E : 0; }

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

// one-shot form: upload, iterate with the reference's stopping rule, download
extern "C" int cz_pagerank(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                           uint64_t E, float damping, double tolerance, uint32_t max_iter, float *scores,
                           uint32_t *iters_run, double *final_err, const volatile uint8_t *poison) {
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (N == 0) return CZ_OK;  // pagerank.rs:43-45 empty input -> empty output
    if (!scores) return cz::set_error(CZ_E_INVALID, "null scores");
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "iterations must be positive");
    if (in_offsets && in_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "in_offsets[N] (%u) != E (%llu)", in_offsets[N], (unsigned long long)E);
    cz_pagerank_plan *plan = nullptr;
    int rc = cz_pagerank_plan_create(in_offsets, in_sources, out_degree, N, 0, N, damping, &plan, 0);
    if (rc) return rc;
    std::unique_ptr<cz_pagerank_plan> guard(plan);
    cz::DevBuf<float> c0, c1;
    cz::DevBuf<double> derr;
    CZ_HIP(c0.alloc(N));
    CZ_HIP(c1.alloc(N));
    CZ_HIP(derr.alloc(1));
    rc = cz_pagerank_plan_init(plan, c0.p, nullptr);
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
    CZ_HIP(hipMemcpy(scores, plan->d_scores, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return CZ_OK;
}
