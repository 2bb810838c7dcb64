// knn_gemm.hip -- exhaustive k-NN as a dense GEMM on the matrix cores (the one place on this path where a query batch
// turns distance into a true GEMM: B queries x N base rows x d, Cosine / IP; north_star keeps MFMA for exactly this).
//
//   dots[q][n] = sum_k Q[q][k] * X[n][k]           v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate: bit for bit a
//                                                  k-ordered fmaf chain per output element (one rounding per product)
//   distance    = 1 - dot / sqrt(|q|^2 |x|^2)  (Cosine)   |   1 - dot  (IP)       f64 tail as hnsw.rs:79-101
//   per query the k nearest of every column chunk (LDS rank-merge, topk.h), then the chunk lists are merged.
//
// L2 is NOT served here: the reference computes dot(a - b, a - b), which is not a GEMM; |a|^2 + |b|^2 - 2ab is a
// different (cancelling) arithmetic.  The squared norms are the same k-ordered fmaf chains (row_norms_seq_kernel), so
// the whole path has a CPU restatement (oracle ORC_DOT_SEQ) it can be compared with bit for bit.
// Used by cz_knn_bruteforce when the caller passes CZ_BF_GEMM (bench.py's recall ground truth); the default remains
// the streaming kernel whose summation tree equals the search kernel's.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "hnsw_index.h"
#include "topk.h"

using namespace czd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KT = 16, LDP = KT + 1;  // block tile, K step, padded LDS row (conflict-free)

// out[r] = fma chain over k of x[r][k]^2 (k ascending, starting from +0): one lane per row; the lanes of a wave walk 64
// consecutive rows, so every 128-byte line is fetched once and served from L1 for the following 7 chunks
__global__ void __launch_bounds__(256)
row_norms_seq_kernel(const float *__restrict__ x, uint32_t n, uint32_t ld, float *__restrict__ out) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float4 *row = (const float4 *)(x + (size_t)r * ld);
    float a = 0.f;
    for (uint32_t c = 0; c < ld / 4; c++) {
        const float4 v = row[c];
        a = fma_(v.x, v.x, a);
        a = fma_(v.y, v.y, a);
        a = fma_(v.z, v.z, a);
        a = fma_(v.w, v.w, a);
    }
    out[r] = a;
}

// C[m][n] = sum_k A[m][k] * B[n][k]; A [M][ld], B [N][ld] row-major, ld a multiple of 4 (zero padded);
// 256 threads = 4 waves, wave (wm, wn) owns a 64 x 64 quadrant = 2 x 2 MFMA tiles of 32 x 32
// FILTER = false: C[m][n] is written (the first stretch of columns, from which bf_select_kernel takes the lists that set the
// thresholds).  FILTER = true: nothing is written but the few products that can still enter a query's list -- the GEMM's epilogue
// compares every score (dot, or dot / |x| for Cosine: both distances fall as it rises) with the query's threshold `need` (the score
// of its k-th nearest so far, less a margin two orders above the rounding of the f32 score: see bf_need_kernel) and appends the
// survivors (column, dot) to the query's candidate list.  After a sample of n0 columns a later stretch of c columns leaves about
// k * c / n0 survivors per query, so the B x N products never reach memory: the selection that cost as much as the GEMM (a 1 GiB
// slab of products written, then read back) is gone.  NaN passes the filter.
struct CandOut {
    const float *need;    // [M] score threshold per query
    const float *xnorm;   // [n] squared norms of the base rows (Cosine), or null
    uint32_t col_base;    // id of column 0 of Bm
    uint32_t *cnt;        // [M] candidates appended per query (may run past cap: the caller checks)
    uint32_t *col;        // [M][cap]
    float *dot;           // [M][cap]
    uint32_t cap;
};
template <bool FILTER>
__global__ void __launch_bounds__(256)
dot_gemm_mfma_kernel(const float *__restrict__ A, uint32_t M, const float *__restrict__ Bm, uint32_t N, uint32_t ld,
                     float *__restrict__ C, uint64_t ldc, CandOut co) {
    __shared__ float As[BM * LDP];
    __shared__ float Bs[BN * LDP];
    __shared__ float need_s[BM];
    if (FILTER && threadIdx.x < BM) {
        const uint32_t qrow = (blockIdx.x % ((M + BM - 1) / BM)) * BM + threadIdx.x;
        need_s[threadIdx.x] = qrow < M ? co.need[qrow] : 0.f;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // FILTER: a 1-D grid with the QUERY tiles fastest -- the workgroups that share a tile of base rows are neighbours in
    // dispatch order and run at the same time, so the tile comes out of HBM once (the other XCDs find it in the Infinity
    // Cache); with the column tiles fastest every base row was fetched once per query tile, B / 128 times
    const uint32_t ny = (M + BM - 1) / BM;
    const uint32_t by = FILTER ? blockIdx.x % ny : blockIdx.y, bx = FILTER ? blockIdx.x / ny : blockIdx.x;
    const uint32_t m0 = by * BM, n0 = bx * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // the K loop is software pipelined through registers: the next 128 x 16 slices of A and B are requested before the
    // MFMAs of the current slice are issued, so the global latency hides behind 32 MFMAs (2 048 cycles) per wave
    float4 ra[2], rbv[2];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + 256 * i, r = idx >> 2, k4 = idx & 3;
            const uint32_t gk = k0 + 4 * k4;
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            rbv[i] = ra[i];
            if (gk < ld) {  // rows past the matrix / k past ld read as zero
                if (m0 + r < M) ra[i] = *(const float4 *)(A + (size_t)(m0 + r) * ld + gk);
                if (n0 + r < N) rbv[i] = *(const float4 *)(Bm + (size_t)(n0 + r) * ld + gk);
            }
        }
    };
    fetch(0);
    for (uint32_t k0 = 0; k0 < ld; k0 += KT) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + 256 * i, r = idx >> 2, k4 = idx & 3;
            float *as = As + r * LDP + 4 * k4, *bs = Bs + r * LDP + 4 * k4;
            as[0] = ra[i].x; as[1] = ra[i].y; as[2] = ra[i].z; as[3] = ra[i].w;
            bs[0] = rbv[i].x; bs[1] = rbv[i].y; bs[2] = rbv[i].z; bs[3] = rbv[i].w;
        }
        __syncthreads();
        if (k0 + KT < ld) fetch(k0 + KT);
#pragma unroll
        for (int kp = 0; kp < KT / 2; kp++) {
            // operand layout of v_mfma_f32_32x32x2_f32: lane l holds A[row = l % 32][k = l / 32] and B[k = l / 32][col = l % 32]
            float af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                af[t] = As[(wm * 64 + t * 32 + (lane & 31)) * LDP + kp * 2 + (lane >> 5)];
                bf[t] = Bs[(wn * 64 + t * 32 + (lane & 31)) * LDP + kp * 2 + (lane >> 5)];
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // accumulator layout: register r of lane l is C[row = 8 (r / 4) + 4 (l / 32) + r % 4][col = l % 32] of the tile
    if constexpr (FILTER) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t col = n0 + wn * 64 + j * 32 + (lane & 31);
            const bool col_ok = col < N;
            const float rs = (col_ok && co.xnorm) ? rsqrtf(co.xnorm[co.col_base + col]) : 1.0f;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t rl = wm * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    const float v = acc[i][j][r];
                    if (col_ok && m0 + rl < M && !(v * rs < need_s[rl])) {  // (a handful per query and stretch: the atomics are rare)
                        const uint32_t at = atomicAdd(&co.cnt[m0 + rl], 1u);
                        if (at < co.cap) {
                            co.col[(size_t)(m0 + rl) * co.cap + at] = co.col_base + col;
                            co.dot[(size_t)(m0 + rl) * co.cap + at] = v;
                        }
                    }
                }
        }
    } else {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t row = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (row < M && col < N) C[(size_t)row * ldc + col] = acc[i][j][r];
            }
        }
    }
}

// the score a column has to reach to enter query q's list, from the list so far (ids / dist [B][k] ascending, CZ_NONE padded):
// its k-th distance d gives s = 1 - d (IP) or (1 - d) * |q| (Cosine: dot / |x|), lowered by 1e-4 relative -- two orders above the
// rounding of the f32 score -- and rounded DOWN to f32; a list that is not full yet (or ends in NaN) lets everything through
__global__ void __launch_bounds__(256)
bf_need_kernel(int metric, const uint32_t *__restrict__ ids, const double *__restrict__ dist, uint32_t B, uint32_t k,
               const float *__restrict__ qnorm, float *__restrict__ need) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= B) return;
    float nd = -__builtin_inff();
    const double d = dist[(size_t)q * k + k - 1];
    if (ids[(size_t)q * k + k - 1] != CZ_NONE && d == d) {
        const double sk = metric == CZ_COSINE ? (1.0 - d) * sqrt((double)qnorm[q]) : (1.0 - d);
        const double lim = sk - 1e-4 * fabs(sk) - 1e-30;
        nd = __double2float_rd(lim);
    }
    need[q] = nd;
}

// one query's candidates -> its k nearest among them, as one more partial list (slot `part` of total_parts)
__global__ void __launch_bounds__(256)
bf_cand_kernel(int metric, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ col, const float *__restrict__ dot, uint32_t cap,
               const float *__restrict__ xnorm, const float *__restrict__ qnorm, uint32_t k, uint32_t part, uint32_t total_parts,
               uint64_t *__restrict__ part_key, uint32_t *__restrict__ part_id) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t *ckey = (uint64_t *)smem_raw;      // [256]
    uint32_t *cid = (uint32_t *)(ckey + 256);    // [256]
    uint64_t *tkey = (uint64_t *)(cid + 256);    // [k]
    uint32_t *tid_ = (uint32_t *)(tkey + k);
    const int tid = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    for (uint32_t i = tid; i < k; i += 256) {
        tkey[i] = ~0ull;
        tid_[i] = CZ_NONE;
    }
    const uint32_t n = min(cnt[qi], cap);
    const float qn = metric == CZ_COSINE ? qnorm[qi] : 0.f;
    int tcnt = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < n; p0 += 256) {
        const uint32_t p = p0 + tid;
        if (p < n) {
            const uint32_t c = col[(size_t)qi * cap + p];
            const float bn = metric == CZ_COSINE ? xnorm[c] : 0.f;
            ckey[tid] = dist_key(finish_distance(metric, dot[(size_t)qi * cap + p], bn, qn));
            cid[tid] = c;
        }
        tcnt = topk_merge_batch((int)min(256u, n - p0), ckey, cid, tkey, tid_, tcnt, (int)k);
    }
    const size_t o = ((size_t)qi * total_parts + part) * k;
    for (uint32_t i = tid; i < k; i += 256) {
        part_key[o + i] = tkey[i];
        part_id[o + i] = tid_[i];
    }
}

// per (column chunk, query): distances from the dot products, the k nearest of the chunk into the partial lists.
// 1024 columns per step (4 per thread, all four loads in flight).  A conservative f32 pre-filter decides which columns
// can still enter the list: both metrics are decreasing in the score s = dot / sqrt(|x|^2) (Cosine) or dot (IP); once the
// list is full its k-th distance gives the score a column has to reach, and a column whose f32 score misses it by more
// than 1e-4 relative (two orders above the rounding of s) skips the f64 tail.  The survivors (a handful per step after
// the first) are compacted into an LDS buffer and rank-merged in one go when the buffer runs full or the chunk ends --
// merging every 256 columns made this kernel slower than the GEMM that feeds it.  NaN passes the filter.
constexpr int kSelCols = 1024, kSelCand = 2048;

__global__ void __launch_bounds__(256)
bf_select_kernel(int metric, const float *__restrict__ dots, uint64_t ldc, uint32_t n_cols, uint32_t col_base,
                 const float *__restrict__ xnorm, const float *__restrict__ qnorm, uint32_t k, uint32_t cols_per_chunk,
                 uint32_t chunk_base, uint32_t total_chunks, uint64_t *__restrict__ part_key, uint32_t *__restrict__ part_id) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t *ckey = (uint64_t *)smem_raw;         // [kSelCand] candidates that passed the filter
    uint32_t *cid = (uint32_t *)(ckey + kSelCand);  // [kSelCand]
    uint64_t *tkey = (uint64_t *)(cid + kSelCand);  // [k] the list
    uint32_t *tid_ = (uint32_t *)(tkey + k);
    int *ctl = (int *)(tid_ + k);                    // [0] candidate count
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t qi = blockIdx.y;
    const uint32_t c0 = blockIdx.x * cols_per_chunk, c1 = min(n_cols, c0 + cols_per_chunk);
    for (uint32_t i = tid; i < k; i += 256) {
        tkey[i] = ~0ull;
        tid_[i] = CZ_NONE;
    }
    if (tid == 0) ctl[0] = 0;
    const float qn = metric == CZ_COSINE ? qnorm[qi] : 0.f;
    const double sqn = sqrt((double)qn);
    const float *row = dots + (size_t)qi * ldc;
    int tcnt = 0, ccnt = 0;
    __syncthreads();
    for (uint32_t b0 = c0; b0 < c1; b0 += kSelCols) {
        double need = -__builtin_inf();
        if (tcnt >= (int)k && tkey[k - 1] != ~0ull) {
            const double sk = metric == CZ_COSINE ? (1.0 - key_dist(tkey[k - 1])) * sqn : (1.0 - key_dist(tkey[k - 1]));
            need = sk - 1e-4 * fabs(sk) - 1e-30;
        }
        float dot[4], bn[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t c = b0 + tid + 256 * i;
            dot[i] = c < c1 ? row[c] : 0.f;
            bn[i] = (c < c1 && metric == CZ_COSINE) ? xnorm[col_base + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t c = b0 + tid + 256 * i;
            const float sc = metric == CZ_COSINE ? dot[i] * rsqrtf(bn[i]) : dot[i];
            const bool pass = c < c1 && !((double)sc < need);
            const unsigned long long m = __ballot(pass);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&ctl[0], __popcll(m));
            base = __shfl(base, 0, 64);
            if (pass) {
                const int p = base + __popcll(m & ((1ull << lane) - 1ull));
                ckey[p] = dist_key(finish_distance(metric, dot[i], bn[i], qn));
                cid[p] = col_base + c;
            }
        }
        __syncthreads();
        ccnt = ctl[0];
        // flush when the list is still filling (the filter is open) or another step could overflow the buffer
        if (ccnt > 0 && (tcnt < (int)k || ccnt > kSelCand - kSelCols || b0 + kSelCols >= c1)) {
            for (int p = 0; p < ccnt; p += 256) tcnt = topk_merge_batch(min(256, ccnt - p), ckey + p, cid + p, tkey, tid_, tcnt, (int)k);
            if (tid == 0) ctl[0] = 0;
            __syncthreads();
        }
    }
    const size_t o = ((size_t)qi * total_chunks + chunk_base + blockIdx.x) * k;
    for (uint32_t i = tid; i < k; i += 256) {
        part_key[o + i] = tkey[i];
        part_id[o + i] = tid_[i];
    }
}

}  // namespace

namespace cz {

// d_q [B][dim] device; d_ids / d_dist [B][k] device.  part_key / part_id are provided by the caller's merge step.
int knn_gemm_device(HnswIndex *ix, const float *d_q, uint32_t B, uint32_t k, uint32_t *d_ids, double *d_dist,
                    hipStream_t stream, void (*merge)(const uint64_t *, const uint32_t *, uint32_t, uint32_t, uint32_t,
                                                      uint32_t *, double *, hipStream_t)) {
    if (ix->metric == CZ_L2)
        return set_error(CZ_E_UNSUPPORTED, "the GEMM form of the exhaustive scan serves Cosine and IP; L2 = dot(a - b, a - b) is not a GEMM");
    const uint32_t n = ix->n, ld = ix->ld;
    // Stage 0: the first `slab` columns as B x slab f32 dot products (<= 1 GiB), chunks of 8192 columns for the selection.
    // Stages 1, 2, ...: stretches of up to 8 x the columns seen so far through the GEMM whose epilogue keeps only what can
    // still enter a list (CandOut): about 8 k survivors per query and stage; 16 k + 1024 slots each, and a stage whose buffer
    // overflows (data sorted towards the queries) is done again the stage-0 way.  CZ_BF_FUSE=0: every stretch the stage-0 way.
    const uint32_t cols_per_chunk = 8192;
    uint32_t slab = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>(cols_per_chunk, ((1ull << 30) / 4 / B) / cols_per_chunk * cols_per_chunk));
    if (const char *sl = getenv("CZ_BF_SLAB"))  // (tests: a small first stretch, so that small corpora go through the fused stages)
        if (atoi(sl) > 0) slab = (uint32_t)std::min<uint64_t>(n, (uint64_t)atoi(sl));
    slab = (slab + BN - 1) / BN * BN;
    const char *fuse_env = getenv("CZ_BF_FUSE");
    const bool fuse = !(fuse_env && atoi(fuse_env) == 0);
    // the plan: (begin, count, fused?) stretches
    struct Stretch {
        uint32_t c0, nc;
        bool fused;
    };
    std::vector<Stretch> plan;
    for (uint32_t c0 = 0; c0 < n;) {
        if (c0 == 0 || !fuse) {
            const uint32_t nc = std::min(slab, n - c0);
            plan.push_back({c0, nc, false});
            c0 += nc;
        } else {
            const uint32_t nc = (uint32_t)std::min<uint64_t>((uint64_t)n - c0, 8ull * c0);
            plan.push_back({c0, nc, true});
            c0 += nc;
        }
    }
    // partial lists: one per 8192-column chunk of an unfused stretch, one per fused stretch (+ room to redo every fused
    // stretch unfused)
    auto parts_of = [&](uint32_t count) {  // lists an unfused pass over `count` columns writes: per slab, one per 8192-column chunk
        uint32_t parts = 0;
        for (uint32_t done = 0; done < count; done += slab) parts += (std::min(slab, count - done) + cols_per_chunk - 1) / cols_per_chunk;
        return parts;
    };
    uint32_t total_parts = 0;
    for (const Stretch &st : plan) total_parts += parts_of(st.nc) + (st.fused ? 1 : 0);
    const uint32_t cap = 16 * k + 1024;
    DevBuf<float> dots, xnorm, qnorm, qpad, need, cdot;
    DevBuf<uint64_t> pkey;
    DevBuf<uint32_t> pid, ccnt, ccol, tids;
    DevBuf<double> tdist;
    CZ_HIP(dots.alloc((size_t)B * slab));
    CZ_HIP(pkey.alloc((size_t)B * total_parts * k));
    CZ_HIP(pid.alloc((size_t)B * total_parts * k));
    CZ_HIP(hipMemsetAsync(pkey.p, 0xFF, (size_t)B * total_parts * k * 8, stream));  // empty lists: key ~0, id CZ_NONE
    CZ_HIP(hipMemsetAsync(pid.p, 0xFF, (size_t)B * total_parts * k * 4, stream));
    const float *q = d_q;
    if (ld != ix->dim) {  // pad the queries like the base rows (zero tail)
        CZ_HIP(qpad.alloc((size_t)B * ld));
        CZ_HIP(hipMemsetAsync(qpad.p, 0, (size_t)B * ld * 4, stream));
        CZ_HIP(hipMemcpy2DAsync(qpad.p, (size_t)ld * 4, d_q, (size_t)ix->dim * 4, (size_t)ix->dim * 4, B, hipMemcpyDeviceToDevice, stream));
        q = qpad.p;
    }
    if (ix->metric == CZ_COSINE) {
        CZ_HIP(xnorm.alloc(n));
        CZ_HIP(qnorm.alloc(B));
        hipLaunchKernelGGL(row_norms_seq_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ix->vec, n, ld, xnorm.p);
        hipLaunchKernelGGL(row_norms_seq_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, q, B, ld, qnorm.p);
    }
    if (plan.size() > 1 && fuse) {
        CZ_HIP(need.alloc(B));
        CZ_HIP(ccnt.alloc(B));
        CZ_HIP(ccol.alloc((size_t)B * cap));
        CZ_HIP(cdot.alloc((size_t)B * cap));
        CZ_HIP(tids.alloc((size_t)B * k));
        CZ_HIP(tdist.alloc((size_t)B * k));
    }
    const size_t smem = (size_t)kSelCand * 12 + (size_t)k * 12 + 16;
    const size_t smem_cand = (size_t)256 * 12 + (size_t)k * 12 + 16;
    uint32_t part = 0;
    const CandOut none{nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0};
    auto unfused = [&](uint32_t c_begin, uint32_t count) {  // GEMM -> products -> bf_select_kernel, a slab at a time
        for (uint32_t c0 = c_begin; c0 < c_begin + count; c0 += slab) {
            const uint32_t nc = std::min(slab, c_begin + count - c0);
            const uint32_t chunks = (nc + cols_per_chunk - 1) / cols_per_chunk;
            hipLaunchKernelGGL(dot_gemm_mfma_kernel<false>, dim3((nc + BN - 1) / BN, (B + BM - 1) / BM), dim3(256), 0, stream, q, B,
                               ix->vec + (size_t)c0 * ld, nc, ld, dots.p, (uint64_t)slab, none);
            hipLaunchKernelGGL(bf_select_kernel, dim3(chunks, B), dim3(256), smem, stream, ix->metric, dots.p, (uint64_t)slab, nc, c0,
                               xnorm.p, qnorm.p, k, cols_per_chunk, part, total_parts, pkey.p, pid.p);
            part += chunks;
        }
    };
    for (const Stretch &st : plan) {
        if (!st.fused) {
            unfused(st.c0, st.nc);
            continue;
        }
        // thresholds from everything seen so far
        merge(pkey.p, pid.p, B, total_parts, k, tids.p, tdist.p, stream);
        hipLaunchKernelGGL(bf_need_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ix->metric, tids.p, tdist.p, B, k, qnorm.p, need.p);
        CZ_HIP(hipMemsetAsync(ccnt.p, 0, (size_t)B * 4, stream));
        const CandOut co{need.p, ix->metric == CZ_COSINE ? xnorm.p : nullptr, st.c0, ccnt.p, ccol.p, cdot.p, cap};
        // (the grid's x extent is the column tiles: a stretch of up to ~9M columns = 70 k of them)
        hipLaunchKernelGGL(dot_gemm_mfma_kernel<true>, dim3(((st.nc + BN - 1) / BN) * ((B + BM - 1) / BM)), dim3(256), 0, stream, q, B,
                           ix->vec + (size_t)st.c0 * ld, st.nc, ld, (float *)nullptr, (uint64_t)0, co);
        std::vector<uint32_t> h_cnt(B);
        CZ_HIP(hipMemcpyAsync(h_cnt.data(), ccnt.p, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        CZ_HIP(hipStreamSynchronize(stream));
        uint32_t worst = 0;
        for (uint32_t c : h_cnt) worst = std::max(worst, c);
        if (worst > cap) {  // the buffer of some query ran over: this stretch again, the plain way (correct whatever the data)
            unfused(st.c0, st.nc);
            continue;
        }
        hipLaunchKernelGGL(bf_cand_kernel, dim3(B), dim3(256), smem_cand, stream, ix->metric, ccnt.p, ccol.p, cdot.p, cap, xnorm.p, qnorm.p, k,
                           part, total_parts, pkey.p, pid.p);
        part += 1;
    }
    merge(pkey.p, pid.p, B, total_parts, k, d_ids, d_dist, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(CZ_E_HIP, "GEMM exhaustive scan launch: %s", hipGetErrorString(e));
    CZ_HIP(hipStreamSynchronize(stream));  // temporaries die with this scope
    return CZ_OK;
}

}  // namespace cz
