// topk.h -- running top-k list of a 256-thread workgroup: rank-merge of a batch of up to 256 (key, id) entries into a
// sorted list in LDS (capacity k <= 1024).  Shared by the exhaustive-scan kernels (streaming and GEMM forms).
#pragma once
#include "hnsw_kernels.h"

// bkey/bid[0..nb): the batch (ids distinct from each other and from the list => (key, id) is a strict order);
// tkey/tid_[0..tcnt): the list, ascending by (key, id).  Returns the new list length (uniform).  Starts and ends with
// the batch / list visible to all threads (barriers inside).
__device__ __forceinline__ int topk_merge_batch(int nb, uint64_t *bkey, uint32_t *bid, uint64_t *tkey, uint32_t *tid_,
                                                int tcnt, int k) {
    const int tid = threadIdx.x;
    __syncthreads();
    // rank-merge the batch into the sorted top list (capacity k); ids are distinct => strict order
    const uint64_t bound_k = tcnt >= k ? tkey[k - 1] : ~0ull;
    const uint32_t bound_i = tcnt >= k ? tid_[k - 1] : CZ_NONE;
    uint64_t mk = 0;
    uint32_t mi = CZ_NONE;
    bool elig = false;
    if (tid < nb) {
        mk = bkey[tid];
        mi = bid[tid];
        elig = tcnt < k || czh::key_lt(mk, mi, bound_k, bound_i);
    }
    int nelig = __syncthreads_count(elig);
    if (nelig > 0) {
        if (tid < nb && !elig) bid[tid] = CZ_NONE;
        __syncthreads();
        // positions of old entries (each thread owns entries tid, tid+256, ...)
        int npos = -1;
        if (elig) {
            int r1c = 0;
            for (int t = 0; t < nb; t++) {
                uint32_t ni = bid[t];
                if (ni != CZ_NONE && czh::key_lt(bkey[t], ni, mk, mi)) r1c++;
            }
            int lo = 0, hi = tcnt;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (czh::key_lt(tkey[mid], tid_[mid], mk, mi)) lo = mid + 1;
                else hi = mid;
            }
            npos = r1c + lo;
        }
        constexpr int R = 4;  // k <= 1024
        uint64_t wk[R];
        uint32_t wi[R];
        int wpos[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            int j = tid + r * 256;
            wpos[r] = -1;
            if (j < tcnt) {
                wk[r] = tkey[j];
                wi[r] = tid_[j];
                int sft = 0;
                for (int t = 0; t < nb; t++) {
                    uint32_t ni = bid[t];
                    if (ni != CZ_NONE && czh::key_lt(bkey[t], ni, wk[r], wi[r])) sft++;
                }
                if (sft > 0) wpos[r] = j + sft;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++)
            if (wpos[r] >= 0 && wpos[r] < k) {
                tkey[wpos[r]] = wk[r];
                tid_[wpos[r]] = wi[r];
            }
        if (elig && npos < k) {
            tkey[npos] = mk;
            tid_[npos] = mi;
        }
        tcnt = min(k, tcnt + nelig);
    }
    __syncthreads();
    return tcnt;
}
