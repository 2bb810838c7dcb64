// hnsw_kernels.h -- HNSW layer search for gfx950: one workgroup per query.
//
// Restates hnsw_search_level / hnsw_knn (cozo-core/src/runtime/hnsw.rs:539-587, 869-1012) in a form
// that maps to a CDNA4 workgroup:
//   * the reference keeps two priority queues (candidates: min, found_nn: max, capped at ef).  Every
//     candidate is pushed to both at once and an element evicted from found_nn can never be expanded
//     (it is farther than the current ef-th best, which only shrinks), so the pair is equivalent to
//     ONE list W, sorted ascending by (distance, id), holding at most ef entries with an "expanded"
//     flag: the next candidate is the nearest un-expanded entry, the loop ends when there is none.
//     (Differs from the reference only when two distinct nodes have bit-identical distances at the
//     eviction boundary, where the reference's priority-queue crate is itself unspecified.)
//   * processing the neighbours of a candidate one by one against a running `furthest` equals merging
//     the whole neighbour batch into W and truncating at ef (streaming top-k).
//   * W lives in LDS; neighbour ids are fetched with one coalesced row load; `visited` is a per-query
//     open-addressing hash set of node ids in global memory (CAS at L2; 128 KiB per query at ef = 192, so the
//     1024 tables of a batch stay in L2 / Infinity Cache -- the per-query BITMAP it replaces is 1.25 MB per query
//     at n = 10M: every test-and-set was an HBM miss and the rows had to be re-zeroed per launch), with the
//     bitmap kept as the overflow form; distances use the wave-wide routines of
//     distance.h, U rows in flight per lane group; the merge computes final positions by rank
//     (binary search over W for new entries, linear count over the <= 64 new entries for W entries).
#pragma once
#include "distance.h"
#include "distance_f64.h"

namespace czh {

using namespace czd;

// Phase timing (profiling builds only: -DCZ_PHASE_TIMING, see scratch/phase_prof.sh): thread 0 of every workgroup
// stamps the shader clock at the phase boundaries of the level-0 loop and the totals are added up per launch.
#ifdef CZ_PHASE_TIMING
__device__ unsigned long long cz_phase_cycles[12];
#define CZ_PH_DECL unsigned long long ph_t0 = 0, ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CZ_PH_START() do { if (tid == 0) ph_t0 = clock64(); } while (0)
#define CZ_PH_MARK(i) do { if (tid == 0) { unsigned long long t_ = clock64(); if (ph_t0) ph_acc[i] += t_ - ph_t0; ph_t0 = t_; } } while (0)
#define CZ_PH_COUNT(i, v) do { if (tid == 0) ph_acc[i] += (v); } while (0)
#define CZ_PH_FLUSH() do { if (tid == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd(&cz_phase_cycles[i_], ph_acc[i_]); } while (0)
#else
#define CZ_PH_DECL
#define CZ_PH_START() do {} while (0)
#define CZ_PH_MARK(i) do {} while (0)
#define CZ_PH_COUNT(i, v) do {} while (0)
#define CZ_PH_FLUSH() do {} while (0)
#endif

// consistency checks of the index construction (debug builds: -DCZ_BUILD_CHECKS): a message and a trap
#ifdef CZ_BUILD_CHECKS
#define CZ_CHECK(cond, ...) do { if (!(cond)) { printf("CZ_CHECK " __VA_ARGS__); __builtin_trap(); } } while (0)
#else
#define CZ_CHECK(cond, ...) do {} while (0)
#endif

constexpr uint32_t kExpanded = 0x80000000u;
constexpr uint32_t kIdMask = 0x7FFFFFFFu;
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kVlogCap = 2048;
// list capacity (x 256) and batch size (x 64) the register form of the merge covers; larger searches (ef > 512 or
// link rows wider than 128) take the ranked form (merge_wide)
constexpr int kMergeR = 2, kMergeEC = 2;

struct IndexDev {
    const float *vec;        // [n][ld]
    const double *vec64;     // [n][ld] of an F64 index (VecElementType::F64; then vec is null and ld = dim rounded up to 2)
    uint32_t n, dim, ld;     // ld = dim rounded up to 4
    int metric;
    const uint32_t *nbr0;    // [n][w0]
    int w0;
    const uint32_t *up_base; // [n]  first upper row of the node (CZ_NONE: top level 0)
    const uint32_t *up_nbrs; // [rows][wu]; row = up_base[node] + (level-1)
    int wu;
    int n_levels;
    uint32_t entry;
};

// LDS carve-up (all offsets multiples of 16 bytes)
struct Smem {
    uint64_t *wkey;   // [efcap]
    uint32_t *wid;    // [efcap]   id | kExpanded
    uint64_t *nkey;   // [wpad]
    uint32_t *nid;    // [wpad]    CZ_NONE = not eligible
    uint32_t *todo;   // [wpad]   ids to evaluate in this step
    uint32_t *vlog;   // [kVlogCap]
    float4 *q;        // [ld/4]
    int *ctl;         // [16] control words
    // select-neighbours heuristic scratch (index construction only)
    uint32_t *tpos;   // [efcap] list position of each pending candidate
    uint32_t *sel;    // [256]   list positions of the selected neighbours
    uint8_t *st;      // [efcap] 0 pending, 1 selected, 2 rejected
};
enum { C_CNT = 0, C_TODO = 1, C_LO = 2, C_VLOG = 3, C_NELIG = 4, C_NDIST_LO = 5, C_NDIST_HI = 6, C_KEEP = 7, C_NUM = 8,
       C_WAVE0 = 9 /* .. C_WAVE0 + kWaves - 1: per-wave counts of the merge's compaction */,
       C_VCNT = 13 /* members of the visited set */, C_VMODE = 14 /* 0 hash set, 1 bitmap */, C_WORDS = 16 };

// `build`: the select-neighbours scratch (tpos / sel) exists only in the index-construction kernels; a search carries `st` alone
// (the pass flags of a filtered search): 15.3 KiB per 1 024 list entries instead of 19.3, i.e. ef = 4 096 in 62 KiB
__host__ __device__ inline size_t smem_bytes(uint32_t efcap, uint32_t wpad, uint32_t ld, bool build = true) {
    size_t b = 0;
    b += (size_t)efcap * 8;
    b += (((size_t)efcap * 4 + 15) / 16) * 16;
    b += (size_t)wpad * 8;
    b += (size_t)wpad * 4 * 2;
    b += (size_t)kVlogCap * 4;
    b += (size_t)ld * 4;
    b += C_WORDS * 4;
    if (build) b += (size_t)efcap * 4 + 256 * 4;
    b += (((size_t)efcap + 15) / 16) * 16;
    return b;
}
__device__ inline Smem carve(char *base, uint32_t efcap, uint32_t wpad, uint32_t ld, bool build = true) {
    Smem s;
    s.wkey = (uint64_t *)base;
    base += (size_t)efcap * 8;
    s.wid = (uint32_t *)base;
    base += (((size_t)efcap * 4 + 15) / 16) * 16;
    s.nkey = (uint64_t *)base;
    base += (size_t)wpad * 8;
    s.nid = (uint32_t *)base;
    base += (size_t)wpad * 4;
    s.todo = (uint32_t *)base;
    base += (size_t)wpad * 4;
    s.vlog = (uint32_t *)base;
    base += (size_t)kVlogCap * 4;
    s.q = (float4 *)base;
    base += (size_t)ld * 4;
    s.ctl = (int *)base;
    base += C_WORDS * 4;
    s.tpos = (uint32_t *)base;
    if (build) base += (size_t)efcap * 4;
    s.sel = (uint32_t *)base;
    if (build) base += 256 * 4;
    s.st = (uint8_t *)base;
    return s;
}

__device__ __forceinline__ bool key_lt(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    return ka < kb || (ka == kb && ia < ib);
}

// The visited set of one query (hnsw.rs:552 `visited`).  Exact (a set of node ids, never a filter):
//   tab     open-addressing hash set, 1 << hbits slots, CZ_NONE = empty; nullptr / hbits 0 when the bitmap is no
//           bigger than a table would be (small indices)
//   bitmap  n bits; the overflow form: when the table reaches 70 % of its slots its members move here
// Between launches every table word is CZ_NONE and every bitmap word 0: the workgroup that used them restores that
// (no per-launch memset: at n = 10M the bitmaps of a 1024-query batch are 1.28 GB).  Only wave 0 of the owning
// workgroup touches either, with L2 atomics.
struct VisitedDev {
    uint32_t *tab;
    uint32_t hbits;
    uint32_t *bitmap;
    uint32_t words;
};

// F64: the index holds f64 vectors (search only; ITERS must be 0: the query is read from LDS, where it sits as doubles, and the
// distances come from distance_f64.h).  The list, the visited set and the merges do not know the element type.
template <int LPV, int ITERS, int U, bool NT = false, bool F64 = false>
struct Searcher {
    static_assert(!F64 || ITERS == 0, "the f64 evaluation reads the query from LDS");
    const IndexDev &ix;
    Smem s;
    VisitedDev vis;
    uint32_t *tcur;  // ids evaluated in this step (s.todo)
    uint32_t *spec = nullptr;  // search_level_spec's scratch: [0, 128) two prefetched link rows, [128..129] whose they are
    // search_level_pending's buffer in front of W (kPend entries: keys, ids | kExpanded, the lower bounds of a flush) and its
    // control words {entries, first possibly un-expanded}
    uint64_t *pkey = nullptr;
    uint32_t *pid = nullptr, *plb = nullptr;
    int *pctl = nullptr;
    static constexpr int kPend = kThreads;
    int tid, lane, wave, glane, group;
    int chunks;
    float4 q[ITERS > 0 ? ITERS : 1];
    float qnorm;
    double qnorm64 = 0.0;
    CZ_PH_DECL

    static constexpr int VPW = 64 / LPV;          // lane groups per wave
    static constexpr int TG = kWaves * VPW;       // lane groups per workgroup
    using Regs = RowRegs<(ITERS > 0 ? ITERS : 1), U>;

    __device__ Searcher(const IndexDev &ix_, Smem s_, VisitedDev vis_) : ix(ix_), s(s_), vis(vis_) {
        tcur = s.todo;
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        glane = lane % LPV;
        group = wave * VPW + lane / LPV;
        chunks = F64 ? (int)(ix.ld / 2) : (int)(ix.ld / 4);
    }

    // stage the query row (zero padded) into LDS and registers
    __device__ void load_query(const float *qrow) {
        if constexpr (F64) {
            const double *qrow64 = reinterpret_cast<const double *>(qrow);
            double *ql = (double *)s.q;
            for (uint32_t i = tid; i < ix.ld; i += kThreads) ql[i] = i < ix.dim ? qrow64[i] : 0.0;
            if (tid < C_WORDS) s.ctl[tid] = tid == C_VMODE ? (vis.tab ? 0 : 1) : 0;
            __syncthreads();
            qnorm = 0.f;
            qnorm64 = ix.metric == CZ_COSINE ? czd64::self_dot<LPV>((const double2 *)s.q, glane, chunks) : 0.0;
            return;
        }
        float *ql = (float *)s.q;
        for (uint32_t i = tid; i < ix.ld; i += kThreads) ql[i] = i < ix.dim ? qrow[i] : 0.f;
        if (tid < C_WORDS) s.ctl[tid] = tid == C_VMODE ? (vis.tab ? 0 : 1) : 0;
        __syncthreads();
        if constexpr (ITERS > 0) {
#pragma unroll
            for (int j = 0; j < ITERS; j++) {
                int c = glane + LPV * j;
                q[j] = c < chunks ? s.q[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        qnorm = ix.metric == CZ_COSINE ? query_norm<LPV, ITERS>(q, s.q, glane, chunks) : 0.f;
    }

    // distances from the register-resident vector (qq, qqn) to todo[0..n) -> nkey/nid   (all waves; ends with the
    // entries written but NOT yet visible: the caller's barrier publishes them).
    // Lane group g owns entries g + TG * t, t = 0, 1, ...; a "round" is U consecutive t.  The next round's rows are
    // requested before the current round is reduced, the reductions are DPP butterflies (no LDS round trips), and the
    // f64 tail of the distance (1 - x / sqrt(..)) is NOT done per row by the whole wave: the group leader parks the raw
    // f32 accumulators in the entry's key slot and, after a barrier, thread j finishes entry j -- one f64 sqrt/div per
    // thread per expansion step instead of one per row.
    // Depth of the row queue, measured (profiles/r01n): the compiler waits for BOTH rounds before it reduces the older
    // one (vmcnt(0) at the join of the `more` branch), i.e. one round per wave is outstanding at a time.  Rewriting the
    // loop as straight-line issue/consume pairs gives true double buffering (vmcnt(6..11)) and is 30 % SLOWER: a
    // step's serial chain (link row -> visited atomics -> first row) of the other workgroups on the CU queues behind
    // the deeper row stream.  Keep it shallow.
    template <int METRIC>
    __device__ __forceinline__ void issue_round(Regs &r, int base, int n, bool full) {
        const float4 *rows[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = min(base + u * TG, n - 1);  // past the end: re-read the last row, result discarded
            rows[u] = (const float4 *)(ix.vec + (size_t)tcur[j] * ix.ld);
        }
        load_rows<LPV, ITERS, U, NT>(r, rows, glane, chunks, full);
    }
    template <int METRIC>
    __device__ __forceinline__ void retire_round(const float4 (&qq)[ITERS > 0 ? ITERS : 1], const Regs &r,
                                                 int base, int n) {
        float m[U], bn[U];
        dot_rows<METRIC, LPV, ITERS, U>(qq, r, m, bn);
        if (glane == 0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = base + u * TG;
                if (j < n) ((float2 *)s.nkey)[j] = make_float2(m[u], bn[u]);
            }
        }
    }
    template <int METRIC>
    __device__ __forceinline__ void eval_rounds(const float4 (&qq)[ITERS > 0 ? ITERS : 1], int n) {
        static_assert(ITERS > 0, "register-resident rows only");
        const bool full = chunks == LPV * ITERS;
        constexpr int STEP = TG * U;
        // every group of a wave runs the same number of rounds (the wave's first group decides), so the
        // cross-lane reductions always see whole groups
        int wbase = wave * VPW;
        if (wbase >= n) return;
        int base = group;
        Regs a, b;
        issue_round<METRIC>(a, base, n, full);
        for (;;) {
            bool more = wbase + STEP < n;
            if (more) issue_round<METRIC>(b, base + STEP, n, full);
            retire_round<METRIC>(qq, a, base, n);
            if (!more) break;
            base += STEP;
            wbase += STEP;
            more = wbase + STEP < n;
            if (more) issue_round<METRIC>(a, base + STEP, n, full);
            retire_round<METRIC>(qq, b, base, n);
            if (!more) break;
            base += STEP;
            wbase += STEP;
        }
    }
    __device__ void eval_list(const float4 (&qq)[ITERS > 0 ? ITERS : 1], float qqn, int n) {
        if constexpr (F64) {
            for (int base = group; base < n; base += TG * U) {
                const double2 *rows[U];
                uint32_t ids[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int j = base + u * TG;
                    ids[u] = j < n ? tcur[j] : CZ_NONE;
                    rows[u] = j < n ? (const double2 *)(ix.vec64 + (size_t)ids[u] * ix.ld) : nullptr;
                }
                double d[U];
                czd64::group_distances<LPV, U>(ix.metric, (const double2 *)s.q, glane, chunks, qnorm64, rows, d);
                if (glane == 0) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int j = base + u * TG;
                        if (j < n) {
                            s.nkey[j] = dist_key(d[u]);
                            s.nid[j] = ids[u];
                        }
                    }
                }
            }
        } else if constexpr (ITERS > 0) {
            if (ix.metric == CZ_COSINE) eval_rounds<CZ_COSINE>(qq, n);
            else if (ix.metric == CZ_L2) eval_rounds<CZ_L2>(qq, n);
            else eval_rounds<CZ_IP>(qq, n);
            __syncthreads();
            CZ_PH_MARK(2);
            for (int j = tid; j < n; j += kThreads) {
                const float2 raw = ((const float2 *)s.nkey)[j];
                s.nkey[j] = dist_key(finish_distance(ix.metric, raw.x, raw.y, qqn));
                s.nid[j] = tcur[j];
            }
        } else {
            // generic dimension (> 2048): the query is read from LDS chunk by chunk
            for (int base = group; base < n; base += TG * U) {
                const float4 *rows[U];
                uint32_t ids[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    int j = base + u * TG;
                    ids[u] = j < n ? tcur[j] : CZ_NONE;
                    rows[u] = j < n ? (const float4 *)(ix.vec + (size_t)ids[u] * ix.ld) : nullptr;
                }
                double d[U];
                group_distances<LPV, ITERS, U>(ix.metric, qq, s.q, glane, chunks, qqn, rows, d);
                if (glane == 0) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        int j = base + u * TG;
                        if (j < n) {
                            s.nkey[j] = dist_key(d[u]);
                            s.nid[j] = ids[u];
                        }
                    }
                }
            }
        }
    }
    __device__ void eval_todo(int n) { eval_list(q, qnorm, n); }

    // hnsw_select_neighbours_heuristic (hnsw.rs:470-538) over the sorted list W (the `found` set, nearest
    // first).  The reference pops candidates nearest-first and rejects one if an already selected neighbour is
    // strictly closer to it than the centre is (:515-523).  Same result, batched: each time a candidate is
    // accepted, its distances to all still-pending farther candidates are evaluated in one pass and the ones it
    // dominates are rejected.  keep_pruned_connections back-fills from the rejected, nearest first (:530-536).
    // Returns the number selected; their list positions are in s.sel.  (extend_candidates: select_extended below.)
    __device__ int select_heuristic(int m, bool keep_pruned) {
        static_assert(ITERS > 0, "index construction needs a register-resident vector (dim <= 2048)");
        const int cnt = s.ctl[C_CNT];
        for (int i = tid; i < cnt; i += kThreads) s.st[i] = 0;
        __syncthreads();
        int nsel = 0;
        for (int i = 0; i < cnt && nsel < m; i++) {
            if (s.st[i] != 0) continue;  // uniform
            __syncthreads();
            if (tid == 0) {
                s.sel[nsel] = (uint32_t)i;
                s.st[i] = 1;
                s.ctl[C_TODO] = 0;
            }
            nsel++;
            if (nsel == m) break;
            reject_dominated(s.wid[i] & kIdMask, i, cnt);
        }
        __syncthreads();
        if (keep_pruned && nsel < m) {
            if (tid == 0) {
                int k2 = nsel;
                for (int i = 0; i < cnt && k2 < m; i++)
                    if (s.st[i] == 2) s.sel[k2++] = (uint32_t)i;
                s.ctl[C_TODO] = k2;
            }
            __syncthreads();
            nsel = s.ctl[C_TODO];
            __syncthreads();
        }
        return nsel;
    }

    // ---------------------------------------------------------------------------------------------------------
    // extend_candidates (hnsw.rs:499-511): the candidate set of the heuristic is W plus every neighbour of W's entries.
    // Up to cnt * (row width + 1) entries, which no LDS list holds: they live in a per-workgroup scratch array in global
    // memory (`gkey` / `gid`, capacity a power of two), are sorted there, and pass through the LDS list one chunk at a time.
    // ---------------------------------------------------------------------------------------------------------
    // W and the not-yet-seen neighbours of its entries, with their distances to the register-resident vector -> the
    // scratch array; returns how many.  The visited set must be empty on entry and is empty again on return (it is the
    // `one entry per key` of PriorityQueue::push here).  W itself is left alone.
    __device__ uint32_t gather_extended(int level, uint64_t *__restrict__ gkey, uint32_t *__restrict__ gid, uint32_t wcap,
                                        uint32_t gcap) {
        static_assert(ITERS > 0, "index construction needs a register-resident vector (dim <= 2048)");
        const int cnt = s.ctl[C_CNT];
        for (int i = tid; i < cnt; i += kThreads) {
            gkey[i] = s.wkey[i];
            gid[i] = s.wid[i] & kIdMask;
        }
        if (wave == 0) {
            for (int b = 0; b < cnt; b += 64) {
                const int j = b + lane;
                uint32_t where;
                const bool fresh = visit(j < cnt ? (s.wid[j] & kIdMask) : CZ_NONE, j < cnt, where);
                if (fresh) log_visit(where, true);
            }
        }
        __syncthreads();
        const int width = level == 0 ? ix.w0 : ix.wu;
        uint32_t total = (uint32_t)cnt;
        int i = 0;
        while (i < cnt) {  // uniform
            if (wave == 0) {  // the link rows of as many entries as the evaluation list takes
                int n = 0, j = i;
                while (j < cnt && n + width <= (int)wcap) {
                    n += expand_row(s.wid[j] & kIdMask, level, width, tcur + n, true);
                    j++;
                }
                if (lane == 0) {
                    s.ctl[C_TODO] = n;
                    s.ctl[C_KEEP] = j;
                    count_dist(n);
                }
            }
            __syncthreads();
            const int n = s.ctl[C_TODO];
            i = s.ctl[C_KEEP];
            if (n > 0) {
                CZ_CHECK(total + (uint32_t)n <= gcap, "gather: %u + %d candidates, room for %u (level %d, cnt %d)\n", total, n, gcap, level, cnt);
                for (int j = tid; j < n; j += kThreads)
                    CZ_CHECK(tcur[j] < ix.n, "gather: candidate id %u of %u nodes (level %d, slot %d of %d)\n", tcur[j], ix.n, level, j, n);
                eval_todo(n);
                __syncthreads();
                for (int j = tid; j < n; j += kThreads) {
                    gkey[total + j] = s.nkey[j];
                    gid[total + j] = s.nid[j];
                }
                total += (uint32_t)n;
            }
            __syncthreads();
        }
        clear_visited();
        return total;
    }

    // ascending by (key, id): a bitonic network over the scratch array (entries past n are padded with the greatest pair)
    __device__ void sort_scratch(uint64_t *__restrict__ gkey, uint32_t *__restrict__ gid, uint32_t n) {
        uint32_t P = 2;
        while (P < n) P <<= 1;
        for (uint32_t i = n + tid; i < P; i += kThreads) {
            gkey[i] = ~0ull;
            gid[i] = CZ_NONE;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < P / 2; t += kThreads) {
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo + j;
                    const uint64_t ka = gkey[lo], kb = gkey[hi];
                    const uint32_t ia = gid[lo], ib = gid[hi];
                    const bool up = (lo & k) == 0;
                    if (key_lt(kb, ib, ka, ia) == up) {
                        gkey[lo] = kb;
                        gid[lo] = ib;
                        gkey[hi] = ka;
                        gid[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
    }

    // one accepted neighbour `a` against the pending candidates at list positions > from: the ones it is strictly closer to
    // than the centre is are rejected (:515-523).  All threads; begins and ends with a barrier.
    __device__ void reject_dominated(uint32_t a, int from, int cnt) {
        __syncthreads();
        if (wave == 0) {
            int total = 0;
            for (int b = from + 1; b < cnt; b += 64) {
                const int j = b + lane;
                const bool pend = j < cnt && s.st[j] == 0;
                const unsigned long long mk = __ballot(pend);
                if (pend) {
                    const int p = total + __popcll(mk & ((1ull << lane) - 1ull));
                    tcur[p] = s.wid[j] & kIdMask;
                    s.tpos[p] = (uint32_t)j;
                }
                total += __popcll(mk);
            }
            if (lane == 0) {
                s.ctl[C_TODO] = total;
                count_dist(total);
            }
        }
        __syncthreads();
        const int n = s.ctl[C_TODO];
        if (n == 0) return;
        const float4 *srow = (const float4 *)(ix.vec + (size_t)a * ix.ld);
        float4 q2[ITERS > 0 ? ITERS : 1];
#pragma unroll
        for (int j = 0; j < ITERS; j++) {
            const int c = glane + LPV * j;
            q2[j] = c < chunks ? srow[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float q2n = ix.metric == CZ_COSINE ? query_norm<LPV, ITERS>(q2, s.q, glane, chunks) : 0.f;
        eval_list(q2, q2n, n);
        __syncthreads();
        for (int j = tid; j < n; j += kThreads) {
            const uint32_t p = s.tpos[j];
            const uint64_t to_center = s.wkey[p], to_sel = s.nkey[j];
            if (to_sel < to_center && to_center != ~0ull) s.st[p] = 2;  // raw `<`: false when either is NaN
        }
        __syncthreads();
    }

    // the selected neighbours of select_extended: ids in s.sel, distance keys here (the log of the visited set is idle
    // while the heuristic runs: kVlogCap words hold 2 x 256 keys and 256 ids)
    __device__ __forceinline__ uint64_t *ext_sel_key() const { return (uint64_t *)s.vlog; }

    // hnsw_select_neighbours_heuristic over the sorted scratch array, `chunk` (<= the LDS list's capacity) entries at a
    // time: a chunk's entries first meet the neighbours accepted from earlier chunks, then each other, exactly the
    // comparisons the reference makes when it pops them one by one.  Overwrites W.  Returns the number selected
    // (s.sel[k], ext_sel_key()[k], in the order of acceptance; back-filled ones after them).
    __device__ int select_extended(const uint64_t *__restrict__ gkey, const uint32_t *__restrict__ gid, uint32_t n_cand, int m,
                                   bool keep_pruned, uint32_t chunk) {
        uint64_t *acc_key = ext_sel_key();
        uint64_t *disc_key = acc_key + 256;
        uint32_t *disc_id = (uint32_t *)(disc_key + 256);
        int nsel = 0, ndisc = 0;
        for (uint32_t base = 0; base < n_cand && nsel < m; base += chunk) {  // uniform
            const int cnt = (int)min(chunk, n_cand - base);
            __syncthreads();
            for (int i = tid; i < cnt; i += kThreads) {
                s.wkey[i] = gkey[base + i];
                s.wid[i] = gid[base + i];
                s.st[i] = 0;
            }
            for (int a = 0; a < nsel; a++) reject_dominated(s.sel[a], -1, cnt);
            __syncthreads();
            for (int i = 0; i < cnt && nsel < m; i++) {
                if (s.st[i] != 0) continue;  // uniform
                __syncthreads();
                const uint32_t a = s.wid[i];
                if (tid == 0) {
                    s.sel[nsel] = a;
                    acc_key[nsel] = s.wkey[i];
                    s.st[i] = 1;
                }
                nsel++;
                if (nsel == m) break;
                reject_dominated(a, i, cnt);
            }
            __syncthreads();
            if (keep_pruned && nsel < m) {  // the whole chunk has been popped: its rejected entries queue up, nearest first
                if (tid == 0) {
                    int k = ndisc;
                    for (int i = 0; i < cnt && k < m; i++)
                        if (s.st[i] == 2) {
                            disc_id[k] = s.wid[i];
                            disc_key[k] = s.wkey[i];
                            k++;
                        }
                    s.ctl[C_TODO] = k;
                }
                __syncthreads();
                ndisc = s.ctl[C_TODO];
            }
        }
        __syncthreads();
        if (keep_pruned && nsel < m && ndisc > 0) {  // :530-536
            const int take = min(ndisc, m - nsel);
            if (tid < take) {
                s.sel[nsel + tid] = disc_id[tid];
                acc_key[nsel + tid] = disc_key[tid];
            }
            nsel += take;
            __syncthreads();
        }
        return nsel;
    }

    // merge nkey/nid[0..n) into W (capacity ef).  Caller guarantees a barrier before; ends with a barrier.
    // The eligible new entries (typically a handful of the ~50 evaluated per step once W is full) are compacted
    // first; then every (W entry, new entry) pair is compared exactly ONCE, in registers: the new entries are
    // broadcast one by one with v_readlane, a wave's 64 W entries answer with one ballot (= how many W entries
    // precede the new one) while each lane counts the new entries that precede its own W entry (= how far it
    // shifts).  No LDS traffic inside the loop (the previous form walked the whole batch through LDS per thread and
    // spent 6 us of a 31 us step there).
    __device__ void merge(int n, int ef) {
        if (ef > kMergeR * kThreads || n > kMergeEC * 64) {  // uniform: beyond what the register form holds: the ranked form
            merge_wide(n, ef);
            return;
        }
        const int cnt = s.ctl[C_CNT];
        const bool full = cnt >= ef;
        const uint64_t bkey = cnt > 0 ? s.wkey[cnt - 1] : 0;
        // eligibility (hnsw.rs:575 `found_nn.len() < ef || neighbour_dist < furthest`, raw f64 compare)
        // NaN corner (zero vector under Cosine): the reference inserts one neighbour at a time; once the list is
        // full with a NaN as its maximum, `x < NaN` is false for ever and nothing else gets in.  Reproduce it: if
        // this batch overflows a not-yet-full list and a NaN is among what fills it, only the first free slots
        // (in neighbour order) are taken.
        const int free_slots = ef - cnt;
        bool nan_gate = false;
        if (!full && cnt + n > ef) {  // uniform
            const bool nan_here = (tid < free_slots && tid < n && s.nkey[tid] == ~0ull) || (tid == 0 && cnt > 0 && bkey == ~0ull);
            nan_gate = __syncthreads_or(nan_here);
        }
        uint64_t mykey = 0;
        uint32_t myid = CZ_NONE;
        bool elig = false;
        if (tid < n) {
            mykey = s.nkey[tid];
            myid = s.nid[tid];
            if (full) elig = mykey < bkey && bkey != ~0ull;
            else elig = nan_gate ? tid < free_slots : true;
        }
        // compaction (batch order kept): per-wave ballot + the wave totals through LDS
        const unsigned long long em = __ballot(elig);
        if (lane == 0) s.ctl[C_WAVE0 + wave] = __popcll(em);
        __syncthreads();  // also orders the nkey/nid reads above before the in-place rewrite below
        int before = 0, nelig = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const int c = s.ctl[C_WAVE0 + w];
            if (w < wave) before += c;
            nelig += c;
        }
        if (nelig == 0) return;
        if (elig) {
            const int e = before + __popcll(em & ((1ull << lane) - 1ull));
            s.nkey[e] = mykey;
            s.nid[e] = myid;
            tcur[e] = 0;  // rank accumulator of compacted entry e (the step's ids were copied to nid)
        }
        __syncthreads();
        // registers: this thread's W entries (W index tid + r * kThreads: a wave holds 64 consecutive ones) and the
        // compacted entries (lane l of EVERY wave holds entries l, 64 + l, ...)
        constexpr int R = kMergeR;    // ef <= R * kThreads
        constexpr int EC = kMergeEC;  // n <= EC * 64
        uint64_t wk[R];
        uint32_t wi[R];
        int sft[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = tid + r * kThreads;
            wk[r] = ~0ull;
            wi[r] = CZ_NONE;
            sft[r] = 0;
            if (j < cnt) {
                wk[r] = s.wkey[j];
                wi[r] = s.wid[j];
            }
        }
        uint64_t ek[EC];
        uint32_t ei[EC];
        int acc[EC];
#pragma unroll
        for (int c = 0; c < EC; c++) {
            const int e = c * 64 + lane;
            ek[c] = ~0ull;
            ei[c] = CZ_NONE;
            acc[c] = 0;
            if (e < nelig) {
                ek[c] = s.nkey[e];
                ei[c] = s.nid[e];
            }
        }
#pragma unroll
        for (int c = 0; c < EC; c++) {
            if (c * 64 >= nelig) break;  // uniform
            const int tn = min(64, nelig - c * 64);
            for (int t = 0; t < tn; t++) {
                const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)ek[c], t);
                const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(ek[c] >> 32), t);
                const uint32_t it = __builtin_amdgcn_readlane(ei[c], t);
                const uint64_t kt = ((uint64_t)khi << 32) | klo;
                int before_t = 0;  // uniform: entries of W (this wave's chunks) / of the batch that precede entry t
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (r * kThreads + wave * 64 >= cnt) break;  // uniform
                    const bool valid = tid + r * kThreads < cnt;
                    const bool lt = valid && key_lt(wk[r], wi[r] & kIdMask, kt, it);
                    before_t += __popcll(__ballot(lt));
                    sft[r] += (valid && !lt) ? 1 : 0;
                }
                if ((t & (kWaves - 1)) == wave) {  // one wave ranks entry t inside the batch
#pragma unroll
                    for (int c2 = 0; c2 < EC; c2++) {
                        if (c2 * 64 >= nelig) break;  // uniform
                        const bool lt2 = c2 * 64 + lane < nelig && key_lt(ek[c2], ei[c2], kt, it);
                        before_t += __popcll(__ballot(lt2));
                    }
                }
                if (lane == t) acc[c] += before_t;
            }
        }
#pragma unroll
        for (int c = 0; c < EC; c++)
            if (c * 64 + lane < nelig && acc[c] != 0) atomicAdd(&tcur[c * 64 + lane], (uint32_t)acc[c]);
        __syncthreads();
        // scatter in place: every W entry and every compacted entry is in a register by now
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = tid + r * kThreads;
            if (j < cnt && sft[r] > 0 && j + sft[r] < ef) {
                s.wkey[j + sft[r]] = wk[r];
                s.wid[j + sft[r]] = wi[r];
            }
        }
        if (tid < nelig) {  // thread e owns compacted entry e: chunk `wave`, lane `lane`
            uint64_t k0 = ek[0];
            uint32_t i0 = ei[0];
#pragma unroll
            for (int c = 1; c < EC; c++)
                if (wave == c) {
                    k0 = ek[c];
                    i0 = ei[c];
                }
            const int npos = (int)tcur[tid];
            if (npos < ef) {
                s.wkey[npos] = k0;
                s.wid[npos] = i0;  // un-expanded
                atomicMin(&s.ctl[C_LO], npos);
            }
        }
        if (tid == 0) s.ctl[C_CNT] = min(ef, cnt + nelig);
        __syncthreads();
    }

    // the general form of the merge (any ef the LDS list holds, any n <= 256).  Round 4: the batch is RANKED instead of walked.
    // Until then every thread compared each of its W entries with every batch entry through LDS (R x n reads per thread and
    // step: the "slow ef > 512 step", 0.58-0.61 of the HBM peak at ef = 768) and W sat in registers (R = 4: ef <= 1024 was the
    // limit).  Now: (a) the eligible entries are compacted; (b) thread e ranks entry e inside the batch (<= n reads) and finds
    // its lower bound in W by bisection (log2 cnt reads): final position = lower bound + rank; the lower bounds, stored by rank,
    // are ascending; (c) a W entry at j moves up by the number of lower bounds <= j -- a bisection over <= 256 sorted words --
    // and since every move goes UP, W is shifted in place block by block from the top, four entries per thread and one barrier
    // per block, and only the blocks at or above the first insertion point are touched; (d) the new entries drop into
    // the holes.  No per-thread copy of W, so ef is bounded by LDS alone (4 096 at 64 KiB, the API's limit is the 160 KiB).
    // Same result as the one-at-a-time push / pop of the reference (hnsw.rs:572-583) like the register form above.
    __device__ void merge_wide(int n, int ef) {
        const int cnt = s.ctl[C_CNT];
        const bool full = cnt >= ef;
        const uint64_t bkey = cnt > 0 ? s.wkey[cnt - 1] : 0;
        // eligibility (hnsw.rs:575 `found_nn.len() < ef || neighbour_dist < furthest`, raw f64 compare)
        // NaN corner (zero vector under Cosine): the reference inserts one neighbour at a time; once the list is
        // full with a NaN as its maximum, `x < NaN` is false for ever and nothing else gets in.  Reproduce it: if
        // this batch overflows a not-yet-full list and a NaN is among what fills it, only the first free slots
        // (in neighbour order) are taken.
        const int free_slots = ef - cnt;
        bool nan_gate = false;
        if (!full && cnt + n > ef) {  // uniform
            const bool nan_here = (tid < free_slots && tid < n && s.nkey[tid] == ~0ull) || (tid == 0 && cnt > 0 && bkey == ~0ull);
            nan_gate = __syncthreads_or(nan_here);
        }
        uint64_t mykey = 0;
        uint32_t myid = CZ_NONE;
        bool elig = false;
        if (tid < n) {
            mykey = s.nkey[tid];
            myid = s.nid[tid];
            if (full) elig = mykey < bkey && bkey != ~0ull;
            else elig = nan_gate ? tid < free_slots : true;
        }
        // (a) compaction, batch order kept: per-wave ballot + the wave totals through LDS
        const unsigned long long em = __ballot(elig);
        if (lane == 0) s.ctl[C_WAVE0 + wave] = __popcll(em);
        __syncthreads();  // also orders the nkey/nid reads above before the in-place rewrite below
        int before = 0, nelig = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const int c = s.ctl[C_WAVE0 + w];
            if (w < wave) before += c;
            nelig += c;
        }
        if (nelig == 0) return;
        if (elig) {
            const int e = before + __popcll(em & ((1ull << lane) - 1ull));
            s.nkey[e] = mykey;
            s.nid[e] = myid;
        }
        __syncthreads();
        CZ_PH_MARK(8);  // (profiling builds: the merge's own phases are 8 = compaction, 9 = rank, 10 = shift; 4 = the rest)
        // (b) thread e < nelig: rank inside the batch, lower bound in W
        int npos = -1;
        if (tid < nelig) {
            mykey = s.nkey[tid];
            myid = s.nid[tid];
            int r1 = 0;
            for (int t = 0; t < nelig; t++)
                if (key_lt(s.nkey[t], s.nid[t], mykey, myid)) r1++;
            int lo = 0, hi = cnt;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (key_lt(s.wkey[mid], s.wid[mid] & kIdMask, mykey, myid)) lo = mid + 1;
                else hi = mid;
            }
            tcur[r1] = (uint32_t)lo;  // lower bounds by rank: ascending (the step's ids were copied to nid)
            npos = lo + r1;
        }
        __syncthreads();
        CZ_PH_MARK(9);
        // (c) shift W in place from the top, kShiftR chunks (kShiftR entries per thread) per barrier.  A W entry at j moves up by
        // the number of lower bounds <= j: all of them above the last one (no search), a count over registers when the step
        // brought at most kLb entries (nearly always), a bisection otherwise.  One barrier per block: a block's entries are read
        // before the barrier and written after it, into their own block and the start of the one above -- which the previous
        // round read before ITS barrier -- while the next round's reads (the block below) touch nothing this round writes.
        // (A list of 8 192 is shifted nearly whole on most steps: with one chunk and two barriers per round the merge was 46 % of
        // a step, now 25 % is the shift and 9 % the rank, profiles/r04_large_ef_phases.txt; moving 8 192 x 12 bytes in and out of
        // LDS is 1 500 cycles at 128 B / clock, the shift takes 3 200: what is left is the move itself.)
        const int first = (int)tcur[0], last = (int)tcur[nelig - 1];
        constexpr int kLb = 8, kShiftR = 4, kBlock = kShiftR * kThreads;
        int lb[kLb];
#pragma unroll
        for (int i = 0; i < kLb; i++) lb[i] = i < nelig ? (int)tcur[i] : 0x7fffffff;
        int base = ((cnt - 1) / kBlock) * kBlock;
        bool more = base >= 0 && base + kBlock > first;  // uniform
        uint64_t wk[kShiftR];
        uint32_t wi[kShiftR];
        int sft[kShiftR];
        auto fetch = [&]() {
#pragma unroll
            for (int r = 0; r < kShiftR; r++) {
                const int j = base + r * kThreads + tid;
                sft[r] = 0;
                if (j < cnt && j >= first) {
                    wk[r] = s.wkey[j];
                    wi[r] = s.wid[j];
                    if (j >= last) sft[r] = nelig;
                    else if (nelig <= kLb) {
#pragma unroll
                        for (int i = 0; i < kLb; i++) sft[r] += lb[i] <= j ? 1 : 0;
                    } else {
                        int lo = 0, hi = nelig;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if ((int)tcur[mid] <= j) lo = mid + 1;
                            else hi = mid;
                        }
                        sft[r] = lo;
                    }
                }
            }
        };
        if (more) fetch();
        while (more) {  // uniform
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kShiftR; r++) {
                const int j = base + r * kThreads + tid;
                if (sft[r] > 0 && j + sft[r] < ef) {
                    s.wkey[j + sft[r]] = wk[r];
                    s.wid[j + sft[r]] = wi[r];
                }
            }
            base -= kBlock;
            more = base >= 0 && base + kBlock > first;
            if (more) fetch();
        }
        CZ_PH_MARK(10);
        // (d) the new entries into the holes
        if (npos >= 0 && npos < ef) {
            s.wkey[npos] = mykey;
            s.wid[npos] = myid;  // un-expanded
            atomicMin(&s.ctl[C_LO], npos);
        }
        if (tid == 0) s.ctl[C_CNT] = min(ef, cnt + nelig);
        __syncthreads();
    }

    // ---------------------------------------------------------------------------------------------------------
    // visited set
    // ---------------------------------------------------------------------------------------------------------
    // restore "every table word CZ_NONE, every bitmap word 0" wholesale (all threads)
    __device__ void clear_all() {
        if (vis.tab) {
            const uint32_t quads = (1u << vis.hbits) / 4;
            uint4 *t4 = (uint4 *)vis.tab;
            for (uint32_t i = tid; i < quads; i += kThreads) t4[i] = make_uint4(CZ_NONE, CZ_NONE, CZ_NONE, CZ_NONE);
        }
        if (s.ctl[C_VMODE] == 1)
            for (uint32_t i = tid; i < vis.words; i += kThreads) vis.bitmap[i] = 0;
        __syncthreads();
        if (tid == 0) {
            s.ctl[C_VLOG] = 0;
            s.ctl[C_VCNT] = 0;
            s.ctl[C_VMODE] = vis.tab ? 0 : 1;
        }
        __syncthreads();
    }
    // forget the visited set of a finished level: by the log of touched slots / ids when it is short and whole
    __device__ void clear_visited() {
        const int nlog = s.ctl[C_VLOG];
        if (nlog > kVlogCap) {  // uniform
            clear_all();
            return;
        }
        if (s.ctl[C_VMODE] == 0) {
            for (int i = tid; i < nlog; i += kThreads) vis.tab[s.vlog[i]] = CZ_NONE;
        } else {
            for (int i = tid; i < nlog; i += kThreads) vis.bitmap[s.vlog[i] >> 5] = 0;
        }
        __syncthreads();
        if (tid == 0) {
            s.ctl[C_VLOG] = 0;
            s.ctl[C_VCNT] = 0;
        }
        __syncthreads();
    }
    // slot (hash form) or id (bitmap form) of a new member
    __device__ __forceinline__ void log_visit(uint32_t what, bool enable) {
        if (!enable) return;
        int p = atomicAdd(&s.ctl[C_VLOG], 1);
        if (p < kVlogCap) s.vlog[p] = what;
    }
    // wave 0: the table is about to pass its load limit: its members move into the query's bitmap row
    __device__ void to_bitmap() {
        const uint32_t slots = 1u << vis.hbits;
        for (uint32_t i = lane; i < slots; i += 64) {
            const uint32_t v = __hip_atomic_load(&vis.tab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != CZ_NONE) atomicOr(&vis.bitmap[v >> 5], 1u << (v & 31));
        }
        if (lane == 0) {
            s.ctl[C_VMODE] = 1;
            s.ctl[C_VLOG] = kVlogCap + 1;  // the log mixes slots and ids from here on: clear wholesale
        }
    }
    // wave 0: test-and-set of up to 64 DISTINCT ids, one per active lane; true where the id was new.
    // `where` = the slot (hash form) or the id (bitmap form), for the log.
    __device__ __forceinline__ bool visit(uint32_t id, bool active, uint32_t &where) {
        int mode = s.ctl[C_VMODE];  // uniform
        if (mode == 0 && (uint32_t)s.ctl[C_VCNT] + 64u > (7u << vis.hbits) / 10u) {
            to_bitmap();
            mode = 1;
        }
        bool fresh = false;
        where = id;
        if (active) {
            if (mode == 0) {
                const uint32_t hmask = (1u << vis.hbits) - 1u;
                uint32_t h = (id * 0x9E3779B1u) >> (32 - vis.hbits);
                for (;;) {
                    const uint32_t old = atomicCAS(&vis.tab[h], CZ_NONE, id);
                    if (old == CZ_NONE) {
                        fresh = true;
                        break;
                    }
                    if (old == id) break;
                    h = (h + 1) & hmask;
                }
                where = h;
            } else {
                const uint32_t bit = 1u << (id & 31);
                fresh = !(atomicOr(&vis.bitmap[id >> 5], bit) & bit);
            }
        }
        const int c = __popcll(__ballot(fresh));
        if (lane == 0 && c) s.ctl[C_VCNT] += c;
        return fresh;
    }

    // ---------------------------------------------------------------------------------------------------------
    // one expansion: link row of `cand` -> its not-yet-visited neighbours, in row order (hnsw.rs:566-571).  Wave 0.
    // ---------------------------------------------------------------------------------------------------------
    __device__ int expand_row(uint32_t cand, int level, int width, uint32_t *dst, bool log) {
        const uint32_t *row = level == 0 ? ix.nbr0 + (size_t)cand * ix.w0
                                         : ix.up_nbrs + ((size_t)ix.up_base[cand] + (level - 1)) * ix.wu;
        int total = 0;
        for (int c0 = 0; c0 < width; c0 += 64) {
            const int c = c0 + lane;
            const uint32_t nb = c < width ? row[c] : CZ_NONE;
            CZ_CHECK(nb == CZ_NONE || nb < ix.n, "expand_row: node %u level %d slot %d holds %u (n = %u)\n", cand, level, c, nb, ix.n);
            uint32_t where;
            const bool fresh = visit(nb, nb != CZ_NONE, where);
            const unsigned long long m = __ballot(fresh);
            if (fresh) {
                const int p = total + __popcll(m & ((1ull << lane) - 1ull));
                dst[p] = nb;
                log_visit(where, log);
            }
            total += __popcll(m);
        }
        return total;
    }
    __device__ __forceinline__ void count_dist(int n) {  // lane 0 of wave 0: 64-bit distance-evaluation counter
        unsigned int lo = (unsigned int)s.ctl[C_NDIST_LO];
        unsigned int nl = lo + (unsigned int)n;
        s.ctl[C_NDIST_LO] = (int)nl;
        if (nl < lo) s.ctl[C_NDIST_HI] += 1;
    }
    // hnsw_search_level (hnsw.rs:539-587) with W carried in and out.  log = keep a log of the visited set's new
    // members so that it can be emptied cheaply afterwards (every level but the last one of a search)
    __device__ void search_level(int level, int ef, bool log) {
        // :554-557 every carried entry is visited and a candidate again
        int cnt = s.ctl[C_CNT];
        for (int i = tid; i < cnt; i += kThreads) s.wid[i] &= kIdMask;
        if (wave == 0) {
            for (int b = 0; b < cnt; b += 64) {
                const int j = b + lane;
                uint32_t where;
                const bool fresh = visit(j < cnt ? (s.wid[j] & kIdMask) : CZ_NONE, j < cnt, where);
                if (fresh) log_visit(where, log);
            }
        }
        if (tid == 0) s.ctl[C_LO] = 0;
        __syncthreads();
        CZ_PH_START();
        const int width = level == 0 ? ix.w0 : ix.wu;
        for (;;) {
            // nearest un-expanded entry (uniform across the workgroup)
            cnt = s.ctl[C_CNT];
            int idx = -1;
            for (int b = s.ctl[C_LO]; b < cnt; b += 64) {
                int j = b + lane;
                bool un = j < cnt && !(s.wid[j] & kExpanded);
                unsigned long long m = __ballot(un);
                if (m) {
                    idx = b + __ffsll((long long)m) - 1;
                    break;
                }
            }
            if (idx < 0) break;
            const uint32_t cand = s.wid[idx] & kIdMask;
            CZ_PH_COUNT(5, 1);
            __syncthreads();  // everyone has read wid[idx] / C_LO before they change
            if (tid == 0) {
                s.wid[idx] = cand | kExpanded;
                s.ctl[C_LO] = idx + 1;
            }
            CZ_PH_MARK(0);
            // neighbour row + visited filter (wave 0), hnsw.rs:566-571
            if (wave == 0) {
                const int total = expand_row(cand, level, width, tcur, log);
                if (lane == 0) {
                    s.ctl[C_TODO] = total;
                    count_dist(total);
                }
            }
            __syncthreads();
            CZ_PH_MARK(1);
            const int n = s.ctl[C_TODO];
            if (n == 0) continue;
            CZ_PH_COUNT(6, n);
            eval_todo(n);
            __syncthreads();
            CZ_PH_MARK(3);
            merge(n, ef);
            CZ_PH_MARK(4);
        }
    }

    // ---- large ef: a sorted PENDING buffer in front of W ------------------------------------------------------------------------
    // With W at 8 192 entries nearly every step's handful of new entries lands somewhere in the middle and merge_wide moves most of
    // the list up by a few places: 25 % of a step for the shift, 9 % for the ranks (profiles/r04_large_ef_phases.txt).  Here the
    // new entries of a step go into a buffer P of at most kPend entries, itself sorted; W is touched only when P is full (one
    // shift per ~20-40 steps) and when the level ends.  What the reference calls found_nn / candidates is then the best ef of
    // W u P: with t = how many of P's entries are among them, these are W[0, ef - t) and P[0, t); their largest key is the
    // `furthest` the eligibility test compares with (hnsw.rs:575), the nearest un-expanded entry among them is the next candidate,
    // and the loop ends when there is none (an entry outside the best ef is farther than `furthest`: hnsw.rs:562-564 breaks on it).
    // Entries outside stay where they are until the next flush drops them; nothing reads them.  P is used only once W is full
    // (while it fills, the step's entries go to W directly: merge()), so W u P never has fewer than ef entries.
    // Same ids / distances / n_dist as search_level (tests/test_gpu_hnsw.py compares the two forms at ef 2 049 .. 8 192).

    // t: the largest t in [0, p] with (t == 0 or P[t - 1] < W[ef - t]); every thread computes it (uniform)
    __device__ __forceinline__ int pending_inside(int p, int ef) const {
        int lo = 0, hi = p;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (key_lt(pkey[mid - 1], pid[mid - 1] & kIdMask, s.wkey[ef - mid], s.wid[ef - mid] & kIdMask)) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    }

    // the same from a guess (t moves by a few entries from step to step: two or three probes instead of eight bisection rounds)
    __device__ __forceinline__ int pending_inside_near(int p, int ef, int guess) const {
        int t = min(max(guess, 0), p);
        auto f = [&](int x) {  // x >= 1
            return key_lt(pkey[x - 1], pid[x - 1] & kIdMask, s.wkey[ef - x], s.wid[ef - x] & kIdMask);
        };
        while (t < p && f(t + 1)) t++;
        while (t > 0 && !f(t)) t--;
        return t;
    }

    // P[0, t) into W (full: cnt == ef), the rest of P dropped.  Caller guarantees a barrier before; ends with a barrier.
    // P is sorted, so entry i's final position is its lower bound in W + i and the lower bounds ascend: merge_wide's shift
    // (steps (c), (d)) with the ranks for free.
    __device__ __forceinline__ void flush_pending(int ef) {  // (inlined at both call sites: a call would put the whole Searcher -- the query registers -- in memory)
        const int p = pctl[0];
        const int t = p > 0 ? pending_inside(p, ef) : 0;
        const int cnt = ef;
        uint64_t mykey = 0;
        uint32_t myid = CZ_NONE;
        int npos = -1;
        if (tid < t) {
            mykey = pkey[tid];
            myid = pid[tid];
            int lo = 0, hi = cnt;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (key_lt(s.wkey[mid], s.wid[mid] & kIdMask, mykey, myid & kIdMask)) lo = mid + 1;
                else hi = mid;
            }
            plb[tid] = (uint32_t)lo;
            npos = lo + tid;
        }
        __syncthreads();
        if (t > 0) {  // uniform
            const int first = (int)plb[0], last = (int)plb[t - 1];
            constexpr int kShiftR = 4, kBlock = kShiftR * kThreads;
            int base = ((cnt - 1) / kBlock) * kBlock;
            bool more = base >= 0 && base + kBlock > first;
            uint64_t wk[kShiftR];
            uint32_t wi[kShiftR];
            int sft[kShiftR];
            auto fetch = [&]() {
#pragma unroll
                for (int r = 0; r < kShiftR; r++) {
                    const int j = base + r * kThreads + tid;
                    sft[r] = 0;
                    if (j < cnt && j >= first) {
                        wk[r] = s.wkey[j];
                        wi[r] = s.wid[j];
                        if (j >= last) sft[r] = t;
                        else {  // how many lower bounds are <= j
                            int lo = 0, hi = t;
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if ((int)plb[mid] <= j) lo = mid + 1;
                                else hi = mid;
                            }
                            sft[r] = lo;
                        }
                    }
                }
            };
            if (more) fetch();
            while (more) {  // one barrier per block, from the top: merge_wide (c)
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kShiftR; r++) {
                    const int j = base + r * kThreads + tid;
                    if (sft[r] > 0 && j + sft[r] < ef) {
                        s.wkey[j + sft[r]] = wk[r];
                        s.wid[j + sft[r]] = wi[r];
                    }
                }
                base -= kBlock;
                more = base >= 0 && base + kBlock > first;
                if (more) fetch();
            }
            if (npos >= 0 && npos < ef) {
                s.wkey[npos] = mykey;
                s.wid[npos] = myid;  // (carries kExpanded if the entry was expanded while it waited)
                if (!(myid & kExpanded)) atomicMin(&s.ctl[C_LO], npos);
            }
        }
        if (tid == 0) {
            pctl[0] = 0;
            pctl[1] = 0;
        }
        __syncthreads();
    }

    // the step's nkey / nid[0, n) into P (W is full).  Caller guarantees a barrier before; ends with a barrier.
    // t: how many of P's entries are inside (the step's value); returns the guess for the next step's
    __device__ __forceinline__ int pending_merge(int n, int ef, int t) {
        int p = pctl[0];
        // `furthest` of the best ef of W u P (hnsw.rs:575 `neighbour_dist < furthest`, raw compare; a NaN furthest admits nothing)
        uint64_t fkey = ef - t > 0 ? s.wkey[ef - t - 1] : 0;
        if (t > 0 && pkey[t - 1] > fkey) fkey = pkey[t - 1];
        uint64_t mykey = 0;
        uint32_t myid = CZ_NONE;
        bool elig = false;
        if (tid < n) {
            mykey = s.nkey[tid];
            myid = s.nid[tid];
            elig = mykey < fkey && fkey != ~0ull;
        }
        const unsigned long long em = __ballot(elig);
        if (lane == 0) s.ctl[C_WAVE0 + wave] = __popcll(em);
        __syncthreads();  // also orders the nkey / nid reads above before the in-place rewrite below
        int before = 0, nelig = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const int c = s.ctl[C_WAVE0 + w];
            if (w < wave) before += c;
            nelig += c;
        }
        if (nelig == 0) return t;
        if (elig) {
            const int e = before + __popcll(em & ((1ull << lane) - 1ull));
            s.nkey[e] = mykey;
            s.nid[e] = myid;
        }
        __syncthreads();
        CZ_PH_MARK(8);
        if (p + nelig > kPend) {  // uniform: no room -- P goes into W first (the best ef of W u P do not change by that)
            flush_pending(ef);
            p = 0;
            t = 0;
            CZ_PH_MARK(10);
        }
        // entry e of the batch: rank inside the batch + lower bound in P; entry j of P moves up by the batch entries before it
        int npos = -1;
        if (tid < nelig) {
            mykey = s.nkey[tid];
            myid = s.nid[tid];
            int r1 = 0;
            for (int x = 0; x < nelig; x++)
                if (key_lt(s.nkey[x], s.nid[x], mykey, myid)) r1++;
            int lo = 0, hi = p;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (key_lt(pkey[mid], pid[mid] & kIdMask, mykey, myid)) lo = mid + 1;
                else hi = mid;
            }
            npos = lo + r1;
        }
        uint64_t pk = 0;
        uint32_t pi = 0;
        int up = 0;
        if (tid < p) {
            pk = pkey[tid];
            pi = pid[tid];
            for (int x = 0; x < nelig; x++)
                if (key_lt(s.nkey[x], s.nid[x], pk, pi & kIdMask)) up++;
        }
        if (tid == 0) pctl[1] = p + nelig;  // the first un-expanded entry of the new P: found below (everybody has read the old value)
        __syncthreads();
        if (tid < p && up > 0) {
            pkey[tid + up] = pk;
            pid[tid + up] = pi;
        }
        if (npos >= 0) {
            pkey[npos] = mykey;
            pid[npos] = myid;  // un-expanded
            atomicMin(&pctl[1], npos);
        }
        {   // P's old entries keep their order: a wave's first un-expanded lane holds its smallest new index
            const unsigned long long um = __ballot(tid < p && !(pi & kExpanded));
            if (um && lane == __ffsll((long long)um) - 1) atomicMin(&pctl[1], tid + up);
        }
        if (tid == 0) pctl[0] = p + nelig;
        __syncthreads();
        CZ_PH_MARK(9);
        return t + nelig;
    }

    // search_level with the pending buffer (see above); any level, any ef: a list of at most 2 kPend entries takes merge() as before
    __device__ __forceinline__ void search_level_pending(int level, int ef, bool log) {
        int cnt = s.ctl[C_CNT];
        for (int i = tid; i < cnt; i += kThreads) s.wid[i] &= kIdMask;
        if (wave == 0) {
            for (int b = 0; b < cnt; b += 64) {
                const int j = b + lane;
                uint32_t where;
                const bool fresh = visit(j < cnt ? (s.wid[j] & kIdMask) : CZ_NONE, j < cnt, where);
                if (fresh) log_visit(where, log);
            }
        }
        if (tid == 0) {
            s.ctl[C_LO] = 0;
            pctl[0] = 0;
            pctl[1] = 0;
        }
        __syncthreads();
        CZ_PH_START();
        const int width = level == 0 ? ix.w0 : ix.wu;
        const bool use_p = ef > 2 * kPend;  // (P never holds more entries than W: pending_inside indexes W[ef - t])
        int t_guess = 0;
        for (;;) {
            cnt = s.ctl[C_CNT];
            const int p = pctl[0];
            const int t = p > 0 ? pending_inside_near(p, ef, t_guess) : 0;  // (p > 0 only when cnt == ef)
            t_guess = t;
            const int wlim = cnt - t;
            // nearest un-expanded entry of W[0, wlim) and of P[0, t) (uniform across the workgroup)
            int iw = -1, ip = -1;
            for (int b = s.ctl[C_LO]; b < wlim; b += 64) {
                const int j = b + lane;
                const bool un = j < wlim && !(s.wid[j] & kExpanded);
                const unsigned long long m = __ballot(un);
                if (m) {
                    iw = b + __ffsll((long long)m) - 1;
                    break;
                }
            }
            for (int b = pctl[1]; b < t; b += 64) {
                const int j = b + lane;
                const bool un = j < t && !(pid[j] & kExpanded);
                const unsigned long long m = __ballot(un);
                if (m) {
                    ip = b + __ffsll((long long)m) - 1;
                    break;
                }
            }
            if (iw < 0 && ip < 0) break;
            const bool from_p = ip >= 0 && (iw < 0 || key_lt(pkey[ip], pid[ip] & kIdMask, s.wkey[iw], s.wid[iw] & kIdMask));
            const uint32_t cand = (from_p ? pid[ip] : s.wid[iw]) & kIdMask;
            CZ_PH_COUNT(5, 1);
            __syncthreads();  // everyone has read the lists' heads before they change
            if (tid == 0) {
                if (from_p) {
                    pid[ip] = cand | kExpanded;
                    pctl[1] = ip + 1;
                    if (iw >= 0) s.ctl[C_LO] = iw;
                } else {
                    s.wid[iw] = cand | kExpanded;
                    s.ctl[C_LO] = iw + 1;
                    if (ip >= 0) pctl[1] = ip;
                }
            }
            CZ_PH_MARK(0);
            if (wave == 0) {  // neighbour row + visited filter, hnsw.rs:566-571
                const int total = expand_row(cand, level, width, tcur, log);
                if (lane == 0) {
                    s.ctl[C_TODO] = total;
                    count_dist(total);
                }
            }
            __syncthreads();
            CZ_PH_MARK(1);
            const int n = s.ctl[C_TODO];
            if (n == 0) continue;
            CZ_PH_COUNT(6, n);
            eval_todo(n);
            __syncthreads();
            CZ_PH_MARK(3);
            if (cnt < ef || !use_p) merge(n, ef);  // W still fills (P is empty) / a short list (the upper levels' ef = 1)
            else t_guess = pending_merge(n, ef, t);
            CZ_PH_MARK(4);
        }
        __syncthreads();
        flush_pending(ef);
    }

    // The same traversal for a batch that leaves the chip empty (HnswSearchRA::iter hands over whatever the parent relation holds --
    // often ONE vector, query/ra.rs:1085-1121).  A step is then a chain of dependent round trips -- link row -> visited atomics ->
    // vector rows -- on a CU that has nothing else to do, and two of the three are taken off the chain without changing anything
    // that is computed:
    //   * the visited set's hash table lives in LDS (the launcher gives this kernel what the list leaves free, up to 128 KiB:
    //     32 768 slots; beyond 70 % load it moves to the global bitmap like the global table does): a test-and-set is an LDS
    //     atomic, not an L2 round trip;
    //   * while a step runs, wave 1 fetches the link row of the nearest un-expanded entry behind the one being expanded -- the next
    //     step's candidate unless this step's merge puts a nearer one in front of it (then the row is fetched as usual).  The load
    //     is issued before the step's barriers and consumed after its evaluation: it rides under the vector rows' round trip.
    // (Evaluating ALL neighbours while the visited test is in flight was tried first: a CU fetches ~50 GB/s, and the 60 % more
    // rows cost what the shorter chain saved -- profiles/r06_small_batch_latency.txt.)
    // Same W, same visited set, same ids / distances / n_dist as search_level; link rows of at most 64 entries.
    __device__ void search_level_spec(int level, int ef, bool log) {
        int cnt = s.ctl[C_CNT];
        for (int i = tid; i < cnt; i += kThreads) s.wid[i] &= kIdMask;
        if (wave == 0) {
            for (int b = 0; b < cnt; b += 64) {
                const int j = b + lane;
                uint32_t where;
                const bool fresh = visit(j < cnt ? (s.wid[j] & kIdMask) : CZ_NONE, j < cnt, where);
                if (fresh) log_visit(where, log);
            }
        }
        if (tid == 0) {
            s.ctl[C_LO] = 0;
            spec[128] = spec[129] = CZ_NONE;
        }
        __syncthreads();
        const int width = level == 0 ? ix.w0 : ix.wu;
        auto row_of = [&](uint32_t node) {
            return level == 0 ? ix.nbr0 + (size_t)node * ix.w0 : ix.up_nbrs + ((size_t)ix.up_base[node] + (level - 1)) * ix.wu;
        };
        for (uint32_t step = 0;; step++) {
            cnt = s.ctl[C_CNT];
            int idx = -1;
            for (int b = s.ctl[C_LO]; b < cnt; b += 64) {
                int j = b + lane;
                bool un = j < cnt && !(s.wid[j] & kExpanded);
                unsigned long long m = __ballot(un);
                if (m) {
                    idx = b + __ffsll((long long)m) - 1;
                    break;
                }
            }
            if (idx < 0) break;
            const uint32_t cand = s.wid[idx] & kIdMask;
            const uint32_t rd = (step & 1u) ^ 1u, wr = step & 1u;  // the buffer the step before filled / the one this step fills
            const bool hit = spec[128 + rd] == cand;
            // wave 1: the next step's link row (the nearest un-expanded entry BEHIND idx: everything before idx is expanded)
            uint32_t pre = CZ_NONE, pre_c = CZ_NONE;
            if (wave == 1) {
                for (int b = idx + 1; b < cnt; b += 64) {
                    const int j = b + lane;
                    const bool un = j < cnt && !(s.wid[j] & kExpanded);
                    const unsigned long long m = __ballot(un);
                    if (m) {
                        pre_c = s.wid[b + __ffsll((long long)m) - 1] & kIdMask;
                        break;
                    }
                }
                if (pre_c != CZ_NONE && lane < width) pre = row_of(pre_c)[lane];
            }
            __syncthreads();  // everyone has read wid[idx] / C_LO / the prefetch tags before they change
            if (tid == 0) {
                s.wid[idx] = cand | kExpanded;
                s.ctl[C_LO] = idx + 1;
            }
            if (wave == 0) {  // the link row -> its not-yet-visited neighbours, in row order (hnsw.rs:566-571)
                const uint32_t nb = lane < width ? (hit ? spec[rd * 64 + lane] : row_of(cand)[lane]) : CZ_NONE;
                uint32_t where;
                const bool fresh = visit(nb, nb != CZ_NONE, where);
                const unsigned long long m = __ballot(fresh);
                if (fresh) {
                    tcur[__popcll(m & ((1ull << lane) - 1ull))] = nb;
                    log_visit(where, log);
                }
                if (lane == 0) {
                    s.ctl[C_TODO] = __popcll(m);
                    count_dist(__popcll(m));
                }
            }
            __syncthreads();
            const int n = s.ctl[C_TODO];
            if (n > 0) {
                eval_todo(n);
                __syncthreads();
            }
            if (wave == 1) {  // (the load has had the whole evaluation to arrive)
                spec[wr * 64 + lane] = pre;
                if (lane == 0) spec[128 + wr] = pre_c;
            }
            if (n > 0) merge(n, ef);
            else __syncthreads();  // (merge ends on a barrier; the prefetch tags are read after one either way)
        }
    }

    // distance to the entry point seeds W (hnsw.rs:915-918): one row, evaluated by the first lane group
    __device__ void seed(uint32_t entry) {
        if (tid == 0) {
            s.ctl[C_CNT] = 0;
            s.ctl[C_NDIST_LO] += 1;
        }
        if constexpr (F64) {
            if (group == 0) {
                const double2 *rows[1] = {(const double2 *)(ix.vec64 + (size_t)entry * ix.ld)};
                double d[1];
                czd64::group_distances<LPV, 1>(ix.metric, (const double2 *)s.q, glane, chunks, qnorm64, rows, d);
                if (glane == 0) {
                    s.wkey[0] = dist_key(d[0]);
                    s.wid[0] = entry;
                    s.ctl[C_CNT] = 1;
                }
            }
            __syncthreads();
            return;
        }
        if (group == 0) {
            const float4 *row = (const float4 *)(ix.vec + (size_t)entry * ix.ld);
            float a0 = 0.f, a1 = 0.f;
            if constexpr (ITERS > 0) {
#pragma unroll
                for (int j = 0; j < ITERS; j++) {
                    const int c = glane + LPV * j;
                    const float4 v = c < chunks ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                    acc_chunk(ix.metric, q[j], v, a0, a1);
                }
            } else {
                for (int c = glane; c < chunks; c += LPV) acc_chunk(ix.metric, s.q[c], row[c], a0, a1);
            }
            float r2[2] = {a0, a1};
            group_reduce_many<LPV, 2>(r2);
            if (glane == 0) {
                s.wkey[0] = dist_key(finish_distance(ix.metric, r2[0], r2[1], qnorm));
                s.wid[0] = entry;
                s.ctl[C_CNT] = 1;
            }
        }
        __syncthreads();
    }
};

// Column predicates of a filtered search, evaluated on the device over the ef candidates (hnsw.rs:943-947 keeps all ef
// when a filter is present, :997-1006 filters then truncates to k): a conjunction of `column[node] OP constant` over
// per-node numeric columns resident in HBM.  Comparison semantics are the reference's (data/functions.rs:298-380 over
// data/value.rs:575-598): Int with Int as integers, Float with Float by f64::total_cmp (so -0.0 < 0.0 and NaN == NaN),
// mixed pairs as IEEE f64 after `as f64`.
struct PredTerm {
    const void *col;  // f64 or i64, [n]
    int col_is_int, const_is_int, op, pad;
    double fv;
    long long iv;
};
constexpr int kMaxPreds = 4;
struct PredSet {
    int n;
    int pad;
    PredTerm t[kMaxPreds];
};
__device__ __forceinline__ long long f64_total_key(double d) {  // order-preserving integer image of f64::total_cmp
    const long long b = __double_as_longlong(d);
    return b ^ (long long)((unsigned long long)(b >> 63) >> 1);
}
__device__ __forceinline__ bool pred_cmp(int op, int c /* -1, 0, 1 */) {
    switch (op) {
        case 0: return c < 0;    // CZ_OP_LT
        case 1: return c <= 0;   // CZ_OP_LE
        case 2: return c == 0;   // CZ_OP_EQ
        case 3: return c >= 0;   // CZ_OP_GE
        case 4: return c > 0;    // CZ_OP_GT
        default: return c != 0;  // CZ_OP_NE
    }
}
__device__ __forceinline__ bool pred_ieee(int op, double l, double r) {
    switch (op) {
        case 0: return l < r;
        case 1: return l <= r;
        case 2: return l == r;
        case 3: return l >= r;
        case 4: return l > r;
        default: return l != r;
    }
}
__device__ __forceinline__ bool pred_pass(const PredSet &ps, uint32_t node) {
    for (int i = 0; i < ps.n; i++) {
        const PredTerm &t = ps.t[i];
        bool ok;
        if (t.col_is_int) {
            const long long v = ((const long long *)t.col)[node];
            if (t.const_is_int) ok = pred_cmp(t.op, v < t.iv ? -1 : (v > t.iv ? 1 : 0));
            else ok = pred_ieee(t.op, (double)v, t.fv);
        } else {
            const double v = ((const double *)t.col)[node];
            if (t.const_is_int) ok = pred_ieee(t.op, v, (double)t.iv);
            else {
                const long long a = f64_total_key(v), b = f64_total_key(t.fv);
                ok = pred_cmp(t.op, a < b ? -1 : (a > b ? 1 : 0));
            }
        }
        if (!ok) return false;
    }
    return true;
}

// hnsw_knn (hnsw.rs:869-1012): one workgroup per query
// The search streams base rows with the non-temporal hint: same-box A/B at 1M x 768, batch 1024: 3.447 -> 3.267 ms
// (the rows are read once; without the hint they push link rows and visited words out of L2).  Index construction
// does NOT (its selection heuristic re-reads rows through L2 / Infinity Cache; NT cost it 40 %).
#ifndef CZ_SEARCH_NT
#define CZ_SEARCH_NT 1
#endif
// The grid is PERSISTENT: min(B, the workgroups the chip holds at once) workgroups, workgroup i takes queries i, i + grid,
// ... and owns ONE visited workspace (slot i) for all of them -- a batch of 4 096 queries then works on the 1 024 tables
// that stay in L2 / the Infinity Cache instead of 4 096 (512 MB) that do not (profiles/r04_batch_size.txt: 0.66 of the peak
// at B = 2 048 / 4 096 against 0.76 at 1 024).  U (rows in flight per lane group and round) is chosen from B by the
// launcher: a batch that leaves most of the chip empty is bound by the latency of a step, and a step by the rounds its
// rows take -- wider rounds (U = 4 / 8: 16 / 32 rows per round, the registers are free at that occupancy) shorten it.
template <int LPV, int ITERS, int U, bool F64 = false, bool SPEC = false, bool PEND = false>
__device__ __forceinline__ void
hnsw_knn_body(const IndexDev &ix, const float *__restrict__ queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t efcap,
              uint32_t wpad, int has_radius, double radius, uint32_t *__restrict__ vtab, uint32_t hbits,
              uint32_t *__restrict__ vbitmap, uint32_t words, const PredSet &preds, uint32_t *__restrict__ out_ids,
              double *__restrict__ out_dist, uint32_t *__restrict__ out_count, unsigned long long *__restrict__ out_n_dist) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem s = carve(smem_raw, efcap, wpad, F64 ? ix.ld * 2 : ix.ld, false);  // (the query row: ld floats, or ld doubles)
    VisitedDev vis;
    vis.tab = hbits ? vtab + ((size_t)blockIdx.x << hbits) : nullptr;
    vis.hbits = hbits;
    if constexpr (SPEC) {  // the hash table of the visited set sits in LDS behind the list (hbits = what the launcher found room for)
        const size_t at = (smem_bytes(efcap, wpad, F64 ? ix.ld * 2 : ix.ld, false) + 15) & ~(size_t)15;
        vis.tab = hbits ? (uint32_t *)(smem_raw + at) : nullptr;
        for (uint32_t i = threadIdx.x; i < (hbits ? (1u << hbits) : 0u); i += kThreads) vis.tab[i] = CZ_NONE;
        __syncthreads();
    }
    vis.bitmap = vbitmap + (size_t)blockIdx.x * words;
    vis.words = words;
    for (uint32_t b = blockIdx.x; b < B; b += gridDim.x) {
    if (b != blockIdx.x) __syncthreads();  // (the output stage of the query before this one has read the list)
    Searcher<LPV, ITERS, U, CZ_SEARCH_NT != 0, F64> S(ix, s, vis);
    if constexpr (SPEC) {
        __shared__ uint32_t spec_words[132];  // two prefetched link rows (64 ids each) and their tags
        S.spec = spec_words;
    }
    if constexpr (PEND) {  // the pending buffer in front of W (search_level_pending): 4 KiB of static LDS
        __shared__ uint64_t pend_key[kThreads];
        __shared__ uint32_t pend_id[kThreads], pend_lb[kThreads];
        __shared__ int pend_ctl[2];
        S.pkey = pend_key;
        S.pid = pend_id;
        S.plb = pend_lb;
        S.pctl = pend_ctl;
    }
    S.load_query(F64 ? reinterpret_cast<const float *>(reinterpret_cast<const double *>(queries) + (size_t)b * ix.dim)
                     : queries + (size_t)b * ix.dim);
    S.seed(ix.entry);
    // :919-938 greedy descent with ef = 1 through the upper levels, then the level-0 search with ef (one call site,
    // so that the traversal is inlined once)
    for (int lv = ix.n_levels - 1; lv >= 0; lv--) {
        if constexpr (SPEC) S.search_level_spec(lv, lv > 0 ? 1 : (int)ef, lv > 0);
        else if constexpr (PEND) S.search_level_pending(lv, lv > 0 ? 1 : (int)ef, lv > 0);  // (one call site here too)
        else S.search_level(lv, lv > 0 ? 1 : (int)ef, lv > 0);
        if (lv > 0) S.clear_visited();
    }
    S.clear_all();  // the table / bitmap go back to the pool empty
#ifdef CZ_PHASE_TIMING
    if (threadIdx.x == 0) {
        for (int i_ = 0; i_ < 12; i_++) atomicAdd(&cz_phase_cycles[i_], S.ph_acc[i_]);
    }
#endif
    const int cnt = s.ctl[C_CNT];
    int total;
    if (preds.n > 0) {
        // filtered search (:943-947, :997-1006): ALL ef candidates are looked at in ascending order -- radius cut, then the
        // predicates -- and the first k survivors are the result
        for (int i = threadIdx.x; i < cnt; i += kThreads) {
            const uint64_t key = s.wkey[i];
            bool ok = !(has_radius && key != ~0ull && key_dist(key) > radius);
            if (ok) ok = pred_pass(preds, s.wid[i] & kIdMask);
            s.st[i] = ok ? 1 : 0;
        }
        for (int j = threadIdx.x; j < (int)k; j += kThreads) {
            out_ids[(size_t)b * k + j] = CZ_NONE;
            out_dist[(size_t)b * k + j] = __longlong_as_double(0x7FF0000000000000ll);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // one wave walks the flags in order
            const int lane = threadIdx.x;
            int kept = 0;
            for (int b0 = 0; b0 < cnt && kept < (int)k; b0 += 64) {
                const int i = b0 + lane;
                const bool ok = i < cnt && s.st[i] != 0;
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int pos = kept + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < (int)k) {
                        out_ids[(size_t)b * k + pos] = s.wid[i] & kIdMask;
                        out_dist[(size_t)b * k + pos] = key_dist(s.wkey[i]);
                    }
                }
                kept += __popcll(m);
            }
            if (lane == 0) s.ctl[C_KEEP] = min(kept, (int)k);
        }
        __syncthreads();
        total = s.ctl[C_KEEP];
    } else {
    // :943-1006 truncate to k, radius cut (`distance > r` => skip; a NaN distance is never > r), ascending
    const int kk = min((int)k, cnt);
    int c_keep = 0, c_num = 0;
    for (int i = threadIdx.x; i < kk; i += kThreads) {
        uint64_t key = s.wkey[i];
        if (key != ~0ull) {
            c_num++;
            if (!(has_radius && key_dist(key) > radius)) c_keep++;
        }
    }
    if (c_keep) atomicAdd(&s.ctl[C_KEEP], c_keep);
    if (c_num) atomicAdd(&s.ctl[C_NUM], c_num);
    __syncthreads();
    // W is sorted: the kept finite entries are a prefix [0,p); NaN entries sit at [first_nan, kk)
    const int p = s.ctl[C_KEEP], first_nan = s.ctl[C_NUM];
    total = p + (kk - first_nan);
    for (int j = threadIdx.x; j < (int)k; j += kThreads) {
        uint32_t id = CZ_NONE;
        double d = __longlong_as_double(0x7FF0000000000000ll);
        int src = j < p ? j : (j < total ? first_nan + (j - p) : -1);
        if (src >= 0) {
            id = s.wid[src] & kIdMask;
            d = key_dist(s.wkey[src]);
        }
        out_ids[(size_t)b * k + j] = id;
        out_dist[(size_t)b * k + j] = d;
    }
    }
    if (threadIdx.x == 0) {
        out_count[b] = (uint32_t)total;
        if (out_n_dist)
            out_n_dist[b] = ((unsigned long long)(unsigned int)s.ctl[C_NDIST_HI] << 32) |
                            (unsigned long long)(unsigned int)s.ctl[C_NDIST_LO];
    }
    }  // the workgroup's next query
}

// the batch fills the chip: four workgroups per CU, i.e. at most 128 VGPRs (said explicitly: the loop over the
// workgroup's queries took the compiler's own choice to 141 and the occupancy to three)
template <int LPV, int ITERS, int U>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
hnsw_knn_kernel(IndexDev ix, const float *__restrict__ queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t efcap,
                uint32_t wpad, int has_radius, double radius, uint32_t *__restrict__ vtab, uint32_t hbits,
                uint32_t *__restrict__ vbitmap, uint32_t words, PredSet preds, uint32_t *__restrict__ out_ids,
                double *__restrict__ out_dist, uint32_t *__restrict__ out_count, unsigned long long *__restrict__ out_n_dist) {
    hnsw_knn_body<LPV, ITERS, U>(ix, queries, B, k, ef, efcap, wpad, has_radius, radius, vtab, hbits, vbitmap, words, preds, out_ids,
                                 out_dist, out_count, out_n_dist);
}
// a batch that leaves most of the chip empty: more rows in flight per lane group, as many registers as that takes
template <int LPV, int ITERS, int U>
__global__ void __launch_bounds__(kThreads)
hnsw_knn_wide_kernel(IndexDev ix, const float *__restrict__ queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t efcap,
                     uint32_t wpad, int has_radius, double radius, uint32_t *__restrict__ vtab, uint32_t hbits,
                     uint32_t *__restrict__ vbitmap, uint32_t words, PredSet preds, uint32_t *__restrict__ out_ids,
                     double *__restrict__ out_dist, uint32_t *__restrict__ out_count, unsigned long long *__restrict__ out_n_dist) {
    hnsw_knn_body<LPV, ITERS, U>(ix, queries, B, k, ef, efcap, wpad, has_radius, radius, vtab, hbits, vbitmap, words, preds, out_ids,
                                 out_dist, out_count, out_n_dist);
}
// a batch of a few queries (at most a workgroup per two CUs): the speculative step (search_level_spec), eight rows in flight
template <int LPV, int ITERS, int U>
__global__ void __launch_bounds__(kThreads)
hnsw_knn_spec_kernel(IndexDev ix, const float *__restrict__ queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t efcap,
                     uint32_t wpad, int has_radius, double radius, uint32_t *__restrict__ vtab, uint32_t hbits,
                     uint32_t *__restrict__ vbitmap, uint32_t words, PredSet preds, uint32_t *__restrict__ out_ids,
                     double *__restrict__ out_dist, uint32_t *__restrict__ out_count, unsigned long long *__restrict__ out_n_dist) {
    hnsw_knn_body<LPV, ITERS, U, false, true>(ix, queries, B, k, ef, efcap, wpad, has_radius, radius, vtab, hbits, vbitmap, words, preds,
                                              out_ids, out_dist, out_count, out_n_dist);
}
// a large ef (the list is thousands of entries): the sorted pending buffer in front of W (search_level_pending)
template <int LPV, int ITERS, int U>
__global__ void __launch_bounds__(kThreads)
hnsw_knn_pend_kernel(IndexDev ix, const float *__restrict__ queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t efcap,
                     uint32_t wpad, int has_radius, double radius, uint32_t *__restrict__ vtab, uint32_t hbits,
                     uint32_t *__restrict__ vbitmap, uint32_t words, PredSet preds, uint32_t *__restrict__ out_ids,
                     double *__restrict__ out_dist, uint32_t *__restrict__ out_count, unsigned long long *__restrict__ out_n_dist) {
    hnsw_knn_body<LPV, ITERS, U, false, false, true>(ix, queries, B, k, ef, efcap, wpad, has_radius, radius, vtab, hbits, vbitmap, words,
                                                     preds, out_ids, out_dist, out_count, out_n_dist);
}
// an F64 index (the query rows are doubles behind the float pointer): LPV lanes per vector, the query in LDS
template <int LPV, int U>
__global__ void __launch_bounds__(kThreads)
hnsw_knn_f64_kernel(IndexDev ix, const float *__restrict__ queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t efcap,
                    uint32_t wpad, int has_radius, double radius, uint32_t *__restrict__ vtab, uint32_t hbits,
                    uint32_t *__restrict__ vbitmap, uint32_t words, PredSet preds, uint32_t *__restrict__ out_ids,
                    double *__restrict__ out_dist, uint32_t *__restrict__ out_count, unsigned long long *__restrict__ out_n_dist) {
    hnsw_knn_body<LPV, 0, U, true>(ix, queries, B, k, ef, efcap, wpad, has_radius, radius, vtab, hbits, vbitmap, words, preds, out_ids,
                                   out_dist, out_count, out_n_dist);
}

}  // namespace czh
