// hnsw_build.hip -- HNSW index construction on gfx950 (C ABI: cz_hnsw_build, cz_hnsw_index_export_*).
//
// Restates hnsw_put_vector / hnsw_select_neighbours_heuristic / hnsw_shrink_neighbour
// (cozo-core/src/runtime/hnsw.rs:155-538) as a batch-parallel insert:
//   K1 insert  : one workgroup per new vector: greedy descent (ef = 1) to its level, then per level
//                hnsw_search_level(ef_construction) + the select-neighbours heuristic (m_max0 on level 0, m_max above,
//                hnsw.rs:243-267); writes the vector's own link row and queues one reverse-link request per neighbour.
//   K2 link    : appends the reverse links (:300-318) with an atomic slot counter per row; a row that grows past its
//                width is queued for shrinking (:338-350).
//   K3 shrink  : one workgroup per over-full row: re-select among its live links with the same heuristic, using the
//                stored link distances (:389-393); dropped links disappear from this row only (one-directional,
//                :434-466 -- the reference soft-deletes them, which is invisible to every reader on this path).
// extend_candidates (:499-511): the heuristic's candidates are the found set plus the neighbours of its members -- gathered
// into a per-workgroup scratch array in global memory, sorted there and fed through the LDS list chunk by chunk
// (hnsw_kernels.h select_extended).  In a shrink the target reaches ITSELF through its neighbours' back links and is
// selected like anything else; the reference writes that "link" onto the target's self row and puts the self row back
// (:413-433, :352-357), so all that remains is a degree one above the number of link rows: `ph` below.  A shrink now reads
// OTHER rows, so the selections of one round are staged and applied afterwards (every shrink of a round sees the rows as
// the round found them), and max_batch = 1 links and shrinks one neighbour at a time, in the order of selection.
// Rows that carry several indexed vectors (:694-706): hnsw_get_neighbours drops every link inside one base row (:609-610), so
// such a link is written and counted into both degrees but never read again.  With T.row_of set, the tables simply do not
// hold them (no reader could tell) and the degree word counts them: word = degree | hidden << 16, hidden = the links the
// degree counts that hold no slot -- links inside a row, and the self link of an extended shrink.
// Vectors of one batch do not see each other (they are linked after the batch's searches), which is the only
// difference from the reference's one-at-a-time insertion: with max_batch = 1 the link tables are identical to the
// sequential algorithm's.  Levels are drawn by the caller (hnsw.rs:46-52 uses an unseedable thread_rng).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "common.h"
#include "hnsw_index.h"
#include "hnsw_kernels.h"

using namespace czd;
using czh::IndexDev;
using czh::kIdMask;
using czh::kThreads;

namespace {

struct BuildTables {
    uint32_t *nbr0;
    double *dst0;
    uint32_t *deg0;        // degree word per row: the self row's degree | hidden << 16 (deg_of / hid_of)
    const uint8_t *hid_in0;  // what an earlier build left hidden (read by the unpack kernel only)
    int w0, cap0;
    const uint32_t *up_base;
    uint32_t *nbrU;
    double *dstU;
    uint32_t *degU;
    const uint8_t *hid_inU;
    int wu, capU;
    const int32_t *level;
    const uint32_t *row_of;  // base row of every node's vector, or nullptr: one vector per row
};
constexpr uint32_t kHiddenOne = 0x10001u;  // one more link in the degree that holds no slot
constexpr int32_t kReqHidden = 0x40000000;  // request flag (in its level word): the two ends lie in one base row
__host__ __device__ __forceinline__ uint32_t deg_of(uint32_t w) { return w & 0xFFFFu; }
__host__ __device__ __forceinline__ uint32_t hid_of(uint32_t w) { return w >> 16; }
__device__ __forceinline__ bool same_row(const BuildTables &T, uint32_t a, uint32_t b) {
    return T.row_of && T.row_of[a] == T.row_of[b];
}
// extend_candidates scratch: per workgroup `cap` candidate slots (a power of two) and a copy of the found list
struct ExtBuf {
    uint64_t *key;
    uint32_t *id;
    uint32_t cap, keep;  // slots per workgroup = cap + keep
    __device__ uint64_t *cand_key(uint32_t wg) const { return key + (size_t)wg * (cap + keep); }
    __device__ uint32_t *cand_id(uint32_t wg) const { return id + (size_t)wg * (cap + keep); }
};
// selections of one round of shrinks, applied after the round (extend_candidates)
struct Stage {
    uint32_t *ids;  // [rows][width]
    double *dst;
    uint32_t *n;    // [rows] links kept
    uint8_t *self;  // [rows] selected but not kept: the target itself, vectors of the target's own base row
    int width;
};
struct Req {
    uint32_t *t, *q;
    int32_t *lv;
    double *d;
};
struct RowRef {
    uint32_t *ids;
    double *dst;
    uint32_t *deg;
    const uint8_t *hid_in;
    int width, cap;
};
__device__ __forceinline__ RowRef row_of(const BuildTables &T, uint32_t node, int lv) {
    RowRef r;
    if (lv == 0) {
        r.ids = T.nbr0 + (size_t)node * T.cap0;
        r.dst = T.dst0 + (size_t)node * T.cap0;
        r.deg = T.deg0 + node;
        r.hid_in = T.hid_in0 + node;
        r.width = T.w0;
        r.cap = T.cap0;
    } else {
        const size_t row = (size_t)T.up_base[node] + (lv - 1);
        r.ids = T.nbrU + row * T.capU;
        r.dst = T.dstU + row * T.capU;
        r.deg = T.degU + row;
        r.hid_in = T.hid_inU + row;
        r.width = T.wu;
        r.cap = T.capU;
    }
    return r;
}

// K1
template <int LPV, int ITERS, int U, bool EXT>
__global__ void __launch_bounds__(kThreads)
build_insert_kernel(IndexDev ix, BuildTables T, uint32_t b0, uint32_t bn, int top, uint32_t entry, int ef_c,
                    uint32_t efcap, uint32_t wcap, int keep_pruned, uint32_t *__restrict__ vtab, uint32_t hbits,
                    uint32_t *__restrict__ vbitmap, uint32_t words, Req req,
                    uint32_t *__restrict__ req_count, uint32_t req_cap, unsigned long long *__restrict__ ndist_total,
                    ExtBuf ext, int defer_out /* the out links are written one by one, by the link kernel */) {
    constexpr bool extend = EXT;  // a template parameter: the plain build keeps its registers (the extended path spills)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    czh::Smem s = czh::carve(smem_raw, efcap, wcap, ix.ld);
    czh::VisitedDev vis;
    vis.tab = hbits ? vtab + ((size_t)blockIdx.x << hbits) : nullptr;
    vis.hbits = hbits;
    vis.bitmap = vbitmap + (size_t)blockIdx.x * words;
    vis.words = words;
    czh::Searcher<LPV, ITERS, U> S(ix, s, vis);
    const int tid = threadIdx.x;
    for (uint32_t qi = blockIdx.x; qi < bn; qi += gridDim.x) {
        const uint32_t q = b0 + qi;
        S.load_query(ix.vec + (size_t)q * ix.ld);
        S.seed(entry);  // hnsw.rs:200-204
        const int lq = T.level[q];
        for (int lv = top; lv >= 0; lv--) {
            const bool above = lv > lq;  // :219-229 greedy descent (ef = 1) above the vector's own top level
            S.search_level(lv, above ? 1 : ef_c, true);  // one call site: the traversal is inlined once
            if (above) {
                S.clear_visited();
                continue;
            }
            // :242-359
            const RowRef r = row_of(T, q, lv);
            int nsel;
            int found_cnt = 0;
            uint64_t *gk = nullptr;
            uint32_t *gi = nullptr;
            if constexpr (EXT) {
                gk = ext.cand_key(blockIdx.x);
                gi = ext.cand_id(blockIdx.x);
                found_cnt = s.ctl[czh::C_CNT];  // found_nn goes on to the next level as it is (:248-256): a copy
                for (int k = tid; k < found_cnt; k += kThreads) {
                    gk[ext.cap + k] = s.wkey[k];
                    gi[ext.cap + k] = s.wid[k];
                }
                S.clear_visited();
                const uint32_t nc = S.gather_extended(lv, gk, gi, wcap, ext.cap);
                S.sort_scratch(gk, gi, nc);
                nsel = S.select_extended(gk, gi, nc, r.width, keep_pruned != 0, efcap);
            } else {
                nsel = S.select_heuristic(r.width, keep_pruned != 0);
            }
            if (tid == 0) s.ctl[czh::C_KEEP] = (int)atomicAdd(req_count, (uint32_t)nsel);
            // a selected neighbour inside the vector's own base row: link rows and both degrees as for any other (:281-357),
            // but no reader will ever see the link (:609-610) -- it gets no slot, here or in the neighbour's row
            uint32_t id = CZ_NONE;
            double d = 0.0;
            bool hidden = false;
            if (tid < nsel) {  // (nsel <= m_max0 <= 192 < kThreads)
                const uint32_t p = s.sel[tid];
                id = EXT ? p : s.wid[p] & kIdMask;
                d = key_dist(EXT ? S.ext_sel_key()[tid] : s.wkey[p]);
                hidden = same_row(T, id, q);
                s.st[tid] = hidden ? 1 : 0;
            }
            const int nhid = __syncthreads_count(hidden);
            const uint32_t base = (uint32_t)s.ctl[czh::C_KEEP];
            if (tid < nsel) {
                int before = 0;
                if (nhid)
                    for (int j = 0; j < tid; j++) before += s.st[j];
                if (!hidden) {
                    r.ids[tid - before] = defer_out ? CZ_NONE : id;
                    r.dst[tid - before] = d;
                }
                if (base + tid < req_cap) {  // the host checks the final count against req_cap
                    req.t[base + tid] = id;
                    req.q[base + tid] = q;
                    req.lv[base + tid] = lv | (hidden ? kReqHidden : 0);
                    req.d[base + tid] = d;
                }
            }
            for (int k = nsel - nhid + tid; k < r.cap; k += kThreads) r.ids[k] = CZ_NONE;
            if (tid == 0) *r.deg = (uint32_t)nsel | ((uint32_t)nhid << 16);  // the self row's degree, :269-277
            if (EXT && lv > 0) {  // W was the heuristic's chunk buffer
                __syncthreads();
                for (int k = tid; k < found_cnt; k += kThreads) {
                    s.wkey[k] = gk[ext.cap + k];
                    s.wid[k] = gi[ext.cap + k];
                }
                if (tid == 0) s.ctl[czh::C_CNT] = found_cnt;
            }
            S.clear_visited();
        }
        if (tid == 0) {
            const unsigned long long nd = ((unsigned long long)(unsigned int)s.ctl[czh::C_NDIST_HI] << 32) |
                                          (unsigned long long)(unsigned int)s.ctl[czh::C_NDIST_LO];
            atomicAdd(ndist_total, nd);
        }
        __syncthreads();
    }
}

// K2
__global__ void __launch_bounds__(256)
build_link_kernel(BuildTables T, Req in, uint32_t n, int lazy, Req retry, uint32_t *__restrict__ retry_count,
                  uint32_t *__restrict__ shrink_t, int32_t *__restrict__ shrink_lv, uint32_t *__restrict__ shrink_count,
                  int out_links, int upsert) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t t = in.t[i], q = in.q[i];
        const int lv = in.lv[i] & ~kReqHidden;
        const bool hidden = (in.lv[i] & kReqHidden) != 0;
        const double d = in.d[i];
        if (out_links && !hidden) {  // (one request per launch) the out link goes in with its reverse link, hnsw.rs:281-318: a
            const RowRef rq = row_of(T, q, lv);  // shrink between two of them sees the new vector's row as far as it has got
            int k = 0;
            while (k < rq.cap - 1 && rq.ids[k] != CZ_NONE) k++;
            rq.ids[k] = t;
            rq.dst[k] = d;
        }
        const RowRef r = row_of(T, t, lv);
        if (hidden) {  // counted (:338), never read (:609-610): no slot; the shrink test is the reference's
            const uint32_t old = atomicAdd(r.deg, kHiddenOne);
            // (lazy builds: such links fill no slot, so the slot test never sees them -- every 16th one onto a row whose degree
            // is past its width asks for the shrink that forgets them, which also keeps the two 16-bit halves far from full)
            if (lazy ? (deg_of(old) >= (uint32_t)r.width && (hid_of(old) & 15u) == 15u) : deg_of(old) == (uint32_t)r.width) {
                const uint32_t p = atomicAdd(shrink_count, 1u);
                shrink_t[p] = t;
                shrink_lv[p] = lv;
            }
            continue;
        }
        if (upsert) {
            // extend_candidates, batched: while this request waited for room, a shrink of `t` may have picked q up through
            // the extension and linked it already.  The reference's put of an existing row replaces it (:300-318): no second
            // entry, no degree change beyond the one the shrink accounted for.
            bool held = false;
            for (int k = 0; k < r.cap; k++) held |= r.ids[k] == q;
            if (held) continue;
        }
        const uint32_t old = atomicAdd(r.deg, 1u);  // the degree of the self row, :338 (one word with the hidden count: a
        const uint32_t slot = deg_of(old) - hid_of(old);  // concurrent hidden link cannot slip between the two)
        if (slot < (uint32_t)r.cap) {
            r.ids[slot] = q;
            r.dst[slot] = d;
            // degree just exceeded the row width: shrink (:339).  Lazy form (batched builds): only when the row's
            // slack slots are used up too -- one re-selection per `slack` appended links instead of one per link
            if (lazy ? slot == (uint32_t)(r.cap - 1) : deg_of(old) == (uint32_t)r.width) {
                const uint32_t p = atomicAdd(shrink_count, 1u);
                shrink_t[p] = t;
                shrink_lv[p] = lv;
            }
        } else {  // no room until the row has been shrunk: try again afterwards
            atomicSub(r.deg, 1u);
            const uint32_t p = atomicAdd(retry_count, 1u);
            retry.t[p] = t;
            retry.q[p] = q;
            retry.lv[p] = lv;
            retry.d[p] = d;
        }
    }
}

// lazy builds: rows still wider than their final width when the last batch is in
__global__ void __launch_bounds__(256)
build_overfull_kernel(BuildTables T, uint32_t n, uint32_t *__restrict__ shrink_t, int32_t *__restrict__ shrink_lv,
                      uint32_t *__restrict__ shrink_count, uint32_t cap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int top = T.level[i];
        for (int lv = 0; lv <= top; lv++) {
            const RowRef r = row_of(T, i, lv);
            if (deg_of(*r.deg) - hid_of(*r.deg) > (uint32_t)r.width) {
                const uint32_t p = atomicAdd(shrink_count, 1u);
                if (p < cap) {
                    shrink_t[p] = i;
                    shrink_lv[p] = lv;
                }
            }
        }
    }
}

// K3
template <int LPV, int ITERS, int U, bool EXT>
__global__ void __launch_bounds__(kThreads)
build_shrink_kernel(IndexDev ix, BuildTables T, const uint32_t *__restrict__ shrink_t, const int32_t *__restrict__ shrink_lv,
                    uint32_t n, const uint32_t *__restrict__ n_dev /* or the count is read here (max_batch = 1 + extend) */,
                    uint32_t efcap, uint32_t wcap, int keep_pruned, unsigned long long *__restrict__ ndist_total,
                    uint32_t *__restrict__ vtab, uint32_t hbits, uint32_t *__restrict__ vbitmap, uint32_t words,
                    ExtBuf ext, Stage stage) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    czh::Smem s = czh::carve(smem_raw, efcap, wcap, ix.ld);
    czh::VisitedDev vis{nullptr, 0, nullptr, 0};
    if constexpr (EXT) {
        vis.tab = hbits ? vtab + ((size_t)blockIdx.x << hbits) : nullptr;
        vis.hbits = hbits;
        vis.bitmap = vbitmap + (size_t)blockIdx.x * words;
        vis.words = words;
    }
    czh::Searcher<LPV, ITERS, U> S(ix, s, vis);
    const int tid = threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t t = shrink_t[i];
        const int lv = shrink_lv[i];
        const RowRef r = row_of(T, t, lv);
        S.load_query(ix.vec + (size_t)t * ix.ld);  // hnsw.rs:386-387
        const uint32_t word = *r.deg;
        const int c = (int)min(deg_of(word) - hid_of(word), (uint32_t)r.cap);
        CZ_CHECK(t < ix.n && deg_of(word) >= hid_of(word) && deg_of(word) - hid_of(word) <= (uint32_t)r.cap,
                 "shrink: node %u level %d degree word %x cap %d\n", t, lv, word, r.cap);
        if (tid < c) CZ_CHECK(r.ids[tid] < ix.n, "shrink: node %u level %d slot %d of %d holds %u\n", t, lv, tid, c, r.ids[tid]);
        // candidates = live links with their stored distances (:389-393), sorted by (distance, id)
        uint64_t mk = 0;
        uint32_t mi = CZ_NONE;
        if (tid < c) {
            mk = dist_key(r.dst[tid]);
            mi = r.ids[tid];
            s.nkey[tid] = mk;
            s.nid[tid] = mi;
        }
        __syncthreads();
        if (tid < c) {
            int rank = 0;
            for (int j = 0; j < c; j++) rank += czh::key_lt(s.nkey[j], s.nid[j], mk, mi);
            s.wkey[rank] = mk;
            s.wid[rank] = mi;
        }
        if (tid == 0) s.ctl[czh::C_CNT] = c;
        __syncthreads();
        if constexpr (!EXT) {
            const int nsel = S.select_heuristic(r.width, keep_pruned != 0);
            for (int k = tid; k < r.cap; k += kThreads) {
                if (k < nsel) {
                    const uint32_t p = s.sel[k];
                    r.ids[k] = s.wid[p] & kIdMask;
                    r.dst[k] = key_dist(s.wkey[p]);
                } else {
                    r.ids[k] = CZ_NONE;
                }
            }
            if (tid == 0) *r.deg = (uint32_t)nsel;  // :352,412: links inside the base row are not among the candidates and
                                                    // no longer counted
        } else {
            // the neighbours' neighbours join in (:499-511) -- the target among them, through the back links; what is
            // selected is staged: the other shrinks of this round still read this row as the round found it
            uint64_t *gk = ext.cand_key(blockIdx.x);
            uint32_t *gi = ext.cand_id(blockIdx.x);
            const uint32_t nc = S.gather_extended(lv, gk, gi, wcap, ext.cap);
            S.sort_scratch(gk, gi, nc);
            const int nsel = S.select_extended(gk, gi, nc, r.width, keep_pruned != 0, efcap);
            CZ_CHECK(nsel <= r.width && nsel <= stage.width + 1, "shrink: %d selected, width %d\n", nsel, r.width);
            // selected without a slot: the target itself (its "link row" is the self row, put back by hnsw_put_vector,
            // :352-357) and vectors of the target's own base row (:609-610) -- counted into the degree, kept nowhere
            bool hidden = false;
            uint32_t id = CZ_NONE;
            if (tid < nsel) {
                id = s.sel[tid];
                hidden = id == t || same_row(T, id, t);
                s.st[tid] = hidden ? 1 : 0;
            }
            const int nhid = __syncthreads_count(hidden);
            uint32_t *sid = stage.ids + (size_t)i * stage.width;
            double *sd = stage.dst + (size_t)i * stage.width;
            if (tid < nsel && !hidden) {
                int before = 0;
                if (nhid)
                    for (int j = 0; j < tid; j++) before += s.st[j];
                sid[tid - before] = id;
                sd[tid - before] = key_dist(S.ext_sel_key()[tid]);
            }
            if (tid == 0) {
                stage.n[i] = (uint32_t)(nsel - nhid);
                stage.self[i] = (uint8_t)nhid;
            }
        }
        if (tid == 0) {
            const unsigned long long nd = ((unsigned long long)(unsigned int)s.ctl[czh::C_NDIST_HI] << 32) |
                                          (unsigned long long)(unsigned int)s.ctl[czh::C_NDIST_LO];
            atomicAdd(ndist_total, nd);
        }
        __syncthreads();
    }
}

// the staged selections of one round of shrinks go into the rows (one wave per row)
__global__ void __launch_bounds__(256)
build_apply_kernel(BuildTables T, const uint32_t *__restrict__ shrink_t, const int32_t *__restrict__ shrink_lv, uint32_t n,
                   const uint32_t *__restrict__ n_dev, Stage stage) {
    if (n_dev) n = min(n, *n_dev);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const RowRef r = row_of(T, shrink_t[i], shrink_lv[i]);
        const uint32_t kept = stage.n[i];
        for (int k = lane; k < r.cap; k += 64) {
            if ((uint32_t)k < kept) {
                r.ids[k] = stage.ids[(size_t)i * stage.width + k];
                r.dst[k] = stage.dst[(size_t)i * stage.width + k];
            } else {
                r.ids[k] = CZ_NONE;
            }
        }
        if (lane == 0) *r.deg = (kept + stage.self[i]) | ((uint32_t)stage.self[i] << 16);  // :412: the selected count
    }
}

// An existing index goes back into build form (cz_hnsw_insert): every live row of the packed tables is copied into the
// build tables and its link distances -- the `dist` column of the tbl:idx rows, which hnsw_shrink_neighbour reads
// (:389-393) -- are evaluated again.  They are the same bits the original insertion stored: the kernels' distance is
// symmetric in its two arguments (products, per-lane order and butterfly do not depend on which side is the query).
template <int LPV, int ITERS, int U>
__global__ void __launch_bounds__(kThreads)
build_unpack_kernel(IndexDev ix, BuildTables T, const uint32_t *__restrict__ old_nbr0, int old_w0,
                    const uint32_t *__restrict__ old_up_base, const uint32_t *__restrict__ old_up, int old_wu, uint32_t n_old,
                    uint32_t efcap, uint32_t wcap,
                    unsigned long long *__restrict__ ndist_total) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    czh::Smem s = czh::carve(smem_raw, efcap, wcap, ix.ld);
    czh::Searcher<LPV, ITERS, U> S(ix, s, czh::VisitedDev{nullptr, 0, nullptr, 0});
    const int tid = threadIdx.x;
    for (uint32_t node = blockIdx.x; node < n_old; node += gridDim.x) {
        const int top = T.level[node];
        if (top < 0) continue;  // a removed node: no rows
        S.load_query(ix.vec + (size_t)node * ix.ld);
        for (int lv = 0; lv <= top; lv++) {
            const RowRef r = row_of(T, node, lv);
            // (the packed upper rows are laid out by the OLD row numbering: removals since then renumber the build tables)
            const uint32_t *row = lv == 0 ? old_nbr0 + (size_t)node * old_w0
                                          : old_up + ((size_t)old_up_base[node] + (lv - 1)) * old_wu;
            const int width = lv == 0 ? old_w0 : old_wu;
            // packed rows hold their live links first (ascending), CZ_NONE after them
            const uint32_t id = tid < width ? row[tid] : CZ_NONE;
            const int c = __syncthreads_count(id != CZ_NONE);
            if (id != CZ_NONE) S.tcur[tid] = id;
            __syncthreads();
            if (c > 0) {
                S.eval_todo(c);
                __syncthreads();
            }
            for (int k = tid; k < r.cap; k += kThreads) {
                if (k < c) {
                    r.ids[k] = s.nid[k];
                    r.dst[k] = key_dist(s.nkey[k]);
                } else {
                    r.ids[k] = CZ_NONE;
                }
            }
            if (tid == 0) {
                *r.deg = ((uint32_t)c + *r.hid_in) | ((uint32_t)*r.hid_in << 16);
                atomicAdd(ndist_total, (unsigned long long)c);
            }
            __syncthreads();
        }
    }
}

// final layout: rows sorted ascending by id, packed to the row width (one wave per row)
__global__ void __launch_bounds__(256)
build_pack_kernel(const uint32_t *__restrict__ src, int cap, uint32_t *__restrict__ dst, int width, uint64_t rows) {
    __shared__ uint32_t buf[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < rows; r += (uint64_t)gridDim.x * 4) {
        const uint32_t *in = src + r * cap;
        for (int j = lane; j < cap; j += 64) buf[wave][j] = in[j];
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < cap; j += 64) {
            const uint32_t v = buf[wave][j];
            int rank = 0;
            for (int i = 0; i < cap; i++) {
                const uint32_t o = buf[wave][i];
                rank += (o < v) || (o == v && i < j);
            }
            if (rank < width) dst[r * width + rank] = v;  // CZ_NONE sorts last
        }
        __builtin_amdgcn_wave_barrier();
    }
}

#define CZ_DISPATCH_BUILD_SHAPE(SH, CALL)                                      \
    do {                                                                       \
        if ((SH).lpv == 16) { CALL(16, 1, 4); }                                \
        else if ((SH).lpv == 32) { CALL(32, 1, 4); }                           \
        else switch ((SH).iters) {                                             \
            case 1: CALL(64, 1, 4); break;                                     \
            case 2: CALL(64, 2, 2); break;                                     \
            case 3: CALL(64, 3, 2); break;                                     \
            case 4: CALL(64, 4, 1); break;                                     \
            case 5: CALL(64, 5, 1); break;                                     \
            case 6: CALL(64, 6, 1); break;                                     \
            case 7: CALL(64, 7, 1); break;                                     \
            default: CALL(64, 8, 1); break;                                    \
        }                                                                      \
    } while (0)

struct ReqBuf {
    cz::DevBuf<uint32_t> t, q;
    cz::DevBuf<int32_t> lv;
    cz::DevBuf<double> d;
    hipError_t alloc(size_t n) {
        hipError_t e;
        if ((e = t.alloc(n)) != hipSuccess) return e;
        if ((e = q.alloc(n)) != hipSuccess) return e;
        if ((e = lv.alloc(n)) != hipSuccess) return e;
        return d.alloc(n);
    }
    Req ref() { return Req{t.p, q.p, lv.p, d.p}; }
};

}  // namespace

namespace {

// hnsw_put over `n_new` more vectors (runtime/hnsw.rs:679-727 -> :155-375) into `ix`, which holds n_old >= 0 nodes in the
// packed search layout.  The whole index goes into build form (rows with slack slots, link distances, degree counters),
// the new vectors are inserted batch by batch, and the tables are packed again.
int build_into(cz::HnswIndex *ix, const float *vectors, uint32_t n_new, uint32_t m, uint32_t ef_construction,
               int keep_pruned_connections, const int32_t *levels, uint64_t seed, uint32_t max_batch, uint64_t *n_dist_out,
               uint32_t flags, hipStream_t stream) {
    const uint32_t dim = ix->dim, ld = ix->ld, n_old = ix->n;
    const uint64_t n64 = (uint64_t)n_old + n_new;
    if (n64 >= 0x7FFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "node ids must be < 2^31");
    const uint32_t n = (uint32_t)n64;
    Shape sh = shape_of(dim);
    if (max_batch == 0) max_batch = 4096;
    // vectors: old rows kept, new rows appended
    {
        float *nv = nullptr;
        bool contiguous = false;
        const size_t table_bytes = std::max<size_t>(16, (size_t)n * ld * 4);
        CZ_HIP(cz::alloc_table((void **)&nv, table_bytes, &contiguous));
        std::unique_ptr<float, void (*)(float *)> guard(nv, [](float *p) { (void)hipFree(p); });
        if (n_old) CZ_HIP(hipMemcpyAsync(nv, ix->vec, (size_t)n_old * ld * 4, hipMemcpyDeviceToDevice, stream));
        float *dst = nv + (size_t)n_old * ld;
        if (flags & CZ_DEVICE_PTRS) {
            if (ld == dim) CZ_HIP(hipMemcpyAsync(dst, vectors, (size_t)n_new * dim * 4, hipMemcpyDeviceToDevice, stream));
            else {
                CZ_HIP(hipMemsetAsync(dst, 0, (size_t)n_new * ld * 4, stream));
                CZ_HIP(hipMemcpy2DAsync(dst, (size_t)ld * 4, vectors, (size_t)dim * 4, (size_t)dim * 4, n_new, hipMemcpyDeviceToDevice,
                                        stream));
            }
        } else {
            CZ_HIP(hipStreamSynchronize(stream));
            if (ld != dim) CZ_HIP(hipMemset(dst, 0, (size_t)n_new * ld * 4));
            CZ_HIP(hipMemcpy2D(dst, (size_t)ld * 4, vectors, (size_t)dim * 4, (size_t)dim * 4, n_new, hipMemcpyHostToDevice));
        }
        CZ_HIP(hipStreamSynchronize(stream));
        const bool had_old = ix->vec != nullptr;
        if (ix->vec) (void)hipFree(ix->vec);
        ix->vec = guard.release();
        if (had_old && !contiguous) contiguous = cz::rehome_table(&ix->vec, table_bytes, stream);  // the old table was in the way
        ix->table_contiguous = contiguous;
    }
    // levels: caller-supplied (non-negative = -layer) or drawn here: floor(-ln(U) / ln(m)), hnsw.rs:46-52.
    // Until the build has gone through, the handle must stay what it was: a failure further down (a bad level, an
    // allocation, a launch) would otherwise leave ix->n == n_old beside a level table of n entries, and every later
    // cz_hnsw_remove would refuse the handle ("level table out of step").  The vector buffer swapped above only grew.
    struct TopGuard {
        std::vector<int32_t> &top;
        size_t n_old;
        bool keep = false;
        ~TopGuard() {
            if (!keep) top.resize(n_old);
        }
    } top_guard{ix->top, n_old};
    ix->top.resize(n);
    if (levels) {
        for (uint32_t i = 0; i < n_new; i++) {
            if (levels[i] < 0 || levels[i] > 60) return cz::set_error(CZ_E_INVALID, "levels[%u] = %d out of range", i, levels[i]);
            ix->top[n_old + i] = levels[i];
        }
    } else {
        std::mt19937_64 rng(seed);
        std::uniform_real_distribution<double> uni(0.0, 1.0);
        const double mult = 1.0 / std::log((double)m);
        for (uint32_t i = 0; i < n_new; i++) {
            double u = uni(rng);
            if (u <= 0.0) u = 1e-300;
            ix->top[n_old + i] = (int32_t)std::min(60.0, std::floor(-std::log(u) * mult));
        }
    }
    // upper-level rows: one per (node, level >= 1), a node's rows consecutive; old nodes keep theirs (appended nodes come after)
    std::vector<uint32_t> base(n, CZ_NONE);
    uint64_t rows = 0, rows_old = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (i == n_old) rows_old = rows;
        if (ix->top[i] > 0) {
            base[i] = (uint32_t)rows;
            rows += (uint32_t)ix->top[i];
        }
    }
    if (n_old == n) rows_old = rows;
    if (rows >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "too many upper-level rows");
    const int old_w0 = ix->w0, old_wu = ix->wu;
    const int w0 = (int)(2 * m), wu = (int)m;
    const int slack = 32;
    const int cap0 = w0 + slack, capU = wu + slack;
    if (n_old && (old_w0 > cap0 || old_wu > capU))
        return cz::set_error(CZ_E_UNSUPPORTED, "the index has rows of %d / %d links, wider than m = %u allows", old_w0, old_wu, m);
    cz::DevBuf<uint32_t> b_nbr0, b_deg0, b_nbrU, b_degU, b_shrink_t, b_misc, b_visited, b_upbase;
    cz::DevBuf<double> b_dst0, b_dstU;
    cz::DevBuf<int32_t> b_level, b_shrink_lv;
    cz::DevBuf<unsigned long long> b_ndist;
    cz::DevBuf<uint8_t> b_ph0, b_phU;  // what earlier builds left hidden in the degrees (hnsw_index.h), for the unpack kernel
    CZ_HIP(b_ph0.alloc(std::max<size_t>(1, n)));
    CZ_HIP(b_phU.alloc(std::max<size_t>(1, rows)));
    CZ_HIP(hipMemsetAsync(b_ph0.p, 0, std::max<size_t>(1, n), stream));
    CZ_HIP(hipMemsetAsync(b_phU.p, 0, std::max<size_t>(1, rows), stream));
    if (!ix->ph0.empty())
        CZ_HIP(hipMemcpyAsync(b_ph0.p, ix->ph0.data(), std::min<size_t>(ix->ph0.size(), n_old), hipMemcpyHostToDevice, stream));
    if (!ix->phU.empty() && rows_old)
        CZ_HIP(hipMemcpyAsync(b_phU.p, ix->phU.data(), std::min<size_t>(ix->phU.size(), rows_old), hipMemcpyHostToDevice, stream));
    cz::DevBuf<uint32_t> b_rowof;  // the base row of every node's vector (cz_hnsw_set_row_of), or none: one vector per row
    if (!ix->row_of.empty()) {
        if (ix->row_of.size() < n)
            return cz::set_error(CZ_E_INVALID, "cz_hnsw_set_row_of covers %zu nodes, the index will hold %u", ix->row_of.size(), n);
        CZ_HIP(b_rowof.alloc(n));
        CZ_HIP(hipMemcpyAsync(b_rowof.p, ix->row_of.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    }
    CZ_HIP(b_nbr0.alloc((size_t)n * cap0));
    CZ_HIP(b_dst0.alloc((size_t)n * cap0));
    CZ_HIP(b_deg0.alloc(n));
    CZ_HIP(b_nbrU.alloc(std::max<size_t>(1, rows) * capU));
    CZ_HIP(b_dstU.alloc(std::max<size_t>(1, rows) * capU));
    CZ_HIP(b_degU.alloc(std::max<size_t>(1, rows)));
    CZ_HIP(b_level.alloc(n));
    CZ_HIP(b_misc.alloc(8));
    CZ_HIP(b_ndist.alloc(1));
    CZ_HIP(b_upbase.alloc(n));
    CZ_HIP(hipMemcpyAsync(b_upbase.p, base.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    CZ_HIP(hipMemcpyAsync(b_level.p, ix->top.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    CZ_HIP(hipMemsetAsync(b_nbr0.p, 0xFF, (size_t)n * cap0 * 4, stream));
    CZ_HIP(hipMemsetAsync(b_nbrU.p, 0xFF, std::max<size_t>(1, rows) * capU * 4, stream));
    CZ_HIP(hipMemsetAsync(b_deg0.p, 0, (size_t)n * 4, stream));
    CZ_HIP(hipMemsetAsync(b_degU.p, 0, std::max<size_t>(1, rows) * 4, stream));
    CZ_HIP(hipMemsetAsync(b_ndist.p, 0, 8, stream));
    // one visited set (hash table + overflow bitmap, hnsw_kernels.h VisitedDev) per resident workgroup
    const int slots = 1024;
    uint32_t hbits = 0, words = 0;
    cz::visited_shape(n, ef_construction, (uint32_t)std::max(cap0, capU), &hbits, &words);
    cz::DevBuf<uint32_t> b_vtab;
    CZ_HIP(b_vtab.alloc(hbits ? ((size_t)slots << hbits) : 4));
    if (hbits) CZ_HIP(hipMemsetAsync(b_vtab.p, 0xFF, ((size_t)slots << hbits) * 4, stream));
    CZ_HIP(b_visited.alloc((size_t)slots * words));
    CZ_HIP(hipMemsetAsync(b_visited.p, 0, (size_t)slots * words * 4, stream));
    // Lazy shrinking (batched builds only; CZ_BUILD_LAZY=0 turns it off): a row is re-selected when its slack slots are
    // used up, not every time a reverse link takes it past its final width, and once more at the end.  On dense data
    // the heuristic keeps almost every link, so the eager form re-selects a row on nearly every reverse link: at
    // 10M x 768 that was 68 k of the 83 k distance evaluations per inserted vector.  max_batch = 1 keeps the
    // reference's eager order (hnsw.rs:338-350) and with it the identical link tables.
    const char *lazy_env = getenv("CZ_BUILD_LAZY");
    const int lazy = max_batch > 1 && !(lazy_env && atoi(lazy_env) == 0);
    const size_t max_req = (size_t)max_batch * (size_t)(w0 + 8 * wu) + 1024;
    ReqBuf reqA, reqB;
    CZ_HIP(reqA.alloc(max_req));
    CZ_HIP(reqB.alloc(max_req));
    CZ_HIP(b_shrink_t.alloc(max_req));
    CZ_HIP(b_shrink_lv.alloc(max_req));

    BuildTables T{b_nbr0.p, b_dst0.p, b_deg0.p, b_ph0.p, w0, cap0, b_upbase.p, b_nbrU.p, b_dstU.p, b_degU.p, b_phU.p, wu, capU,
                  b_level.p, ix->row_of.empty() ? nullptr : b_rowof.p};
    IndexDev dev = ix->dev();
    dev.n = n;
    dev.nbr0 = b_nbr0.p;
    dev.w0 = cap0;  // rows are scanned at their build stride; unused slots hold CZ_NONE
    dev.up_base = b_upbase.p;
    dev.up_nbrs = b_nbrU.p;
    dev.wu = capU;
    const uint32_t efcap = (std::max<uint32_t>(ef_construction, (uint32_t)std::max(cap0, capU)) + 63) & ~63u;
    const uint32_t wcap = std::max<uint32_t>(efcap, (uint32_t)((std::max(cap0, capU) + 63) & ~63));
    const size_t smem = czh::smem_bytes(efcap, wcap, ld);
    if (smem > 160 * 1024) return cz::set_error(CZ_E_UNSUPPORTED, "dim/ef_construction need %zu bytes of LDS", smem);
    // extend_candidates: per resident workgroup the candidate scratch (every member of the found set -- or of a row --
    // brings at most a row of neighbours) and the copy of the found list; per shrink of a round its staged selection
    const int extend = (flags & CZ_HNSW_EXTEND_CANDIDATES) ? 1 : 0;
    uint32_t stage_rows = 0;  // rows the staging arrays hold: grown to the largest round of shrinks (extend_candidates only)
    cz::DevBuf<uint64_t> b_ext_key;
    cz::DevBuf<uint32_t> b_ext_id, b_stage_ids, b_stage_n;
    cz::DevBuf<double> b_stage_dst;
    cz::DevBuf<uint8_t> b_stage_self;
    ExtBuf ext{nullptr, nullptr, 0, 0};
    Stage stage{nullptr, nullptr, nullptr, nullptr, std::max(w0, wu)};
    if (extend) {
        const uint64_t members = std::max<uint64_t>(ef_construction, (uint64_t)std::max(cap0, capU));
        const uint64_t need = members * (uint64_t)(std::max(cap0, capU) + 1);
        uint64_t cap = 2;
        while (cap < need) cap <<= 1;
        ext.cap = (uint32_t)cap;
        ext.keep = efcap;
        CZ_HIP(b_ext_key.alloc((size_t)slots * (cap + efcap)));
        CZ_HIP(b_ext_id.alloc((size_t)slots * (cap + efcap)));
        ext.key = b_ext_key.p;
        ext.id = b_ext_id.p;
    }
    // the staging arrays of a round of `rows` shrinks (width ids + width distances + two counters per row: ~1 KB per row)
    auto stage_for = [&](uint32_t rows) -> int {
        if (rows <= stage_rows) return CZ_OK;
        uint32_t want = std::max<uint32_t>(rows, std::max<uint32_t>(4096, stage_rows + stage_rows / 2));
        CZ_HIP(hipStreamSynchronize(stream));  // (nothing of the old arrays is in flight any more)
        // A round is staged WHOLE (its shrinks must all see the rows as the round found them: applying it part by part is the
        // ADVICE r3 bug), so the final lazy round of a multi-million-row batched build asks for 12 * width + 5 bytes per row at once.
        // When that does not fit, say so -- with what would fit -- instead of a bare allocation failure.
        // CZ_BUILD_STAGE_CAP_BYTES caps it for the tests.
        {
            const size_t per_row = (size_t)stage.width * 12 + 5;
            size_t free_b = 0, total_b = 0, held = (size_t)stage_rows * per_row;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = ~size_t(0) >> 1, (void)hipGetLastError();
            size_t room = free_b + held;  // the old arrays are released by the allocations below
            room = room > (size_t(256) << 20) ? room - (size_t(256) << 20) : 0;
            if (const char *c = getenv("CZ_BUILD_STAGE_CAP_BYTES")) room = std::min<size_t>(room, (size_t)strtoull(c, nullptr, 10));
            if ((size_t)want * per_row > room) want = rows;  // no head room for growth: exactly this round
            if ((size_t)want * per_row > room)
                return cz::set_error(CZ_E_OOM, "extend_candidates: a round of %u shrinks needs %zu MiB of staging (%zu B per row), %zu MiB are free; "
                                               "a round is applied whole, so build or insert in smaller batches (max_batch) -- at most %zu rows per "
                                               "round fit", rows, ((size_t)rows * per_row) >> 20, per_row, room >> 20, room / per_row);
            b_stage_ids.reset();
            b_stage_dst.reset();
            b_stage_n.reset();
            b_stage_self.reset();
            stage_rows = 0;
        }
        CZ_HIP(b_stage_ids.alloc((size_t)want * stage.width));
        CZ_HIP(b_stage_dst.alloc((size_t)want * stage.width));
        CZ_HIP(b_stage_n.alloc(want));
        CZ_HIP(b_stage_self.alloc(want));
        stage.ids = b_stage_ids.p;
        stage.dst = b_stage_dst.p;
        stage.n = b_stage_n.p;
        stage.self = b_stage_self.p;
        stage_rows = want;
        return CZ_OK;
    };
    // CZ_BUILD_TRACE=1: wait after every stage and name it (a faulting kernel is the one after the last name printed)
    const bool trace = getenv("CZ_BUILD_TRACE") && atoi(getenv("CZ_BUILD_TRACE")) != 0;
    auto stage_done = [&](const char *what, uint32_t count) {
        if (!trace) return;
        (void)hipStreamSynchronize(stream);
        fprintf(stderr, "[build] %s (%u) done\n", what, count);
    };
    // one round of shrinks: rows [0, count) of the request arrays (count on the host, or read on the device when `count_dev`).
    // With extend_candidates a shrink reads OTHER rows, so the selections of the WHOLE round are staged and applied afterwards:
    // every shrink of a round sees the rows as the round found them, whatever the round's size (until round 5 the staging held
    // 65 536 requests and a larger round -- the final lazy round of a large batched build -- was applied part by part, a later
    // part reading rows an earlier one had rewritten: ADVICE r3).
    auto launch_shrinks = [&](const uint32_t *sh_t, const int32_t *sh_lv, uint32_t count, const uint32_t *count_dev) -> int {
        if (count == 0) return CZ_OK;
        if (extend) {
            int src = stage_for(count);
            if (src) return src;
        }
        const uint32_t g3 = std::min<uint32_t>(count, (uint32_t)slots);
#define CZ_LAUNCH_SHRINK(LPV, ITERS, U)                                                                                  \
    do {                                                                                                                 \
        auto kern = extend ? build_shrink_kernel<LPV, ITERS, U, true> : build_shrink_kernel<LPV, ITERS, U, false>;      \
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                                        (int)smem);                                                     \
        hipLaunchKernelGGL(kern, dim3(g3), dim3(kThreads), smem, stream, dev, T, sh_t, sh_lv, count, count_dev,          \
                           efcap, wcap, keep_pruned_connections, b_ndist.p, b_vtab.p, hbits, b_visited.p, words, ext,     \
                           stage);                                                                                       \
    } while (0)
        CZ_DISPATCH_BUILD_SHAPE(sh, CZ_LAUNCH_SHRINK);
#undef CZ_LAUNCH_SHRINK
        stage_done("shrink", count);
        if (extend)
            hipLaunchKernelGGL(build_apply_kernel, dim3(std::max<uint32_t>(1, std::min<uint32_t>((count + 3) / 4, 4096))),
                               dim3(256), 0, stream, T, sh_t, sh_lv, count, count_dev, stage);
        stage_done("apply", count);
        return CZ_OK;
    };

    int top = -1;
    uint32_t entry = 0;
    uint32_t i = n_old;
    if (n_old > 0 && ix->n_levels > 0) {
        // the existing rows, with their link distances, into the build tables
        top = ix->n_levels - 1;
        entry = ix->entry;
        const uint32_t g0 = std::min<uint32_t>(n_old, 4096);
#define CZ_LAUNCH_UNPACK(LPV, ITERS, U)                                                                                  \
    do {                                                                                                                 \
        auto kern = build_unpack_kernel<LPV, ITERS, U>;                                                                  \
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                                        (int)smem);                                                     \
        hipLaunchKernelGGL(kern, dim3(g0), dim3(kThreads), smem, stream, dev, T, ix->nbr0, old_w0, ix->up_base, ix->up_nbrs,     \
                           old_wu, n_old, efcap, wcap, b_ndist.p);                                                       \
    } while (0)
        CZ_DISPATCH_BUILD_SHAPE(sh, CZ_LAUNCH_UNPACK);
#undef CZ_LAUNCH_UNPACK
        CZ_HIP(hipStreamSynchronize(stream));
    } else if (n > 0) {
        // an empty index: the first vector is its own entry point (hnsw.rs:206-218 with no rows yet)
        uint32_t first = n_old;
        top = ix->top[first];
        entry = first;
        i = first + 1;
    }
    (void)rows_old;
    while (i < n) {
        // a vector that raises the top level is inserted alone and becomes the entry point (hnsw.rs:206-218)
        uint32_t bn = 1;
        if (ix->top[i] <= top) {
            const uint32_t want = std::min<uint32_t>(max_batch, std::max<uint32_t>(1, i / 4));
            while (bn < want && i + bn < n && ix->top[i + bn] <= top) bn++;
        }
        CZ_HIP(hipMemsetAsync(b_misc.p, 0, 32, stream));
        const uint32_t grid = std::min<uint32_t>(bn, (uint32_t)slots);
#define CZ_LAUNCH_INSERT(LPV, ITERS, U)                                                                                  \
    do {                                                                                                                 \
        auto kern = extend ? build_insert_kernel<LPV, ITERS, U, true> : build_insert_kernel<LPV, ITERS, U, false>;      \
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                                        (int)smem);                                                     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), smem, stream, dev, T, i, bn, top, entry,                     \
                           (int)ef_construction, efcap, wcap, keep_pruned_connections, b_vtab.p, hbits, b_visited.p, words, \
                           reqA.ref(),                                                                                   \
                           b_misc.p + 0, (uint32_t)max_req, b_ndist.p, ext, (extend && max_batch == 1) ? 1 : 0);          \
    } while (0)
        CZ_DISPATCH_BUILD_SHAPE(sh, CZ_LAUNCH_INSERT);
#undef CZ_LAUNCH_INSERT
        stage_done("insert", bn);
        uint32_t h[8];
        CZ_HIP(hipMemcpyAsync(h, b_misc.p, 32, hipMemcpyDeviceToHost, stream));
        CZ_HIP(hipStreamSynchronize(stream));
        uint32_t nreq = h[0];
        if (nreq > max_req) return cz::set_error(CZ_E_HIP, "internal: request buffer overflow (%u > %zu)", nreq, max_req);
        ReqBuf *cur = &reqA, *nxt = &reqB;
        int rounds = 0;
        if (extend && max_batch == 1) {
            // The reference's order (:280-358): one neighbour after the other gets its reverse link and, if that takes it
            // past the row width, its shrink -- which with extend_candidates reads the rows of the neighbours still to
            // come.  No host round trip in between: the shrink and the apply kernel read the count the link kernel left.
            for (uint32_t k = 0; k < nreq; k++) {
                CZ_HIP(hipMemsetAsync(b_misc.p + 1, 0, 8, stream));
                const Req one{cur->t.p + k, cur->q.p + k, cur->lv.p + k, cur->d.p + k};
                hipLaunchKernelGGL(build_link_kernel, dim3(1), dim3(64), 0, stream, T, one, 1u, 0, nxt->ref(), b_misc.p + 1,
                                   b_shrink_t.p, b_shrink_lv.p, b_misc.p + 2, 1, 0);
                if (int src = launch_shrinks(b_shrink_t.p, b_shrink_lv.p, 1u, b_misc.p + 2)) return src;
            }
            nreq = 0;
        }
        while (nreq > 0) {
            CZ_HIP(hipMemsetAsync(b_misc.p + 1, 0, 8, stream));  // [1] retry count, [2] shrink count
            hipLaunchKernelGGL(build_link_kernel, dim3(std::max<uint32_t>(1, std::min<uint32_t>((nreq + 255) / 256, 4096))),
                               dim3(256), 0, stream, T, cur->ref(), nreq, lazy, nxt->ref(), b_misc.p + 1, b_shrink_t.p,
                               b_shrink_lv.p, b_misc.p + 2, 0, extend);
            stage_done("link", nreq);
            CZ_HIP(hipMemcpyAsync(h, b_misc.p, 32, hipMemcpyDeviceToHost, stream));
            CZ_HIP(hipStreamSynchronize(stream));
            const uint32_t nretry = h[1], nshrink = h[2];
            if (nshrink > 0)
                if (int src = launch_shrinks(b_shrink_t.p, b_shrink_lv.p, nshrink, nullptr)) return src;
            std::swap(cur, nxt);
            if (nretry >= nreq && nshrink == 0)
                return cz::set_error(CZ_E_HIP, "internal: reverse-link requests made no progress");
            nreq = nretry;
            if (++rounds > 64) return cz::set_error(CZ_E_HIP, "internal: reverse-link retry did not converge");
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "hnsw build launch: %s", hipGetErrorString(e));
        if (ix->top[i] > top) {
            top = ix->top[i];
            entry = i;
        } else {  // a row on the top layer whose KEY sorts before the entry point's becomes the first row of the index
            for (uint32_t k = i; k < i + bn; k++)
                if (ix->top[k] == top && ix->key_before(k, entry)) entry = k;
        }
        i += bn;
    }
    if (lazy || n_old > 0) {  // rows that still hold more links than their final width
        cz::DevBuf<uint32_t> f_t;
        cz::DevBuf<int32_t> f_lv;
        const size_t fcap = (size_t)n + rows;
        CZ_HIP(f_t.alloc(fcap));
        CZ_HIP(f_lv.alloc(fcap));
        CZ_HIP(hipMemsetAsync(b_misc.p + 2, 0, 4, stream));
        hipLaunchKernelGGL(build_overfull_kernel, dim3(4096), dim3(256), 0, stream, T, n, f_t.p, f_lv.p, b_misc.p + 2,
                           (uint32_t)std::min<size_t>(fcap, 0xFFFFFFFFu));
        uint32_t nfinal = 0;
        CZ_HIP(hipMemcpyAsync(&nfinal, b_misc.p + 2, 4, hipMemcpyDeviceToHost, stream));
        CZ_HIP(hipStreamSynchronize(stream));
        if (nfinal > 0) {
            if (int src = launch_shrinks(f_t.p, f_lv.p, nfinal, nullptr)) return src;
            CZ_HIP(hipStreamSynchronize(stream));  // f_t / f_lv die with this scope
        }
        hipError_t fe = hipGetLastError();
        if (fe != hipSuccess) return cz::set_error(CZ_E_HIP, "hnsw build final shrink: %s", hipGetErrorString(fe));
    }
    // final layout
    uint32_t *new_nbr0 = nullptr, *new_up = nullptr;
    CZ_HIP(hipMalloc((void **)&new_nbr0, (size_t)n * w0 * 4));
    if (hipMalloc((void **)&new_up, std::max<size_t>(1, rows) * wu * 4) != hipSuccess) {
        (void)hipFree(new_nbr0);
        return cz::set_error(CZ_E_OOM, "out of device memory for the packed link tables");
    }
    CZ_HIP(hipMemsetAsync(new_up, 0xFF, std::max<size_t>(1, rows) * wu * 4, stream));
    hipLaunchKernelGGL(build_pack_kernel, dim3(4096), dim3(256), 0, stream, b_nbr0.p, cap0, new_nbr0, w0, (uint64_t)n);
    if (rows) hipLaunchKernelGGL(build_pack_kernel, dim3(1024), dim3(256), 0, stream, b_nbrU.p, capU, new_up, wu, rows);
    unsigned long long nd = 0;
    CZ_HIP(hipMemcpyAsync(&nd, b_ndist.p, 8, hipMemcpyDeviceToHost, stream));
    CZ_HIP(hipStreamSynchronize(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(new_nbr0);
        (void)hipFree(new_up);
        return cz::set_error(CZ_E_HIP, "hnsw pack launch: %s", hipGetErrorString(e));
    }
    if (ix->nbr0) (void)hipFree(ix->nbr0);
    if (ix->up_nbrs) (void)hipFree(ix->up_nbrs);
    if (ix->up_base) (void)hipFree(ix->up_base);
    ix->nbr0 = new_nbr0;
    ix->up_nbrs = new_up;
    ix->up_base = b_upbase.release();
    ix->n = n;
    top_guard.keep = true;
    ix->w0 = w0;
    ix->wu = wu;
    ix->up_rows = rows;
    ix->n_levels = top + 1;
    ix->entry = entry;
    ix->layout_top.assign(ix->top.begin(), ix->top.end());
    if (extend || !ix->ph0.empty() || !ix->row_of.empty()) {
        // degrees that count links without a slot (the self link of an extended shrink, links inside a base row), for the next
        // insert and the write-back: the high half of every degree word
        std::vector<uint32_t> w0v(n), wUv((size_t)rows);
        CZ_HIP(hipMemcpy(w0v.data(), b_deg0.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (rows) CZ_HIP(hipMemcpy(wUv.data(), b_degU.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
        ix->ph0.assign(n, 0);
        ix->phU.assign((size_t)rows, 0);
        // (kept as one byte per row; a row that hides more than 255 links -- a base row carrying more than 256 indexed vectors that all
        // select each other -- is refused rather than written back with a degree that no longer matches what the build counted)
        uint32_t hid_max = 0;
        for (uint32_t v = 0; v < n; v++) hid_max = std::max(hid_max, hid_of(w0v[v]));
        for (uint64_t r = 0; r < rows; r++) hid_max = std::max(hid_max, hid_of(wUv[r]));
        if (hid_max > 255)
            return cz::set_error(CZ_E_UNSUPPORTED, "a row's degree counts %u links that have no slot (links inside one base row): at most 255 are kept",
                                 hid_max);
        for (uint32_t v = 0; v < n; v++) ix->ph0[v] = (uint8_t)hid_of(w0v[v]);
        for (uint64_t r = 0; r < rows; r++) ix->phU[r] = (uint8_t)hid_of(wUv[r]);
    }
    {   // the visited workspaces were sized for the old n
        std::lock_guard<std::mutex> lk(ix->mu);
        for (auto &w : ix->pool) cz::HnswIndex::destroy(w);
        ix->pool.clear();
    }
    if (n_dist_out) *n_dist_out = nd;
    return CZ_OK;
}

int check_build_args(uint32_t dim, int metric, uint32_t m, uint32_t ef_construction) {
    if (dim == 0) return cz::set_error(CZ_E_INVALID, "dim must be > 0");
    if (metric < CZ_L2 || metric > CZ_IP) return cz::set_error(CZ_E_INVALID, "bad metric %d", metric);
    if (m < 2) return cz::set_error(CZ_E_INVALID, "m must be >= 2");  // level_multiplier = 1/ln(m)
    if (2 * m > 192) return cz::set_error(CZ_E_UNSUPPORTED, "m = %u: m_max0 = 2m must be <= 192 for the GPU build", m);
    if (ef_construction == 0) return cz::set_error(CZ_E_INVALID, "ef_construction must be > 0");
    // (no other limit than the LDS the lists take, checked where they are sized: ~4 400 entries next to a 768-d vector; round 3: 1 024)
    Shape sh = shape_of(dim);
    if (sh.lpv == 64 && sh.iters > 8) return cz::set_error(CZ_E_UNSUPPORTED, "GPU index construction supports dim <= 2048");
    return CZ_OK;
}

// drop every link to a removed node, keep the rows' order (one wave per row)
__global__ void __launch_bounds__(256)
remove_links_kernel(uint32_t *__restrict__ tab, int width, uint64_t rows, const uint32_t *__restrict__ removed_bits,
                    const uint32_t *__restrict__ row_owner /* level 0: nullptr (row = node) */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < rows; r += (uint64_t)gridDim.x * 4) {
        uint32_t *row = tab + r * width;
        const uint32_t owner = row_owner ? row_owner[r] : (uint32_t)r;
        const bool dead_row = owner != CZ_NONE && ((removed_bits[owner >> 5] >> (owner & 31)) & 1u);
        int kept = 0;
        for (int c0 = 0; c0 < width; c0 += 64) {
            const int c = c0 + lane;
            const uint32_t id = c < width ? row[c] : CZ_NONE;
            const bool keep = !dead_row && id != CZ_NONE && !((removed_bits[id >> 5] >> (id & 31)) & 1u);
            const unsigned long long mk = __ballot(keep);
            __builtin_amdgcn_wave_barrier();
            if (keep) row[kept + __popcll(mk & ((1ull << lane) - 1ull))] = id;  // never ahead of the slots read so far
            kept += __popcll(mk);
        }
        for (int c = kept + lane; c < width; c += 64) row[c] = CZ_NONE;
    }
}

}  // namespace

extern "C" int cz_hnsw_build(const float *vectors, uint32_t n, uint32_t dim, int metric, uint32_t m,
                             uint32_t ef_construction, int keep_pruned_connections, const int32_t *levels, uint64_t seed,
                             uint32_t max_batch, uint64_t *n_dist_out, cz_hnsw_index **out, uint32_t flags,
                             void *stream_) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    if (n_dist_out) *n_dist_out = 0;
    int rc = cz::ensure_device();
    if (rc) return rc;
    if ((rc = check_build_args(dim, metric, m, ef_construction))) return rc;
    if (n >= 0x7FFFFFFFu) return cz::set_error(CZ_E_UNSUPPORTED, "node ids must be < 2^31");
    if (n > 0 && !vectors) return cz::set_error(CZ_E_INVALID, "vectors is null");
    std::unique_ptr<cz::HnswIndex> ix(new cz::HnswIndex());
    ix->n = 0;
    ix->dim = dim;
    ix->ld = (dim + 3) & ~3u;
    ix->metric = metric;
    ix->w0 = (int)(2 * m);
    ix->wu = (int)m;
    ix->n_levels = 0;
    if (n == 0) {
        CZ_HIP(hipMalloc((void **)&ix->vec, 16));
        *out = reinterpret_cast<cz_hnsw_index *>(ix.release());
        return CZ_OK;
    }
    rc = build_into(ix.get(), vectors, n, m, ef_construction, keep_pruned_connections, levels, seed, max_batch, n_dist_out, flags,
                    (hipStream_t)stream_);
    if (rc) return rc;
    // the build's scratch is gone: now the arrays a search reads can be given their place (hnsw_api.hip, settle_placement).
    // cz_hnsw_insert does not do this by itself -- a caller that inserts in a loop settles once at the end (cz_hnsw_index_settle).
    if ((rc = cz::settle_if_large(ix.get(), (hipStream_t)stream_))) return rc;
    *out = reinterpret_cast<cz_hnsw_index *>(ix.release());
    return CZ_OK;
}

extern "C" int cz_hnsw_insert(cz_hnsw_index *h, const float *vectors, uint32_t n_new, uint32_t m, uint32_t ef_construction,
                              int keep_pruned_connections, const int32_t *levels, uint64_t seed, uint32_t max_batch,
                              uint64_t *n_dist_out, uint32_t flags, void *stream_) {
    if (n_dist_out) *n_dist_out = 0;
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    int rc = cz::ensure_device();
    if (rc) return rc;
    auto *ix = reinterpret_cast<cz::HnswIndex *>(h);
    if (ix->f64()) return cz::set_error(CZ_E_UNSUPPORTED, "the index holds F64 vectors: it is searched on the device, maintained on the CPU path");
    if ((rc = check_build_args(ix->dim, ix->metric, m, ef_construction))) return rc;
    if (n_new == 0) return CZ_OK;
    if (!vectors) return cz::set_error(CZ_E_INVALID, "vectors is null");
    return build_into(ix, vectors, n_new, m, ef_construction, keep_pruned_connections, levels, seed, max_batch, n_dist_out, flags,
                      (hipStream_t)stream_);
}

extern "C" int cz_hnsw_set_key_order(cz_hnsw_index *h, const uint32_t *rank, uint32_t n) {
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    cz::HnswIndex *ix = (cz::HnswIndex *)h;
    if (!rank || n == 0) {
        ix->key_rank.clear();
        return CZ_OK;
    }
    if (n < ix->n) return cz::set_error(CZ_E_INVALID, "key order for %u nodes, the index holds %u", n, ix->n);
    ix->key_rank.assign(rank, rank + n);
    return CZ_OK;
}

extern "C" int cz_hnsw_set_row_of(cz_hnsw_index *h, const uint32_t *row_of, uint32_t n) {
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    cz::HnswIndex *ix = (cz::HnswIndex *)h;
    if (!row_of || n == 0) {
        ix->row_of.clear();
        return CZ_OK;
    }
    if (n < ix->n) return cz::set_error(CZ_E_INVALID, "base rows for %u nodes, the index holds %u", n, ix->n);
    ix->row_of.assign(row_of, row_of + n);
    return CZ_OK;
}

extern "C" int cz_hnsw_remove(cz_hnsw_index *h, const uint32_t *nodes, uint32_t n_nodes) {
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    int rc = cz::ensure_device();
    if (rc) return rc;
    auto *ix = reinterpret_cast<cz::HnswIndex *>(h);
    if (ix->f64()) return cz::set_error(CZ_E_UNSUPPORTED, "the index holds F64 vectors: it is searched on the device, maintained on the CPU path");
    if (n_nodes == 0 || ix->n == 0) return CZ_OK;
    if (!nodes) return cz::set_error(CZ_E_INVALID, "null nodes");
    if (ix->top.size() != ix->n || ix->layout_top.size() != ix->n) return cz::set_error(CZ_E_HIP, "internal: level table out of step");
    std::vector<uint32_t> bits((ix->n + 31) / 32, 0);
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (nodes[i] >= ix->n) return cz::set_error(CZ_E_INVALID, "node %u out of range", nodes[i]);
        bits[nodes[i] >> 5] |= 1u << (nodes[i] & 31);
    }
    // owner of every upper-level row (for the rows of the removed nodes themselves)
    std::vector<uint32_t> owner((size_t)std::max<uint64_t>(1, ix->up_rows), CZ_NONE);
    {
        uint64_t r = 0;
        for (uint32_t i = 0; i < ix->n; i++)
            for (int l = 0; l < ix->layout_top[i]; l++) owner[(size_t)r++] = i;
    }
    cz::DevBuf<uint32_t> d_bits, d_owner;
    CZ_HIP(d_bits.alloc(bits.size()));
    CZ_HIP(d_owner.alloc(owner.size()));
    CZ_HIP(hipMemcpy(d_bits.p, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
    CZ_HIP(hipMemcpy(d_owner.p, owner.data(), owner.size() * 4, hipMemcpyHostToDevice));
    if (ix->n_levels > 0) {
        hipLaunchKernelGGL(remove_links_kernel, dim3(4096), dim3(256), 0, nullptr, ix->nbr0, ix->w0, (uint64_t)ix->n, d_bits.p,
                           (const uint32_t *)nullptr);
        if (ix->up_rows)
            hipLaunchKernelGGL(remove_links_kernel, dim3(1024), dim3(256), 0, nullptr, ix->up_nbrs, ix->wu, ix->up_rows, d_bits.p,
                               d_owner.p);
        CZ_HIP(hipDeviceSynchronize());
    }
    // the removed nodes leave every level (their upper rows stay allocated, empty); the entry point is positional: the
    // smallest node on the highest level that still has one (hnsw.rs:184-191, 891-899)
    for (uint32_t i = 0; i < n_nodes; i++) ix->top[nodes[i]] = -1;
    int top = -1;
    uint32_t entry = CZ_NONE;
    for (uint32_t i = 0; i < ix->n; i++)
        if (ix->top[i] > top || (ix->top[i] == top && top >= 0 && ix->key_before(i, entry))) {
            top = ix->top[i];
            entry = i;
        }
    ix->n_levels = top + 1;
    ix->entry = top >= 0 ? entry : CZ_NONE;
    return CZ_OK;
}

// ---- export of a device-resident index back to the flat host layout (cz_hnsw_desc) ----
extern "C" int cz_hnsw_index_info(const cz_hnsw_index *h, uint32_t *n, uint32_t *dim, int32_t *metric, int32_t *n_levels,
                                  uint32_t *entry) {
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    auto *ix = reinterpret_cast<const cz::HnswIndex *>(h);
    if (n) *n = ix->n;
    if (dim) *dim = ix->dim;
    if (metric) *metric = ix->metric;
    if (n_levels) *n_levels = ix->n_levels;
    if (entry) *entry = ix->entry;
    return CZ_OK;
}

extern "C" int cz_hnsw_index_level_info(const cz_hnsw_index *h, int32_t level, uint32_t *size, int32_t *width) {
    if (!h) return cz::set_error(CZ_E_INVALID, "null index");
    auto *ix = reinterpret_cast<const cz::HnswIndex *>(h);
    if (level < 0 || level >= ix->n_levels) return cz::set_error(CZ_E_INVALID, "level %d out of range", level);
    uint32_t c = 0;
    if (level == 0) c = ix->n;
    else
        for (uint32_t i = 0; i < ix->n; i++) c += ix->top[i] >= level;
    if (size) *size = c;
    if (width) *width = level == 0 ? ix->w0 : ix->wu;
    return CZ_OK;
}

extern "C" int cz_hnsw_index_export_level(const cz_hnsw_index *h, int32_t level, uint32_t *node_ids, uint32_t *nbrs) {
    if (!h || !nbrs) return cz::set_error(CZ_E_INVALID, "null argument");
    int rc = cz::ensure_device();
    if (rc) return rc;
    auto *ix = reinterpret_cast<const cz::HnswIndex *>(h);
    if (level < 0 || level >= ix->n_levels) return cz::set_error(CZ_E_INVALID, "level %d out of range", level);
    if (level == 0) {
        if (node_ids)
            for (uint32_t i = 0; i < ix->n; i++) node_ids[i] = i;
        CZ_HIP(hipMemcpy(nbrs, ix->nbr0, (size_t)ix->n * ix->w0 * 4, hipMemcpyDeviceToHost));
        return CZ_OK;
    }
    std::vector<uint32_t> base(ix->n);
    CZ_HIP(hipMemcpy(base.data(), ix->up_base, (size_t)ix->n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> up((size_t)std::max<uint64_t>(1, ix->up_rows) * ix->wu);
    CZ_HIP(hipMemcpy(up.data(), ix->up_nbrs, up.size() * 4, hipMemcpyDeviceToHost));
    uint32_t r = 0;
    for (uint32_t i = 0; i < ix->n; i++) {
        if (ix->top[i] < level) continue;
        if (node_ids) node_ids[r] = i;
        memcpy(nbrs + (size_t)r * ix->wu, &up[((size_t)base[i] + (level - 1)) * ix->wu], (size_t)ix->wu * 4);
        r++;
    }
    return CZ_OK;
}

extern "C" int cz_hnsw_index_export_degrees(const cz_hnsw_index *h, int32_t level, double *degree) {
    if (!h || !degree) return cz::set_error(CZ_E_INVALID, "null argument");
    auto *ix = reinterpret_cast<const cz::HnswIndex *>(h);
    if (level < 0 || level >= ix->n_levels) return cz::set_error(CZ_E_INVALID, "level %d out of range", level);
    uint32_t size = 0;
    int32_t width = 0;
    int rc = cz_hnsw_index_level_info(h, level, &size, &width);
    if (rc) return rc;
    std::vector<uint32_t> ids(std::max<uint32_t>(1, size)), tab((size_t)std::max<uint32_t>(1, size) * width);
    if ((rc = cz_hnsw_index_export_level(h, level, ids.data(), tab.data()))) return rc;
    std::vector<uint32_t> base;
    if (level > 0 && !ix->phU.empty()) {
        base.resize(ix->n);
        CZ_HIP(hipMemcpy(base.data(), ix->up_base, (size_t)ix->n * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t r = 0; r < size; r++) {
        uint32_t live = 0;
        for (int k = 0; k < width; k++) live += tab[(size_t)r * width + k] != CZ_NONE;
        uint32_t self = 0;
        if (level == 0) self = ids[r] < ix->ph0.size() ? ix->ph0[ids[r]] : 0;
        else if (!base.empty()) {
            const size_t row = (size_t)base[ids[r]] + (level - 1);
            self = row < ix->phU.size() ? ix->phU[row] : 0;
        }
        degree[r] = (double)(live + self);
    }
    return CZ_OK;
}

extern "C" int cz_hnsw_index_export_vectors(const cz_hnsw_index *h, float *out) {
    if (!h || !out) return cz::set_error(CZ_E_INVALID, "null argument");
    int rc = cz::ensure_device();
    if (rc) return rc;
    auto *ix = reinterpret_cast<const cz::HnswIndex *>(h);
    if (ix->f64()) return cz::set_error(CZ_E_UNSUPPORTED, "the index holds F64 vectors");
    if (ix->n == 0) return CZ_OK;
    CZ_HIP(hipMemcpy2D(out, (size_t)ix->dim * 4, ix->vec, (size_t)ix->ld * 4, (size_t)ix->dim * 4, ix->n, hipMemcpyDeviceToHost));
    return CZ_OK;
}
