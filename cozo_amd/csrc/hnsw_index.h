// hnsw_index.h -- device-resident HNSW index handle (the object behind cz_hnsw_index*).
//
// HBM layout (sized for 288 GB; everything a search touches is contiguous per node):
//   vec     f32 [n][ld]        ld = dim rounded up to 4 floats, zero padded, rows 16-byte aligned
//   nbr0    u32 [n][w0]        level-0 link rows (w0 = m_max0), ascending ids, CZ_NONE padded
//   up_base u32 [n]            first upper-level row of a node, CZ_NONE if the node lives on level 0 only
//   up_nbrs u32 [rows][wu]     rows of one node are consecutive: level 1, 2, ... top  (wu = m_max)
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace czh {
struct IndexDev;
struct PredSet;
}

namespace cz {

struct HnswIndex {
    uint32_t n = 0, dim = 0, ld = 0;
    int metric = 0;
    int n_levels = 0;
    uint32_t entry = CZ_NONE;
    int w0 = 1, wu = 1;
    uint64_t up_rows = 0;
    float *vec = nullptr;
    bool table_contiguous = false;  // the vector table sits in one physically contiguous range (cz::alloc_table)
    double *vec64 = nullptr;  // an F64 index (cz_hnsw_index_create_f64): vec is null, ld = dim rounded up to 2; search only
    uint32_t *nbr0 = nullptr;
    uint32_t *up_base = nullptr;
    uint32_t *up_nbrs = nullptr;
    std::vector<int32_t> top;         // host copy: top level of every node (-1: removed, cz_hnsw_remove)
    std::vector<int32_t> layout_top;  // the same when the upper-level rows were last laid out (row numbering)
    // Position of every node's KEY among all keys (cz_hnsw_set_key_order); empty = node ids are in key order, which is what
    // an index read from the store or built over rows in key order has.  The reference's entry point is positional -- the
    // first row of the index relation = the smallest KEY on the top layer (hnsw.rs:184-191, 891-899) -- so rows inserted
    // later with keys that sort before existing ones need their rank, not their id, to be compared.
    std::vector<uint32_t> key_rank;
    // How far a row's stored degree lies above its number of visible link rows: the self link an extend_candidates shrink
    // selected and hnsw_put_vector overwrote with the self row again (hnsw.rs:413-433, 352-357), and links inside one base row,
    // which are written and counted but never read (:609-610).  ph0[node], phU[upper row]; empty = none.  Kept between builds
    // and inserts: the next reverse link onto such a row triggers its shrink earlier (:338-339), and the self row written
    // back carries the degree.
    std::vector<uint8_t> ph0, phU;
    // the base row every node's vector comes from (cz_hnsw_set_row_of); empty = one vector per row
    std::vector<uint32_t> row_of;
    bool key_before(uint32_t a, uint32_t b) const {
        return key_rank.empty() ? a < b : (key_rank[a] != key_rank[b] ? key_rank[a] < key_rank[b] : a < b);
    }

    // per-call scratch (the visited sets of a batch: hash tables + overflow bitmaps, hnsw_kernels.h VisitedDev):
    // cached, handed out under a mutex, stream-ordered by an event.  Invariant while pooled: every word of `tab` is
    // CZ_NONE and every word of `bitmap` 0 -- set once when the buffers are allocated, restored by the kernels that use
    // them (so a search never pays a per-launch memset).  A workspace whose kernel failed to launch is destroyed, not pooled.
    struct Workspace {
        void *tab = nullptr;
        size_t tab_bytes = 0;
        void *bitmap = nullptr;
        size_t bitmap_bytes = 0;
        hipEvent_t ready = nullptr;
    };
    std::mutex mu;
    std::vector<Workspace> pool;

    // what cz_hnsw_index_settle found: calibration launch before / after, candidate placements tried (0: never run)
    double settle_ms_before = 0.0, settle_ms_after = 0.0;
    uint32_t settle_trials = 0;

    bool f64() const { return vec64 != nullptr; }
    ~HnswIndex();
    int acquire(size_t tab_bytes, size_t bitmap_bytes, hipStream_t stream, Workspace *out);
    int release(Workspace w, hipStream_t stream);
    static void destroy(Workspace &w);
    czh::IndexDev dev() const;
};

// d_queries: [B][dim] floats -- doubles for an F64 index
int hnsw_search_device(HnswIndex *ix, const float *d_queries, uint32_t B, uint32_t k, uint32_t ef, int has_radius,
                       double radius, uint32_t *d_ids, double *d_dist, uint32_t *d_count, uint64_t *d_ndist,
                       hipStream_t stream, const czh::PredSet *preds = nullptr);

// shape of the per-query visited set for a traversal with list size `ef` over link rows of `width` slots on an index
// of n nodes: hash-table bits (0 = bitmap only) and bitmap words.  CZ_HNSW_VISITED = bitmap | hash and
// CZ_HNSW_VSLOTS = <slots> override the choice (experiments, and the test of the overflow path).
void visited_shape(uint32_t n, uint32_t ef, uint32_t width, uint32_t *hbits, uint32_t *words);

// cz_hnsw_index_settle on an index nobody else holds yet (create / build / insert call it on large indices)
int settle_placement(HnswIndex *ix, uint32_t ef, uint32_t trials, hipStream_t stream);
// the policy: tables of at least 1 GiB, 3 candidates per array (CZ_TABLE_SETTLE=0: never; =n: n candidates)
int settle_if_large(HnswIndex *ix, hipStream_t stream);

}  // namespace cz
