/*
 * cozo_gpu.h -- C ABI of libcozo_gpu.so: the MI355X (gfx950) implementation of CozoDB's
 * HNSW k-NN search and whole-graph fixed-rule hot path.
 *
 * This header is the drop-in boundary.  cozo-core (Rust) has no C-level hook for this path
 * (cozo-lib-c/cozo_c.h is string/JSON only), so each entry point states the reference
 * interface it replaces (paths relative to /root/reference/cozo-core/src/); the Rust-side
 * `extern "C"` declarations and the `impl FixedRule` / `HnswSearchRA::iter` patches that bind
 * them are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (CZ_OK) or a negative cz_status; a human-readable message for the
 *     last failure on the calling thread is available from cz_last_error().
 *   - the caller owns every pointer it passes in and every output buffer; the library owns device
 *     memory behind the opaque handles.  No callbacks into the caller.
 *   - `flags & CZ_DEVICE_PTRS`: the array arguments marked [dev-able] are device pointers (already
 *     resident in HBM) and `stream` (a hipStream_t, may be NULL) orders the work; otherwise they are
 *     host pointers and the call is synchronous (H2D, kernels, D2H).
 *   - `poison` (may be NULL) mirrors runtime/db.rs:1926-1942 `Poison(Arc<AtomicBool>)`: a host byte
 *     polled between kernel launches; non-zero => CZ_E_CANCELLED.
 *   - all entry points are thread-safe; index / plan handles are immutable after creation.
 *   - cz_init(device) selects the GPU of the process; multi-GPU is either one process per GPU (cz_comm_create_rank)
 *     or one process driving several (cz_pagerank_multi) -- see the multi-GPU section below and INTEGRATION.md.
 *   - node ids are dense u32 (< 2^31); CZ_NONE pads id arrays.
 */
#ifndef COZO_GPU_H
#define COZO_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CZ_NONE 0xFFFFFFFFu
#define CZ_DEVICE_PTRS 1u
/* cz_hnsw_build / cz_hnsw_insert: HnswIndexManifest::extend_candidates (runtime/hnsw.rs:499-511) */
#define CZ_HNSW_EXTEND_CANDIDATES 256u
/* cz_pagerank_plan_create: force one of the three device formulations of the sweep (default: chosen from the
 * shard's shape; the environment variable CZ_PR_MODE = gather | blocked | accumulate overrides the default too) */
#define CZ_PR_GATHER 2u
#define CZ_PR_BLOCKED 4u
#define CZ_PR_ACCUMULATE 1024u /* needs ascending in-lists (CsrLayout::Sorted); refused with CZ_E_INVALID otherwise (the default choice falls back to CZ_PR_BLOCKED) */
/* cz_knn_bruteforce: compute the B x N dot products as one dense f32 GEMM on the matrix cores (Cosine / IP only;
 * every dot product is then a k-ordered fmaf chain instead of the search kernel's lane-parallel tree) */
#define CZ_BF_GEMM 8u
/* (16u was CZ_PR_RELAXED until round 3 -- reordered sums for long rows, out of north_star's 1e-5 on R-MAT.  Removed:
 * long rows are now summed in parallel AND in the reference's sequential f32 order, csrc/exact_sum.h.) */

typedef enum {
    CZ_OK = 0,
    CZ_E_INVALID = -1,     /* bad argument */
    CZ_E_NO_DEVICE = -2,   /* no gfx950 device / HIP runtime unusable: the product path never falls back to CPU */
    CZ_E_HIP = -3,         /* a HIP call failed */
    CZ_E_CANCELLED = -4,   /* *poison != 0 -> the shim raises ProcessKilled (runtime/db.rs:1932-1940) */
    CZ_E_OOM = -5,
    CZ_E_UNSUPPORTED = -6
} cz_status;

/* HnswDistance, parse/sys.rs:76-98 */
typedef enum { CZ_L2 = 0, CZ_COSINE = 1, CZ_IP = 2 } cz_metric;

/* ---- runtime ---- */
int cz_init(int device);              /* hipSetDevice + capability check (gfx950) */
void cz_shutdown(void);
int cz_device_count(void);
const char *cz_last_error(void);      /* thread-local */
const char *cz_version(void);
/* What this box's HBM delivers under the path's two access patterns (measurement aid; SURVEY 8d asks for measured ceilings
 * beside the nominal 8 TB/s): a contiguous non-temporal read (`stream_gbs`) and whole `row_bytes`-byte rows at pseudo-random
 * row numbers of `table` [rows][row_bytes] (a DEVICE pointer -- e.g. an index' vector table, so that the size, the allocation
 * and the translation footprint are the real ones; NULL = 4 GiB of the library's own), `n_fetch` rows per launch (0 = 4M),
 * `reps` timed launches each (0 = 5).  GB/s = 1e9 bytes per second of bytes requested. */
int cz_hbm_probe(const void *table, uint64_t rows, uint32_t row_bytes, uint64_t n_fetch, uint32_t reps, double *stream_gbs,
                 double *row_fetch_gbs);
/* The same for the traversal rules' access pattern: independent accesses to pseudo-random words of a per-node array of `n_words`
 * words of `word_bytes` (4 | 8) bytes (the library's own allocation), `n_access` per launch (0 = 256M), `reps` timed launches (0 = 3):
 * plain loads, and atomicMin without a returned value, in 1e9 accesses per second. */
int cz_random_access_probe(uint64_t n_words, uint32_t word_bytes, uint64_t n_access, uint32_t reps, double *loads_g_per_s,
                           double *atomic_min_g_per_s);
/* TEST HOOK: the wave-parallel form of a sequential f32 sum (csrc/exact_sum.h, what PageRank's long rows use) on arbitrary rows:
 * out[r] = init[r] + terms[row_off[r]] + terms[row_off[r] + 1] + ... added one after the other in f32, by a group of `lanes`
 * (16 | 64) lanes taking `per_lane` (4 | 8 | 16) terms each per pass.  Host pointers.  Exists so that the paths PageRank's
 * non-negative finite terms never reach (negative terms, inf / nan, denormal sums ...) are tested on the device. */
int cz_debug_seq_sum(const float *terms, const uint64_t *row_off, const float *init, uint32_t n_rows, int lanes, int per_lane,
                     float *out);
/* TEST HOOK: the PageRank plan build's own device primitives (csrc/sort_scan.h; no library sort / scan on the product path):
 * a STABLE sort of n (key, value) pairs by the low `bits` bits of the key (keys < 2^bits), and the exclusive scan of `vals`
 * (out_scan, optional).  Host pointers. */
int cz_debug_sort_pairs(const uint32_t *keys, const uint32_t *vals, uint64_t n, uint32_t bits, uint32_t *out_keys, uint32_t *out_vals,
                        uint32_t *out_scan);

/* =====================================================================================
 * Vectors / HNSW           replaces: runtime/hnsw.rs
 * ===================================================================================== */

/* Flat export of one `tbl:idx` index relation (schema runtime/relation.rs:1064-1126) plus the
 * indexed vectors of the base relation.  A "node" is one CompoundKey (row key, field, sub-index)
 * (runtime/hnsw.rs:55); the shim keeps the node -> CompoundKey table.  Level l here is layer -l of
 * the reference (level 0 = dense bottom layer).  Neighbour rows hold the live links of
 * hnsw_get_neighbours(include_deleted = false) (runtime/hnsw.rs:588-629): ascending node id,
 * self-loop row, same-row-key links and ignore_link rows already dropped, CZ_NONE padded.  */
typedef struct {
    uint32_t n;                 /* number of nodes = rows of `vectors`; level 0 holds all of them */
    uint32_t dim;               /* HnswIndexManifest::vec_dim (runtime/hnsw.rs:31) */
    int32_t metric;             /* cz_metric */
    int32_t n_levels;           /* 0 => empty index (hnsw_knn returns no rows, runtime/hnsw.rs:903-909) */
    uint32_t entry;             /* first row of the index relation = smallest key on the top layer (:891-899) */
    const uint32_t *level_size; /* [n_levels] nodes present on each level; level_size[0] == n */
    const int32_t *level_width; /* [n_levels] row width: m_max0 on level 0, m_max above (:243-247) */
    const uint32_t *const *level_nodes; /* [n_levels] ascending node ids per level; [0] may be NULL (identity) */
    const uint32_t *const *level_nbrs;  /* [n_levels] -> [level_size][level_width] */
} cz_hnsw_desc;

typedef struct cz_hnsw_index cz_hnsw_index;

/* Upload an index (host pointers).  vectors: f32 [n][dim] row-major (Vector::F32, data/value.rs:207-213). */
int cz_hnsw_index_create(const cz_hnsw_desc *desc, const float *vectors, cz_hnsw_index **out);
/* The same for an index of f64 vectors (VecElementType::F64, parse/sys.rs; Vector::F64, data/value.rs:207-213): vectors f64
 * [n][dim].  Such a handle is SEARCHED on the device (cz_hnsw_search_batch_f64 / _filtered_f64: VectorCache::dist's F64 arms,
 * runtime/hnsw.rs:73-78, 86-95, 102-106 -- every dot product and the final arithmetic in f64); cz_hnsw_insert / _remove /
 * cz_knn_bruteforce / cz_hnsw_index_export_vectors refuse it (CZ_E_UNSUPPORTED: F64 indices are built and maintained by the
 * reference's CPU path and uploaded from their stored rows). */
int cz_hnsw_index_create_f64(const cz_hnsw_desc *desc, const double *vectors, cz_hnsw_index **out);
void cz_hnsw_index_destroy(cz_hnsw_index *ix);
/* device bytes held by the index (vectors + link tables) */
uint64_t cz_hnsw_index_bytes(const cz_hnsw_index *ix);
/* 1 when the vector table (the VectorCache's rows, runtime/hnsw.rs:110-151, resident) sits in ONE physically contiguous range of
 * device memory, 0 when the allocator had none left and it is paged like any hipMalloc.  Placement moves the random whole-row
 * fetch rate of a 30 GB table by 2-8 %; tables >= 64 MB ask for a contiguous range first
 * (CZ_TABLE_CONTIGUOUS=0: never).  A diagnostic: results do not depend on it. */
int cz_hnsw_index_table_contiguous(const cz_hnsw_index *ix);
/* Placement by trial.  Where the vector table and the visited workspaces land in device memory moves the search kernel by up to
 * 16 % (same box, same binary: profiles/r05_landing.txt), nothing in the allocation API says where that is, and only the search
 * itself can tell.  So: time a calibration batch (1 024 rows of the table as queries, k = 10, `ef`; 0 = 128), give one array a
 * second place while the first is still held, time again, keep the faster; rounds of (a table candidate, a workspace candidate) until
 * the calibration launch reaches 0.72 of the nominal 8 TB/s by its algorithmic bytes (CZ_TABLE_SETTLE_TARGET; a landing that good is
 * not touched at all) or 3 x trials rounds are spent (needs room for a second copy of the table while it runs; skipped otherwise).
 * A caller that knows the ef of its queries settles with it: the landing that suits one list size does not always suit another.  cz_hnsw_index_create(_f64) and cz_hnsw_build do this by themselves for tables
 * of at least 1 GiB (CZ_TABLE_SETTLE=0: never, =n: n candidates; a 30 GB table: ~8 s); cz_hnsw_insert does not -- settle once after
 * a run of inserts.  trials = 0 only reports.  Not to be called while other threads search the handle.  Results never depend on it.
 *   ms_before / ms_after: the calibration launch before the first and after the last settle; n_tried: candidates timed. */
int cz_hnsw_index_settle(cz_hnsw_index *ix, uint32_t ef, uint32_t trials, double *ms_before, double *ms_after, uint32_t *n_tried);
/* the table's device address (placement experiments only) */
uint64_t cz_debug_index_table_address(const cz_hnsw_index *ix);
/* placement experiments only: one of the index' arrays re-allocated elsewhere, contents kept (what = 0 vector table, 1 level-0 links,
 * 2 upper-level tables, 3 drop the pooled visited workspaces; contiguous != 0: ask for a physically contiguous range) */
int cz_debug_index_rehome(cz_hnsw_index *ix, int what, int contiguous);
/* cz_hbm_probe over THIS index' vector table (its rows, its allocation): the ceiling the search kernels run under on this box */
int cz_hnsw_index_probe(const cz_hnsw_index *ix, uint64_t n_fetch, uint32_t reps, double *stream_gbs, double *row_fetch_gbs);

/* Index construction on the GPU: hnsw_put over all rows in key order (runtime/hnsw.rs:155-538, 679-727;
 * create_hnsw_index, runtime/relation.rs:1010-1201), as a batch-parallel insert.
 *   vectors [n][dim] f32 [dev-able]; m, ef_construction, keep_pruned_connections as HnswIndexManifest
 *   (m_max = m, m_max0 = 2m); flags & CZ_HNSW_EXTEND_CANDIDATES = HnswIndexManifest::extend_candidates (:499-511).
 *   levels [n] (host, may be NULL): level of every vector as the non-negative -layer of get_random_level
 *   (:46-52); NULL draws floor(-ln(U)/ln(m)) from `seed` (the reference uses an unseedable thread_rng).
 *   max_batch: vectors inserted concurrently (0 = default 4096; 1 reproduces the sequential algorithm and its
 *   link tables exactly).  n_dist (optional): distance evaluations spent.
 * The result is an ordinary index handle; cz_hnsw_index_export_* copies the link rows back so that the host
 * can store them as `tbl:idx` rows (store_tx.put, :277-357). */
int cz_hnsw_build(const float *vectors, uint32_t n, uint32_t dim, int metric, uint32_t m, uint32_t ef_construction,
                  int keep_pruned_connections, const int32_t *levels, uint64_t seed, uint32_t max_batch,
                  uint64_t *n_dist, cz_hnsw_index **out, uint32_t flags, void *stream);

/* Index maintenance on the device.
 * cz_hnsw_insert: hnsw_put for `n_new` more rows on a later write (query/stored.rs:431-450 -> runtime/hnsw.rs:679-727),
 *   batched like cz_hnsw_build: the new vectors become nodes n .. n + n_new - 1 of the same handle.  The existing link
 *   rows go back into build form first (their link distances -- the `dist` column hnsw_shrink_neighbour reads,
 *   :389-393 -- are evaluated again by the same arithmetic, bit for bit), the new vectors are inserted, the tables are
 *   packed again.  m / ef_construction / keep_pruned_connections as at build time; max_batch = 1 reproduces the
 *   sequential algorithm's tables.  The write-back of the changed rows is the shim's (cz_hnsw_index_export_level +
 *   czi_hnsw_encode_rows).  Not concurrent with searches on the same handle.
 * cz_hnsw_remove: hnsw_remove / hnsw_remove_vec (:728-868) for a set of nodes: the nodes leave every level, the links
 *   from and to them disappear, the entry point moves to the smallest node of the highest level that is left (the
 *   reference's entry point is positional, :184-191).  Node ids are not renumbered: a removed node keeps its id, holds no
 *   rows and can never be reached.  Unlike the reference, links INTO a removed node from nodes it did not link back to
 *   are dropped as well (the reference leaves those rows behind and a later search that follows one fails with
 *   "corrupted index"). */
int cz_hnsw_insert(cz_hnsw_index *ix, const float *vectors /* [n_new][dim], dev-able */, uint32_t n_new, uint32_t m,
                   uint32_t ef_construction, int keep_pruned_connections, const int32_t *levels, uint64_t seed,
                   uint32_t max_batch, uint64_t *n_dist, uint32_t flags, void *stream);
int cz_hnsw_remove(cz_hnsw_index *ix, const uint32_t *nodes, uint32_t n_nodes);
/* The reference's entry point is POSITIONAL: the first row of the index relation, i.e. the smallest KEY on the top layer
 * (hnsw.rs:184-191, 891-899).  Node ids follow key order when the index was read from the store or built over rows in key
 * order; rows inserted later get ids n, n+1, ... whatever their keys are.  rank[node] = the position of the node's
 * (row key, field, sub-index) among all of them, for the nodes the index holds AND the ones the next cz_hnsw_insert adds
 * (n >= both); cz_hnsw_insert and cz_hnsw_remove then choose the entry point by rank.  rank = NULL: ids are key order. */
int cz_hnsw_set_key_order(cz_hnsw_index *ix, const uint32_t *rank, uint32_t n);

/* Rows that carry several indexed vectors (a List of vectors, several vec_fields: runtime/hnsw.rs:694-706).
 * row_of[node] = the base row the node's vector comes from, for the nodes held and the ones the next cz_hnsw_insert adds
 * (NULL / 0: one vector per row).  hnsw_get_neighbours drops every link inside one base row (:609-610): it is written and
 * counted into both degrees (:281-357) and never read again -- not by the search, the extension, the shrink or the removal.
 * The device tables do not hold such links; the degrees count them (cz_hnsw_index_export_degrees).  To BUILD an index over
 * such rows: cz_hnsw_build with n = 0, cz_hnsw_set_row_of, cz_hnsw_insert. */
int cz_hnsw_set_row_of(cz_hnsw_index *ix, const uint32_t *row_of, uint32_t n);

/* flat export of a device-resident index (the inverse of cz_hnsw_index_create; host buffers) */
int cz_hnsw_index_info(const cz_hnsw_index *ix, uint32_t *n, uint32_t *dim, int32_t *metric, int32_t *n_levels,
                       uint32_t *entry);
int cz_hnsw_index_level_info(const cz_hnsw_index *ix, int32_t level, uint32_t *size, int32_t *width);
int cz_hnsw_index_export_level(const cz_hnsw_index *ix, int32_t level, uint32_t *node_ids /* [size] or NULL */,
                               uint32_t *nbrs /* [size][width] */);
int cz_hnsw_index_export_vectors(const cz_hnsw_index *ix, float *out /* [n][dim] */);
/* the f64 of every self row of a level (runtime/hnsw.rs:270, 338-357), in cz_hnsw_index_export_level's node order: the
 * number of visible link rows, plus one where an extend_candidates shrink selected the node itself (:413-433), plus the
 * links into the node's own base row that are counted and never read (:609-610) */
int cz_hnsw_index_export_degrees(const cz_hnsw_index *ix, int32_t level, double *degree /* [size] */);

/* SessionTx::hnsw_knn (runtime/hnsw.rs:869-1012) for a whole batch of parent tuples
 * (HnswSearchRA::iter, query/ra.rs:1085-1121, calls it once per tuple).
 *   queries [B][dim] f32 [dev-able]; k, ef as HnswSearch (data/program.rs:975-991);
 *   has_radius/radius: `distance > r => skip` (:952-956), applied to the same (squared-L2 / 1-cos /
 *   1-dot) value the reference compares.
 *   out_ids [B][k] (CZ_NONE padded), out_dist [B][k] f64 (the reference returns f64 distances),
 *   out_count [B]: results per query, ascending distance (:1005-1006)           [all dev-able]
 *   out_n_dist [B] or NULL: distance evaluations per query (for roofline accounting) [dev-able]
 * A filtered search (filter bytecode stays in Rust) asks for k = ef and filters the rows it gets
 * back, which is what :943-947/:997-1006 do. */
int cz_hnsw_search_batch(cz_hnsw_index *ix, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                         int has_radius, double radius, uint32_t *out_ids, double *out_dist, uint32_t *out_count,
                         uint64_t *out_n_dist, const volatile uint8_t *poison, uint32_t flags, void *stream);
/* queries [B][dim] f64 on an index created with cz_hnsw_index_create_f64.  The reference converts a query to the index' element type
 * before it searches (hnsw.rs:879-884: `x as f64` / `x as f32` per element): that cast is the caller's; either entry point
 * refuses an index of the other element type (CZ_E_INVALID). */
int cz_hnsw_search_batch_f64(cz_hnsw_index *ix, const double *queries, uint32_t B, uint32_t k, uint32_t ef, int has_radius,
                             double radius, uint32_t *out_ids, double *out_dist, uint32_t *out_count, uint64_t *out_n_dist,
                             const volatile uint8_t *poison, uint32_t flags, void *stream);

/* Filtered search with the filter on the device.  The reference keeps all ef candidates when a
 * filter is present, walks them in ascending distance -- radius cut, fetch the row, bind columns, filter bytecode -- and
 * truncates to k afterwards (runtime/hnsw.rs:943-947, 951-1006).  When the filter is a conjunction of comparisons of a
 * numeric column of the candidate's row with a constant, the shim uploads those columns once per index (one value per
 * node: cz_column_upload) and hands the comparisons over: the kernel's output stage evaluates them on the ef candidates
 * and only k rows per query leave the device instead of ef.  Comparison semantics are op_lt / op_le / op_eq / op_ge /
 * op_gt / op_neq of data/functions.rs:298-380: Int with Int as integers, Float with Float by total order, mixed as f64.
 * Anything else (Null / non-numeric values, other expressions) stays with the host path (k = ef, filter in Rust). */
typedef struct cz_column cz_column;
typedef enum { CZ_COL_F64 = 0, CZ_COL_I64 = 1 } cz_col_type;
typedef enum { CZ_OP_LT = 0, CZ_OP_LE = 1, CZ_OP_EQ = 2, CZ_OP_GE = 3, CZ_OP_GT = 4, CZ_OP_NE = 5 } cz_cmp_op;
typedef struct {
    const cz_column *column; /* [n nodes of the index] */
    int32_t op;              /* cz_cmp_op: column[node] OP constant */
    int32_t const_type;      /* cz_col_type of the constant */
    double f64_value;
    int64_t i64_value;
} cz_predicate;
/* values: f64 [n] or i64 [n] (host), one per node of the index the column will be used with */
int cz_column_upload(const void *values, uint32_t n, int32_t type, cz_column **out);
void cz_column_destroy(cz_column *c);
/* cz_hnsw_search_batch with 1..4 predicates (AND): out_ids / out_dist [B][k], the first k of the ef candidates that pass
 * the radius cut and every predicate, ascending distance; out_count [B]. */
int cz_hnsw_search_filtered(cz_hnsw_index *ix, const float *queries, uint32_t B, uint32_t k, uint32_t ef, int has_radius,
                            double radius, const cz_predicate *preds, uint32_t n_preds, uint32_t *out_ids, double *out_dist,
                            uint32_t *out_count, uint64_t *out_n_dist, const volatile uint8_t *poison, uint32_t flags,
                            void *stream);
int cz_hnsw_search_filtered_f64(cz_hnsw_index *ix, const double *queries, uint32_t B, uint32_t k, uint32_t ef, int has_radius,
                                double radius, const cz_predicate *preds, uint32_t n_preds, uint32_t *out_ids, double *out_dist,
                                uint32_t *out_count, uint64_t *out_n_dist, const volatile uint8_t *poison, uint32_t flags,
                                void *stream);

/* VectorCache::dist (runtime/hnsw.rs:66-109) == op_l2_dist / op_cos_dist / op_ip_dist
 * (data/functions.rs:2185-2255) over P (query, node) pairs:
 *   base [n][dim], queries [nq][dim], pairs [P][2] = (query row, base row), out [P] f64  [all dev-able] */
int cz_distance_batch(int metric, const float *base, uint32_t n, uint32_t dim, const float *queries, uint32_t nq,
                      const uint32_t *pairs, uint64_t P, double *out, uint32_t flags, void *stream);
/* The same over the vectors of an INDEX (VectorCache::dist is only ever called on index nodes): pairs [P][2] = (query row, node).
 * The base table is the index's resident, settled one (cz_hnsw_index_settle), not a bare array wherever the caller's allocation
 * landed -- the form the batched-distance roofline is quoted on.  Results are cz_distance_batch's, bit for bit.  [dev-able] */
int cz_hnsw_index_distance_batch(cz_hnsw_index *ix, const float *queries, uint32_t nq, const uint32_t *pairs, uint64_t P,
                                 double *out, uint32_t flags, void *stream);
/* the F64 arms of VectorCache::dist (hnsw.rs:73-78, 86-95, 102-106): base / queries f64, every dot product in f64 */
int cz_distance_batch_f64(int metric, const double *base, uint32_t n, uint32_t dim, const double *queries, uint32_t nq,
                          const uint32_t *pairs, uint64_t P, double *out, uint32_t flags, void *stream);

/* exact k-NN by exhaustive scan over an uploaded index' vectors (recall ground truth; the "query batch
 * turns distance into a dense GEMM" case).  Same outputs as cz_hnsw_search_batch.  flags: CZ_DEVICE_PTRS,
 * CZ_BF_GEMM (MFMA form, Cosine / IP). */
int cz_knn_bruteforce(cz_hnsw_index *ix, const float *queries, uint32_t B, uint32_t k, uint32_t *out_ids,
                      double *out_dist, uint32_t flags, void *stream);

/* =====================================================================================
 * Whole-graph fixed rules   replaces: fixed_rule/algos/ (one .rs per rule) behind `trait FixedRule`
 * (fixed_rule/mod.rs:538-567).  Graphs arrive as the CSR that
 * FixedRuleInputRelation::as_directed_graph builds (fixed_rule/mod.rs:136-200): dense u32 ids in
 * first-appearance order, CsrLayout::Sorted adjacency, parallel edges kept.
 * ===================================================================================== */

/* PageRank::run (fixed_rule/algos/pagerank.rs:29-56) -> graph::page_rank.
 *   in_offsets [N+1], in_sources [E] : in-adjacency (pull form)     out_degree [N]
 *   damping = `theta` as f32, tolerance = `epsilon` (f32 widened to f64), max_iter = `iterations`
 *   scores [N] f32 out (the shim emits `score as f64`), iters_run / final_err optional.
 * Host pointers only (one-shot); the resident / sharded form is the plan API below. */
int cz_pagerank(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                uint64_t E, float damping, double tolerance, uint32_t max_iter, float *scores, uint32_t *iters_run,
                double *final_err, const volatile uint8_t *poison);

/* The same rule under the OTHER reading of graph 0.3.1's loop (crate source absent from the reference tree): the
 * contribution of node u is refreshed INSIDE the sweep, right after its score, so nodes later in the sweep pull the new value --
 * on one rayon thread an ascending Gauss-Seidel sweep, which is the execution reproduced here bit for bit (oracle:
 * orc_pagerank_mode(ORC_PR_INPLACE)); with several threads the reference would depend on their schedule.  Level-scheduled on the
 * device (csrc/pagerank_inplace.hip).  Same arguments as cz_pagerank; flags: CZ_PR_ERR_F64_DIFF = the error term is
 * |f64(new) - f64(old)| instead of the f32 difference widened (scores do not depend on it); n_levels (optional): launches per sweep.
 * Which reading is the reference's is decided by tests/test_ref_fixtures.py on a box with cargo; until then the two readings
 * have equal standing (bench.py reports both, event-timed). */
#define CZ_PR_ERR_F64_DIFF 128u
int cz_pagerank_inplace(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N, uint64_t E,
                        float damping, double tolerance, uint32_t max_iter, uint32_t flags, float *scores, uint32_t *iters_run,
                        double *final_err, uint32_t *n_levels, const volatile uint8_t *poison);

/* The resident form of the in-place reading (round 6): the static layout of the level-scheduled sweep (csrc/inplace_plan.hpp:
 * level-major numbering, LDS-staged value streams, one hipGraph per sweep parity) is built once and kept in HBM.
 *   create      in_offsets [N+1], in_sources [E] ascending per row, out_degree [N]: host memory, or device memory with
 *               CZ_DEVICE_PTRS (the layout is built on the host either way); flags also takes CZ_PR_ERR_F64_DIFF.
 *               CZ_E_INVALID for lists that do not ascend / sources that are not nodes, CZ_E_UNSUPPORTED for a chain-like graph
 *               (more than CZ_PR_INPLACE_MAX_LEVELS = 4096 dependence levels).
 *   run         graph::page_rank's loop from the initial state: sweeps until err < tolerance or max_iter (pagerank.rs:47-50)
 *   init/sweeps the same loop in the caller's hands: init, then n sweeps on `stream` with nothing read back (what bench.py
 *               brackets with HIP events)
 *   read_scores scores [N] in the caller's numbering (host memory, or device memory with CZ_DEVICE_PTRS)
 *   A plan carries the sweep's state (contributions, value streams, parity): ONE thread drives it at a time; different plans are
 *   independent.  It belongs to the device that was current when it was created.
 *   info        shape [16] u64: levels, row blocks, phase-A items, long rows, urgent gap, slice width, launches per sweep, graph
 *               replay (1/0), X edges, Y edges, urgent edges, long-row edges, X positions, Y positions; host build / upload ms */
/* cz_pagerank_inplace_plan_create: lay the JACOBI reading out the same way (one level, every edge reads the previous sweep's
 * contribution; scores == cz_pagerank's, bit for bit): the grouped tile formulation as a sweep of two launches.  Measurement /
 * cross-check only; cz_pagerank and cz_pagerank_plan_* stay the product path of that reading. */
#define CZ_PR_INPLACE_AS_JACOBI 2048u
typedef struct cz_pagerank_inplace_plan cz_pagerank_inplace_plan;
int cz_pagerank_inplace_plan_create(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                                    uint64_t E, float damping, uint32_t flags, cz_pagerank_inplace_plan **out);
void cz_pagerank_inplace_plan_destroy(cz_pagerank_inplace_plan *p);
int cz_pagerank_inplace_plan_run(cz_pagerank_inplace_plan *p, double tolerance, uint32_t max_iter, uint32_t *iters_run,
                                 double *final_err, const volatile uint8_t *poison, void *stream);
int cz_pagerank_inplace_plan_init(cz_pagerank_inplace_plan *p, void *stream);
int cz_pagerank_inplace_plan_sweeps(cz_pagerank_inplace_plan *p, uint32_t n, void *stream);
int cz_pagerank_inplace_plan_read_scores(cz_pagerank_inplace_plan *p, float *scores, uint32_t flags, void *stream);
int cz_pagerank_inplace_plan_info(const cz_pagerank_inplace_plan *p, uint64_t *shape, double *build_ms, double *h2d_ms);

/* The same with the static device layout (CSR upload + blocked plan) kept between calls: `key_hi:key_lo` is the
 * caller's identity of (relation, snapshot) -- e.g. the stored relation's id and the transaction's snapshot; 0:0 = do
 * not cache.  On a hit with the same N, E, damping and flags the arrays are not read again (the key is the caller's
 * promise that they are unchanged), which takes a repeated `?[] <~ PageRank(*rel[])` from upload + plan + iterations
 * to iterations alone.  Up to CZ_PR_CACHE_PLANS (default 4) plans are kept, least recently used dropped first.
 * flags: CZ_PR_GATHER | CZ_PR_BLOCKED | CZ_PR_ACCUMULATE.  timing (optional): where the call's time went. */
typedef struct {
    double h2d_ms;        /* CSR upload (0 on a cache hit) */
    double plan_build_ms; /* static layout of the sweep (0 on a cache hit) */
    double iterate_ms;    /* init + iterations incl. the per-iteration error read-back */
    double d2h_ms;        /* scores back to the host */
    int32_t cache_hit;
    int32_t reserved;
} cz_pagerank_timing;
int cz_pagerank_cached(uint64_t key_hi, uint64_t key_lo, const uint32_t *in_offsets, const uint32_t *in_sources,
                       const uint32_t *out_degree, uint32_t N, uint64_t E, float damping, double tolerance,
                       uint32_t max_iter, uint32_t flags, float *scores, uint32_t *iters_run, double *final_err,
                       const volatile uint8_t *poison, cz_pagerank_timing *timing);
void cz_pagerank_cache_clear(void);

/* Resident / row-sharded PageRank.  A plan owns rows [row_begin, row_end) of the in-CSR:
 *   in_offsets [rows+1] relative to the shard (in_offsets[0] == 0), in_sources [E_local] GLOBAL ids,
 *   out_degree [N] for all nodes.                                   [dev-able]
 * One iteration = cz_pagerank_plan_step: reads the full contribution vector contrib_in [N] (device),
 * writes this shard's scores and its slice of contrib_out [N] (device, may alias a gathered buffer
 * the host then all-gathers over RCCL), and adds the shard's sum |new-old| into *err_out (device f64). */
typedef struct cz_pagerank_plan cz_pagerank_plan;
int cz_pagerank_plan_create(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree,
                            uint32_t N, uint32_t row_begin, uint32_t row_end, float damping,
                            cz_pagerank_plan **out, uint32_t flags /* CZ_DEVICE_PTRS: the three arrays are in HBM */);
void cz_pagerank_plan_destroy(cz_pagerank_plan *p);
/* contrib [N] device: init/out_degree for every node (graph::page_rank initial state); zeroes scores */
int cz_pagerank_plan_init(cz_pagerank_plan *p, float *contrib_dev, void *stream);
int cz_pagerank_plan_step(cz_pagerank_plan *p, const float *contrib_in_dev, float *contrib_out_dev,
                          double *err_out_dev, void *stream);
/* device pointer to this shard's scores [row_end-row_begin] */
float *cz_pagerank_plan_scores(cz_pagerank_plan *p);
uint64_t cz_pagerank_plan_edges(const cz_pagerank_plan *p);
/* 1 when the plan runs the source-blocked two-phase sweep with LDS tiles, 0 otherwise */
int cz_pagerank_plan_is_blocked(const cz_pagerank_plan *p);
/* which formulation the plan runs: 1 = CSR-stream gather, 2 = source-blocked two-phase sweep with tiles, 3 = two-phase
 * sweep with in-order accumulation (rows of the front part in groups, one wave each; csrc/pagerank.hip) */
int cz_pagerank_plan_formulation(const cz_pagerank_plan *p);
/* the plan's shape, for measurement scripts: {slices, slice width, groups, waves per workgroup, rows per group (LDS words),
 * accumulate workgroups, tile blocks, hub rows, pieces, in-edges of the groups' rows, value-stream positions, 0} */
int cz_pagerank_plan_shape(const cz_pagerank_plan *p, uint32_t *out12);
/* what creating the plan cost: CSR upload and static layout, milliseconds */
int cz_pagerank_plan_timing(const cz_pagerank_plan *p, double *h2d_ms, double *build_ms);
/* copy this shard's scores [row_end-row_begin] to `out` (host, or device with CZ_DEVICE_PTRS) */
int cz_pagerank_plan_read_scores(cz_pagerank_plan *p, float *out, uint32_t flags, void *stream);

uint32_t cz_pagerank_plan_nodes(const cz_pagerank_plan *p);

/* =====================================================================================
 * Multi-GPU, one node (RCCL over xGMI).  SURVEY section 8b proposed `cz_pagerank(..., int n_gpus, ...)`; a cozo
 * process is ONE process, so that form is cz_pagerank_multi below (one host thread and one RCCL communicator per GPU).
 * Launchers that run one process per GPU (MPI-style) use cz_comm_create_rank + cz_pagerank_sharded /
 * cz_hnsw_search_sharded.  RCCL is bound at run time (dlopen "librccl.so.1", or $COZO_RCCL_LIB): CZ_E_UNSUPPORTED when
 * the process cannot load it.
 * ===================================================================================== */
typedef struct cz_comm cz_comm;
#define CZ_UNIQUE_ID_BYTES 128
/* rank 0: a fresh RCCL unique id (ncclGetUniqueId); the caller ships the 128 bytes to the other ranks by its own means */
int cz_comm_unique_id(uint8_t *id /* [CZ_UNIQUE_ID_BYTES] */);
/* every rank, collectively: this process = rank `rank` of `world` on the device chosen by cz_init */
int cz_comm_create_rank(const uint8_t *id, int rank, int world, cz_comm **out);
void cz_comm_destroy(cz_comm *c);
int cz_comm_rank(const cz_comm *c);
int cz_comm_size(const cz_comm *c);
/* the two exchange steps of the path on device buffers, stream-ordered.  all_gather is in place: rank r's part sits at
 * buf + r * bytes_per_rank before the call, every part everywhere after it. */
int cz_comm_all_gather(cz_comm *c, void *buf_dev, uint64_t bytes_per_rank, void *stream);
int cz_comm_all_reduce_sum_f64(cz_comm *c, double *buf_dev, uint64_t n, void *stream);

/* cz_pagerank_sharded / cz_pagerank_multi: exchange the next contribution vector by an all-reduce(sum) of the full
 * vector with the other ranks' slices zeroed (north_star's literal wording) instead of the in-place all-gather of the
 * slices.  Same values bit for bit, ~2(world-1)/world * 4N bytes per link instead of 4N/world: a labelled comparison. */
#define CZ_PR_EXCHANGE_ALLREDUCE 32u
/* cz_pagerank_multi: every rank's rows as TWO plans cut in the middle; the exchange of the first part's contributions runs (on
 * a stream of its own) while the second part is swept -- cz_pagerank_sharded_overlapped per rank.  Scores unchanged bit for
 * bit; hides min(second-part sweep, first-part exchange) per iteration.  Ignored together with CZ_PR_EXCHANGE_ALLREDUCE. */
#define CZ_PR_OVERLAP_EXCHANGE 64u

/* graph::page_rank over row shards, collectively on every rank: `plan` owns this rank's rows
 * [rank * rows_per_rank, min(N, (rank + 1) * rows_per_rank)) (cz_pagerank_plan_create with those bounds).  Per iteration:
 * the plan's sweep, the exchange of the contribution slices, an all-reduce of {sum |new - old|, cancellation flag}.
 * Every rank stops at the same iteration: by the reference's rule on the reduced error, or with CZ_E_CANCELLED as soon
 * as ANY rank's poison flag is set.  The rank's scores stay in the plan (cz_pagerank_plan_read_scores). */
int cz_pagerank_sharded(cz_comm *comm, cz_pagerank_plan *plan, uint32_t rows_per_rank, double tolerance, uint32_t max_iter,
                        uint32_t flags /* CZ_PR_EXCHANGE_ALLREDUCE */, uint32_t *iters_run, double *final_err,
                        const volatile uint8_t *poison, void *stream);
/* The same loop with the rank's rows in two plans (rows [rb, rb + half_rows) and the rest; half_rows counts rows of the padded
 * per-rank range and is the same on every rank): sweep part 1 -> its pieces start travelling -> sweep part 2 meanwhile -> its
 * pieces -> join -> the f64 all-reduce.  Pieces land at their natural places of the full contribution vector (one grouped
 * ncclBroadcast per rank: a part's pieces sit rows_per_rank floats apart), so the scores equal cz_pagerank_sharded's. */
int cz_pagerank_sharded_overlapped(cz_comm *comm, cz_pagerank_plan *plan_first, cz_pagerank_plan *plan_second,
                                   uint32_t rows_per_rank, uint32_t half_rows, double tolerance, uint32_t max_iter,
                                   uint32_t *iters_run, double *final_err, const volatile uint8_t *poison, void *stream);
/* cz_pagerank on n_gpus devices of this process (devices 0 .. n_gpus-1): host CSR in, scores [N] out.
 * flags: CZ_PR_GATHER | CZ_PR_BLOCKED | CZ_PR_ACCUMULATE | CZ_PR_EXCHANGE_ALLREDUCE | CZ_PR_OVERLAP_EXCHANGE. */
int cz_pagerank_multi(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                      uint64_t E, float damping, double tolerance, uint32_t max_iter, int n_gpus, uint32_t flags,
                      float *scores, uint32_t *iters_run, double *final_err, const volatile uint8_t *poison);
/* The single-process forms (cz_pagerank_multi, cz_*_multi) keep their RCCL communicators between calls (creating a set costs ~0.5 s
 * even for one device); cz_shutdown releases them, or this on its own. */
void cz_comm_multi_shutdown(void);
/* hnsw_knn over an index partitioned into one independent sub-index per rank (BASELINE.json configs[3]), collectively:
 * rank 0's `queries_dev` [B][dim] are broadcast, every rank searches ITS shard with the same k / ef, the per-shard lists
 * are all-gathered (B * k * 16 bytes per rank) and merged by (distance, id) on every rank.
 * out_ids_dev [B][k] u64: global ids = shard-local id + id_offset of the owning rank, ~0 padded; out_dist_dev [B][k] f64;
 * out_count_dev [B].  Device pointers; returns after the stream has drained. */
int cz_hnsw_search_sharded(cz_comm *comm, cz_hnsw_index *shard, const float *queries_dev, uint32_t B, uint32_t k,
                           uint32_t ef, uint64_t id_offset, uint64_t *out_ids_dev, double *out_dist_dev,
                           uint32_t *out_count_dev, void *stream);

/* The same partitioned index held by ONE process (what a cozo process with several GPUs uses; HnswSearchRA::iter,
 * query/ra.rs:1085-1121, calls hnsw_knn per parent tuple -- here once per batch): sub-index r lives on GPU r, with one host
 * thread and one RCCL communicator per GPU for the lifetime of the handle.
 *   cz_hnsw_multi_build: `::hnsw create` over shards -- rows [r * ceil(n / n_gpus), ...) of `vectors` (host, in key order) are
 *     built into an index on GPU r with cz_hnsw_build's parameters, all devices at once; n_dist (optional) = evaluations spent.
 *   cz_hnsw_multi_create: shards read from the store (cz_hnsw_desc + vectors per shard, id_offsets[r] = the global id of
 *     shard r's node 0).
 *   cz_hnsw_multi_search: queries [B][dim] (host) -> ids [B][k] u64 global ids (~0 = none), dist [B][k] f64, count [B]
 *     (host): cz_hnsw_search_sharded on every device at once.
 *   cz_hnsw_multi_shards: the number of shards (and their id offsets). */
typedef struct cz_hnsw_multi cz_hnsw_multi;
int cz_hnsw_multi_build(const float *vectors, uint32_t n, uint32_t dim, int metric, uint32_t m, uint32_t ef_construction,
                        int keep_pruned_connections, uint64_t seed, uint32_t max_batch, int n_gpus, uint32_t flags,
                        uint64_t *n_dist, cz_hnsw_multi **out);
int cz_hnsw_multi_create(const cz_hnsw_desc *const *shards, const float *const *vectors, const uint64_t *id_offsets,
                         int n_gpus, cz_hnsw_multi **out);
int cz_hnsw_multi_search(cz_hnsw_multi *m, const float *queries, uint32_t B, uint32_t k, uint32_t ef, uint64_t *ids,
                         double *dist, uint32_t *count);
int cz_hnsw_multi_shards(const cz_hnsw_multi *m, uint64_t *id_offsets);
void cz_hnsw_multi_destroy(cz_hnsw_multi *m);

/* ONE traversal over a graph whose vertices are partitioned across ranks (third row), collectively:
 * rank r passes the out-adjacency of the nodes [row_begin, row_end) -- out_offsets_local [row_end-row_begin+1] relative to
 * the shard, out_targets / weights [E_local] with GLOBAL target ids; starts / goals and every output (full length N per
 * start, as for cz_bfs / cz_sssp) are the same on every rank.
 * cz_bfs_sharded keeps the reference's FIFO semantics across the ranks (parents = first discoverers, discovery order):
 * per level an all-reduce(min) of the N claim words, an all-reduce(sum) of the frontier's counts and one of the next
 * frontier; results are bit-identical to cz_bfs on the whole graph.  cz_sssp_sharded: the one-GPU rule's near-far schedule with a
 * SPARSE exchange -- per round an all-gather of one count per rank and one of the ranks' (target, proposed (cost, parent) word)
 * lists, which every rank applies to its copy of the state (csrc/sharded_traversal.hpp); costs and parents identical to cz_sssp's.
 * cz_sssp_sharded_last_stats: what this thread's last call exchanged -- out[0] rounds, out[1] pairs over all ranks and rounds,
 * out[2] compactions of the far pile, out[3] times the threshold moved (summed over the starts).
 * A set poison flag on ANY rank cancels every rank at the same level / round. */
int cz_bfs_sharded(cz_comm *comm, const uint32_t *out_offsets_local, const uint32_t *out_targets, uint32_t N,
                   uint32_t row_begin, uint32_t row_end, uint64_t E_local, const uint32_t *starts, uint32_t n_starts,
                   const uint32_t *goals, uint32_t n_goals, int share_visited, uint32_t *parent, uint32_t *depth,
                   uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison);
int cz_sssp_sharded(cz_comm *comm, const uint32_t *out_offsets_local, const uint32_t *out_targets, const float *weights,
                    uint32_t N, uint32_t row_begin, uint32_t row_end, uint64_t E_local, const uint32_t *starts,
                    uint32_t n_starts, float *dist, uint32_t *parent, const volatile uint8_t *poison);
int cz_sssp_sharded_last_stats(uint64_t *out4);

/* ConnectedComponents (strongly_connected_components.rs:42-77, strong = false) over a vertex partition of the SYMMETRISED
 * graph, collectively: this rank passes the adjacency of [row_begin, row_end) (offsets relative to the shard).  Per round every
 * rank links the endpoints of its rows' edges in its copy of one forest over all N nodes (pointers lead to lower indices of
 * the same component), the N pointers are all-reduced(min), and the rounds stop when one leaves the forest unchanged; group
 * ids are numbered like cz_connected_components' (rank of the component by its smallest node), identical on every rank.
 * rounds (optional) = rounds run. */
int cz_connected_components_sharded(cz_comm *comm, const uint32_t *offsets_local, const uint32_t *targets, uint32_t N,
                                    uint32_t row_begin, uint32_t row_end, uint64_t E_local, uint32_t *group, uint32_t *n_groups,
                                    uint32_t *rounds, const volatile uint8_t *poison);

/* The same three traversals on n_gpus devices of THIS process (devices 0 .. n_gpus-1; one host thread and one RCCL
 * communicator per device, rows split evenly): host CSR of the whole graph in, the results of cz_bfs / cz_sssp /
 * cz_connected_components out.  What a cozo process with several GPUs calls; the per-rank forms above are what a launcher
 * with one process per GPU calls. */
int cz_bfs_multi(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, int n_gpus,
                 const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, int share_visited,
                 uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison);
int cz_sssp_multi(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                  int n_gpus, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent,
                  const volatile uint8_t *poison);
int cz_connected_components_multi(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E, int n_gpus,
                                  uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison);

/* ShortestPathBFS::run (fixed_rule/algos/shortest_path_bfs.rs:35-113) and the traversal of Bfs::run
 * (algos/bfs.rs:25-113) on the out-CSR (neighbours in sorted order = the KV prefix-scan order).
 *   starts [n_starts]: one BFS per start.  goals [n_goals] or NULL (NULL: full traversal; with goals the
 *   traversal stops after the level in which the last goal was discovered, as the reference does).
 *   parent [n_starts][N] out: first discoverer of every reached node in FIFO order (CZ_NONE: not reached
 *   / the start itself) -- exactly the reference's `backtrace` for every node on a goal path.
 *   depth  [n_starts][N] (optional): BFS level, CZ_NONE if unreached.
 *   order  [n_starts][N] (optional): nodes in the reference's discovery (FIFO push) order, start excluded;
 *   n_reached [n_starts] (optional): how many entries of `order` are valid.
 *   share_visited != 0: Bfs semantics, `visited` shared across starts (algos/bfs.rs:43): a start that was
 *   already reached is skipped and later traversals do not re-enter earlier territory. */
int cz_bfs(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
           uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, int share_visited, uint32_t *parent,
           uint32_t *depth, uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison);
/* Bfs::run's traversal (algos/bfs.rs:43-98: `visited` and `backtrace` shared by all starting nodes) with outputs of O(N) whatever
 * the number of starts -- the rule's default is EVERY node of the `nodes` relation (bfs.rs:33), for which cz_bfs's row of N per
 * start is O(N^2) memory and time:
 *   parent [N]: the one backtrace (every node has one discoverer; CZ_NONE: a start or unreached);
 *   order [N]: the discovery sequences of the starts one after the other; first [n_starts + 1]: start i discovered
 *   order[first[i] .. first[i + 1]) (nothing when an earlier start had reached it: bfs.rs:52-54).
 * A skipped start costs no device work, a traversal resets only what it touched: O(N + E) in all. */
int cz_bfs_shared(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
                  uint32_t n_starts, uint32_t *parent, uint32_t *order, uint32_t *first, const volatile uint8_t *poison);
/* The same traversal, stopped by the caller: Bfs::run evaluates its `condition` on every node as it is discovered and leaves ALL
 * loops once `limit` nodes passed (algos/bfs.rs:78-91).  After every level the nodes that level discovered (FIFO order, already in
 * `order`) are handed to on_level(ctx, start, nodes, n): 0 = go on, > 0 = enough (no further level, no further start), < 0 = the
 * caller failed (CZ_E_INVALID).  The caller sees exactly the nodes the reference's loop would look at, in its order, so the first
 * `limit` passing ones are the reference's `found`; the level in flight when it says stop is delivered whole (a superset of what the
 * reference visited before its break -- `parent` of a found node is the same).  on_level runs on the calling thread. */
typedef int (*cz_bfs_level_fn)(void *ctx, uint32_t start, const uint32_t *nodes, uint32_t n);
int cz_bfs_shared_until(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
                        uint32_t n_starts, cz_bfs_level_fn on_level, void *ctx, uint32_t *parent, uint32_t *order, uint32_t *first,
                        const volatile uint8_t *poison);

/* StronglyConnectedComponent{strong:false}::run = ConnectedComponents
 * (algos/strongly_connected_components.rs:42-77): adjacency of the symmetrised graph
 * (as_directed_graph(undirected = true)); group [N] out = dense rank of each component by its
 * smallest node index, which is what Tarjan over ascending roots yields; n_groups out. */
int cz_connected_components(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E, uint32_t *group,
                            uint32_t *n_groups, const volatile uint8_t *poison);

/* ClusteringCoefficients::run (fixed_rule/algos/triangles.rs:25-110) on the adjacency of the symmetrised graph
 * (as_directed_graph(undirected = true): ascending lists, parallel edges kept):
 *   n_triangles [N] out: #{(i, j) list positions of node v : A[i] > A[j] and A[j] is an out-neighbour of A[i]}
 *   degree [N] out: list length (multiplicity counted).  The shim emits
 *   (node, 2 t / (d (d - 1)) as f64 -- 0.0 when d < 2 --, t, d) like :58-66, :102. */
/* flags: CZ_ADJ_SYMMETRIC = the caller built the adjacency with as_directed_graph(undirected = true) (symmetric, symmetric
 * multiplicities -- what the rule always passes): the kernel that finds every triangle once is taken without further ado.
 * Without the flag the library VERIFIES that precondition exactly (every entry above its node matched in the other list with
 * the same multiplicity, as many entries below as above) and counts with the general kernel when it does not hold. */
#define CZ_ADJ_SYMMETRIC 512u
int cz_clustering_coefficients(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E,
                               uint64_t *n_triangles, uint32_t *degree, const volatile uint8_t *poison, uint32_t flags);

/* dijkstra (algos/shortest_path_dijkstra.rs:274-339) for n_starts sources on the weighted out-CSR
 * (as_directed_weighted_graph, fixed_rule/mod.rs:208-328; f32 weights >= 0):
 *   dist [n_starts][N] f32 (inf = unreachable), parent [n_starts][N]: the SMALLEST predecessor p with
 *   dist[p] + w(p,v) == dist[v] in f32 and dist[p] < dist[v] -- the same on every run, whatever the schedule (a node all
 *   of whose such predecessors sit at its own cost, zero-weight edges, keeps the one that set its cost); CZ_NONE for the
 *   start / unreachable.  (The reference's choice among equal-cost predecessors is its heap's pop order.) */
int cz_sssp(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
            const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent, const volatile uint8_t *poison);
/* dijkstra's early exit (shortest_path_dijkstra.rs:300-306: the search stops once its goal set is exhausted): the same traversal,
 * stopped as soon as every goal of every start is SETTLED (its cost lies below everything still waiting in the piles; checked once
 * per bucket).  dist / parent of the goals and of every node on their shortest paths are cz_sssp's; a node the search had not
 * settled by then is reported unreached (inf / CZ_NONE) -- its tentative cost is never exposed.  An unreachable goal makes the
 * call a full cz_sssp, like the reference.  n_goals == 0: cz_sssp. */
int cz_sssp_goals(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                  const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, float *dist,
                  uint32_t *parent, const volatile uint8_t *poison);

/* A relation's CSR resident on the device.  FixedRule::run (fixed_rule/mod.rs:538-567) is handed the relation anew on every
 * call; on a 10M-node / 100M-edge graph the upload is 8-15 ms of a 20-32 ms call.  cz_graph_upload keeps the arrays of one
 * graph (weights may be null: BFS / ConnectedComponents only) until cz_graph_destroy; cz_graph_acquire / cz_graph_release do the
 * same under the caller's (relation id, snapshot) key like cz_pagerank_cached: acquire hands out the cached graph (taken out
 * of the cache: an entry is never shared between threads) or uploads, release gives it back (CZ_GRAPH_CACHE entries, default
 * 4).  The *_on forms of the rules are the rules of the host-array forms, minus the upload. */
typedef struct cz_graph cz_graph;
int cz_graph_upload(const uint32_t *offsets, const uint32_t *targets, const float *weights, uint32_t N, uint64_t E,
                    cz_graph **out);
void cz_graph_destroy(cz_graph *g);
int cz_graph_acquire(uint64_t key_hi, uint64_t key_lo, const uint32_t *offsets, const uint32_t *targets, const float *weights,
                     uint32_t N, uint64_t E, cz_graph **out, int *cache_hit);
void cz_graph_release(uint64_t key_hi, uint64_t key_lo, cz_graph *g);
void cz_graph_cache_clear(void);
int cz_bfs_on(const cz_graph *g, const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals,
              int share_visited, uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached,
              const volatile uint8_t *poison);
int cz_connected_components_on(const cz_graph *g, uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison);
int cz_sssp_on(const cz_graph *g, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent,
               const volatile uint8_t *poison);
int cz_sssp_goals_on(const cz_graph *g, const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, float *dist,
                     uint32_t *parent, const volatile uint8_t *poison);

/* Where the last cz_bfs / cz_connected_components / cz_sssp / cz_clustering_coefficients / cz_closeness /
 * cz_betweenness / cz_label_propagation call of THIS
 * host thread spent its wall time, in milliseconds (any pointer may be null): these entry points take host arrays
 * (FixedRule::run hands over a relation, fixed_rule/mod.rs:538-567), so a call is upload (allocation + CSR over PCIe) +
 * device (kernels and their control round trips) + download (per-node results back). */
int cz_graph_last_timing(double *upload_ms, double *device_ms, double *download_ms);

/* ClosenessCentrality::run (fixed_rule/algos/all_pairs_shortest_path.rs:97-176) on the weighted out-CSR (weights >= 0):
 *   centrality [N] f64 out = nc * nc / total / (N - 1) computed in f32 like :118-122, total = the finite dijkstra_cost_only
 *   distances from the node summed one after the other in node order, nc = their count (bit-identical to the reference). */
int cz_closeness(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                 double *centrality, const volatile uint8_t *poison);

/* BetweennessCentrality::run (fixed_rule/algos/all_pairs_shortest_path.rs:31-95 over dijkstra_keep_ties,
 * shortest_path_dijkstra.rs:341-450) on the weighted out-CSR, weights > 0:
 *   centrality [N] f64 out: the sum, over every (start, target) pair and every one of the l shortest paths between them,
 *   of 1 / l for each inner node of the path -- computed from path counts over the tight edges (dist[u] + w == dist[v]
 *   in f32, the reference's back pointers) instead of enumerating paths; f64 sums (the reference adds f32 terms path by
 *   path): equal within 1e-5 relative.  CZ_E_UNSUPPORTED for a weight <= 0 or one absorbed by an f32 path cost
 *   (dist[u] + w == dist[u]: shortest-path counts are not defined; the reference's enumeration would not end). */
int cz_betweenness(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                   double *centrality, const volatile uint8_t *poison);

/* LabelPropagation (fixed_rule/algos/label_propagation.rs:56-109) on the weighted out-CSR
 * (as_directed_weighted_graph(undirected, allow_negative_weights = true); finite f32 weights, default 1.0), as ONE fixed
 * execution of the reference's loop -- which shuffles the node order every iteration and breaks ties with thread_rng
 * (:63-66, :85), so that no two of its own runs agree:
 *   order  every iteration visits the colour classes of a deterministic colouring in ascending order, ids ascending
 *          inside a class (round r of the colouring: every uncoloured node whose key hash(id) << 32 | id is the largest
 *          among its uncoloured neighbours, edges in either direction, takes colour r; hash = the murmur3 finaliser);
 *   tie    the smallest label among those whose score equals the largest one.
 * Scores are the reference's: per label the f32 sum of the edge values in adjacency order.  labels [N] out (a label is
 * a node index, :61), iters_run / n_colours optional.  CZ_E_INVALID when a best score is NaN (the reference panics). */
/* flags: CZ_ADJ_SYMMETRIC = the adjacency is symmetric with symmetric multiplicities (the rule under `undirected: true`): the
 * colouring then reads the out-lists alone; without the flag the library verifies exactly whether that holds (the transposed
 * adjacency -- 16 of the rule's 55 ms on the 10M / 200M graph -- is only built when it does not).  Same labels either way. */
int cz_label_propagation(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N,
                         uint64_t E, uint32_t max_iter, uint32_t *labels, uint32_t *iters_run, uint32_t *n_colours,
                         const volatile uint8_t *poison, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
