/*
 * cozo_oracle.h -- CPU ORACLE for the cozo HNSW / fixed-rule hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithms (cozodb/cozo v0.7.6, cozo-core).  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it; the shipped
 * product path (libcozo_gpu.so) never links, imports or calls anything here.
 *
 * PARITY STATUS: "parity unpinned" for distances / HNSW / PageRank / CC /
 * Dijkstra: the reference's own tests hold no numeric golden vectors for
 * those and the reference cannot be compiled in this
 * environment (no cargo/rustc).  Pinned (tiny) cases: the `love` graph of
 * algos/shortest_path_bfs.rs:124-174 and the hand-computable distance
 * examples of runtime/tests.rs:691-697; both are in tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * cozo-core/src/).
 */
#ifndef COZO_ORACLE_H
#define COZO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_L2 = 0, ORC_COSINE = 1, ORC_IP = 2 };
/* ORC_DOT_NDARRAY: ndarray 0.15.6 `unrolled_dot` order (what the reference runs).
 * ORC_DOT_GPU:     the summation tree of the HIP kernels (per-lane fma chains + xor butterfly);
 *                  used to show the GPU traversal is bit-identical given identical arithmetic. */
enum { ORC_DOT_NDARRAY = 0, ORC_DOT_GPU = 1, ORC_DOT_SEQ = 2 /* k-ordered fmaf chain = the MFMA GEMM form (Cosine / IP) */ };

#define ORC_NONE 0xFFFFFFFFu

/* ---- distances (runtime/hnsw.rs:66-109, data/functions.rs:2185-2255) ---- */
float orc_dot_ndarray(const float *a, const float *b, size_t n);
float orc_dot_gpu(const float *a, const float *b, int dim);
float orc_dot_seq(const float *a, const float *b, int dim);
float orc_l2_gpu(const float *a, const float *b, int dim);
double orc_distance(int metric, int dot_mode, const float *a, const float *b, int dim);
void orc_distance_pairs(int metric, int dot_mode, const float *base, const float *queries, int dim,
                        const uint32_t *pairs /* [P][2] = (query, node) */, uint64_t P, double *out);

/* ---- HNSW index construction (runtime/hnsw.rs:155-538, 630-678) ---- */
typedef struct orc_hnsw orc_hnsw;
orc_hnsw *orc_hnsw_new(int dim, int metric, int m, int ef_construction, int extend_candidates,
                       int keep_pruned_connections, int dot_mode);
void orc_hnsw_free(orc_hnsw *h);
/* insert vectors id = n_existing .. n_existing+n-1 in order; levels[i] >= 0 is -layer of hnsw.rs:46-52 */
int orc_hnsw_insert(orc_hnsw *h, const float *vectors, uint32_t n, const int32_t *levels);
/* hnsw_remove_vec (hnsw.rs:754-868) for one node; 1 when it was indexed.  Rows the reference leaves pointing at the removed
 * node are counted by orc_hnsw_dangling_links and skipped by the exports. */
int orc_hnsw_remove(orc_hnsw *h, uint32_t node);
/* rank[node] = position of the node's key among all keys (nodes held and to come); NULL = ids are key order.  The entry point
 * is the smallest KEY on the top layer (hnsw.rs:184-191, 891-899). */
void orc_hnsw_set_key_order(orc_hnsw *h, const uint32_t *rank, uint32_t n);
/* row_of[node] = the base row the node's vector comes from, for the nodes held and the ones to come; links inside one row are
 * stored and counted but never read (hnsw.rs:609-610).  NULL: one vector per row. */
void orc_hnsw_set_row_of(orc_hnsw *h, const uint32_t *row_of, uint32_t n);
uint64_t orc_hnsw_dangling_links(const orc_hnsw *h);
double orc_hnsw_degree(const orc_hnsw *h, uint32_t node, int level);
uint32_t orc_hnsw_size(const orc_hnsw *h);
int orc_hnsw_n_levels(const orc_hnsw *h);
uint32_t orc_hnsw_entry(const orc_hnsw *h);
uint32_t orc_hnsw_level_size(const orc_hnsw *h, int level);
int orc_hnsw_level_width(const orc_hnsw *h, int level);
/* flat export of one level: node_ids[level_size] ascending, nbrs[level_size][width] (ascending, ORC_NONE padded) */
void orc_hnsw_export_level(const orc_hnsw *h, int level, uint32_t *node_ids, uint32_t *nbrs);
uint64_t orc_hnsw_dist_count(const orc_hnsw *h);
/* number of link rows (incl. soft-deleted / self rows) -- structural counters used by tests */
uint64_t orc_hnsw_link_rows(const orc_hnsw *h, int include_ignored);

/* ---- HNSW search over the flat layout (runtime/hnsw.rs:539-629, 869-1012) ---- */
typedef struct {
    uint32_t n;
    int dim;
    int metric;
    int dot_mode;
    const float *vectors;          /* [n][dim] */
    int n_levels;                  /* 0 => empty index */
    const uint32_t *level_size;    /* [n_levels] */
    const int32_t *level_width;    /* [n_levels] */
    const uint32_t *const *level_nodes; /* [n_levels] -> ids ascending (level 0 may be NULL = identity) */
    const uint32_t *const *level_nbrs;  /* [n_levels] -> [size][width] */
    uint32_t entry;
    const double *vectors64; /* an F64 index (VecElementType::F64): [n][dim] f64, `vectors` unused; queries are then f64 rows behind
                                the float pointers of orc_hnsw_knn / orc_hnsw_knn_batch */
} orc_flat_index;

/* VectorCache::dist, the F64 arms (runtime/hnsw.rs:73-78, 86-95, 102-106): every dot product in f64 -- ndarray's unrolled_dot
 * (ORC_DOT_NDARRAY) or the HIP kernels' tree over 16-byte chunks of two doubles (ORC_DOT_GPU, cozo_amd/csrc/distance_f64.h) */
double orc_dot_ndarray_f64(const double *a, const double *b, size_t n);
double orc_dot_gpu_f64(const double *a, const double *b, int dim);
double orc_distance_f64(int metric, int dot_mode, const double *a, const double *b, int dim);
void orc_distance_pairs_f64(int metric, int dot_mode, const double *base, const double *queries, int dim, const uint32_t *pairs,
                            uint64_t P, double *out);

/* returns number of results (<= k), ascending distance; n_dist accumulates distance evaluations */
int orc_hnsw_knn(const orc_flat_index *ix, const float *q, int k, int ef, int has_radius, double radius,
                 uint32_t *out_ids, double *out_dist, uint64_t *n_dist);
/* batch helper (OpenMP over queries when threads > 1); out arrays [B][k], counts [B] */
void orc_hnsw_knn_batch(const orc_flat_index *ix, const float *queries, uint32_t B, int k, int ef,
                        int has_radius, double radius, uint32_t *out_ids, double *out_dist,
                        uint32_t *out_count, uint64_t *n_dist_total, int threads);
/* exact k-NN by exhaustive scan (recall ground truth), (dist,id)-ordered */
void orc_bruteforce_knn(int metric, int dot_mode, const float *base, uint32_t n, int dim, const float *queries,
                        uint32_t B, int k, uint32_t *out_ids, double *out_dist, int threads);

/* ---- relation -> graph (fixed_rule/mod.rs:136-328) ---- */
/* first-appearance id assignment over rows scanned in key order; returns node count, fills
 * from_idx/to_idx [E] and indices[<=2E] (original key of each id) */
uint32_t orc_assign_ids(const int64_t *from, const int64_t *to, uint64_t E, uint32_t *from_idx, uint32_t *to_idx,
                        int64_t *indices);
/* CsrLayout::Sorted adjacency, duplicates kept.  If undirected, every row is mirrored first.
 * off[n+1], tgt[E'] (E' = E or 2E); w_in/w_out optional */
void orc_build_csr(uint32_t n, uint64_t E, const uint32_t *src, const uint32_t *dst, const float *w_in,
                   int undirected, uint64_t *off, uint32_t *tgt, float *w_out);

/* ---- PageRank (fixed_rule/algos/pagerank.rs:29-56 -> graph 0.3.1 page_rank) ---- */
int orc_pagerank(uint32_t n, const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, float damping,
                 double tolerance, uint32_t max_iter, float *scores, uint32_t *iters_run, double *final_err,
                 int threads);

/* both readings of the crate's loop (see the .c): mode ORC_PR_JACOBI / ORC_PR_INPLACE (one thread, ascending nodes),
 * err_f64_diff 0 / 1 */
#define ORC_PR_JACOBI 0
#define ORC_PR_INPLACE 1
int orc_pagerank_mode(uint32_t n, const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, float damping,
                      double tolerance, uint32_t max_iter, int mode, int err_f64_diff, float *scores, uint32_t *iters_run,
                      double *final_err);
/* the in-place reading on `threads` rayon threads under ONE deterministic lockstep schedule of the crate's 16 384-node chunks (see the .c) */
int orc_pagerank_inplace_lockstep(uint32_t n, const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, float damping,
                                  double tolerance, uint32_t max_iter, uint32_t threads, uint32_t chunk, float *scores,
                                  uint32_t *iters_run, double *final_err);

/* ---- ShortestPathBFS (fixed_rule/algos/shortest_path_bfs.rs:35-113) ---- */
/* parent[n]: ORC_NONE when no backtrace entry.  Goal semantics as the reference (start itself has no entry). */
void orc_shortest_path_bfs(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t start,
                           const uint32_t *goals, uint32_t n_goals, uint32_t *parent);
/* full FIFO BFS (algos/bfs.rs:25-113 traversal order): order[] = discovery order (excluding already
 * visited), parent[]; visited[] is in/out so that several starts can share it.  returns #discovered */
uint32_t orc_bfs_order(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t start, uint8_t *visited,
                       uint32_t *parent, uint32_t *order);

/* ---- (Strongly)ConnectedComponents (algos/strongly_connected_components.rs:42-149) ---- */
/* Tarjan exactly as TarjanSccG (explicit stack), grp[n] = rank of the component's root discovery id */
uint32_t orc_tarjan_groups(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t *grp);

/* ---- ClusteringCoefficients (algos/triangles.rs:25-110) on the symmetrised out-CSR (duplicates kept) ---- */
void orc_clustering_coefficients(uint32_t n, const uint64_t *off, const uint32_t *tgt, double *cc, uint64_t *n_tri,
                                 uint32_t *degree);
/* the same loop over the nodes first, first + step, ... until max_seconds are spent (bench.py's bounded CPU baseline) */
uint64_t orc_clustering_coefficients_sample(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t first, uint32_t step,
                                            double max_seconds, uint64_t *n_tri, uint64_t *edges);

/* ---- ShortestPathDijkstra (algos/shortest_path_dijkstra.rs:274-339) ---- */
/* goals NULL => all nodes.  dist[n] f32 (inf unreachable), parent[n] (ORC_NONE) */
void orc_dijkstra(uint32_t n, const uint64_t *off, const uint32_t *tgt, const float *w, uint32_t start,
                  const uint32_t *goals, uint32_t n_goals, float *dist, uint32_t *parent);

/* ---- BetweennessCentrality (algos/all_pairs_shortest_path.rs:31-95 over dijkstra_keep_ties, shortest_path_dijkstra.rs:341-450) ---- */
/* literal path enumeration (small graphs); out[n] f32 as the reference accumulates it; -1 when a start has > max_paths paths */
int orc_betweenness(uint32_t n, const uint64_t *off, const uint32_t *tgt, const float *w, float *out, uint64_t max_paths);

/* ---- LabelPropagation (algos/label_propagation.rs:56-109) ---- */
/* The reference shuffles the node order every iteration and breaks score ties with thread_rng (:63-66, :90): no two runs agree.
 * orc_label_propagation_in_order is its loop with both choices handed in: the SAME node order in every iteration and the SMALLEST
 * label among the best-scored ones -- one of the executions the reference can produce.  labels[n] out, returns iterations run
 * (the one that changed nothing included), -1 when a best score is NaN (the reference panics there: `choose` on an empty list). */
int orc_label_propagation_in_order(uint32_t n, const uint64_t *off, const uint32_t *tgt, const float *w, const uint32_t *order,
                                   uint32_t max_iter, uint32_t *labels);
/* the order the GPU rule fixes: colour classes of a deterministic colouring of the graph (both edge directions):
 * in round r every still uncoloured node whose key (hash(id) << 32 | id) is the largest among its uncoloured neighbours takes colour r.
 * colour[n] out, returns the number of colours */
uint32_t orc_lp_colouring(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t *colour);

#ifdef __cplusplus
}
#endif
#endif
