/*
 * cozo_ingest.h -- C ABI of libcozo_ingest.so: stored rows -> the flat arrays libcozo_gpu.so takes.
 *
 * Host code only (no device, no HIP): the step either side of the GPU path that SURVEY.md section 8 (f1) names.
 * cozo-core reaches the same arrays by decoding every stored row into a Vec<DataValue> and looking every endpoint up
 * in a BTreeMap<DataValue, u32> (fixed_rule/mod.rs:144-195), or, for a vector index, by one KV get + msgpack decode
 * per touched node at query time (runtime/hnsw.rs:110-151, 588-629).  Here the rows are taken as the storage
 * iterator yields them -- key bytes and value bytes -- and are never materialised as values:
 *   key   = 8-byte big-endian relation id, then the key columns in the memcmp encoding (data/memcmp.rs:46-163;
 *           RelationHandle::encode_key_for_store, runtime/relation.rs:247-267; data/tuple.rs:27-52)
 *   value = 8-byte prefix, then the non-key columns as ONE msgpack array of DataValue in rmp-serde 1.2.0's
 *           representation (encode_val_for_store, runtime/relation.rs:275-296; extend_tuple_from_v :526-531):
 *           unit variants are strings ("Null"), the others one-entry maps ({"Num": {"Int": 5}}, {"Str": "a"},
 *           {"Bytes": bin}, {"Bool": true}, {"List": [..]}, {"Vec": [0, bin(little-endian f32)]}; data/value.rs:143-175,
 *           226-252, 493-499).  The decoder also accepts the variant INDEX in place of its name.
 * The memcmp bytes of a value are its identity: DataValue's Ord never returns Equal for values whose encodings
 * differ (Int(1) < Float(1.0), data/value.rs:575-598), so byte equality == the BTreeMap's key equality.
 *
 * Conventions as cozo_gpu.h: 0 / negative status, czi_last_error() per thread, the caller owns what it passes in,
 * the library owns what is behind a handle (pointers returned by accessors live until the handle is freed).
 * Re-entrant, no global state.  Inputs of 2^18 rows and more are spread over min(16, cores) threads of the library's own
 * (environment: CZI_THREADS = n; 1 keeps a call on the calling thread); results do not depend on the thread count.
 */
#ifndef COZO_INGEST_H
#define COZO_INGEST_H
#include <stddef.h>
#include <stdint.h>

#include "cozo_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CZI_OK = 0,
    CZI_E_INVALID = -1,
    CZI_E_CORRUPT = -2,       /* bytes that are not a stored row of the stated shape */
    CZI_E_NOT_AN_EDGE = -3,   /* NotAnEdgeError, fixed_rule/mod.rs:846-850: a row with fewer than two columns */
    CZI_E_BAD_WEIGHT = -4,    /* BadEdgeWeightError, fixed_rule/mod.rs:852-860 */
    CZI_E_UNSUPPORTED = -5,   /* F64 vectors; Json / Regex / Validity node values that live in the VALUE part of a row */
    CZI_E_TOO_LARGE = -6,     /* more than 2^32 - 2 nodes or 2^32 - 1 CSR entries (the reference's ids are u32 too) */
    CZI_E_MISSING_ROW = -7,   /* an index row names a base row / vector that is not there ("corrupted index") */
    CZI_E_OOM = -8            /* host memory */
} czi_status;

const char *czi_last_error(void);
const char *czi_version(void);

/* n_rows stored rows in ascending key order (what StoreTx::range_scan yields, storage/mod.rs):
 * row i = keys[key_off[i] .. key_off[i+1])  /  vals[val_off[i] .. val_off[i+1])  (vals may be NULL when every column
 * that is read is a key column; a row whose value is empty has val_off[i] == val_off[i+1]). */
typedef struct {
    const uint8_t *keys;
    const uint64_t *key_off; /* [n_rows + 1] */
    const uint8_t *vals;
    const uint64_t *val_off; /* [n_rows + 1] or NULL */
    uint64_t n_rows;
    uint32_t n_key_cols;     /* metadata.keys.len() */
} czi_rows;

/* =====================================================================================
 * edge relation -> CSR          replaces: FixedRuleInputRelation::as_directed_graph / as_directed_weighted_graph
 *                                         (fixed_rule/mod.rs:136-328)
 * =====================================================================================
 * Columns 0 and 1 of every row are the endpoints; dense ids in FIRST-APPEARANCE order over the scan, `from` before `to`
 * within a row (:164-179); CZI_UNDIRECTED: every row is mirrored after id assignment (:187-191); adjacency lists
 * ascending by id, parallel edges kept, ties in scan order (CsrLayout::Sorted).  CZI_WEIGHTED: column 2 is the weight
 * (absent => 1.0; not a number / not finite / negative without CZI_ALLOW_NEGATIVE_WEIGHTS => CZI_E_BAD_WEIGHT,
 * :226-262), stored as f32.
 * CZI_ORDERED_IDS: ids are instead the RANK of the node value's key bytes, i.e. the order a stored relation is scanned
 * in.  ShortestPathBFS / Bfs walk `prefix_iter(node)` (algos/shortest_path_bfs.rs:64-72, algos/bfs.rs:58-66), i.e. they
 * meet neighbours in key order of the `to` value; with rank ids "ascending id" in the CSR is that order.  (Byte order is
 * DataValue::cmp for every variant except Vec and Json / Validity, whose tags are out of enum order, memcmp.rs:22-36.) */
#define CZI_UNDIRECTED 1u
#define CZI_WEIGHTED 2u
#define CZI_ALLOW_NEGATIVE_WEIGHTS 4u
#define CZI_ORDERED_IDS 8u
typedef struct czi_graph czi_graph;
int czi_graph_ingest(const czi_rows *rel, uint32_t flags, czi_graph **out);
void czi_graph_free(czi_graph *g);
uint32_t czi_graph_node_count(const czi_graph *g);
uint64_t czi_graph_edge_count(const czi_graph *g); /* CSR entries per direction (2x the rows when undirected) */
/* out-adjacency (inverse = 0: what cz_bfs / cz_sssp / cz_connected_components take) or in-adjacency (inverse = 1:
 * cz_pagerank's in_offsets / in_sources).  offsets [N+1], targets [E], weights [E] or NULL. */
int czi_graph_csr(const czi_graph *g, int inverse, uint32_t *offsets, uint32_t *targets, float *weights);
/* `indices` of the reference (id -> node value), as the memcmp bytes of each value, concatenated:
 * value of id i = bytes[off[i] .. off[i+1]) -- DataValue::decode_from_key reads it back (N decodes instead of 2E). */
int czi_graph_node_keys(const czi_graph *g, const uint8_t **bytes, const uint64_t **off);
/* `inv_indices.get(node)` (the BTreeMap<DataValue, u32> as_directed_graph returns, fixed_rule/mod.rs:143-145; used e.g. by
 * algos/shortest_path_dijkstra.rs:47-66 to map start / goal nodes): id of the node whose memcmp bytes are `key`, CZ_NONE if absent */
uint32_t czi_graph_lookup(const czi_graph *g, const uint8_t *key, uint64_t len);

/* =====================================================================================
 * `tbl:idx` + base relation -> cz_hnsw_desc + vectors        replaces: the per-query VectorCache / hnsw_get_neighbours
 *                                                            reads (runtime/hnsw.rs:110-151, 588-629, 891-915)
 * =====================================================================================
 * idx : every row of the index relation (schema runtime/relation.rs:1064-1126; K = base->n_key_cols):
 *       key [layer, fr_key x K, fr__field, fr__sub_idx, to_key x K, to__field, to__sub_idx], value [dist, hash, ignore_link].
 * base: every row of the base relation.  vec_fields = HnswIndexManifest::vec_fields (positions in the base tuple),
 *       dim = vec_dim, m_max / m_max0 = m_neighbours / 2 m_neighbours (row widths of the upper levels / of level 0; a longer
 *       live row widens level 0 or all upper levels together).
 * A node is one (row key, field, sub index) (:55).  Node ids follow the key order of the self-loop rows of layer 0, so
 * "ascending id" is the order hnsw_get_neighbours yields neighbours in.  Rows dropped exactly as :609-624 drops them:
 * the self-loop row, links to a vector of the same base row, ignore_link rows.  The canary row (layer 1, :641-669) is
 * skipped; an index holding nothing else comes out with n_levels = 0.  Entry point = `fr` of the first row (:891-915). */
typedef struct czi_hnsw czi_hnsw;
int czi_hnsw_ingest(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields, uint32_t dim,
                    int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw **out);
void czi_hnsw_free(czi_hnsw *h);
/* the descriptor + vectors cz_hnsw_index_create takes (pointers into the handle) */
int czi_hnsw_desc(const czi_hnsw *h, cz_hnsw_desc *desc, const float **vectors);
/* The same for an index whose manifest says VecElementType::F64 (runtime/hnsw.rs:33-44): the rows' vectors are taken as f64
 * and handed to cz_hnsw_index_create_f64 unconverted.  A node whose stored vector has the other element type is an error in
 * either form (hnsw_put indexes a vector only when its type is the manifest's, :694-706). */
int czi_hnsw_ingest_f64(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields, uint32_t dim,
                        int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw **out);
int czi_hnsw_desc_f64(const czi_hnsw *h, cz_hnsw_desc *desc, const double **vectors);
/* node id -> CompoundKey: the base row's position in `base`, the field (column position) and the sub index (-1: the
 * column is the vector itself); [n] each */
int czi_hnsw_nodes(const czi_hnsw *h, const uint64_t **base_row, const uint32_t **field, const int32_t **sub);
/* rows of the index relation seen / kept as live links (structural counters, runtime/tests.rs:730,737) */
int czi_hnsw_row_counts(const czi_hnsw *h, uint64_t *n_rows, uint64_t *n_self, uint64_t *n_live_links, uint64_t *n_ignored);

/* =====================================================================================
 * flat index -> the stored rows of its `tbl:idx` relation        replaces: the puts of hnsw_put_vector /
 *                                                                hnsw_put_fresh_at_levels (runtime/hnsw.rs:270-330, 630-678)
 * =====================================================================================
 * The way back: what `::hnsw create` leaves in the store for an index built by cz_hnsw_build, as
 * key / value bytes ready for `store_tx.put`, in key order:
 *   per level l (layer -l) and node present there
 *     the self-loop row  [layer, fr.., fr..]  ->  [degree as f64, Bytes(SHA-256 of the vector's little-endian bytes,
 *                                                  data/value.rs:333-348), false]
 *     per live link      [layer, fr.., to..]  ->  [distance f64, Null, false]
 *   the canary row       [1, Null x (2K+4)]    ->  [bottom layer as an Int, Bytes(key of the entry's self row with a Null
 *                                                  layer), false]                                        (:641-669)
 * node_keys: per node the memcmp bytes of its CompoundKey columns [row key x K, field, sub index], concatenated
 * (node i = node_keys[node_key_off[i] .. node_key_off[i+1])); level_dist[l] = [size][width] f64, the distance of every
 * link slot of desc->level_nbrs[l] (cz_distance_batch over the (node, neighbour) pairs; slots holding CZ_NONE are not
 * read).  relation_id = the index relation's id (the 8-byte prefix of every key and value). */
typedef struct czi_row_buf czi_row_buf;
int czi_hnsw_encode_rows(const cz_hnsw_desc *desc, const float *vectors, const uint8_t *node_keys,
                         const uint64_t *node_key_off, const double *const *level_dist, uint64_t relation_id,
                         czi_row_buf **out);
/* the same with the f64 of the self rows handed in: level_degree[l] = [size] (cz_hnsw_index_export_degrees), or NULL /
 * a NULL level = the number of link rows.  With extend_candidates a shrink can leave the stored degree one above the link
 * rows (the target selects itself and the self row is put back, runtime/hnsw.rs:413-433, 352-357). */
int czi_hnsw_encode_rows_degrees(const cz_hnsw_desc *desc, const float *vectors, const uint8_t *node_keys,
                                 const uint64_t *node_key_off, const double *const *level_dist,
                                 const double *const *level_degree, uint64_t relation_id, czi_row_buf **out);
/* the rows (pointers into the buffer; n_key_cols is left 0 -- the caller knows K) */
int czi_row_buf_rows(const czi_row_buf *b, czi_rows *rows);
void czi_row_buf_free(czi_row_buf *b);

#ifdef __cplusplus
}
#endif
#endif
