// hnsw.hpp -- C++ host mirror of the reference's HNSW operator surface over libcozo_gpu.
//
//   HnswIndexManifest   runtime/hnsw.rs:27-43      (fields as stored in the index relation's metadata)
//   CompoundKey         runtime/hnsw.rs:55         (row key tuple, field index, sub-index): one node of the graph
//   HnswSearch          data/program.rs:975-991    per-query parameters (k, ef, bind_*, radius, filter)
//   GpuHnswIndex        the device-resident flat export of `tbl:idx` + the indexed vectors of the base relation;
//                       hnsw_put order (:679-727): rows in key order, vec_fields in manifest order, List positions
//   hnsw_knn            runtime/hnsw.rs:869-1012   one parent tuple
//   HnswSearchRA::iter  query/ra.rs:1085-1121      the parent iterator is drained into ONE batch, searched by one
//                       cz_hnsw_search_batch launch, rows re-emitted in parent order as `parent ++ result`
// SCOPE (frozen in round 6).  The NORMATIVE executable host mirror is the Python one (cozo_amd/hnsw.py,
// cozo_amd/fixed_rule.py): it is the one every parity test and bench.py drive, and it grows with the C ABI.  This C++ mirror is kept
// as the compiled twin of the part the Rust shim (integration/rust/) needs a second opinion on -- the FixedRule surface, the F32
// HnswSearchRA batching and row assembly -- and is held to the Python one by tests/test_mirrors_agree.py, row for row.  Declared
// gaps (tests/test_mirrors_agree.py::CPP_MIRROR_GAPS lists them and asserts each is REFUSED here, not silently different):
//   * F64 indices (VecElementType::F64): searched by the Python mirror and the Rust source only; GpuHnswIndex here throws.
//   * cz_pagerank_inplace_plan_* (the resident in-place plan), cz_hnsw_index_distance_batch, cz_hnsw_index_settle: Python only.
// Radius, bind columns and the filter predicate are applied on the host to the (node, distance) rows the GPU
// returns, exactly where the reference applies them (:943-1006).  No CPU fallback: without the device library or
// a gfx950 device every search throws GpuError.
#pragma once
#include <cmath>
#include <map>
#include <optional>
#include <string>
#include <vector>

#include "codec.hpp"
#include "fixed_rule.hpp"

struct cz_hnsw_index;

namespace cozo {

enum class HnswDistance { L2 = 0, Cosine = 1, InnerProduct = 2 };  // parse/sys.rs:76-98 (values = cz_metric)
enum class VecElementType { F32, F64 };

struct HnswIndexManifest {
    std::string base_relation, index_name;
    size_t vec_dim = 0;
    VecElementType dtype = VecElementType::F32;
    std::vector<size_t> vec_fields;  // column positions in the base relation (keys first, then non-keys)
    HnswDistance distance = HnswDistance::L2;
    size_t ef_construction = 100;
    size_t m_neighbours = 16;
    size_t m_max = 16, m_max0 = 32;       // runtime/relation.rs:1136-1151: m, 2m
    double level_multiplier = 1.0 / std::log(16.0);
    bool extend_candidates = false, keep_pruned_connections = false;

    // ::hnsw create option handling (parse/sys.rs:515-624 defaults, relation.rs:1136-1151 derived fields)
    static HnswIndexManifest create(std::string base, std::string index, size_t dim, std::vector<size_t> fields,
                                    HnswDistance distance, size_t m, size_t ef_construction);
};

// the stored relation an index hangs off: key columns then non-key columns; rows sorted by key
struct BaseRelation {
    std::vector<std::string> keys, non_keys;
    std::vector<Tuple> rows;
    const std::string &column_name(size_t idx) const { return idx < keys.size() ? keys[idx] : non_keys[idx - keys.size()]; }
};

struct CompoundKey {
    uint32_t row;    // position of the base row (stands for the row's key tuple)
    uint32_t field;  // column position
    int32_t sub;     // -1, or the position inside a List of vectors
};

// one term of a filter that is a conjunction of `base column OP constant` over numeric columns (what the shim extracts from
// the filter bytecode; the reference evaluates the same comparisons on the ef candidate rows, hnsw.rs:994-998)
struct ColumnPredicate {
    size_t column;     // position in the base relation's row
    int op;            // cz_cmp_op
    DataValue constant;  // Int or Float
};

struct HnswSearch {
    size_t k = 10, ef = 10;
    bool bind_field = false, bind_field_idx = false, bind_distance = false, bind_vector = false;
    std::optional<double> radius;
    std::optional<TuplePredicate> filter;  // the compiled filter expression over the bound result tuple
    // evaluated on the DEVICE over all ef candidates (cz_hnsw_search_filtered) when `filter` is absent, there are 1..4 of
    // them and every value of the named columns is an Int (or every value a Float); otherwise on the host rows with the
    // reference's comparison semantics (data/functions.rs:298-380)
    std::vector<ColumnPredicate> predicates;
};

class GpuHnswIndex {
    cz_hnsw_index *h_ = nullptr;
    HnswIndexManifest manifest_;
    const BaseRelation *base_ = nullptr;
    std::vector<CompoundKey> nodes_;  // node id -> CompoundKey
    std::vector<uint8_t> removed_;    // node id -> taken out by remove_rows (keeps its id, has no index rows)
    uint64_t build_n_dist_ = 0;
    // per base column: the device copy of its per-node values (nullptr: the column is not purely Int / purely Float)
    mutable std::map<size_t, cz_column *> columns_;
    cz_column *device_column(size_t column) const;

public:
    GpuHnswIndex() = default;
    GpuHnswIndex(const GpuHnswIndex &) = delete;
    GpuHnswIndex &operator=(const GpuHnswIndex &) = delete;
    GpuHnswIndex(GpuHnswIndex &&o) noexcept { *this = std::move(o); }
    GpuHnswIndex &operator=(GpuHnswIndex &&o) noexcept;
    ~GpuHnswIndex();

    // `::hnsw create` (create_hnsw_index, runtime/relation.rs:1010-1201 -> hnsw_put per row) on the GPU.
    // `levels` (optional) fixes every node's level (the reference draws them from an unseedable thread_rng);
    // max_batch = 1 reproduces the sequential insertion order exactly.
    static GpuHnswIndex create(const HnswIndexManifest &manifest, const BaseRelation &base, uint64_t seed = 0,
                               uint32_t max_batch = 0, const std::vector<int32_t> *levels = nullptr);

    // The index as it lives in the store: the key / value bytes of `tbl:idx` and of the base relation.
    // libcozo_ingest turns them into the flat layout (include/cozo_ingest.h: node ids in key order, rows dropped as
    // hnsw_get_neighbours drops them), cz_hnsw_index_create uploads it.  `base` holds the decoded rows for the row
    // assembly of hnsw_knn and must be the same relation, in key order.
    static GpuHnswIndex from_stored(const HnswIndexManifest &manifest, const StoredRows &idx, const StoredRows &base_rows,
                                    const BaseRelation &base);
    // The way back (section 8 f2): every `tbl:idx` row of this index as key / value bytes in key order, ready for
    // store_tx.put -- link tables exported from the device, link distances recomputed by cz_distance_batch (the values
    // the kernels work with), self-loop rows with degree and vector hash, the canary row (hnsw.rs:270-330, 630-678).
    StoredRows index_rows(uint64_t relation_id) const;

    // Index maintenance on a later write (query/stored.rs:431-450, 486-503 -> runtime/hnsw.rs:679-727, 728-868), on the device:
    //   put_rows     hnsw_put for the rows base.rows[first_row ..) the caller appended to the base relation since the index was
    //                built / last extended (cz_hnsw_insert; max_batch = 1 == the reference's one-at-a-time order)
    //   remove_rows  hnsw_remove for base rows: every vector of each row leaves every level, no link names it any more
    // then index_rows() again and stored_rows_delta (codec.hpp) against the rows the store holds = what to put / delete.
    void put_rows(uint32_t first_row, uint64_t seed = 0, uint32_t max_batch = 0, const std::vector<int32_t> *levels = nullptr);
    void remove_rows(const std::vector<uint32_t> &rows);

    size_t node_count() const { return nodes_.size(); }
    const CompoundKey &node(uint32_t id) const { return nodes_[id]; }
    const HnswIndexManifest &manifest() const { return manifest_; }
    uint64_t build_distance_evaluations() const { return build_n_dist_; }
    uint64_t device_bytes() const;

    // SessionTx::hnsw_knn for a batch of query vectors; result[i] = the reference's Vec<Tuple> for queries[i]
    std::vector<std::vector<Tuple>> hnsw_knn_batch(const std::vector<const std::vector<float> *> &queries,
                                                   const HnswSearch &config, const Poison &poison) const;
    std::vector<Tuple> hnsw_knn(const std::vector<float> &q, const HnswSearch &config, const Poison &poison) const {
        return hnsw_knn_batch({&q}, config, poison)[0];
    }
    // raw (node id, distance) rows, ascending, <= k per query
    void search_raw(const float *queries, uint32_t B, uint32_t k, uint32_t ef, std::vector<uint32_t> &ids,
                    std::vector<double> &dist, std::vector<uint32_t> &count, const Poison &poison,
                    const std::optional<double> &radius = {}) const;
};

// HnswSearchRA (query/ra.rs:1085-1121): `parent` yields tuples carrying a DataValue::Vec at `bind_idx`
struct HnswSearchRA {
    const GpuHnswIndex *index;
    HnswSearch hnsw_search;
    size_t bind_idx;
    std::vector<Tuple> iter(const std::vector<Tuple> &parent, const Poison &poison) const;
};

}  // namespace cozo
