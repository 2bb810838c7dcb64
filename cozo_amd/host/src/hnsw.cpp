// hnsw.cpp -- HNSW half of the C++ host mirror (see hnsw.hpp for the reference map).
#include "cozo_host/hnsw.hpp"

#include <cstring>

#include <algorithm>
#include <memory>

#include "cozo_gpu.h"
#include "cozo_ingest.h"

namespace cozo {

HnswIndexManifest HnswIndexManifest::create(std::string base, std::string index, size_t dim, std::vector<size_t> fields,
                                            HnswDistance distance, size_t m, size_t ef_construction) {
    HnswIndexManifest mf;
    mf.base_relation = std::move(base);
    mf.index_name = std::move(index);
    mf.vec_dim = dim;
    mf.vec_fields = std::move(fields);
    mf.distance = distance;
    mf.m_neighbours = m;
    mf.ef_construction = ef_construction;
    mf.m_max = m;       // runtime/relation.rs:1145
    mf.m_max0 = m * 2;  // :1146
    mf.level_multiplier = 1.0 / std::log((double)m);  // :1147
    return mf;
}

GpuHnswIndex &GpuHnswIndex::operator=(GpuHnswIndex &&o) noexcept {
    if (this != &o) {
        if (h_) cz_hnsw_index_destroy(h_);
        h_ = o.h_;
        o.h_ = nullptr;
        manifest_ = std::move(o.manifest_);
        base_ = o.base_;
        nodes_ = std::move(o.nodes_);
        removed_ = std::move(o.removed_);
        build_n_dist_ = o.build_n_dist_;
        for (auto &kv : columns_)
            if (kv.second) cz_column_destroy(kv.second);
        columns_ = std::move(o.columns_);
        o.columns_.clear();
    }
    return *this;
}

GpuHnswIndex::~GpuHnswIndex() {
    for (auto &kv : columns_)
        if (kv.second) cz_column_destroy(kv.second);
    if (h_) cz_hnsw_index_destroy(h_);
}

uint64_t GpuHnswIndex::device_bytes() const { return h_ ? cz_hnsw_index_bytes(h_) : 0; }

// nodes are appended row by row: a row that carries several vectors has them side by side
static bool shares_rows(const std::vector<CompoundKey> &nodes) {
    for (size_t i = 1; i < nodes.size(); i++)
        if (nodes[i].row == nodes[i - 1].row) return true;
    return false;
}
static std::vector<uint32_t> rows_of(const std::vector<CompoundKey> &nodes) {
    std::vector<uint32_t> r(nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) r[i] = nodes[i].row;
    return r;
}

GpuHnswIndex GpuHnswIndex::create(const HnswIndexManifest &manifest, const BaseRelation &base, uint64_t seed,
                                  uint32_t max_batch, const std::vector<int32_t> *levels) {
    if (manifest.dtype != VecElementType::F32)
        throw GpuError(CZ_E_UNSUPPORTED, "only F32 vector indices are GPU-resident");
    GpuHnswIndex ix;
    ix.manifest_ = manifest;
    ix.base_ = &base;
    // hnsw_put (runtime/hnsw.rs:679-727): per row, per indexed field, a Vec or every Vec inside a List
    std::vector<float> flat;
    auto push = [&](const std::vector<float> &v, uint32_t row, uint32_t field, int32_t sub) {
        if (v.size() != manifest.vec_dim)
            throw CozoError("hnsw::dim_mismatch", "vector of length " + std::to_string(v.size()) + " in an index of dimension " +
                                                      std::to_string(manifest.vec_dim));
        flat.insert(flat.end(), v.begin(), v.end());
        ix.nodes_.push_back({row, field, sub});
    };
    for (uint32_t r = 0; r < base.rows.size(); r++) {
        const Tuple &t = base.rows[r];
        for (size_t f : manifest.vec_fields) {
            if (f >= t.size()) continue;
            if (const std::vector<float> *v = t[f].get_vec()) {
                push(*v, r, (uint32_t)f, -1);
            } else if (const std::vector<DataValue> *l = t[f].get_slice()) {
                for (size_t s = 0; s < l->size(); s++)
                    if (const std::vector<float> *v2 = (*l)[s].get_vec()) push(*v2, r, (uint32_t)f, (int32_t)s);
            }
        }
    }
    if (levels && levels->size() != ix.nodes_.size())
        throw CozoError("hnsw::bad_levels", "levels must hold one entry per indexed vector");
    if (ix.nodes_.empty()) return ix;  // empty index: hnsw_knn returns no rows (:903-909)
    const uint32_t build_flags = manifest.extend_candidates ? CZ_HNSW_EXTEND_CANDIDATES : 0u;
    if (shares_rows(ix.nodes_)) {
        // Some row carries several vectors: hnsw_get_neighbours drops every link inside one base row (:609-610), also while
        // the index is being built.  The library needs every node's base row for that: an empty handle, the rows, the insert.
        const std::vector<uint32_t> rows = rows_of(ix.nodes_);
        check_gpu(cz_hnsw_build(nullptr, 0, (uint32_t)manifest.vec_dim, (int)manifest.distance, (uint32_t)manifest.m_neighbours,
                                (uint32_t)manifest.ef_construction, manifest.keep_pruned_connections ? 1 : 0, nullptr, 0, 0, nullptr,
                                &ix.h_, 0, nullptr));
        check_gpu(cz_hnsw_set_row_of(ix.h_, rows.data(), (uint32_t)rows.size()));
        check_gpu(cz_hnsw_insert(ix.h_, flat.data(), (uint32_t)ix.nodes_.size(), (uint32_t)manifest.m_neighbours,
                                 (uint32_t)manifest.ef_construction, manifest.keep_pruned_connections ? 1 : 0,
                                 levels ? levels->data() : nullptr, seed, max_batch, &ix.build_n_dist_, build_flags, nullptr));
        return ix;
    }
    check_gpu(cz_hnsw_build(flat.data(), (uint32_t)ix.nodes_.size(), (uint32_t)manifest.vec_dim, (int)manifest.distance,
                            (uint32_t)manifest.m_neighbours, (uint32_t)manifest.ef_construction,
                            manifest.keep_pruned_connections ? 1 : 0, levels ? levels->data() : nullptr, seed, max_batch,
                            &ix.build_n_dist_, &ix.h_, manifest.extend_candidates ? CZ_HNSW_EXTEND_CANDIDATES : 0u, nullptr));
    return ix;
}

void GpuHnswIndex::put_rows(uint32_t first_row, uint64_t seed, uint32_t max_batch, const std::vector<int32_t> *levels) {
    std::vector<float> flat;
    const size_t n_before = nodes_.size();
    for (uint32_t r = first_row; r < base_->rows.size(); r++) {  // hnsw_put (runtime/hnsw.rs:679-727): a Vec or every Vec inside a List
        const Tuple &t = base_->rows[r];
        for (size_t f : manifest_.vec_fields) {
            if (f >= t.size()) continue;
            auto push = [&](const std::vector<float> &v, int32_t sub) {
                if (v.size() != manifest_.vec_dim)
                    throw CozoError("hnsw::dim_mismatch", "vector of length " + std::to_string(v.size()) + " in an index of dimension " +
                                                              std::to_string(manifest_.vec_dim));
                flat.insert(flat.end(), v.begin(), v.end());
                nodes_.push_back({r, (uint32_t)f, sub});
            };
            if (const std::vector<float> *v = t[f].get_vec()) push(*v, -1);
            else if (const std::vector<DataValue> *l = t[f].get_slice())
                for (size_t s = 0; s < l->size(); s++)
                    if (const std::vector<float> *v2 = (*l)[s].get_vec()) push(*v2, (int32_t)s);
        }
    }
    const size_t n_new = nodes_.size() - n_before;
    if (levels && levels->size() != n_new) {
        nodes_.resize(n_before);
        throw CozoError("hnsw::bad_levels", "levels must hold one entry per new vector");
    }
    if (n_new == 0) return;
    uint64_t nd = 0;
    int rc;
    const bool shared = shares_rows(nodes_);
    if (!h_ && shared) {  // (see create)
        rc = cz_hnsw_build(nullptr, 0, (uint32_t)manifest_.vec_dim, (int)manifest_.distance, (uint32_t)manifest_.m_neighbours,
                           (uint32_t)manifest_.ef_construction, manifest_.keep_pruned_connections ? 1 : 0, nullptr, 0, 0, nullptr, &h_, 0,
                           nullptr);
        if (rc != CZ_OK) {
            nodes_.resize(n_before);
            check_gpu(rc);
        }
    }
    if (!h_)  // the first rows of an index that was empty so far
        rc = cz_hnsw_build(flat.data(), (uint32_t)n_new, (uint32_t)manifest_.vec_dim, (int)manifest_.distance, (uint32_t)manifest_.m_neighbours,
                           (uint32_t)manifest_.ef_construction, manifest_.keep_pruned_connections ? 1 : 0, levels ? levels->data() : nullptr,
                           seed, max_batch, &nd, &h_, manifest_.extend_candidates ? CZ_HNSW_EXTEND_CANDIDATES : 0u, nullptr);
    else {
        // Node ids follow insertion order, the reference's entry point follows KEY order (the first row of the index relation,
        // hnsw.rs:184-191, 891-899): hand the library every node's position among the (row key, field, sub-index) triples so
        // that a later row whose key sorts before the entry point's takes its place exactly as in the reference.
        const size_t K = base_->keys.size();
        std::vector<uint32_t> by_key(nodes_.size());
        for (uint32_t i = 0; i < by_key.size(); i++) by_key[i] = i;
        std::stable_sort(by_key.begin(), by_key.end(), [&](uint32_t a, uint32_t b) {
            const Tuple &ta = base_->rows[nodes_[a].row], &tb = base_->rows[nodes_[b].row];
            for (size_t c = 0; c < K; c++) {
                const int cmp = DataValue::compare(ta[c], tb[c]);
                if (cmp) return cmp < 0;
            }
            if (nodes_[a].field != nodes_[b].field) return nodes_[a].field < nodes_[b].field;
            return nodes_[a].sub < nodes_[b].sub;
        });
        std::vector<uint32_t> rank(nodes_.size());
        for (uint32_t pos = 0; pos < by_key.size(); pos++) rank[by_key[pos]] = pos;
        rc = cz_hnsw_set_key_order(h_, rank.data(), (uint32_t)rank.size());
        if (rc == CZ_OK && shared) {  // links inside one base row are never read (hnsw.rs:609-610): the library needs the rows
            const std::vector<uint32_t> rows = rows_of(nodes_);
            rc = cz_hnsw_set_row_of(h_, rows.data(), (uint32_t)rows.size());
        }
        if (rc == CZ_OK)
            rc = cz_hnsw_insert(h_, flat.data(), (uint32_t)n_new, (uint32_t)manifest_.m_neighbours, (uint32_t)manifest_.ef_construction,
                                manifest_.keep_pruned_connections ? 1 : 0, levels ? levels->data() : nullptr, seed, max_batch, &nd,
                                manifest_.extend_candidates ? CZ_HNSW_EXTEND_CANDIDATES : 0u, nullptr);
    }
    if (rc != CZ_OK) nodes_.resize(n_before);
    check_gpu(rc);
    build_n_dist_ += nd;
    for (auto &kv : columns_)  // the per-node copies of base columns no longer cover every node
        if (kv.second) cz_column_destroy(kv.second);
    columns_.clear();
}

void GpuHnswIndex::remove_rows(const std::vector<uint32_t> &rows) {
    if (!h_ || rows.empty()) return;
    std::vector<uint8_t> hit(base_->rows.size(), 0);
    for (uint32_t r : rows)
        if (r < hit.size()) hit[r] = 1;
    std::vector<uint32_t> ids;
    removed_.resize(nodes_.size(), 0);
    for (uint32_t v = 0; v < nodes_.size(); v++)
        if (hit[nodes_[v].row] && !removed_[v]) ids.push_back(v);
    if (ids.empty()) return;
    check_gpu(cz_hnsw_remove(h_, ids.data(), (uint32_t)ids.size()));
    for (uint32_t v : ids) removed_[v] = 1;
}

GpuHnswIndex GpuHnswIndex::from_stored(const HnswIndexManifest &manifest, const StoredRows &idx, const StoredRows &base_rows,
                                       const BaseRelation &base) {
    if (manifest.dtype != VecElementType::F32) throw GpuError(CZ_E_UNSUPPORTED, "only F32 vector indices are GPU-resident");
    if (base.rows.size() != base_rows.size()) throw CozoError("hnsw::base_mismatch", "`base` and its stored rows differ in length");
    GpuHnswIndex ix;
    ix.manifest_ = manifest;
    ix.base_ = &base;
    std::vector<uint32_t> fields(manifest.vec_fields.begin(), manifest.vec_fields.end());
    const czi_rows vi = idx.view(), vb = base_rows.view();
    czi_hnsw *h = nullptr;
    if (czi_hnsw_ingest(&vi, &vb, fields.data(), (uint32_t)fields.size(), (uint32_t)manifest.vec_dim, (int32_t)manifest.distance,
                        (uint32_t)manifest.m_max, (uint32_t)manifest.m_max0, &h) != CZI_OK)
        throw CozoError("ingest::error", std::string("libcozo_ingest: ") + czi_last_error());
    std::unique_ptr<czi_hnsw, void (*)(czi_hnsw *)> hold(h, czi_hnsw_free);
    cz_hnsw_desc desc;
    const float *vectors = nullptr;
    czi_hnsw_desc(h, &desc, &vectors);
    const uint64_t *row = nullptr;
    const uint32_t *field = nullptr;
    const int32_t *sub = nullptr;
    czi_hnsw_nodes(h, &row, &field, &sub);
    for (uint32_t v = 0; v < desc.n; v++) ix.nodes_.push_back({(uint32_t)row[v], field[v], sub[v]});
    if (desc.n_levels == 0) return ix;  // an empty index: hnsw_knn returns no rows (hnsw.rs:903-909)
    check_gpu(cz_hnsw_index_create(&desc, vectors, &ix.h_));
    return ix;
}

StoredRows GpuHnswIndex::index_rows(uint64_t relation_id) const {
    StoredRows out;
    const uint32_t K = (uint32_t)base_->keys.size();
    out.n_key_cols = 2 * K + 5;
    if (!h_) return out;
    uint32_t n = 0, dim = 0, entry = 0;
    int32_t metric = 0, n_levels = 0;
    check_gpu(cz_hnsw_index_info(h_, &n, &dim, &metric, &n_levels, &entry));
    std::vector<float> vectors((size_t)n * dim);
    check_gpu(cz_hnsw_index_export_vectors(h_, vectors.data()));
    std::vector<uint32_t> sizes(n_levels);
    std::vector<int32_t> widths(n_levels);
    std::vector<std::vector<uint32_t>> ids(n_levels), nbrs(n_levels);
    std::vector<std::vector<double>> dist(n_levels), degree(n_levels);
    for (int32_t lv = 0; lv < n_levels; lv++) {
        check_gpu(cz_hnsw_index_level_info(h_, lv, &sizes[lv], &widths[lv]));
        ids[lv].resize(sizes[lv]);
        nbrs[lv].resize((size_t)sizes[lv] * widths[lv]);
        check_gpu(cz_hnsw_index_export_level(h_, lv, ids[lv].data(), nbrs[lv].data()));
        degree[lv].resize(sizes[lv]);  // the f64 of the self rows: with extend_candidates not always the number of link rows
        check_gpu(cz_hnsw_index_export_degrees(h_, lv, degree[lv].data()));
        if (lv == 0 && !removed_.empty()) {  // a removed node keeps its id on the device and has no rows in the store (hnsw_remove, :728-868)
            uint32_t keep = 0;
            for (uint32_t r = 0; r < sizes[0]; r++) {
                const uint32_t v = ids[0][r];
                if (v < removed_.size() && removed_[v]) continue;
                ids[0][keep] = v;
                degree[0][keep] = degree[0][r];
                std::copy(nbrs[0].begin() + (size_t)r * widths[0], nbrs[0].begin() + (size_t)(r + 1) * widths[0],
                          nbrs[0].begin() + (size_t)keep * widths[0]);
                keep++;
            }
            sizes[0] = keep;
            ids[0].resize(keep);
            degree[0].resize(keep);
            nbrs[0].resize((size_t)keep * widths[0]);
        }
        // the distance column of every link row: the same arithmetic the search uses
        std::vector<uint32_t> pairs;
        std::vector<size_t> slot;
        for (uint32_t r = 0; r < sizes[lv]; r++)
            for (int32_t s = 0; s < widths[lv]; s++) {
                const uint32_t t = nbrs[lv][(size_t)r * widths[lv] + s];
                if (t == CZ_NONE) continue;
                pairs.push_back(ids[lv][r]);
                pairs.push_back(t);
                slot.push_back((size_t)r * widths[lv] + s);
            }
        dist[lv].assign(nbrs[lv].size(), 0.0);
        if (!slot.empty()) {
            std::vector<double> d(slot.size());
            check_gpu(cz_distance_batch(metric, vectors.data(), n, dim, vectors.data(), n, pairs.data(), slot.size(), d.data(), 0, nullptr));
            for (size_t i = 0; i < slot.size(); i++) dist[lv][slot[i]] = d[i];
        }
    }
    // CompoundKey columns of every node in their key encoding: [row key x K, field, sub index]
    std::vector<uint8_t> node_keys;
    std::vector<uint64_t> node_key_off{0};
    for (const CompoundKey &ck : nodes_) {
        const Tuple &t = base_->rows[ck.row];
        for (uint32_t c = 0; c < K; c++) encode_datavalue(node_keys, t[c]);
        encode_datavalue(node_keys, DataValue((int64_t)ck.field));
        encode_datavalue(node_keys, DataValue((int64_t)ck.sub));
        node_key_off.push_back(node_keys.size());
    }
    std::vector<const uint32_t *> ids_p, nbrs_p;
    std::vector<const double *> dist_p, degree_p;
    for (int32_t lv = 0; lv < n_levels; lv++) {
        ids_p.push_back(ids[lv].data());
        nbrs_p.push_back(nbrs[lv].data());
        dist_p.push_back(dist[lv].data());
        degree_p.push_back(degree[lv].data());
    }
    cz_hnsw_desc desc{n, dim, metric, n_levels, entry, sizes.data(), widths.data(), ids_p.data(), nbrs_p.data()};
    czi_row_buf *buf = nullptr;
    if (czi_hnsw_encode_rows_degrees(&desc, vectors.data(), node_keys.data(), node_key_off.data(), dist_p.data(), degree_p.data(),
                                     relation_id, &buf) != CZI_OK)
        throw CozoError("ingest::error", std::string("libcozo_ingest: ") + czi_last_error());
    std::unique_ptr<czi_row_buf, void (*)(czi_row_buf *)> hold(buf, czi_row_buf_free);
    czi_rows rows;
    czi_row_buf_rows(buf, &rows);
    out.keys.assign(rows.keys, rows.keys + rows.key_off[rows.n_rows]);
    out.vals.assign(rows.vals, rows.vals + rows.val_off[rows.n_rows]);
    out.key_off.assign(rows.key_off, rows.key_off + rows.n_rows + 1);
    out.val_off.assign(rows.val_off, rows.val_off + rows.n_rows + 1);
    return out;
}

// op_lt / op_le / op_eq / op_ge / op_gt / op_neq (data/functions.rs:298-380).  Numbers: Int with Int as integers, Float
// with Float by total order (data/value.rs:595), mixed pairs as f64.  op_eq / op_neq never check types (:298-304, :337-343):
// anything that is not a pair of numbers is compared as DataValues (Null == 5 is false, 'a' != 5 is true); the ordering
// operators call ensure_same_value_type first and fail on such a pair -- that error is kept.
static bool compare_values(const DataValue &a, int op, const DataValue &b) {
    int64_t ai = 0, bi = 0;
    double af = 0, bf = 0;
    const bool a_int = a.is_int(), b_int = b.is_int();
    if (!a.is_num() || !b.is_num()) {
        if (op == CZ_OP_EQ) return a == b;
        if (op != CZ_OP_LT && op != CZ_OP_LE && op != CZ_OP_GE && op != CZ_OP_GT) return !(a == b);
        throw CozoError("", "comparison can only be done between the same datatypes");
    }
    if (a_int) a.get_int(&ai);
    else a.get_float(&af);
    if (b_int) b.get_int(&bi);
    else b.get_float(&bf);
    int c;
    if (a_int && b_int) c = ai < bi ? -1 : (ai > bi ? 1 : 0);
    else if (!a_int && !b_int) {
        auto key = [](double d) {
            int64_t u;
            memcpy(&u, &d, 8);
            return u ^ (int64_t)((uint64_t)(u >> 63) >> 1);
        };
        const int64_t ka = key(af), kb = key(bf);
        c = ka < kb ? -1 : (ka > kb ? 1 : 0);
    } else {
        const double l = a_int ? (double)ai : af, r = b_int ? (double)bi : bf;
        switch (op) {
            case CZ_OP_LT: return l < r;
            case CZ_OP_LE: return l <= r;
            case CZ_OP_EQ: return l == r;
            case CZ_OP_GE: return l >= r;
            case CZ_OP_GT: return l > r;
            default: return l != r;
        }
    }
    switch (op) {
        case CZ_OP_LT: return c < 0;
        case CZ_OP_LE: return c <= 0;
        case CZ_OP_EQ: return c == 0;
        case CZ_OP_GE: return c >= 0;
        case CZ_OP_GT: return c > 0;
        default: return c != 0;
    }
}

cz_column *GpuHnswIndex::device_column(size_t column) const {
    auto it = columns_.find(column);
    if (it != columns_.end()) return it->second;
    cz_column *col = nullptr;
    std::vector<int64_t> iv;
    std::vector<double> fv;
    bool all_int = true, all_float = true;
    for (const CompoundKey &ck : nodes_) {
        const Tuple &t = base_->rows[ck.row];
        int64_t i = 0;
        double f = 0;
        if (column >= t.size()) {
            all_int = all_float = false;
            break;
        }
        const bool is_float = t[column].is_float() && t[column].get_float(&f);
        const bool is_int = t[column].is_int() && t[column].get_int(&i);
        all_int = all_int && is_int;
        all_float = all_float && is_float;
        if (!all_int && !all_float) break;
        if (is_int) iv.push_back(i);
        else fv.push_back(f);
    }
    if (!nodes_.empty() && all_int) check_gpu(cz_column_upload(iv.data(), (uint32_t)iv.size(), CZ_COL_I64, &col));
    else if (!nodes_.empty() && all_float) check_gpu(cz_column_upload(fv.data(), (uint32_t)fv.size(), CZ_COL_F64, &col));
    columns_[column] = col;
    return col;
}

void GpuHnswIndex::search_raw(const float *queries, uint32_t B, uint32_t k, uint32_t ef, std::vector<uint32_t> &ids,
                              std::vector<double> &dist, std::vector<uint32_t> &count, const Poison &poison,
                              const std::optional<double> &radius) const {
    ids.assign((size_t)B * k, CZ_NONE);
    dist.assign((size_t)B * k, 0.0);
    count.assign(B, 0);
    if (!h_ || B == 0) return;
    // the radius cut (`distance > r => skip`, hnsw.rs:952-956) is applied on the device to the rows that come back
    check_gpu(cz_hnsw_search_batch(h_, queries, B, k, ef, radius ? 1 : 0, radius ? *radius : 0.0, ids.data(), dist.data(),
                                   count.data(), nullptr, poison.flag_ptr(), 0, nullptr));
}

std::vector<std::vector<Tuple>> GpuHnswIndex::hnsw_knn_batch(const std::vector<const std::vector<float> *> &queries,
                                                             const HnswSearch &config, const Poison &poison) const {
    const uint32_t B = (uint32_t)queries.size();
    std::vector<std::vector<Tuple>> result(B);
    std::vector<float> q((size_t)B * manifest_.vec_dim);
    for (uint32_t i = 0; i < B; i++) {
        if (queries[i]->size() != manifest_.vec_dim)  // runtime/hnsw.rs:876-878
            throw CozoError("", "query vector dimension mismatch");
        std::copy(queries[i]->begin(), queries[i]->end(), q.begin() + (size_t)i * manifest_.vec_dim);
    }
    if (nodes_.empty() || B == 0) return result;
    // column predicates: on the device when every named column is purely Int or purely Float over the indexed rows
    std::vector<cz_predicate> dev_preds;
    bool on_device = !config.predicates.empty() && !config.filter && config.predicates.size() <= 4;
    for (const ColumnPredicate &p : config.predicates) {
        if (!on_device) break;
        cz_column *c = device_column(p.column);
        int64_t iv = 0;
        double fv = 0;
        const bool is_int = p.constant.is_int();
        if (is_int) p.constant.get_int(&iv);
        if (!c || (!is_int && !(p.constant.is_float() && p.constant.get_float(&fv))) || p.op < CZ_OP_LT || p.op > CZ_OP_NE) {
            on_device = false;
            break;
        }
        dev_preds.push_back(cz_predicate{c, p.op, is_int ? CZ_COL_I64 : CZ_COL_F64, fv, iv});
    }
    const bool host_filter = config.filter.has_value() || (!config.predicates.empty() && !on_device);
    // without a filter the candidate set is cut to k before rows are fetched; with one, all ef survive until the
    // filter has run (:943-947) -- ask the device for the same number of rows
    const uint32_t kk = (uint32_t)(host_filter ? config.ef : std::min(config.k, config.ef));
    std::vector<uint32_t> ids, count;
    std::vector<double> dist;
    if (on_device) {
        ids.assign((size_t)B * kk, CZ_NONE);
        dist.assign((size_t)B * kk, 0.0);
        count.assign(B, 0);
        check_gpu(cz_hnsw_search_filtered(h_, q.data(), B, kk, (uint32_t)config.ef, config.radius ? 1 : 0,
                                          config.radius ? *config.radius : 0.0, dev_preds.data(), (uint32_t)dev_preds.size(),
                                          ids.data(), dist.data(), count.data(), nullptr, poison.flag_ptr(), 0, nullptr));
    } else {
        search_raw(q.data(), B, kk, (uint32_t)config.ef, ids, dist, count, poison, config.radius);
    }
    for (uint32_t i = 0; i < B; i++) {
        std::vector<Tuple> &ret = result[i];
        for (uint32_t j = 0; j < count[i]; j++) {
            const double distance = dist[(size_t)i * kk + j];
            if (config.radius && distance > *config.radius) continue;  // :952-956
            const CompoundKey &ck = nodes_[ids[(size_t)i * kk + j]];
            Tuple cand = base_->rows[ck.row];  // base_handle.get(cand_key.0)
            const DataValue field_val = cand[ck.field];
            // "make sure the order is the same as in all_bindings()" (:962): field, field_idx, distance, vector
            if (config.bind_field) cand.push_back(DataValue(base_->column_name(ck.field)));
            if (config.bind_field_idx) cand.push_back(ck.sub < 0 ? DataValue() : DataValue((int64_t)ck.sub));
            if (config.bind_distance) cand.push_back(DataValue(distance));
            if (config.bind_vector) {
                if (ck.sub < 0) cand.push_back(field_val);
                else cand.push_back((*field_val.get_slice())[(size_t)ck.sub]);
            }
            if (config.filter && !(*config.filter)(cand)) continue;  // :994-998
            if (!on_device && !config.predicates.empty()) {  // the same comparisons on the host row (op_lt .. op_neq)
                bool ok = true;
                for (const ColumnPredicate &p : config.predicates) ok = ok && compare_values(cand[p.column], p.op, p.constant);
                if (!ok) continue;
            }
            ret.push_back(std::move(cand));
        }
        if (ret.size() > config.k) ret.resize(config.k);  // :1005-1006 (rows already ascending by distance)
    }
    return result;
}

std::vector<Tuple> HnswSearchRA::iter(const std::vector<Tuple> &parent, const Poison &poison) const {
    std::vector<const std::vector<float> *> queries;
    queries.reserve(parent.size());
    for (const Tuple &t : parent) {
        const std::vector<float> *v = bind_idx < t.size() ? t[bind_idx].get_vec() : nullptr;
        if (!v)  // query/ra.rs:1106-1109
            throw CozoError("", "Expected vector, got " + (bind_idx < t.size() ? t[bind_idx].to_string() : std::string("nothing")));
        queries.push_back(v);
    }
    std::vector<std::vector<Tuple>> res = index->hnsw_knn_batch(queries, hnsw_search, poison);
    std::vector<Tuple> out;
    for (size_t i = 0; i < parent.size(); i++) {
        for (Tuple &r : res[i]) {
            Tuple joined = parent[i];
            joined.insert(joined.end(), std::make_move_iterator(r.begin()), std::make_move_iterator(r.end()));
            out.push_back(std::move(joined));
        }
    }
    return out;
}

}  // namespace cozo
