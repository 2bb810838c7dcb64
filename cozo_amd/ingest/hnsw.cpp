// hnsw.cpp -- libcozo_ingest.so: the stored rows of `tbl:idx` + the base relation -> cz_hnsw_desc + vectors (czi_hnsw_*).
#include "common.hpp"

using namespace czi;

// ==================================================================================================== hnsw
struct czi_hnsw {
    uint32_t n = 0, dim = 0;
    int32_t metric = 0, n_levels = 0;
    uint32_t entry = 0;
    std::vector<float> vectors;
    std::vector<double> vectors64;  // an index over F64 vectors (czi_hnsw_ingest_f64): `vectors` stays empty
    bool f64 = false;
    std::vector<uint32_t> level_size;
    std::vector<int32_t> level_width;
    std::vector<std::vector<uint32_t>> level_nodes, level_nbrs;
    std::vector<const uint32_t *> level_nodes_p, level_nbrs_p;
    std::vector<uint64_t> base_row;
    std::vector<uint32_t> field;
    std::vector<int32_t> sub;
    uint64_t n_rows = 0, n_self = 0, n_live = 0, n_ignored = 0;
};

namespace {

struct IdxRow {
    int64_t layer;
    const uint8_t *fr, *fr_end, *to, *to_end;  // [key x K, field, sub] of either end, as raw key bytes
    const uint8_t *fr_key_end, *to_key_end;    // end of the K row-key columns inside each
    uint64_t to_hash;
    bool ignore;
};

int64_t key_int(const uint8_t *a, const uint8_t *b, const char *what) {
    if (a >= b || *a != NUM_TAG) raise(CZI_E_CORRUPT, "index row: %s is not a number", what);
    const NumVal x = mc_num(a + 1, b);
    if (!x.is_int) raise(CZI_E_CORRUPT, "index row: %s is not an integer", what);
    return x.i;
}

// one row of `tbl:idx`; returns false for the canary row (layer 1, all-Null ends; hnsw.rs:641-669)
bool parse_idx_row(const czi_rows *idx, uint64_t i, uint32_t K, IdxRow &r) {
    const Row row = row_at(idx, i);
    const uint8_t *p = row.k, *q = mc_skip(p, row.kend);
    r.layer = key_int(p, q, "layer");
    if (r.layer > 0) return false;
    p = q;
    for (int side = 0; side < 2; side++) {
        const uint8_t *start = p;
        for (uint32_t c = 0; c < K; c++) p = mc_skip(p, row.kend);
        const uint8_t *key_end = p;
        p = mc_skip(p, row.kend);
        p = mc_skip(p, row.kend);
        if (side == 0) { r.fr = start; r.fr_key_end = key_end; r.fr_end = p; }
        else { r.to = start; r.to_key_end = key_end; r.to_end = p; }
    }
    if (p != row.kend) raise(CZI_E_CORRUPT, "index row %llu: %zu bytes after the last key column", (unsigned long long)i, (size_t)(row.kend - p));
    // value [dist, hash, ignore_link]: only ignore_link matters for the link tables
    r.ignore = false;
    if (row.v && row.v < row.vend) {
        Mp m{row.v, row.vend};
        if (m.array() != 3) raise(CZI_E_CORRUPT, "index row %llu: the value is not [dist, hash, ignore_link]", (unsigned long long)i);
        m.skip();
        m.skip();
        if (mp_value_head(m) != V_BOOL) raise(CZI_E_CORRUPT, "index row %llu: ignore_link is not a bool", (unsigned long long)i);
        r.ignore = m.boolean();
    } else {
        raise(CZI_E_CORRUPT, "index row %llu has no value", (unsigned long long)i);
    }
    return true;
}

// the vector a node names, copied as T[dim] (the index' element type, VecElementType: f32 / f64) into dst; col = the field's
// column of the base row.  A vector of the other element type is not converted: hnsw_put indexes a row's vector only when its
// type is the manifest's (runtime/hnsw.rs:694-706), so such a node cannot exist in a sound index.
template <typename T>
void copy_vector(const czi_rows *base, uint64_t brow, uint32_t fld, int32_t sub_idx, uint32_t dim, T *dst,
                 std::vector<uint8_t> &scratch) {
    constexpr bool kF64 = sizeof(T) == 8;
    ColumnCursor cur(row_at(base, brow), base->n_key_cols);
    const uint8_t *a = nullptr, *b = nullptr;
    int where = 0;
    for (uint32_t c = 0; c <= fld; c++) {
        where = cur.next(a, b);
        if (where == 0) raise(CZI_E_MISSING_ROW, "base row %llu has no column %u", (unsigned long long)brow, fld);
        if (where == 2 && c < fld) cur.m.skip();
    }
    if (where == 1) {  // a key column: [LIST_TAG elements.. INIT] or VEC_TAG
        if (sub_idx >= 0) {
            if (*a != LIST_TAG) raise(CZI_E_MISSING_ROW, "base row %llu column %u is not a list", (unsigned long long)brow, fld);
            a++;
            for (int32_t s = 0; s < sub_idx; s++) {
                if (a >= b || *a == INIT_TAG) raise(CZI_E_MISSING_ROW, "base row %llu column %u has no element %d", (unsigned long long)brow, fld, sub_idx);
                a = mc_skip(a, b);
            }
        }
        if (a >= b || *a != VEC_TAG) raise(CZI_E_MISSING_ROW, "base row %llu column %u: not a vector", (unsigned long long)brow, fld);
        if (a[1] != (kF64 ? VEC_F64 : VEC_F32))
            raise(CZI_E_UNSUPPORTED, "base row %llu column %u: the vector's element type is not the index' (%s)", (unsigned long long)brow, fld, kF64 ? "F64" : "F32");
        if (be64(a + 2) != dim) raise(CZI_E_CORRUPT, "base row %llu: vector of length %llu, index dimension %u", (unsigned long long)brow, (unsigned long long)be64(a + 2), dim);
        a += 10;
        for (uint32_t d = 0; d < dim; d++) {
            if (kF64) {
                const uint64_t u = be64(a + 8 * d);
                memcpy(dst + d, &u, 8);
            } else {
                const uint32_t u = be32(a + 4 * d);
                memcpy(dst + d, &u, 4);
            }
        }
        return;
    }
    Mp &m = cur.m;
    Variant v = mp_value_head(m);
    if (sub_idx >= 0) {
        if (v != V_LIST) raise(CZI_E_MISSING_ROW, "base row %llu column %u is not a list", (unsigned long long)brow, fld);
        const uint32_t k = m.array();
        if ((uint32_t)sub_idx >= k) raise(CZI_E_MISSING_ROW, "base row %llu column %u has no element %d", (unsigned long long)brow, fld, sub_idx);
        for (int32_t s = 0; s < sub_idx; s++) m.skip();
        v = mp_value_head(m);
    }
    if (v != V_VEC) raise(CZI_E_MISSING_ROW, "base row %llu column %u: not a vector", (unsigned long long)brow, fld);
    int el;
    const uint8_t *s;
    uint32_t nb;
    mp_vec(m, el, s, nb, scratch);
    if (el != (kF64 ? 1 : 0))
        raise(CZI_E_UNSUPPORTED, "base row %llu column %u: the vector's element type is not the index' (%s)", (unsigned long long)brow, fld, kF64 ? "F64" : "F32");
    if (nb != dim * sizeof(T)) raise(CZI_E_CORRUPT, "base row %llu: vector of length %u, index dimension %u", (unsigned long long)brow, nb / (uint32_t)sizeof(T), dim);
    memcpy(dst, s, nb);
}

void ingest_hnsw(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields, uint32_t dim,
                 int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw &h) {
    check_rows(idx, "czi_hnsw_ingest(idx)");
    check_rows(base, "czi_hnsw_ingest(base)");
    if (!dim || !m_max || !m_max0) raise(CZI_E_INVALID, "dim, m_max and m_max0 must be > 0");
    if (n_fields && !vec_fields) raise(CZI_E_INVALID, "null vec_fields");
    const uint32_t K = base->n_key_cols;
    if (idx->n_rows && idx->n_key_cols != 2 * K + 5)
        raise(CZI_E_INVALID, "the index relation of a %u-key base relation has %u key columns, not %u", K, 2 * K + 5, idx->n_key_cols);
    h.dim = dim;
    h.metric = metric;
    h.n_rows = idx->n_rows;

    const uint32_t T = ingest_threads(idx->n_rows);
    WorkerError err;
    auto rethrow = [&] {
        if (err.set) {
            g_err = err.msg;
            throw Error{err.code};
        }
    };
    auto guarded_worker = [&](uint64_t order, auto &&body) {
        try {
            body();
        } catch (const Error &e) {
            err.report(order, e.code, g_err);
        } catch (const std::exception &e) {
            err.report(order, CZI_E_INVALID, e.what());
        }
    };

    // pass 1 (threads over rows): parse every row once.  Rows are in key order, so the layers never decrease and the
    // canary rows (layer 1) are a suffix.
    const uint64_t R = idx->n_rows;
    std::unique_ptr<IdxRow[]> rows(new IdxRow[R + 1]);
    parallel_for(T, [&](uint32_t t) {
        uint64_t i = R * t / T;
        guarded_worker(i, [&] {
            for (; i < R * (t + 1) / T; i++)
                if (parse_idx_row(idx, i, K, rows[i])) rows[i].to_hash = hash_bytes(rows[i].to, (size_t)(rows[i].to_end - rows[i].to));
        });
    });
    rethrow();
    uint64_t nv = 0;  // rows that are not the canary
    for (uint64_t i = 0; i < R; i++) {
        if (i && rows[i].layer < rows[i - 1].layer) raise(CZI_E_CORRUPT, "index rows are not in key order (row %llu)", (unsigned long long)i);
        if (rows[i].layer <= 0) nv = i + 1;
    }
    if (nv == 0) return;  // only the canary, or nothing: an empty index (hnsw.rs:903-909)
    const int64_t min_layer = rows[0].layer;
    if (min_layer < -62) raise(CZI_E_CORRUPT, "index has layer %lld", (long long)min_layer);
    h.n_levels = (int32_t)(-min_layer) + 1;

    // group starts: a (layer, fr) group is one node's rows on one layer
    std::unique_ptr<uint8_t[]> group_start(new uint8_t[nv + 1]);
    parallel_for(T, [&](uint32_t t) {
        for (uint64_t j = nv * t / T; j < nv * (t + 1) / T; j++) {
            const size_t len = (size_t)(rows[j].fr_end - rows[j].fr);
            group_start[j] = j == 0 || rows[j].layer != rows[j - 1].layer || (size_t)(rows[j - 1].fr_end - rows[j - 1].fr) != len ||
                             memcmp(rows[j - 1].fr, rows[j].fr, len) != 0;
        }
    });

    // pass 2: node ids = order of the `fr` groups of layer 0 (every node has its self-loop row there, hnsw.rs:630-678)
    ByteTable nodes;
    for (uint64_t j = 0; j < nv; j++) {
        if (!group_start[j] || rows[j].layer != 0) continue;
        const uint32_t before = nodes.size();
        if (nodes.find_or_insert(rows[j].fr, (size_t)(rows[j].fr_end - rows[j].fr)) != before)
            raise(CZI_E_CORRUPT, "index rows are not in key order (layer 0)");
    }
    h.n = nodes.size();
    if (!h.n) raise(CZI_E_CORRUPT, "the index has upper layers but no layer 0");

    // node -> CompoundKey -> base row -> vector (threads over nodes)
    ByteTable base_keys;
    for (uint64_t i = 0; i < base->n_rows; i++) {
        const Row r = row_at(base, i);
        const uint8_t *p = r.k;
        for (uint32_t c = 0; c < K; c++) p = mc_skip(p, r.kend);
        if (base_keys.find_or_insert(r.k, (size_t)(p - r.k)) != (uint32_t)i) raise(CZI_E_CORRUPT, "base row %llu repeats a key", (unsigned long long)i);
    }
    h.base_row.resize(h.n);
    h.field.resize(h.n);
    h.sub.resize(h.n);
    if (h.f64) h.vectors64.resize((size_t)h.n * dim);
    else h.vectors.resize((size_t)h.n * dim);
    parallel_for(T, [&](uint32_t t) {
        std::vector<uint8_t> scratch;
        uint32_t v = (uint32_t)((uint64_t)h.n * t / T);
        guarded_worker(v, [&] {
            for (; v < (uint32_t)((uint64_t)h.n * (t + 1) / T); v++) {
                const uint8_t *a = nodes.bytes.data() + nodes.off[v], *b = nodes.bytes.data() + nodes.off[v + 1];
                const uint8_t *p = a;
                for (uint32_t c = 0; c < K; c++) p = mc_skip(p, b);
                const uint8_t *q = mc_skip(p, b);
                const int64_t fld = key_int(p, q, "fr__field");
                const int64_t sb = key_int(q, b, "fr__sub_idx");
                const uint32_t br = base_keys.find(a, (size_t)(p - a));
                if (br == CZ_NONE) raise(CZI_E_MISSING_ROW, "node %u of the index has no base row (corrupted index, hnsw.rs:131-140)", v);
                bool known = n_fields == 0;
                for (uint32_t f = 0; f < n_fields; f++) known |= vec_fields[f] == (uint32_t)fld;
                if (fld < 0 || !known) raise(CZI_E_CORRUPT, "node %u: field %lld is not one of the index' vec_fields", v, (long long)fld);
                h.base_row[v] = br;
                h.field[v] = (uint32_t)fld;
                h.sub[v] = (int32_t)sb;
                if (h.f64) copy_vector(base, br, (uint32_t)fld, (int32_t)sb, dim, h.vectors64.data() + (size_t)v * dim, scratch);
                else copy_vector(base, br, (uint32_t)fld, (int32_t)sb, dim, h.vectors.data() + (size_t)v * dim, scratch);
            }
        });
    });
    rethrow();

    // pass 3a (threads over rows): what every row is -- dropped exactly as hnsw_get_neighbours drops it -- and the id of `to`
    enum : uint8_t { LIVE = 0, SELF = 1, SAME_ROW = 2, IGNORED = 3 };
    std::unique_ptr<uint8_t[]> kind(new uint8_t[nv + 1]);
    std::unique_ptr<uint32_t[]> to_id(new uint32_t[nv + 1]);
    parallel_for(T, [&](uint32_t t) {
        uint64_t j = nv * t / T;
        const uint64_t hi = nv * (t + 1) / T;
        guarded_worker(j, [&] {
            for (; j < hi; j++) {
                const IdxRow &r = rows[j];
                // the `to` lookups are random probes of the node table: keep two stages of them in flight
                if (j + 32 < hi) nodes.hint_slot(rows[j + 32].to_hash);
                if (j + 16 < hi) nodes.hint_bytes(rows[j + 16].to_hash);
                const size_t len = (size_t)(r.fr_end - r.fr), klen = (size_t)(r.fr_key_end - r.fr);
                const bool same_row = (size_t)(r.to_key_end - r.to) == klen && memcmp(r.to, r.fr, klen) == 0;
                if (same_row) {  // hnsw.rs:609-610: the self-loop row and links between vectors of one base row
                    kind[j] = ((size_t)(r.to_end - r.to) == len && memcmp(r.to, r.fr, len) == 0) ? SELF : SAME_ROW;
                } else if (r.ignore) {  // :616-619
                    kind[j] = IGNORED;
                } else {
                    kind[j] = LIVE;
                    to_id[j] = nodes.find_h(r.to, (size_t)(r.to_end - r.to), r.to_hash);
                    if (to_id[j] == CZ_NONE) raise(CZI_E_CORRUPT, "a link points at a node with no layer-0 row");
                }
            }
        });
    });
    rethrow();

    // pass 3b: per level, the nodes present (ascending id = key order) and their live rows
    const int L = h.n_levels;
    h.level_size.assign(L, 0);
    h.level_width.assign(L, 0);
    h.level_nodes.assign(L, {});
    h.level_nbrs.assign(L, {});
    std::vector<std::vector<uint64_t>> row_at_flat(L);  // per present node: where its live links start in flat[lv]
    std::vector<std::vector<uint32_t>> flat(L);         // concatenated live links per level
    for (uint64_t i = 0; i < nv;) {
        const int lv = (int)(-rows[i].layer);
        const uint32_t fr = nodes.find(rows[i].fr, (size_t)(rows[i].fr_end - rows[i].fr));
        if (fr == CZ_NONE) raise(CZI_E_CORRUPT, "a layer %lld row starts at a node with no layer-0 row", (long long)rows[i].layer);
        if (!h.level_nodes[lv].empty() && h.level_nodes[lv].back() >= fr) raise(CZI_E_CORRUPT, "index rows are not in key order");
        row_at_flat[lv].push_back(flat[lv].size());
        bool self = false;
        uint64_t j = i;
        do {
            switch (kind[j]) {
            case LIVE: flat[lv].push_back(to_id[j]); h.n_live++; break;
            case SELF: self = true; h.n_self++; break;
            case IGNORED: h.n_ignored++; break;
            default: break;
            }
            j++;
        } while (j < nv && !group_start[j]);
        if (!self) raise(CZI_E_CORRUPT, "node %u has rows on layer %lld but no self-loop row there", fr, (long long)rows[i].layer);
        h.level_nodes[lv].push_back(fr);
        i = j;
    }
    if (h.level_nodes[0].size() != h.n) raise(CZI_E_CORRUPT, "layer 0 holds %zu of %u nodes", h.level_nodes[0].size(), h.n);
    // row widths: m_max0 on level 0, m_max above -- one width for ALL upper levels (cz_hnsw_index_create keeps them in one
    // table); a live row longer than that (never written by hnsw_put_vector, which shrinks to m_max) widens its group
    uint32_t width0 = m_max0, width_up = m_max;
    for (int lv = 0; lv < L; lv++) {
        if (h.level_nodes[lv].empty()) raise(CZI_E_CORRUPT, "layer %d is empty", -lv);
        row_at_flat[lv].push_back(flat[lv].size());
        for (size_t r = 0; r < h.level_nodes[lv].size(); r++) {
            const uint32_t len = (uint32_t)(row_at_flat[lv][r + 1] - row_at_flat[lv][r]);
            if (lv == 0) width0 = std::max(width0, len); else width_up = std::max(width_up, len);
        }
    }
    for (int lv = 0; lv < L; lv++) {
        const size_t sz = h.level_nodes[lv].size();
        const uint32_t width = lv == 0 ? width0 : width_up;
        h.level_size[lv] = (uint32_t)sz;
        h.level_width[lv] = (int32_t)width;
        std::vector<uint32_t> &tab = h.level_nbrs[lv];
        tab.assign(sz * width, CZ_NONE);
        parallel_for(T, [&](uint32_t t) {
            for (size_t r = sz * t / T; r < sz * (t + 1) / T; r++) {
                // the scan yields `to` ends in key order = ascending id already; sort defensively (ids are what the kernels need)
                const uint64_t a0 = row_at_flat[lv][r], a1 = row_at_flat[lv][r + 1];
                std::copy(flat[lv].begin() + a0, flat[lv].begin() + a1, tab.begin() + r * width);
                std::sort(tab.begin() + r * width, tab.begin() + r * width + (a1 - a0));
            }
        });
    }
    h.entry = nodes.find(rows[0].fr, (size_t)(rows[0].fr_end - rows[0].fr));  // hnsw.rs:891-915
    for (int lv = 0; lv < L; lv++) {
        h.level_nodes_p.push_back(h.level_nodes[lv].data());
        h.level_nbrs_p.push_back(h.level_nbrs[lv].data());
    }
}

}  // namespace

static int ingest_any(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields, uint32_t dim,
                      int32_t metric, uint32_t m_max, uint32_t m_max0, bool f64, czi_hnsw **out) {
    if (!out) return fail(CZI_E_INVALID, "null out");
    *out = nullptr;
    std::unique_ptr<czi_hnsw> h(new (std::nothrow) czi_hnsw);
    if (!h) return fail(CZI_E_OOM, "out of host memory");
    h->f64 = f64;
    const int rc = guarded([&] { ingest_hnsw(idx, base, vec_fields, n_fields, dim, metric, m_max, m_max0, *h); });
    if (rc) return rc;
    *out = h.release();
    return CZI_OK;
}

extern "C" int czi_hnsw_ingest(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields,
                               uint32_t dim, int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw **out) {
    return ingest_any(idx, base, vec_fields, n_fields, dim, metric, m_max, m_max0, false, out);
}

extern "C" int czi_hnsw_ingest_f64(const czi_rows *idx, const czi_rows *base, const uint32_t *vec_fields, uint32_t n_fields,
                                   uint32_t dim, int32_t metric, uint32_t m_max, uint32_t m_max0, czi_hnsw **out) {
    return ingest_any(idx, base, vec_fields, n_fields, dim, metric, m_max, m_max0, true, out);
}

extern "C" void czi_hnsw_free(czi_hnsw *h) { delete h; }

extern "C" int czi_hnsw_desc(const czi_hnsw *h, cz_hnsw_desc *desc, const float **vectors) {
    if (!h || !desc || !vectors) return fail(CZI_E_INVALID, "null argument");
    desc->n = h->n;
    desc->dim = h->dim;
    desc->metric = h->metric;
    desc->n_levels = h->n_levels;
    desc->entry = h->entry;
    desc->level_size = h->level_size.data();
    desc->level_width = h->level_width.data();
    desc->level_nodes = h->level_nodes_p.data();
    desc->level_nbrs = h->level_nbrs_p.data();
    if (h->f64) return fail(CZI_E_INVALID, "an F64 index: czi_hnsw_desc_f64");
    *vectors = h->vectors.data();
    return CZI_OK;
}

extern "C" int czi_hnsw_desc_f64(const czi_hnsw *h, cz_hnsw_desc *desc, const double **vectors) {
    if (!h || !desc || !vectors) return fail(CZI_E_INVALID, "null argument");
    if (!h->f64) return fail(CZI_E_INVALID, "an F32 index: czi_hnsw_desc");
    desc->n = h->n;
    desc->dim = h->dim;
    desc->metric = h->metric;
    desc->n_levels = h->n_levels;
    desc->entry = h->entry;
    desc->level_size = h->level_size.data();
    desc->level_width = h->level_width.data();
    desc->level_nodes = h->level_nodes_p.data();
    desc->level_nbrs = h->level_nbrs_p.data();
    *vectors = h->vectors64.data();
    return CZI_OK;
}

extern "C" int czi_hnsw_nodes(const czi_hnsw *h, const uint64_t **base_row, const uint32_t **field, const int32_t **sub) {
    if (!h) return fail(CZI_E_INVALID, "null handle");
    if (base_row) *base_row = h->base_row.data();
    if (field) *field = h->field.data();
    if (sub) *sub = h->sub.data();
    return CZI_OK;
}

extern "C" int czi_hnsw_row_counts(const czi_hnsw *h, uint64_t *n_rows, uint64_t *n_self, uint64_t *n_live_links, uint64_t *n_ignored) {
    if (!h) return fail(CZI_E_INVALID, "null handle");
    if (n_rows) *n_rows = h->n_rows;
    if (n_self) *n_self = h->n_self;
    if (n_live_links) *n_live_links = h->n_live;
    if (n_ignored) *n_ignored = h->n_ignored;
    return CZI_OK;
}
