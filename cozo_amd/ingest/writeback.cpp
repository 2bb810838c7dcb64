// writeback.cpp -- libcozo_ingest.so: a flat index -> the stored rows of its `tbl:idx` relation (czi_hnsw_encode_rows).
#include "common.hpp"

using namespace czi;

// ==================================================================================================== write-back
struct czi_row_buf {
    std::vector<uint8_t> keys, vals;
    std::vector<uint64_t> key_off{0}, val_off{0};
};

namespace {

// FIPS 180-4 SHA-256 (Vector::get_hash, data/value.rs:333-348, hashes the little-endian element bytes)
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t block[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
            0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
            0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
            0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
            0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = be32(p + 4 * i);
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t *p, size_t n) {
        total += n;
        while (n) {
            const size_t take = std::min(n, 64 - fill);
            memcpy(block + fill, p, take);
            fill += take;
            p += take;
            n -= take;
            if (fill == 64) {
                compress(block);
                fill = 0;
            }
        }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)(h[i] >> 24);
            out[4 * i + 1] = (uint8_t)(h[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h[i] >> 8);
            out[4 * i + 3] = (uint8_t)h[i];
        }
    }
};

// msgpack writers for the three value columns, in rmp-serde 1.2.0's shape of the derived enums
void mp_put_str(std::vector<uint8_t> &o, const char *s) {
    const size_t n = strlen(s);  // variant names are < 32 bytes: fixstr
    o.push_back((uint8_t)(0xa0 | n));
    o.insert(o.end(), s, s + n);
}
void mp_put_variant(std::vector<uint8_t> &o, const char *name) {
    o.push_back(0x81);  // a one-entry map
    mp_put_str(o, name);
}
void mp_put_f64(std::vector<uint8_t> &o, double f) {
    mp_put_variant(o, "Num");
    mp_put_variant(o, "Float");
    uint64_t u;
    memcpy(&u, &f, 8);
    o.push_back(0xcb);
    for (int i = 7; i >= 0; i--) o.push_back((uint8_t)(u >> (8 * i)));
}
void mp_put_int(std::vector<uint8_t> &o, int64_t v) {  // the most compact form, as rmp's write_sint picks it
    mp_put_variant(o, "Num");
    mp_put_variant(o, "Int");
    if (v >= 0) {
        if (v < 128) o.push_back((uint8_t)v);
        else if (v < 256) { o.push_back(0xcc); o.push_back((uint8_t)v); }
        else if (v < 65536) { o.push_back(0xcd); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
        else if (v < 4294967296ll) { o.push_back(0xce); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
        else { o.push_back(0xcf); for (int i = 7; i >= 0; i--) o.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
    } else {
        if (v >= -32) o.push_back((uint8_t)v);
        else if (v >= -128) { o.push_back(0xd0); o.push_back((uint8_t)v); }
        else if (v >= -32768) { o.push_back(0xd1); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
        else if (v >= -2147483648ll) { o.push_back(0xd2); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
        else { o.push_back(0xd3); for (int i = 7; i >= 0; i--) o.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
    }
}
void mp_put_bytes(std::vector<uint8_t> &o, const uint8_t *p, size_t n) {
    mp_put_variant(o, "Bytes");
    if (n < 256) { o.push_back(0xc4); o.push_back((uint8_t)n); }
    else if (n < 65536) { o.push_back(0xc5); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n); }
    else { o.push_back(0xc6); for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(n >> (8 * i))); }
    o.insert(o.end(), p, p + n);
}
void mp_put_bool(std::vector<uint8_t> &o, bool b) {
    mp_put_variant(o, "Bool");
    o.push_back(b ? 0xc3 : 0xc2);
}
void put_prefix(std::vector<uint8_t> &o, uint64_t relation_id) {
    for (int i = 7; i >= 0; i--) o.push_back((uint8_t)(relation_id >> (8 * i)));
}

void encode_index_rows(const cz_hnsw_desc *d, const float *vectors, const uint8_t *nk, const uint64_t *nko,
                       const double *const *level_dist, const double *const *level_degree, uint64_t rid, czi_row_buf &out) {
    if (d->n_levels <= 0 || d->n == 0) return;
    const uint32_t n = d->n;
    // key order of the nodes: rows of one layer are ordered by the `fr` bytes, then by the `to` bytes
    std::vector<uint32_t> order(n), rank(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    auto less = [&](uint32_t x, uint32_t y) {
        const size_t lx = nko[x + 1] - nko[x], ly = nko[y + 1] - nko[y];
        const int c = memcmp(nk + nko[x], nk + nko[y], std::min(lx, ly));
        return c ? c < 0 : lx < ly;
    };
    std::sort(order.begin(), order.end(), less);
    for (uint32_t r = 0; r < n; r++) rank[order[r]] = r;
    for (uint32_t r = 1; r < n; r++)
        if (!less(order[r - 1], order[r])) raise(CZI_E_INVALID, "nodes %u and %u have the same key", order[r - 1], order[r]);
    if (d->entry >= n) raise(CZI_E_INVALID, "entry %u of %u nodes", d->entry, n);

    // the two value shapes are constant up to the number / the hash: build them once, patch per row
    std::vector<uint8_t> link_val, self_val;
    put_prefix(link_val, rid);
    link_val.push_back(0x93);  // [dist, hash, ignore_link]
    mp_put_f64(link_val, 0.0);
    const size_t link_num_at = link_val.size() - 8;
    mp_put_str(link_val, "Null");
    mp_put_bool(link_val, false);
    const uint8_t zero_hash[32] = {0};
    put_prefix(self_val, rid);
    self_val.push_back(0x93);
    mp_put_f64(self_val, 0.0);
    const size_t self_num_at = self_val.size() - 8;
    mp_put_bytes(self_val, zero_hash, 32);
    const size_t self_hash_at = self_val.size() - 32;
    mp_put_bool(self_val, false);
    auto patch_f64 = [](uint8_t *at, double f) {
        uint64_t u;
        memcpy(&u, &f, 8);
        u = __builtin_bswap64(u);
        memcpy(at, &u, 8);
    };

    // pass 1: validate, and lay the output out exactly.  An item = one node on one level (its self-loop row + its link
    // rows); items in output order: top layer first (most negative layer = smallest key), nodes by key order.
    struct Item {
        int lv;
        uint32_t r, fr;            // row in the level's tables, node id
        uint64_t row0, key0, val0;  // where its rows start in the output
    };
    std::vector<Item> items;
    uint64_t n_rows = 0, key_bytes = 0, val_bytes = 0;
    std::vector<std::pair<uint32_t, uint32_t>> present;  // (rank of node, row in the level's tables)
    for (int lv = d->n_levels - 1; lv >= 0; lv--) {
        const uint32_t sz = d->level_size[lv], width = (uint32_t)d->level_width[lv];
        const uint32_t *ids = d->level_nodes[lv], *tab = d->level_nbrs[lv];
        if (!tab || (lv > 0 && !ids)) raise(CZI_E_INVALID, "level %d: null table", lv);
        present.clear();
        for (uint32_t r = 0; r < sz; r++) {
            const uint32_t fr = ids ? ids[r] : r;
            if (fr >= n) raise(CZI_E_INVALID, "level %d names node %u of %u", lv, fr, n);
            present.push_back({rank[fr], r});
        }
        std::sort(present.begin(), present.end());
        for (const auto &pr : present) {
            const uint32_t r = pr.second, fr = ids ? ids[r] : r;
            const uint64_t flen = nko[fr + 1] - nko[fr];
            items.push_back({lv, r, fr, n_rows, key_bytes, val_bytes});
            n_rows++;
            key_bytes += 18 + 2 * flen;
            val_bytes += self_val.size();
            for (uint32_t s = 0; s < width; s++) {
                const uint32_t t = tab[(size_t)r * width + s];
                if (t == CZ_NONE) continue;
                if (t >= n) raise(CZI_E_INVALID, "level %d: link to node %u of %u", lv, t, n);
                if (t == fr) raise(CZI_E_INVALID, "level %d: node %u links to itself", lv, fr);
                n_rows++;
                key_bytes += 18 + flen + (nko[t + 1] - nko[t]);
                val_bytes += link_val.size();
            }
        }
    }
    // the canary row: layer 1, every other key column Null; the number of those = the columns of two CompoundKeys
    size_t cols = 0;
    for (const uint8_t *p = nk + nko[d->entry], *e = nk + nko[d->entry + 1]; p < e; p = mc_skip(p, e)) cols++;
    Buf target;  // the entry's self-row key with a Null layer (hnsw.rs:641-656)
    put_prefix(target.b, rid);
    target.u8(NULL_TAG);
    for (int side = 0; side < 2; side++) target.raw(nk + nko[d->entry], (size_t)(nko[d->entry + 1] - nko[d->entry]));
    std::vector<uint8_t> canary_val;
    put_prefix(canary_val, rid);
    canary_val.push_back(0x93);
    mp_put_int(canary_val, -(int64_t)(d->n_levels - 1));
    mp_put_bytes(canary_val, target.b.data(), target.b.size());
    mp_put_bool(canary_val, false);
    const uint64_t canary_row = n_rows, canary_key = key_bytes, canary_valat = val_bytes;
    n_rows++;
    key_bytes += 18 + 2 * cols;
    val_bytes += canary_val.size();

    out.keys.resize(key_bytes);
    out.vals.resize(val_bytes);
    out.key_off.resize(n_rows + 1);
    out.val_off.resize(n_rows + 1);
    out.key_off[n_rows] = key_bytes;
    out.val_off[n_rows] = val_bytes;

    const uint32_t T = ingest_threads(n_rows);
    // SHA-256 of every vector (Vector::get_hash): threads over nodes
    std::vector<uint8_t> hashes((size_t)n * 32);
    parallel_for(T, [&](uint32_t t) {
        for (uint32_t v = (uint32_t)((uint64_t)n * t / T); v < (uint32_t)((uint64_t)n * (t + 1) / T); v++) {
            Sha256 sha;
            sha.update((const uint8_t *)(vectors + (size_t)v * d->dim), (size_t)d->dim * 4);  // host is little-endian
            sha.finish(hashes.data() + (size_t)v * 32);
        }
    });
    std::vector<std::array<uint8_t, 10>> layer_keys(d->n_levels);
    for (int lv = 0; lv < d->n_levels; lv++) {
        Buf k;
        k.num_int(-(int64_t)lv);  // |layer| < 2^53: 10 bytes
        memcpy(layer_keys[lv].data(), k.b.data(), 10);
    }
    // pass 2: write (threads over items; every item knows where its rows go)
    parallel_for(T, [&](uint32_t t) {
        struct Link {
            uint32_t rank, to;
            double dist;
        };
        std::vector<Link> links;
        uint8_t head[18];  // relation id + the layer column
        for (int i = 0; i < 8; i++) head[i] = (uint8_t)(rid >> (56 - 8 * i));
        for (size_t it = items.size() * t / T; it < items.size() * (t + 1) / T; it++) {
            const Item &item = items[it];
            const uint32_t width = (uint32_t)d->level_width[item.lv], fr = item.fr;
            const uint32_t *tab = d->level_nbrs[item.lv];
            const double *dist = level_dist ? level_dist[item.lv] : nullptr;
            memcpy(head + 8, layer_keys[item.lv].data(), 10);
            uint8_t *kw = out.keys.data() + item.key0, *vw = out.vals.data() + item.val0;
            uint64_t row = item.row0;
            auto begin_row = [&] {
                out.key_off[row] = (uint64_t)(kw - out.keys.data());
                out.val_off[row] = (uint64_t)(vw - out.vals.data());
                row++;
            };
            auto put_key = [&](uint32_t to) {
                memcpy(kw, head, 18);
                kw += 18;
                memcpy(kw, nk + nko[fr], nko[fr + 1] - nko[fr]);
                kw += nko[fr + 1] - nko[fr];
                memcpy(kw, nk + nko[to], nko[to + 1] - nko[to]);
                kw += nko[to + 1] - nko[to];
            };
            links.clear();
            for (uint32_t s = 0; s < width; s++) {
                const uint32_t to = tab[(size_t)item.r * width + s];
                if (to != CZ_NONE) links.push_back({rank[to], to, dist ? dist[(size_t)item.r * width + s] : 0.0});
            }
            std::sort(links.begin(), links.end(), [](const Link &a, const Link &b) { return a.rank < b.rank; });
            bool self_done = false;
            auto put_self = [&] {
                begin_row();
                put_key(fr);
                memcpy(vw, self_val.data(), self_val.size());
                // the degree: the link rows, unless the caller knows better (cz_hnsw_index_export_degrees: an
                // extend_candidates shrink leaves it one above them, hnsw.rs:413-433 + 352-357)
                patch_f64(vw + self_num_at, level_degree && level_degree[item.lv] ? level_degree[item.lv][item.r] : (double)links.size());
                memcpy(vw + self_hash_at, hashes.data() + (size_t)fr * 32, 32);
                vw += self_val.size();
                self_done = true;
            };
            for (const Link &l : links) {
                if (!self_done && rank[fr] < l.rank) put_self();
                begin_row();
                put_key(l.to);
                memcpy(vw, link_val.data(), link_val.size());
                patch_f64(vw + link_num_at, l.dist);
                vw += link_val.size();
            }
            if (!self_done) put_self();
        }
    });
    {
        uint8_t *kw = out.keys.data() + canary_key;
        for (int i = 0; i < 8; i++) kw[i] = (uint8_t)(rid >> (56 - 8 * i));
        Buf k;
        k.num_int(1);
        memcpy(kw + 8, k.b.data(), 10);
        memset(kw + 18, NULL_TAG, 2 * cols);
        memcpy(out.vals.data() + canary_valat, canary_val.data(), canary_val.size());
        out.key_off[canary_row] = canary_key;
        out.val_off[canary_row] = canary_valat;
    }
}

}  // namespace

extern "C" int czi_hnsw_encode_rows(const cz_hnsw_desc *desc, const float *vectors, const uint8_t *node_keys,
                                    const uint64_t *node_key_off, const double *const *level_dist, uint64_t relation_id,
                                    czi_row_buf **out) {
    return czi_hnsw_encode_rows_degrees(desc, vectors, node_keys, node_key_off, level_dist, nullptr, relation_id, out);
}

extern "C" int czi_hnsw_encode_rows_degrees(const cz_hnsw_desc *desc, const float *vectors, const uint8_t *node_keys,
                                            const uint64_t *node_key_off, const double *const *level_dist,
                                            const double *const *level_degree, uint64_t relation_id, czi_row_buf **out) {
    if (!out) return fail(CZI_E_INVALID, "null out");
    *out = nullptr;
    if (!desc || (desc->n && (!vectors || !node_keys || !node_key_off))) return fail(CZI_E_INVALID, "null argument");
    std::unique_ptr<czi_row_buf> b(new (std::nothrow) czi_row_buf);
    if (!b) return fail(CZI_E_OOM, "out of host memory");
    const int rc = guarded([&] { encode_index_rows(desc, vectors, node_keys, node_key_off, level_dist, level_degree, relation_id, *b); });
    if (rc) return rc;
    *out = b.release();
    return CZI_OK;
}

extern "C" int czi_row_buf_rows(const czi_row_buf *b, czi_rows *rows) {
    if (!b || !rows) return fail(CZI_E_INVALID, "null argument");
    rows->keys = b->keys.data();
    rows->key_off = b->key_off.data();
    rows->vals = b->vals.data();
    rows->val_off = b->val_off.data();
    rows->n_rows = b->key_off.size() - 1;
    rows->n_key_cols = 0;
    return CZI_OK;
}

extern "C" void czi_row_buf_free(czi_row_buf *b) { delete b; }
