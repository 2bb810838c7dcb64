// graph.cpp -- libcozo_ingest.so: the stored rows of an edge relation -> dense ids + CSR (czi_graph_*), and the two
// library-wide entry points czi_last_error / czi_version.
#include "common.hpp"

using namespace czi;

// ==================================================================================================== graph
struct czi_graph {
    ByteTable nodes;  // one-thread path: the table; threaded path: only its bytes / off (node keys in id order) are filled
    // threaded path: the endpoints are partitioned by hash, one table per partition (local ids) + local -> global id
    std::vector<ByteTable> parts;
    std::vector<std::vector<uint32_t>> gid;
    std::vector<uint32_t> src, dst;
    std::vector<float> w;
    bool weighted = false, undirected = false;
};

extern "C" const char *czi_last_error(void) { return g_err.c_str(); }
extern "C" const char *czi_version(void) { return "cozo_ingest 0.1 (memcmp keys + rmp-serde 1.2 values of cozo 0.7.6)"; }

namespace {

struct End {
    const uint8_t *a;
    size_t len;
    uint64_t h;
};

// columns 0 and 1 of row i as memcmp byte slices (+ hash), column 2 as the weight when asked for (fixed_rule/mod.rs:146-262)
inline void parse_edge_row(const czi_rows *rel, uint64_t i, End ends[2], Buf tmp[2], std::vector<uint8_t> &scratch, bool weighted,
                           bool allow_negative_weights, float *w) {
    ColumnCursor cur(row_at(rel, i), rel->n_key_cols);
    for (int c = 0; c < 2; c++) {
        const uint8_t *a, *b;
        const int where = cur.next(a, b);
        if (where == 0) raise(CZI_E_NOT_AN_EDGE, "The relation cannot be interpreted as an edge");  // mod.rs:846-850
        if (where == 2) {  // the endpoint lives in the value part: its memcmp form is its identity
            tmp[c].b.clear();
            mp_to_memcmp(cur.m, tmp[c], scratch);
            a = tmp[c].b.data();
            b = a + tmp[c].b.size();
        }
        ends[c].a = a;
        ends[c].len = (size_t)(b - a);
        ends[c].h = hash_bytes(a, ends[c].len);
    }
    if (!weighted) return;
    const uint8_t *a, *b;
    const int where = cur.next(a, b);
    double f = 1.0;
    if (where == 1) {
        if (*a != NUM_TAG) raise(CZI_E_BAD_WEIGHT, "row %llu: the value cannot be interpreted as an edge weight", (unsigned long long)i);
        f = mc_num(a + 1, b).f;
    } else if (where == 2) {
        if (mp_value_head(cur.m) != V_NUM) raise(CZI_E_BAD_WEIGHT, "row %llu: the value cannot be interpreted as an edge weight", (unsigned long long)i);
        f = mp_num(cur.m).f;
    }
    if (where != 0 && (!std::isfinite(f) || (f < 0.0 && !allow_negative_weights)))
        raise(CZI_E_BAD_WEIGHT, "row %llu: the value %g cannot be interpreted as an edge weight", (unsigned long long)i, f);
    *w = (float)f;
}

// One thread.  Rows are handled in blocks: parse a block (endpoint slices + weights), hash every endpoint and start its two
// cache lines moving (slot, then candidate bytes), and only then resolve the endpoints IN ROW ORDER -- the
// first-appearance numbering is untouched, the lookups of a block overlap instead of queueing behind each other.
void assign_ids_serial(const czi_rows *rel, czi_graph &g, bool allow_negative_weights) {
    const uint64_t E = rel->n_rows;
    constexpr uint32_t kBlock = 32;
    End ends[2 * kBlock];
    Buf tmp[2 * kBlock];
    std::vector<uint8_t> scratch;
    float unused = 0;
    for (uint64_t i0 = 0; i0 < E; i0 += kBlock) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(kBlock, E - i0);
        for (uint32_t r = 0; r < nb; r++) {
            parse_edge_row(rel, i0 + r, ends + 2 * r, tmp + 2 * r, scratch, g.weighted, allow_negative_weights,
                           g.weighted ? &g.w[i0 + r] : &unused);
            g.nodes.hint_slot(ends[2 * r].h);
            g.nodes.hint_slot(ends[2 * r + 1].h);
        }
        for (uint32_t x = 0; x < 2 * nb; x++) g.nodes.hint_bytes(ends[x].h);
        for (uint32_t r = 0; r < nb; r++) {
            g.src[i0 + r] = g.nodes.find_or_insert_h(ends[2 * r].a, ends[2 * r].len, ends[2 * r].h);
            g.dst[i0 + r] = g.nodes.find_or_insert_h(ends[2 * r + 1].a, ends[2 * r + 1].len, ends[2 * r + 1].h);
        }
    }
}


// T threads.  The first-appearance numbering looks sequential but is not: the id of a node is the number of FIRST
// appearances before its own, and whether an endpoint is a first appearance only depends on the endpoints with the same
// hash partition before it.  So: (A) rows in parallel: parse, hash, weights; (B) partitions in parallel: every thread
// walks the hash array in order, resolves the endpoints of ITS partition in its own table (local ids in order of first
// appearance) and marks first appearances in a bitmap over endpoint positions; (C) global id = rank of the first position
// in that bitmap (a prefix popcount); (D) rows in parallel: local -> global.  Same ids as the one-thread path, bit for bit.
void assign_ids_threaded(const czi_rows *rel, czi_graph &g, bool allow_negative_weights, uint32_t T) {
    const uint64_t E = rel->n_rows;
    const uint32_t P = T;
    WorkerError err;
    const bool trace = getenv("CZI_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[czi] assign_ids %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // per-endpoint scratch, deliberately NOT zero-filled: the threads that write it take the page faults, in parallel
    std::unique_ptr<uint64_t[]> hashes(new uint64_t[2 * E + 1]);
    std::unique_ptr<const uint8_t *[]> eptr(new const uint8_t *[2 * E + 1]);  // the endpoint's memcmp bytes: inside the caller's key
    std::unique_ptr<uint32_t[]> elen(new uint32_t[2 * E + 1]);  // buffer, or (value-part endpoints) in the parsing thread's arena
    std::unique_ptr<uint32_t[]> loc(new uint32_t[2 * E + 1]);
    struct Arena {
        std::vector<std::unique_ptr<uint8_t[]>> chunks;
        size_t left = 0;
        uint8_t *at = nullptr;
        const uint8_t *keep(const uint8_t *p, size_t n) {
            if (n > left) {
                const size_t sz = std::max<size_t>(n, 1u << 20);
                chunks.emplace_back(new uint8_t[sz]);
                at = chunks.back().get();
                left = sz;
            }
            memcpy(at, p, n);
            const uint8_t *r = at;
            at += n;
            left -= n;
            return r;
        }
    };
    std::vector<Arena> arenas(T);
    // (A)
    parallel_for(T, [&](uint32_t t) {
        const uint64_t lo = E * t / T, hi = E * (t + 1) / T;
        End ends[2];
        Buf tmp[2];
        std::vector<uint8_t> scratch;
        float unused = 0;
        uint64_t i = lo;
        try {
            for (; i < hi; i++) {
                parse_edge_row(rel, i, ends, tmp, scratch, g.weighted, allow_negative_weights, g.weighted ? &g.w[i] : &unused);
                for (int c = 0; c < 2; c++) {
                    if (ends[c].len > 0xFFFFFFFFull) raise(CZI_E_TOO_LARGE, "a node value of %zu bytes", ends[c].len);
                    hashes[2 * i + c] = ends[c].h;
                    elen[2 * i + c] = (uint32_t)ends[c].len;
                    eptr[2 * i + c] = ends[c].a == tmp[c].b.data() ? arenas[t].keep(ends[c].a, ends[c].len) : ends[c].a;
                }
            }
        } catch (const Error &e) {
            err.report(i, e.code, g_err);
        } catch (const std::exception &e) {
            err.report(i, CZI_E_INVALID, e.what());
        }
    });
    if (err.set) {
        g_err = err.msg;
        throw Error{err.code};
    }
    lap("(A) parse + hash");
    // (B)
    g.parts.resize(P);
    g.gid.resize(P);
    std::vector<uint64_t> first_bits((2 * E + 63) / 64 + 1, 0);
    std::vector<std::vector<uint64_t>> first_pos(P);
    parallel_for(T, [&](uint32_t p) {
        ByteTable &tab = g.parts[p];
        constexpr uint32_t kBlock = 64;
        uint64_t pend[kBlock];
        uint32_t np = 0;
        try {
            auto flush = [&] {
                for (uint32_t x = 0; x < np; x++) tab.hint_bytes(hashes[pend[x]]);
                for (uint32_t x = 0; x < np; x++) {
                    const uint64_t j = pend[x];
                    const uint32_t before = tab.size();
                    const uint32_t id = tab.find_or_insert_h(eptr[j], elen[j], hashes[j]);
                    loc[j] = id;
                    if (id == before) {  // a first appearance
                        first_pos[p].push_back(j);
                        __atomic_fetch_or(&first_bits[j >> 6], 1ull << (j & 63), __ATOMIC_RELAXED);
                    }
                }
                np = 0;
            };
            for (uint64_t j = 0; j < 2 * E; j++) {
                if (part_of(hashes[j], P) != p) continue;
                tab.hint_slot(hashes[j]);
                __builtin_prefetch(eptr[j]);
                pend[np++] = j;
                if (np == kBlock) flush();
            }
            flush();
        } catch (const Error &e) {
            err.report(0, e.code, g_err);
        } catch (const std::exception &e) {
            err.report(0, CZI_E_INVALID, e.what());
        }
    });
    if (err.set) {
        g_err = err.msg;
        throw Error{err.code};
    }
    lap("(B) partitions resolve");
    // (C) rank of every first position
    std::vector<uint32_t> word_rank(first_bits.size());
    uint64_t total = 0;
    for (size_t wd = 0; wd < first_bits.size(); wd++) {
        word_rank[wd] = (uint32_t)total;
        total += (uint64_t)__builtin_popcountll(first_bits[wd]);
    }
    if (total >= 0xFFFFFFFEull) raise(CZI_E_TOO_LARGE, "more than 2^32 - 2 distinct nodes");
    const uint32_t n = (uint32_t)total;
    std::vector<uint64_t> &off = g.nodes.off;
    off.assign((size_t)n + 1, 0);
    parallel_for(T, [&](uint32_t p) {
        const ByteTable &tab = g.parts[p];
        g.gid[p].resize(tab.size());
        for (uint32_t l = 0; l < tab.size(); l++) {
            const uint64_t pos = first_pos[p][l];
            const uint32_t id = word_rank[pos >> 6] + (uint32_t)__builtin_popcountll(first_bits[pos >> 6] & ((1ull << (pos & 63)) - 1));
            g.gid[p][l] = id;
            off[(size_t)id + 1] = tab.off[l + 1] - tab.off[l];  // lengths first, prefix sum below
        }
    });
    for (uint32_t v = 0; v < n; v++) off[v + 1] += off[v];
    g.nodes.bytes.resize(off[n]);
    parallel_for(T, [&](uint32_t p) {
        const ByteTable &tab = g.parts[p];
        for (uint32_t l = 0; l < tab.size(); l++)
            memcpy(g.nodes.bytes.data() + off[g.gid[p][l]], tab.bytes.data() + tab.off[l], tab.off[l + 1] - tab.off[l]);
    });
    lap("(C) ranks + node keys");
    // (D)
    parallel_for(T, [&](uint32_t t) {
        for (uint64_t i = E * t / T; i < E * (t + 1) / T; i++) {
            g.src[i] = g.gid[part_of(hashes[2 * i], P)][loc[2 * i]];
            g.dst[i] = g.gid[part_of(hashes[2 * i + 1], P)][loc[2 * i + 1]];
        }
    });
    lap("(D) local -> global");
}


// CsrLayout::Sorted: lists ascending by target, parallel edges kept, ties in input order.  The entries are sorted as
// 64-bit (source << 32 | target) keys by a stable LSD radix sort over just the bits two node ids need: every pass
// streams the arrays through 2^11 write cursors, where a counting sort over N buckets would pay a cache miss per entry.
// Threads split every pass by position: per-thread histograms, one prefix over (digit, thread), per-thread scatter --
// stable, so the result does not depend on the thread count.
// `undirected` feeds every row a second time with the ends swapped, right after it (fixed_rule/mod.rs:187-191).
void build_csr(const czi_graph &g, bool inverse, uint32_t *offsets, uint32_t *targets, float *weights) {
    const uint32_t n = g.nodes.size();
    const uint64_t rows = g.src.size();
    const uint64_t e = g.undirected ? rows * 2 : rows;
    const std::vector<uint32_t> &A = inverse ? g.dst : g.src, &B = inverse ? g.src : g.dst;
    const bool carry = weights != nullptr && g.weighted;
    const uint32_t T = ingest_threads(e);
    std::unique_ptr<uint64_t[]> key(new uint64_t[e + 1]), key2(new uint64_t[e + 1]);
    std::unique_ptr<uint32_t[]> pay, pay2;  // the row a key came from (weights follow their edges through the sort)
    if (carry) {
        pay.reset(new uint32_t[e + 1]);
        pay2.reset(new uint32_t[e + 1]);
    }
    int id_bits = 1;
    while (id_bits < 32 && (1ull << id_bits) < (uint64_t)n) id_bits++;
    constexpr int kRadix = 11;
    struct Digit {
        int shift, bits;
    };
    std::vector<Digit> digits;  // target bits first (least significant), then source bits
    for (int half = 0; half < 2; half++)
        for (int done = 0; done < id_bits; done += kRadix) digits.push_back({32 * half + done, std::min(kRadix, id_bits - done)});
    const size_t D = digits.size();
    // cnt[t][d][x]: how many keys of thread t's slice have value x in digit d
    std::vector<std::vector<uint64_t>> cnt(T, std::vector<uint64_t>(D << kRadix, 0));
    parallel_for(T, [&](uint32_t t) {
        uint64_t *c = cnt[t].data();
        // digit 0 always; with one thread the slice is the whole array in every pass, so all digits can be counted now
        const size_t Dnow = T == 1 ? D : 1;
        auto tally = [&](uint64_t k) {
            for (size_t d = 0; d < Dnow; d++) c[(d << kRadix) + ((k >> digits[d].shift) & ((1ull << digits[d].bits) - 1))]++;
        };
        for (uint64_t r = rows * t / T; r < rows * (t + 1) / T; r++) {
            const uint64_t a = A[r], b = B[r];
            if (g.undirected) {
                key[2 * r] = a << 32 | b;
                key[2 * r + 1] = b << 32 | a;
                tally(key[2 * r]);
                tally(key[2 * r + 1]);
                if (carry) pay[2 * r] = pay[2 * r + 1] = (uint32_t)r;
            } else {
                key[r] = a << 32 | b;
                tally(key[r]);
                if (carry) pay[r] = (uint32_t)r;
            }
        }
    });
    // a thread's slice of the ENTRIES is the image of its slice of the rows, in every pass (positions, not values)
    auto lo_of = [&](uint32_t t) { return (rows * t / T) * (g.undirected ? 2 : 1); };
    for (size_t d = 0; d < D; d++) {
        const int shift = digits[d].shift;
        const uint64_t m = (1ull << digits[d].bits) - 1;
        if (d > 0 && T > 1) {  // digit 0 was counted while the keys were built; the slices of later passes are known only now
            parallel_for(T, [&](uint32_t t) {
                uint64_t *c = cnt[t].data() + (d << kRadix);
                std::fill(c, c + m + 1, 0);
                for (uint64_t j = lo_of(t); j < lo_of(t + 1); j++) c[(key[j] >> shift) & m]++;
            });
        }
        uint64_t sum = 0;
        for (uint64_t x = 0; x <= m; x++)
            for (uint32_t t = 0; t < T; t++) {
                uint64_t &c = cnt[t][(d << kRadix) + x];
                const uint64_t v = c;
                c = sum;
                sum += v;
            }
        parallel_for(T, [&](uint32_t t) {
            uint64_t *c = cnt[t].data() + (d << kRadix);
            if (carry) {
                for (uint64_t j = lo_of(t); j < lo_of(t + 1); j++) {
                    const uint64_t at = c[(key[j] >> shift) & m]++;
                    key2[at] = key[j];
                    pay2[at] = pay[j];
                }
            } else {
                for (uint64_t j = lo_of(t); j < lo_of(t + 1); j++) key2[c[(key[j] >> shift) & m]++] = key[j];
            }
        });
        key.swap(key2);
        if (carry) pay.swap(pay2);
    }
    // sorted by (source, target): targets are the low halves; offsets[v] = first position whose source is >= v
    parallel_for(T, [&](uint32_t t) {
        const uint64_t lo = e * t / T, hi = e * (t + 1) / T;
        for (uint64_t j = lo; j < hi; j++) {
            targets[j] = (uint32_t)key[j];
            if (weights) weights[j] = carry ? g.w[pay[j]] : 1.0f;
            const uint32_t sj = (uint32_t)(key[j] >> 32);
            const uint32_t prev = j ? (uint32_t)(key[j - 1] >> 32) + 1 : 0;
            for (uint32_t v = prev; v <= sj; v++) offsets[v] = (uint32_t)j;  // empty when the source repeats
        }
    });
    const uint32_t last = e ? (uint32_t)(key[e - 1] >> 32) + 1 : 0;
    for (uint64_t v = last; v <= n; v++) offsets[v] = (uint32_t)e;
}

}  // namespace

extern "C" int czi_graph_ingest(const czi_rows *rel, uint32_t flags, czi_graph **out) {
    if (!out) return fail(CZI_E_INVALID, "null out");
    *out = nullptr;
    std::unique_ptr<czi_graph> g(new (std::nothrow) czi_graph);
    if (!g) return fail(CZI_E_OOM, "out of host memory");
    const int rc = guarded([&] {
        check_rows(rel, "czi_graph_ingest");
        g->weighted = (flags & CZI_WEIGHTED) != 0;
        g->undirected = (flags & CZI_UNDIRECTED) != 0;
        const bool allow_negative_weights = (flags & CZI_ALLOW_NEGATIVE_WEIGHTS) != 0;
        const uint64_t E = rel->n_rows;
        if ((g->undirected ? E * 2 : E) >= 0xFFFFFFFFull) raise(CZI_E_TOO_LARGE, "%llu rows do not fit u32 CSR offsets", (unsigned long long)E);
        g->src.resize(E);
        g->dst.resize(E);
        if (g->weighted) g->w.resize(E);
        const uint32_t T = ingest_threads(E);
        if (T > 1) assign_ids_threaded(rel, *g, allow_negative_weights, T);
        else assign_ids_serial(rel, *g, allow_negative_weights);
        if (flags & CZI_ORDERED_IDS) {
            const std::vector<uint32_t> rank = g->nodes.relabel_by_rank(g->src, g->dst);
            for (std::vector<uint32_t> &m : g->gid)
                for (uint32_t &x : m) x = rank[x];
        }
    });
    if (rc) return rc;
    *out = g.release();
    return CZI_OK;
}

extern "C" void czi_graph_free(czi_graph *g) { delete g; }
extern "C" uint32_t czi_graph_node_count(const czi_graph *g) { return g ? g->nodes.size() : 0; }
extern "C" uint64_t czi_graph_edge_count(const czi_graph *g) { return g ? (uint64_t)g->src.size() * (g->undirected ? 2 : 1) : 0; }

extern "C" int czi_graph_csr(const czi_graph *g, int inverse, uint32_t *offsets, uint32_t *targets, float *weights) {
    if (!g || !offsets || (!targets && !g->src.empty())) return fail(CZI_E_INVALID, "null argument");
    return guarded([&] { build_csr(*g, inverse != 0, offsets, targets, weights); });
}

extern "C" int czi_graph_node_keys(const czi_graph *g, const uint8_t **bytes, const uint64_t **off) {
    if (!g || !bytes || !off) return fail(CZI_E_INVALID, "null argument");
    *bytes = g->nodes.bytes.data();
    *off = g->nodes.off.data();
    return CZI_OK;
}

extern "C" uint32_t czi_graph_lookup(const czi_graph *g, const uint8_t *key, uint64_t len) {
    if (!g || (!key && len)) return CZ_NONE;
    if (g->parts.empty()) return g->nodes.find(key, (size_t)len);
    const uint64_t h = hash_bytes(key, (size_t)len);
    const uint32_t p = part_of(h, (uint32_t)g->parts.size());
    const uint32_t l = g->parts[p].find_h(key, (size_t)len, h);
    return l == CZ_NONE ? CZ_NONE : g->gid[p][l];
}
