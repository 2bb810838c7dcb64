/*
 * cozo_oracle.c -- CPU ORACLE (test infrastructure, see cozo_oracle.h).
 *
 * Restates, function by function, the reference hot path of cozodb/cozo v0.7.6
 * (file:line citations are relative to cozo-core/src/).  Must be compiled with
 * -ffp-contract=off: Rust never contracts a*b+c into an fma, and the "GPU order"
 * mode uses explicit fmaf() where the HIP kernels use v_fma_f32.
 *
 * Third-party arithmetic that lives outside /root/reference (restated from the
 * published crates, pinned versions from Cargo.lock):
 *   ndarray 0.15.6  numeric_util::unrolled_dot   (8 accumulators, see orc_dot_ndarray)
 *   graph   0.3.1   page_rank                    (GAP-style pull PageRank; Jacobi contrib refresh = orc_pagerank, the in-place
 *                                                 reading = orc_pagerank_mode(ORC_PR_INPLACE): which one the crate is, only a run of it can say)
 *   priority-queue 1.4.0 / ordered-float 4.2.0   (pop order among EQUAL priorities is
 *        implementation-defined there; this oracle breaks ties by node id -- documented deviation)
 */
#include "cozo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Distances
 * ---------------------------------------------------------------------------------------- */

/* ndarray 0.15.6 src/numeric_util.rs `unrolled_dot` (call sites runtime/hnsw.rs:70-71,81-83,99):
 * eight running products p0..p7 over blocks of 8, combined (p0+p4),(p1+p5),(p2+p6),(p3+p7)
 * left to right into `sum`, then the <8 tail added sequentially.  No fma. */
float orc_dot_ndarray(const float *a, const float *b, size_t n) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int j = 0; j < 8; j++) p[j] = p[j] + a[i + j] * b[i + j];
    }
    float sum = 0.0f;
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (; i < n; i++) sum = sum + a[i] * b[i];
    return sum;
}

/* Summation tree of the HIP kernels (cozo_amd/csrc/distance.h): a vector of `dim` f32 is cut into
 * 16-byte chunks; LPV lanes (16/32/64, smallest power of two >= #chunks) own chunks lane, lane+LPV, ...;
 * each lane runs one fma chain over its elements in address order; lanes are combined by an xor
 * butterfly with offsets LPV/2 ... 1.  Zero padding participates (fma(0,0,acc)). */
static int gpu_lpv(int dim) {
    int chunks = (dim + 3) / 4;
    int lpv = 16;
    while (lpv < chunks && lpv < 64) lpv <<= 1;
    return lpv;
}
static float gpu_butterfly(float *p, int lpv) {
    float t[64];
    for (int off = lpv / 2; off >= 1; off >>= 1) {
        for (int i = 0; i < lpv; i++) t[i] = p[i] + p[i ^ off];
        memcpy(p, t, sizeof(float) * (size_t)lpv);
    }
    return p[0];
}
float orc_dot_gpu(const float *a, const float *b, int dim) {
    int chunks = (dim + 3) / 4, lpv = gpu_lpv(dim);
    float p[64];
    memset(p, 0, sizeof p);
    for (int c = 0; c < chunks; c++) {
        int lane = c % lpv;
        for (int e = 0; e < 4; e++) {
            int idx = 4 * c + e;
            float x = idx < dim ? a[idx] : 0.0f, y = idx < dim ? b[idx] : 0.0f;
            p[lane] = fmaf(x, y, p[lane]);
        }
    }
    return gpu_butterfly(p, lpv);
}
float orc_l2_gpu(const float *a, const float *b, int dim) {
    int chunks = (dim + 3) / 4, lpv = gpu_lpv(dim);
    float p[64];
    memset(p, 0, sizeof p);
    for (int c = 0; c < chunks; c++) {
        int lane = c % lpv;
        for (int e = 0; e < 4; e++) {
            int idx = 4 * c + e;
            float x = idx < dim ? a[idx] : 0.0f, y = idx < dim ? b[idx] : 0.0f;
            float d = x - y;
            p[lane] = fmaf(d, d, p[lane]);
        }
    }
    return gpu_butterfly(p, lpv);
}

/* VectorCache::dist, runtime/hnsw.rs:66-109 (F32 arms).  a = query side (v1), b = stored (v2).
 *   L2     : diff = a - b (f32 array); diff.dot(diff) as f64          (:68-72, squared, no sqrt)
 *   Cosine : 1 - dot/sqrt(a_norm*b_norm), every dot f32 widened first (:79-85)
 *   IP     : 1 - dot as f64                                            (:97-101) */
/* ORC_DOT_SEQ: one k-ordered fmaf chain per dot product, starting from +0 -- what v_mfma_f32_32x32x2_f32 computes
 * per output element (the GEMM form of the exhaustive scan, cozo_amd/csrc/knn_gemm.hip).  Cosine / IP only. */
float orc_dot_seq(const float *a, const float *b, int dim) {
    float acc = 0.0f;
    for (int i = 0; i < dim; i++) acc = fmaf(a[i], b[i], acc);
    return acc;
}

double orc_distance(int metric, int dot_mode, const float *a, const float *b, int dim) {
    if (dot_mode == ORC_DOT_SEQ && metric != ORC_L2) {
        const double d = (double)orc_dot_seq(a, b, dim);
        if (metric == ORC_COSINE) return 1.0 - d / sqrt((double)orc_dot_seq(a, a, dim) * (double)orc_dot_seq(b, b, dim));
        return 1.0 - d;
    }
    if (metric == ORC_L2) {
        if (dot_mode == ORC_DOT_GPU) return (double)orc_l2_gpu(a, b, dim);
        float stackbuf[2048];
        float *diff = dim <= 2048 ? stackbuf : (float *)malloc(sizeof(float) * (size_t)dim);
        for (int i = 0; i < dim; i++) diff[i] = a[i] - b[i];
        double r = (double)orc_dot_ndarray(diff, diff, (size_t)dim);
        if (diff != stackbuf) free(diff);
        return r;
    }
    if (metric == ORC_COSINE) {
        double an, bn, d;
        if (dot_mode == ORC_DOT_GPU) {
            an = (double)orc_dot_gpu(a, a, dim);
            bn = (double)orc_dot_gpu(b, b, dim);
            d = (double)orc_dot_gpu(a, b, dim);
        } else {
            an = (double)orc_dot_ndarray(a, a, (size_t)dim);
            bn = (double)orc_dot_ndarray(b, b, (size_t)dim);
            d = (double)orc_dot_ndarray(a, b, (size_t)dim);
        }
        return 1.0 - d / sqrt(an * bn);
    }
    float d = dot_mode == ORC_DOT_GPU ? orc_dot_gpu(a, b, dim) : orc_dot_ndarray(a, b, (size_t)dim);
    return 1.0 - (double)d;
}

void orc_distance_pairs(int metric, int dot_mode, const float *base, const float *queries, int dim,
                        const uint32_t *pairs, uint64_t P, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)P; i++) {
        out[i] = orc_distance(metric, dot_mode, queries + (size_t)pairs[2 * i] * dim,
                              base + (size_t)pairs[2 * i + 1] * dim, dim);
    }
}

/* ---- VectorCache::dist, the F64 arms (runtime/hnsw.rs:73-78 L2, 86-95 Cosine, 102-106 IP) -----------------------------------
 * ndarray 0.15.6 unrolled_dot is generic over the element type: the same eight running products, in f64. */
double orc_dot_ndarray_f64(const double *a, const double *b, size_t n) {
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int j = 0; j < 8; j++) p[j] = p[j] + a[i + j] * b[i + j];
    }
    double sum = 0.0;
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (; i < n; i++) sum = sum + a[i] * b[i];
    return sum;
}
/* the HIP kernels' tree (cozo_amd/csrc/distance_f64.h): 16-byte chunks of TWO doubles; LPV lanes (16 / 32 / 64: the smallest
 * power of two >= the chunk count) own chunks lane, lane + LPV, ...; one fma chain per lane in address order; xor butterfly. */
static int gpu_lpv_f64(int dim) {
    int chunks = (dim + 1) / 2;
    int lpv = 16;
    while (lpv < chunks && lpv < 64) lpv <<= 1;
    return lpv;
}
static double gpu_butterfly_f64(double *p, int lpv) {
    double t[64];
    for (int off = lpv / 2; off >= 1; off >>= 1) {
        for (int i = 0; i < lpv; i++) t[i] = p[i] + p[i ^ off];
        memcpy(p, t, sizeof(double) * (size_t)lpv);
    }
    return p[0];
}
double orc_dot_gpu_f64(const double *a, const double *b, int dim) {
    int chunks = (dim + 1) / 2, lpv = gpu_lpv_f64(dim);
    double p[64];
    memset(p, 0, sizeof p);
    for (int c = 0; c < chunks; c++) {
        int lane = c % lpv;
        for (int e = 0; e < 2; e++) {
            int idx = 2 * c + e;
            double x = idx < dim ? a[idx] : 0.0, y = idx < dim ? b[idx] : 0.0;
            p[lane] = fma(x, y, p[lane]);
        }
    }
    return gpu_butterfly_f64(p, lpv);
}
static double orc_l2_gpu_f64(const double *a, const double *b, int dim) {
    int chunks = (dim + 1) / 2, lpv = gpu_lpv_f64(dim);
    double p[64];
    memset(p, 0, sizeof p);
    for (int c = 0; c < chunks; c++) {
        int lane = c % lpv;
        for (int e = 0; e < 2; e++) {
            int idx = 2 * c + e;
            double x = idx < dim ? a[idx] : 0.0, y = idx < dim ? b[idx] : 0.0;
            double d = x - y;
            p[lane] = fma(d, d, p[lane]);
        }
    }
    return gpu_butterfly_f64(p, lpv);
}
double orc_distance_f64(int metric, int dot_mode, const double *a, const double *b, int dim) {
    const int gpu = dot_mode == ORC_DOT_GPU;
    if (metric == ORC_L2) { /* :73-78 diff = a - b; diff.dot(&diff) */
        if (gpu) return orc_l2_gpu_f64(a, b, dim);
        double stackbuf[2048];
        double *diff = dim <= 2048 ? stackbuf : (double *)malloc(sizeof(double) * (size_t)dim);
        for (int i = 0; i < dim; i++) diff[i] = a[i] - b[i];
        double r = orc_dot_ndarray_f64(diff, diff, (size_t)dim);
        if (diff != stackbuf) free(diff);
        return r;
    }
    if (metric == ORC_COSINE) { /* :86-95 */
        const double an = gpu ? orc_dot_gpu_f64(a, a, dim) : orc_dot_ndarray_f64(a, a, (size_t)dim);
        const double bn = gpu ? orc_dot_gpu_f64(b, b, dim) : orc_dot_ndarray_f64(b, b, (size_t)dim);
        const double d = gpu ? orc_dot_gpu_f64(a, b, dim) : orc_dot_ndarray_f64(a, b, (size_t)dim);
        return 1.0 - d / sqrt(an * bn);
    }
    return 1.0 - (gpu ? orc_dot_gpu_f64(a, b, dim) : orc_dot_ndarray_f64(a, b, (size_t)dim)); /* :102-106 */
}
void orc_distance_pairs_f64(int metric, int dot_mode, const double *base, const double *queries, int dim, const uint32_t *pairs,
                            uint64_t P, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)P; i++)
        out[i] = orc_distance_f64(metric, dot_mode, queries + (size_t)pairs[2 * i] * dim, base + (size_t)pairs[2 * i + 1] * dim, dim);
}

/* ------------------------------------------------------------------------------------------
 * (dist, id) priority queues.  OrderedFloat: NaN sorts greatest and equals itself.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double d;
    uint32_t id;
} pq_item;

static int key_less(pq_item a, pq_item b) { /* total order (dist with NaN greatest, then id) */
    int an = isnan(a.d), bn = isnan(b.d);
    if (an != bn) return bn; /* a<b iff b is NaN and a is not */
    if (!an && a.d != b.d) return a.d < b.d;
    return a.id < b.id;
}
typedef struct {
    pq_item *v;
    int n, cap;
    int is_max;
} heap_t;
static void heap_init(heap_t *h, int is_max) {
    h->v = NULL;
    h->n = h->cap = 0;
    h->is_max = is_max;
}
static void heap_free(heap_t *h) {
    free(h->v);
    h->v = NULL;
    h->n = h->cap = 0;
}
static int heap_before(const heap_t *h, pq_item a, pq_item b) { return h->is_max ? key_less(b, a) : key_less(a, b); }
static void heap_push(heap_t *h, pq_item it) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->v = (pq_item *)realloc(h->v, sizeof(pq_item) * (size_t)h->cap);
    }
    int i = h->n++;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!heap_before(h, it, h->v[p])) break;
        h->v[i] = h->v[p];
        i = p;
    }
    h->v[i] = it;
}
static pq_item heap_pop(heap_t *h) {
    pq_item top = h->v[0];
    pq_item last = h->v[--h->n];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, c;
        if (l >= h->n) break;
        c = (r < h->n && heap_before(h, h->v[r], h->v[l])) ? r : l;
        if (!heap_before(h, h->v[c], last)) break;
        h->v[i] = h->v[c];
        i = c;
    }
    if (h->n > 0) h->v[i] = last;
    return top;
}

/* ------------------------------------------------------------------------------------------
 * HNSW: shared search-level over an abstract neighbour accessor
 * ---------------------------------------------------------------------------------------- */
typedef int (*nbr_fn)(const void *ctx, uint32_t node, int level, uint32_t *out);
typedef struct {
    const void *ctx;
    nbr_fn nbrs;
    const float *vecs;
    int dim, metric, dot_mode;
    const double *vecs64; /* an F64 index: the vectors, and every query pointer is a row of doubles */
    uint32_t *stamp; /* visited stamps [n] */
    uint32_t epoch;
    uint64_t n_dist;
    int max_width;
} search_env;

static double env_dist(const search_env *E, const float *q, uint32_t v) {
    if (E->vecs64) return orc_distance_f64(E->metric, E->dot_mode, (const double *)q, E->vecs64 + (size_t)v * E->dim, E->dim);
    return orc_distance(E->metric, E->dot_mode, q, E->vecs + (size_t)v * E->dim, E->dim);
}
/* hnsw_search_level, runtime/hnsw.rs:539-587.  found_nn is a max-queue carried in and out. */
static void search_level(search_env *E, const float *q, int ef, int level, heap_t *found_nn) {
    heap_t cand;
    heap_init(&cand, 0);
    E->epoch++;
    for (int i = 0; i < found_nn->n; i++) { /* :554-557 */
        E->stamp[found_nn->v[i].id] = E->epoch;
        heap_push(&cand, found_nn->v[i]);
    }
    uint32_t *nb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(E->max_width > 0 ? E->max_width : 1));
    while (cand.n > 0) { /* :559 */
        pq_item c = heap_pop(&cand);
        double furthest = found_nn->v[0].d; /* :560 peek */
        if (c.d > furthest) break;          /* :562 raw f64 compare */
        int cnt = E->nbrs(E->ctx, c.id, level, nb);
        for (int j = 0; j < cnt; j++) { /* :566 ascending key order */
            uint32_t v = nb[j];
            if (E->stamp[v] == E->epoch) continue; /* :569 */
            double nd = env_dist(E, q, v);
            E->n_dist++;
            double cf = found_nn->v[0].d;             /* :574 */
            if (found_nn->n < ef || nd < cf) {         /* :575 */
                pq_item it = {nd, v};
                heap_push(&cand, it);                  /* :576 */
                heap_push(found_nn, it);               /* :577 */
                if (found_nn->n > ef) heap_pop(found_nn); /* :578-580 */
            }
            E->stamp[v] = E->epoch; /* :582 */
        }
    }
    free(nb);
    heap_free(&cand);
}

/* ------------------------------------------------------------------------------------------
 * HNSW index construction
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t to;
    uint8_t ignore;
    double dist;
} link_t;
typedef struct {
    link_t *v;
    int n, cap;
    double degree; /* the f64 kept in the self-loop row's `dist` column, hnsw.rs:270,338-357 */
} adj_t;

struct orc_hnsw {
    int dim, metric, dot_mode;
    int m, m_max, m_max0, ef_c, extend, keep_pruned;
    uint32_t n, cap;
    float *vecs;
    int32_t *top; /* node exists on levels 0..top */
    adj_t **adj;  /* adj[node][level] */
    int max_level;
    uint32_t entry;
    uint32_t *stamp;
    uint32_t epoch;
    uint64_t n_dist;
    uint32_t *key_rank; /* position of every node's key among all keys; NULL = ids are key order (orc_hnsw_set_key_order) */
    uint32_t n_rank;
    uint32_t *row_of; /* the base row every node's vector comes from; NULL = one vector per row (orc_hnsw_set_row_of) */
    uint32_t n_row_of;
};
/* hnsw_get_neighbours drops every link whose far end lies in the SAME base row as the node asked about (key_tup == cand_key.0,
 * runtime/hnsw.rs:609-610): rows that carry several indexed vectors (a List of vectors, several vec_fields, :694-706).  Such a
 * link row is written like any other (:281-318) and counted into both degrees, but no reader ever sees it -- not the search, not
 * the extension, not the shrink (which therefore never drops it and does not count it into the new degree), not the removal. */
static int same_row(const orc_hnsw *h, uint32_t a, uint32_t b) {
    return h->row_of && a < h->n_row_of && b < h->n_row_of && h->row_of[a] == h->row_of[b];
}
/* the index relation is scanned in KEY order: its first row -- the entry point -- is the smallest key on the top layer */
static int key_before(const orc_hnsw *h, uint32_t a, uint32_t b) {
    if (!h->key_rank || a >= h->n_rank || b >= h->n_rank || h->key_rank[a] == h->key_rank[b]) return a < b;
    return h->key_rank[a] < h->key_rank[b];
}

orc_hnsw *orc_hnsw_new(int dim, int metric, int m, int ef_construction, int extend_candidates,
                       int keep_pruned_connections, int dot_mode) {
    orc_hnsw *h = (orc_hnsw *)calloc(1, sizeof(orc_hnsw));
    h->dim = dim;
    h->metric = metric;
    h->dot_mode = dot_mode;
    h->m = m;
    h->m_max = m;      /* runtime/relation.rs:1136-1151: m_max = m */
    h->m_max0 = 2 * m; /* m_max0 = 2m */
    h->ef_c = ef_construction;
    h->extend = extend_candidates;
    h->keep_pruned = keep_pruned_connections;
    h->max_level = -1;
    h->entry = ORC_NONE;
    return h;
}
void orc_hnsw_free(orc_hnsw *h) {
    if (!h) return;
    for (uint32_t i = 0; i < h->n; i++) {
        for (int l = 0; l <= h->top[i]; l++) free(h->adj[i][l].v);
        free(h->adj[i]);
    }
    free(h->adj);
    free(h->top);
    free(h->vecs);
    free(h->stamp);
    free(h->key_rank);
    free(h->row_of);
    free(h);
}
void orc_hnsw_set_key_order(orc_hnsw *h, const uint32_t *rank, uint32_t n) {
    free(h->key_rank);
    h->key_rank = NULL;
    h->n_rank = 0;
    if (rank && n) {
        h->key_rank = (uint32_t *)malloc(sizeof(uint32_t) * n);
        memcpy(h->key_rank, rank, sizeof(uint32_t) * n);
        h->n_rank = n;
    }
}
void orc_hnsw_set_row_of(orc_hnsw *h, const uint32_t *row_of, uint32_t n) {
    free(h->row_of);
    h->row_of = NULL;
    h->n_row_of = 0;
    if (row_of && n) {
        h->row_of = (uint32_t *)malloc(sizeof(uint32_t) * n);
        memcpy(h->row_of, row_of, sizeof(uint32_t) * n);
        h->n_row_of = n;
    }
}
static void adj_upsert(adj_t *a, uint32_t to, double dist, uint8_t ignore) { /* store_tx.put of a link row */
    int lo = 0, hi = a->n;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (a->v[mid].to < to) lo = mid + 1;
        else hi = mid;
    }
    if (lo < a->n && a->v[lo].to == to) {
        a->v[lo].dist = dist;
        a->v[lo].ignore = ignore;
        return;
    }
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 8;
        a->v = (link_t *)realloc(a->v, sizeof(link_t) * (size_t)a->cap);
    }
    memmove(a->v + lo + 1, a->v + lo, sizeof(link_t) * (size_t)(a->n - lo));
    a->v[lo].to = to;
    a->v[lo].dist = dist;
    a->v[lo].ignore = ignore;
    a->n++;
}
/* hnsw_get_neighbours(include_deleted=false), runtime/hnsw.rs:588-629: ascending `to` key, skipping
 * soft-deleted rows (the self-loop row is held separately in adj_t.degree). */
static int dyn_nbrs(const void *ctx, uint32_t node, int level, uint32_t *out) {
    const orc_hnsw *h = (const orc_hnsw *)ctx;
    if (level > h->top[node]) return 0;
    const adj_t *a = &h->adj[node][level];
    int c = 0;
    for (int i = 0; i < a->n; i++)
        if (!a->v[i].ignore && h->top[a->v[i].to] >= level && !same_row(h, node, a->v[i].to))
            out[c++] = a->v[i].to; /* (rows left dangling by a removal: see orc_hnsw_remove) */
    return c;
}
static double hdist(orc_hnsw *h, const float *a, const float *b) {
    h->n_dist++;
    return orc_distance(h->metric, h->dot_mode, a, b, h->dim);
}
static const float *hvec(const orc_hnsw *h, uint32_t id) { return h->vecs + (size_t)id * h->dim; }

/* hnsw_select_neighbours_heuristic, runtime/hnsw.rs:470-538.
 * `found` = (id, dist-to-q) items; result written to sel (<= m items, nearest-first order of acceptance). */
static int select_heuristic(orc_hnsw *h, const float *q, const pq_item *found, int nfound, int m, int level,
                            pq_item *sel) {
    heap_t cand, disc;
    heap_init(&cand, 0);
    heap_init(&disc, 0);
    for (int i = 0; i < nfound; i++) heap_push(&cand, found[i]); /* :495-498 */
    if (h->extend) { /* :499-511 ; PriorityQueue::push on an existing key keeps one entry per key */
        uint32_t *nb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(2 * h->m_max0 + 8));
        h->epoch++;
        for (int i = 0; i < nfound; i++) h->stamp[found[i].id] = h->epoch;
        for (int i = 0; i < nfound; i++) {
            int cnt = dyn_nbrs(h, found[i].id, level, nb);
            for (int j = 0; j < cnt; j++) {
                if (h->stamp[nb[j]] == h->epoch) continue; /* same key, same distance: push is a no-op */
                h->stamp[nb[j]] = h->epoch;
                pq_item it = {hdist(h, q, hvec(h, nb[j])), nb[j]};
                heap_push(&cand, it);
            }
        }
        free(nb);
    }
    int nsel = 0;
    while (cand.n > 0 && nsel < m) { /* :512 */
        pq_item c = heap_pop(&cand);
        int add = 1;
        for (int i = 0; i < nsel; i++) { /* :515-523 */
            double de = hdist(h, hvec(h, sel[i].id), hvec(h, c.id));
            if (de < c.d) {
                add = 0;
                break;
            }
        }
        if (add) sel[nsel++] = c;
        else if (h->keep_pruned) heap_push(&disc, c);
    }
    if (h->keep_pruned) { /* :530-536 */
        while (disc.n > 0 && nsel < m) sel[nsel++] = heap_pop(&disc);
    }
    heap_free(&cand);
    heap_free(&disc);
    return nsel;
}

/* hnsw_shrink_neighbour, runtime/hnsw.rs:376-469 */
static int shrink_neighbour(orc_hnsw *h, uint32_t target, int m, int level) {
    adj_t *a = &h->adj[target][level];
    int nold = 0;
    pq_item *old = (pq_item *)malloc(sizeof(pq_item) * (size_t)(a->n + 1));
    for (int i = 0; i < a->n; i++)
        if (!a->v[i].ignore && !same_row(h, target, a->v[i].to)) { /* :389-393 stored distances, not recomputed */
            old[nold].id = a->v[i].to;
            old[nold].d = a->v[i].dist;
            nold++;
        }
    pq_item *sel = (pq_item *)malloc(sizeof(pq_item) * (size_t)(m + 1));
    int nsel = select_heuristic(h, hvec(h, target), old, nold, m, level, sel);
    for (int i = 0; i < nsel; i++) { /* :413-433 new rows (only possible with extend_candidates) */
        int was_old = 0;
        for (int j = 0; j < nold; j++)
            if (old[j].id == sel[i].id) {
                was_old = 1;
                break;
            }
        /* With extend_candidates the target is reachable through its neighbours' back links, at the distance of a vector
         * to itself, and is selected like anything else.  Its "new row" [layer, target, target] IS the self row: the put
         * of :413-433 overwrites it and hnsw_put_vector puts the self row back right after this call returns (:352-357).
         * Net effect: no link row, one of the m slots spent, and the stored degree counts it (tests/literal_hnsw_store.py
         * plays the same steps on a literal row store). */
        if (!was_old && sel[i].id != target) adj_upsert(&h->adj[target][level], sel[i].id, sel[i].d, 0);
    }
    a = &h->adj[target][level];
    for (int j = 0; j < nold; j++) { /* :434-466 dropped links: soft delete (ignore_link = true) */
        int kept = 0;
        for (int i = 0; i < nsel; i++)
            if (sel[i].id == old[j].id) {
                kept = 1;
                break;
            }
        if (!kept) adj_upsert(a, old[j].id, old[j].d, 1);
    }
    free(old);
    free(sel);
    return nsel; /* :412,468 */
}

/* hnsw_put_vector, runtime/hnsw.rs:155-375 (fresh key; level drawn by the caller, :46-52) */
static void put_vector(orc_hnsw *h, uint32_t id, int target_lv /* = -target_level >= 0 */) {
    const float *q = hvec(h, id);
    h->top[id] = target_lv;
    h->adj[id] = (adj_t *)calloc((size_t)target_lv + 1, sizeof(adj_t));
    if (h->max_level < 0) { /* :360-373 first vector */
        h->max_level = target_lv;
        h->entry = id;
        return;
    }
    int bottom_lv = h->max_level; /* `bottom_level` of :195 is the TOP (most negative layer) */
    uint32_t ep = h->entry;       /* :184-199 first row of the index = smallest key on the top layer */
    heap_t found;
    heap_init(&found, 1);
    pq_item epi = {hdist(h, q, hvec(h, ep)), ep}; /* :200-204 */
    heap_push(&found, epi);
    search_env E = {h, dyn_nbrs, h->vecs, h->dim, h->metric, h->dot_mode, NULL, h->stamp, h->epoch, 0, 2 * h->m_max0 + 8};
    /* :219-229 greedy descent on layers above the target */
    for (int lv = bottom_lv; lv > target_lv; lv--) search_level(&E, q, 1, lv, &found);
    pq_item *sel = (pq_item *)malloc(sizeof(pq_item) * (size_t)(h->m_max0 + 1));
    int start_lv = target_lv < bottom_lv ? target_lv : bottom_lv; /* :242 max(target_level, bottom_level) */
    for (int lv = start_lv; lv >= 0; lv--) {
        int m_max = lv == 0 ? h->m_max0 : h->m_max; /* :243-247 */
        search_level(&E, q, h->ef_c, lv, &found);   /* :248-256 found_nn carried un-truncated */
        h->epoch = E.epoch;
        int nsel = select_heuristic(h, q, found.v, found.n, m_max, lv, sel); /* :258-267 */
        E.epoch = h->epoch;
        h->adj[id][lv].degree = (double)nsel; /* :269-277 */
        for (int i = 0; i < nsel; i++) {      /* :280-358 */
            uint32_t nb = sel[i].id;
            adj_upsert(&h->adj[id][lv], nb, sel[i].d, 0); /* out link */
            adj_upsert(&h->adj[nb][lv], id, sel[i].d, 0); /* in link  */
            int target_degree = (int)h->adj[nb][lv].degree + 1; /* :338 */
            if (target_degree > m_max) {
                target_degree = shrink_neighbour(h, nb, m_max, lv); /* :339-350 */
                E.epoch = h->epoch;
            }
            h->adj[nb][lv].degree = (double)target_degree; /* :352 */
        }
    }
    h->n_dist += E.n_dist;
    h->epoch = E.epoch;
    free(sel);
    heap_free(&found);
    if (target_lv > h->max_level) { /* :206-218 the new vector becomes the entry point */
        h->max_level = target_lv;
        h->entry = id;
    } else if (target_lv == h->max_level && key_before(h, id, h->entry)) {
        h->entry = id; /* :184-191: a row on the top layer with a smaller key is now the first row of the index */
    }
}

int orc_hnsw_insert(orc_hnsw *h, const float *vectors, uint32_t n, const int32_t *levels) {
    uint32_t need = h->n + n;
    if (need > h->cap) {
        uint32_t nc = h->cap ? h->cap : 1024;
        while (nc < need) nc *= 2;
        h->vecs = (float *)realloc(h->vecs, sizeof(float) * (size_t)nc * h->dim);
        h->top = (int32_t *)realloc(h->top, sizeof(int32_t) * nc);
        h->adj = (adj_t **)realloc(h->adj, sizeof(adj_t *) * nc);
        h->stamp = (uint32_t *)realloc(h->stamp, sizeof(uint32_t) * nc);
        memset(h->stamp + h->cap, 0, sizeof(uint32_t) * (nc - h->cap));
        h->cap = nc;
    }
    memcpy(h->vecs + (size_t)h->n * h->dim, vectors, sizeof(float) * (size_t)n * h->dim);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t id = h->n;
        h->n++;
        if (levels[i] < 0) return -1;
        put_vector(h, id, levels[i]);
    }
    return 0;
}
/* hnsw_remove_vec, runtime/hnsw.rs:754-868, on the same row model as put_vector (adj[node][level] = the link rows
 * `[layer, fr.., to..]` with their ignore_link flag, adj_t.degree = the f64 in the self row):
 *   per layer 0, -1, ... while the node's self row exists (:766-778): delete the self row; walk the node's out-rows
 *   INCLUDING the soft-deleted ones (hnsw_get_neighbours(.., include_deleted = true), :780-782); for every neighbour delete
 *   the out-row and the reverse row -- whether or not that one exists (:786-805) -- and take ONE off the neighbour's stored
 *   degree (:806-823) even when it held no live link back.
 * What the reference leaves behind: rows Z -> X of nodes Z that X did not link to (X was shrunk out of by nobody's choice
 * but Z's own shrink never ran the other way).  A later search that follows such a row fails in ensure_key ("Cannot find
 * compound key for HNSW", :133) because the base row is gone.  orc_hnsw_dangling_links counts them; the exports skip
 * them, which is the state the device's cz_hnsw_remove produces.
 * The entry point is positional (:184-191, :891-899): the smallest node of the highest layer that still has a row. */
static void adj_remove(adj_t *a, uint32_t to) {
    for (int i = 0; i < a->n; i++)
        if (a->v[i].to == to) {
            memmove(a->v + i, a->v + i + 1, sizeof(link_t) * (size_t)(a->n - i - 1));
            a->n--;
            return;
        }
}
int orc_hnsw_remove(orc_hnsw *h, uint32_t node) {
    if (node >= h->n || h->top[node] < 0) return 0; /* no self row at layer 0: nothing is indexed under this key */
    for (int lv = 0; lv <= h->top[node]; lv++) {
        adj_t *a = &h->adj[node][lv];
        for (int i = 0; i < a->n; i++) {
            const uint32_t nb = a->v[i].to;
            if (same_row(h, node, nb)) continue; /* the walk is hnsw_get_neighbours too: the row and its reverse stay behind, dead */
            if (nb < h->n && h->top[nb] >= lv) {
                adj_remove(&h->adj[nb][lv], node); /* :796-805 the reverse row, present or not */
                h->adj[nb][lv].degree -= 1.0;      /* :806-823 */
            }
        }
        free(a->v);
        a->v = NULL;
        a->n = a->cap = 0;
        a->degree = 0.0;
    }
    free(h->adj[node]);
    h->adj[node] = NULL;
    h->top[node] = -1;
    /* the first row of what is left */
    h->max_level = -1;
    h->entry = ORC_NONE;
    for (uint32_t i = 0; i < h->n; i++)
        if (h->top[i] > h->max_level || (h->top[i] >= 0 && h->top[i] == h->max_level && key_before(h, i, h->entry))) {
            h->max_level = h->top[i];
            h->entry = i;
        }
    return 1;
}
uint64_t orc_hnsw_dangling_links(const orc_hnsw *h) {
    uint64_t c = 0;
    for (uint32_t i = 0; i < h->n; i++)
        for (int l = 0; l <= h->top[i]; l++) {
            const adj_t *a = &h->adj[i][l];
            for (int k = 0; k < a->n; k++) c += h->top[a->v[k].to] < l;
        }
    return c;
}
/* stored degree of a node's self row at a level (NaN when the node has no row there) */
double orc_hnsw_degree(const orc_hnsw *h, uint32_t node, int level) {
    if (node >= h->n || h->top[node] < level) return NAN;
    return h->adj[node][level].degree;
}
uint32_t orc_hnsw_size(const orc_hnsw *h) { return h->n; }
int orc_hnsw_n_levels(const orc_hnsw *h) { return h->max_level + 1; }
uint32_t orc_hnsw_entry(const orc_hnsw *h) { return h->entry; }
uint32_t orc_hnsw_level_size(const orc_hnsw *h, int level) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < h->n; i++) c += h->top[i] >= level;
    return c;
}
int orc_hnsw_level_width(const orc_hnsw *h, int level) { return level == 0 ? h->m_max0 : h->m_max; }
void orc_hnsw_export_level(const orc_hnsw *h, int level, uint32_t *node_ids, uint32_t *nbrs) {
    int w = orc_hnsw_level_width(h, level);
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(4 * w + 64));
    uint32_t r = 0;
    for (uint32_t i = 0; i < h->n; i++) {
        if (h->top[i] < level) continue;
        node_ids[r] = i;
        const adj_t *a = &h->adj[i][level];
        int c = 0;
        for (int k = 0; k < a->n; k++)
            if (!a->v[k].ignore && c < w && h->top[a->v[k].to] >= level && !same_row(h, i, a->v[k].to))
                tmp[c++] = a->v[k].to; /* (not a row left dangling by a removal, not a link inside one base row: what readers see) */
        for (int k = 0; k < w; k++) nbrs[(size_t)r * w + k] = k < c ? tmp[k] : ORC_NONE;
        r++;
    }
    free(tmp);
}
uint64_t orc_hnsw_dist_count(const orc_hnsw *h) { return h->n_dist; }
uint64_t orc_hnsw_link_rows(const orc_hnsw *h, int include_ignored) {
    uint64_t c = 0;
    for (uint32_t i = 0; i < h->n; i++)
        for (int l = 0; l <= h->top[i]; l++) {
            const adj_t *a = &h->adj[i][l];
            for (int k = 0; k < a->n; k++) c += include_ignored || !a->v[k].ignore;
        }
    return c;
}

/* ------------------------------------------------------------------------------------------
 * HNSW search over the flat layout
 * ---------------------------------------------------------------------------------------- */
static int flat_nbrs(const void *ctx, uint32_t node, int level, uint32_t *out) {
    const orc_flat_index *ix = (const orc_flat_index *)ctx;
    int w = ix->level_width[level];
    size_t row;
    if (level == 0 && ix->level_nodes[0] == NULL) row = node;
    else { /* binary search of the ascending node list */
        const uint32_t *ids = ix->level_nodes[level];
        uint32_t lo = 0, hi = ix->level_size[level];
        while (lo < hi) {
            uint32_t mid = lo + (hi - lo) / 2;
            if (ids[mid] < node) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= ix->level_size[level] || ids[lo] != node) return 0;
        row = lo;
    }
    const uint32_t *r = ix->level_nbrs[level] + row * (size_t)w;
    int c = 0;
    for (int k = 0; k < w; k++)
        if (r[k] != ORC_NONE) out[c++] = r[k];
    return c;
}

/* hnsw_knn, runtime/hnsw.rs:869-1012 (no filter bytecode: the caller asks for k = ef when it filters) */
static int knn_one(const orc_flat_index *ix, const float *q, int k, int ef, int has_radius, double radius,
                   uint32_t *out_ids, double *out_dist, uint64_t *n_dist, uint32_t *stamp, uint32_t *epoch) {
    if (ix->n_levels <= 0 || ix->n == 0) return 0; /* :900-909,1009-1011 */
    int maxw = 1;
    for (int l = 0; l < ix->n_levels; l++)
        if (ix->level_width[l] > maxw) maxw = ix->level_width[l];
    search_env E = {ix, flat_nbrs, ix->vectors, ix->dim, ix->metric, ix->dot_mode, ix->vectors64, stamp, *epoch, 0, maxw};
    heap_t found;
    heap_init(&found, 1);
    pq_item epi = {env_dist(&E, q, ix->entry), ix->entry}; /* :915-918 */
    E.n_dist++;
    heap_push(&found, epi);
    for (int lv = ix->n_levels - 1; lv > 0; lv--) search_level(&E, q, 1, lv, &found); /* :919-929 */
    search_level(&E, q, ef, 0, &found);                                               /* :930-938 */
    while (found.n > k) heap_pop(&found);                                             /* :943-947 */
    int cnt = 0;
    while (found.n > 0) { /* :951-1004 farthest first */
        pq_item it = heap_pop(&found);
        if (has_radius && it.d > radius) continue; /* :952-956 */
        out_ids[cnt] = it.id;
        out_dist[cnt] = it.d;
        cnt++;
    }
    for (int i = 0; i < cnt / 2; i++) { /* :1005 reverse */
        uint32_t ti = out_ids[i];
        out_ids[i] = out_ids[cnt - 1 - i];
        out_ids[cnt - 1 - i] = ti;
        double td = out_dist[i];
        out_dist[i] = out_dist[cnt - 1 - i];
        out_dist[cnt - 1 - i] = td;
    }
    if (cnt > k) cnt = k; /* :1006 */
    heap_free(&found);
    *epoch = E.epoch;
    if (n_dist) *n_dist += E.n_dist;
    return cnt;
}
int orc_hnsw_knn(const orc_flat_index *ix, const float *q, int k, int ef, int has_radius, double radius,
                 uint32_t *out_ids, double *out_dist, uint64_t *n_dist) {
    uint32_t *stamp = (uint32_t *)calloc(ix->n ? ix->n : 1, sizeof(uint32_t));
    uint32_t epoch = 0;
    int c = knn_one(ix, q, k, ef, has_radius, radius, out_ids, out_dist, n_dist, stamp, &epoch);
    free(stamp);
    return c;
}
void orc_hnsw_knn_batch(const orc_flat_index *ix, const float *queries, uint32_t B, int k, int ef, int has_radius,
                        double radius, uint32_t *out_ids, double *out_dist, uint32_t *out_count,
                        uint64_t *n_dist_total, int threads) {
    uint64_t total = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : total)
    {
        uint32_t *stamp = (uint32_t *)calloc(ix->n ? ix->n : 1, sizeof(uint32_t));
        uint32_t epoch = 0;
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            uint64_t nd = 0;
            for (int j = 0; j < k; j++) {
                out_ids[(size_t)b * k + j] = ORC_NONE;
                out_dist[(size_t)b * k + j] = INFINITY;
            }
            if (epoch > 0xFFFF0000u) {
                memset(stamp, 0, sizeof(uint32_t) * ix->n);
                epoch = 0;
            }
            out_count[b] = (uint32_t)knn_one(ix, queries + (size_t)b * ix->dim * (ix->vectors64 ? 2 : 1), k, ef, has_radius, radius,
                                             out_ids + (size_t)b * k, out_dist + (size_t)b * k, &nd, stamp, &epoch);
            total += nd;
        }
        free(stamp);
    }
    if (n_dist_total) *n_dist_total = total;
}

void orc_bruteforce_knn(int metric, int dot_mode, const float *base, uint32_t n, int dim, const float *queries,
                        uint32_t B, int k, uint32_t *out_ids, double *out_dist, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        heap_t h;
        heap_init(&h, 1);
        const float *q = queries + (size_t)b * dim;
        for (uint32_t i = 0; i < n; i++) {
            pq_item it = {orc_distance(metric, dot_mode, q, base + (size_t)i * dim, dim), i};
            if (h.n < k) heap_push(&h, it);
            else if (key_less(it, h.v[0])) {
                heap_pop(&h);
                heap_push(&h, it);
            }
        }
        int cnt = h.n;
        for (int j = cnt - 1; j >= 0; j--) {
            pq_item it = heap_pop(&h);
            out_ids[(size_t)b * k + j] = it.id;
            out_dist[(size_t)b * k + j] = it.d;
        }
        for (int j = cnt; j < k; j++) {
            out_ids[(size_t)b * k + j] = ORC_NONE;
            out_dist[(size_t)b * k + j] = INFINITY;
        }
        heap_free(&h);
    }
}

/* ------------------------------------------------------------------------------------------
 * relation -> graph
 * ---------------------------------------------------------------------------------------- */
/* as_directed_graph id assignment, fixed_rule/mod.rs:144-186: rows in scan order, `from` before `to`,
 * a new key gets id = indices.len().  (Open-addressing hash instead of the BTreeMap: same mapping.) */
uint32_t orc_assign_ids(const int64_t *from, const int64_t *to, uint64_t E, uint32_t *from_idx, uint32_t *to_idx,
                        int64_t *indices) {
    uint64_t cap = 16;
    while (cap < 4 * E + 16) cap <<= 1;
    int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * cap);
    uint32_t *vals = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    memset(vals, 0xFF, sizeof(uint32_t) * cap);
    uint32_t n = 0;
    for (uint64_t e = 0; e < E; e++) {
        for (int side = 0; side < 2; side++) {
            int64_t key = side == 0 ? from[e] : to[e];
            uint64_t hsh = (uint64_t)key * 0x9E3779B97F4A7C15ull;
            hsh ^= hsh >> 29;
            uint64_t pos = hsh & (cap - 1);
            while (vals[pos] != ORC_NONE && keys[pos] != key) pos = (pos + 1) & (cap - 1);
            if (vals[pos] == ORC_NONE) {
                keys[pos] = key;
                vals[pos] = n;
                indices[n] = key;
                n++;
            }
            if (side == 0) from_idx[e] = vals[pos];
            else to_idx[e] = vals[pos];
        }
    }
    free(keys);
    free(vals);
    return n;
}

/* GraphBuilder::csr_layout(CsrLayout::Sorted), fixed_rule/mod.rs:187-195 (graph_builder 0.4.0):
 * per-node target lists sorted ascending, parallel edges kept. */
typedef struct {
    uint32_t t;
    float w;
    uint64_t seq;
} tw_t;
static int tw_cmp(const void *a, const void *b) {
    const tw_t *x = (const tw_t *)a, *y = (const tw_t *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}
void orc_build_csr(uint32_t n, uint64_t E, const uint32_t *src, const uint32_t *dst, const float *w_in, int undirected,
                   uint64_t *off, uint32_t *tgt, float *w_out) {
    uint64_t Et = undirected ? 2 * E : E;
    memset(off, 0, sizeof(uint64_t) * ((size_t)n + 1));
    for (uint64_t e = 0; e < E; e++) {
        off[src[e] + 1]++;
        if (undirected) off[dst[e] + 1]++;
    }
    for (uint32_t i = 0; i < n; i++) off[i + 1] += off[i];
    uint64_t *cur = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    memcpy(cur, off, sizeof(uint64_t) * ((size_t)n + 1));
    tw_t *tmp = (tw_t *)malloc(sizeof(tw_t) * (Et ? Et : 1));
    uint64_t seq = 0;
    for (uint64_t e = 0; e < E; e++) {
        float w = w_in ? w_in[e] : 1.0f;
        tw_t a = {dst[e], w, seq++};
        tmp[cur[src[e]]++] = a;
        if (undirected) {
            tw_t b = {src[e], w, seq++};
            tmp[cur[dst[e]]++] = b;
        }
    }
    for (uint32_t i = 0; i < n; i++) qsort(tmp + off[i], off[i + 1] - off[i], sizeof(tw_t), tw_cmp);
    for (uint64_t e = 0; e < Et; e++) {
        tgt[e] = tmp[e].t;
        if (w_out) w_out[e] = tmp[e].w;
    }
    free(tmp);
    free(cur);
}

/* ------------------------------------------------------------------------------------------
 * PageRank: fixed_rule/algos/pagerank.rs:29-56 -> graph 0.3.1 `page_rank` (GAP pr.cc pull form)
 *   init = 1/N ; base = (1-d)/N ; contrib[v] = score[v] / out_degree(v)       (all f32)
 *   per iteration, every u: new = base + d * sum_{v in in(u)} contrib[v]       (sequential f32 sum,
 *       in-neighbours in sorted order) ; err += |new - old| accumulated in f64
 *   contrib refreshed in a separate pass (Jacobi) ; stop when err < tolerance or iter == max_iter.
 *   No dangling-mass redistribution, no renormalisation (a sink's contrib is inf but never read).
 * threads > 1: 16384-node dynamic chunks (the crate's scheduler); only the f64 error sum order varies.
 * ---------------------------------------------------------------------------------------- */
int orc_pagerank(uint32_t n, const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, float damping,
                 double tolerance, uint32_t max_iter, float *scores, uint32_t *iters_run, double *final_err,
                 int threads) {
    if (n == 0) {
        if (iters_run) *iters_run = 0;
        if (final_err) *final_err = 0;
        return 0;
    }
    if (threads < 1) threads = 1;
    float init = 1.0f / (float)n;
    float base = (1.0f - damping) / (float)n;
    float *contrib = (float *)malloc(sizeof(float) * n);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t v = 0; v < (int64_t)n; v++) {
        scores[v] = init;
        contrib[v] = init / (float)out_deg[v];
    }
    uint32_t it = 0;
    double err = 0;
    for (;;) {
        err = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16384) reduction(+ : err)
        for (int64_t u = 0; u < (int64_t)n; u++) {
            float s = 0.0f;
            for (uint64_t e = in_off[u]; e < in_off[u + 1]; e++) s = s + contrib[in_src[e]];
            float old = scores[u];
            float nw = base + damping * s;
            scores[u] = nw;
            err += fabs((double)(nw - old));
        }
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t v = 0; v < (int64_t)n; v++) contrib[v] = scores[v] / (float)out_deg[v];
        it++;
        if (err < tolerance || it == max_iter) break;
    }
    free(contrib);
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return 0;
}

/* The OTHER reading of graph 0.3.1 `page_rank` (SURVEY 8 a10's flagged uncertainty; crate source absent, Cargo.lock:1562-1565).
 * ORC_PR_JACOBI (0) is orc_pagerank above: contributions refreshed in a pass of their own, after the sweep.
 * ORC_PR_INPLACE (1): `out_scores[u] = new_score / out_degree(u)` is written INSIDE the per-node loop, right after
 *   `scores[u] = new_score`, so every node later in the same sweep already pulls the updated contribution -- on one thread
 *   (and below one 16 384-node chunk) that is a Gauss-Seidel sweep in ascending node order, deterministic; with several rayon
 *   threads it depends on the schedule and no restatement can pin it.  This mode is the ONE-THREAD execution.
 * err_f64_diff: how `error += |new - old|` is formed -- 0: the f32 difference widened to f64 (`(new - old).abs() as f64`),
 *   1: the difference of the widened values (`f64::abs(new as f64 - old as f64)`).  Scores do not depend on it; the
 *   iteration count does when the error sits near `tolerance`.
 * tests/test_ref_fixtures.py runs the reference's rows against all four combinations and reports which one they equal. */
int orc_pagerank_mode(uint32_t n, const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, float damping,
                      double tolerance, uint32_t max_iter, int mode, int err_f64_diff, float *scores, uint32_t *iters_run,
                      double *final_err) {
    if (n == 0) {
        if (iters_run) *iters_run = 0;
        if (final_err) *final_err = 0;
        return 0;
    }
    const float init = 1.0f / (float)n;
    const float base = (1.0f - damping) / (float)n;
    float *contrib = (float *)malloc(sizeof(float) * n);
    for (uint32_t v = 0; v < n; v++) {
        scores[v] = init;
        contrib[v] = init / (float)out_deg[v];
    }
    uint32_t it = 0;
    double err = 0;
    for (;;) {
        err = 0;
        for (uint32_t u = 0; u < n; u++) {
            float s = 0.0f;
            for (uint64_t e = in_off[u]; e < in_off[u + 1]; e++) s = s + contrib[in_src[e]];
            const float old = scores[u];
            const float nw = base + damping * s;
            scores[u] = nw;
            if (mode == 1) contrib[u] = nw / (float)out_deg[u];
            err += err_f64_diff ? fabs((double)nw - (double)old) : fabs((double)(nw - old));
        }
        if (mode != 1)
            for (uint32_t v = 0; v < n; v++) contrib[v] = scores[v] / (float)out_deg[v];
        it++;
        if (err < tolerance || it == max_iter) break;
    }
    free(contrib);
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return 0;
}

/* The in-place reading on T rayon threads, as ONE deterministic schedule: the crate hands out 16 384-node
 * chunks through an atomic counter; here the T threads run in lockstep -- in round k thread t owns chunk k*T + t, and the threads
 * advance node by node together (step s: thread 0's node s, thread 1's node s, ...), each writing its contribution at once.  No
 * real run follows this schedule exactly; it is a representative of "several threads, contributions refreshed inside the sweep",
 * used to state how far such a run can be from the one-thread sweep and from the Jacobi reading (bench.py pagerank.readings).
 * threads == 1 is orc_pagerank_mode(ORC_PR_INPLACE). */
int orc_pagerank_inplace_lockstep(uint32_t n, const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, float damping,
                                  double tolerance, uint32_t max_iter, uint32_t threads, uint32_t chunk, float *scores,
                                  uint32_t *iters_run, double *final_err) {
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0;
    if (n == 0) return 0;
    if (threads == 0) threads = 1;
    if (chunk == 0) chunk = 16384;
    const float init = 1.0f / (float)n;
    const float base = (1.0f - damping) / (float)n;
    float *contrib = (float *)malloc(sizeof(float) * n);
    for (uint32_t v = 0; v < n; v++) {
        scores[v] = init;
        contrib[v] = init / (float)out_deg[v];
    }
    const uint64_t n_chunks = ((uint64_t)n + chunk - 1) / chunk;
    uint32_t it = 0;
    double err = 0;
    for (;;) {
        err = 0;
        for (uint64_t c0 = 0; c0 < n_chunks; c0 += threads)
            for (uint32_t s = 0; s < chunk; s++)
                for (uint32_t t = 0; t < threads && c0 + t < n_chunks; t++) {
                    const uint64_t u64 = (c0 + t) * chunk + s;
                    if (u64 >= n) continue;
                    const uint32_t u = (uint32_t)u64;
                    float sum = 0.0f;
                    for (uint64_t e = in_off[u]; e < in_off[u + 1]; e++) sum = sum + contrib[in_src[e]];
                    const float old = scores[u];
                    const float nw = base + damping * sum;
                    scores[u] = nw;
                    contrib[u] = nw / (float)out_deg[u];
                    err += fabs((double)(nw - old));
                }
        it++;
        if (err < tolerance || it == max_iter) break;
    }
    free(contrib);
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ShortestPathBFS, fixed_rule/algos/shortest_path_bfs.rs:65-94.  FIFO queue, neighbours in sorted
 * key order, parent = first discoverer, `pending.is_empty()` only breaks the inner loop.
 * ---------------------------------------------------------------------------------------- */
void orc_shortest_path_bfs(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t start,
                           const uint32_t *goals, uint32_t n_goals, uint32_t *parent) {
    uint8_t *visited = (uint8_t *)calloc(n ? n : 1, 1);
    uint8_t *is_goal = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t *queue = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) parent[i] = ORC_NONE;
    uint32_t pending = 0;
    for (uint32_t i = 0; i < n_goals; i++)
        if (goals[i] < n && !is_goal[goals[i]]) {
            is_goal[goals[i]] = 1;
            pending++;
        }
    uint32_t head = 0, tail = 0;
    if (start < n) {
        visited[start] = 1;
        queue[tail++] = start;
    }
    while (head < tail) {
        uint32_t c = queue[head++];
        for (uint64_t e = off[c]; e < off[c + 1]; e++) {
            uint32_t t = tgt[e];
            if (visited[t]) continue;
            visited[t] = 1;
            parent[t] = c;
            if (is_goal[t]) {
                is_goal[t] = 0;
                pending--;
            }
            if (pending == 0) break;
            queue[tail++] = t;
        }
    }
    free(visited);
    free(is_goal);
    free(queue);
}

/* Bfs traversal order, fixed_rule/algos/bfs.rs:49-98 (visited/backtrace shared across starts, :43-45) */
uint32_t orc_bfs_order(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t start, uint8_t *visited,
                       uint32_t *parent, uint32_t *order) {
    if (start >= n || visited[start]) return 0;
    uint32_t *queue = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t head = 0, tail = 0, cnt = 0;
    visited[start] = 1;
    queue[tail++] = start;
    while (head < tail) {
        uint32_t c = queue[head++];
        for (uint64_t e = off[c]; e < off[c + 1]; e++) {
            uint32_t t = tgt[e];
            if (visited[t]) continue;
            visited[t] = 1;
            parent[t] = c;
            order[cnt++] = t;
            queue[tail++] = t;
        }
    }
    free(queue);
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * TarjanSccG, fixed_rule/algos/strongly_connected_components.rs:89-149, with an explicit stack.
 * Note the reference updates low[at] from low[to] (not ids[to]) whenever `to` is on the stack, AFTER
 * the recursive call returns (:134-141), and rewrites low[] of a finished component to the root id.
 * Groups = BTreeMap<low, nodes> values in ascending key order (:103-108).
 * ---------------------------------------------------------------------------------------- */
uint32_t orc_tarjan_groups(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t *grp) {
    uint32_t *ids = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t)); /* 0 = None */
    uint32_t *low = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    uint8_t *on_stack = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t *stack = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *cs_node = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t *cs_edge = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint32_t sp = 0, id = 0;
    for (uint32_t root = 0; root < n; root++) {
        if (ids[root]) continue;
        uint32_t cp = 0;
        cs_node[cp] = root;
        cs_edge[cp] = off[root];
        cp++;
        stack[sp++] = root;
        on_stack[root] = 1;
        ids[root] = low[root] = ++id;
        while (cp > 0) {
            uint32_t at = cs_node[cp - 1];
            if (cs_edge[cp - 1] < off[at + 1]) {
                uint32_t to = tgt[cs_edge[cp - 1]];
                if (!ids[to]) { /* recurse; the edge is re-examined (on_stack check) when we come back */
                    cs_node[cp] = to;
                    cs_edge[cp] = off[to];
                    cp++;
                    stack[sp++] = to;
                    on_stack[to] = 1;
                    ids[to] = low[to] = ++id;
                    continue;
                }
                if (on_stack[to] && low[to] < low[at]) low[at] = low[to];
                cs_edge[cp - 1]++;
            } else {
                if (ids[at] == low[at]) {
                    for (;;) {
                        uint32_t node = stack[--sp];
                        on_stack[node] = 0;
                        low[node] = ids[at];
                        if (node == at) break;
                    }
                }
                cp--;
            }
        }
    }
    /* rank of each distinct low value, ascending */
    uint8_t *is_root = (uint8_t *)calloc((size_t)id + 2, 1);
    for (uint32_t i = 0; i < n; i++) is_root[low[i]] = 1;
    uint32_t *rank = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)id + 2));
    uint32_t r = 0;
    for (uint32_t v = 0; v <= id; v++) {
        rank[v] = r;
        r += is_root[v];
    }
    for (uint32_t i = 0; i < n; i++) grp[i] = rank[low[i]];
    free(ids);
    free(low);
    free(on_stack);
    free(stack);
    free(cs_node);
    free(cs_edge);
    free(is_root);
    free(rank);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * dijkstra, fixed_rule/algos/shortest_path_dijkstra.rs:274-339.  f32 costs, strict `<` relaxation,
 * one queue entry per node (push_increase).  Equal-cost pop order is by node id here.
 * ---------------------------------------------------------------------------------------- */
void orc_dijkstra(uint32_t n, const uint64_t *off, const uint32_t *tgt, const float *w, uint32_t start,
                  const uint32_t *goals, uint32_t n_goals, float *dist, uint32_t *parent) {
    uint32_t *heap = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *pos = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    float *pri = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint8_t *is_goal = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t remaining = 0, hn = 0;
    for (uint32_t i = 0; i < n; i++) {
        dist[i] = INFINITY;
        parent[i] = ORC_NONE;
        pos[i] = ORC_NONE;
    }
    if (goals)
        for (uint32_t i = 0; i < n_goals; i++)
            if (goals[i] < n && !is_goal[goals[i]]) {
                is_goal[goals[i]] = 1;
                remaining++;
            }
#define HLESS(a, b) (pri[a] < pri[b] || (pri[a] == pri[b] && (a) < (b)))
#define HSWAP(i, j)                 \
    do {                            \
        uint32_t _t = heap[i];      \
        heap[i] = heap[j];          \
        heap[j] = _t;               \
        pos[heap[i]] = (uint32_t)(i); \
        pos[heap[j]] = (uint32_t)(j); \
    } while (0)
    if (start < n) {
        dist[start] = 0.0f;
        pri[start] = 0.0f;
        heap[0] = start;
        pos[start] = 0;
        hn = 1;
    }
    while (hn > 0) {
        uint32_t node = heap[0];
        float cost = pri[node];
        hn--;
        if (hn > 0) {
            heap[0] = heap[hn];
            pos[heap[0]] = 0;
            uint32_t i = 0;
            for (;;) {
                uint32_t l = 2 * i + 1, r = l + 1, c;
                if (l >= hn) break;
                c = (r < hn && HLESS(heap[r], heap[l])) ? r : l;
                if (!HLESS(heap[c], heap[i])) break;
                HSWAP(i, c);
                i = c;
            }
        }
        pos[node] = ORC_NONE;
        if (!(cost > dist[node])) {
            for (uint64_t e = off[node]; e < off[node + 1]; e++) {
                uint32_t nx = tgt[e];
                float nc = cost + w[e];
                if (nc < dist[nx]) {
                    dist[nx] = nc;
                    parent[nx] = node;
                    pri[nx] = nc;
                    uint32_t i;
                    if (pos[nx] == ORC_NONE) {
                        i = hn++;
                        heap[i] = nx;
                        pos[nx] = i;
                    } else i = pos[nx];
                    while (i > 0) {
                        uint32_t p = (i - 1) / 2;
                        if (!HLESS(heap[i], heap[p])) break;
                        HSWAP(i, p);
                        i = p;
                    }
                }
            }
            if (goals) {
                if (is_goal[node]) {
                    is_goal[node] = 0;
                    remaining--;
                }
                if (remaining == 0) break;
            }
        }
    }
#undef HLESS
#undef HSWAP
    free(heap);
    free(pos);
    free(pri);
    free(is_goal);
}

/* ---- ClusteringCoefficients (fixed_rule/algos/triangles.rs:70-110) ----
 * Restated loop for loop: for every node, `edges` = its out-neighbour list on the symmetrised graph (duplicates
 * kept); n_triangles = sum over e_src in edges of #{ e_dst in edges : e_src > e_dst and e_dst is among
 * out_neighbors(e_src) } (:84-101); cc = 2 t / (d (d - 1)) in f64, (0, 0, d) when d < 2 (:80-82, :102). */
void orc_clustering_coefficients(uint32_t n, const uint64_t *off, const uint32_t *tgt, double *cc, uint64_t *n_tri,
                                 uint32_t *degree) {
    for (uint32_t v = 0; v < n; v++) {
        const uint64_t a = off[v], b = off[v + 1];
        const uint64_t d = b - a;
        uint64_t t = 0;
        if (d >= 2) {
            for (uint64_t i = a; i < b; i++) {
                const uint32_t e_src = tgt[i];
                for (uint64_t j = a; j < b; j++) {
                    const uint32_t e_dst = tgt[j];
                    if (e_src <= e_dst) continue;
                    for (uint64_t k = off[e_src]; k < off[e_src + 1]; k++) {
                        if (tgt[k] == e_dst) {
                            t++;
                            break;
                        }
                    }
                }
            }
        }
        degree[v] = (uint32_t)d;
        n_tri[v] = t;
        cc[v] = d < 2 ? 0.0 : 2.0 * (double)t / ((double)d * ((double)d - 1.0));
    }
}


/* the same loop on a SAMPLE of the nodes (v = first, first + step, ...), stopped after max_seconds: what bench.py times as the CPU
 * baseline of ClusteringCoefficients where the whole graph would take minutes (the literal loop is cubic in a hub's degree).
 * n_tri[v] is written for the processed nodes only; returns their number, *edges = the adjacency entries of their rows. */
#include <time.h>
uint64_t orc_clustering_coefficients_sample(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t first, uint32_t step,
                                            double max_seconds, uint64_t *n_tri, uint64_t *edges) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    uint64_t done = 0, ne = 0;
    if (step == 0) step = 1;
    for (uint64_t v64 = first; v64 < n; v64 += step) {
        const uint32_t v = (uint32_t)v64;
        const uint64_t a = off[v], b = off[v + 1];
        uint64_t t = 0;
        if (b - a >= 2) {
            for (uint64_t i = a; i < b; i++) {
                const uint32_t e_src = tgt[i];
                for (uint64_t j = a; j < b; j++) {
                    const uint32_t e_dst = tgt[j];
                    if (e_src <= e_dst) continue;
                    for (uint64_t k = off[e_src]; k < off[e_src + 1]; k++) {
                        if (tgt[k] == e_dst) {
                            t++;
                            break;
                        }
                    }
                }
            }
        }
        n_tri[v] = t;
        done++;
        ne += b - a;
        if ((done & 1023u) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > max_seconds) break;
        }
    }
    if (edges) *edges = ne;
    return done;
}

/* ------------------------------------------------------------------------------------------
 * BetweennessCentrality::run, fixed_rule/algos/all_pairs_shortest_path.rs:31-95, over dijkstra_keep_ties
 * (shortest_path_dijkstra.rs:341-450).  Per start: the f32 distances of Dijkstra; back_pointers[v] = every in-edge (u, v)
 * with dist[u] + w == dist[v] in f32 (one entry per edge occurrence: parallel edges multiply paths, :371-380); ALL
 * shortest paths to every target are enumerated through them (:397-430), and every path of >= 3 nodes adds 1 / l (f32;
 * l = the number of paths to that target) to each of its middle nodes (all_pairs:57-69).  The per-start maps are then
 * added up in start order in f32 (:74-79).  The order of additions within one target is immaterial (equal addends), so
 * the result is fully determined.  Literal enumeration: exponential in ties, for small graphs only (returns -1 past
 * `max_paths` per start).  Weights must be > 0 (a zero-weight cycle makes the reference's recursion endless).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint32_t *bp_off, *bp;
    uint32_t start;
    float add;
    float *seg;
    uint32_t *chain;
    uint64_t paths, max_paths;
} orc_bc_ctx;

static void orc_bc_walk(orc_bc_ctx *c, uint32_t depth, int count_only) {
    const uint32_t last = c->chain[depth - 1];
    for (uint32_t e = c->bp_off[last]; e < c->bp_off[last + 1]; e++) {
        const uint32_t nxt = c->bp[e];
        if (c->paths > c->max_paths) return;
        c->chain[depth] = nxt;
        if (nxt == c->start) {
            c->paths++;
            if (!count_only && depth + 1 >= 3)
                for (uint32_t i = 1; i < depth; i++) c->seg[c->chain[i]] += c->add; /* the middle nodes */
        } else {
            orc_bc_walk(c, depth + 1, count_only);
        }
    }
}

int orc_betweenness(uint32_t n, const uint64_t *off, const uint32_t *tgt, const float *w, float *out, uint64_t max_paths) {
    const uint64_t E = n ? off[n] : 0;
    /* in-adjacency with weights, in out-CSR scan order */
    uint32_t *in_off = (uint32_t *)calloc((size_t)n + 2, sizeof(uint32_t));
    uint32_t *in_src = (uint32_t *)malloc(sizeof(uint32_t) * (E ? E : 1));
    float *in_w = (float *)malloc(sizeof(float) * (E ? E : 1));
    for (uint64_t e = 0; e < E; e++) in_off[tgt[e] + 1]++;
    for (uint32_t v = 0; v < n; v++) in_off[v + 1] += in_off[v];
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    memcpy(cur, in_off, sizeof(uint32_t) * ((size_t)n + 1));
    for (uint32_t u = 0; u < n; u++)
        for (uint64_t e = off[u]; e < off[u + 1]; e++) {
            in_src[cur[tgt[e]]] = u;
            in_w[cur[tgt[e]]++] = w[e];
        }
    float *dist = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t *parent = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *bp_off = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    uint32_t *bp = (uint32_t *)malloc(sizeof(uint32_t) * (E ? E : 1));
    float *seg = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t *chain = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 2));
    int rc = 0;
    for (uint32_t i = 0; i < n; i++) out[i] = 0.0f;
    for (uint32_t s = 0; s < n && rc == 0; s++) {
        orc_dijkstra(n, off, tgt, w, s, NULL, 0, dist, parent);
        uint32_t k = 0;
        for (uint32_t v = 0; v < n; v++) {
            bp_off[v] = k;
            if (v == s || !isfinite(dist[v])) continue;
            for (uint32_t e = in_off[v]; e < in_off[v + 1]; e++) {
                const uint32_t u = in_src[e];
                if (isfinite(dist[u]) && (float)(dist[u] + in_w[e]) == dist[v]) bp[k++] = u;
            }
        }
        bp_off[n] = k;
        for (uint32_t i = 0; i < n; i++) seg[i] = 0.0f;
        orc_bc_ctx c = {bp_off, bp, s, 0.0f, seg, chain, 0, max_paths};
        for (uint32_t t = 0; t < n; t++) {
            if (t == s || !isfinite(dist[t])) continue;
            c.chain[0] = t;
            c.paths = 0;
            orc_bc_walk(&c, 1, 1);
            if (c.paths > max_paths) { rc = -1; break; }
            c.add = 1.0f / (float)c.paths; /* `1. / l`, l = grp.len() as f32 */
            c.paths = 0;
            orc_bc_walk(&c, 1, 0);
        }
        for (uint32_t i = 0; i < n; i++) out[i] += seg[i];
    }
    free(in_off); free(in_src); free(in_w); free(cur); free(dist); free(parent); free(bp_off); free(bp); free(seg); free(chain);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * LabelPropagation (fixed_rule/algos/label_propagation.rs:56-109)
 * ---------------------------------------------------------------------------------------- */
static uint32_t lp_total_key(float f) { /* f32::total_cmp as an unsigned compare */
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

int orc_label_propagation_in_order(uint32_t n, const uint64_t *off, const uint32_t *tgt, const float *w, const uint32_t *order,
                                   uint32_t max_iter, uint32_t *labels) {
    /* :61 labels = 0..n; the BTreeMap<u32, f32> of :69 as a dense score array + the list of labels it holds */
    float *score = (float *)calloc(n ? n : 1, sizeof(float));
    uint8_t *present = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t *seen = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) labels[i] = i;
    int iters = 0, rc = 0;
    for (uint32_t it = 0; it < max_iter && rc == 0; it++) {
        int changed = 0;
        iters++;
        for (uint32_t oi = 0; oi < n; oi++) {
            const uint32_t node = order[oi];
            uint32_t ns = 0;
            for (uint64_t e = off[node]; e < off[node + 1]; e++) { /* :70-73, adjacency order, f32 adds starting from 0.0 */
                const uint32_t l = labels[tgt[e]];
                if (!present[l]) {
                    present[l] = 1;
                    score[l] = 0.0f;
                    seen[ns++] = l;
                }
                score[l] += w[e];
            }
            if (ns == 0) continue; /* :74-76 */
            /* :77-84: the largest score under total_cmp; candidates = the labels whose score == it */
            uint32_t best = 0;
            for (uint32_t k = 1; k < ns; k++)
                if (lp_total_key(score[seen[k]]) > lp_total_key(score[seen[best]])) best = k;
            const float max_score = score[seen[best]];
            uint32_t new_label = ORC_NONE;
            for (uint32_t k = 0; k < ns; k++)
                if (score[seen[k]] == max_score && seen[k] < new_label) new_label = seen[k]; /* `choose`: here the smallest */
            for (uint32_t k = 0; k < ns; k++) present[seen[k]] = 0;
            if (new_label == ORC_NONE) { /* max_score is NaN: nothing equals it, `choose` on an empty list -> the reference panics */
                rc = -1;
                break;
            }
            if (new_label != labels[node]) { /* :86-89 */
                changed = 1;
                labels[node] = new_label;
            }
        }
        if (!changed) break; /* :92-94 */
    }
    free(score);
    free(present);
    free(seen);
    return rc ? rc : iters;
}

static uint64_t lp_priority(uint32_t v) {
    uint32_t x = v;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return ((uint64_t)x << 32) | v;
}

uint32_t orc_lp_colouring(uint32_t n, const uint64_t *off, const uint32_t *tgt, uint32_t *colour) {
    const uint64_t E = n ? off[n] : 0;
    /* the transposed adjacency: independence is about edges in either direction */
    uint64_t *in_off = (uint64_t *)calloc((size_t)n + 2, sizeof(uint64_t));
    uint32_t *in_src = (uint32_t *)malloc(sizeof(uint32_t) * (E ? E : 1));
    for (uint64_t e = 0; e < E; e++) in_off[tgt[e] + 1]++;
    for (uint32_t v = 0; v < n; v++) in_off[v + 1] += in_off[v];
    uint64_t *cur = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    memcpy(cur, in_off, sizeof(uint64_t) * ((size_t)n + 1));
    for (uint32_t u = 0; u < n; u++)
        for (uint64_t e = off[u]; e < off[u + 1]; e++) in_src[cur[tgt[e]]++] = u;
    for (uint32_t v = 0; v < n; v++) colour[v] = ORC_NONE;
    uint32_t left = n, round = 0;
    uint8_t *take = (uint8_t *)malloc(n ? n : 1);
    while (left > 0) {
        for (uint32_t v = 0; v < n; v++) {
            take[v] = 0;
            if (colour[v] != ORC_NONE) continue;
            const uint64_t kv = lp_priority(v);
            int is_max = 1;
            for (uint64_t e = off[v]; e < off[v + 1] && is_max; e++) {
                const uint32_t u = tgt[e];
                if (u != v && colour[u] == ORC_NONE && lp_priority(u) > kv) is_max = 0;
            }
            for (uint64_t e = in_off[v]; e < in_off[v + 1] && is_max; e++) {
                const uint32_t u = in_src[e];
                if (u != v && colour[u] == ORC_NONE && lp_priority(u) > kv) is_max = 0;
            }
            take[v] = (uint8_t)is_max;
        }
        for (uint32_t v = 0; v < n; v++)
            if (take[v]) {
                colour[v] = round;
                left--;
            }
        round++;
    }
    free(in_off);
    free(in_src);
    free(cur);
    free(take);
    return round;
}
