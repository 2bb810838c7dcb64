/* oracle_shim.c -- TEST ONLY.  The graph entry points of include/cozo_gpu.h implemented with the CPU oracle, so that the
 * C++ host mirror's rule logic (options, id mapping, CSR build, row emission) can be exercised on a box without a GPU:
 * tests/cpp/test_host is linked against THIS library ahead of libcozo_gpu.so for its `rules-cpu` mode, exactly like
 * tests/util.py's OracleGraphBackend stands in for cozo_amd.graph in the Python rule tests.  Never shipped, never loaded
 * by the product. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cozo_gpu.h"
#include "cozo_oracle.h"

static const char *g_err = "";
const char *cz_last_error(void) { return g_err; }
int cz_device_count(void) { return 1; }
int cz_init(int device) { (void)device; return CZ_OK; }

static uint64_t *widen(const uint32_t *off, uint32_t n) {
    uint64_t *o = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    for (uint32_t i = 0; i <= n; i++) o[i] = off[i];
    return o;
}
static int poisoned(const volatile uint8_t *p) { return p && *p; }

int cz_pagerank(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N, uint64_t E,
                float damping, double tolerance, uint32_t max_iter, float *scores, uint32_t *iters_run, double *final_err,
                const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    if (N == 0) return CZ_OK;
    uint64_t *off = widen(in_offsets, N);
    uint32_t it = 0;
    double err = 0;
    orc_pagerank(N, off, in_sources, out_degree, damping, tolerance, max_iter, scores, &it, &err, 1);
    free(off);
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return CZ_OK;
}

int cz_pagerank_inplace(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N, uint64_t E,
                        float damping, double tolerance, uint32_t max_iter, uint32_t flags, float *scores, uint32_t *iters_run,
                        double *final_err, uint32_t *n_levels, const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    if (n_levels) *n_levels = 1;
    if (N == 0) return CZ_OK;
    uint64_t *off = widen(in_offsets, N);
    uint32_t it = 0;
    double err = 0;
    orc_pagerank_mode(N, off, in_sources, out_degree, damping, tolerance, max_iter, ORC_PR_INPLACE, (flags & CZ_PR_ERR_F64_DIFF) ? 1 : 0, scores,
                      &it, &err);
    free(off);
    if (iters_run) *iters_run = it;
    if (final_err) *final_err = err;
    return CZ_OK;
}

int cz_bfs(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
           uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, int share_visited, uint32_t *parent, uint32_t *depth,
           uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison) {
    (void)E; (void)depth;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    uint8_t *visited = (uint8_t *)calloc(N ? N : 1, 1);
    uint32_t *ord = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t s = 0; s < n_starts; s++) {
        uint32_t *par = parent + (size_t)s * N;
        for (uint32_t v = 0; v < N; v++) par[v] = CZ_NONE;
        if (order) for (uint32_t v = 0; v < N; v++) order[(size_t)s * N + v] = CZ_NONE;
        if (n_reached) n_reached[s] = 0;
        if (goals) {
            orc_shortest_path_bfs(N, off, out_targets, starts[s], goals, n_goals, par);
            continue;
        }
        if (!share_visited) memset(visited, 0, N);
        if (visited[starts[s]]) continue; /* algos/bfs.rs:52-54 */
        const uint32_t c = orc_bfs_order(N, off, out_targets, starts[s], visited, par, ord);
        if (n_reached) n_reached[s] = c;
        if (order) memcpy(order + (size_t)s * N, ord, sizeof(uint32_t) * c);
    }
    free(ord);
    free(visited);
    free(off);
    return CZ_OK;
}

/* the reference's loop (algos/bfs.rs:43-98): one `visited` / `backtrace`, a start already reached is skipped */
int cz_bfs_shared(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
                  uint32_t n_starts, uint32_t *parent, uint32_t *order, uint32_t *first, const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    uint8_t *visited = (uint8_t *)calloc(N ? N : 1, 1);
    for (uint32_t v = 0; v < N; v++) parent[v] = CZ_NONE;
    uint32_t at = 0;
    first[0] = 0;
    for (uint32_t s = 0; s < n_starts; s++) {
        if (starts[s] < N && !visited[starts[s]]) at += orc_bfs_order(N, off, out_targets, starts[s], visited, parent, order + at);
        first[s + 1] = at;
    }
    free(visited);
    free(off);
    return CZ_OK;
}

/* cz_bfs_shared_until on the CPU: a start's whole discovery sequence is handed over as one "level" (the contract allows any
 * grouping that keeps the order: the caller stops at its own `limit` inside the sequence) */
int cz_bfs_shared_until(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
                        uint32_t n_starts, cz_bfs_level_fn on_level, void *ctx, uint32_t *parent, uint32_t *order, uint32_t *first,
                        const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    uint8_t *visited = (uint8_t *)calloc(N ? N : 1, 1);
    for (uint32_t v = 0; v < N; v++) parent[v] = CZ_NONE;
    uint32_t at = 0;
    int stop = 0, rc = CZ_OK;
    first[0] = 0;
    for (uint32_t s = 0; s < n_starts; s++) {
        if (!stop && starts[s] < N && !visited[starts[s]]) {
            const uint32_t c = orc_bfs_order(N, off, out_targets, starts[s], visited, parent, order + at);
            if (c && on_level) {
                const int verdict = on_level(ctx, starts[s], order + at, c);
                if (verdict < 0) { g_err = "the level callback failed"; rc = CZ_E_INVALID; stop = 1; }
                if (verdict > 0) stop = 1;
            }
            at += c;
        }
        first[s + 1] = at;
    }
    free(visited);
    free(off);
    return rc;
}

int cz_connected_components(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E, uint32_t *group,
                            uint32_t *n_groups, const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(offsets, N);
    const uint32_t k = orc_tarjan_groups(N, off, targets, group);
    free(off);
    if (n_groups) *n_groups = k;
    return CZ_OK;
}

int cz_clustering_coefficients(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E, uint64_t *n_triangles,
                               uint32_t *degree, const volatile uint8_t *poison, uint32_t flags) {
    (void)E;
    (void)flags;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(offsets, N);
    double *cc = (double *)malloc(sizeof(double) * (N ? N : 1));
    orc_clustering_coefficients(N, off, targets, cc, n_triangles, degree);
    free(cc);
    free(off);
    return CZ_OK;
}

int cz_sssp(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
            const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent, const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    for (uint32_t s = 0; s < n_starts; s++)
        orc_dijkstra(N, off, out_targets, weights, starts[s], NULL, 0, dist + (size_t)s * N, parent + (size_t)s * N);
    free(off);
    return CZ_OK;
}

/* (the full run: a goal set only lets the device stop early; the rows the rule emits -- the goals' -- are the same) */
int cz_sssp_goals(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                  const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, float *dist, uint32_t *parent,
                  const volatile uint8_t *poison) {
    (void)goals; (void)n_goals;
    return cz_sssp(out_offsets, out_targets, weights, N, E, starts, n_starts, dist, parent, poison);
}

int cz_betweenness(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                   double *centrality, const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    float *c = (float *)malloc(((size_t)N + 1) * sizeof(float));
    const int rc = orc_betweenness(N, off, out_targets, weights, c, 100000000ull);
    for (uint32_t v = 0; v < N; v++) centrality[v] = c[v];
    free(c);
    free(off);
    if (rc) { g_err = "oracle: too many shortest paths"; return CZ_E_UNSUPPORTED; }
    return CZ_OK;
}

int cz_label_propagation(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                         uint32_t max_iter, uint32_t *labels, uint32_t *iters_run, uint32_t *n_colours, const volatile uint8_t *poison,
                         uint32_t flags) {
    (void)E;
    (void)flags;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    uint32_t *colour = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
    uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
    const uint32_t k = orc_lp_colouring(N, off, out_targets, colour);
    uint32_t at = 0;
    for (uint32_t c = 0; c < k; c++)
        for (uint32_t v = 0; v < N; v++)
            if (colour[v] == c) order[at++] = v;
    const int it = orc_label_propagation_in_order(N, off, out_targets, weights, order, max_iter, labels);
    free(colour);
    free(order);
    free(off);
    if (it < 0) { g_err = "a best score is NaN"; return CZ_E_INVALID; }
    if (iters_run) *iters_run = (uint32_t)it;
    if (n_colours) *n_colours = k;
    return CZ_OK;
}

int cz_closeness(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                 double *centrality, const volatile uint8_t *poison) {
    (void)E;
    if (poisoned(poison)) { g_err = "cancelled"; return CZ_E_CANCELLED; }
    uint64_t *off = widen(out_offsets, N);
    float *dist = (float *)malloc(sizeof(float) * (N ? N : 1));
    uint32_t *par = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t s = 0; s < N; s++) { /* all_pairs_shortest_path.rs:118-122, f32 throughout */
        orc_dijkstra(N, off, out_targets, weights, s, NULL, 0, dist, par);
        float total = 0.0f, nc = 0.0f;
        for (uint32_t v = 0; v < N; v++)
            if (isfinite(dist[v])) {
                total = total + dist[v];
                nc = nc + 1.0f;
            }
        centrality[s] = (double)(nc * nc / total / (float)(N - 1));
    }
    free(dist);
    free(par);
    free(off);
    return CZ_OK;
}
