// sharded_driver_test.cpp -- TEST ONLY.  Instantiates the multi-GPU PageRank loop of libcozo_gpu
// (cozo_amd/csrc/sharded_pagerank.hpp, the code cz_pagerank_sharded / cz_pagerank_multi run over HIP + RCCL) with a
// HOST backend: the local step is one Jacobi sweep over the rank's rows exactly as graph::page_rank does it (sequential
// f32 sums in sorted in-neighbour order, f64 error; restated from oracle/cozo_oracle.c orc_pagerank), and the exchange
// steps are handed to callbacks -- tests/test_sharded_driver.py runs them over torch.distributed/gloo with world_size 2.
// What is under test is the loop itself: the order of sweep / exchange / error reduction, the stopping rule on the
// reduced error, the all-reduce comparison variant, and the collective cancellation.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../cozo_amd/csrc/sharded_pagerank.hpp"

extern "C" {
typedef int (*cz_test_all_gather_f32)(void *ctx, float *buf, uint64_t per);   // in place, slice r at buf + r * per
typedef int (*cz_test_all_reduce_f32)(void *ctx, float *buf, uint64_t n);
typedef int (*cz_test_all_reduce_f64)(void *ctx, double *buf, uint64_t n);
// every rank's piece [lo, lo + cnt) of its slice: piece r at buf + r * per + lo
typedef int (*cz_test_exchange_piece)(void *ctx, float *buf, uint64_t per, uint64_t lo, uint64_t cnt);
}

namespace {

struct HostBackend {
    uint32_t N, rb, re, per;
    int rank, world;
    const uint64_t *off;  // local: off[i] for row rb + i, off[0] == 0
    const uint32_t *src, *outdeg;
    float damping, base, init0;
    std::vector<float> c[2], scores;
    double e2[2];
    void *ctx;
    cz_test_all_gather_f32 ag;
    cz_test_all_reduce_f32 ar32;
    cz_test_all_reduce_f64 ar64;
    int steps = 0, gathers = 0, reduces = 0;

    float *contrib(int i) { return c[i].data(); }
    int init(float *cf) {
        for (uint32_t v = 0; v < N; v++) cf[v] = init0 / (float)outdeg[v];
        for (auto &s : scores) s = init0;
        return 0;
    }
    int begin_iteration(double flag) {
        e2[0] = 0.0;
        e2[1] = flag;
        return 0;
    }
    int step(const float *cin, float *cout) {
        double err = 0.0;
        for (uint32_t r = rb; r < re; r++) {
            float s = 0.0f;
            for (uint64_t e = off[r - rb]; e < off[r - rb + 1]; e++) s = s + cin[src[e]];
            const float old = scores[r - rb];
            const float nw = base + damping * s;
            scores[r - rb] = nw;
            cout[r] = nw / (float)outdeg[r];
            err += std::fabs((double)(nw - old));
        }
        e2[0] += err;
        steps++;
        return 0;
    }
    int all_gather_slices(float *buf) {
        gathers++;
        return ag(ctx, buf, per);
    }
    // the overlapped form: two parts of the rank's rows, cut `half` rows into the padded per-rank range
    uint32_t half = 0;
    cz_test_exchange_piece xp = nullptr;
    int pieces = 0;
    int step_part(int part, const float *cin, float *cout) {
        const uint32_t cut = std::min<uint32_t>(re, rb + half);
        const uint32_t r0 = part ? cut : rb, r1 = part ? re : cut;
        double err = 0.0;
        for (uint32_t r = r0; r < r1; r++) {
            float s = 0.0f;
            for (uint64_t e = off[r - rb]; e < off[r - rb + 1]; e++) s = s + cin[src[e]];
            const float old = scores[r - rb];
            const float nw = base + damping * s;
            scores[r - rb] = nw;
            cout[r] = nw / (float)outdeg[r];
            err += std::fabs((double)(nw - old));
        }
        e2[0] += err;
        steps++;
        return 0;
    }
    int exchange_part_begin(int part, float *buf) {
        pieces++;
        const uint64_t lo = part ? half : 0, cnt = part ? per - half : half;
        return cnt ? xp(ctx, buf, per, lo, cnt) : 0;  // (the host callback completes the exchange before it returns)
    }
    int exchange_join() { return 0; }
    int zero_other_slices(float *buf) {
        const size_t lo = (size_t)rank * per, total = (size_t)per * world;
        std::memset(buf, 0, lo * 4);
        if (lo + per < total) std::memset(buf + lo + per, 0, (total - lo - per) * 4);
        return 0;
    }
    int all_reduce_sum_f32(float *buf, size_t n) { return ar32(ctx, buf, n); }
    int all_reduce_err2() {
        reduces++;
        return ar64(ctx, e2, 2);
    }
    int read_err2(double out[2]) {
        out[0] = e2[0];
        out[1] = e2[1];
        return 0;
    }
};

}  // namespace

// returns czs::RUN_OK / czs::RUN_CANCELLED / a callback's error; scores_out [re - rb]; counters [3] = sweeps, gathers, reduces
extern "C" int cz_test_sharded_pagerank_host(uint32_t N, uint32_t per, int rank, int world, const uint64_t *off_local,
                                             const uint32_t *src, const uint32_t *outdeg, float damping, double tolerance,
                                             uint32_t max_iter, int exchange, const volatile uint8_t *poison, void *ctx,
                                             cz_test_all_gather_f32 ag, cz_test_all_reduce_f32 ar32, cz_test_all_reduce_f64 ar64,
                                             float *scores_out, uint32_t *iters_run, double *final_err, int *counters) {
    HostBackend b;
    b.N = N;
    b.per = per;
    b.rank = rank;
    b.world = world;
    b.rb = (uint32_t)std::min<uint64_t>(N, (uint64_t)rank * per);
    b.re = (uint32_t)std::min<uint64_t>(N, (uint64_t)(rank + 1) * per);
    b.off = off_local;
    b.src = src;
    b.outdeg = outdeg;
    b.damping = damping;
    b.init0 = 1.0f / (float)N;
    b.base = (1.0f - damping) / (float)N;
    b.c[0].assign((size_t)per * world, 0.f);
    b.c[1].assign((size_t)per * world, 0.f);
    b.scores.assign(b.re - b.rb, 0.f);
    b.ctx = ctx;
    b.ag = ag;
    b.ar32 = ar32;
    b.ar64 = ar64;
    const int rc = czs::run_sharded_pagerank(b, world, per, tolerance, max_iter, exchange, poison, iters_run, final_err);
    if (scores_out) std::memcpy(scores_out, b.scores.data(), b.scores.size() * 4);
    if (counters) {
        counters[0] = b.steps;
        counters[1] = b.gathers;
        counters[2] = b.reduces;
    }
    return rc;
}

// the overlapped form of the same loop (czs::run_sharded_pagerank_overlapped); counters [3] = part sweeps, pieces begun, reduces
extern "C" int cz_test_sharded_pagerank_overlapped_host(uint32_t N, uint32_t per, uint32_t half, int rank, int world,
                                                        const uint64_t *off_local, const uint32_t *src, const uint32_t *outdeg,
                                                        float damping, double tolerance, uint32_t max_iter,
                                                        const volatile uint8_t *poison, void *ctx, cz_test_exchange_piece xp,
                                                        cz_test_all_reduce_f64 ar64, float *scores_out, uint32_t *iters_run,
                                                        double *final_err, int *counters) {
    HostBackend b;
    b.N = N;
    b.per = per;
    b.half = half;
    b.rank = rank;
    b.world = world;
    b.rb = (uint32_t)std::min<uint64_t>(N, (uint64_t)rank * per);
    b.re = (uint32_t)std::min<uint64_t>(N, (uint64_t)(rank + 1) * per);
    b.off = off_local;
    b.src = src;
    b.outdeg = outdeg;
    b.damping = damping;
    b.init0 = 1.0f / (float)N;
    b.base = (1.0f - damping) / (float)N;
    b.c[0].assign((size_t)per * world, 0.f);
    b.c[1].assign((size_t)per * world, 0.f);
    b.scores.assign(b.re - b.rb, 0.f);
    b.ctx = ctx;
    b.ag = nullptr;
    b.ar32 = nullptr;
    b.ar64 = ar64;
    b.xp = xp;
    const int rc = czs::run_sharded_pagerank_overlapped(b, world, per, tolerance, max_iter, poison, iters_run, final_err);
    if (scores_out) std::memcpy(scores_out, b.scores.data(), b.scores.size() * 4);
    if (counters) {
        counters[0] = b.steps;
        counters[1] = b.pieces;
        counters[2] = b.reduces;
    }
    return rc;
}

// =====================================================================================================================
// the vertex-partitioned traversals (cozo_amd/csrc/sharded_traversal.hpp) with a host backend
// =====================================================================================================================
#include "../../cozo_amd/csrc/sharded_traversal.hpp"

extern "C" {
typedef int (*cz_test_all_reduce_u32)(void *ctx, uint32_t *buf, uint64_t n, int op /* 0 sum, 1 min */);
typedef int (*cz_test_all_reduce_u64)(void *ctx, uint64_t *buf, uint64_t n, int op);
}

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct HostTraversal {
    uint32_t N, rb, re;
    const uint64_t *off;  // local
    const uint32_t *tgt;
    const float *w;
    const uint32_t *goals;
    uint32_t n_goals;
    void *ctx;
    cz_test_all_reduce_u32 ar32;
    cz_test_all_reduce_u64 ar64;
    const volatile uint8_t *poison_after;  // set by the test between levels
    std::vector<uint32_t> depth, parent, claim, order, cnt, pos, buf, canon;
    std::vector<uint64_t> dp, prop;
    int exchanges = 0;

    bool owned(uint32_t u) const { return u >= rb && u < re; }
    int any_poisoned(bool mine, bool *any) {
        uint32_t v = mine ? 1u : 0u;
        int rc = ar32(ctx, &v, 1, 0);
        exchanges++;
        *any = v != 0;
        return rc;
    }
    // ---- BFS ----
    int bfs_reset(bool keep) {
        parent.assign(N, kNone);
        if (!keep) {
            depth.assign(N, kNone);
            claim.assign(N, kNone);
        }
        order.assign((size_t)N + 1, kNone);
        cnt.assign(N, 0);
        pos.assign(N, 0);
        buf.assign(N, 0);
        return 0;
    }
    int bfs_seed(uint32_t start, bool *already) {
        *already = depth[start] != kNone;
        if (*already) return 0;
        depth[start] = 0;
        order[0] = start;
        return 0;
    }
    int bfs_claim(uint32_t lo, uint32_t fsize) {
        for (uint32_t i = 0; i < fsize; i++) {
            const uint32_t u = order[lo + i];
            if (!owned(u)) continue;
            for (uint64_t e = off[u - rb]; e < off[u - rb + 1]; e++) {
                const uint32_t v = tgt[e];
                if (depth[v] == kNone && i < claim[v]) claim[v] = i;
            }
        }
        return 0;
    }
    int reduce_claim() { return ar32(ctx, claim.data(), N, 1); }
    int bfs_count(uint32_t lo, uint32_t fsize) {
        for (uint32_t i = 0; i < fsize; i++) {
            cnt[i] = 0;
            const uint32_t u = order[lo + i];
            if (!owned(u)) continue;
            const uint64_t e0 = off[u - rb], e1 = off[u - rb + 1];
            for (uint64_t e = e0; e < e1; e++) {
                const uint32_t v = tgt[e];
                if ((e == e0 || tgt[e - 1] != v) && depth[v] == kNone && claim[v] == i) cnt[i]++;
            }
        }
        return 0;
    }
    int reduce_counts(uint32_t fsize) { return ar32(ctx, cnt.data(), fsize, 0); }
    int bfs_scan(uint32_t fsize, uint32_t *total) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < fsize; i++) {
            pos[i] = run;
            run += cnt[i];
        }
        *total = run;
        return 0;
    }
    int bfs_emit(uint32_t lo, uint32_t fsize, uint32_t total, uint32_t next_depth) {
        std::fill(buf.begin(), buf.begin() + total, 0u);
        for (uint32_t i = 0; i < fsize; i++) {
            const uint32_t u = order[lo + i];
            if (!owned(u)) continue;
            uint32_t o = pos[i];
            const uint64_t e0 = off[u - rb], e1 = off[u - rb + 1];
            for (uint64_t e = e0; e < e1; e++) {
                const uint32_t v = tgt[e];
                if ((e == e0 || tgt[e - 1] != v) && depth[v] == kNone && claim[v] == i) {
                    buf[o++] = v + 1;
                    parent[v] = u;
                }
            }
        }
        (void)next_depth;
        return 0;
    }
    int reduce_next(uint32_t total) { return ar32(ctx, buf.data(), total, 0); }
    int bfs_commit(uint32_t at, uint32_t total, uint32_t next_depth) {
        for (uint32_t j = 0; j < total; j++) {
            const uint32_t v = buf[j] - 1;
            order[at + j] = v;
            depth[v] = next_depth;
        }
        return 0;
    }
    int goals_left(uint32_t start, uint32_t *left) {
        uint32_t c = 0;
        for (uint32_t i = 0; i < n_goals; i++)
            if (goals[i] < N && (depth[goals[i]] == kNone || goals[i] == start)) c++;
        *left = c;
        return 0;
    }
    int reduce_parents() { return ar32(ctx, parent.data(), N, 1); }
    // ---- SSSP ----
    static uint32_t bits(float f) {
        uint32_t b;
        std::memcpy(&b, &f, 4);
        return b;
    }
    static float val(uint32_t b) {
        float f;
        std::memcpy(&f, &b, 4);
        return f;
    }
    // near-far piles + the sparse exchange (the all-gathers are all-reduce(sum)s of buffers zero outside the rank's own slot)
    int rank = 0, world = 1;
    float delta = 0.f;
    uint32_t thr_bits = 0, rounds = 0;
    uint64_t pairs_exchanged = 0, words_exchanged = 0;
    std::vector<uint32_t> near, touched;
    std::vector<uint64_t> far, pairs, counts;
    uint32_t threshold_over(uint32_t min_bits) const {
        if (!(delta > 0.f) || !std::isfinite(delta)) return 0x7F800000u;
        const float m = val(min_bits);
        float t = m + delta;
        if (!(t > m)) t = std::nextafter(m, INFINITY);
        return bits(t);
    }
    int sssp_seed(uint32_t start, uint32_t *n_near) {
        dp.assign(N, ((uint64_t)0x7F800000u << 32) | kNone);
        prop.assign(N, ~0ull);
        canon.assign(N, kNone);
        near.clear();
        far.clear();
        touched.clear();
        thr_bits = threshold_over(0u);
        if (start < N) {
            dp[start] = 0x00000000FFFFFFFFull;
            near.push_back(start);
        }
        *n_near = (uint32_t)near.size();
        return 0;
    }
    int sssp_relax(uint32_t n_near) {
        for (uint32_t i = 0; i < n_near; i++) {
            const uint32_t u = near[i];
            if (!owned(u)) continue;
            const float du = val((uint32_t)(dp[u] >> 32));
            for (uint64_t e = off[u - rb]; e < off[u - rb + 1]; e++) {
                const uint32_t v = tgt[e];
                const uint32_t nb = bits(du + w[e]);
                if (nb < (uint32_t)(dp[v] >> 32)) {
                    if (prop[v] == ~0ull) touched.push_back(v);
                    prop[v] = std::min(prop[v], ((uint64_t)nb << 32) | u);
                }
            }
        }
        return 0;
    }
    int exchange_counts(bool poisoned, uint32_t *longest, bool *any) {
        counts.assign(world, 0);
        counts[rank] = (uint64_t)touched.size() | ((uint64_t)(poisoned ? 1 : 0) << 40);
        const int rc = ar64(ctx, counts.data(), world, 0);
        exchanges++;
        *longest = 0;
        *any = false;
        for (int r = 0; r < world; r++) {
            *longest = std::max(*longest, (uint32_t)counts[r]);
            *any = *any || (counts[r] >> 40) != 0;
            pairs_exchanged += (uint32_t)counts[r];
        }
        rounds++;
        return rc;
    }
    int exchange_pairs(uint32_t longest) {
        pairs.assign(2 * (size_t)world * longest, 0);
        uint64_t *slot = pairs.data() + 2 * (size_t)rank * longest;
        for (uint32_t i = 0; i < longest; i++) {
            uint64_t word = ~0ull, node = kNone;
            if (i < touched.size()) {
                node = touched[i];
                word = prop[node];
                prop[node] = ~0ull;
            }
            slot[2 * (size_t)i] = word;
            slot[2 * (size_t)i + 1] = node;
        }
        touched.clear();
        exchanges++;
        words_exchanged += pairs.size();
        return ar64(ctx, pairs.data(), pairs.size(), 0);
    }
    int sssp_apply(uint32_t longest, uint32_t *n_near, uint32_t *n_far) {
        const size_t total = (size_t)world * longest;
        std::vector<uint8_t> lowered(total, 0);
        for (size_t i = 0; i < total; i++) {
            const uint64_t word = pairs[2 * i], node = pairs[2 * i + 1];
            if (node == kNone) continue;
            if (word < dp[node]) {
                dp[node] = word;
                lowered[i] = 1;
            }
        }
        near.clear();
        for (size_t i = 0; i < total; i++) {
            if (!lowered[i]) continue;
            const uint64_t word = pairs[2 * i];
            const uint32_t v = (uint32_t)pairs[2 * i + 1];
            if (dp[v] != word) continue;  // a later pair went lower: that one is the winner
            const uint32_t cost = (uint32_t)(word >> 32);
            if (cost < thr_bits) near.push_back(v);
            else far.push_back(((uint64_t)cost << 32) | v);
        }
        *n_near = (uint32_t)near.size();
        *n_far = (uint32_t)far.size();
        return 0;
    }
    int sssp_next_bucket(uint32_t *n_near, uint32_t *n_far) {
        uint32_t m = 0xFFFFFFFFu;
        for (uint64_t ent : far)
            if ((uint32_t)(dp[(uint32_t)ent] >> 32) == (uint32_t)(ent >> 32)) m = std::min(m, (uint32_t)(ent >> 32));
        near.clear();
        std::vector<uint64_t> keep;
        if (m != 0xFFFFFFFFu) {
            thr_bits = threshold_over(m);
            for (uint64_t ent : far) {
                const uint32_t cost = (uint32_t)(ent >> 32);
                if ((uint32_t)(dp[(uint32_t)ent] >> 32) != cost) continue;  // stale
                if (cost < thr_bits) near.push_back((uint32_t)ent);
                else keep.push_back(ent);
            }
        }
        far.swap(keep);
        *n_near = (uint32_t)near.size();
        *n_far = (uint32_t)far.size();
        return 0;
    }
    int sssp_canonical_parents() {
        canon.assign(N, kNone);
        for (uint32_t u = rb; u < re; u++) {
            const uint32_t cu = (uint32_t)(dp[u] >> 32);
            if (cu == 0x7F800000u) continue;
            for (uint64_t e = off[u - rb]; e < off[u - rb + 1]; e++) {
                const uint32_t v = tgt[e], cv = (uint32_t)(dp[v] >> 32);
                if (cu < cv && bits(val(cu) + w[e]) == cv) canon[v] = std::min(canon[v], u);
            }
        }
        return 0;
    }
    int reduce_canonical() { return ar32(ctx, canon.data(), N, 1); }
};

}  // namespace

// BFS from `start` over this rank's rows; outputs full length N (+ order without the start) on every rank
extern "C" int cz_test_sharded_bfs_host(uint32_t N, uint32_t rb, uint32_t re, const uint64_t *off_local, const uint32_t *tgt,
                                        uint32_t start, const uint32_t *goals, uint32_t n_goals, int has_goals,
                                        const volatile uint8_t *poison, void *ctx, cz_test_all_reduce_u32 ar32,
                                        cz_test_all_reduce_u64 ar64, uint32_t *parent, uint32_t *depth, uint32_t *order,
                                        uint32_t *reached) {
    HostTraversal b;
    b.N = N;
    b.rb = rb;
    b.re = re;
    b.off = off_local;
    b.tgt = tgt;
    b.w = nullptr;
    b.goals = goals;
    b.n_goals = n_goals;
    b.ctx = ctx;
    b.ar32 = ar32;
    b.ar64 = ar64;
    const int rc = czs::run_sharded_bfs(b, start, N, has_goals != 0, false, poison, reached);
    std::memcpy(parent, b.parent.data(), (size_t)N * 4);
    std::memcpy(depth, b.depth.data(), (size_t)N * 4);
    std::memcpy(order, b.order.data() + 1, (size_t)N * 4);
    return rc;
}

// counters [4] = rounds, exchanges, pairs exchanged (all ranks' lists), u64 words that crossed the exchange (padding included)
extern "C" int cz_test_sharded_sssp_host(uint32_t N, uint32_t rb, uint32_t re, int rank, int world, float delta, const uint64_t *off_local,
                                         const uint32_t *tgt, const float *w, uint32_t start, const volatile uint8_t *poison, void *ctx,
                                         cz_test_all_reduce_u32 ar32, cz_test_all_reduce_u64 ar64, float *dist, uint32_t *parent,
                                         uint64_t *counters) {
    HostTraversal b;
    b.N = N;
    b.rb = rb;
    b.re = re;
    b.off = off_local;
    b.tgt = tgt;
    b.w = w;
    b.goals = nullptr;
    b.n_goals = 0;
    b.ctx = ctx;
    b.ar32 = ar32;
    b.ar64 = ar64;
    b.rank = rank;
    b.world = world;
    b.delta = delta;
    const int rc = czs::run_sharded_sssp(b, start, N, poison);
    if (counters) {
        counters[0] = b.rounds;
        counters[1] = (uint64_t)b.exchanges;
        counters[2] = b.pairs_exchanged;
        counters[3] = b.words_exchanged;
    }
    if (rc) return rc;
    for (uint32_t v = 0; v < N; v++) {
        dist[v] = HostTraversal::val((uint32_t)(b.dp[v] >> 32));
        parent[v] = b.canon[v] != kNone ? b.canon[v] : (uint32_t)b.dp[v];
    }
    return rc;
}

// ---- ConnectedComponents over a vertex partition (czs::run_sharded_cc) with a host backend ---------------------------------------
namespace {

struct HostCc {
    uint32_t N, rb, re;
    const uint64_t *off;  // local rows of the symmetrised graph
    const uint32_t *tgt;
    void *ctx;
    cz_test_all_reduce_u32 ar32;
    std::vector<uint32_t> comp, prev, group;
    uint32_t n_groups = 0;
    int exchanges = 0;

    uint32_t root(uint32_t v) const {
        while (comp[v] != v) v = comp[v];
        return v;
    }
    void compress() {
        for (uint32_t v = 0; v < N; v++) comp[v] = root(v);  // ascending: comp[comp[v]] is final already
    }
    int any_poisoned(bool mine, bool *any) {
        uint32_t v = mine ? 1u : 0u;
        const int rc = ar32(ctx, &v, 1, 0);
        exchanges++;
        *any = v != 0;
        return rc;
    }
    int cc_init() {
        comp.resize(N);
        for (uint32_t v = 0; v < N; v++) comp[v] = v;
        prev = comp;
        return 0;
    }
    int cc_local_round() {
        for (uint32_t u = rb; u < re; u++)
            for (uint64_t e = off[u - rb]; e < off[u - rb + 1]; e++) {
                const uint32_t a = root(u), b = root(tgt[e]);
                if (a != b) comp[std::max(a, b)] = std::min(a, b);  // the higher root goes under the lower
            }
        compress();
        return 0;
    }
    int reduce_labels() {
        exchanges++;
        return ar32(ctx, comp.data(), N, 1);
    }
    int cc_settle(bool *changed) {
        compress();
        *changed = comp != prev;
        prev = comp;
        return 0;
    }
    int cc_number_groups() {
        std::vector<uint32_t> rank(N, 0);
        uint32_t k = 0;
        for (uint32_t v = 0; v < N; v++)
            if (comp[v] == v) rank[v] = k++;
        group.resize(N);
        for (uint32_t v = 0; v < N; v++) group[v] = rank[comp[v]];
        n_groups = k;
        return 0;
    }
};

}  // namespace

// counters [2] = rounds, exchanges
extern "C" int cz_test_sharded_cc_host(uint32_t N, uint32_t rb, uint32_t re, const uint64_t *off_local, const uint32_t *tgt,
                                       const volatile uint8_t *poison, void *ctx, cz_test_all_reduce_u32 ar32, uint32_t *group,
                                       uint32_t *n_groups, uint32_t *counters) {
    HostCc b;
    b.N = N;
    b.rb = rb;
    b.re = re;
    b.off = off_local;
    b.tgt = tgt;
    b.ctx = ctx;
    b.ar32 = ar32;
    uint32_t rounds = 0;
    const int rc = czs::run_sharded_cc(b, poison, &rounds);
    if (rc == 0) {
        std::memcpy(group, b.group.data(), (size_t)N * 4);
        *n_groups = b.n_groups;
    }
    counters[0] = rounds;
    counters[1] = (uint32_t)b.exchanges;
    return rc;
}

