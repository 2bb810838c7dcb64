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
