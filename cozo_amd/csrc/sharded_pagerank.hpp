// sharded_pagerank.hpp -- the multi-GPU loop of graph::page_rank (PageRank::run, fixed_rule/algos/pagerank.rs:29-56),
// written once against a small backend interface:
//   * libcozo_gpu instantiates it with the HIP + RCCL backend (comm.hip: cz_pagerank_sharded / cz_pagerank_multi);
//   * tests/cpp/sharded_driver_test.cpp instantiates the SAME loop with a host backend (the CPU oracle's sweep as the
//     local step, the exchange steps handed to callbacks that run over torch.distributed/gloo with world_size 2), so
//     that the exchange order, the stopping rule and the collective cancellation are exercised without a GPU.
// Plain C++17, no HIP in here.
//
// Partition: 1-D by destination row, `per` rows per rank (the last rank's range may be short; buffers are padded to
// per * world), every rank holds the full contribution vector.  Per iteration:
//   sweep of the rank's rows -> its slice [rank * per, (rank + 1) * per) of the next contribution vector
//   exchange                 -> EXCHANGE_ALLGATHER (default): in-place all-gather of the slices, per * 4 bytes per rank, one
//                               hop per peer on the xGMI mesh;  EXCHANGE_ALLREDUCE (north_star's literal wording, kept as a
//                               labelled comparison): the other ranks' slices are zeroed and the whole vector is all-reduced
//                               (sum) -- the same values bit for bit (x + 0 + ... + 0), ~2 (world-1)/world * 4N bytes per link
//   all-reduce of two f64    -> {sum |new - old| (the reference's stopping rule), cancellation flag}
// Cancellation is collective: a rank whose poison flag is set stops sweeping but keeps taking part in the exchanges, its
// flag travels with the error, and every rank leaves at the same iteration (a rank that raised on its own would leave the
// others blocked in the next collective).
//
// Backend interface (all device-side work is stream-ordered; read_err2 is the only host synchronisation):
//   float *contrib(int i)                              i = 0, 1: two buffers of per * world floats
//   int init(float *contrib_full)                      contribution of every node = (1/N)/out_degree, local scores = 1/N
//   int begin_iteration(double poison_flag)            err2 = {0, poison_flag}
//   int step(const float *cin, float *cout)            sweep of the local rows; adds the local sum |new - old| into err2[0]
//   int all_gather_slices(float *buf)                  in place, slice r at buf + r * per
//   int zero_other_slices(float *buf)                  EXCHANGE_ALLREDUCE only
//   int all_reduce_sum_f32(float *buf, size_t n)
//   int all_reduce_err2()                              sum over ranks of the two f64
//   int read_err2(double out[2])                       waits for the stream, copies err2 to the host
#pragma once
#include <cstddef>
#include <cstdint>

namespace czs {

enum { EXCHANGE_ALLGATHER = 0, EXCHANGE_ALLREDUCE = 1 };
enum { RUN_OK = 0, RUN_CANCELLED = 1 };  // anything else: the backend's own error code, passed through

template <class Backend>
int run_sharded_pagerank(Backend &b, int world, uint32_t per, double tolerance, uint32_t max_iter, int exchange,
                         const volatile uint8_t *poison, uint32_t *iters_run, double *final_err) {
    float *cin = b.contrib(0), *cout = b.contrib(1);
    int rc = b.init(cin);
    if (rc) return rc;
    uint32_t it = 0;
    // `err < tolerance` can never hold for tolerance <= 0 (err is a sum of absolute values): the sweeps then run back to
    // back and the host looks at the reduced {error, flag} pair only every 8th iteration (for the flag) and at the end
    const bool never_stops_early = !(tolerance > 0.0);
    for (;;) {
        const bool last = it + 1 == max_iter;
        const bool p = poison && *poison;
        if ((rc = b.begin_iteration(p ? 1.0 : 0.0))) return rc;
        if (!p && (rc = b.step(cin, cout))) return rc;
        if (exchange == EXCHANGE_ALLREDUCE) {
            if ((rc = b.zero_other_slices(cout))) return rc;
            if ((rc = b.all_reduce_sum_f32(cout, (size_t)per * (size_t)world))) return rc;
        } else {
            if ((rc = b.all_gather_slices(cout))) return rc;
        }
        if ((rc = b.all_reduce_err2())) return rc;
        float *t = cin;
        cin = cout;
        cout = t;
        it++;
        if (never_stops_early && !last && !(poison && it % 8 == 0)) continue;
        double h[2] = {0.0, 0.0};
        if ((rc = b.read_err2(h))) return rc;
        if (h[1] > 0.0) return RUN_CANCELLED;  // some rank's Poison is set: every rank sees the same sum and leaves here
        if (h[0] < tolerance || it == max_iter) {
            if (iters_run) *iters_run = it;
            if (final_err) *final_err = h[0];
            return RUN_OK;
        }
    }
}

// The same loop with the exchange of the FIRST part of a rank's rows in flight while the SECOND part is swept (round 3;
// until then only cozo_amd/distributed.py had this form, over torch.distributed).  The exchange of iteration k feeds
// iteration k + 1, so the only overlap there is lies inside an iteration: the rank's rows are cut at `half` (the same
// cut, in rows of the padded per-rank range, on every rank) into two plans;
//   sweep part 0 -> begin the exchange of every rank's part-0 piece (on the backend's exchange stream, behind the sweep)
//   sweep part 1 (runs meanwhile) -> begin the exchange of the part-1 pieces -> join -> all-reduce of the two f64.
// Pieces land at their natural places in the full contribution vector, sources keep their ids, every row sum keeps its
// order: the scores equal the unsplit run's bit for bit.  Hides min(part-1 sweep, part-0 exchange) per iteration.
// Backend additions:
//   int step_part(int part, const float *cin, float *cout)     rows [rb, rb + half) / [rb + half, re); adds into err2[0]
//   int exchange_part_begin(int part, float *buf)              piece r at buf + r * per + (part ? half : 0), `half` or
//                                                              `per - half` floats; ordered behind the work queued so far
//   int exchange_join()                                        the work queued after it waits for the exchanges begun
template <class Backend>
int run_sharded_pagerank_overlapped(Backend &b, int world, uint32_t per, double tolerance, uint32_t max_iter,
                                    const volatile uint8_t *poison, uint32_t *iters_run, double *final_err) {
    (void)world;
    (void)per;
    float *cin = b.contrib(0), *cout = b.contrib(1);
    int rc = b.init(cin);
    if (rc) return rc;
    uint32_t it = 0;
    const bool never_stops_early = !(tolerance > 0.0);
    for (;;) {
        const bool last = it + 1 == max_iter;
        const bool p = poison && *poison;
        if ((rc = b.begin_iteration(p ? 1.0 : 0.0))) return rc;
        if (!p && (rc = b.step_part(0, cin, cout))) return rc;
        if ((rc = b.exchange_part_begin(0, cout))) return rc;
        if (!p && (rc = b.step_part(1, cin, cout))) return rc;
        if ((rc = b.exchange_part_begin(1, cout))) return rc;
        if ((rc = b.exchange_join())) return rc;
        if ((rc = b.all_reduce_err2())) return rc;
        float *t = cin;
        cin = cout;
        cout = t;
        it++;
        if (never_stops_early && !last && !(poison && it % 8 == 0)) continue;
        double h[2] = {0.0, 0.0};
        if ((rc = b.read_err2(h))) return rc;
        if (h[1] > 0.0) return RUN_CANCELLED;
        if (h[0] < tolerance || it == max_iter) {
            if (iters_run) *iters_run = it;
            if (final_err) *final_err = h[0];
            return RUN_OK;
        }
    }
}

}  // namespace czs
