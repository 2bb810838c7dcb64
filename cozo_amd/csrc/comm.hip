// comm.hip -- the multi-GPU side of the C ABI: RCCL communicators (cz_comm_*), row-sharded PageRank behind the
// boundary (cz_pagerank_sharded for one process per GPU, cz_pagerank_multi for one process driving several GPUs -- what
// a cozo process is), and hnsw_knn over an index partitioned into one sub-index per rank (cz_hnsw_search_sharded).
//
// RCCL is bound at run time (dlopen of librccl.so.1): libcozo_gpu.so must not drag a second copy of the collectives
// library -- or of the HIP runtime it depends on -- into a process that already holds one (a PyTorch process carries its
// own librccl.so with the same SONAME; the loader then hands back that copy).  COZO_RCCL_LIB names another file.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "distance.h"
#include "hnsw_index.h"
#include "sharded_pagerank.hpp"

// ---- the few RCCL declarations this file needs (rccl.h: NCCL 2.x ABI) -------------------------------------------------
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char *e = getenv("COZO_RCCL_LIB")) names.push_back(e);
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        const char *rocm = getenv("ROCM_PATH");
        names.push_back(std::string(rocm ? rocm : "/opt/rocm") + "/lib/librccl.so.1");
        for (const auto &n : names) {
            r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            r.why += n + ": " + dlerror() + "; ";
        }
        if (!r.h) return;
#define CZ_SYM(field, name)                                  \
    r.field = (decltype(r.field))dlsym(r.h, name);          \
    if (!r.field) {                                          \
        r.why = std::string("librccl lacks ") + name;        \
        r.h = nullptr;                                       \
        return;                                              \
    }
        CZ_SYM(GetUniqueId, "ncclGetUniqueId")
        CZ_SYM(CommInitRank, "ncclCommInitRank")
        CZ_SYM(CommInitAll, "ncclCommInitAll")
        CZ_SYM(CommDestroy, "ncclCommDestroy")
        CZ_SYM(AllGather, "ncclAllGather")
        CZ_SYM(AllReduce, "ncclAllReduce")
        CZ_SYM(Broadcast, "ncclBroadcast")
        CZ_SYM(GroupStart, "ncclGroupStart")
        CZ_SYM(GroupEnd, "ncclGroupEnd")
        CZ_SYM(GetErrorString, "ncclGetErrorString")
#undef CZ_SYM
    });
    return r.h ? &r : nullptr;
}

std::string g_rccl_why;
int need_rccl(Rccl **out) {
    Rccl *r = rccl();
    if (!r) return cz::set_error(CZ_E_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded into this process; set COZO_RCCL_LIB");
    *out = r;
    return CZ_OK;
}

#define CZ_NCCL(R, expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess)                                                                             \
            return cz::set_error(CZ_E_HIP, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct cz_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;
};

namespace cz {
int comm_world(const cz_comm *c) { return c ? c->world : 0; }
int comm_rank(const cz_comm *c) { return c ? c->rank : -1; }
int comm_all_reduce(cz_comm *c, void *buf, size_t count, int dtype, int op, hipStream_t stream) {
    if (!c || !buf) return set_error(CZ_E_INVALID, "null argument");
    if (c->world == 1 || count == 0) return CZ_OK;
    Rccl *R = nullptr;
    int rc = need_rccl(&R);
    if (rc) return rc;
    static const ncclDataType_t types[] = {ncclUint32, ncclUint64, ncclFloat32, ncclFloat64};
    CZ_NCCL(R, R->AllReduce(buf, buf, count, types[dtype], op == COMM_MIN ? ncclMin : ncclSum, c->nccl, stream));
    return CZ_OK;
}
int comm_all_gather(cz_comm *c, const void *send, void *recv, size_t count, int dtype, hipStream_t stream) {
    if (!c || !send || !recv) return set_error(CZ_E_INVALID, "null argument");
    if (count == 0) return CZ_OK;
    static const size_t width[] = {4, 8, 4, 8};
    if (c->world == 1) {
        if (send != recv) CZ_HIP(hipMemcpyAsync(recv, send, count * width[dtype], hipMemcpyDeviceToDevice, stream));
        return CZ_OK;
    }
    Rccl *R = nullptr;
    int rc = need_rccl(&R);
    if (rc) return rc;
    static const ncclDataType_t types[] = {ncclUint32, ncclUint64, ncclFloat32, ncclFloat64};
    CZ_NCCL(R, R->AllGather(send, recv, count, types[dtype], c->nccl, stream));
    return CZ_OK;
}
}  // namespace cz

extern "C" int cz_comm_unique_id(uint8_t *id) {
    if (!id) return cz::set_error(CZ_E_INVALID, "null id");
    Rccl *R = nullptr;
    int rc = need_rccl(&R);
    if (rc) return rc;
    ncclUniqueId u;
    CZ_NCCL(R, R->GetUniqueId(&u));
    static_assert(sizeof(u) == CZ_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof u);
    return CZ_OK;
}

extern "C" int cz_comm_create_rank(const uint8_t *id, int rank, int world, cz_comm **out) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return cz::set_error(CZ_E_INVALID, "bad rank %d of %d", rank, world);
    int rc = cz::ensure_device();
    if (rc) return rc;
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    std::unique_ptr<cz_comm> c(new cz_comm());
    c->rank = rank;
    c->world = world;
    CZ_HIP(hipGetDevice(&c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    CZ_NCCL(R, R->CommInitRank(&c->nccl, world, u, rank));
    *out = c.release();
    return CZ_OK;
}

extern "C" void cz_comm_destroy(cz_comm *c) {
    if (!c) return;
    if (c->nccl && !getenv("CZ_COMM_NO_DESTROY")) {
        if (Rccl *R = rccl()) {
            (void)hipSetDevice(c->device);
            (void)R->CommDestroy(c->nccl);
        }
    }
    delete c;
}

extern "C" int cz_comm_rank(const cz_comm *c) { return c ? c->rank : -1; }
extern "C" int cz_comm_size(const cz_comm *c) { return c ? c->world : 0; }

extern "C" int cz_comm_all_gather(cz_comm *c, void *buf_dev, uint64_t bytes_per_rank, void *stream) {
    if (!c || !buf_dev) return cz::set_error(CZ_E_INVALID, "null argument");
    // one rank: the gather is the identity.  (Not handed to RCCL: an in-place ncclAllGather of a 1-rank communicator on
    // memory owned by PyTorch's allocator left the process with a double free at exit -- scratch/r2_rccl_exit.py.)
    if (c->world == 1) return CZ_OK;
    Rccl *R = nullptr;
    int rc = need_rccl(&R);
    if (rc) return rc;
    CZ_NCCL(R, R->AllGather((const char *)buf_dev + (size_t)c->rank * bytes_per_rank, buf_dev, (size_t)bytes_per_rank, ncclUint8,
                            c->nccl, (hipStream_t)stream));
    return CZ_OK;
}

extern "C" int cz_comm_all_reduce_sum_f64(cz_comm *c, double *buf_dev, uint64_t n, void *stream) {
    if (!c || !buf_dev) return cz::set_error(CZ_E_INVALID, "null argument");
    Rccl *R = nullptr;
    int rc = need_rccl(&R);
    if (rc) return rc;
    CZ_NCCL(R, R->AllReduce(buf_dev, buf_dev, (size_t)n, ncclFloat64, ncclSum, c->nccl, (hipStream_t)stream));
    return CZ_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// row-sharded PageRank
// ------------------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) err2_set_kernel(double *e2, double flag) {
    if (threadIdx.x == 0) {
        e2[0] = 0.0;
        e2[1] = flag;
    }
}

struct HipPagerankBackend {
    Rccl *R;
    cz_comm *comm;
    cz_pagerank_plan *plan;
    hipStream_t stream;
    uint32_t per;
    cz::DevBuf<float> c0, c1;
    cz::DevBuf<double> e2;

    int alloc() {
        const size_t padded = (size_t)per * (size_t)comm->world;
        CZ_HIP(c0.alloc(padded));
        CZ_HIP(c1.alloc(padded));
        CZ_HIP(e2.alloc(2));
        // the padding beyond N is gathered but never read; keep it defined
        CZ_HIP(hipMemsetAsync(c0.p, 0, padded * 4, stream));
        CZ_HIP(hipMemsetAsync(c1.p, 0, padded * 4, stream));
        return CZ_OK;
    }
    float *contrib(int i) { return i ? c1.p : c0.p; }
    int init(float *c) { return cz_pagerank_plan_init(plan, c, stream); }
    int begin_iteration(double flag) {
        hipLaunchKernelGGL(err2_set_kernel, dim3(1), dim3(64), 0, stream, e2.p, flag);
        return CZ_OK;
    }
    int step(const float *cin, float *cout) { return cz_pagerank_plan_step(plan, cin, cout, e2.p, stream); }
    int all_gather_slices(float *buf) {
        if (comm->world == 1) return CZ_OK;  // the identity
        CZ_NCCL(R, R->AllGather(buf + (size_t)comm->rank * per, buf, (size_t)per, ncclFloat32, comm->nccl, stream));
        return CZ_OK;
    }
    int zero_other_slices(float *buf) {
        const size_t lo = (size_t)comm->rank * per, total = (size_t)per * comm->world;
        if (lo) CZ_HIP(hipMemsetAsync(buf, 0, lo * 4, stream));
        if (lo + per < total) CZ_HIP(hipMemsetAsync(buf + lo + per, 0, (total - lo - per) * 4, stream));
        return CZ_OK;
    }
    int all_reduce_sum_f32(float *buf, size_t n) {
        CZ_NCCL(R, R->AllReduce(buf, buf, n, ncclFloat32, ncclSum, comm->nccl, stream));
        return CZ_OK;
    }
    int all_reduce_err2() {
        CZ_NCCL(R, R->AllReduce(e2.p, e2.p, 2, ncclFloat64, ncclSum, comm->nccl, stream));
        return CZ_OK;
    }
    int read_err2(double out[2]) {
        CZ_HIP(hipMemcpyAsync(out, e2.p, 16, hipMemcpyDeviceToHost, stream));
        CZ_HIP(hipStreamSynchronize(stream));
        return CZ_OK;
    }

    // ---- the overlapped form (czs::run_sharded_pagerank_overlapped): two plans over the two parts of the rank's rows, the
    // exchanges on a stream of their own.  A part's pieces sit `per` floats apart (piece r at buf + r * per + lo), which an
    // all-gather cannot address: one broadcast per rank, grouped into one RCCL operation.
    cz_pagerank_plan *plan2 = nullptr;
    uint32_t half = 0;
    hipStream_t xs = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    int overlapped_setup() {
        CZ_HIP(hipStreamCreateWithFlags(&xs, hipStreamNonBlocking));
        CZ_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
        CZ_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        return CZ_OK;
    }
    void overlapped_teardown() {
        if (xs) (void)hipStreamSynchronize(xs);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (xs) (void)hipStreamDestroy(xs);
        xs = nullptr;
        ev_ready = ev_done = nullptr;
    }
    int init_both(float *c) {
        int rc = cz_pagerank_plan_init(plan, c, stream);
        return rc ? rc : cz_pagerank_plan_init(plan2, c, stream);  // (the contribution vector is written twice, alike)
    }
    int step_part(int part, const float *cin, float *cout) {
        return cz_pagerank_plan_step(part ? plan2 : plan, cin, cout, e2.p, stream);
    }
    int exchange_part_begin(int part, float *buf) {
        if (comm->world == 1) return CZ_OK;
        const size_t lo = part ? half : 0, cnt = part ? (size_t)per - half : half;
        if (cnt == 0) return CZ_OK;
        CZ_HIP(hipEventRecord(ev_ready, stream));  // behind the sweep of this part
        CZ_HIP(hipStreamWaitEvent(xs, ev_ready, 0));
        CZ_NCCL(R, R->GroupStart());
        for (int r = 0; r < comm->world; r++) {
            float *piece = buf + (size_t)r * per + lo;
            ncclResult_t nr = R->Broadcast(piece, piece, cnt, ncclFloat32, r, comm->nccl, xs);
            if (nr != ncclSuccess) {
                (void)R->GroupEnd();
                return cz::set_error(CZ_E_HIP, "ncclBroadcast failed: %s", R->GetErrorString(nr));
            }
        }
        CZ_NCCL(R, R->GroupEnd());
        return CZ_OK;
    }
    int exchange_join() {
        if (comm->world == 1) return CZ_OK;
        CZ_HIP(hipEventRecord(ev_done, xs));
        CZ_HIP(hipStreamWaitEvent(stream, ev_done, 0));
        return CZ_OK;
    }
};

// the loop hands `init` the first buffer; the overlapped form has two plans to initialise
struct HipPagerankBackendOverlapped : HipPagerankBackend {
    int init(float *c) { return init_both(c); }
};

}  // namespace

extern "C" int cz_pagerank_sharded_overlapped(cz_comm *comm, cz_pagerank_plan *plan_first, cz_pagerank_plan *plan_second,
                                              uint32_t rows_per_rank, uint32_t half_rows, double tolerance, uint32_t max_iter,
                                              uint32_t *iters_run, double *final_err, const volatile uint8_t *poison, void *stream) {
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (!comm || !plan_first || !plan_second) return cz::set_error(CZ_E_INVALID, "null argument");
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "iterations must be positive");
    if (half_rows > rows_per_rank) return cz::set_error(CZ_E_INVALID, "half_rows %u > rows_per_rank %u", half_rows, rows_per_rank);
    const uint32_t N = cz_pagerank_plan_nodes(plan_first);
    if (cz_pagerank_plan_nodes(plan_second) != N) return cz::set_error(CZ_E_INVALID, "the two plans belong to different graphs");
    if ((uint64_t)rows_per_rank * (uint64_t)comm->world < N)
        return cz::set_error(CZ_E_INVALID, "rows_per_rank %u x %d ranks does not cover %u nodes", rows_per_rank, comm->world, N);
    int rc = cz::ensure_device();
    if (rc) return rc;
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    HipPagerankBackendOverlapped b;
    b.R = R;
    b.comm = comm;
    b.plan = plan_first;
    b.plan2 = plan_second;
    b.stream = (hipStream_t)stream;
    b.per = rows_per_rank;
    b.half = half_rows;
    if ((rc = b.alloc())) return rc;
    if ((rc = b.overlapped_setup())) {
        b.overlapped_teardown();
        return rc;
    }
    rc = czs::run_sharded_pagerank_overlapped(b, comm->world, rows_per_rank, tolerance, max_iter, poison, iters_run, final_err);
    (void)hipStreamSynchronize((hipStream_t)stream);  // the buffers die with this scope
    b.overlapped_teardown();
    if (rc == czs::RUN_CANCELLED) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    return rc;
}

extern "C" int cz_pagerank_sharded(cz_comm *comm, cz_pagerank_plan *plan, uint32_t rows_per_rank, double tolerance,
                                   uint32_t max_iter, uint32_t flags, uint32_t *iters_run, double *final_err,
                                   const volatile uint8_t *poison, void *stream) {
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (!comm || !plan) return cz::set_error(CZ_E_INVALID, "null argument");
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "iterations must be positive");
    const uint32_t N = cz_pagerank_plan_nodes(plan);
    if ((uint64_t)rows_per_rank * (uint64_t)comm->world < N)
        return cz::set_error(CZ_E_INVALID, "rows_per_rank %u x %d ranks does not cover %u nodes", rows_per_rank, comm->world, N);
    int rc = cz::ensure_device();
    if (rc) return rc;
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    HipPagerankBackend b;
    b.R = R;
    b.comm = comm;
    b.plan = plan;
    b.stream = (hipStream_t)stream;
    b.per = rows_per_rank;
    if ((rc = b.alloc())) return rc;
    rc = czs::run_sharded_pagerank(b, comm->world, rows_per_rank, tolerance, max_iter,
                                   (flags & CZ_PR_EXCHANGE_ALLREDUCE) ? czs::EXCHANGE_ALLREDUCE : czs::EXCHANGE_ALLGATHER, poison,
                                   iters_run, final_err);
    (void)hipStreamSynchronize((hipStream_t)stream);  // the buffers die with this scope
    if (rc == czs::RUN_CANCELLED) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    return rc;
}

// The communicators of the single-process forms (cz_pagerank_multi, cz_{bfs,sssp,connected_components}_multi): ncclCommInitAll costs
// ~0.5 s even for ONE device (measured, round 4: every *_multi call on a 200k-node graph took 538 ms), so a set is created on the first
// call for a device count and kept for the life of the process (cz_shutdown destroys them).  The lock is held for the whole call: one
// collective job per device set at a time.  A call that failed drops its set -- the next one starts from fresh communicators.
namespace {
struct MultiComms {
    std::mutex mu;
    std::map<int, std::vector<ncclComm_t>> by_world;
};
MultiComms &multi_comms() {
    static MultiComms m;
    return m;
}
int multi_comms_get(Rccl *R, int world, std::vector<ncclComm_t> **out) {  // caller holds multi_comms().mu
    auto &m = multi_comms().by_world;
    auto it = m.find(world);
    if (it == m.end()) {
        std::vector<int> devs(world);
        for (int i = 0; i < world; i++) devs[i] = i;
        std::vector<ncclComm_t> comms(world, nullptr);
        CZ_NCCL(R, R->CommInitAll(comms.data(), world, devs.data()));
        it = m.emplace(world, std::move(comms)).first;
    }
    *out = &it->second;
    return CZ_OK;
}
void multi_comms_drop(Rccl *R, int world) {  // caller holds the lock
    auto &m = multi_comms().by_world;
    auto it = m.find(world);
    if (it == m.end()) return;
    if (!getenv("CZ_COMM_NO_DESTROY"))
        for (int r = 0; r < world; r++) {
            (void)hipSetDevice(r);
            if (it->second[r]) (void)R->CommDestroy(it->second[r]);
        }
    m.erase(it);
    (void)cz::ensure_device();
}
}  // namespace

extern "C" void cz_comm_multi_shutdown(void) {
    std::lock_guard<std::mutex> lk(multi_comms().mu);
    if (multi_comms().by_world.empty()) return;  // (RCCL is not even loaded then)
    Rccl *R = nullptr;
    if (need_rccl(&R)) return;
    std::vector<int> worlds;
    for (auto &kv : multi_comms().by_world) worlds.push_back(kv.first);
    for (int w : worlds) multi_comms_drop(R, w);
}

// One process, n_gpus devices: rows split evenly, one host thread per GPU, one RCCL communicator per GPU (ncclCommInitAll).
extern "C" int cz_pagerank_multi(const uint32_t *in_offsets, const uint32_t *in_sources, const uint32_t *out_degree, uint32_t N,
                                 uint64_t E, float damping, double tolerance, uint32_t max_iter, int n_gpus, uint32_t flags,
                                 float *scores, uint32_t *iters_run, double *final_err, const volatile uint8_t *poison) {
    if (iters_run) *iters_run = 0;
    if (final_err) *final_err = 0.0;
    if (N == 0) return CZ_OK;  // pagerank.rs:43-45
    if (!scores || !in_offsets || !out_degree || (E && !in_sources)) return cz::set_error(CZ_E_INVALID, "null argument");
    if (max_iter == 0) return cz::set_error(CZ_E_INVALID, "iterations must be positive");
    if (in_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "in_offsets[N] (%u) != E (%llu)", in_offsets[N], (unsigned long long)E);
    int have = cz_device_count();
    if (n_gpus < 1 || n_gpus > have) return cz::set_error(CZ_E_INVALID, "n_gpus = %d, %d device(s) visible", n_gpus, have);
    int rc = cz::ensure_device();
    if (rc) return rc;
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    const int world = n_gpus;
    const uint32_t per = (uint32_t)(((uint64_t)N + world - 1) / world);
    std::vector<int> devs(world);
    for (int i = 0; i < world; i++) devs[i] = i;
    std::lock_guard<std::mutex> comms_lock(multi_comms().mu);
    std::vector<ncclComm_t> *comms_p = nullptr;
    if ((rc = multi_comms_get(R, world, &comms_p))) return rc;
    std::vector<ncclComm_t> &comms = *comms_p;
    std::vector<int> rcs(world, CZ_OK);
    std::vector<std::string> msgs(world);
    std::vector<uint32_t> its(world, 0);
    std::vector<double> errs(world, 0.0);
    auto worker = [&](int r) {
        cz::t_device_override = devs[r];
        auto fail = [&](int code) {
            rcs[r] = code;
            msgs[r] = cz_last_error();
        };
        int wrc = cz::ensure_device();
        if (wrc) return fail(wrc);
        const uint32_t rb = std::min<uint64_t>(N, (uint64_t)r * per), re = std::min<uint64_t>(N, (uint64_t)(r + 1) * per);
        std::vector<uint32_t> off((size_t)(re - rb) + 1);
        for (uint32_t i = 0; i <= re - rb; i++) off[i] = in_offsets[rb + i] - in_offsets[rb];
        cz_pagerank_plan *plan = nullptr, *plan_b = nullptr;
        // CZ_PR_OVERLAP_EXCHANGE: the rank's rows as two plans cut in the middle of the padded range; the first part's
        // exchange runs while the second part is swept (cz_pagerank_sharded_overlapped)
        const bool overlap = (flags & CZ_PR_OVERLAP_EXCHANGE) != 0 && !(flags & CZ_PR_EXCHANGE_ALLREDUCE);
        const uint32_t half = per / 2;
        const uint32_t mid = overlap ? std::min<uint32_t>(re, rb + half) : re;
        wrc = cz_pagerank_plan_create(off.data(), in_sources + in_offsets[rb], out_degree, N, rb, mid, damping, &plan,
                                      flags & (CZ_PR_GATHER | CZ_PR_BLOCKED | CZ_PR_ACCUMULATE));
        if (!wrc && overlap) {
            std::vector<uint32_t> off2((size_t)(re - mid) + 1);
            for (uint32_t i = 0; i <= re - mid; i++) off2[i] = in_offsets[mid + i] - in_offsets[mid];
            wrc = cz_pagerank_plan_create(off2.data(), in_sources + in_offsets[mid], out_degree, N, mid, re, damping, &plan_b,
                                          flags & (CZ_PR_GATHER | CZ_PR_BLOCKED | CZ_PR_ACCUMULATE));
        }
        // a rank that failed before the loop would leave the others blocked in the first collective: every rank enters
        // the loop, a failed one with its poison flag raised
        cz_comm c;
        c.rank = r;
        c.world = world;
        c.device = devs[r];
        c.nccl = comms[r];
        hipStream_t st = nullptr;
        if (!wrc && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) wrc = cz::set_error(CZ_E_HIP, "hipStreamCreate failed");
        if (wrc) fail(wrc);
        static const uint8_t kSet = 1;
        std::unique_ptr<cz_pagerank_plan, void (*)(cz_pagerank_plan *)> guard(plan, cz_pagerank_plan_destroy), guard_b(plan_b, cz_pagerank_plan_destroy);
        cz_pagerank_plan *use = plan, *use_b = plan_b;
        cz_pagerank_plan *empty = nullptr, *empty_b = nullptr;
        uint32_t z = 0;
        if (!use) {  // an empty stand-in so that this rank can still take part in the exchanges
            if (cz_pagerank_plan_create(&z, nullptr, out_degree, N, rb, rb, damping, &empty, 0) == CZ_OK) use = empty;
        }
        if (overlap && !use_b) {
            if (cz_pagerank_plan_create(&z, nullptr, out_degree, N, mid, mid, damping, &empty_b, 0) == CZ_OK) use_b = empty_b;
        }
        if (use && (!overlap || use_b)) {
            int lrc = overlap ? cz_pagerank_sharded_overlapped(&c, use, use_b, per, half, tolerance, max_iter, &its[r], &errs[r],
                                                               wrc ? &kSet : poison, st)
                              : cz_pagerank_sharded(&c, use, per, tolerance, max_iter, flags, &its[r], &errs[r], wrc ? &kSet : poison, st);
            if (!wrc && lrc) fail(lrc);
            if (!wrc && !lrc && mid > rb) {
                lrc = cz_pagerank_plan_read_scores(use, scores + rb, 0, st);
                if (lrc) fail(lrc);
            }
            if (!wrc && !lrc && overlap && re > mid) {
                lrc = cz_pagerank_plan_read_scores(use_b, scores + mid, 0, st);
                if (lrc) fail(lrc);
            }
        }
        if (empty_b) cz_pagerank_plan_destroy(empty_b);
        if (empty) cz_pagerank_plan_destroy(empty);
        if (st) (void)hipStreamDestroy(st);
        c.nccl = nullptr;
        cz::t_device_override = -1;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < world; r++) th.emplace_back(worker, r);
    worker(0);
    for (auto &t : th) t.join();
    (void)cz::ensure_device();
    for (int r = 0; r < world; r++)
        if (rcs[r]) {
            multi_comms_drop(R, world);  // (a failed collective job: do not trust its communicators again)
            return cz::set_error(rcs[r], "GPU %d: %s", r, msgs[r].c_str());
        }
    if (iters_run) *iters_run = its[0];
    if (final_err) *final_err = errs[0];
    return CZ_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// One cozo process, n GPUs: the vertex-partitioned traversals without a process per GPU (round 3; until then only PageRank
// had this form).  One host thread per device, one RCCL communicator per device (ncclCommInitAll), rows split evenly; every
// thread runs the collective entry point on its range of the host CSR.  The results are the same on every rank: rank 0
// writes the caller's buffers, the others scratch.
// ------------------------------------------------------------------------------------------------------------------
namespace {

template <class F>
int run_on_devices(int n_gpus, F fn /* int(int rank, cz_comm *comm) */) {
    int have = cz_device_count();
    if (n_gpus < 1 || n_gpus > have) return cz::set_error(CZ_E_INVALID, "n_gpus = %d, %d device(s) visible", n_gpus, have);
    int rc = cz::ensure_device();
    if (rc) return rc;
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    std::vector<int> devs(n_gpus);
    for (int i = 0; i < n_gpus; i++) devs[i] = i;
    std::lock_guard<std::mutex> comms_lock(multi_comms().mu);
    std::vector<ncclComm_t> *comms_p = nullptr;
    if ((rc = multi_comms_get(R, n_gpus, &comms_p))) return rc;
    std::vector<ncclComm_t> &comms = *comms_p;
    std::vector<int> rcs(n_gpus, CZ_OK);
    std::vector<std::string> msgs(n_gpus);
    auto worker = [&](int r) {
        cz::t_device_override = devs[r];
        int wrc = cz::ensure_device();
        cz_comm c;
        c.rank = r;
        c.world = n_gpus;
        c.device = devs[r];
        c.nccl = comms[r];
        if (!wrc) wrc = fn(r, &c);
        if (wrc) {
            rcs[r] = wrc;
            msgs[r] = cz_last_error();
        }
        c.nccl = nullptr;
        cz::t_device_override = -1;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n_gpus; r++) th.emplace_back(worker, r);
    worker(0);
    for (auto &t : th) t.join();
    (void)cz::ensure_device();
    for (int r = 0; r < n_gpus; r++)
        if (rcs[r]) {
            multi_comms_drop(R, n_gpus);  // (a failed collective job: do not trust its communicators again)
            return cz::set_error(rcs[r], "GPU %d: %s", r, msgs[r].c_str());
        }
    return CZ_OK;
}

// this rank's rows [rb, re) of a host CSR: offsets relative to the shard
struct RowShard {
    uint32_t rb, re;
    std::vector<uint32_t> off;
    RowShard(const uint32_t *offsets, uint32_t N, int rank, int world) {
        const uint32_t per = (uint32_t)(((uint64_t)N + world - 1) / world);
        rb = (uint32_t)std::min<uint64_t>(N, (uint64_t)rank * per);
        re = (uint32_t)std::min<uint64_t>(N, (uint64_t)(rank + 1) * per);
        off.resize((size_t)(re - rb) + 1);
        for (uint32_t i = 0; i <= re - rb; i++) off[i] = offsets[rb + i] - offsets[rb];
    }
};

}  // namespace

extern "C" int cz_bfs_multi(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, int n_gpus,
                            const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, int share_visited,
                            uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison) {
    if (N == 0 || n_starts == 0) return CZ_OK;
    if (!out_offsets || !starts || !parent || (E && !out_targets)) return cz::set_error(CZ_E_INVALID, "null argument");
    if (out_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "out_offsets[N] (%u) != E (%llu)", out_offsets[N], (unsigned long long)E);
    return run_on_devices(n_gpus, [&](int r, cz_comm *c) {
        RowShard sh(out_offsets, N, r, c->world);
        const size_t full = (size_t)n_starts * N;
        std::vector<uint32_t> sp, sd, so, sr;  // the other ranks' copies of the (identical) results
        if (r) {
            sp.resize(full);
            if (depth) sd.resize(full);
            if (order) so.resize(full);
            sr.resize(n_starts);
        }
        return cz_bfs_sharded(c, sh.off.data(), out_targets + out_offsets[sh.rb], N, sh.rb, sh.re, sh.off.back(), starts, n_starts, goals,
                              n_goals, share_visited, r ? sp.data() : parent, r ? (depth ? sd.data() : nullptr) : depth,
                              r ? (order ? so.data() : nullptr) : order, r ? sr.data() : n_reached, poison);
    });
}

extern "C" int cz_sssp_multi(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                             int n_gpus, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent,
                             const volatile uint8_t *poison) {
    if (N == 0 || n_starts == 0) return CZ_OK;
    if (!out_offsets || !starts || !dist || !parent || (E && (!out_targets || !weights))) return cz::set_error(CZ_E_INVALID, "null argument");
    if (out_offsets[N] != E) return cz::set_error(CZ_E_INVALID, "out_offsets[N] (%u) != E (%llu)", out_offsets[N], (unsigned long long)E);
    return run_on_devices(n_gpus, [&](int r, cz_comm *c) {
        RowShard sh(out_offsets, N, r, c->world);
        const size_t full = (size_t)n_starts * N;
        std::vector<float> sdist;
        std::vector<uint32_t> spar;
        if (r) {
            sdist.resize(full);
            spar.resize(full);
        }
        return cz_sssp_sharded(c, sh.off.data(), out_targets + out_offsets[sh.rb], weights + out_offsets[sh.rb], N, sh.rb, sh.re,
                               sh.off.back(), starts, n_starts, r ? sdist.data() : dist, r ? spar.data() : parent, poison);
    });
}

extern "C" int cz_connected_components_multi(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E, int n_gpus,
                                             uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison) {
    if (n_groups) *n_groups = 0;
    if (N == 0) return CZ_OK;
    if (!offsets || !group || (E && !targets)) return cz::set_error(CZ_E_INVALID, "null argument");
    if (offsets[N] != E) return cz::set_error(CZ_E_INVALID, "offsets[N] (%u) != E (%llu)", offsets[N], (unsigned long long)E);
    return run_on_devices(n_gpus, [&](int r, cz_comm *c) {
        RowShard sh(offsets, N, r, c->world);
        std::vector<uint32_t> sg;
        uint32_t k = 0;
        if (r) sg.resize(N);
        const int rc = cz_connected_components_sharded(c, sh.off.data(), targets + offsets[sh.rb], N, sh.rb, sh.re, sh.off.back(),
                                                       r ? sg.data() : group, r ? &k : n_groups, nullptr, poison);
        return rc;
    });
}

// ------------------------------------------------------------------------------------------------------------------
// hnsw_knn over an index partitioned into one independent sub-index per rank (BASELINE.json configs[3])
// ------------------------------------------------------------------------------------------------------------------
namespace {

// per query: `world` sorted lists of k (key = order-preserving image of the f64 distance, id = global node id or
// 0xFFFF... for an empty slot) -> the k smallest by (key, id).  One workgroup per query, one thread per list entry:
// rank = sum over lists of the lower bound of (key, id) in that list.
__global__ void __launch_bounds__(256)
shard_merge_kernel(const double *__restrict__ all_dist /* [world][B][k] */, const uint64_t *__restrict__ all_ids, uint32_t world,
                   uint32_t B, uint32_t k, uint64_t *__restrict__ out_ids, double *__restrict__ out_dist,
                   uint32_t *__restrict__ out_count) {
    const uint32_t q = blockIdx.x;
    __shared__ uint32_t found;
    if (threadIdx.x == 0) found = 0;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(size_t)q * k + i] = ~0ull;
        out_dist[(size_t)q * k + i] = __longlong_as_double(0x7FF0000000000000ll);
    }
    __syncthreads();
    const uint32_t total = world * k;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        const uint32_t w = e / k, j = e % k;
        const size_t at = ((size_t)w * B + q) * k + j;
        const uint64_t id = all_ids[at];
        if (id == ~0ull) continue;
        const uint64_t key = czd::dist_key(all_dist[at]);
        uint32_t rank = 0;
        for (uint32_t c = 0; c < world && rank < k; c++) {
            const size_t base = ((size_t)c * B + q) * k;
            uint32_t lo = 0, hi = k;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t mid_id = all_ids[base + mid];
                const uint64_t mid_key = czd::dist_key(all_dist[base + mid]);
                const bool lt = mid_id != ~0ull && (mid_key < key || (mid_key == key && mid_id < id));
                if (lt) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_ids[(size_t)q * k + rank] = id;
            out_dist[(size_t)q * k + rank] = all_dist[at];
            atomicAdd(&found, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out_count[q] = found;
}

__global__ void __launch_bounds__(256)
shard_pack_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ cnt, uint32_t B, uint32_t k, uint64_t id_offset,
                  uint64_t *__restrict__ out) {
    const size_t n = (size_t)B * k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t q = (uint32_t)(i / k), j = (uint32_t)(i % k);
        out[i] = (j < cnt[q] && ids[i] != CZ_NONE) ? (uint64_t)ids[i] + id_offset : ~0ull;
    }
}

}  // namespace

// `entry_status`: what the caller already knows about this rank (a failed allocation of its own): it joins the first agreement
static int cz_hnsw_search_sharded_status(cz_comm *comm, cz_hnsw_index *shard, const float *queries_dev, uint32_t B, uint32_t k,
                                         uint32_t ef, uint64_t id_offset, uint64_t *out_ids_dev, double *out_dist_dev,
                                         uint32_t *out_count_dev, void *stream_, int entry_status) {
    if (!comm) return cz::set_error(CZ_E_INVALID, "null communicator");
    if (B == 0) return CZ_OK;
    int rc = cz::ensure_device();
    if (rc) return rc;
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    if (!entry_status && (!shard || !queries_dev || !out_ids_dev || !out_dist_dev || !out_count_dev))
        entry_status = cz::set_error(CZ_E_INVALID, "null argument");
    hipStream_t stream = (hipStream_t)stream_;
    auto *ix = reinterpret_cast<cz::HnswIndex *>(shard);
    if (!entry_status && ix && ix->f64()) entry_status = cz::set_error(CZ_E_UNSUPPORTED, "the sharded search takes F32 indices");
    const uint32_t world = (uint32_t)comm->world;
    const size_t nk = (size_t)B * k;
    cz::DevBuf<float> q;
    cz::DevBuf<uint32_t> ids, cnt, flag;
    cz::DevBuf<double> all_d;
    cz::DevBuf<uint64_t> all_i;
    // A rank that fails on its own (an allocation, a shard the search refuses) must not leave the others blocked in the next
    // collective: every rank's status is summed through the communicator before each collective step and all ranks
    // leave together.  `agree` returns this rank's own error, or CZ_E_INVALID naming how many OTHER ranks failed.
    auto agree = [&](int mine) -> int {
        if (world <= 1) return mine;
        const std::string my_msg = mine ? std::string(cz_last_error()) : std::string();
        uint32_t h = mine ? 1u : 0u;
        if (!flag.p && flag.alloc(1) != hipSuccess) return mine ? mine : cz::set_error(CZ_E_OOM, "out of device memory");
        if (hipMemcpyAsync(flag.p, &h, 4, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return mine ? mine : cz::set_error(CZ_E_HIP, "status upload failed");
        if (R->AllReduce(flag.p, flag.p, 1, ncclUint32, ncclSum, comm->nccl, stream) != 0) return mine ? mine : cz::set_error(CZ_E_HIP, "status all-reduce failed");
        if (hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return mine ? mine : cz::set_error(CZ_E_HIP, "status download failed");
        if (mine) return cz::set_error(mine, "%s", my_msg.c_str());
        if (h) return cz::set_error(CZ_E_INVALID, "%u other rank(s) failed in this sharded search", h);
        return CZ_OK;
    };
    auto allocs = [&]() -> int {
        CZ_HIP(q.alloc((size_t)B * ix->dim));
        CZ_HIP(ids.alloc(nk));
        CZ_HIP(cnt.alloc(B));
        CZ_HIP(all_d.alloc(nk * world));
        CZ_HIP(all_i.alloc(nk * world));
        if (k == 0 || ef == 0) return cz::set_error(CZ_E_INVALID, "k and ef must be > 0");
        return CZ_OK;
    };
    if ((rc = agree(entry_status ? entry_status : allocs()))) return rc;
    // rank 0's parent tuples go to every shard (B x dim x 4 bytes)
    if (comm->rank == 0) CZ_HIP(hipMemcpyAsync(q.p, queries_dev, (size_t)B * ix->dim * 4, hipMemcpyDeviceToDevice, stream));
    CZ_NCCL(R, R->Broadcast(q.p, q.p, (size_t)B * ix->dim, ncclFloat32, 0, comm->nccl, stream));
    double *my_d = all_d.p + nk * comm->rank;
    uint64_t *my_i = all_i.p + nk * comm->rank;
    rc = cz::hnsw_search_device(ix, q.p, B, k, ef, 0, 0.0, ids.p, my_d, cnt.p, nullptr, stream);
    if ((rc = agree(rc))) return rc;  // (an empty shard -- n < n_gpus -- searches fine: zero rows; a refused one stops every rank here)
    hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)std::min<size_t>(1024, (nk + 255) / 256)), dim3(256), 0, stream, ids.p, cnt.p,
                       B, k, id_offset, my_i);
    if (world > 1) {
        CZ_NCCL(R, R->AllGather(my_d, all_d.p, nk, ncclFloat64, comm->nccl, stream));
        CZ_NCCL(R, R->AllGather(my_i, all_i.p, nk, ncclUint64, comm->nccl, stream));
    }
    hipLaunchKernelGGL(shard_merge_kernel, dim3(B), dim3(256), 0, stream, all_d.p, all_i.p, world, B, k, out_ids_dev, out_dist_dev,
                       out_count_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sharded search launch: %s", hipGetErrorString(e));
    CZ_HIP(hipStreamSynchronize(stream));  // temporaries die with this scope
    return CZ_OK;
}

extern "C" int cz_hnsw_search_sharded(cz_comm *comm, cz_hnsw_index *shard, const float *queries_dev, uint32_t B, uint32_t k,
                                      uint32_t ef, uint64_t id_offset, uint64_t *out_ids_dev, double *out_dist_dev,
                                      uint32_t *out_count_dev, void *stream_) {
    return cz_hnsw_search_sharded_status(comm, shard, queries_dev, B, k, ef, id_offset, out_ids_dev, out_dist_dev, out_count_dev, stream_,
                                         CZ_OK);
}

// ------------------------------------------------------------------------------------------------------------------
// The same partitioned index held by ONE process: sub-index r on GPU r, one host thread and one RCCL communicator per GPU
// for the lifetime of the handle (a cozo process is one process; building the handle is `::hnsw create` over shards, a search
// is cz_hnsw_search_sharded on every device at once).
// ------------------------------------------------------------------------------------------------------------------
struct cz_hnsw_multi {
    int n_gpus = 0;
    uint32_t dim = 0;
    std::vector<ncclComm_t> comms;
    std::vector<cz_hnsw_index *> shards;
    std::vector<uint64_t> id_offset;
};

namespace {

// fn(rank, comm) on every device of the handle at once, each on its own host thread with that device current
template <class F>
int on_devices_of(cz_hnsw_multi *m, F fn) {
    std::vector<int> rcs(m->n_gpus, CZ_OK);
    std::vector<std::string> msgs(m->n_gpus);
    auto worker = [&](int r) {
        cz::t_device_override = r;
        int wrc = cz::ensure_device();
        cz_comm c;
        c.rank = r;
        c.world = m->n_gpus;
        c.device = r;
        c.nccl = m->comms.empty() ? nullptr : m->comms[r];
        if (!wrc) wrc = fn(r, &c);
        if (wrc) {
            rcs[r] = wrc;
            msgs[r] = cz_last_error();
        }
        c.nccl = nullptr;
        cz::t_device_override = -1;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < m->n_gpus; r++) th.emplace_back(worker, r);
    worker(0);
    for (auto &t : th) t.join();
    (void)cz::ensure_device();
    for (int r = 0; r < m->n_gpus; r++)
        if (rcs[r]) return cz::set_error(rcs[r], "GPU %d: %s", r, msgs[r].c_str());
    return CZ_OK;
}

int multi_open(int n_gpus, cz_hnsw_multi **out) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    const int have = cz_device_count();
    if (n_gpus < 1 || n_gpus > have) return cz::set_error(CZ_E_INVALID, "n_gpus = %d, %d device(s) visible", n_gpus, have);
    int rc = cz::ensure_device();
    if (rc) return rc;
    std::unique_ptr<cz_hnsw_multi> m(new cz_hnsw_multi());
    m->n_gpus = n_gpus;
    m->shards.assign(n_gpus, nullptr);
    m->id_offset.assign(n_gpus, 0);
    Rccl *R = nullptr;
    if ((rc = need_rccl(&R))) return rc;
    std::vector<int> devs(n_gpus);
    for (int i = 0; i < n_gpus; i++) devs[i] = i;
    m->comms.assign(n_gpus, nullptr);
    CZ_NCCL(R, R->CommInitAll(m->comms.data(), n_gpus, devs.data()));
    *out = m.release();
    return CZ_OK;
}

}  // namespace

extern "C" void cz_hnsw_multi_destroy(cz_hnsw_multi *m) {
    if (!m) return;
    Rccl *R = rccl();
    for (int r = 0; r < m->n_gpus; r++) {
        (void)hipSetDevice(r);
        if (m->shards[r]) cz_hnsw_index_destroy(m->shards[r]);
        if (R && !m->comms.empty() && m->comms[r] && !getenv("CZ_COMM_NO_DESTROY")) (void)R->CommDestroy(m->comms[r]);
    }
    (void)cz::ensure_device();
    delete m;
}

extern "C" int cz_hnsw_multi_create(const cz_hnsw_desc *const *shards, const float *const *vectors, const uint64_t *id_offsets,
                                    int n_gpus, cz_hnsw_multi **out) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    if (!shards || !vectors || !id_offsets) return cz::set_error(CZ_E_INVALID, "null argument");
    cz_hnsw_multi *m = nullptr;
    int rc = multi_open(n_gpus, &m);
    if (rc) return rc;
    for (int r = 0; r < n_gpus; r++) {
        if (!shards[r] || shards[r]->dim != shards[0]->dim || shards[r]->metric != shards[0]->metric) {
            cz_hnsw_multi_destroy(m);
            return cz::set_error(CZ_E_INVALID, "shard %d: null, or dimension / metric differ from shard 0", r);
        }
        m->id_offset[r] = id_offsets[r];
    }
    m->dim = shards[0]->dim;
    rc = on_devices_of(m, [&](int r, cz_comm *) { return cz_hnsw_index_create(shards[r], vectors[r], &m->shards[r]); });
    if (rc) {
        cz_hnsw_multi_destroy(m);
        return rc;
    }
    *out = m;
    return CZ_OK;
}

extern "C" int cz_hnsw_multi_build(const float *vectors, uint32_t n, uint32_t dim, int metric, uint32_t m_neighbours,
                                   uint32_t ef_construction, int keep_pruned_connections, uint64_t seed, uint32_t max_batch, int n_gpus,
                                   uint32_t flags, uint64_t *n_dist, cz_hnsw_multi **out) {
    if (n_dist) *n_dist = 0;
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    if (n > 0 && !vectors) return cz::set_error(CZ_E_INVALID, "vectors is null");
    if (flags & CZ_DEVICE_PTRS) return cz::set_error(CZ_E_INVALID, "cz_hnsw_multi_build takes host vectors (they go to several devices)");
    cz_hnsw_multi *m = nullptr;
    int rc = multi_open(n_gpus, &m);
    if (rc) return rc;
    m->dim = dim;
    const uint32_t per = (uint32_t)(((uint64_t)n + n_gpus - 1) / n_gpus);
    std::vector<uint64_t> nd(n_gpus, 0);
    rc = on_devices_of(m, [&](int r, cz_comm *) {
        const uint32_t rb = (uint32_t)std::min<uint64_t>(n, (uint64_t)r * per), re = (uint32_t)std::min<uint64_t>(n, (uint64_t)(r + 1) * per);
        m->id_offset[r] = rb;  // rows [rb, re) of the relation in key order: a global id is rb + the shard's node id
        return cz_hnsw_build(vectors + (size_t)rb * dim, re - rb, dim, metric, m_neighbours, ef_construction, keep_pruned_connections,
                             nullptr, seed + (uint64_t)r, max_batch, &nd[r], &m->shards[r], flags, nullptr);
    });
    if (rc) {
        cz_hnsw_multi_destroy(m);
        return rc;
    }
    if (n_dist)
        for (uint64_t v : nd) *n_dist += v;
    *out = m;
    return CZ_OK;
}

extern "C" int cz_hnsw_multi_search(cz_hnsw_multi *m, const float *queries, uint32_t B, uint32_t k, uint32_t ef, uint64_t *ids,
                                    double *dist, uint32_t *count) {
    if (!m || !queries || !ids || !dist || !count) return cz::set_error(CZ_E_INVALID, "null argument");
    if (B == 0) return CZ_OK;
    const size_t nk = (size_t)B * k;
    return on_devices_of(m, [&](int r, cz_comm *c) {
        cz::DevBuf<float> q;
        cz::DevBuf<uint64_t> oi;
        cz::DevBuf<double> od;
        cz::DevBuf<uint32_t> oc;
        // (a rank that fails HERE still enters the collective call below, with null buffers: cz_hnsw_search_sharded rejects them
        // AFTER the ranks have agreed on a status, so nobody is left waiting in a broadcast -- ADVICE r3)
        auto prepare = [&]() -> int {
            CZ_HIP(q.alloc((size_t)B * m->dim));
            CZ_HIP(oi.alloc(nk));
            CZ_HIP(od.alloc(nk));
            CZ_HIP(oc.alloc(B));
            if (r == 0) CZ_HIP(hipMemcpy(q.p, queries, (size_t)B * m->dim * 4, hipMemcpyHostToDevice));  // (the others receive the broadcast)
            return CZ_OK;
        };
        const int prc = prepare();
        int rc = cz_hnsw_search_sharded_status(c, m->shards[r], q.p, B, k, ef, m->id_offset[r], oi.p, od.p, oc.p, nullptr, prc);
        if (rc) return rc;
        if (r == 0) {  // every rank holds the same merged lists
            CZ_HIP(hipMemcpy(ids, oi.p, nk * 8, hipMemcpyDeviceToHost));
            CZ_HIP(hipMemcpy(dist, od.p, nk * 8, hipMemcpyDeviceToHost));
            CZ_HIP(hipMemcpy(count, oc.p, (size_t)B * 4, hipMemcpyDeviceToHost));
        }
        return (int)CZ_OK;
    });
}

extern "C" int cz_hnsw_multi_shards(const cz_hnsw_multi *m, uint64_t *id_offsets /* [n_gpus] or NULL */) {
    if (!m) return 0;
    if (id_offsets)
        for (int r = 0; r < m->n_gpus; r++) id_offsets[r] = m->id_offset[r];
    return m->n_gpus;
}
