// common.h -- host-side helpers shared by the libcozo_gpu translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/cozo_gpu.h"

namespace cz {

std::string &last_error_ref();
int set_error(int code, const char *fmt, ...);
int ensure_device();  // CZ_OK or CZ_E_NO_DEVICE; lazily runs cz_init(0)
// the device a worker thread of cz_pagerank_multi drives instead of the process-wide one (-1: none).  HIP's current
// device is per thread and every entry point re-selects it through ensure_device().
extern thread_local int t_device_override;

#define CZ_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            return cz::set_error(_e == hipErrorOutOfMemory ? CZ_E_OOM : CZ_E_HIP, "%s failed: %s (%s:%d)", #expr, \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                       \
        }                                                                                          \
    } while (0)

static inline bool poisoned(const volatile uint8_t *p) { return p && *p; }

// collectives of a cz_comm for the other translation units (comm.hip): in place, stream-ordered; one rank = the identity
enum { COMM_U32 = 0, COMM_U64 = 1, COMM_F32 = 2, COMM_F64 = 3 };
enum { COMM_SUM = 0, COMM_MIN = 1 };
int comm_all_reduce(cz_comm *c, void *buf_dev, size_t count, int dtype, int op, hipStream_t stream);
// every rank's `count` elements at `send` -> recv[rank * count ...] on every rank; in place when send == recv + rank * count
int comm_all_gather(cz_comm *c, const void *send_dev, void *recv_dev, size_t count, int dtype, hipStream_t stream);
int comm_world(const cz_comm *c);
// A table the search kernels read at random (an index' vectors): physically contiguous VRAM when the driver can find it
// (hipExtMallocWithFlags(hipDeviceMallocContiguous): lands the same every time -- profiles/r04_built_vs_created.txt,
// profiles/r05_contiguous_table.txt), plain hipMalloc otherwise or under CZ_TABLE_CONTIGUOUS=0.  Freed with hipFree either
// way.  *contiguous (optional) says which one it was; CZ_TABLE_TRACE=1 says it on stderr.
inline hipError_t alloc_table(void **p, size_t bytes, bool *contiguous = nullptr) {
    const char *e = getenv("CZ_TABLE_CONTIGUOUS");
    const bool trace = getenv("CZ_TABLE_TRACE") != nullptr;
    if (contiguous) *contiguous = false;
    if ((!e || atoi(e) != 0) && bytes >= (64u << 20)) {
        if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) {
            if (contiguous) *contiguous = true;
            if (trace) fprintf(stderr, "[table] %.2f GB contiguous\n", bytes / 1e9);
            return hipSuccess;
        }
        (void)hipGetLastError();
        if (trace) fprintf(stderr, "[table] %.2f GB: no contiguous range, plain hipMalloc\n", bytes / 1e9);
    }
    return hipMalloc(p, bytes);
}
// The other arrays a search step reads at random places -- link tables, the visited workspaces: their fetches sit on the step's
// dependent chain (candidate -> link row -> visited -> rows), so where THEY land shows up as latency just like the vector table's
// landing does.  Same policy from 1 MB up (CZ_AUX_CONTIGUOUS=0: plain hipMalloc).
inline hipError_t alloc_aux(void **p, size_t bytes) {
    const char *e = getenv("CZ_AUX_CONTIGUOUS");
    if ((!e || atoi(e) != 0) && bytes >= (1u << 20)) {
        if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
    }
    return hipMalloc(p, bytes);
}
// A table that had to be allocated while the one it replaces was still held (an insert: the old rows are copied over) may
// have missed its contiguous range only because of that.  Once the old table is gone: try again, move the rows, free the
// first copy.  Leaves *table alone when the second attempt fails too.  -> the table is in a contiguous range now
inline bool rehome_table(float **table, size_t bytes, hipStream_t stream) {
    const char *e = getenv("CZ_TABLE_CONTIGUOUS");
    if ((e && atoi(e) == 0) || bytes < (64u << 20)) return false;
    void *q = nullptr;
    if (hipExtMallocWithFlags(&q, bytes, hipDeviceMallocContiguous) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (hipMemcpyAsync(q, *table, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(q);
        return false;
    }
    (void)hipFree(*table);
    *table = (float *)q;
    if (getenv("CZ_TABLE_TRACE")) fprintf(stderr, "[table] %.2f GB moved into a contiguous range\n", bytes / 1e9);
    return true;
}
int comm_rank(const cz_comm *c);

// RAII device buffer (freed on scope exit unless released)
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        reset();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
    T *release() {
        T *r = p;
        p = nullptr;
        n = 0;
        return r;
    }
};

// The same for scratch that a call allocates and frees again: from the device's stream-ordered pool, which is told to keep
// what is freed (release threshold = max) -- hipMalloc / hipFree of a few hundred MB per call is driver work (map / unmap)
// that grows with what the process has mapped already.
inline void pool_keep_freed() {
    static const bool once = [] {
        int dev = 0;
        hipMemPool_t pool = nullptr;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        return true;
    }();
    (void)once;
}
template <typename T>
struct PoolBuf {
    T *p = nullptr;
    size_t n = 0;
    PoolBuf() = default;
    PoolBuf(const PoolBuf &) = delete;
    PoolBuf &operator=(const PoolBuf &) = delete;
    ~PoolBuf() { reset(); }
    void reset() {
        if (p) (void)hipFreeAsync(p, nullptr);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        reset();
        pool_keep_freed();
        if (count == 0) count = 1;
        hipError_t e = hipMallocAsync((void **)&p, count * sizeof(T), nullptr);
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
};

}  // namespace cz
