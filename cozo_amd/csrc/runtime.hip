// runtime.hip -- device selection, error reporting (C ABI: cz_init / cz_shutdown / cz_last_error ...)
#include <atomic>
#include <cstring>
#include <mutex>

#include "common.h"

namespace cz {

std::string &last_error_ref() {
    static thread_local std::string e;
    return e;
}

int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

thread_local int t_device_override = -1;
static std::atomic<int> g_device{-1};
static std::mutex g_init_mu;

static int init_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(CZ_E_NO_DEVICE, "no HIP device visible (%s); libcozo_gpu has no CPU fallback",
                         e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return set_error(CZ_E_INVALID, "device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    CZ_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(CZ_E_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device,
                         prop.gcnArchName);
    CZ_HIP(hipSetDevice(device));
    g_device.store(device);
    return CZ_OK;
}

int ensure_device() {
    int d = g_device.load();
    if (d >= 0 && t_device_override >= 0) d = t_device_override;
    if (d >= 0) {
        // HIP's current device is per thread: make this thread target the selected GPU
        hipError_t e = hipSetDevice(d);
        if (e != hipSuccess) return set_error(CZ_E_HIP, "hipSetDevice(%d): %s", d, hipGetErrorString(e));
        return CZ_OK;
    }
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_device.load() >= 0) return CZ_OK;
    return init_device(0);
}

}  // namespace cz

extern "C" {

int cz_init(int device) {
    std::lock_guard<std::mutex> lk(cz::g_init_mu);
    return cz::init_device(device);
}

void cz_comm_multi_shutdown(void);  // comm.hip: the communicators the single-process *_multi forms keep between calls
void cz_shutdown(void) {
    cz_comm_multi_shutdown();
    cz::g_device.store(-1);
}

int cz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *cz_last_error(void) { return cz::last_error_ref().c_str(); }

const char *cz_version(void) { return "cozo_gpu 0.1.0 (gfx950)"; }
}
