// Per-device launch state shared by every launcher.
//
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair, and one process
// may hold a context per GPU and drive them from different threads (the reference's
// one-replica-per-device thread pool, deprecated_package/embedder.py:73-82,208-224).  The
// launchers therefore ask here before every launch that needs more than the default 64 KiB of
// dynamic LDS: the attribute is set once per (device, kernel) up to the largest size seen, under a
// mutex, keyed by the device that is current on the calling thread.
#include <map>
#include <mutex>
#include <utility>

#include "kernels.h"

namespace {
std::mutex g_mu;
std::map<std::pair<int, const void*>, int> g_lds;  // (device, kernel) -> bytes granted so far
}  // namespace

hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_mu);
    int& have = g_lds[{dev, kernel}];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}
