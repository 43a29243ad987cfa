// Per-device launch state shared by every launcher.
//
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair, and one process
// may hold a context per GPU and drive them from different threads (the reference's
// one-replica-per-device thread pool, deprecated_package/embedder.py:73-82,208-224).  The
// launchers therefore ask here before every launch that needs more than the default 64 KiB of
// dynamic LDS: the attribute is set once per (device, kernel) up to the largest size seen.
//
// A context is driven by one thread, so the answer is cached per thread: after the first launch of a
// kernel on a device the check is a handful of compares in a thread-local table (no mutex, no map
// walk -- a pass of the encoder makes ~90 launches, the small-batch calls of the reference's shape
// are launch-chain bound).  Only a miss takes the process-wide mutex and the (device, kernel) map.
#include <map>
#include <mutex>
#include <utility>

#include "kernels.h"

namespace {
std::mutex g_mu;
std::map<std::pair<int, const void*>, int> g_lds;  // (device, kernel) -> bytes granted so far

struct Granted {
    const void* kernel;
    int dev, bytes;
};
constexpr int TL_SLOTS = 64;  // kernels x devices one thread launches; beyond that the slow path still answers correctly
thread_local Granted tl_cache[TL_SLOTS];
thread_local int tl_used = 0;
}  // namespace

hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    for (int i = 0; i < tl_used; ++i)
        if (tl_cache[i].kernel == kernel && tl_cache[i].dev == dev) {
            if (bytes <= tl_cache[i].bytes) return hipSuccess;
            break;
        }
    int granted;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        int& have = g_lds[{dev, kernel}];
        if (bytes > have) {
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            have = bytes;
        }
        granted = have;
    }
    for (int i = 0; i < tl_used; ++i)
        if (tl_cache[i].kernel == kernel && tl_cache[i].dev == dev) {
            tl_cache[i].bytes = granted;
            return hipSuccess;
        }
    if (tl_used < TL_SLOTS) tl_cache[tl_used++] = Granted{kernel, dev, granted};
    return hipSuccess;
}
