// Error plumbing and version of the amtx C ABI (include/amtx.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <utility>

#include "../../include/amtx.h"

static thread_local char g_err[1024] = "";

void amtx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* amtx_last_error(void) { return g_err; }
extern "C" int amtx_version(void) { return 100; }


// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: the grant is tracked per (kernel, device)
// under a mutex (a process-wide `static bool` would leave the kernel at the 64 KiB default on a second GPU and raced between
// threads making their first call).
int amtx_grant_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return AMTX_OK;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(mu);
        size_t& g = granted[std::make_pair(kernel, dev)];
        if (bytes <= g) return AMTX_OK;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) g = bytes;
    }
    if (e != hipSuccess) {
        amtx_set_error("amtx_grant_lds(%zu bytes): %s", bytes, hipGetErrorString(e));
        return AMTX_ERR_HIP;
    }
    return AMTX_OK;
}
