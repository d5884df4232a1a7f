// TEST INFRASTRUCTURE ONLY (tests/test_sanitized_host.py): host-memory stand-ins for the handful of HIP runtime entry points that the
// HOST halves of amt_tools_amd/csrc call, linked into the AddressSanitizer / UBSan build of those host halves (libamtx_san.so, CPU only,
// never shipped, never loaded by the product).  "Device" allocations are plain heap blocks -- so every byte the weight packers, plan
// builders and amtx_of_model_finalize upload is written through instrumented code into instrumented memory -- and a kernel launch
// reports hipErrorNoDevice, which the library turns into its ordinary error return.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

extern "C" {
static hipError_t g_last = hipSuccess;
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t i = 0; i < h; ++i) memcpy((char*)d + i * dp, (const char*)s + i * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { g_last = hipErrorNoDevice; return hipErrorNoDevice; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "no device (sanitizer build: host halves only)"; }
hipError_t hipEventCreate(hipEvent_t*) { return hipErrorNoDevice; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipErrorNoDevice; }
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t) { return hipErrorNoDevice; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
// kernel registration of the host-only objects: nothing to register
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
static dim3 g_grid, g_block; static size_t g_shmem; static hipStream_t g_stream;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) { g_grid = g; g_block = b; g_shmem = sh; g_stream = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) { *g = g_grid; *b = g_block; *sh = g_shmem; *s = g_stream; return hipSuccess; }
}
