// Read-only HBM streaming rate of this box: what a kernel that only reads (the pitch-head GEMM, the scaling kernels) can be held against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_dbg/hbm_read tools/hbm_read.hip && tools/_dbg/hbm_read
// Grid-stride 16-byte loads, U of them in flight per thread, XOR-reduced so that nothing is optimised away; 4 GiB per pass (16 x the L2 + MALL).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, size_t n, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same bytes fetched in MFMA A-fragment shape: a wave instruction reads 16 rows x 64 bytes (lane = row + 16 x 16-byte chunk) of 128-byte
// rows, the next one the rows' other halves -- what a GEMM that loads its A operand straight into registers issues
template <int U>
__global__ __launch_bounds__(256) void rd_frag(const uint4* __restrict__ p, size_t n, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
    const size_t lo = (size_t)(lane & 15) * 8 + (lane >> 4);          // uint4 index inside a 16-row x 128-byte group (2 KiB = 128 uint4)
    unsigned acc = 0;
    for (size_t grp = wave; grp + (U / 2 - 1) * nwaves < n / 128; grp += (U / 2) * nwaves) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U / 2; ++u) {
            v[2 * u] = p[(grp + u * nwaves) * 128 + lo];
            v[2 * u + 1] = p[(grp + u * nwaves) * 128 + lo + 4];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U>
void run_frag(const uint4* p, size_t n, unsigned* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rd_frag<U>, dim3(blocks), dim3(256), 0, 0, p, n, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rd_frag<U>, dim3(blocks), dim3(256), 0, 0, p, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  fragment-shaped, %d loads in flight per thread, %5d blocks: %.2f TB/s\n", U, blocks, 5.0 * n * 16 / (ms * 1e-3) / 1e12);
}

template <int U>
void run(const uint4* p, size_t n, unsigned* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rd<U>, dim3(blocks), dim3(256), 0, 0, p, n, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rd<U>, dim3(blocks), dim3(256), 0, 0, p, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %d loads in flight per thread, %5d blocks: %.2f TB/s\n", U, blocks, 5.0 * n * 16 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    uint4* p; unsigned* out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 1, bytes);
    printf("read-only streaming, 4 GiB per pass:\n");
    for (int blocks : {1024, 2048, 4096, 8192}) { run<4>(p, n, out, blocks); run<8>(p, n, out, blocks); run<16>(p, n, out, blocks); }
    for (int blocks : {2048, 4096, 8192}) { run_frag<8>(p, n, out, blocks); run_frag<16>(p, n, out, blocks); }
    return 0;
}
