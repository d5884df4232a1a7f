// Microbenchmark: the matrix rate gfx950 SUSTAINS under a dense bf16 MFMA stream (the yardstick bench.py's
// `frac_of_sustained_matrix_rate` divides by; profiles/mfma_sustained.json is this program's output on an MI355X).
//
// Every wave keeps NACC independent accumulator tiles and issues v_mfma_f32_16x16x32_bf16 back to back out of registers -- no LDS, no memory
// traffic inside the loop, so the result is the pipe's own rate at the clock the chip settles to under its power budget (DVFS), which depends on
// the operand DATA: zeros toggle no multiplier bits and clock higher than random values.  Reported for both, plus the 32x32x16 shape.
// Build + run: python tools/mfma_sustained.py   (hipcc --offload-arch=gfx950 -O3 tools/mfma_sustained.hip -o <tmp>/mfma_sustained)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(const uint4* __restrict__ ops, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a[4], b[2];                                    // NACC <= 8 distinct (a, b) pairs: no two accumulators compute the same tile
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(u32x4, ops[i * 64 + lane]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(u32x4, ops[(4 + i) * 64 + lane]);
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            // inline asm: with the builtin hipcc 7.2 allocates this loop's accumulators as a[24:27] <- a[22:25] style shifted ranges (a dependent
            // chain); pinned in-place accumulation is what the product kernels' loops compile to.  Eight independent tiles between two uses of
            // one accumulator cover the 8-pass latency without wait states
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + 2.f * acc[i][1] + 3.f * acc[i][2] + 5.f * acc[i][3] * (float)(i + 1);
    if (s == 12345.678f) sink[blockIdx.x] = s;          // keeps the loop alive; practically never true
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma32_kernel(const uint4* __restrict__ ops, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a[4], b[2];                                    // NACC <= 8 distinct (a, b) pairs: no two accumulators compute the same tile
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(u32x4, ops[i * 64 + lane]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(u32x4, ops[(4 + i) * 64 + lane]);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 2) & 1]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) sink[blockIdx.x] = s;
}

static uint16_t bf16_of(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char** argv) {
    const double target_ms = argc > 1 ? atof(argv[1]) : 300.0;      // long enough for the clock to settle
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint4* d_ops;
    float* d_sink;
    CHECK(hipMalloc(&d_ops, 6 * 64 * sizeof(uint4)));
    CHECK(hipMalloc(&d_sink, 65536 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz_max\": %d, \"runs\": [", prop.name, prop.gcnArchName, cus, prop.clockRate / 1000);
    bool first = true;
    for (int data = 0; data < 2; ++data) {
        std::vector<uint16_t> h(6 * 64 * 8);
        uint32_t st = 12345u;
        for (auto& v : h) {
            st = st * 1664525u + 1013904223u;
            v = data ? bf16_of(((st >> 8) / 8388608.0f) - 1.0f) : 0;      // uniform in [-1, 1) or zeros
        }
        CHECK(hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int shape = 0; shape < 2; ++shape) {
            for (int wpc = 4; wpc <= 8; wpc += 4) {                        // waves per CU: one or two per SIMD
                const int blocks = cus * (wpc / 4);
                const double flops_per_mfma = shape ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
                const int nacc = shape ? 4 : 8;
                auto launch = [&](int iters) {
                    if (shape) hipLaunchKernelGGL(mfma32_kernel<4>, dim3(blocks), dim3(256), 0, 0, d_ops, d_sink, iters);
                    else hipLaunchKernelGGL(mfma16_kernel<8>, dim3(blocks), dim3(256), 0, 0, d_ops, d_sink, iters);
                };
                int iters = 20000;
                launch(iters);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                launch(iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                iters = (int)(iters * target_ms / ms);                     // one long launch: ~target_ms of uninterrupted MFMA issue
                CHECK(hipEventRecord(e0));
                launch(iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double tf = flops_per_mfma * nacc * (double)iters * blocks * 4 / (ms * 1e-3) / 1e12;
                // implied clock: a CU's four matrix pipes retire 4 x 1024 bf16 FLOP per cycle at the dense peak (2.5 PF / 256 CUs / 2.4 GHz)
                const double mhz = tf * 1e12 / (cus * 4096.0) / 1e6;
                printf("%s{\"data\": \"%s\", \"shape\": \"%s\", \"waves_per_cu\": %d, \"ms\": %.2f, \"tflops\": %.1f, \"implied_mfma_clock_mhz\": %.0f}", first ? "" : ", ",
                       data ? "random" : "zeros", shape ? "32x32x16" : "16x16x32", wpc, ms, tf, mhz);
                first = false;
                fflush(stdout);
            }
        }
    }
    printf("]}\n");
    return 0;
}
