// Prototype: bf16 GEMM C[M][N] = A[M][K] . W[N][K]^T, 256 x 256 block tile, FOUR waves (one per SIMD, 512-register budget), every wave
// a 128 x 128 output tile (256 accumulator registers), 32-deep k-stages in a four-buffer LDS ring fed by global_load_lds, counted waits.
#include "../../amt_tools_amd/csrc/amtx_common.h"
#include <cstdio>
#include <vector>
#include <cmath>

typedef unsigned gb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int TBM = 256, TBN = 256, BK = 32, NST = 4;
constexpr int STAGE = (TBM + TBN) * BK * 2;          // 32 KiB
constexpr int ROWB = BK * 2;                         // 64 bytes per row: 4 chunks of 16 B

// 64-byte rows: fragment read = 16 rows x 16 B at chunk c = lane >> 4.  Swizzle chunk with (row >> 1) & 3 -> 16 rows hit distinct 16-byte
// slots of 8 x 64 B = 512 B?  (checked on the device by SQ_LDS_BANK_CONFLICT)
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 3; }

__global__ __launch_bounds__(256, 1) void gemm_big(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K,
                                                   int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbn = N / TBN;
    const int nk = K / BK;
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem);

    // DMA: a stage has 256 A rows + 256 W rows of 64 B = 512 rows; one instruction moves 16 rows (64 lanes x 16 B: 4 lanes per row);
    // wave w issues rows [64 w, 64 w + 64) of A and of W: 4 + 4 instructions per stage
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n0 = (tile % nbn) * TBN;
        const int64_t m0 = (int64_t)(tile / nbn) * TBM;
        const bf16_t* a_src[4];
        const bf16_t* w_src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 64 + i * 16 + (lane >> 2);
            const int c = (lane & 3) ^ swz(row);
            int64_t mr = m0 + row; if (mr >= M) mr = M - 1;
            a_src[i] = A + mr * K + c * 8;
            w_src[i] = W + (int64_t)(n0 + row) * K + c * 8;
        }
        auto issue = [&](int kt) {
            const unsigned sb = lds0 + (kt % NST) * STAGE + wave * 64 * ROWB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                glds16(a_src[i] + kt * BK, sb + i * 1024);
                glds16(w_src[i] + kt * BK, sb + TBM * ROWB + i * 1024);
            }
        };
        uint4 af[2][8], wf[2][8];
        auto read = [&](int kt, int set) {
            const char* b = smem + (kt % NST) * STAGE;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int ar = wm * 128 + t * 16 + frow;
                af[set][t] = *reinterpret_cast<const uint4*>(b + ar * ROWB + ((fchunk ^ swz(ar)) << 4));
                const int wr = wn * 128 + t * 16 + frow;
                wf[set][t] = *reinterpret_cast<const uint4*>(b + TBM * ROWB + wr * ROWB + ((fchunk ^ swz(wr)) << 4));
            }
        };
        f32x4_t acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        __syncthreads();                 // everybody is done with the previous tile's stages
        issue(0); if (nk > 1) issue(1); if (nk > 2) issue(2);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read(0, 0);
#define GB_STEP(KT, CUR, NXT)                                                                                       \
        do {                                                                                                        \
            /* tile KT + 1 must have landed before it is read below; tiles KT + 2 (and KT + 3, issued next) stay in flight */ \
            if ((KT) + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
            __syncthreads();                                                                                        \
            if ((KT) + 3 < nk) issue((KT) + 3);                                                                     \
            if ((KT) + 1 < nk) read((KT) + 1, NXT);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            _Pragma("unroll") for (int nt = 0; nt < 8; ++nt)                                                        \
                _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                                                    \
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nt][mt]) : "v"(__builtin_bit_cast(gb_u32x4, wf[CUR][nt])), "v"(__builtin_bit_cast(gb_u32x4, af[CUR][mt])));  \
        } while (0)
        for (int kt = 0; kt < nk; kt += 2) {
            GB_STEP(kt, 0, 1);
            if (kt + 1 < nk) GB_STEP(kt + 1, 1, 0);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results are read by vector instructions below (no interlock)
        // epilogue: D'[n][m]: lane holds m = lane & 15, n = 4 (lane >> 4) + r
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int64_t m = m0 + wm * 128 + mt * 16 + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const int n = n0 + wn * 128 + nt * 16 + 4 * (lane >> 4);
                const f32x4_t v = acc[nt][mt];
                uint2 o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                if (m < M) *reinterpret_cast<uint2*>(C + m * N + n) = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

int main() {
    const int M = 640000, N = 512, K = 3648;
    std::vector<bf16_t> hA((size_t)4096 * K), hW((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f32_to_bf16_rn(rnd());
    for (auto& v : hW) v = f32_to_bf16_rn(rnd() * 0.05f);
    bf16_t *dA, *dW, *dC;
    hipMalloc(&dA, (size_t)M * K * 2); hipMalloc(&dW, (size_t)N * K * 2); hipMalloc(&dC, (size_t)M * N * 2);
    for (int64_t r = 0; r < M; r += 4096) hipMemcpy(dA + r * K, hA.data(), (size_t)std::min<int64_t>(4096, M - r) * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    const int ntiles = ((M + TBM - 1) / TBM) * (N / TBN);
    const size_t lds = (size_t)NST * STAGE;
    hipFuncSetAttribute((const void*)gemm_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(gemm_big, dim3(256), dim3(256), lds, 0, dA, dW, dC, M, N, K, ntiles);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(gemm_big, dim3(256), dim3(256), lds, 0, dA, dW, dC, M, N, K, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%s gemm_big M=%d N=%d K=%d: %.3f ms  %.0f TFLOP/s\n", hipGetErrorString(hipGetLastError()), M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
    // check a few entries
    std::vector<bf16_t> hC((size_t)512 * N);
    hipMemcpy(hC.data(), dC + (size_t)4096 * 3 * N, hC.size() * 2, hipMemcpyDeviceToHost);   // rows 12288.. = hA rows 0..
    double maxerr = 0;
    for (int m = 0; m < 512; m += 37)
        for (int n = 0; n < N; n += 29) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)bf16_to_f32(hA[(size_t)m * K + k]) * bf16_to_f32(hW[(size_t)n * K + k]);
            maxerr = std::max(maxerr, std::fabs(acc - bf16_to_f32(hC[(size_t)m * N + n])) / (std::fabs(acc) + 1e-2));
        }
    printf("max rel err %.3g\n", maxerr);
    return 0;
}
