// The whole convolution stack of the Onsets & Frames acoustic model in ONE kernel for gfx950
// (amt_tools/models/onsetsframes.py:375-412: layer1 Conv(1->32)+BN+ReLU, layer2 Conv(32->32)+BN+ReLU+MaxPool(1,2),
// layer3 Conv(32->64)+BN+ReLU+MaxPool(1,2); Dropout is the identity in eval mode), bf16 operands, fp32 accumulation.
//
// Why: as two kernels (conv.hip) the 32-channel map behind layer2 -- 14.6 kB per frame, 9.3 GB per 1024 clips -- is written to
// HBM by one kernel and read back by the next: 37 % of all bytes a forward pass moves, and the kernels run power-managed at
// 1.8 GHz under that traffic.  Here neither intermediate map exists outside LDS.
//
// Shape of the kernel.  One 512-thread block per CU, wave-specialised: waves 0-3 ("producers") run layer1 + layer2, waves
// 4-7 ("consumers") run layer3, so that each SIMD hosts one wave of either kind and their (equal) matrix work shares the SIMD's
// matrix pipe; every wave keeps ITS layers' folded weights stationary in registers (producer 104, consumer 144 VGPRs), which is
// why the split is by layer: one wave cannot hold both sets.
// A block owns "strips" = (head, clip, 46 consecutive frames) and STREAMS each strip along the frequency axis in steps of 16
// layer2 columns (= 8 pooled columns = 4 output columns):
//     step k:   producers   layer1 -> a1 ring (50 rows x 18 columns x 32 ch)      | consumers  layer3 on the a2 columns of step k-1
//               barrier
//               producers   layer2 (a1 ring) -> a2 ring (48 rows x 18 pooled cols) | consumers  ... -> 16-channel x 64-byte stores
//               barrier
// Rings are indexed by a running column counter (mod 18), so the two columns a 3x3 window needs from the previous step are simply
// still there, and a strip's last step flows into the next strip's first one without a drain.  Streaming along frequency means
// there is NO halo in frequency at all (the zero padding at both ends is the real padding); in time a strip computes 48 layer2
// rows for 46 layer3 rows (+4 %) and 50 layer1 rows.  MFMA N-tile = 16 frames at one frequency column, "swapped" product
// D = W . X^T exactly as in conv.hip (a lane ends up with consecutive channels of one position: MaxPool over the frequency pair,
// ReLU and the channels-last store are lane-local); the accumulation order of every output equals conv.hip's, so the result is
// BIT-IDENTICAL to the two-kernel path (tests/test_gpu_model.py::test_fused_conv_stack_is_bit_identical_to_the_two_kernel_path).
//
// LDS (124 KB of the CU's 160): a1 ring 4 chunk planes x 50 x 19 x 16 B, a2 ring 4 x 48 x 19 x 16 B (chunk-major: 16-byte chunk c of
// position (row, slot) at c * PLANE + (row * 19 + slot) * 16; row pitch 19 is odd, so the 16 rows of a ds_read_b128 lane group hit 16
// distinct 16-byte bank groups, and the 8 rows of a ds_write_b128 lane group 8 distinct ones), two bf16 feature slabs (52 rows x 20
// columns, dB-scaled while they are staged: amtx_of_forward_power), layer3's shift table.

#include "amtx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int RT = 3;                 // 16-row tiles per strip
constexpr int R2 = 16 * RT;           // layer2 rows per strip (48): strip row i2 <-> frame r0 - 1 + i2
constexpr int R3 = R2 - 2;            // layer3 (output) rows per strip (46): o <-> frame r0 + o
constexpr int R1 = R2 + 2;            // layer1 rows (50): i1 <-> frame r0 - 2 + i1
constexpr int RF = R1 + 2;            // feature rows (52): fi <-> frame r0 - 3 + fi
constexpr int CS = 16;                // layer2 columns per step
constexpr int RC = CS + 2;            // ring columns: a1 = 16 new + 2 carried, a2 = 2 x 8 + 2 carried
constexpr int PITCH = RC + 1;         // 19 slots per ring row (odd)
constexpr int ROWB = PITCH * 16;      // bytes between rows of one chunk plane
constexpr int PLANE1 = (R1 * ROWB + 255) / 256 * 256;
constexpr int PLANE2 = (R2 * ROWB + 255) / 256 * 256;
constexpr int A1_OFF = 0;
constexpr int A2_OFF = 4 * PLANE1;
constexpr int SLAB_COLS = CS + 4;     // 20: feature columns 16 j - 2 .. 16 j + 17 of step j (the last two only ever meet zero weights)
constexpr int SLAB_LOAD = CS + 2;     // 18 of them are loaded
constexpr int SLABP = SLAB_COLS * 2;  // 40 bytes per slab row = 8 x odd: the 17 rows of a ds_read_b64 lane group on distinct banks
constexpr int SLAB_BYTES = (RF * SLABP + 15) / 16 * 16;
constexpr int SLAB_OFF = A2_OFF + 4 * PLANE2;
constexpr int SH3_OFF = SLAB_OFF + 2 * SLAB_BYTES + 128;   // + slack: the halo unit's discarded lanes read past their slab
constexpr int LDS_BYTES = SH3_OFF + 64 * 4;
constexpr int FITEMS = RF * SLAB_LOAD;                     // feature values per step (936)
constexpr int FPRE = (FITEMS + 255) / 256;                 // per producer thread (4)
constexpr int NUNITS = RT * (CS / 4) + 1;                  // layer1 units per step: 16 rows x 4 columns each + one unit for rows 48, 49

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void settle(const uint4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void settle(float v) { asm volatile("" ::"v"(v)); }

struct ConvFArgs {
    const float* feats; int64_t f_stride_b, f_stride_t, f_stride_f;   // (B, T, F) view of the one-channel features (or raw power)
    const float* f_clip_max; const float* f_ref;                      // raw power: dB-scaled while staged (null: features as is)
    const bf16_t* w1frag; int64_t w1_gs; const float* shift1;         // amtx_conv1_pack_host(c_in = 1), [groups][32]
    const bf16_t* w2frag; int64_t w2_gs; const float* shift2;         // amtx_conv3x3_pack_host(32), [groups][32]
    const bf16_t* w3frag; int64_t w3_gs; const float* shift3;         // amtx_conv3x3_pack_host(64), [groups][64]
    bf16_t* out; int64_t out_gs;                                      // [groups][B][T][F / 4][64]
    int B, T, F;
};

struct StepCoord { int b, r0, j; };
__device__ __forceinline__ StepCoord step_coord(int kk, int first, int nstep, int ntt) {
    const int si = kk / nstep;
    const int strip = first + si;
    StepCoord c;
    c.j = kk - si * nstep;
    c.b = strip / ntt;
    c.r0 = (strip - c.b * ntt) * R3;
    return c;
}

__device__ __forceinline__ int ring(int x) { return x >= RC ? x - RC : x; }   // x in [0, 2 RC)

__global__ __launch_bounds__(512, 2) void convf_kernel(ConvFArgs a, int nstep, int ntt, int nstrips, int per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.y;
    const int g = lane >> 4, trow = lane & 15;
    const int first = blockIdx.x * per_block;
    const int mine = min(per_block, nstrips - first);
    if (mine <= 0) return;
    const int K = mine * nstep;                       // steps of this block; iteration K only drains the consumers
    const int F2 = a.F >> 1, F4 = F2 >> 1;
    const bool db = a.f_clip_max != nullptr;

    if (wave < 4) {
        // =================================================================== producers: layer1 + layer2
        uint4 w1t[4][2];
        uint4 wf[9][2];
        f32x4_t sh1[2], sh2[2];
        {
            const uint4* wp = reinterpret_cast<const uint4*>(a.w1frag + (int64_t)grp * a.w1_gs) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) w1t[q][nt] = wp[(q * 2 + nt) * 64];
            const uint4* w = reinterpret_cast<const uint4*>(a.w2frag + (int64_t)grp * a.w2_gs) + lane;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) wf[tap][nt] = w[(tap * 2 + nt) * 64];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sh1[nt][r] = a.shift1[(int64_t)grp * 32 + g * 8 + 4 * nt + r];
                    sh2[nt][r] = a.shift2[(int64_t)grp * 32 + g * 8 + 4 * nt + r];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) settle(w1t[q][nt]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) settle(wf[tap][nt]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { settle(sh1[nt][r]); settle(sh2[nt][r]); }
        }
        // the two slab columns no step ever stages (they meet zero weights, but NaN x 0 is NaN): zero once, both slabs
        for (int i = tid; i < 2 * RF * 2; i += 256) {
            const int s = i / (RF * 2), r = i % (RF * 2);
            *reinterpret_cast<uint16_t*>(smem + SLAB_OFF + s * SLAB_BYTES + (r >> 1) * SLABP + (SLAB_LOAD + (r & 1)) * 2) = 0;
        }

        float fpre[FPRE];
        float fown = 0.f, fref = 0.f;
        // feature values of step kk -> registers (consumed a whole step later)
#define CONVF_ISSUE_LOADS(KK)                                                                               \
        do {                                                                                                \
            const StepCoord sc_ = step_coord((KK), first, nstep, ntt);                                      \
            const float* fb = a.feats + (int64_t)sc_.b * a.f_stride_b;                                      \
            int tid_l = tid;                                                                                \
            asm volatile("" : "+v"(tid_l)); /* (row, column) of the cells recomputed per step, not kept in registers */ \
            _Pragma("unroll") for (int n = 0; n < FPRE; ++n) {                                              \
                const int it = tid_l + 256 * n;                                                             \
                const int fi = it / SLAB_LOAD, fc = it - fi * SLAB_LOAD;                                    \
                const int t = sc_.r0 - 3 + fi, f = CS * sc_.j - 2 + fc;                                     \
                fpre[n] = db ? -1.f : 0.f; /* power is never negative: -1 marks the zero padding */         \
                if (it < FITEMS && t >= 0 && t < a.T && f >= 0 && f < a.F) fpre[n] = fb[t * a.f_stride_t + f * a.f_stride_f]; \
            }                                                                                               \
            if (db) {                                                                                       \
                fown = a.f_clip_max[sc_.b];                                                                 \
                fref = a.f_ref ? a.f_ref[sc_.b] : fown;                                                     \
            }                                                                                               \
        } while (0)
        // registers -> slab (KK & 1) as bf16, dB-scaled on the way when the input is raw power (db_scale_apply: the bits
        // amtx_spec_scale would have written, then the round-to-nearest-even conv.hip applies when it reads its fp32 tile)
#define CONVF_WRITE_SLAB(KK)                                                                                \
        do {                                                                                                \
            DbScale dbs = {0.f, 0.f};                                                                       \
            if (db) dbs = db_scale_make(fown, fref);                                                        \
            char* slab = smem + SLAB_OFF + ((KK) & 1) * SLAB_BYTES;                                         \
            int tid_l = tid;                                                                                \
            asm volatile("" : "+v"(tid_l));                                                                 \
            _Pragma("unroll") for (int n = 0; n < FPRE; ++n) {                                              \
                const int it = tid_l + 256 * n;                                                             \
                const int fi = it / SLAB_LOAD, fc = it - fi * SLAB_LOAD;                                    \
                float v = fpre[n];                                                                          \
                if (db) {                                                                                   \
                    const float sv = db_scale_apply(v, dbs);                                                \
                    v = v < 0.f ? 0.f : sv;                                                                 \
                }                                                                                           \
                if (it < FITEMS) *reinterpret_cast<uint16_t*>(slab + fi * SLABP + fc * 2) = (uint16_t)pack_bf16x2(v, 0.f); \
            }                                                                                               \
        } while (0)

        CONVF_ISSUE_LOADS(0);
        CONVF_WRITE_SLAB(0);
        if (K > 1) CONVF_ISSUE_LOADS(1);
        __syncthreads();

        const int n16 = trow;
        const int gg = min(g, 2);                                  // k-group 3 has zero weights: re-read group 2's row
        const int lane_a1 = A1_OFF + g * PLANE1 + trow * ROWB;      // lane part of an a1 address (fragment reads and layer1 stores)
        const int lane_a2 = A2_OFF + g * PLANE2 + trow * ROWB;      // lane part of an a2 store address
        typedef short s16x2 __attribute__((ext_vector_type(2)));

        for (int kk = 0; kk <= K; ++kk) {
            const bool live = kk < K;
            StepCoord sc = step_coord(live ? kk : K - 1, first, nstep, ntt);
            const int base1 = (kk * CS) % RC;                       // ring slot of this step's first new a1 column (16 j - 1)
            if (live) {
                if (kk + 1 < K) CONVF_WRITE_SLAB(kk + 1);
                if (kk + 2 < K) CONVF_ISSUE_LOADS(kk + 2);
                // ---------------------------------------------------------------- layer1 (Toeplitz product, see conv.hip)
                // unit u < NUNITS - 1: a1 rows 16 rt + n16, columns 16 j - 1 + 4 xb + q (rt = u / 4, xb = u % 4); the last unit: rows
                // 48 + (n16 & 1), column block n16 >> 1 (lanes with block >= 4 compute garbage that is not stored)
                const char* slab = smem + SLAB_OFF + (kk & 1) * SLAB_BYTES;
                for (int u = wave; u < NUNITS; u += 4) {
                    const bool mainu = u < NUNITS - 1;              // scalar
                    const int rt = u >> 2, xbu = u & 3;
                    const int row1 = mainu ? 16 * rt + n16 : R2 + (n16 & 1);
                    const int xb = mainu ? xbu : (n16 >> 1);
                    const char* fp = slab + (row1 + gg) * SLABP + 8 * xb;
                    const uint2 b0 = *reinterpret_cast<const uint2*>(fp);
                    const uint2 b1 = *reinterpret_cast<const uint2*>(fp + 8);
                    const uint4 bh = make_uint4(b0.x, b0.y, b1.x, b1.y);
                    f32x4_t acc1[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc1[q][nt] = mfma16(w1t[q][nt], bh, sh1[nt]);
                    const int t1 = sc.r0 - 2 + row1;
                    const uint32_t rowmask = (unsigned)t1 < (unsigned)a.T ? 0xffffffffu : 0u;
                    const int c10 = CS * sc.j - 1 + 4 * xb;         // first a1 column of this lane's four positions
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t keep = (unsigned)(c10 + q) < (unsigned)a.F ? rowmask : 0u;
                        uint32_t pk[4];
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            // round first, then ReLU on the packed pairs as a signed 16-bit max with 0, then the zero padding of the map
                            const uint32_t v = pack_bf16x2(acc1[q][h >> 1][2 * (h & 1)], acc1[q][h >> 1][2 * (h & 1) + 1]);
                            pk[h] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0})) & keep;
                        }
                        const int slot = ring(ring(base1 + 4 * xb + q));      // base1 + 4 xb + q < 18 + 16
                        char* dst = smem + A1_OFF + g * PLANE1 + (row1 * PITCH + slot) * 16;
                        if (mainu || xb < CS / 4) *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    }
                }
            }
            lds_only_barrier();
            if (live) {
                // ---------------------------------------------------------------- layer2 + MaxPool(1,2): a1 ring -> a2 ring
                // pooled column p = 8 j - 1 + pi (pi = wave, wave + 4), its four a1 columns at ring offsets 2 pi - 2 + cc behind base1
                const int base2 = (kk * (CS / 2)) % RC;             // ring slot of this step's first pooled column (8 j - 1)
                for (int pi = wave; pi < CS / 2; pi += 4) {
                    const int p = (CS / 2) * sc.j - 1 + pi;
                    const bool pvalid = p >= 0 && p < F2;           // scalar
                    int addr[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) addr[cc] = lane_a1 + ((base1 + RC + 2 * pi - 2 + cc) % RC) * 16;
                    const int slot2 = ring(base2 + pi);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int t2 = sc.r0 - 1 + 16 * rt + trow;
                        uint4 o = make_uint4(0, 0, 0, 0);
                        if (pvalid && sc.r0 - 1 + 16 * rt < a.T) {
                            uint4 x[3][4];
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc) x[kh][cc] = *reinterpret_cast<const uint4*>(smem + addr[cc] + (16 * rt + kh) * ROWB);
                            f32x4_t acc[2][2];
#pragma unroll
                            for (int e = 0; e < 2; ++e)
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt) acc[e][nt] = sh2[nt];
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        const int kw = cc - e;
                                        if (kw < 0 || kw > 2) continue;
#pragma unroll
                                        for (int nt = 0; nt < 2; ++nt) acc[e][nt] = mfma16(wf[kh * 3 + kw][nt], x[kh][cc], acc[e][nt]);
                                    }
                            float v[8];
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = fmaxf(fmaxf(acc[0][nt][r], acc[1][nt][r]), 0.f);
                            const uint32_t rowmask = (unsigned)t2 < (unsigned)a.T ? 0xffffffffu : 0u;
                            o = make_uint4(pack_bf16x2(v[0], v[1]) & rowmask, pack_bf16x2(v[2], v[3]) & rowmask,
                                           pack_bf16x2(v[4], v[5]) & rowmask, pack_bf16x2(v[6], v[7]) & rowmask);
                        }
                        *reinterpret_cast<uint4*>(smem + lane_a2 + (16 * rt) * ROWB + slot2 * 16) = o;
                    }
                }
            }
            lds_only_barrier();
        }
#undef CONVF_ISSUE_LOADS
#undef CONVF_WRITE_SLAB
    } else {
        // =================================================================== consumers: layer3 + MaxPool(1,2) -> HBM
        // weight rows re-dealt at load (conv.hip WIDE_ST): store q of the four lane groups covers 64 CONTIGUOUS bytes
        // (channels 32 q + 8 g ..): row (4 gr + r) of tile nt <- row (4 (2 (nt >> 1) + (gr >> 1)) + r) of packed tile 2 (gr & 1) + (nt & 1)
        uint4 wf[9][4];
        {
            const int gr = (lane & 15) >> 2;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int src_lane = (lane & 48) + 4 * (2 * (nt >> 1) + (gr >> 1)) + (lane & 3);
                const uint4* w = reinterpret_cast<const uint4*>(a.w3frag + (int64_t)grp * a.w3_gs) + src_lane;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) wf[tap][nt] = w[(tap * 4 + 2 * (gr & 1) + (nt & 1)) * 64];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) settle(wf[tap][nt]);
        }
        if (tid - 256 < 64) reinterpret_cast<float*>(smem + SH3_OFF)[tid - 256] = a.shift3[(int64_t)grp * 64 + tid - 256];
        const f32x4_t* shl0 = reinterpret_cast<const f32x4_t*>(smem + SH3_OFF);
        __syncthreads();

        const int mi = wave - 4;                                    // this wave's output column within a step
        const int lane_a2 = A2_OFF + g * PLANE2 + trow * ROWB;
        for (int kk = 0; kk <= K; ++kk) {
            const bool live = kk >= 1;
            const StepCoord sc = step_coord(live ? kk - 1 : 0, first, nstep, ntt);
            const int base2 = ((kk - 1 + RC) * (CS / 2)) % RC;      // ring slot of pooled column 8 j - 1 of the step consumed now
            const int m = (CS / 4) * sc.j - 1 + mi;                 // output column; its a2 columns 2 m - 1 + cc at ring offsets 2 mi - 2 + cc
            const bool mvalid = live && m >= 0 && m < F4;
            int addr[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) addr[cc] = lane_a2 + ((base2 + RC + 2 * mi - 2 + cc) % RC) * 16;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (mvalid && sc.r0 + 16 * rt < a.T) {
                    uint4 x[3][4];
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) x[kh][cc] = *reinterpret_cast<const uint4*>(smem + addr[cc] + (16 * rt + kh) * ROWB);
                    f32x4_t acc[2][4];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[e][nt] = shl0[(nt >> 1) * 8 + g * 2 + (nt & 1)];
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int kw = cc - e;
                                if (kw < 0 || kw > 2) continue;
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt) acc[e][nt] = mfma16(wf[kh * 3 + kw][nt], x[kh][cc], acc[e][nt]);
                            }
                    const int o = 16 * rt + trow;
                    const int t = sc.r0 + o;
                    if (o < R3 && t < a.T) {
                        float v[16];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = fmaxf(fmaxf(acc[0][nt][r], acc[1][nt][r]), 0.f);
                        uint4* dst = reinterpret_cast<uint4*>(a.out + (int64_t)grp * a.out_gs + (((int64_t)sc.b * a.T + t) * F4 + m) * 64 + g * 8);
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            dst[4 * q] = make_uint4(pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                                                    pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7]));
                    }
                }
                if (rt == 0) lds_only_barrier();
            }
            lds_only_barrier();
        }
    }
}

}  // namespace

// Steps per strip: the smallest n with 4 (n - 1) + 2 >= F / 4 - 1 (the last output column falls into step n - 1; the zero column
// behind the pooled map, 2 (F / 4) at most, is then inside that step's 8 pooled columns 8 (n - 1) - 1 .. 8 (n - 1) + 6 as well)
static int convf_steps(int F) {
    const int F4 = (F >> 1) >> 1;
    int n = 1;
    while (4 * (n - 1) + 2 < F4 - 1) ++n;
    return n;
}

// One launch handles `groups` heads x B clips x T frames.  The caller decides WHEN (amtx_conv_stack_fused_ok): below ~one strip per
// CU the two-kernel path's finer tiles fill the chip better.
bool amtx_conv_stack_fused_ok(int B, int T, int F, int groups) {
    if (F < 4 || B <= 0 || T <= 0 || groups <= 0) return false;
    const int64_t strips = (int64_t)B * ((T + R3 - 1) / R3) * groups;
    return strips >= 256 && (int64_t)B * ((T + R3 - 1) / R3) < (1ll << 30);
}

int amtx_launch_conv_stack(const ConvArgs& c2, const bf16_t* w3frag, int64_t w3_gs, const float* shift3, void* out, int64_t out_gs,
                           hipStream_t stream) {
    AMTX_REQUIRE(c2.feats && c2.c_in == 1 && c2.w1frag && c2.shift1 && c2.wfrag && c2.shift && w3frag && shift3 && out,
                 "conv_stack: null pointer / not a one-channel input");
    AMTX_REQUIRE(c2.planes == 1 && c2.c_out == 32 && c2.out_type == AMTX_T_BF16, "conv_stack: bf16, 32 -> 32 -> 64 channels only");
    AMTX_REQUIRE(c2.B > 0 && c2.T > 0 && c2.F >= 4 && c2.groups > 0, "conv_stack: bad sizes");
    ConvFArgs a;
    a.feats = c2.feats; a.f_stride_b = c2.f_stride_b; a.f_stride_t = c2.f_stride_t; a.f_stride_f = c2.f_stride_f;
    a.f_clip_max = c2.f_clip_max; a.f_ref = c2.f_ref;
    a.w1frag = c2.w1frag; a.w1_gs = c2.w1_gs; a.shift1 = c2.shift1;
    a.w2frag = c2.wfrag; a.w2_gs = c2.w_gs; a.shift2 = c2.shift;
    a.w3frag = w3frag; a.w3_gs = w3_gs; a.shift3 = shift3;
    a.out = (bf16_t*)out; a.out_gs = out_gs;
    a.B = c2.B; a.T = c2.T; a.F = c2.F;
    const int ntt = (c2.T + R3 - 1) / R3;
    const int64_t nstrips = (int64_t)c2.B * ntt;
    AMTX_REQUIRE(nstrips < (1ll << 30), "conv_stack: grid too large");
    const int nstep = convf_steps(c2.F);
    AMTX_REQUIRE((int64_t)nstep * (nstrips + 1) < (1ll << 31), "conv_stack: too many steps");
    int gx = std::max(1, 256 / c2.groups);             // one resident block per CU in total
    if (gx > nstrips) gx = (int)nstrips;
    const int per_block = (int)((nstrips + gx - 1) / gx);
    gx = (int)((nstrips + per_block - 1) / per_block);
    auto kern = convf_kernel;
    AMTX_GRANT_LDS(kern, LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)c2.groups), dim3(512), LDS_BYTES, stream, a, nstep, ntt, (int)nstrips, per_block);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}
