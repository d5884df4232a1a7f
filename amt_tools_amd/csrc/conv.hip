// Convolution stages of the Onsets & Frames acoustic model for gfx950
// (amt_tools/models/onsetsframes.py:375-416: Conv2d 3x3 pad 1 + BatchNorm2d + ReLU [+ MaxPool(1,2)]).
//
// conv1  (C_in = 1 or 6 -> 32/48): direct VALU kernel, BN folded, channels-last output.
// conv3x3 (C_in = 32 -> 32/64..): implicit GEMM on v_mfma_f32_16x16x32_bf16 with
//   * K = 32 input channels = exactly one MFMA per 3x3 tap,
//   * the folded (BN-scaled) weights stationary in registers for the whole block (9 x C_out/16 fragments),
//   * the input tile (18 x (FT+2) positions x 32 ch) staged once in LDS, every tap is a shifted 16-byte
//     fragment read (row pitch odd + XOR chunk swizzle by row -> 16 distinct 16-byte slots per read),
//   * computed "swapped" (D' = W . X^T): a lane holds 4*NT consecutive output channels of one position,
//     so bias + ReLU + MaxPool(1,2) over the frequency pair and the channels-last store are lane-local.
// Eval-mode BatchNorm is folded on the host: scale into the weights, shift (+ conv bias) into `shift`.
// Algorithmic HBM bytes per output position: 32 ch in + C_out/2 ch out (pooled), element size of the mode.

#include "amtx_f16_names.h"
#include "amtx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int CIN = 32;
constexpr int TT = 16;              // frames per block tile = one MFMA N-tile
constexpr int FT_MAX = 46;          // frequency columns per block tile (even)
constexpr int PITCH = FT_MAX + 3;   // LDS positions per tile row: 49 = 1 (mod 4), see tile_off
constexpr int ROWS = TT + 2;
constexpr int PLANE_BYTES = ROWS * PITCH * 64;
// The fused first conv writes its 32-channel map CHUNK-major: 16-byte chunk c of position (i, j) sits at
// c * CM_PLANE + (i * PITCH + j) * 16.  Its lanes hold (chunk = lane >> 4, column = lane & 15), so every 8-lane group of a
// ds_write_b128 covers 128 contiguous bytes; in the position-major layout of tile_off the same store put 8 lanes on two
// 16-byte bank groups (4-way conflict, 32 LDS cycles per store instead of 13: 39 % of that kernel's LDS cycles were conflict
// cycles, PMC SQ_LDS_BANK_CONFLICT).  Fragment reads stay conflict-free without any XOR: rows are PITCH = 1 (mod 16) slots
// apart and CM_PLANE is a multiple of 256 bytes, so a ds_read_b128 lane group (8 rows of chunk c, the other 8 rows of chunk
// c + 1) lands on 16 distinct slots.
constexpr int CM_PLANE = (ROWS * PITCH * 16 + 255) / 256 * 256;
constexpr int CM_BYTES = 3 * CM_PLANE + ROWS * PITCH * 16;   // the last chunk plane needs no tail padding

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
    return amtx_mfma_16x16x32(a, b, c);
}

// Byte offset of 16-byte chunk c of tile position (row i, column j).  A fragment read takes, per lane,
// row = (lane & 15) + kh and chunk = lane >> 4.  With PITCH = 1 (mod 4) consecutive rows rotate through the four
// 64-byte quarters of a 256-byte bank row; XOR-ing the chunk with 2*((i >> 2) & 1) then makes every
// ds_read_b128 lane group ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte slots for every kh and column
// (found by exhaustive search, tools/lds_swizzle_search.py).
typedef __attribute__((ext_vector_type(4))) short mfma_s16x4;
__device__ __forceinline__ f32x4_t mfma16k16(uint2 a, uint2 b, f32x4_t c) {
    return amtx_mfma_16x16x16(a, b, c);
}

__device__ __forceinline__ int tile_off(int i, int j, int c) { return ((i * PITCH + j) * 4 + (c ^ (((i >> 2) & 1) << 1))) * 16; }

__device__ __forceinline__ void cvt8(const float (&f)[8], bool split, uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        l[i] = 0;
        if (split) split_bf16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
        else h[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Makes the compiler finish the loads that produced v before the persistent loop: otherwise the first use inside the
// loop carries an s_waitcnt vmcnt(0) that also drains the next tile's prefetch loads every iteration.
__device__ __forceinline__ void settle(const uint4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void settle(const uint2& v) { asm volatile("" ::"v"(v.x), "v"(v.y)); }
__device__ __forceinline__ void settle(float v) { asm volatile("" ::"v"(v)); }

#ifdef AMTX_CONV_TIMING
// Debug build only (AMTX_EXTRA_FLAGS=-DAMTX_CONV_TIMING): cycles wave 0 of every block spends per phase of the persistent loop,
// summed over blocks: [0] staging / feature store, [1] barrier after it, [2] first-conv phase, [3] barrier, [4] conv phase,
// [5] trailing barrier, [6] tiles.  Read with amtxdbg_conv_prof().
__device__ unsigned long long g_conv_prof[32];   // [16..22]: sections of the first-conv loop (pack, gather issue, MFMA issue, epilogue 0, epilogue 1, -, iterations)
// layout of the first 16:   // [0..7] fused first conv + conv2, [8..15] plain conv
#define CONV_TICK(SLOT)                                                    \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        prof_acc[SLOT] += now_ - prof_t;                                   \
        prof_t = now_;                                                     \
    } while (0)
#else
#define CONV_TICK(SLOT) do {} while (0)
#endif
// ticks inside the first-conv loops: every s_memtime costs a scalar-memory round trip (hundreds of cycles with lgkmcnt(0) behind
// it), so the per-section numbers are only meaningful relative to each other; off unless asked for
#ifdef AMTX_CONV_TIMING_FINE
#define CONV_TICK_FINE(SLOT) CONV_TICK(SLOT)
#else
#define CONV_TICK_FINE(SLOT) do {} while (0)
#endif

constexpr int FROWS = ROWS + 2;      // feature tile rows of the fused first conv
// Feature tile row pitch (floats).  c_in = 1 (KS = 1, the Toeplitz first conv below): lane (row n, k-group g) reads 8 consecutive
// floats of row n + g with two ds_read_b128; a pitch of 4 floats (mod 64) puts the 16 rows of a lane group on 16 distinct 16-byte
// slots of the 256-byte bank row.  c_in > 1 (KS = 4): the tightest pitch, so that two blocks still fit one CU's LDS.
constexpr int fw_pitch(int ks) { return ks == 1 ? 68 : FT_MAX + 4; }

constexpr int ITEMS = (ROWS * (FT_MAX + 2) * 4 + 255) / 256;   // 16-byte staging items per thread (14)
constexpr int FSLACK = 16;          // floats after the feature tile: gathers of columns past the tile stay in LDS we own
constexpr int FPRE = 4;                                        // prefetched feature values per thread (c_in = 1)

struct TileCoord { int b, t0, f0; };
// `tile` is the virtual id blockIdx.x + k * gridDim.x.  Blocks whose ids agree mod 8 share an XCD (and its L2): xcd_remap
// hands each XCD one contiguous range of tiles, so the tiles in flight on an XCD are spatial neighbours (same clip, adjacent
// frequency / time tiles) and the halo rows and columns they share are read from HBM once instead of once per XCD.
__device__ __forceinline__ TileCoord tile_coord(int tile, int ntf, int ntt, int ft, int ntiles) {
    tile = (int)xcd_remap((unsigned)tile, (unsigned)ntiles);
    TileCoord c;
    const int tf = tile % ntf; tile /= ntf;
    const int tt = tile % ntt; tile /= ntt;
    c.b = tile; c.t0 = tt * TT; c.f0 = tf * ft;
    return c;
}

// Persistent blocks: each block keeps its (folded) weights in registers and walks tiles blockIdx.x, +gridDim.x, ...
// For bf16 inputs the NEXT tile's 14 x 16 B per thread are already in flight (registers) while the current tile is
// on the matrix cores, so the HBM/L2 latency of staging is hidden behind ~3300 MFMA cycles per wave.
// KS: compile-time bound of the fused first conv's K steps (1 for c_in = 1, 4 otherwise; 0 when not fused)
// Threads per block: four waves everywhere.  The kernel is written for any multiple of 64 (NTH / NW below), but more waves per
// SIMD do not pay on gfx950: eight per block need <= 128 VGPRs and spill (72 weight + 48 fragment-ring + 16 accumulator registers
// alone), six (168 VGPRs) leave one block resident per CU (measured 5.9 vs 2.85 ms for the fused c_in = 1 kernel).
constexpr int conv_threads(bool fuse1, int ks, int ns) { return 256; }

template <int NT, int NS, int IN_TYPE, int OUT_TYPE, bool FUSE1, int KS>
__global__ __launch_bounds__(conv_threads(FUSE1, KS, NS), (NS == 1 ? conv_threads(FUSE1, KS, NS) / 128 : 1)) void conv3x3_kernel(ConvArgs a, int ft, int ntf, int ntt, int inv_cols, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int COUT = NT * 16;
    constexpr int NTH = conv_threads(FUSE1, KS, NS), NW = NTH / 64;
    constexpr int PB = FUSE1 ? CM_BYTES : PLANE_BYTES;      // bytes of one input-tile plane in LDS
    constexpr int FW = fw_pitch(KS);
    // next-tile register prefetch only where the register budget keeps 2 waves per SIMD (C_out = 32)
    constexpr bool PREFETCH = (IN_TYPE == AMTX_T_BF16) && !FUSE1 && NT <= 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // dB staging of raw power features (ConvArgs::f_clip_max): c_in = 1 kernels only (compiled out elsewhere: those are out of registers)
    const float* const f_clip_max = (FUSE1 && KS == 1) ? a.f_clip_max : nullptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // the same value, provably uniform: scalar loop counters
    const int grp = blockIdx.y;
    const int g = lane >> 4, trow = lane & 15;

    // ---- stationary weights: 9 taps x NT tiles (x NS planes), one 16-byte fragment per lane each
    uint4 wf[9][NT][NS];
    // C_out = 64, bf16 out: the packed order gives a lane 16 consecutive channels (two 16-byte stores 32 bytes apart per position, four
    // lane groups interleaved).  WIDE_ST re-deals the weight rows once at load so that store q of the four lane groups covers 64
    // CONTIGUOUS bytes (channels 32 q + 8 g ..): row (4 gr + r) of tile nt <- row (4 (2 (nt >> 1) + (gr >> 1)) + r) of packed tile 2 (gr & 1) + (nt & 1)
    constexpr bool WIDE_ST = NT == 4 && NS == 1 && OUT_TYPE == AMTX_T_BF16 && !FUSE1;
    if constexpr (WIDE_ST) {
        const int gr = (lane & 15) >> 2;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int src_lane = (lane & 48) + 4 * (2 * (nt >> 1) + (gr >> 1)) + (lane & 3);
            const uint4* w = reinterpret_cast<const uint4*>(a.wfrag + (int64_t)grp * a.w_gs) + src_lane;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wf[tap][nt][0] = w[(tap * NT + 2 * (gr & 1) + (nt & 1)) * 64];
        }
    } else {
        const uint4* w = reinterpret_cast<const uint4*>(a.wfrag + (int64_t)grp * a.w_gs) + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) wf[tap][nt][p] = w[((tap * NT + nt) * NS + p) * 64];
    }
    // folded BN shift: a C_out-float table at the end of LDS; the accumulators of every column pair are initialised
    // from it (lane -> its 4*NT consecutive channels), so the epilogue is max3(y0, y1, 0) only
    const f32x4_t* shl0 = reinterpret_cast<const f32x4_t*>(smem + NS * PB + (FUSE1 ? (a.c_in * FROWS * FW + FSLACK) * 4 : 0));
    const f32x4_t* shl = shl0 + g * NT;
    if (tid < COUT) reinterpret_cast<float*>(smem + NS * PB + (FUSE1 ? (a.c_in * FROWS * FW + FSLACK) * 4 : 0))[tid] = a.shift[(int64_t)grp * a.shift_gs + tid];

    // C_out = 32: the shift also sits in 8 registers and is the C operand of each pair's first MFMAs directly (the LDS table read
    // at the top of every column pair put ~100 cycles of LDS latency in front of the first MFMA); C_out = 64 has no registers left
    constexpr bool SH_REGS = (NT == 2 && NS == 1 && KS <= 1);   // not the c_in > 1 variant: it is out of registers already
    f32x4_t shr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) shr[nt][r] = SH_REGS ? a.shift[(int64_t)grp * a.shift_gs + (g * NT + nt) * 4 + r] : 0.f;
    if (SH_REGS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) settle(shr[nt][r]);
    }
    const int cols = ft + 2;
    const int npos = ROWS * cols;
    const int Fo = a.F >> 1;
    const int c = tid & 3;                              // staging chunk handled by this thread (256 % 4 == 0)
    const char* in_grp = reinterpret_cast<const char*>(a.in) + (int64_t)grp * a.in_gs * (IN_TYPE == AMTX_T_F32 ? 4 : 2);

    // ---- constants of the fused first conv
    float* ftile = reinterpret_cast<float*>(smem + NS * PB);
    const int fcols = ft + 4;
    const int inv_fcols = 65536 / fcols + 1;            // c_in = 1: it / fcols == (it * inv_fcols) >> 16 for it < FPRE * NTH (checked at launch)
    const int fitems = FUSE1 ? a.c_in * FROWS * fcols : 0;
    const bool fprefetch = FUSE1 && fitems <= FPRE * NTH;
    constexpr int KSA = KS > 0 ? KS : 1;
    int koff[KSA][4];
    uint2 w1[KSA][2][NS];
    float sh1[2][4];
    int ksteps = 0;
    int kaddr[KSA][4];                                  // koff + the lane's column inside a 16-column block
    const int wlane = g * CM_PLANE + (lane & 15) * 16;   // lane part of the chunk-major store address
    // c_in = 1: Toeplitz form of the first conv (see the phase below): 4 output columns x 2 channel halves of stationary A fragments
    uint4 w1t[KS == 1 ? 4 : 1][2][NS];
    if constexpr (FUSE1 && KS == 1) {
        const uint4* wp = reinterpret_cast<const uint4*>(a.w1frag + (int64_t)grp * a.w1_gs) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) w1t[q][nt][p] = wp[((q * 2 + nt) * NS + p) * 64];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sh1[nt][r] = a.shift1[(int64_t)grp * 32 + g * 8 + 4 * nt + r];
        // feature columns past the staged ones (fcols .. FW) are only ever multiplied by zero weights, but must hold finite
        // values (NaN x 0 is NaN): zero them once (disjoint from the cells staging writes, so no barrier is needed in between)
        for (int i = tid; i < FROWS * FW + FSLACK; i += NTH)
            if (i >= FROWS * FW || i % FW >= fcols) ftile[i] = 0.f;
    }
    if constexpr (FUSE1 && KS != 1) {
        const int kvalid = 9 * a.c_in;
        ksteps = (kvalid + 15) >> 4;                    // K = 16 per MFMA step, <= 4 steps
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * ks + 4 * g + j;
                const int ci = k / 9, tap = k - ci * 9, kh = tap / 3, kw = tap - kh * 3;
                koff[ks][j] = k < kvalid ? (ci * FROWS + kh) * FW + kw : 0;   // padded k: zero weights, any finite value
                // c_in = 1: slot (g, j) = tap (kh = g, kw = j), see the packing; the zero-weight slots (g = 3, j = 3) read initialised
                // cells too (row 2 again; one feature column more is staged): uninitialised LDS may hold NaN, and NaN x 0 is not 0
                if (KS == 1) koff[ks][j] = min(g, 2) * FW + j;
                kaddr[ks][j] = koff[ks][j] + (lane & 15);
            }
            const uint2* wp = reinterpret_cast<const uint2*>(a.w1frag + (int64_t)grp * a.w1_gs) + lane;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) w1[ks][nt][p] = ks < ksteps ? wp[((ks * 2 + nt) * NS + p) * 64] : make_uint2(0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sh1[nt][r] = a.shift1[(int64_t)grp * 32 + g * 8 + 4 * nt + r];
    }

    uint4 pre[PREFETCH ? ITEMS : 1];
    float fpre[FPRE];
    float fown = 0.f, fref = 0.f;   // dB staging (f_clip_max): the prefetched tile's clip maximum and reference power
    int tile = blockIdx.x;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < NS; ++p) settle(wf[tap][nt][p]);
    if constexpr (FUSE1 && KS == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) settle(w1t[q][nt][p]);
    }
    if constexpr (FUSE1) {
#pragma unroll
        for (int ks = 0; ks < (KS == 1 ? 0 : KS); ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) settle(w1[ks][nt][p]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) settle(sh1[nt][r]);
    }

#define CONV_ISSUE_TILE_LOADS(TC)                                                                          \
    do {                                                                                                   \
        const char* inb = in_grp + (int64_t)(TC).b * a.T * a.F * CIN * 2;                                  \
        _Pragma("unroll") for (int n = 0; n < ITEMS; ++n) {                                                \
            const int pos = (tid >> 2) + 64 * n;                                                           \
            const int i = (pos * inv_cols) >> 16;                                                          \
            const int j = pos - i * cols;                                                                  \
            const int t = (TC).t0 - 1 + i, f = (TC).f0 - 1 + j;                                            \
            pre[n] = make_uint4(0, 0, 0, 0);                                                               \
            if (pos < npos && t >= 0 && t < a.T && f >= 0 && f < a.F)                                      \
                pre[n] = *reinterpret_cast<const uint4*>(inb + (((int64_t)t * a.F + f) * CIN + c * 8) * 2); \
        }                                                                                                  \
    } while (0)
#define CONV_ISSUE_FEAT_LOADS(TC)                                                                          \
    do {                                                                                                   \
        const float* fb = a.feats + (int64_t)(TC).b * a.f_stride_b;                                        \
        int tid_l = tid;                                                                                   \
        asm volatile("" : "+v"(tid_l)); /* the (row, column) of the FPRE cells: recomputed per tile, not kept in registers */ \
        const bool c1ch = a.c_in == 1;                                                                     \
        _Pragma("unroll") for (int n = 0; n < FPRE; ++n) {                                                 \
            const int it = tid_l + NTH * n;                                                                \
            const int fi = c1ch ? (it * inv_fcols) >> 16 : it / fcols, fj = it - fi * fcols;               \
            const int t = (TC).t0 - 2 + fi, f = (TC).f0 - 2 + fj;                                          \
            fpre[n] = f_clip_max ? -1.f : 0.f; /* power is never negative: -1 marks the zero padding */  \
            if (it < fitems && t >= 0 && t < a.T && f >= 0 && f < a.F) fpre[n] = fb[t * a.f_stride_t + f * a.f_stride_f]; \
        }                                                                                                  \
        if (f_clip_max) {                                                                                \
            fown = f_clip_max[(TC).b];                                                                   \
            fref = a.f_ref ? a.f_ref[(TC).b] : fown;                                                       \
        }                                                                                                  \
    } while (0)

    if (tile < ntiles) {
        const TileCoord tc0 = tile_coord(tile, ntf, ntt, ft, ntiles);
        if constexpr (PREFETCH) CONV_ISSUE_TILE_LOADS(tc0);
        if (fprefetch) CONV_ISSUE_FEAT_LOADS(tc0);
    }

#ifdef AMTX_CONV_TIMING
    unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        CONV_TICK(5);
#ifdef AMTX_CONV_TIMING
        prof_acc[6] += 1;
#endif
        const TileCoord tc = tile_coord(tile, ntf, ntt, ft, ntiles);
        const int t0 = tc.t0, f0 = tc.f0;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const TileCoord tn = tile_coord(has_next ? tile + (int)gridDim.x : tile, ntf, ntt, ft, ntiles);

        if constexpr (FUSE1) {
            // ---- fused first conv: features (c_in, 20 x (ft+4)) -> LDS, then Conv(c_in->32)+BN+ReLU on the matrix cores
            // (K = 9*c_in in steps of 16, im2col patches gathered from LDS as the MFMA B operand), written straight
            // into this kernel's input tile in fragment order.  a1 never touches HBM.
            // raw power values are dB-scaled here, while they are staged (amtx_of_forward_power; db_scale_apply is the function
            // spec_scale_kernel uses, so the staged values are the bits amtx_spec_scale would have written)
            DbScale dbs = {0.f, 0.f};
            if (f_clip_max) {
                if (!fprefetch) {
                    fown = f_clip_max[tc.b];
                    fref = a.f_ref ? a.f_ref[tc.b] : fown;
                }
                dbs = db_scale_make(fown, fref);
            }
            if (fprefetch) {
                // opaque copy of the thread index: the (row, column) of the FPRE cells are tile-invariant, and hoisted out of the
                // persistent loop they sit in registers next to the stationary weights and spill (a scratch reload = vmcnt(0) =
                // a wait for the previous tile's stores at every tile top)
                int tid_f = tid;
                asm volatile("" : "+v"(tid_f));
#pragma unroll
                for (int n = 0; n < FPRE; ++n) {
                    const int it = tid_f + NTH * n;
                    const int fi = a.c_in == 1 ? (it * inv_fcols) >> 16 : it / fcols, fj = it - fi * fcols;
                    float v = fpre[n];
                    if (f_clip_max) {
                        const float sv = db_scale_apply(v, dbs);   // branch-free: a handful of instructions, the select drops the padding
                        v = v < 0.f ? 0.f : sv;
                    }
                    if (it < fitems) ftile[fi * FW + fj] = v;
                }
            } else {
                const float* fb = a.feats + (int64_t)tc.b * a.f_stride_b;
                for (int it = tid; it < fitems; it += NTH) {
                    const int ci = it / (FROWS * fcols), r = it - ci * (FROWS * fcols);
                    const int fi = r / fcols, fj = r - fi * fcols;
                    const int t = t0 - 2 + fi, f = f0 - 2 + fj;
                    float v = 0.f;
                    if (t >= 0 && t < a.T && f >= 0 && f < a.F) {
                        v = fb[ci * a.f_stride_c + t * a.f_stride_t + f * a.f_stride_f];
                        if (f_clip_max) v = db_scale_apply(v, dbs);
                    }
                    ftile[(ci * FROWS + fi) * FW + fj] = v;
                }
            }
            CONV_TICK(0);
            __syncthreads();
            CONV_TICK(1);
            if (fprefetch && has_next) CONV_ISSUE_FEAT_LOADS(tn);
            if constexpr (KS == 1) {
            // ---- c_in = 1: the first conv as a TOEPLITZ product on the matrix cores.  One "unit" = 16 tile rows x 4 tile columns:
            //        D[(column q, channel), row n] = sum_{(dy, e)} Tw[(q, channel), (dy, e)] . f[n + dy, 4 xb + e],   e = 0..7, dy = 0..2
            // with Tw[(q, co), (dy, e)] = w[co, dy, e - q] for 0 <= e - q <= 2 and 0 elsewhere (host-packed, stationary in 32
            // registers).  The MFMA's B operand of lane (row n, k-group g = dy) is then 8 CONSECUTIVE floats of feature row n + g:
            // two ds_read_b128 + four conversions feed the 8 MFMAs of 64 positions x 32 channels -- the im2col form gathered four
            // scalars per lane and 16 positions (24 ds_read_b32 + 12 conversions for the same 64 positions), and that gather, not
            // the matrix work, was half of a tile's issue time.  The two halo rows (16, 17) of all column blocks form their own
            // units with lane n -> (row 16 + (n & 1), column block n >> 1).  A lane ends up with 8 consecutive channels of one
            // position per (q): one ds_write_b128 into the chunk-major input tile of conv2, as before.
            const int n16 = lane & 15;
            const int nmain = (cols + 3) >> 2;
            const int nunits = nmain + ((nmain + 7) >> 3);
            const int gg = min(g, 2);                                   // k-group 3 has zero weights: re-read group 2's row
            const int fa_main = (n16 + gg) * FW * 4;
            const int fa_halo = ((TT + (n16 & 1) + gg) * FW + 4 * (n16 >> 1)) * 4;
            const int oa_main = g * CM_PLANE + n16 * PITCH * 16;
            const int oa_halo = g * CM_PLANE + ((TT + (n16 & 1)) * PITCH + 4 * (n16 >> 1)) * 16;
            const char* fbytes = reinterpret_cast<const char*>(ftile);
            const bool interior = t0 >= 1 && t0 + TT < a.T && f0 >= 1 && f0 + ft < a.F;   // no position of the tile is padding
            // scratch lines for masked stores (one 16-byte slot per lane and plane) behind the shift table
            const int scratch_off = NS * PB + (a.c_in * FROWS * FW + FSLACK) * 4 + COUT * 4 + lane * 16;
            f32x4_t c1[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) c1[nt] = (f32x4_t){sh1[nt][0], sh1[nt][1], sh1[nt][2], sh1[nt][3]};
            __builtin_amdgcn_s_setprio(2);
            float4 fv0 = make_float4(0.f, 0.f, 0.f, 0.f), fv1 = fv0;
#define CONV1T_LOAD(U)                                                                                     \
            do {                                                                                           \
                const int uu = min((U), nunits - 1);                                                       \
                const int fa = uu < nmain ? fa_main + uu * 16 : fa_halo + (uu - nmain) * 128;              \
                fv0 = *reinterpret_cast<const float4*>(fbytes + fa);                                       \
                fv1 = *reinterpret_cast<const float4*>(fbytes + fa + 16);                                  \
            } while (0)
            CONV1T_LOAD(wave_u);
            for (int u = wave_u; u < nunits; u += NW) {
                uint4 bh, bl = make_uint4(0, 0, 0, 0);
                if (NS == 2) {
                    split_bf16x2(fv0.x, fv0.y, bh.x, bl.x);
                    split_bf16x2(fv0.z, fv0.w, bh.y, bl.y);
                    split_bf16x2(fv1.x, fv1.y, bh.z, bl.z);
                    split_bf16x2(fv1.z, fv1.w, bh.w, bl.w);
                } else {
                    bh = make_uint4(pack_bf16x2(fv0.x, fv0.y), pack_bf16x2(fv0.z, fv0.w), pack_bf16x2(fv1.x, fv1.y), pack_bf16x2(fv1.z, fv1.w));
                }
                __builtin_amdgcn_sched_barrier(0);
                CONV_TICK_FINE(8);
                CONV1T_LOAD(u + NW);
                __builtin_amdgcn_sched_barrier(0);
                CONV_TICK_FINE(9);
                f32x4_t acc1[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc1[q][nt] = mfma16(w1t[q][nt][0], bh, c1[nt]);
                        if (NS == 2) {
                            acc1[q][nt] = mfma16(w1t[q][nt][0], bl, acc1[q][nt]);
                            acc1[q][nt] = mfma16(w1t[q][nt][1], bh, acc1[q][nt]);
                        }
                    }
                CONV_TICK_FINE(10);
                const bool mainu = u < nmain;                                                    // scalar
                const int oa = mainu ? oa_main + u * 64 : oa_halo + (u - nmain) * 512;
                // wave-uniform: no position of this unit is padding or past the tile's last column -> unconditional epilogue
                const bool fast = __builtin_amdgcn_readfirstlane((int)(interior && mainu && 4 * u + 3 < cols)) != 0;
                if (fast) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (NS == 1) {
                            // round first, then ReLU on the packed pairs as a signed 16-bit max with 0 (a negative bf16 is a
                            // negative int16, -0 included)
                            typedef short s16x2 __attribute__((ext_vector_type(2)));
                            uint32_t pk[4];
#pragma unroll
                            for (int h = 0; h < 4; ++h) {
                                const uint32_t v = pack_bf16x2(acc1[q][h >> 1][2 * (h & 1)], acc1[q][h >> 1][2 * (h & 1) + 1]);
                                pk[h] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
                            }
                            *reinterpret_cast<uint4*>(smem + oa + q * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                        } else {
                            float y[8];
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) y[nt * 4 + r] = fmaxf(acc1[q][nt][r], 0.f);
                            uint4 hi, lo;
                            cvt8(y, true, hi, lo);
                            *reinterpret_cast<uint4*>(smem + oa + q * 16) = hi;
                            *reinterpret_cast<uint4*>(smem + PB + oa + q * 16) = lo;
                        }
                    }
                } else {
                    // border tiles, the halo-row units and a ragged last column block: branch-free per-lane padding (ReLU and the
                    // zero padding of the map in one v_med3 against inf / 0) and a per-lane store address (positions past the
                    // tile's last column go to a scratch line) -- a version with per-position branches cost 2000 cycles per tile
                    const int jl = mainu ? 4 * u : 4 * (8 * (u - nmain) + (n16 >> 1));           // first tile column of this lane's positions
                    const int tl = t0 - 1 + (mainu ? n16 : TT + (n16 & 1));
                    const bool row_ok = (unsigned)tl < (unsigned)a.T;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = row_ok && (unsigned)(f0 - 1 + jl + q) < (unsigned)a.F;
                        const float lim = ok ? __builtin_inff() : 0.f;
                        float y[8];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) y[nt * 4 + r] = __builtin_amdgcn_fmed3f(acc1[q][nt][r], 0.f, lim);
                        uint4 hi, lo;
                        cvt8(y, NS == 2, hi, lo);
                        const bool in_tile = jl + q < cols;
                        *reinterpret_cast<uint4*>(smem + (in_tile ? oa + q * 16 : scratch_off)) = hi;
                        if (NS == 2) *reinterpret_cast<uint4*>(smem + (in_tile ? PB + oa + q * 16 : scratch_off + 1024)) = lo;
                    }
                }
                CONV_TICK_FINE(12);
#ifdef AMTX_CONV_TIMING
                prof_acc[14] += 1;
#endif
            }
#undef CONV1T_LOAD
            } else {
            // Position groups are (tile row i, 16-column block jb): row and block are wave-uniform (scalar registers), only
            // the column inside the block is per lane, so the position arithmetic costs no vector instructions.
            // Two groups per iteration; the im2col gathers of the NEXT iteration are issued (unconditionally: padded k slots
            // read slot 0 against zero weights, columns past the tile read finite slack) before this iteration's epilogue,
            // so the LDS latency of the gathers hides behind the MFMAs + ReLU/pack of the current groups.
            const int nblk = (cols + 15) >> 4;
            const int ngroups = ROWS * nblk;
            const int inv_nblk = 65536 / nblk + 1;               // exact for gq < 1024 (ROWS * nblk <= 72)
            float pv[2][KSA][4];
#define CONV1_GATHER(GI0)                                                                                  \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                \
                const int gq = min((GI0) + NW * u, ngroups - 1);                                            \
                const int gi = (gq * inv_nblk) >> 16, gb = gq - gi * nblk;                                 \
                const float* fbase = ftile + (gi * FW + gb * 16);                                          \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                          \
                    if (KS == 1 || ks < ksteps) {                                                          \
                        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) pv[u][ks][jj] = fbase[kaddr[ks][jj]]; \
                    }                                                                                      \
            }
            // This phase is a latency chain with few instructions in flight (gather -> cvt -> 4 small MFMAs -> clamp/pack -> LDS
            // write); the other block on the CU is usually in its MFMA-bound conv phase.  Raising this wave's issue priority
            // lets every instruction it has ready go first, which shortens the chain without starving the matrix cores
            // (measured: conv1+conv2 3.16-3.33 -> 3.00 ms; raising the conv phase or its epilogue instead costs time).
            __builtin_amdgcn_s_setprio(2);
            if (wave_u < ngroups) { CONV1_GATHER(wave_u) }
            for (int gi0 = wave_u; gi0 < ngroups; gi0 += 2 * NW) {
                uint2 ph[2][KSA], pl[2][KSA];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        pl[u][ks] = make_uint2(0, 0);
                        if (KS == 1 || ks < ksteps) {
                            if (NS == 2) {
                                split_bf16x2(pv[u][ks][0], pv[u][ks][1], ph[u][ks].x, pl[u][ks].x);
                                split_bf16x2(pv[u][ks][2], pv[u][ks][3], ph[u][ks].y, pl[u][ks].y);
                            } else {
                                ph[u][ks] = make_uint2(pack_bf16x2(pv[u][ks][0], pv[u][ks][1]), pack_bf16x2(pv[u][ks][2], pv[u][ks][3]));
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                CONV_TICK(8);
                { CONV1_GATHER(gi0 + 2 * NW) }
                __builtin_amdgcn_sched_barrier(0);
                CONV_TICK(9);
                f32x4_t acc1[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc1[u][0] = (f32x4_t){sh1[0][0], sh1[0][1], sh1[0][2], sh1[0][3]};
                    acc1[u][1] = (f32x4_t){sh1[1][0], sh1[1][1], sh1[1][2], sh1[1][3]};
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (KS == 1 || ks < ksteps) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                acc1[u][nt] = mfma16k16(w1[ks][nt][0], ph[u][ks], acc1[u][nt]);
                                if (NS == 2) {
                                    acc1[u][nt] = mfma16k16(w1[ks][nt][0], pl[u][ks], acc1[u][nt]);
                                    acc1[u][nt] = mfma16k16(w1[ks][nt][1], ph[u][ks], acc1[u][nt]);
                                }
                            }
                        }
                    }
                }
                CONV_TICK(10);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int gq = gi0 + NW * u;
                    if (u == 1) CONV_TICK(11);
                    if (gq < ngroups) {
                        const int gi = (gq * inv_nblk) >> 16, gb = gq - gi * nblk;          // scalar
                        const int t = t0 - 1 + gi;
                        const int cj = gb * 16 + (lane & 15);
                        const int f = f0 - 1 + cj;
                        if (cj < cols) {
                            const int off = (gi * PITCH + gb * 16) * 16 + wlane;
                            if constexpr (NS == 1) {
                                // bf16 mode: round first, then ReLU on the packed pairs as a signed 16-bit max with 0 (a negative
                                // bf16 is a negative int16, -0 included) -- 4 + 4 instructions for 8 values instead of 8 + 4.
                                // The zero padding of the map only exists at the image border: a group whose row and 16 columns
                                // are all inside (decided on scalars) skips the mask.
                                typedef short s16x2 __attribute__((ext_vector_type(2)));
                                uint32_t pk[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const uint32_t v = pack_bf16x2(acc1[u][q >> 1][2 * (q & 1)], acc1[u][q >> 1][2 * (q & 1) + 1]);
                                    pk[q] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
                                }
                                const int fb0 = f0 - 1 + gb * 16;                          // scalar
                                const bool inside = t >= 0 && t < a.T && fb0 >= 0 && fb0 + 15 < a.F;
                                if (!inside) {
                                    const uint32_t keep = (t >= 0 && t < a.T && f >= 0 && f < a.F) ? 0xffffffffu : 0u;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) pk[q] &= keep;
                                }
                                *reinterpret_cast<uint4*>(smem + off) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                            } else {
                                // ReLU and the zero padding of the 32-channel map in one v_med3: clamp to [0, inf) inside, [0, 0] outside
                                const float lim = (t >= 0 && t < a.T && f >= 0 && f < a.F) ? __builtin_inff() : 0.f;
                                float y[8];
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) y[nt * 4 + r] = __builtin_amdgcn_fmed3f(acc1[u][nt][r], 0.f, lim);
                                uint4 hi, lo;
                                cvt8(y, true, hi, lo);
                                *reinterpret_cast<uint4*>(smem + off) = hi;
                                *reinterpret_cast<uint4*>(smem + PB + off) = lo;
                            }
                        }
                    }
                }
                CONV_TICK(12);
#ifdef AMTX_CONV_TIMING
                prof_acc[14] += 1;
#endif
            }
#undef CONV1_GATHER
            }
            __builtin_amdgcn_s_setprio(0);
            CONV_TICK(2);
        } else if constexpr (PREFETCH) {
            // ---- the tile was fetched while the previous one was computed: registers -> LDS
#pragma unroll
            for (int n = 0; n < ITEMS; ++n) {
                const int pos = (tid >> 2) + 64 * n;
                const int i = (pos * inv_cols) >> 16;
                const int j = pos - i * cols;
                if (pos < npos) *reinterpret_cast<uint4*>(smem + tile_off(i, j, c)) = pre[n];
            }
        } else {
            // ---- stage the (TT+2) x (ft+2) x 32 tile in two batches (all loads of a batch issued before the first
            // LDS store)
            const char* in = in_grp + (int64_t)tc.b * a.T * a.F * CIN * (IN_TYPE == AMTX_T_F32 ? 4 : 2);
            constexpr int BATCH = ITEMS / 2;
            // opaque copy of the thread index: keeps the compiler from hoisting the 14 tile-invariant (row, column, LDS offset)
            // triples out of the persistent loop, where they would sit in registers next to the stationary weights and spill
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint4 v0[BATCH], v1[BATCH];
                int off[BATCH];
#pragma unroll
                for (int n = 0; n < BATCH; ++n) {
                    const int pos = (tid_o >> 2) + 64 * (half * BATCH + n);
                    const int i = (pos * inv_cols) >> 16;
                    const int j = pos - i * cols;
                    const int t = t0 - 1 + i, f = f0 - 1 + j;
                    const bool ok = pos < npos && t >= 0 && t < a.T && f >= 0 && f < a.F;
                    off[n] = pos < npos ? tile_off(i, j, c) : -1;
                    v0[n] = make_uint4(0, 0, 0, 0);
                    v1[n] = make_uint4(0, 0, 0, 0);
                    if (ok) {
                        const int64_t e = ((int64_t)t * a.F + f) * CIN + c * 8;
                        if (IN_TYPE == AMTX_T_BF16) {
                            v0[n] = *reinterpret_cast<const uint4*>(in + e * 2);
                        } else if (IN_TYPE == AMTX_T_SPLIT) {
                            // the two planes as the previous layer's epilogue wrote them: nothing to convert
                            v0[n] = *reinterpret_cast<const uint4*>(in + e * 2);
                            v1[n] = *reinterpret_cast<const uint4*>(in + (a.in_split + e) * 2);
                        } else {
                            v0[n] = *reinterpret_cast<const uint4*>(in + e * 4);
                            v1[n] = *reinterpret_cast<const uint4*>(in + e * 4 + 16);
                        }
                    }
                }
#pragma unroll
                for (int n = 0; n < BATCH; ++n) {
                    if (off[n] < 0) continue;
                    uint4 hi = v0[n], lo = IN_TYPE == AMTX_T_SPLIT ? v1[n] : make_uint4(0, 0, 0, 0);
                    if (IN_TYPE == AMTX_T_F32) {
                        const float fv[8] = {__uint_as_float(v0[n].x), __uint_as_float(v0[n].y), __uint_as_float(v0[n].z), __uint_as_float(v0[n].w),
                                             __uint_as_float(v1[n].x), __uint_as_float(v1[n].y), __uint_as_float(v1[n].z), __uint_as_float(v1[n].w)};
                        cvt8(fv, NS == 2, hi, lo);
                    }
                    *reinterpret_cast<uint4*>(smem + off[n]) = hi;
                    if (NS == 2) *reinterpret_cast<uint4*>(smem + PB + off[n]) = lo;
                }
            }
        }
        if constexpr (!FUSE1) CONV_TICK(0);
        __syncthreads();
        CONV_TICK(3);
        if constexpr (PREFETCH) {
            if (has_next) CONV_ISSUE_TILE_LOADS(tn);
        }

        const int t = t0 + trow;
        char* out = reinterpret_cast<char*>(a.out) + ((int64_t)grp * a.out_gs + ((int64_t)tc.b * a.T + t) * Fo * COUT) * (OUT_TYPE == AMTX_T_F32 ? 4 : 2);
        // Fragment reads are software-pipelined by tap row: while the MFMAs of row kh run, the ds_reads of the same
        // row of this wave's NEXT column pair are already in flight (x[kh] is refilled right after its last use), so the
        // LDS latency never sits between two MFMA groups and the epilogue overlaps the next pair's reads.
        const int npairs = ft >> 1;
        int rbase[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) rbase[kh] = FUSE1 ? g * CM_PLANE + (trow + kh) * PITCH * 16 : tile_off(trow + kh, 0, g);
        constexpr int CSTEP = FUSE1 ? 16 : 64;              // bytes between neighbouring columns of one chunk
        uint4 x[3][4][NS];
#define CONV_LOAD_ROW(KH, JP)                                                                              \
        _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                 \
            const int off = rbase[KH] + (2 * (JP) + cc) * CSTEP;                                             \
            x[KH][cc][0] = *reinterpret_cast<const uint4*>(smem + off);                                    \
            if (NS == 2) x[KH][cc][1] = *reinterpret_cast<const uint4*>(smem + PB + off);         \
        }
        if (wave_u < npairs) {
            CONV_LOAD_ROW(0, wave_u)
            CONV_LOAD_ROW(1, wave_u)
            CONV_LOAD_ROW(2, wave_u)
        }
        for (int jp = wave_u; jp < npairs; jp += NW) {
            const int jn = min(jp + NW, npairs - 1);     // past the end: re-read a valid pair, never used
            f32x4_t acc[2][NT];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[e][nt] = SH_REGS ? shr[nt] : (WIDE_ST ? shl0[(nt >> 1) * 8 + g * 2 + (nt & 1)] : shl[nt]);

#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int kw = cc - e;
                        if (kw < 0 || kw > 2) continue;
                        const int tap = kh * 3 + kw;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[e][nt] = mfma16(wf[tap][nt][0], x[kh][cc][0], acc[e][nt]);
                            if (NS == 2) {
                                acc[e][nt] = mfma16(wf[tap][nt][0], x[kh][cc][1], acc[e][nt]);
                                acc[e][nt] = mfma16(wf[tap][nt][1], x[kh][cc][0], acc[e][nt]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kh == 0) { CONV_LOAD_ROW(0, jn) } else if (kh == 1) { CONV_LOAD_ROW(1, jn) } else { CONV_LOAD_ROW(2, jn) }
                __builtin_amdgcn_sched_barrier(0);
            }

            // ---- + shift, ReLU, MaxPool(1,2) over the (f, f+1) pair, channels-last store
            const int fo = (f0 >> 1) + jp;
            if (t < a.T && fo < Fo) {
                float v[NT * 4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[nt * 4 + r] = fmaxf(fmaxf(acc[0][nt][r], acc[1][nt][r]), 0.f);
                    }
                if (OUT_TYPE == AMTX_T_BF16) {
                    uint4* dst = reinterpret_cast<uint4*>(out + ((int64_t)fo * COUT + (WIDE_ST ? g * 8 : g * 4 * NT)) * 2);
#pragma unroll
                    for (int q = 0; q < NT / 2; ++q)
                        dst[WIDE_ST ? 4 * q : q] = make_uint4(pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                                            pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7]));
                } else if (OUT_TYPE == AMTX_T_SPLIT) {
                    // the next layer's operand planes, split here once (split_bf16x2: what that layer's staging would have computed)
                    uint4* dh = reinterpret_cast<uint4*>(out + ((int64_t)fo * COUT + g * 4 * NT) * 2);
                    uint4* dl = reinterpret_cast<uint4*>(out + ((int64_t)fo * COUT + g * 4 * NT + a.out_split) * 2);
#pragma unroll
                    for (int q = 0; q < NT / 2; ++q) {
                        const float y[8] = {v[8 * q], v[8 * q + 1], v[8 * q + 2], v[8 * q + 3], v[8 * q + 4], v[8 * q + 5], v[8 * q + 6], v[8 * q + 7]};
                        uint4 hi, lo;
                        cvt8(y, true, hi, lo);
                        dh[q] = hi;
                        dl[q] = lo;
                    }
                } else {
                    float4* dst = reinterpret_cast<float4*>(out + ((int64_t)fo * COUT + g * 4 * NT) * 4);
#pragma unroll
                    for (int q = 0; q < NT; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
            }
        }
#undef CONV_LOAD_ROW
        CONV_TICK(4);
        __syncthreads();   // every wave is done with this LDS tile before the next one is staged
    }
#ifdef AMTX_CONV_TIMING
    if (tid == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_conv_prof[(FUSE1 ? 0 : 8) + i], prof_acc[i]);
        if (FUSE1) for (int i = 8; i < 15; ++i) atomicAdd(&g_conv_prof[8 + i], prof_acc[i]);
        atomicAdd(&g_conv_prof[(FUSE1 ? 0 : 8) + 7], 1ull);
    }
#endif
#undef CONV_ISSUE_TILE_LOADS
#undef CONV_ISSUE_FEAT_LOADS
}

template <int NT, int NS, int IN_TYPE, int OUT_TYPE, bool FUSE1, int KS = 0>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    const int fe = (a.F + 1) & ~1;
    const int ntf = (fe + FT_MAX - 1) / FT_MAX;
    const int ft = 2 * (((fe >> 1) + ntf - 1) / ntf);
    const int ntt = (a.T + TT - 1) / TT;
    const int64_t nblocks = (int64_t)ntf * ntt * a.B;
    AMTX_REQUIRE(nblocks < (1ll << 31), "conv3x3: grid too large");
    const size_t lds = (size_t)NS * (FUSE1 ? CM_BYTES : PLANE_BYTES) + (FUSE1 ? ((size_t)a.c_in * FROWS * fw_pitch(KS) + FSLACK) * sizeof(float) : 0) + (size_t)NT * 16 * sizeof(float) +
                       (FUSE1 && KS == 1 ? (size_t)NS * 1024 : 0);   // + the Toeplitz first conv's scratch lines
    auto kern = conv3x3_kernel<NT, NS, IN_TYPE, OUT_TYPE, FUSE1, KS>;
    AMTX_GRANT_LDS(kern, lds);   // the size grows with c_in
    const int cols = ft + 2;
    const int inv_cols = 65536 / cols + 1;
    for (int pos = 0; pos < 1024; ++pos)
        if (((pos * inv_cols) >> 16) != pos / cols) {
            amtx_set_error("conv3x3: internal: reciprocal division inexact for cols=%d", cols);
            return AMTX_ERR_ARG;
        }
    if (FUSE1 && a.c_in == 1) {
        const int fcols = ft + 4, inv_fcols = 65536 / fcols + 1;
        for (int it = 0; it < FPRE * 256; ++it)
            if (((it * inv_fcols) >> 16) != it / fcols) {
                amtx_set_error("conv3x3: internal: reciprocal division inexact for fcols=%d", fcols);
                return AMTX_ERR_ARG;
            }
    }
    // persistent grid: two resident blocks per CU in total
    int64_t gx = nblocks;
    const int64_t per_group = std::max<int64_t>(1, 512 / std::max(1, a.groups));
    if (gx > per_group) gx = per_group;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)a.groups), dim3(conv_threads(FUSE1, KS, NS)), lds, stream, a, ft, ntf, ntt, inv_cols, (int)nblocks);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int NT, int NS>
int dispatch_types(const ConvArgs& a, hipStream_t s) {
    if (a.feats) {
        if (NT != 2) {
            amtx_set_error("conv3x3: the fused first conv feeds a 32 -> 32 layer only");
            return AMTX_ERR_UNSUPPORTED;
        }
        if (a.c_in * 9 <= 16) {
            if (a.out_type == AMTX_T_BF16) return launch_conv<2, NS, AMTX_T_BF16, AMTX_T_BF16, true, 1>(a, s);
            if (a.out_type == AMTX_T_SPLIT) {
                if constexpr (NS == 2) {
                    AMTX_REQUIRE(a.out_split > 0 && a.out_split % 8 == 0, "conv3x3: two-plane output needs a plane stride");
                    static const bool no_convx = getenv("AMTX_NO_CONVX") != nullptr;      // A/B switch, as below
                    if (!no_convx && a.c_in == 1) return amtx_launch_convx12(a, s);
                    return launch_conv<2, NS, AMTX_T_F32, AMTX_T_SPLIT, true, 1>(a, s);
                }
                amtx_set_error("conv3x3: two-plane maps exist in the two-plane mode only");
                return AMTX_ERR_UNSUPPORTED;
            }
            return launch_conv<2, NS, AMTX_T_F32, AMTX_T_F32, true, 1>(a, s);
        }
        if (a.out_type == AMTX_T_BF16) return launch_conv<2, NS, AMTX_T_BF16, AMTX_T_BF16, true, 4>(a, s);
        return launch_conv<2, NS, AMTX_T_F32, AMTX_T_F32, true, 4>(a, s);
    }
    if (a.in_type == AMTX_T_BF16 && a.out_type == AMTX_T_BF16) {
        // (an LDS-DMA double-buffered variant of this layer was measured in rounds 1 - 2: 4 % slower on conv3, which is bandwidth-bound,
        // and never the default; it and its AMTX_CONV_DMA switch are gone)
        return launch_conv<NT, NS, AMTX_T_BF16, AMTX_T_BF16, false>(a, s);
    }
    if (a.in_type == AMTX_T_F32 && a.out_type == AMTX_T_F32) return launch_conv<NT, NS, AMTX_T_F32, AMTX_T_F32, false>(a, s);
    if (a.in_type == AMTX_T_SPLIT && a.out_type == AMTX_T_SPLIT) {
        if constexpr (NS == 2) {
            AMTX_REQUIRE(a.in_split > 0 && a.in_split % 8 == 0 && a.out_split > 0 && a.out_split % 8 == 0, "conv3x3: two-plane maps need plane strides");
            // A/B switch: AMTX_NO_CONVX=1 keeps this file's register-staged kernel on the two-plane maps
            static const bool no_convx = getenv("AMTX_NO_CONVX") != nullptr;
            if (!no_convx) return amtx_launch_convx3(a, s);
            return launch_conv<NT, NS, AMTX_T_SPLIT, AMTX_T_SPLIT, false>(a, s);
        }
    }
    amtx_set_error("conv3x3: in/out element types must match (bf16/bf16 or f32/f32)");
    return AMTX_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------ first conv: direct, fp32 math
template <int OUT_TYPE>
__global__ __launch_bounds__(256) void conv1_kernel(Conv1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                 // [c_in*9][c_out]
    float* sl = wl + a.c_in * 9 * a.c_out;                      // [c_out]
    const int grp = blockIdx.y;
    const float* w = a.w + (int64_t)grp * a.w_gs;
    const int nw = a.c_out * a.c_in * 9;
    for (int i = threadIdx.x; i < nw; i += 256) {
        const int co = i / (a.c_in * 9), rest = i % (a.c_in * 9);
        wl[rest * a.c_out + co] = w[i];
    }
    for (int i = threadIdx.x; i < a.c_out; i += 256) sl[i] = a.shift[(int64_t)grp * a.shift_gs + i];
    __syncthreads();

    const int cgroups = a.c_out >> 3;                           // 8 output channels per thread
    const int64_t total = (int64_t)a.B * a.T * a.F * cgroups;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int cg = (int)(idx % cgroups);
        int64_t pos = idx / cgroups;
        const int f = (int)(pos % a.F); pos /= a.F;
        const int t = (int)(pos % a.T);
        const int b = (int)(pos / a.T);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        const float* inb = a.in + (int64_t)b * a.stride_b;
        for (int ci = 0; ci < a.c_in; ++ci) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int tt = t + kh - 1;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ff = f + kw - 1;
                    float x = 0.f;
                    if (tt >= 0 && tt < a.T && ff >= 0 && ff < a.F) x = inb[ci * a.stride_c + tt * a.stride_t + ff * a.stride_f];
                    const float4* wp = reinterpret_cast<const float4*>(wl + ((ci * 3 + kh) * 3 + kw) * a.c_out + cg * 8);
                    const float4 w0 = wp[0], w1 = wp[1];
                    acc[0] = fmaf(x, w0.x, acc[0]); acc[1] = fmaf(x, w0.y, acc[1]);
                    acc[2] = fmaf(x, w0.z, acc[2]); acc[3] = fmaf(x, w0.w, acc[3]);
                    acc[4] = fmaf(x, w1.x, acc[4]); acc[5] = fmaf(x, w1.y, acc[5]);
                    acc[6] = fmaf(x, w1.z, acc[6]); acc[7] = fmaf(x, w1.w, acc[7]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = a.relu ? fmaxf(acc[k] + sl[cg * 8 + k], 0.f) : acc[k] + sl[cg * 8 + k];
        const int64_t o = (((int64_t)b * a.T + t) * a.F + f) * a.c_out + cg * 8;
        if (OUT_TYPE == AMTX_T_BF16) {
            bf16_t* dst = reinterpret_cast<bf16_t*>(a.out) + (int64_t)grp * a.out_gs + o;
            *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                                                        pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
        } else {
            float* dst = reinterpret_cast<float*>(a.out) + (int64_t)grp * a.out_gs + o;
            reinterpret_cast<float4*>(dst)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            reinterpret_cast<float4*>(dst)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

}  // namespace

size_t amtx_conv3x3_wfrag_elems(int c_out, int planes) { return (size_t)9 * (c_out / 16) * planes * 64 * 8; }

void amtx_conv3x3_pack_host(const float* w, const float* scale, int c_out, int planes, bf16_t* out) {
    const int NT = c_out / 16;
    for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15;
                const int co = (row >> 2) * (4 * NT) + 4 * nt + (row & 3);
                for (int j = 0; j < 8; ++j) {
                    const int ci = (l >> 4) * 8 + j;
                    const float v = w[((size_t)co * CIN + ci) * 9 + tap] * (scale ? scale[co] : 1.0f);
                    const bf16_t hi = f32_to_bf16_rn(v);
                    const size_t base = ((size_t)(tap * NT + nt) * planes) * 64 * 8 + (size_t)l * 8 + j;
                    out[base] = hi;
                    if (planes == 2) out[base + 64 * 8] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                }
            }
}

size_t amtx_conv1_wfrag_elems(int c_in, int planes) {
    if (c_in == 1) return (size_t)4 * 2 * planes * 64 * 8;    // Toeplitz fragments: 4 output columns x 2 channel halves, 32-deep
    return (size_t)((9 * c_in + 15) / 16) * 2 * planes * 64 * 4;
}

void amtx_conv1_pack_host(const float* w, const float* scale, int c_in, int planes, bf16_t* out) {
    if (c_in == 1) {
        // Toeplitz A fragments of the fused first conv (conv3x3_kernel, KS == 1): fragment (q, nt), lane l = (row, k-group g):
        // row -> channel co = 8 (row >> 2) + 4 nt + (row & 3) (a lane of the D tile then holds 8 consecutive channels over nt = 0, 1),
        // k = 8 g + e -> tap (dy = g, kw = e - q) of output column q within a 4-column unit; everything else is zero
        for (int q = 0; q < 4; ++q)
            for (int nt = 0; nt < 2; ++nt)
                for (int l = 0; l < 64; ++l) {
                    const int row = l & 15, g = l >> 4;
                    const int co = (row >> 2) * 8 + 4 * nt + (row & 3);
                    for (int e = 0; e < 8; ++e) {
                        const int kw = e - q;
                        const float v = (g < 3 && kw >= 0 && kw <= 2) ? w[(size_t)co * 9 + g * 3 + kw] * (scale ? scale[co] : 1.0f) : 0.0f;
                        const bf16_t hi = f32_to_bf16_rn(v);
                        const size_t base = ((size_t)(q * 2 + nt) * planes) * 64 * 8 + (size_t)l * 8 + e;
                        out[base] = hi;
                        if (planes == 2) out[base + 64 * 8] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                    }
                }
        return;
    }
    const int kvalid = 9 * c_in, ksteps = (kvalid + 15) / 16;
    for (int ks = 0; ks < ksteps; ++ks)
        for (int nt = 0; nt < 2; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15;
                const int co = (row >> 2) * 8 + 4 * nt + (row & 3);
                for (int j = 0; j < 4; ++j) {
                    // c_in = 1: K slot (g, j) of the single 16-deep step is tap (kh = g, kw = j), g, j < 3 (the other 7 slots are
                    // zero), so a lane's four im2col values are one LDS row segment: one address, immediate offsets.
                    // Otherwise k runs over (ci, kh, kw) in the weight tensor's own order.
                    int k = 16 * ks + 4 * (l >> 4) + j;
                    if (c_in == 1) k = ((l >> 4) < 3 && j < 3) ? 3 * (l >> 4) + j : kvalid;
                    const float v = k < kvalid ? w[(size_t)co * kvalid + k] * (scale ? scale[co] : 1.0f) : 0.0f;
                    const bf16_t hi = f32_to_bf16_rn(v);
                    const size_t base = ((size_t)(ks * 2 + nt) * planes) * 64 * 4 + (size_t)l * 4 + j;
                    out[base] = hi;
                    if (planes == 2) out[base + 64 * 4] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                }
            }
}

int amtx_launch_conv3x3(const ConvArgs& a, hipStream_t stream) {
    AMTX_REQUIRE((a.in || a.feats) && a.wfrag && a.shift && a.out, "conv3x3: null pointer");
    if (a.feats) {
        AMTX_REQUIRE(a.w1frag && a.shift1 && a.c_in > 0 && 9 * a.c_in <= 64, "conv3x3: fused first conv needs w1frag/shift1 and 9*c_in <= 64");
    }
    AMTX_REQUIRE(a.B > 0 && a.T > 0 && a.F >= 2 && a.groups > 0, "conv3x3: bad sizes");
    AMTX_REQUIRE(a.planes == 1 || a.planes == 2, "conv3x3: planes must be 1 or 2");
    const int key = (a.c_out / 16) * 10 + a.planes;
    switch (key) {
        case 21: return dispatch_types<2, 1>(a, stream);
        case 22: return dispatch_types<2, 2>(a, stream);
        case 41: return dispatch_types<4, 1>(a, stream);
        case 42: return dispatch_types<4, 2>(a, stream);
    }
    amtx_set_error("conv3x3: unsupported c_out=%d (supported: 32 and 64, with C_in = 32)", a.c_out);
    return AMTX_ERR_UNSUPPORTED;
}

int amtx_launch_conv1(const Conv1Args& a, hipStream_t stream) {
    AMTX_REQUIRE(a.in && a.w && a.shift && a.out, "conv1: null pointer");
    AMTX_REQUIRE(a.B > 0 && a.T > 0 && a.F > 0 && a.c_in > 0 && a.c_out % 8 == 0 && a.groups > 0, "conv1: bad sizes");
    const size_t lds = ((size_t)a.c_in * 9 * a.c_out + a.c_out) * sizeof(float);
    AMTX_REQUIRE(lds <= 64 * 1024, "conv1: weights do not fit LDS");
    const int64_t total = (int64_t)a.B * a.T * a.F * (a.c_out / 8);
    int64_t nblocks = (total + 255) / 256;
    if (nblocks > 256 * 32) nblocks = 256 * 32;
    if (a.out_type == AMTX_T_BF16)
        hipLaunchKernelGGL(conv1_kernel<AMTX_T_BF16>, dim3((unsigned)nblocks, (unsigned)a.groups), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL(conv1_kernel<AMTX_T_F32>, dim3((unsigned)nblocks, (unsigned)a.groups), dim3(256), lds, stream, a);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

#ifdef AMTX_CONV_TIMING
extern "C" int amtxdbg_conv_prof(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_conv_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
