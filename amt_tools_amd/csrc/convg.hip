// Conv2d 3x3 pad 1 + folded BatchNorm2d + ReLU + MaxPool(1,2) for the channel counts conv.hip's register-stationary
// kernel does not cover (amt_tools/models/onsetsframes.py:375-416 at model_complexity != 2; OnsetsFrames2's default
// model_complexity = 3 has 48 -> 48 and 48 -> 96 channel layers, onsetsframes.py:199-233).
//
// Same implicit GEMM as conv.hip (swapped product D' = W . X^T on the matrix cores, one 16-frame column of positions per
// MFMA N-tile, pooling pair and channels-last store lane-local), but C_in is any multiple of 16 and the folded weights do not
// fit the register file next to two waves per SIMD, so they are staged in LDS in fragment order, one C_out chunk of NTC
// 16-channel tiles at a time:
//   * K per tap = C_in = n32 x 32 (v_mfma_f32_16x16x32_bf16) + (C_in % 32 ? one more 32-deep step whose upper half (lane groups
//     2, 3) multiplies zero weights : none).  The legacy 16-deep v_mfma_f32_16x16x16_bf16 is NOT used for that tail: mixed into
//     the 32-deep accumulation chains it gave wrong, run-to-run varying values in the later result registers on MI355X (first
//     version of this file; hipcc 7.2 schedules it with too few wait states), and it costs the same issue cycles anyway,
//   * the input tile (18 x (FT+2) positions) is held chunk-major (16-byte chunk c of a position at c * CPLANE + position * 16,
//     CPLANE = 48 (mod 256) bytes, row pitch FT + 2 = 2 (mod 16) positions): the fragment reads of 16 consecutive rows are
//     conflict-free, the staging stores of consecutive chunks spread over the banks,
//   * one wave = four neighbouring output columns (two pooled outputs) x all NTC tiles of the chunk: a weight fragment read
//     from LDS feeds 4 MFMAs, an input fragment NTC x (up to) 3,
//   * x3 precision: hi/lo planes of both operands, 3 MFMAs per product (as everywhere else in this library).
// Algorithmic HBM bytes per output position: C_in in + C_out / 2 out, element size of the mode.

#include "amtx_f16_names.h"
#include "amtx_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

#ifndef CONVG_FCL_GP
#define CONVG_FCL_GP 5
#endif
#ifndef CONVG_FCL_GP2
#define CONVG_FCL_GP2 6              // ... in the two-plane mode (six groups per wave: all at once, 315 registers; 2: 4.67, 3: 4.65, 6: 4.54 ms per 512 HCQT clips)
#endif
#ifndef CONVG_FCL_MINW
#define CONVG_FCL_MINW 1            // waves per SIMD the 32-channel tap-major variant is compiled for: 4 = two 512-thread blocks per CU (<= 128 VGPRs)
#endif
constexpr int GTT = 16;             // frames per tile
constexpr int GROWS = GTT + 2;

typedef __attribute__((ext_vector_type(8))) __bf16 g_bf16x8;
__device__ __forceinline__ f32x4_t gm32(uint4 a, uint4 b, f32x4_t c) {
    return amtx_mfma_16x16x32(a, b, c);
}

// compile-time loop: the body sees its index as a constant expression, so register arrays indexed through it stay in registers
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#ifdef AMTX_CONV_TIMING
// Debug build only (AMTX_EXTRA_FLAGS=-DAMTX_CONV_TIMING): cycles wave 0 of every block spends per phase of the persistent loop, summed
// over blocks: [0] tile store / feature store + barrier, [1] fused first conv, [2] weight store + barrier, [3] MFMA loop, [4] epilogue
// stores, [5] trailing barrier, [6] chunk-tiles.  Read with amtxdbg_convg_prof().
__device__ unsigned long long g_convg_prof[8];
#define CG_TICK(SLOT)                                                      \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        cg_acc[SLOT] += now_ - cg_t;                                       \
        cg_t = now_;                                                       \
    } while (0)
#else
#define CG_TICK(SLOT) do {} while (0)
#endif

constexpr int g_cplane(int ft) {
    // bytes of one chunk plane: ROWS x PC positions x 16 B, rounded up to 48 (mod 256)
    const int pc = ft + 2;
    const int raw = GROWS * pc * 16;
    return (raw + 255 - 48) / 256 * 256 + 48;
}

// per (tap, tile) weight bytes of one plane in LDS / in the packed global buffer
// Weight fragments (1024 bytes per plane each) of one 16-channel output tile: 9 taps x n32 full 32-deep steps, plus, when C_in has a
// 16-channel tail, 5 PAIRED tail steps: the two 16-channel tails of two taps share one 32-deep MFMA (lane groups 0, 1 = first tap,
// 2, 3 = second tap): (kh 0, kh 1) for each kw, then (2,0)+(2,1), then (2,2) alone (upper half zero).  14 MFMAs per output column
// and channel tile at C_in = 48 instead of 18 with one half-empty tail step per tap.
constexpr int g_wfrags_per_tile(int ci16) { return 9 * (ci16 / 2) + 5 * (ci16 % 2); }

// KS1 > 0: the first convolution (Conv2d(c_in -> C_in of this layer) + BN + ReLU, models/onsetsframes.py:375-384) is computed
// inside this kernel, from the fp32 features, straight into the LDS input tile (the C_in-channel map never exists in HBM):
// K = 9 c_in taps padded to KS1 32-deep steps, im2col gathered per lane from a small feature tile in LDS, D' = W1 . P^T per
// 16 tile positions, epilogue shift + ReLU + zero outside the map + bf16 (hi/lo) -> 8-byte LDS stores.
// FCL (with KS1 = 3, one plane): the tap-major form of the fused first conv (amtx_conv1g_tapk): features staged channels-last in bf16.
// CMAX: the most input channels this instantiation stages (sizes the per-thread item slots: 6 -> 9 loads per tile instead of 12; 1 -> 2
// instead of a rolled ten-slot loop with an early exit); 0 = the variant's limit (8 tap-major, 7 otherwise).
// F16IN (tap-major only): the features arrive as that very tile format -- ConvArgs.feats16, [B][T][F][8] 16-bit channels-last, what
// amtx_cqt_forward16 writes -- and a position is ONE 16-byte load and ONE 16-byte LDS store instead of c_in strided 4-byte loads, conversions
// and 2-byte stores.
// STRIP (with F16IN, or with a one-channel fp32 first conv): the tile holds THREE strips of 8 output columns -- the same columns f0 .. f0 + 7 of three consecutive 16-frame blocks -- side
// by side instead of 32 neighbouring columns: strip s sits at tile columns 10 s .. 10 s + 9 (its own halo columns included; 12 s .. 12 s + 11 of
// the feature tile) and belongs to waves 2 s, 2 s + 1.  What the launcher gives the last <= 8 columns of a map whose width is not a multiple of
// 32 (the HCQT shape: 72 = 2 x 32 + 8): as a 32-column tile those columns kept two waves of eight busy, a quarter of the kernel's time.
template <int CI16, int NTC, int NS, int FT, int IN_TYPE, int OUT_TYPE, int KS1, bool FCL = false, int CMAX = 0, bool F16IN = false, bool STRIP = false>
__global__ __launch_bounds__(16 * FT, (FCL && CI16 == 2) ? CONVG_FCL_MINW : 1) void conv3x3_gen_kernel(ConvArgs a, int ntf, int ntt, int nchunks, int ntiles, int w_all, int sh_off, int f_base) {
    static_assert(!FCL || KS1 == 3, "tap-major first conv: three 32-deep steps");
    static_assert(!F16IN || FCL, "16-bit channels-last features: the tap-major first conv only");
    static_assert(!STRIP || (FT == 32 && KS1 > 0 && NS == 1 && (F16IN || (!FCL && CMAX == 1))), "strip tiles: 32-column tiles with a fused first conv from 16-bit features or from one fp32 channel");
    constexpr int SPW = 8, NSTRIP = 3;           // output columns of a strip; strips per tile (3 x (8 + 2) <= FT + 2, 3 x (8 + 4) = FT + 4)
    constexpr int NTH = 16 * FT;                 // one wave per 4 output columns
    constexpr int CIN = 16 * CI16;
    constexpr int NCH = CIN / 8;                 // 16-byte chunks per position
    constexpr int N32 = CI16 / 2, N16 = CI16 % 2;
    constexpr int PC = FT + 2;                   // positions per tile row: FT + 2 = 2 (mod 16) with CPLANE = 48 (mod 256) bytes keeps
                                                 // every ds_read_b128 lane group on 16 distinct 16-byte slots (tools/lds_swizzle_search.py model)
    constexpr int COLS = FT + 2;
    constexpr int CPLANE = g_cplane(FT);
    constexpr int XPLANE = NCH * CPLANE;         // one precision plane of the input tile
    constexpr int NMAIN = 9 * NTC * N32;         // full 32-deep (tap, tile, step) fragments of one chunk
    constexpr int NTAIL = 5 * NTC * N16;         // paired tail fragments
    constexpr int WCHUNK = (NMAIN + NTAIL) * NS * 1024;   // all planes of one C_out chunk
    constexpr int IES = IN_TYPE == AMTX_T_BF16 ? 2 : 4;
    constexpr int NITEMS = GROWS * COLS * NCH;
    constexpr int NIT = (NITEMS + NTH - 1) / NTH;
    constexpr int NWIT = (WCHUNK / 16 + NTH - 1) / NTH;
    constexpr int NRAW = IN_TYPE == AMTX_T_BF16 ? 1 : 2;
    constexpr bool FUSE1 = KS1 > 0;
    constexpr int FROWS1 = GROWS + 2, FP1 = FT + 4 + 1;      // feature tile: rows t0-2 .. t0+17, columns f0-2 .. f0+FT+1 (+1 pad)
    constexpr int NPOS = GROWS * COLS;                        // positions of the input tile
    constexpr int NNT1 = (NPOS + 15) / 16;                    // 16-position groups of the fused first conv
    constexpr int NW = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PIPE = FCL && CI16 == 2 && NS == 1;      // two-tile software pipeline (below): two input tiles and two feature tiles in LDS
    char* xs = smem;
    char* ws = smem + (PIPE ? 2 : 1) * NS * XPLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int grp = blockIdx.y;
    const int F = a.F, T = a.T, F2 = F >> 1;
    const int64_t out_ts = a.out_ts ? a.out_ts : (int64_t)F2 * a.c_out;
    // FUSE1: first-conv weight fragments [tile][ks][plane][lane] + shift [CIN] (copied once: a global load per tile would put an
    // L2 round trip in front of every tile's first MFMA), then [c_in][FROWS1][FP1] fp32 features
    constexpr int W1BYTES = FUSE1 ? CI16 * KS1 * NS * 1024 + CIN * 4 : 0;
    char* w1s = ws + (w_all == 1 ? nchunks : (w_all == 2 ? 2 : 1)) * WCHUNK;
    float* fs = reinterpret_cast<float*>(w1s + W1BYTES);
    // FCL: the feature tile as bf16 [FROWS1][FT + 4][8 channel slots]: 16 bytes per position = one k-group of a tap
    char* fs16 = w1s + W1BYTES;
    constexpr int FW = FT + 4;
    constexpr int FSPL = FROWS1 * FW * 16;       // bytes of one plane of the channels-last feature tile (two-plane mode: hi plane, then lo plane)
    char* const fscratch = w1s + W1BYTES + (PIPE ? 2 : 1) * NS * FSPL;             // NS x 128 bytes behind the feature tile(s): slots without an item
    const int c_in1 = a.c_in;
    const int nfeat = c_in1 * FROWS1 * (FT + 4);
    if (FCL) {   // channel slots c_in .. 7 meet zero weights, but NaN x 0 is NaN: zero the tile once (the staging only writes real channels)
        for (int it = tid; it < NS * FROWS1 * FW; it += NTH) reinterpret_cast<uint4*>(fs16)[it] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    if (FUSE1) {
        const uint4* w1g = reinterpret_cast<const uint4*>(a.w1frag + (int64_t)grp * a.w1_gs);
        for (int it = tid; it < CI16 * KS1 * NS * 64; it += NTH) reinterpret_cast<uint4*>(w1s)[it] = w1g[it];
        if (tid < CIN) reinterpret_cast<float*>(w1s + CI16 * KS1 * NS * 1024)[tid] = a.shift1[(int64_t)grp * CIN + tid];
    }
    // the folded BatchNorm shift of this layer, [c_out] fp32, in LDS for the launch: a global load per chunk would queue behind the next
    // tile's prefetch loads (vmcnt retires in order) and put that round trip in front of every tile's first MFMA
    const float* shs = reinterpret_cast<const float*>(smem + sh_off);
    for (int i = tid; i < a.c_out; i += NTH) reinterpret_cast<float*>(smem + sh_off)[i] = a.shift[(int64_t)grp * a.shift_gs + i];
    const char* in_g = reinterpret_cast<const char*>(a.in) + (int64_t)grp * a.in_gs * IES;
    const uint4* wsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.wfrag) + (int64_t)grp * a.w_gs * 2);

    // tile id -> (clip, first frame, first column); consecutive ids of one XCD are spatial neighbours (halo rows / columns come
    // from that XCD's L2 instead of HBM)
    auto coord = [&](int tile, int& b, int& t0, int& f0) {
        tile = (int)xcd_remap((unsigned)tile, (unsigned)ntiles);
        const int tf = tile % ntf; tile /= ntf;
        const int tt = tile % ntt; tile /= ntt;
        b = tile; t0 = tt * GTT * (STRIP ? NSTRIP : 1); f0 = f_base + tf * FT;
    };
    // All loads of a thread are issued before the first LDS store (clamped addresses + a select instead of a branch: a branchy
    // loop costs one full memory round trip per item).  Rows t0-1 .. t0+16, columns f0-1 .. f0+FT, zero outside the map.
    // Item k of a thread = 16-byte chunk c of tile position (i, j), fixed for the launch: (byte offset of the chunk inside a position) |
    // i << 10 | j << 16, decoded once.  Per tile an item then costs a dozen 32-bit vector instructions (clamp, validity, one offset from the
    // clip's scalar base: a clip's map is < 4 GiB, the launcher checks) instead of ~30 with 64-bit addresses -- staging runs with the
    // matrix pipe idle, two waves per SIMD deep.
    unsigned xmeta[FUSE1 ? 1 : NIT];
    if (!FUSE1) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            int it = tid + k * NTH;
            it = it < NITEMS ? it : NITEMS - 1;
            const int c = it % NCH, pos = it / NCH;
            xmeta[k] = (unsigned)(c * 8 * IES) | (unsigned)(pos / COLS) << 10 | (unsigned)(pos % COLS) << 16;
        }
    }
    auto load_x = [&](int tile, uint4 (&raw)[NIT][NRAW], unsigned& okmask) {
        int b, t0, f0;
        coord(tile, b, t0, f0);
        const char* in = in_g + (int64_t)b * T * F * CIN * IES;
        okmask = 0;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const unsigned m = xmeta[FUSE1 ? 0 : k];
            const int t = t0 - 1 + (int)((m >> 10) & 63), f = f0 - 1 + (int)(m >> 16);
            const int tc = min(max(t, 0), T - 1), fc = min(max(f, 0), F - 1);
            if (t == tc && f == fc) okmask |= 1u << k;
            const unsigned off = (unsigned)(tc * F + fc) * (unsigned)(CIN * IES) + (m & 1023u);
            const char* p = in + off;
            raw[k][0] = *reinterpret_cast<const uint4*>(p);
            if (NRAW == 2) raw[k][NRAW - 1] = reinterpret_cast<const uint4*>(p)[1];
        }
    };
    auto store_x = [&](const uint4 (&raw)[NIT][NRAW], unsigned okmask) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = tid + k * NTH;
            const int c = it % NCH, pos = it / NCH;
            uint4 hi = raw[k][0], lo = make_uint4(0, 0, 0, 0);
            if (NRAW == 2) {
                const float4 v0 = __builtin_bit_cast(float4, raw[k][0]), v1 = __builtin_bit_cast(float4, raw[k][NRAW - 1]);
                if (NS == 2) {
                    split_bf16x2(v0.x, v0.y, hi.x, lo.x); split_bf16x2(v0.z, v0.w, hi.y, lo.y);
                    split_bf16x2(v1.x, v1.y, hi.z, lo.z); split_bf16x2(v1.z, v1.w, hi.w, lo.w);
                } else {
                    hi = make_uint4(pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w));
                }
            }
            if (!((okmask >> k) & 1)) { hi = make_uint4(0, 0, 0, 0); lo = hi; }
            if (it < NITEMS) {
                const int off = c * CPLANE + pos * 16;        // PC == COLS: tile position (i, j) is slot i * COLS + j
                *reinterpret_cast<uint4*>(xs + off) = hi;
                if (NS == 2) *reinterpret_cast<uint4*>(xs + XPLANE + off) = lo;
            }
        }
    };
    auto load_w = [&](int ch, uint4 (&wreg)[NWIT]) {
#pragma unroll
        for (int k = 0; k < NWIT; ++k) {
            const int it = tid + k * NTH;
            wreg[k] = wsrc[(int64_t)ch * (WCHUNK / 16) + (it < WCHUNK / 16 ? it : 0)];
        }
    };
    auto store_w = [&](const uint4 (&wreg)[NWIT]) {
#pragma unroll
        for (int k = 0; k < NWIT; ++k) {
            const int it = tid + k * NTH;
            if (it < WCHUNK / 16) reinterpret_cast<uint4*>(ws)[it] = wreg[k];
        }
    };

    // first of this wave's four output columns: jb in the tile (strip tiles: strip wave / 2 at tile column 10 (wave / 2), its second half 4
    // further), jbo relative to the map column f0
    const int jb = STRIP ? (SPW + 2) * (wave >> 1) + 4 * (wave & 1) : 4 * wave;
    const int jbo = STRIP ? 4 * (wave & 1) : 4 * wave;
    const int xrow = r16 * PC * 16;                            // byte offset of this lane's row (kh = 0) in a chunk plane
    const int x32 = g * CPLANE;                                // + ks * 4 * CPLANE: chunk 4 ks + g
    const int x16 = (4 * N32 + (g & 1)) * CPLANE;              // tail steps: lane groups (0, 1) and (2, 3) each read the two tail chunks

    // Persistent blocks (one per CU: the tile and one weight chunk fill most of the LDS): the NEXT tile's input and the NEXT
    // weight chunk travel HBM/L2 -> registers while the current ones are on the matrix cores, so neither the block start-up nor
    // a memory round trip is paid per tile.  With a single C_out chunk the weights stay in LDS for the whole launch.
    // ---- fused first conv helpers
    constexpr int CMX = CMAX > 0 ? CMAX : (FCL ? 8 : 7);      // channels the item slots are sized for
    constexpr bool EXACT = FCL || CMAX == 1;                  // every slot loads unconditionally (no early exit from the item loops)
    constexpr int NF1 = !FUSE1 ? 1 : F16IN ? (FROWS1 * (FT + 4) + NTH - 1) / NTH      // F16IN: an item is a position (16 bytes)
                                           : (CMX * FROWS1 * (FT + 4) + NTH - 1) / NTH;   // feature values per thread (c_in <= 7; tap-major: <= 8)
    using fraw_t = typename std::conditional<F16IN, uint4, float>::type;
    // A thread's feature items are the same tile-relative (channel, row, column) for every tile: decoded once (the div / mod chains
    // per item and tile were a sixth of the kernel), bit 31 = item exists.  The zeroing of values outside the map waits for
    // store_f: a select right behind the load would wait for the load here.
    unsigned fdesc[NF1];
#pragma unroll
    for (int k = 0; k < NF1; ++k) {
        const int it = tid + k * NTH;
        if constexpr (F16IN) {
            // bits 0-7 / 8-15: column / row of the position in the feature TILE; bits 16-21 / 22-29: its column / row in the MAP relative to
            // the tile's (f0 - 2, t0 - 2) -- the same, except in a strip tile
            const bool has = it < FROWS1 * (FT + 4);
            const int itc = has ? it : 0;
            const int i = itc / (FT + 4), j = itc % (FT + 4);
            const int mi = STRIP ? i + GTT * (j / (SPW + 4)) : i, mj = STRIP ? j % (SPW + 4) : j;
            fdesc[k] = (has ? 0x80000000u : 0u) | ((unsigned)mi << 22) | ((unsigned)mj << 16) | ((unsigned)i << 8) | (unsigned)j;
            continue;
        }
        const int itc = it < nfeat ? it : 0;
        if constexpr (STRIP) {       // one channel: bits 16-21 / 22-29 = the item's column / row in the MAP relative to (f0 - 2, t0 - 2), as for F16IN
            int i, j;
            if (a.f_stride_t < a.f_stride_f) { i = itc % FROWS1; j = (itc / FROWS1) % (FT + 4); }
            else { j = itc % (FT + 4); i = (itc / (FT + 4)) % FROWS1; }
            fdesc[k] = (it < nfeat ? 0x80000000u : 0u) | ((unsigned)(i + GTT * (j / (SPW + 4))) << 22) | ((unsigned)(j % (SPW + 4)) << 16) | ((unsigned)i << 8) | (unsigned)j;
            continue;
        }
        int ci, i, j;
        if (a.f_stride_t < a.f_stride_f) {        // frames contiguous (a (B,C,F,T) tensor): consecutive lanes = consecutive frames of one column
            i = itc % FROWS1; j = (itc / FROWS1) % (FT + 4); ci = itc / (FROWS1 * (FT + 4));
        } else {                                  // columns contiguous (the model layout the log-mel kernels write)
            j = itc % (FT + 4); i = (itc / (FT + 4)) % FROWS1; ci = itc / (FROWS1 * (FT + 4));
        }
        fdesc[k] = (it < nfeat ? 0x80000000u : 0u) | ((unsigned)ci << 16) | ((unsigned)i << 8) | (unsigned)j;
    }
    const int nfk = (nfeat + NTH - 1) / NTH;                   // items per thread that exist for this c_in (uniform)
    // FCL: element offset of every item inside a tile's feature window, tile-independent (32 bits: a clip's view spans < 2^31 elements, checked
    // at launch).  Per tile and item that leaves two adds, two unsigned compares and a select in front of the load; the 64-bit clamp /
    // multiply chains of the other variants are 27 vector instructions per loaded value -- the largest block of the HCQT model's conv2 kernel.
    // (Only in the tap-major variants: the others sit at 256 registers and ten more spill hundreds.)
    int foff[FCL ? NF1 : 1];
    if constexpr (FCL) {
#pragma unroll
        for (int k = 0; k < NF1; ++k) {
            const int j = fdesc[k] & 0xff, i = (fdesc[k] >> 8) & 0xff, ci = (fdesc[k] >> 16) & 0xff;
            foff[k] = F16IN ? (int)((fdesc[k] >> 22) & 0xff) * F + (int)((fdesc[k] >> 16) & 0x3f) : ci * (int)a.f_stride_c + i * (int)a.f_stride_t + j * (int)a.f_stride_f;
        }
    }
    auto load_f = [&](int tile, fraw_t (&fr)[NF1], unsigned& okmask) {
        int b, t0, f0;
        coord(tile, b, t0, f0);
        okmask = 0;
        if constexpr (F16IN) {
            const uint4* fb = reinterpret_cast<const uint4*>(a.feats16) + (int64_t)b * T * F;      // one uint4 per position
            const int wb = (t0 - 2) * F + (f0 - 2);
#pragma unroll
            for (int k = 0; k < NF1; ++k) {
                const int j = (fdesc[k] >> 16) & 0x3f, i = (fdesc[k] >> 22) & 0xff;
                const bool ok = (int)fdesc[k] < 0 && (unsigned)(t0 - 2 + i) < (unsigned)T && (unsigned)(f0 - 2 + j) < (unsigned)F;
                fr[k] = fb[(unsigned)(ok ? wb + foff[k] : 0)];
                okmask |= ok ? (1u << k) : 0u;
            }
            return;
        }
        const float* fb = a.feats + (int64_t)b * a.f_stride_b;
        if constexpr (FCL && !F16IN) {
            const int wb = (t0 - 2) * (int)a.f_stride_t + (f0 - 2) * (int)a.f_stride_f;   // window origin; negative at the map's edges: only used when valid
            // straight-line: every item slot loads (a slot without an item, or a padding cell, re-reads the clip's first value) -- a
            // guarded load is waited for at the join of its branch, one memory round trip per item (the rolled, branchy form of this loop
            // was 43 % of a tile)
#pragma unroll
            for (int k = 0; k < NF1; ++k) {
                const int j = fdesc[k] & 0xff, i = (fdesc[k] >> 8) & 0xff;
                const bool ok = (int)fdesc[k] < 0 && (unsigned)(t0 - 2 + i) < (unsigned)T && (unsigned)(f0 - 2 + j) < (unsigned)F;
                fr[k] = fb[(unsigned)(ok ? wb + foff[k] : 0)];
                okmask |= ok ? (1u << k) : 0u;
            }
        } else if constexpr (!F16IN) {
#pragma unroll
            for (int k = 0; k < NF1; ++k) {
                if (!EXACT && k >= nfk) break;
                const int j = STRIP ? (fdesc[k] >> 16) & 0x3f : fdesc[k] & 0xff, i = STRIP ? (fdesc[k] >> 22) & 0xff : (fdesc[k] >> 8) & 0xff;
                const int ci = STRIP ? 0 : (fdesc[k] >> 16) & 0xff;
                const int t = t0 - 2 + i, f = f0 - 2 + j;
                const bool ok = (fdesc[k] >> 31) && t >= 0 && t < T && f >= 0 && f < F;
                const int tc = min(max(t, 0), T - 1), fc = min(max(f, 0), F - 1);
                fr[k] = fb[(int64_t)ci * a.f_stride_c + (int64_t)tc * a.f_stride_t + (int64_t)fc * a.f_stride_f];
                if (ok) okmask |= 1u << k;
            }
        }
    };
    auto store_f = [&](const fraw_t (&fr)[NF1], unsigned okmask) {
        if constexpr (F16IN) {
#pragma unroll
            for (int k = 0; k < NF1; ++k) {
                const int j = fdesc[k] & 0xff, i = (fdesc[k] >> 8) & 0xff;
                const uint4 v = ((okmask >> k) & 1) ? fr[k] : make_uint4(0, 0, 0, 0);
                if ((int)fdesc[k] < 0) *reinterpret_cast<uint4*>(fs16 + (i * FW + j) * 16) = v;
            }
            return;
        }
        if constexpr (FCL && !F16IN) {
            // straight-line as well: a slot without an item writes its (zero) value to a scratch line behind the feature tiles
#pragma unroll
            for (int k = 0; k < NF1; ++k) {
                const int j = fdesc[k] & 0xff, ci = (fdesc[k] >> 16) & 0xff, i = (fdesc[k] >> 8) & 0xff;
                const float v = ((okmask >> k) & 1) ? fr[k] : 0.f;
                char* dst = (int)fdesc[k] < 0 ? fs16 + ((i * FW + j) * 8 + ci) * 2 : fscratch + (tid & 63) * 2;
                if constexpr (NS == 2) {                      // both planes of the value, split ONCE (the K-major form split it at every gather)
                    uint32_t vh, vl;
                    split_bf16x2(v, 0.f, vh, vl);
                    *reinterpret_cast<uint16_t*>(dst) = (uint16_t)vh;
                    *reinterpret_cast<uint16_t*>((int)fdesc[k] < 0 ? dst + FSPL : dst + 128) = (uint16_t)vl;
                } else
                *reinterpret_cast<uint16_t*>(dst) = (uint16_t)pack_bf16x2(v, 0.f);
            }
            return;
        }
        if constexpr (!F16IN) {
#pragma unroll
            for (int k = 0; k < NF1; ++k) {
                if (!EXACT && k >= nfk) break;
                if (fdesc[k] >> 31) {
                    const int j = fdesc[k] & 0xff, ci = STRIP ? 0 : (fdesc[k] >> 16) & 0xff, i = (fdesc[k] >> 8) & 0xff;
                    fs[(ci * FROWS1 + i) * FP1 + j] = ((okmask >> k) & 1) ? fr[k] : 0.f;
                }
            }
        }
    };
    // per-lane im2col offsets (floats) of the K slots this lane feeds: k = 32 ks + 8 g + jj -> (ci, kh, kw) in the weight
    // tensor's own order; slots past 9 c_in read a valid address (their weights are zero)
    int koff[KS1 > 0 ? KS1 : 1][8];
    if (FUSE1 && !FCL) {
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                int k = 32 * ks + 8 * g + jj;
                k = k < 9 * c_in1 ? k : 0;
                const int ci = k / 9, tap = k % 9;
                koff[ks][jj] = (ci * FROWS1 + tap / 3) * FP1 + tap % 3;
            }
    }
    // FCL: what a lane's fragment reads need is the same for every tile -- the byte offset of its k-group's tap inside the feature tile, and per
    // 16-position group its position's offset and (row, column): decoded once (divisions by 3 and by COLS per group and tile otherwise)
    constexpr int NGRP1 = (NNT1 + NW - 1) / NW;
    int tapoff[FCL ? 3 : 1], gsrc[NGRP1], gij[NGRP1];
    if constexpr (FCL) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int tap = 4 * ks + g, tp = tap < 9 ? tap : 0;
            tapoff[ks] = ((tp / 3) * FW + tp % 3) * 16;
        }
    }
    if constexpr (FUSE1) {
#pragma unroll
        for (int q = 0; q < NGRP1; ++q) {
            const int pos = (wave + q * NW) * 16 + r16, posc = pos < NPOS ? pos : NPOS - 1;
            const int i = posc / COLS, j = posc % COLS;
            if constexpr (STRIP) {
                // tile column j = strip j / 10, column j % 10 of it; its taps start at feature column 12 s + (j % 10); map row i + 16 s.
                // Columns 30 .. 33 belong to no strip: a row outside every map
                const int st = j / (SPW + 2), lc = j % (SPW + 2);
                gsrc[q] = FCL ? (i * FW + (SPW + 4) * st + lc) * 16 : i * FP1 + (SPW + 4) * st + lc;
                gij[q] = (st < NSTRIP ? i + GTT * st : 0x7fffff) << 8 | lc;
                continue;
            }
            gsrc[q] = FCL ? (i * FW + j) * 16 : i * FP1 + j;     // feature tile offset of the position: bytes (bf16 channels-last) / floats
            gij[q] = i << 8 | j;
        }
    }
    auto first_conv = [&](int t0, int f0) {
        const uint4* w1 = reinterpret_cast<const uint4*>(w1s);                                   // [tile][ks][plane][lane]
        const float* sh1 = reinterpret_cast<const float*>(w1s + CI16 * KS1 * NS * 1024);
        uint4 wf[CI16][KS1 > 0 ? KS1 : 1][NS];
#pragma unroll
        for (int nt = 0; nt < CI16; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int p = 0; p < NS; ++p) wf[nt][ks][p] = w1[((nt * KS1 + ks) * NS + p) * 64 + lane];
        // Straight-line code, two 16-position groups at a time: their gathers, then their MFMAs, then their epilogues, so that no
        // instruction waits on the one right before it (a loop of gather -> MFMA -> epilogue per channel tile ran at ~1500 cycles per
        // group, a third of the conv2 kernel).
        float4 s4[CI16];
#pragma unroll
        for (int nt = 0; nt < CI16; ++nt) s4[nt] = *reinterpret_cast<const float4*>(sh1 + 16 * nt + 4 * g);
        // groups in flight at once: their LDS reads, then their MFMA chains (independent of each other), then their epilogues.  The tap-major
        // form reads three 16-byte chunks per group instead of sixteen scalars: all of a wave's groups fit the registers at once, and five
        // serial read -> MFMA -> convert -> store chains (~1000 cycles each) become one pass
        constexpr int NIT1 = (NNT1 + NW - 1) / NW, GP = FCL ? (CI16 <= 2 ? (NS == 1 ? CONVG_FCL_GP : CONVG_FCL_GP2) : 2) : ((KS1 * NS > 2) ? 1 : 2);   // (48 / 64 mid channels: two at a time, or the accumulators spill)
        static_for<0, (NIT1 + GP - 1) / GP>([&](auto gc) {
            constexpr int it0 = decltype(gc)::value * GP;
            constexpr int cnt = it0 + GP <= NIT1 ? GP : NIT1 - it0;
            int pos[GP];
            bool inside[GP];
            uint4 ph[GP][KS1 > 0 ? KS1 : 1], pl[GP][KS1 > 0 ? KS1 : 1];
            static_for<0, cnt>([&](auto ic) {
                constexpr int i_ = decltype(ic)::value;
                pos[i_] = (wave + (it0 + i_) * NW) * 16 + r16;
                if constexpr (FCL) {
                    // k-group g of step ks = tap 4 ks + g (taps 9 .. 11: zero weights, any valid address): the 8 channel slots of the
                    // position (i + kh, j + kw) of the feature tile, already bf16
                    const int i = gij[it0 + i_] >> 8, j = gij[it0 + i_] & 0xff;
                    inside[i_] = (unsigned)(t0 - 1 + i) < (unsigned)T && (unsigned)(f0 - 1 + j) < (unsigned)F;
                    static_for<0, KS1>([&](auto kc) {
                        constexpr int ks = decltype(kc)::value;
                        ph[i_][ks] = *reinterpret_cast<const uint4*>(fs16 + gsrc[it0 + i_] + tapoff[ks]);
                        if constexpr (NS == 2) pl[i_][ks] = *reinterpret_cast<const uint4*>(fs16 + FSPL + gsrc[it0 + i_] + tapoff[ks]);
                    });
                    return;
                }
                const int i = gij[it0 + i_] >> 8, j = gij[it0 + i_] & 0xff;
                const float* fp = fs + gsrc[it0 + i_];
                inside[i_] = (unsigned)(t0 - 1 + i) < (unsigned)T && (unsigned)(f0 - 1 + j) < (unsigned)F;
                static_for<0, KS1>([&](auto kc) {
                    constexpr int ks = decltype(kc)::value;
                    float v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v[jj] = fp[koff[ks][jj]];
                    if (NS == 2) {
                        split_bf16x2(v[0], v[1], ph[i_][ks].x, pl[i_][ks].x); split_bf16x2(v[2], v[3], ph[i_][ks].y, pl[i_][ks].y);
                        split_bf16x2(v[4], v[5], ph[i_][ks].z, pl[i_][ks].z); split_bf16x2(v[6], v[7], ph[i_][ks].w, pl[i_][ks].w);
                    } else {
                        ph[i_][ks] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                    }
                });
            });
            f32x4_t d[GP][CI16];
            static_for<0, cnt>([&](auto ic) {
                constexpr int i_ = decltype(ic)::value;
                static_for<0, CI16>([&](auto nc) {
                    constexpr int nt = decltype(nc)::value;
                    f32x4_t dd = (f32x4_t){s4[nt].x, s4[nt].y, s4[nt].z, s4[nt].w};
                    static_for<0, KS1>([&](auto kc) {
                        constexpr int ks = decltype(kc)::value;
                        dd = gm32(wf[nt][ks][0], ph[i_][ks], dd);
                        if constexpr (NS == 2) {
                            dd = gm32(wf[nt][ks][0], pl[i_][ks], dd);
                            dd = gm32(wf[nt][ks][NS - 1], ph[i_][ks], dd);
                        }
                    });
                    d[i_][nt] = dd;
                });
            });
            static_for<0, cnt>([&](auto ic) {
                constexpr int i_ = decltype(ic)::value;
                static_for<0, CI16>([&](auto nc) {
                    constexpr int nt = decltype(nc)::value;
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = inside[i_] ? fmaxf(d[i_][nt][r], 0.f) : 0.f;
                    uint2 hi, lo = make_uint2(0, 0);
                    if (NS == 2) { split_bf16x2(o[0], o[1], hi.x, lo.x); split_bf16x2(o[2], o[3], hi.y, lo.y); }
                    else hi = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                    if (pos[i_] < NPOS) {
                        // channels 16 nt + 4 g .. + 4 = half (g & 1) of 16-byte chunk 2 nt + (g >> 1)
                        const int off = (2 * nt + (g >> 1)) * CPLANE + pos[i_] * 16 + (g & 1) * 8;
                        *reinterpret_cast<uint2*>(xs + off) = hi;
                        if (NS == 2) *reinterpret_cast<uint2*>(xs + XPLANE + off) = lo;
                    }
                });
            });
            __builtin_amdgcn_sched_barrier(0);                 // one pair at a time: hoisting every pair's gathers to the top spills
        });
    };

#ifdef AMTX_CONV_TIMING
    unsigned long long cg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cg_t = __builtin_readcyclecounter();
#endif
    // One C_out chunk of one tile: the MFMA loop over the tile `xs` points to and the chunk's weights, then ReLU + MaxPool + stores.
    auto mma_chunk = [&](int b, int t0, int f0, int ch, const char* wsc) {
        // a wave whose four columns all lie past the last pooled column has nothing to store: no matrix work either (F = 229: 28 of the
        // 256 tile columns, 114: 14 of 128, the HCQT shape's 72: 24 of 96; the waves that do work are not faster for it -- each runs at
        // the pace of its own instruction stream -- so this saves energy, not time)
        if (STRIP ? (wave >= 2 * NSTRIP || t0 + GTT * (wave >> 1) >= T) : f0 + jbo >= (F & ~1)) return;
        const int t_out = t0 + r16 + (STRIP ? GTT * (wave >> 1) : 0);
        // accumulators start at the folded BatchNorm shift of the lane's channels: chunk channel g * 4 NTC + 4 nt + r
        f32x4_t acc[4][NTC];
        {
            const float* sh = shs + ch * 16 * NTC + g * 4 * NTC;
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                const float4 s = *reinterpret_cast<const float4*>(sh + 4 * nt);
#pragma unroll
                for (int col = 0; col < 4; ++col) acc[col][nt] = (f32x4_t){s.x, s.y, s.z, s.w};
            }
        }

        // One "item" = one weight fragment (16 B per lane) x the wave's four columns = 4 MFMAs (12 in the two-plane mode).
        // Items run row-major over five "rows" of six input-column fragments: tap rows kh = 0, 1, 2 (full 32-deep steps), then
        // row 3 = the channel tails of tap rows 0 | 1 side by side, row 4 = the tails of tap row 2 at columns c | c + 1.
        // Weight fragments are read from LDS WD items ahead of their use (ring of WD + 1 register slots); the six column
        // fragments of a row are read at the row boundary (double-buffering them as well costs 48 more VGPRs and spills).
        constexpr int IPM = 3 * NTC * N32;                   // items of a main row: (kw, tile, step)
        constexpr int NITM = 3 * IPM + (3 * NTC + 2 * NTC) * N16;
        constexpr int WD = 2;
        uint4 xa[6][N32 > 0 ? N32 : 1][NS];
        uint4 wq[WD + 1][NS];
        // item v -> row, kw (column shift of the X fragment), tile, step, index of its weight fragment in the chunk
        auto load_xrow = [&](auto rowc) {
            constexpr int row = decltype(rowc)::value;
            static_for<0, 6>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                static_for<0, NS>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    if constexpr (row < 3) {
                        static_for<0, N32>([&](auto sc) {
                            constexpr int st = decltype(sc)::value;
                            xa[c][st][pl] = *reinterpret_cast<const uint4*>(xs + pl * XPLANE + xrow + (row * PC + jb + c) * 16 + x32 + st * 4 * CPLANE);
                        });
                    } else if constexpr (row == 3) {           // lane groups 0, 1: tap row 0; 2, 3: tap row 1
                        xa[c][0][pl] = *reinterpret_cast<const uint4*>(xs + pl * XPLANE + xrow + ((g >> 1) * PC + jb + c) * 16 + x16);
                    } else {                                   // tap row 2: lane groups 0, 1: column c; 2, 3: column c + 1
                        // (columns 4, 5 only meet the lone (2,2) tail, whose upper weights are zero: stay inside the tile)
                        xa[c][0][pl] = *reinterpret_cast<const uint4*>(xs + pl * XPLANE + xrow + (2 * PC + jb + c + (c < 4 ? (g >> 1) : 0)) * 16 + x16);
                    }
                });
            });
        };
        auto load_witem = [&](auto vc) {
            constexpr int v = decltype(vc)::value;
            constexpr int widx = v < 3 * IPM ? v                                      // main: ((kh * 3 + kw) * NTC + nt) * N32 + st == v
                                             : NMAIN + (v - 3 * IPM);                 // tails: A (kw, nt), B (nt), C (nt) in item order
            static_for<0, NS>([&](auto pc) {
                constexpr int pl = decltype(pc)::value;
                wq[v % (WD + 1)][pl] = *reinterpret_cast<const uint4*>(wsc + (widx * NS + pl) * 1024 + lane * 16);
            });
        };
        load_xrow(std::integral_constant<int, 0>{});
        static_for<0, WD>([&](auto vc) { load_witem(vc); });
        static_for<0, NITM>([&](auto vc) {
            constexpr int v = decltype(vc)::value;
            constexpr bool is_main = v < 3 * IPM;
            constexpr int t = v - 3 * IPM;                                            // tail item index
            constexpr int row = is_main ? v / IPM : (t < 3 * NTC ? 3 : 4);
            constexpr int kw = is_main ? (v % IPM) / (NTC * N32) : (t < 3 * NTC ? t / NTC : (t < 4 * NTC ? 0 : 2));
            constexpr int nt = is_main ? ((v % IPM) / N32) % NTC : t % NTC;
            constexpr int st = is_main ? v % N32 : 0;
            constexpr bool row_start = is_main ? (v % IPM == 0) : (t == 0 || t == 3 * NTC);
            if constexpr (row_start && v > 0) load_xrow(std::integral_constant<int, row>{});
            if constexpr (v + WD < NITM) load_witem(std::integral_constant<int, v + WD>{});
            static_for<0, 4>([&](auto colc) {
                constexpr int col = decltype(colc)::value;
                acc[col][nt] = gm32(wq[v % (WD + 1)][0], xa[col + kw][st][0], acc[col][nt]);
                if constexpr (NS == 2) {
                    acc[col][nt] = gm32(wq[v % (WD + 1)][0], xa[col + kw][st][NS - 1], acc[col][nt]);
                    acc[col][nt] = gm32(wq[v % (WD + 1)][NS - 1], xa[col + kw][st][0], acc[col][nt]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });

        CG_TICK(3);
        // ---- ReLU + MaxPool(1,2) over the column pair, channels-last store of the lane's 4 NTC channels
        if (t_out < T) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int fo = (f0 + jbo + 2 * pr) >> 1;
                if (fo >= F2) continue;
                const int64_t o = (int64_t)grp * a.out_gs + ((int64_t)b * T + t_out) * out_ts + (int64_t)fo * a.c_out + ch * 16 * NTC + g * 4 * NTC;
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaxf(acc[2 * pr][nt][r], acc[2 * pr + 1][nt][r]), 0.f);
                    if (OUT_TYPE == AMTX_T_SPLIT) {             // the next layer's two 16-bit planes (x3: conv3 and everything behind it take AMTX_T_SPLIT maps)
                        uint2 hi, lo;
                        split_bf16x2(v[0], v[1], hi.x, lo.x);
                        split_bf16x2(v[2], v[3], hi.y, lo.y);
                        bf16_t* dsth = reinterpret_cast<bf16_t*>(a.out) + o + 4 * nt;
                        *reinterpret_cast<uint2*>(dsth) = hi;
                        *reinterpret_cast<uint2*>(dsth + a.out_split) = lo;
                    } else if (OUT_TYPE == AMTX_T_BF16)
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out) + o + 4 * nt) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    else
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o + 4 * nt) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };

    if constexpr (PIPE) {
        // ---- The 32-channel tap-major variant (BASELINE config 3: HCQT, 6 x 72 bins) as a two-tile software pipeline, ONE barrier per tile:
        // between two barriers a wave runs the first conv of tile k + 1 (vector / LDS work: ~40 % of a tile when it was a phase of its own
        // with the matrix pipe idle) AND the matrix loop of tile k; waves 0-3 take them in one order, waves 4-7 in the other, so that each
        // SIMD's two waves are in different kinds of work.  Two input tiles and two feature tiles in LDS (126 KB), one C_out chunk whose
        // weights stay resident.  Requires nchunks == 1 (the launcher checks).
        char* const fs_base = fs16;
        auto xs2 = [&](int i) { return smem + i * (NS * XPLANE); };
        auto fsb = [&](int i) { return fs_base + i * (FROWS1 * FW * 16); };
        fraw_t fraw[NF1];
        unsigned xok = 0;
        const int stride = (int)gridDim.x;
        int tile = blockIdx.x;
        // prologue: weights, features of tile 0 -> fs[0], first conv of tile 0 -> xs[0], features of tile 1 -> fs[1]
        for (int it = tid; it < WCHUNK / 16; it += NTH) reinterpret_cast<uint4*>(ws)[it] = wsrc[it];
        for (int it = tid; it < FROWS1 * FW; it += NTH) reinterpret_cast<uint4*>(fsb(1))[it] = make_uint4(0, 0, 0, 0);
        load_f(tile, fraw, xok);
        fs16 = fsb(0);
        store_f(fraw, xok);
        if (tile + stride < ntiles) load_f(tile + stride, fraw, xok);
        __syncthreads();
        {
            int b, t0, f0;
            coord(tile, b, t0, f0);
            xs = xs2(0);
            first_conv(t0, f0);
        }
        if (tile + stride < ntiles) { fs16 = fsb(1); store_f(fraw, xok); }
        if (tile + 2 * stride < ntiles) load_f(tile + 2 * stride, fraw, xok);
        __syncthreads();
        int cur = 0;
        int bn = 0, t0n = 0, f0n = 0;
        coord(tile, bn, t0n, f0n);
        for (; tile < ntiles; tile += stride, cur ^= 1) {
            const int b = bn, t0 = t0n, f0 = f0n;                  // decoded one iteration ago
            const int next = tile + stride;
            const bool has_next = next < ntiles;
            if (has_next) coord(next, bn, t0n, f0n);
            auto fc_next = [&]() {
                if (has_next) { xs = xs2(cur ^ 1); fs16 = fsb(cur ^ 1); first_conv(t0n, f0n); }
            };
            auto mma_cur = [&]() { xs = xs2(cur); mma_chunk(b, t0, f0, 0, ws); };
            CG_TICK(7);
            if (wave < NW / 2) { fc_next(); CG_TICK(1); mma_cur(); } else { mma_cur(); fc_next(); }
            // features of tile k + 2 (in registers since the last iteration) -> the feature tile the first conv of tile k read an iteration ago
            if (tile + 2 * stride < ntiles) { fs16 = fsb(cur); store_f(fraw, xok); }
            if (tile + 3 * stride < ntiles) load_f(tile + 3 * stride, fraw, xok);
            CG_TICK(0);
            __syncthreads();
            CG_TICK(5);
#ifdef AMTX_CONV_TIMING
            cg_acc[6] += 1;
#endif
        }
#ifdef AMTX_CONV_TIMING
        if (tid == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_convg_prof[i], cg_acc[i]);
#endif
        return;
    }

    uint4 xraw[FUSE1 ? 1 : NIT][NRAW], wreg[NWIT];
    fraw_t fraw[NF1];
    unsigned xok = 0;
    // w_all == 3: ONE C_out chunk per block, resident for the launch: blocks i and i + 8 k (the same XCD) walk the same tiles with different
    // chunks (the grid is a multiple of 8 nchunks).  Every tile is staged nchunks times, but no weight ever moves again: what the two-plane
    // modes run when their chunks do not all fit (their matrix phase is three times as long for the same staging).
    const bool csplit = w_all == 3;
    const int bq = (int)blockIdx.x >> 3;
    const int ch_first = csplit ? bq % nchunks : 0, ch_end = csplit ? ch_first + 1 : nchunks;
    const int tstride = csplit ? (int)gridDim.x / nchunks : (int)gridDim.x;
    int tile = csplit ? (bq / nchunks) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    if (FUSE1) load_f(tile, fraw, xok);
    else load_x(tile, reinterpret_cast<uint4 (&)[NIT][NRAW]>(xraw), xok);
    bool w_resident = false;
    // w_all == 2: the chunks do not all fit, but two do (64 -> 64 / 128 and 80 -> 80 / 160 channels in the one-plane modes): two weight
    // buffers filled by LDS-DMA, the next chunk in flight under the current one's matrix loop, one barrier per chunk.  (The register
    // staging of w_all == 0 -- what the two-plane modes still use -- ends up in scratch memory and needs two barriers per chunk.)
    const bool wdma = w_all == 2;
    const unsigned ws_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)ws);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_w = [&](int ch, int buf) {                        // this wave's 1-KiB fragments of chunk ch -> weight buffer buf
        const char* src = reinterpret_cast<const char*>(wsrc) + (int64_t)ch * WCHUNK + lane * 16;
        for (int f = wave_u; f < WCHUNK / 1024; f += NW) glds16(src + f * 1024, ws_lds + buf * WCHUNK + f * 1024);
    };
    int wcur = 0;
    if (wdma) {
        dma_w(0, 0);
        w_resident = true;
    } else if (csplit) {
        for (int it = tid; it < WCHUNK / 16; it += NTH) reinterpret_cast<uint4*>(ws)[it] = wsrc[(int64_t)ch_first * (WCHUNK / 16) + it];
        w_resident = true;
    } else if (w_all) {                                        // every C_out chunk fits next to the tile: weights stay in LDS for the launch
        for (int it = tid; it < nchunks * (WCHUNK / 16); it += NTH) reinterpret_cast<uint4*>(ws)[it] = wsrc[it];
        w_resident = true;
    } else {
        load_w(0, wreg);
    }

    for (; tile < ntiles; tile += tstride) {
        int b, t0, f0;
        coord(tile, b, t0, f0);
        CG_TICK(7);
        const int next = tile + tstride;
        const bool has_next = next < ntiles;
        if (FUSE1) {
            store_f(fraw, xok);
            __syncthreads();                                   // feature tile visible
            CG_TICK(0);
            if (has_next) load_f(next, fraw, xok);
            first_conv(t0, f0);
            CG_TICK(1);
        } else {
            store_x(reinterpret_cast<const uint4 (&)[NIT][NRAW]>(xraw), xok);
            CG_TICK(0);
        }

        for (int ch = ch_first; ch < ch_end; ++ch) {
            if (wdma) wait_vm<0>();                            // this wave's pieces of the chunk have landed (issued a whole chunk ago)
            else if (!w_resident) store_w(wreg);
            if (w_all != 1 || ch == 0) __syncthreads();        // tile and weight chunk visible
            CG_TICK(2);
            if (!FUSE1 && ch == ch_first && has_next) load_x(next, reinterpret_cast<uint4 (&)[NIT][NRAW]>(xraw), xok);  // in flight during the MFMA phase
            if (wdma) {                                        // the buffer the previous chunk read is free: every wave is past the barrier
                if (ch + 1 < nchunks) dma_w(ch + 1, wcur ^ 1);
                else if (has_next) dma_w(0, wcur ^ 1);
            } else if (!w_all) {
                if (nchunks > 1) {
                    if (ch + 1 < nchunks) load_w(ch + 1, wreg);
                    else if (has_next) load_w(0, wreg);
                } else {
                    w_resident = true;
                }
            }

            mma_chunk(b, t0, f0, ch, ws + (w_all == 1 ? ch : wcur) * WCHUNK);
            CG_TICK(4);
            if (wdma) wcur ^= 1;
            if (!w_all || ch + 1 == ch_end) __syncthreads();   // every wave is done reading this chunk's weights / (last chunk) the tile
            CG_TICK(5);
#ifdef AMTX_CONV_TIMING
            cg_acc[6] += 1;
#endif
        }
    }
#ifdef AMTX_CONV_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_convg_prof[i], cg_acc[i]);
#endif
}

#ifdef AMTX_CONV_TIMING
extern "C" int amtxdbg_convg_prof(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_convg_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_convg_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// f_base / ntf_only: the launch covers the column tiles [f_base, f_base + ntf_only FT) of the map only (0: all of it).  STRIP: ONE column of
// strip tiles (three 16-frame blocks x 8 columns each) at f_base.
template <int CI16, int NTC, int NS, int FT, int IN_TYPE, int OUT_TYPE, int KS1 = 0, bool FCL = false, int CMAX = 0, bool F16IN = false, bool STRIP = false>
int launch_gen(const ConvArgs& a, hipStream_t stream, int f_base = 0, int ntf_only = 0) {
    const int fe = a.F & ~1;                                  // columns that reach a pooled output
    const int ntf = STRIP ? 1 : ntf_only > 0 ? ntf_only : (fe + FT - 1) / FT;
    const int ntt = STRIP ? ((a.T + GTT - 1) / GTT + 2) / 3 : (a.T + GTT - 1) / GTT;
    const int64_t ntiles = (int64_t)ntf * ntt * a.B;
    AMTX_REQUIRE(ntiles < (1ll << 31), "conv3x3: grid too large");
    if (KS1 == 0) AMTX_REQUIRE((int64_t)a.T * a.F * 16 * CI16 * (IN_TYPE == AMTX_T_BF16 ? 2 : 4) < (1ll << 32),
                               "conv3x3 (general): a clip's input map must be smaller than 4 GiB");
    const int nchunks = a.c_out / (16 * NTC);
    const size_t lds_x = (size_t)NS * (16 * CI16 / 8) * g_cplane(FT), wchunk = (size_t)NTC * NS * g_wfrags_per_tile(CI16) * 1024;
    const size_t lds_f = KS1 > 0 ? (FCL ? (size_t)NS * ((GROWS + 2) * (FT + 4) * 16 + 128) : (size_t)a.c_in * (GROWS + 2) * (FT + 5) * sizeof(float)) +
                                   (size_t)CI16 * KS1 * NS * 1024 + 16 * CI16 * 4 : 0;
    constexpr bool PIPE = FCL && CI16 == 2 && NS == 1;        // two input tiles + two feature tiles, one resident weight chunk
    const size_t lds_sh = 16 + (size_t)a.c_out * sizeof(float);
    static const bool no_wdma = getenv("AMTX_CONVG_NO_WDMA") != nullptr;     // A/B switch
    int w_all = !PIPE && nchunks > 1 && lds_x + nchunks * wchunk + lds_f + lds_sh <= 160 * 1024;
    if (!w_all && !PIPE && NS == 1 && nchunks > 1 && !no_wdma && lds_x + 2 * wchunk + lds_f + lds_sh <= 160 * 1024) w_all = 2;
    static const bool no_csplit = getenv("AMTX_CONVG_NO_CSPLIT") != nullptr;  // A/B switch
    if (!w_all && !PIPE && NS == 2 && nchunks > 1 && !no_csplit) w_all = 3;
    size_t lds = PIPE ? 2 * lds_x + wchunk + lds_f + (size_t)(GROWS + 2) * (FT + 4) * 16
                      : lds_x + (w_all == 1 ? nchunks : (w_all == 2 ? 2 : 1)) * wchunk + lds_f;
    if (PIPE) AMTX_REQUIRE(nchunks == 1, "conv3x3 (general): the pipelined 32-channel variant takes one C_out chunk");
    const int sh_off = (int)((lds + 15) / 16 * 16);           // [c_out] fp32 shift behind everything else
    lds = (size_t)sh_off + (size_t)a.c_out * sizeof(float);
    AMTX_REQUIRE(lds <= 160 * 1024, "conv3x3 (general): tile + weights + features do not fit the LDS (%zu bytes)", lds);
    if (F16IN) AMTX_REQUIRE((int64_t)a.T * a.F < (1ll << 27), "conv3x3 (general): a clip's 16-bit feature map must be smaller than 2 GiB");
    auto kern = conv3x3_gen_kernel<CI16, NTC, NS, FT, IN_TYPE, OUT_TYPE, KS1, FCL, CMAX, F16IN, STRIP>;
    AMTX_GRANT_LDS(kern, lds);
    // persistent grid: as many blocks as fit the chip at once (LDS allows 160 KiB / lds per CU), a multiple of 8 per group so a
    // block's tiles stay on its XCD
    const int per_cu = std::max(1, (int)(160 * 1024 / lds));
    int64_t gx = std::max<int64_t>(8, (256 * per_cu / std::max(1, a.groups)) / 8 * 8);
    if (w_all == 3) {                                         // a multiple of 8 nchunks blocks, every one with a first tile
        gx = std::max<int64_t>(8 * nchunks, gx / (8 * nchunks) * (8 * nchunks));
        while (gx > 8 * nchunks && gx / nchunks > ntiles) gx -= 8 * nchunks;
        if (gx / nchunks > ntiles) { w_all = 0; gx = std::min<int64_t>(std::max<int64_t>(8, (256 * per_cu / std::max(1, a.groups)) / 8 * 8), ntiles); }
    } else if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)a.groups), dim3(16 * FT), lds, stream, a, ntf, ntt, nchunks, (int)ntiles, w_all, sh_off, f_base);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int CI16, int NTC>
int dispatch_gen(const ConvArgs& a, hipStream_t s) {
    if (a.feats16) {                                            // fused first conv from 16-bit channels-last features (amtx_cqt_forward16)
        if constexpr (CI16 == 2) {
            if (amtx_conv1g_tapk(a.c_in, a.planes) && a.out_type == AMTX_T_BF16) {
                // a map whose last column tile would hold 8 columns or fewer (72 = 2 x 32 + 8): those columns as strip tiles, in a launch of
                // their own (AMTX_CONVG_NO_STRIP=1: one launch of 32-column tiles, the A/B switch)
                static const bool no_strip = getenv("AMTX_CONVG_NO_STRIP") != nullptr;
                const int fe = a.F & ~1, rem = fe % 32;
                if (!no_strip && fe > 32 && rem > 0 && rem <= 8) {
                    int rc = launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 3, true, 0, true>(a, s, 0, fe / 32);
                    if (rc != AMTX_OK) return rc;
                    return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 3, true, 0, true, true>(a, s, fe - rem, 0);
                }
                return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 3, true, 0, true>(a, s);
            }
        }
        amtx_set_error("conv3x3 (general): 16-bit channels-last features: 2 .. 8 input channels, 32 first-layer channels, one-plane modes only");
        return AMTX_ERR_UNSUPPORTED;
    }
    if (a.feats) {                                              // fused first conv
        const int ks1 = (9 * a.c_in + 31) / 32;
        if (amtx_conv1g_tapk(a.c_in, a.planes) && a.out_type == AMTX_T_BF16) {
            if constexpr (CI16 == 2) {       // the HCQT shape (6 harmonics, amt_tools/features/hvqt.py:107-133) has its own item count
                if (a.c_in <= 6) return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 3, true, 6>(a, s);
            }
            return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 3, true>(a, s);
        }
        if (a.planes == 1 && a.out_type == AMTX_T_BF16 && a.c_in == 1) {   // one input channel (2 .. 8 channels: the tap-major variant above)
            if constexpr (CI16 == 3) {       // OnsetsFrames2 as shipped (model_complexity 3, 229 mel bins = 7 x 32 + 4 columns that reach a pooled output)
                static const bool no_strip = getenv("AMTX_CONVG_NO_STRIP") != nullptr;
                const int fe = a.F & ~1, rem = fe % 32;
                if (!no_strip && fe > 32 && rem > 0 && rem <= 8) {
                    int rc = launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 1, false, 1>(a, s, 0, fe / 32);
                    if (rc != AMTX_OK) return rc;
                    return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 1, false, 1, false, true>(a, s, fe - rem, 0);
                }
            }
            return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16, 1, false, 1>(a, s);
        }
        // two-plane modes.  2 .. 8 input channels: tap-major as well (round 5: the K-major form gathered and split 8 fp32 values per lane, k-step
        // and 16 positions -- 52 % of the HCQT model's x3 conv2; the packed weights follow amtx_conv1g_tapk, so there is no run-time switch back)
        // (32-column tiles for the 32-channel shape -- eight waves, 149 KB of LDS, 256 registers with 12 spilled -- measured the same: 4.80 vs 4.83 ms)
        if (a.planes == 2 && amtx_conv1g_tapk(a.c_in, 2)) {
            if (a.out_type == AMTX_T_F32) return launch_gen<CI16, NTC, 2, 16, AMTX_T_F32, AMTX_T_F32, 3, true>(a, s);
            if constexpr (CI16 == 2) {   // a2 as split planes for convx.hip's conv3
                if (a.out_type == AMTX_T_SPLIT && a.out_split > 0) return launch_gen<CI16, NTC, 2, 16, AMTX_T_F32, AMTX_T_SPLIT, 3, true>(a, s);
            }
        }
        if (a.planes == 2 && a.out_type == AMTX_T_F32 && ks1 == 1) return launch_gen<CI16, NTC, 2, 16, AMTX_T_F32, AMTX_T_F32, 1>(a, s);      // one input channel
        amtx_set_error("conv3x3 (general): fused first conv: unsupported c_in / precision");
        return AMTX_ERR_UNSUPPORTED;
    }
    // (16-column tiles = two 256-thread blocks per CU were measured too: 22.5 vs 15.9 ms for conv2 at mc 3, the kernels need more
    // than 256 VGPRs so only one of the two blocks is resident)
    if (a.planes == 1 && a.in_type == AMTX_T_BF16 && a.out_type == AMTX_T_BF16) return launch_gen<CI16, NTC, 1, 32, AMTX_T_BF16, AMTX_T_BF16>(a, s);
    if (a.planes == 2 && a.in_type == AMTX_T_F32 && a.out_type == AMTX_T_F32) return launch_gen<CI16, NTC, 2, 16, AMTX_T_F32, AMTX_T_F32>(a, s);
    amtx_set_error("conv3x3 (general): unsupported precision/type combination");
    return AMTX_ERR_UNSUPPORTED;
}

}  // namespace

// C_out chunk (in 16-channel tiles) the general kernel keeps in LDS at a time
int amtx_conv3x3_gen_ntc(int c_in, int c_out) {
    if (c_in == 48 && c_out % 48 == 0) return 3;
    if (c_in == 32 && c_out == 32) return 2;          // model_complexity 2 with a multi-channel first conv (HCQT), see ofmodel.hip
    if (c_in == 64 && c_out % 32 == 0) return 2;      // model_complexity 4 (64 -> 64, 64 -> 128): 36 KiB of weights per chunk and plane
    if (c_in == 80 && c_out % 16 == 0) return 1;      // model_complexity 5 (80 -> 80, 80 -> 160: 5 and 10 tiles): 23 KiB per chunk and plane
    return 0;
}

size_t amtx_conv3x3_gen_wfrag_elems(int c_in, int c_out, int planes) { return (size_t)(c_out / 16) * planes * g_wfrags_per_tile(c_in / 16) * 512; }

// host packing: weight (c_out, c_in, 3, 3) fp32 * scale[c_out] -> [chunk][fragment][plane][lane][8]; fragments of a chunk: the full
// steps in (tap, tile, step) order, then the paired tails A (kw, tile): taps (0,kw) | (1,kw); B (tile): (2,0) | (2,1); C (tile): (2,2) | 0
void amtx_conv3x3_gen_pack_host(const float* w, const float* scale, int c_in, int c_out, int planes, bf16_t* out) {
    const int ntc = amtx_conv3x3_gen_ntc(c_in, c_out);
    const int ci16 = c_in / 16, n32 = ci16 / 2, n16 = ci16 % 2;
    const int nmain = 9 * ntc * n32, nfrag = nmain + 5 * ntc * n16;
    const int nchunks = c_out / (16 * ntc);
    auto put = [&](bf16_t* frag, int l, int j, float v) {
        const bf16_t hi = f32_to_bf16_rn(v);
        frag[l * 8 + j] = hi;
        if (planes == 2) frag[512 + l * 8 + j] = f32_to_bf16_rn(v - bf16_to_f32(hi));
    };
    for (int ch = 0; ch < nchunks; ++ch) {
        bf16_t* cbase = out + (size_t)ch * nfrag * planes * 512;
        for (int nt = 0; nt < ntc; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15, gq = l >> 4;
                const int co = ch * 16 * ntc + (row >> 2) * (4 * ntc) + 4 * nt + (row & 3);
                const float sc = scale ? scale[co] : 1.0f;
                auto wv = [&](int ci, int tap) { return w[((size_t)co * c_in + ci) * 9 + tap] * sc; };
                for (int tap = 0; tap < 9; ++tap)
                    for (int ks = 0; ks < n32; ++ks)
                        for (int j = 0; j < 8; ++j)
                            put(cbase + (size_t)(((tap * ntc + nt) * n32) + ks) * planes * 512, l, j, wv(32 * ks + 8 * gq + j, tap));
                if (!n16) continue;
                const int ct = 32 * n32 + 8 * (gq & 1);                // tail channels of this lane group: ct .. ct + 7
                for (int j = 0; j < 8; ++j) {
                    for (int kw = 0; kw < 3; ++kw)                   // A: tap rows 0 | 1 at column shift kw
                        put(cbase + (size_t)(nmain + kw * ntc + nt) * planes * 512, l, j, wv(ct + j, (gq < 2 ? 0 : 3) + kw));
                    put(cbase + (size_t)(nmain + 3 * ntc + nt) * planes * 512, l, j, wv(ct + j, gq < 2 ? 6 : 7));      // B: (2,0) | (2,1)
                    put(cbase + (size_t)(nmain + 4 * ntc + nt) * planes * 512, l, j, gq < 2 ? wv(ct + j, 8) : 0.0f);  // C: (2,2) | zero
                }
            }
    }
}

// fused first conv (c_in -> c_mid channels, c_mid = this layer's C_in): fragments [tile of 16 channels][k-step][plane][lane][8],
// k = 32 ks + 8 (lane >> 4) + j over (ci, kh, kw) in the weight tensor's order, zero past 9 c_in
// (amtx_conv1g_tapk: k = 8 tap + ci instead, three steps)
size_t amtx_conv1g_wfrag_elems(int c_in, int c_mid, int planes) {
    return (size_t)(c_mid / 16) * (amtx_conv1g_tapk(c_in, planes) ? 3 : (9 * c_in + 31) / 32) * planes * 512;
}

void amtx_conv1g_pack_host(const float* w, const float* scale, int c_in, int c_mid, int planes, bf16_t* out) {
    if (amtx_conv1g_tapk(c_in, planes)) {
        for (int nt = 0; nt < c_mid / 16; ++nt)
            for (int ks = 0; ks < 3; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 16 * nt + (l & 15), tap = 4 * ks + (l >> 4);
                        const float v = (tap < 9 && j < c_in) ? w[((size_t)co * c_in + j) * 9 + tap] * (scale ? scale[co] : 1.0f) : 0.0f;
                        const bf16_t hi = f32_to_bf16_rn(v);
                        const size_t base = ((size_t)(nt * 3 + ks) * planes) * 512 + (size_t)l * 8 + j;
                        out[base] = hi;
                        if (planes == 2) out[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                    }
        return;
    }
    const int kvalid = 9 * c_in, ks1 = (kvalid + 31) / 32;
    for (int nt = 0; nt < c_mid / 16; ++nt)
        for (int ks = 0; ks < ks1; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int co = 16 * nt + (l & 15), k = 32 * ks + 8 * (l >> 4) + j;
                    const float v = k < kvalid ? w[(size_t)co * kvalid + k] * (scale ? scale[co] : 1.0f) : 0.0f;
                    const bf16_t hi = f32_to_bf16_rn(v);
                    const size_t base = ((size_t)(nt * ks1 + ks) * planes) * 512 + (size_t)l * 8 + j;
                    out[base] = hi;
                    if (planes == 2) out[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                }
}

// does the fused-first-conv variant fit the LDS for this layer?
bool amtx_conv3x3_gen_can_fuse1(int c_in, int c_mid, int c_out, int planes) {
    const bool tapk = amtx_conv1g_tapk(c_in, planes);
    if (c_in < 1 || c_in > (tapk ? 8 : 7) || !amtx_conv3x3_gen_ntc(c_mid, c_out)) return false;
    const int ft = planes == 2 ? 16 : 32, ntc = amtx_conv3x3_gen_ntc(c_mid, c_out);
    const bool pipe = tapk && c_mid == 32 && planes == 1;      // two input tiles and two feature tiles (conv3x3_gen_kernel PIPE)
    const size_t feat = tapk ? (size_t)(pipe ? 2 : 1) * planes * ((GROWS + 2) * (ft + 4) * 16 + 128) : (size_t)c_in * (GROWS + 2) * (ft + 5) * sizeof(float);
    const size_t lds = (size_t)(pipe ? 2 : 1) * planes * (c_mid / 8) * g_cplane(ft) + (size_t)ntc * planes * g_wfrags_per_tile(c_mid / 16) * 1024 + feat +
                       (size_t)(c_mid / 16) * (tapk ? 3 : (9 * c_in + 31) / 32) * planes * 1024 + c_mid * 4 + 16 + (size_t)c_out * 4;   // + this layer's shift
    return lds <= 160 * 1024;
}

int amtx_launch_conv3x3_gen(const ConvArgs& a, int c_in, hipStream_t stream) {
    AMTX_REQUIRE((a.in || a.feats || a.feats16) && a.wfrag && a.shift && a.out, "conv3x3 (general): null pointer");
    if (a.feats16) AMTX_REQUIRE(!a.feats && !a.f_clip_max && a.w1frag && a.shift1 && ((uintptr_t)a.feats16 % 16) == 0,
                                "conv3x3 (general): 16-bit features: 16-byte aligned, instead of `feats`, with w1frag / shift1");
    if (a.feats && amtx_conv1g_tapk(a.c_in, a.planes))
        AMTX_REQUIRE(a.f_stride_c >= 0 && a.f_stride_t >= 0 && a.f_stride_f >= 0 &&
                         (int64_t)(a.c_in - 1) * a.f_stride_c + (int64_t)(a.T - 1) * a.f_stride_t + (int64_t)(a.F - 1) * a.f_stride_f < (1ll << 31),
                     "conv3x3 (general): a clip's feature view must span fewer than 2^31 elements with non-negative strides");
    if (a.feats) AMTX_REQUIRE(a.w1frag && a.shift1 && a.c_in > 0 && a.c_in <= (amtx_conv1g_tapk(a.c_in, a.planes) ? 8 : 7),
                              "conv3x3 (general): fused first conv needs w1frag / shift1 and c_in <= 7 (8 in the one-plane modes)");
    AMTX_REQUIRE(a.B > 0 && a.T > 0 && a.F >= 2 && a.groups > 0, "conv3x3 (general): bad sizes");
    AMTX_REQUIRE(a.planes == 1 || a.planes == 2, "conv3x3 (general): planes must be 1 or 2");
    if (c_in == 48 && amtx_conv3x3_gen_ntc(c_in, a.c_out) == 3) return dispatch_gen<3, 3>(a, stream);
    if (c_in == 32 && amtx_conv3x3_gen_ntc(c_in, a.c_out) == 2) return dispatch_gen<2, 2>(a, stream);
    if (c_in == 64 && amtx_conv3x3_gen_ntc(c_in, a.c_out) == 2) return dispatch_gen<4, 2>(a, stream);
    if (c_in == 80 && amtx_conv3x3_gen_ntc(c_in, a.c_out) == 1) return dispatch_gen<5, 1>(a, stream);
    amtx_set_error("conv3x3 (general): unsupported channel counts %d -> %d", c_in, a.c_out);
    return AMTX_ERR_UNSUPPORTED;
}
