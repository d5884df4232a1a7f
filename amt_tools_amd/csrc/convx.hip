// Two-plane ("x3") convolution stages on split-plane maps, gfx950 -- the precision that is inside north_star's 1e-4
// (amt_tools/models/onsetsframes.py:375-416: Conv2d 3x3 pad 1 + BatchNorm2d + ReLU + MaxPool(1,2); the reference computes in fp32,
// amt_tools/inference.py:35).
//
// conv.hip's two-plane variants keep both weight planes in registers (288 for the 64-channel layer), so they run ONE wave per SIMD and
// one block per CU -- and with nobody else on the CU every tile's staging (global loads, fp32 -> hi / lo conversion, LDS stores, two
// barriers) is exposed: conv3 reached 43 % of the three-MFMA roof.  Here the maps are AMTX_T_SPLIT (two 16-bit planes, written once by
// the producing kernel's epilogue), so a tile is a pure copy and travels HBM / L2 -> LDS by DMA (global_load_lds_dwordx4) into the
// OTHER of two LDS buffers while the current tile is on the matrix cores: no staging instructions but ~19 address computations per
// wave and tile, one barrier per tile.
//
// Same swapped implicit GEMM (D' = W . X^T), same weight fragments (amtx_conv3x3_pack_host), same tap order and the same three
// products per fragment pair (hi.hi, hi.lo, lo.hi) as conv3x3_kernel<NT, 2, ...>: identical bits.

#include "amtx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int CIN = 32;
constexpr int XT = 16;                          // frames per tile = one MFMA N-tile
constexpr int XROWS = XT + 2;
constexpr int XFT = 30;                         // frequency columns per tile (even)
constexpr int XPITCH = XFT + 3;                 // LDS positions per tile row: 33 = 1 (mod 4), see xoff
constexpr int XNPOS = XROWS * XPITCH;           // 594
constexpr int XPIECES = (XNPOS + 15) / 16;      // DMA pieces (16 positions x 64 bytes) per plane: 38
constexpr int XPLANE = XPIECES * 1024;          // 38 912 bytes
constexpr int XBUF = 2 * XPLANE;                // one tile, both planes
constexpr int XQ = (2 * XPIECES + 7) / 8;      // pieces per wave and tile: 10 (the last round: waves 0 - 3 only)

// what the padding cells of a tile are copied from (a DMA cannot write a constant)
__device__ uint4 g_convx_zero[4];

__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) { return amtx_mfma_16x16x32(a, b, c); }

// Byte offset of 16-byte chunk c of tile position (row i, column j): conv.hip's tile_off with this tile's pitch.  PITCH = 1 (mod 4)
// rotates consecutive rows through the four 64-byte quarters of a 256-byte bank row; XOR-ing the chunk with 2 ((i >> 2) & 1) puts every
// ds_read_b128 lane group (rows r .. r + 3, r + 12 .. r + 15 of chunk g, rows r + 4 .. r + 11 of chunk g + 1) on 16 distinct slots.
__device__ __forceinline__ int xswz(int i) { return ((i >> 2) & 1) << 1; }

struct XTile { int b, t0, f0; };
__device__ __forceinline__ XTile xtile(int tile, int ntf, int ntt, int ft, int ntiles) {
    tile = (int)xcd_remap((unsigned)tile, (unsigned)ntiles);
    XTile c;
    const int tf = tile % ntf; tile /= ntf;
    const int tt = tile % ntt; tile /= ntt;
    c.b = tile; c.t0 = tt * XT; c.f0 = tf * ft;
    return c;
}

// 8 floats -> the two 16-bit planes (split_bf16x2 pairwise: conv.hip's cvt8)
__device__ __forceinline__ void xcvt8(const float (&f)[8], uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_bf16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

__device__ __forceinline__ void xsettle(const uint4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }

// C_in = 32 -> C_out = 16 NT; in [2 planes][B][T][F][32], out [2 planes][B][T][F / 2][C_out] (channels-last, planes in_split / out_split apart).
// EIGHT waves: wave = (pair group pg = wave >> 1, channel half ch = wave & 1).  A wave holds the weights of ITS half of the output
// channels (9 taps x NT / 2 tiles x 2 planes = 144 registers at 64 channels) and therefore fits the 256 registers of a two-waves-per-SIMD
// block: one wave's epilogue (max / split / stores) and fragment-read waits sit under the other wave's matrix work.  (conv.hip's
// two-plane kernel keeps all 288 weight registers in one wave per SIMD, whose matrix pipe then idles through every epilogue: 74 % busy
// inside its matrix phase, tools/conv_phase_prof.py.)  Each input fragment is read from LDS by both channel halves.
template <int NT>
__global__ __launch_bounds__(512) void convx3_kernel(ConvArgs a, int ft, int ntf, int ntt, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][hi | lo]
    constexpr int COUT = NT * 16;
    constexpr int NW = NT / 2;                                    // 16-channel tiles per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, pg = wave >> 1;
    const int grp = blockIdx.y;
    const int g = lane >> 4, trow = lane & 15;
    const int Fo = a.F >> 1;

    // ---- stationary weights (9 taps x NW tiles x 2 planes) and the folded BatchNorm shift of this lane's 4 NW channels
    uint4 wf[9][NW][2];
    {
        const uint4* w = reinterpret_cast<const uint4*>(a.wfrag + (int64_t)grp * a.w_gs) + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int k = 0; k < NW; ++k)
#pragma unroll
                for (int p = 0; p < 2; ++p) wf[tap][k][p] = w[((tap * NT + NW * ch + k) * 2 + p) * 64];
    }
    const int c0 = g * 4 * NT + 4 * NW * ch;                      // this lane's first output channel
    f32x4_t shr[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) shr[k][r] = a.shift[(int64_t)grp * a.shift_gs + c0 + 4 * k + r];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) xsettle(wf[tap][k][p]);

    const bf16_t* in_grp = reinterpret_cast<const bf16_t*>(a.in) + (int64_t)grp * a.in_gs;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem);
    const int cols = ft + 2;

    // ---- this wave's part of a tile's DMA: pieces q = wave + 8 n of the 2 x 38; lane -> (position p = 16 piece + (lane >> 2), physical
    // chunk lane & 3).  The (row, column) of a piece's cell are tile-invariant but are recomputed per tile from an opaque copy of the
    // lane id (~8 vector instructions per piece and tile) rather than held in 20 registers next to the weights.
    auto issue_tile = [&](const XTile& tc, int buf) {
        // first halo cell of the tile (may lie outside the map: only ever added to offsets of cells that exist)
        const int64_t origin = (((int64_t)tc.b * a.T + (tc.t0 - 1)) * a.F + (tc.f0 - 1)) * CIN;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_convx_zero);
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int lq = lane_o >> 2, lc0 = lane_o & 3;
#pragma unroll
        for (int n = 0; n < XQ; ++n) {
            const int q = wave + 8 * n;
            if (q >= 2 * XPIECES) break;                                   // wave-uniform
            const int pl = q >= XPIECES ? 1 : 0, pp = q - pl * XPIECES;
            const int p = pp * 16 + lq;
            const int i = (p * 1986) >> 16, j = p - i * XPITCH;            // p / 33 (exact for p < 2048)
            const int lc = lc0 ^ xswz(i);
            const int t = tc.t0 - 1 + i, f = tc.f0 - 1 + j;
            const bool ok = p < XNPOS && j < cols && (unsigned)t < (unsigned)a.T && (unsigned)f < (unsigned)a.F;
            const bf16_t* src = ok ? in_grp + (pl ? a.in_split : 0) + origin + ((i * a.F + j) * CIN + lc * 8) : zero;
            glds16(src, lds_base + buf * XBUF + pl * XPLANE + pp * 1024);
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue_tile(xtile(tile, ntf, ntt, ft, ntiles), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int rbase[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) rbase[kh] = ((trow + kh) * XPITCH * 4 + (g ^ xswz(trow + kh))) * 16;

    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const XTile tc = xtile(tile, ntf, ntt, ft, ntiles);
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) issue_tile(xtile(next, ntf, ntt, ft, ntiles), cur ^ 1);

        const char* tb = smem + cur * XBUF;
        const int t = tc.t0 + trow;
        bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + (int64_t)grp * a.out_gs + ((int64_t)tc.b * a.T + t) * Fo * COUT + c0;
        // column pairs of this tile that exist (the last tile of a row of tiles may be narrower); this wave's: pg, pg + 4, ...
        const int npairs = min(ft, ((a.F + 1) & ~1) - tc.f0) >> 1;
        const int np = pg < npairs ? (npairs - pg + 3) >> 2 : 0;
        // Fragment rows travel through a TWO-row register ring: while the 6 NW x 3 MFMAs of one tap row run, the next row (of this pair or
        // of the wave's next pair) is in flight.  A pair takes three rows, so pairs alternate the ring phase: the loop body is two pairs.
        uint4 x[2][4][2];
#define XLOAD(BUF, KH, JP)                                                                                 \
        _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                 \
            const int off = rbase[KH] + (2 * (JP) + cc) * 64;                                              \
            x[BUF][cc][0] = *reinterpret_cast<const uint4*>(tb + off);                                     \
            x[BUF][cc][1] = *reinterpret_cast<const uint4*>(tb + XPLANE + off);                            \
        }
        // per accumulator the order is conv.hip's (taps kw = 0, 1, 2, each hi.hi, hi.lo, lo.hi); the four accumulators (2 columns x NW
        // tiles) take turns and the order is pinned, so that an MFMA never waits for the one or two issued right before it
#define XROW(BUF, KH)                                                                                      \
        _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                                   \
            _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) {                                             \
                _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
                    _Pragma("unroll") for (int k = 0; k < NW; ++k)                                         \
                        acc[e][k] = mfma16(wf[(KH) * 3 + kw][k][pr == 2 ? 1 : 0], x[BUF][kw + e][pr == 1 ? 1 : 0], acc[e][k]); \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }
        // one pair whose first row is in ring slot B0: + shift (the accumulators' initial value), ReLU, MaxPool(1,2) over the (f, f + 1)
        // pair, the two planes of the result
#define XPAIR(B0, JP, JNEXT, LAST)                                                                         \
        {                                                                                                  \
            f32x4_t acc[2][NW];                                                                            \
            _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                  \
                _Pragma("unroll") for (int k = 0; k < NW; ++k) acc[e][k] = shr[k];                         \
            XLOAD((B0) ^ 1, 1, JP)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            XROW(B0, 0)                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            XLOAD(B0, 2, JP)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            XROW((B0) ^ 1, 1)                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            XLOAD((B0) ^ 1, 0, JNEXT)                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            XROW(B0, 2)                                                                                    \
            if (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* see the end of the tile loop */  \
            const int fo = (tc.f0 >> 1) + (JP);                                                            \
            if (t < a.T && fo < Fo) {                                                                      \
                uint32_t h[2 * NW], l[2 * NW];                                                             \
                _Pragma("unroll") for (int k = 0; k < NW; ++k)                                             \
                    _Pragma("unroll") for (int r = 0; r < 4; r += 2) {                                     \
                        const float v0 = fmaxf(fmaxf(acc[0][k][r], acc[1][k][r]), 0.f);                    \
                        const float v1 = fmaxf(fmaxf(acc[0][k][r + 1], acc[1][k][r + 1]), 0.f);            \
                        split_bf16x2(v0, v1, h[2 * k + (r >> 1)], l[2 * k + (r >> 1)]);                    \
                    }                                                                                      \
                bf16_t* d = out + (int64_t)fo * COUT;                                                      \
                if constexpr (NW == 2) {                                                                   \
                    *reinterpret_cast<uint4*>(d) = make_uint4(h[0], h[1], h[2], h[3]);                     \
                    *reinterpret_cast<uint4*>(d + a.out_split) = make_uint4(l[0], l[1], l[2], l[3]);       \
                } else {                                                                                   \
                    *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);                                 \
                    *reinterpret_cast<uint2*>(d + a.out_split) = make_uint2(l[0], l[1]);                   \
                }                                                                                          \
            }                                                                                              \
        }
        if (np > 0) {
            XLOAD(0, 0, pg)
            for (int pi = 0; pi < np; pi += 2) {
                const int ja = pg + 4 * pi;
                const int jb = min(ja + 4, npairs - 1), jc = min(ja + 8, npairs - 1);   // past the end: re-read a valid pair, never used
                XPAIR(0, ja, jb, pi + 1 >= np)
                if (pi + 1 < np) {
                    XPAIR(1, jb, jc, pi + 2 >= np)
                }
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#undef XPAIR
#undef XROW
#undef XLOAD
        // The next tile has landed: every wave waited for its own pieces (vmcnt(0) in front of its LAST pair's stores -- the pieces were
        // issued a whole tile of matrix work ago) and the barrier covers everybody's; everybody is done reading this tile.  The barrier
        // waits for LDS traffic only: the last pair's global stores stay in flight across it (a __syncthreads() here waited for their
        // acknowledgement, ~a quarter of the kernel's wave cycles at s_waitcnt / barrier in the PMC pass).
        lds_only_barrier();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// layer1 + layer2 of a one-channel model in the two-plane mode: Conv(1 -> 32) + BN + ReLU computed into the LDS input tile of
// Conv(32 -> 32) + BN + ReLU + MaxPool(1,2) (conv.hip's fused kernel, KS == 1), a2 written as AMTX_T_SPLIT planes.
//
// conv.hip's version is one wave per SIMD (208 weight registers) and three phases per tile separated by barriers: the first conv --
// 3 % of the flops, but ~200 vector instructions per 64 positions (hi / lo splits of its operands and of its 2048 results) around a
// dependent read -> convert -> MFMA -> convert -> write chain -- is 40 % of a tile and the matrix pipe idles through it
// (tools/conv_phase_prof.py: 10 850 of 27 100 cycles).  Here the block is WAVE-SPECIALISED BY LAYER, as convf.hip is for the one-plane modes:
//   * waves 0 - 3 ("layer2 waves", one per SIMD) hold layer2's 144 weight registers and do nothing but its matrix loop: pairs w, w + 4, ..
//     of tile k out of one of TWO a1 tiles in LDS (16 x 24 outputs: 18 x 26 positions x 32 channels, 33 KB per plane, position-major, pitch 29);
//   * waves 4 - 7 ("layer1 waves", the other wave of every SIMD) stage the features of tile k + 2 (split into the two 16-bit planes ONCE, at
//     the store), run layer1 of tile k + 1 into the OTHER a1 tile (Toeplitz product, conv.hip) and convert its results -- vector work that
//     issues beside the layer2 wave's MFMAs instead of in front of them;
//   * ONE barrier per tile.  A first version that gave every wave both layers' work for half the channels ran at conv.hip's speed (8.5 - 8.9
//     against 9.1 ms per 512 clips): matrix time and vector time of a wave ADD, whatever the partner does (measured by switching parts off:
//     layer2 alone 6.6 ms -- 5.65 without its reads and epilogue --, layer1 2.0, staging 0.7).
// Both roles index ONE weight-register array (a kernel's waves share one register allocation: two arrays would not fit 256).
// Same arithmetic as conv.hip's kernel (Toeplitz first conv, the same fragments, tap order and product order): identical bits.
// MC = false: the one-channel kernel described above.  MC = true (round 6): 2 .. 8 input channels (HCQT: one per harmonic) with the features
// handed over as the two 16-bit planes themselves ([2][B][T][F][8] channels-last, amtx_cqt_forward16_split): the first conv is the TAP-MAJOR
// product of convg.hip's fused first conv (k = 8 tap + channel, three 32-deep steps; k-group g of step s = the 8 channel slots of the feature
// position at tap 4 s + g: ONE 16-byte LDS read per plane), same fragments (amtx_conv1g_pack_host), same step and product order -> the
// same a1 bits; a tile's feature staging is 3.4 sixteen-byte copies per thread instead of strided fp32 loads + splits.  Tiles are 18 output
// columns wide (the HCQT map's 72 bins = 4 tiles; 24 would need 178 KB of LDS with channels-last feature tiles).
template <bool MC>
struct Y12 {
    static constexpr int FT = MC ? 18 : 24;                      // output columns per tile
    static constexpr int P = MC ? 21 : 29;                       // a1 tile pitch in positions (>= FT + 2, = 1 mod 4)
    static constexpr int PLANE = XROWS * P * 64;                 // 33 408 / 24 192 bytes
    static constexpr int BUF = 2 * PLANE;
    static constexpr int FROWS = XROWS + 2;                      // feature rows of a tile
    // MC = false: one 16-bit value per cell, row pitch FW values (80 bytes -> 16 consecutive rows on 16 distinct 8-byte slots of a bank row)
    // MC = true: 16 bytes (8 channel slots) per cell, row pitch FW cells = 23 x 16 = 112 (mod 256) bytes: the 16 rows of a ds_read_b128 lane
    // group on 16 distinct 16-byte slots
    static constexpr int FW = MC ? 23 : 40;
    static constexpr int FPLANE = MC ? FROWS * FW * 16 : (FROWS * FW + 16) * 2;   // one 16-bit plane of a feature tile
    static constexpr int FEAT = 2 * FPLANE;                      // hi | lo: the features are split ONCE (MC: by the front-end)
    static constexpr int LDS = 2 * BUF + 2 * FEAT + 2 * 256 * 16 + 256;   // + one 16-byte scratch slot per layer1 thread and plane for masked stores + the shift tables
};
constexpr int YFT = Y12<false>::FT, YP = Y12<false>::P, YPLANE = Y12<false>::PLANE, YBUF = Y12<false>::BUF, YFROWS = Y12<false>::FROWS, YFW = Y12<false>::FW,
              YFPLANE = Y12<false>::FPLANE, YFEAT = Y12<false>::FEAT, YLDS = Y12<false>::LDS;
constexpr int YFPRE = 3;                        // feature values per layer1-wave thread and tile (20 x 28 = 560 <= 768)
constexpr int YMPRE = 4;                        // MC: 16-byte feature cells per layer1-wave thread and tile (2 planes x 20 x 22 = 880 <= 1024)
static_assert(Y12<true>::LDS <= 160 * 1024 && Y12<false>::LDS <= 160 * 1024, "LDS");

// Chunk swizzle of the a1 tile: layer1's lanes run along ROWS (8 rows of one column and chunk per ds_write_b128 lane group; LDS stores bank
// mod 128 bytes), layer2's fragment reads take rows r .. r + 15 of one column (ds_read_b128, mod 256 bytes).  (i >> 1) & 3 keeps the reads
// conflict-free for every tap row and the stores conflict-free too (xswz: 4-way stores; tools/lds_swizzle_check_convx.py).
__device__ __forceinline__ int yswz(int i) { return (i >> 1) & 3; }

template <bool MC>
__global__ __launch_bounds__(512) void convx12_kernel(ConvArgs a, int ft, int ntf, int ntt, int ntiles, int inv_fcols) {
    // this instantiation's tile shape (shadows the one-channel constants of the same names)
    constexpr int YP = Y12<MC>::P, YPLANE = Y12<MC>::PLANE, YBUF = Y12<MC>::BUF, YFROWS = Y12<MC>::FROWS, YFW = Y12<MC>::FW, YFPLANE = Y12<MC>::FPLANE,
                  YFEAT = Y12<MC>::FEAT, YLDS = Y12<MC>::LDS;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][a1 hi | a1 lo] | [2] feature tiles | scratch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool l2role = wave < 4;                                 // waves w and w + 4 share a SIMD: one of each role
    const int wq = wave & 3;                                      // index inside the role
    const int grp = blockIdx.y;
    const int g = lane >> 4;
    const int Fo = a.F >> 1;
    const int cols = ft + 2, fcols = ft + 4;

    // ---- stationary weights, ONE array for both roles: layer2's 9 taps x 2 tiles x 2 planes (36 fragments), or layer1's Toeplitz fragments
    // (4 columns x 2 tiles x 2 planes = the first 16)
    uint4 wreg[36];
    {
        const uint4* w = reinterpret_cast<const uint4*>(l2role ? a.wfrag + (int64_t)grp * a.w_gs : a.w1frag + (int64_t)grp * a.w1_gs) + lane;
#pragma unroll
        for (int i = 0; i < 36; ++i) wreg[i] = w[(l2role || i < (MC ? 12 : 16) ? i : 0) * 64];
    }
    const int c0 = g * 8;                                         // this lane's first channel (of both layers)
#pragma unroll
    for (int i = 0; i < 36; ++i) xsettle(wreg[i]);
    // the folded BatchNorm shifts (the accumulators' initial values) live in LDS: 8 registers per lane are what this kernel does not have
    float* shtab = reinterpret_cast<float*>(smem + YLDS - 256);                 // [layer2's 32 | layer1's 32]
    if (tid < 64) shtab[tid] = tid < 32 ? a.shift[(int64_t)grp * a.shift_gs + tid] : a.shift1[(int64_t)grp * 32 + tid - 32];
    const f32x4_t* sh = reinterpret_cast<const f32x4_t*>(shtab + (l2role ? 0 : 32) + c0);   // sh[k]: channels c0 + 4 k ..

    char* feat0 = smem + 2 * YBUF;
    const int ltid = tid & 255;                                   // thread index inside the layer1 half
    char* scratch = smem + 2 * YBUF + 2 * YFEAT + ltid * 16;
    // feature columns past the staged ones are only ever multiplied by zero weights but must hold finite values (NaN x 0 is NaN): zero
    // both tiles once (the staged cells are rewritten per tile)
    for (int i = tid; i < 2 * YFEAT / 4; i += 512) reinterpret_cast<float*>(feat0)[i] = 0.f;
    __syncthreads();

    // ---- feature staging (layer1 waves): cells it = ltid + 256 n of the 20 x fcols tile; issue -> registers, (dB scaling +) split + store later
    float fpre[YFPRE];
    float fown = 0.f, fref = 0.f;
    // MC: cells it = ltid + 256 n of [2 planes][20 rows][fcols] 16-byte cells, straight copies of the caller's two planes
    uint4 mpre[MC ? YMPRE : 1];
    const int64_t f16_plane = MC ? a.in_split / 8 : 0;            // 16-byte cells between the hi and the lo plane of feats16
    auto feat_issue = [&](const XTile& tc) {
        if constexpr (MC) {
            const uint4* fb = reinterpret_cast<const uint4*>(a.feats16) + (int64_t)tc.b * a.T * a.F;
            int tid_l = ltid;
            asm volatile("" : "+v"(tid_l));
#pragma unroll
            for (int n = 0; n < YMPRE; ++n) {
                const int it = tid_l + 256 * n;
                const int cells = YFROWS * fcols;
                const int pl = it >= cells ? 1 : 0, ic = it - pl * cells;
                const int fi = (ic * inv_fcols) >> 16, fj = ic - fi * fcols;
                const int t = tc.t0 - 2 + fi, f = tc.f0 - 2 + fj;
                const bool ok = it < 2 * cells && (unsigned)t < (unsigned)a.T && (unsigned)f < (unsigned)a.F;
                // unconditional load from a clamped offset (a guarded load is waited for at the join of its branch); masked at the store
                const uint4 v = fb[(ok ? (int64_t)t * a.F + f : 0) + (pl ? f16_plane : 0)];
                mpre[n] = ok ? v : make_uint4(0, 0, 0, 0);
            }
            return;
        }
        const float* fb = a.feats + (int64_t)tc.b * a.f_stride_b;
        int tid_l = ltid;
        asm volatile("" : "+v"(tid_l));
#pragma unroll
        for (int n = 0; n < YFPRE; ++n) {
            const int it = tid_l + 256 * n;
            const int fi = (it * inv_fcols) >> 16, fj = it - fi * fcols;      // it / fcols, exact for it < 1024 (checked at launch)
            const int t = tc.t0 - 2 + fi, f = tc.f0 - 2 + fj;
            fpre[n] = a.f_clip_max ? -1.f : 0.f;                  // power is never negative: -1 marks the zero padding
            if (it < YFROWS * fcols && (unsigned)t < (unsigned)a.T && (unsigned)f < (unsigned)a.F) fpre[n] = fb[(int64_t)t * a.f_stride_t + f * a.f_stride_f];
        }
        if (a.f_clip_max) {
            fown = a.f_clip_max[tc.b];
            fref = a.f_ref ? a.f_ref[tc.b] : fown;
        }
    };
    auto feat_store = [&](int buf) {
        if constexpr (MC) {
            int tid_l = ltid;
            asm volatile("" : "+v"(tid_l));
#pragma unroll
            for (int n = 0; n < YMPRE; ++n) {
                const int it = tid_l + 256 * n;
                const int cells = YFROWS * fcols;
                const int pl = it >= cells ? 1 : 0, ic = it - pl * cells;
                const int fi = (ic * inv_fcols) >> 16, fj = ic - fi * fcols;
                if (it < 2 * cells) *reinterpret_cast<uint4*>(feat0 + buf * YFEAT + pl * YFPLANE + (fi * YFW + fj) * 16) = mpre[n];
            }
            return;
        }
        bf16_t* ftile = reinterpret_cast<bf16_t*>(feat0 + buf * YFEAT);
        DbScale dbs = {0.f, 0.f};
        if (a.f_clip_max) dbs = db_scale_make(fown, fref);
        int tid_l = ltid;
        asm volatile("" : "+v"(tid_l));
#pragma unroll
        for (int n = 0; n < YFPRE; ++n) {
            const int it = tid_l + 256 * n;
            const int fi = (it * inv_fcols) >> 16, fj = it - fi * fcols;
            float v = fpre[n];
            if (a.f_clip_max) {
                const float sv = db_scale_apply(v, dbs);
                v = v < 0.f ? 0.f : sv;
            }
            uint32_t h, l;
            split_bf16x2(v, 0.f, h, l);               // the conversion layer1 applied to its B operand
            if (it < YFROWS * fcols) {
                ftile[fi * YFW + fj] = (bf16_t)h;
                ftile[YFPLANE / 2 + fi * YFW + fj] = (bf16_t)l;
            }
        }
    };

    // ---- layer1 of one tile (layer1 waves): units u = wq, wq + 4 of the 7 main units (16 rows x 4 columns) + the halo unit (rows 16 - 17 x 8
    // column blocks), all 32 channels.  See conv.hip (KS == 1) for the Toeplitz form.
    const int nmain = (cols + 3) >> 2;
    const int nunits = nmain + 1;
    auto layer1 = [&](const XTile& tc, int buf) {
        // per-lane offsets, recomputed per tile from an opaque copy of the lane id rather than held in registers across the tile loop
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int n16 = lane_o & 15, g = lane_o >> 4;
        const int gg = min(g, 2);                                     // k-group 3 has zero weights: re-read group 2's row
        const int fa_main = (n16 + gg) * YFW * 2;
        const int fa_halo = ((XT + (n16 & 1) + gg) * YFW + 4 * (n16 >> 1)) * 2;
        const int oa_main = (n16 * YP * 4 + (g ^ yswz(n16))) * 16;
        const int hrow = XT + (n16 & 1);
        const int oa_halo = ((hrow * YP + 4 * (n16 >> 1)) * 4 + (g ^ yswz(hrow))) * 16;
        const char* fbytes = feat0 + buf * YFEAT;
        char* ob = smem + buf * YBUF;
        const bool interior = tc.t0 >= 1 && tc.t0 + XT < a.T && tc.f0 >= 1 && tc.f0 + ft < a.F;   // no position of the tile is padding
        for (int u = wq; u < nunits; u += 4) {
            const bool mainu = u < nmain;                                                              // scalar
            // B operand of lane (row n, k-group g): 8 consecutive staged values of feature row n + g from column 4 u, both planes (8-byte aligned)
            const int fa = mainu ? fa_main + u * 8 : fa_halo;
            const uint2 h0 = *reinterpret_cast<const uint2*>(fbytes + fa), h1 = *reinterpret_cast<const uint2*>(fbytes + fa + 8);
            const uint2 l0 = *reinterpret_cast<const uint2*>(fbytes + YFPLANE + fa), l1 = *reinterpret_cast<const uint2*>(fbytes + YFPLANE + fa + 8);
            const uint4 bh = make_uint4(h0.x, h0.y, h1.x, h1.y), bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
            f32x4_t acc1[4][2];
            // fragment (q, nt, plane) = wreg[(q * 2 + nt) * 2 + plane]; per accumulator hi.hi, hi.lo, lo.hi as in conv.hip
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc1[q][nt] = mfma16(wreg[(q * 2 + nt) * 2], bh, sh[nt]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc1[q][nt] = mfma16(wreg[(q * 2 + nt) * 2], bl, acc1[q][nt]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc1[q][nt] = mfma16(wreg[(q * 2 + nt) * 2 + 1], bh, acc1[q][nt]);
            const int oa = mainu ? oa_main + u * 256 : oa_halo;
            const bool fast = __builtin_amdgcn_readfirstlane((int)(interior && mainu)) != 0;
            if (fast) {
                // no position of this unit is padding (columns past the tile's last one land in the row's pad cells, which nobody reads)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float y[8];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[nt * 4 + r] = fmaxf(acc1[q][nt][r], 0.f);
                    uint4 hi, lo;
                    xcvt8(y, hi, lo);
                    *reinterpret_cast<uint4*>(ob + oa + q * 64) = hi;
                    *reinterpret_cast<uint4*>(ob + YPLANE + oa + q * 64) = lo;
                }
            } else {
                // border tiles and the halo-row unit: branch-free per-lane padding (ReLU and the zero padding of the map in one v_med3
                // against inf / 0); positions past the tile's pitch go to a scratch slot
                const int jl = mainu ? 4 * u : 4 * (n16 >> 1);
                const int tl = tc.t0 - 1 + (mainu ? n16 : hrow);
                const bool row_ok = (unsigned)tl < (unsigned)a.T;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ok = row_ok && (unsigned)(tc.f0 - 1 + jl + q) < (unsigned)a.F;
                    const float lim = ok ? __builtin_inff() : 0.f;
                    float y[8];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[nt * 4 + r] = __builtin_amdgcn_fmed3f(acc1[q][nt][r], 0.f, lim);
                    uint4 hi, lo;
                    xcvt8(y, hi, lo);
                    const bool in_tile = jl + q < YP - 1;
                    *reinterpret_cast<uint4*>(in_tile ? ob + oa + q * 64 : scratch) = hi;
                    *reinterpret_cast<uint4*>(in_tile ? ob + YPLANE + oa + q * 64 : scratch + 4096) = lo;
                }
            }
        }
    };

    // ---- MC: layer1 of one tile as the tap-major product (convg.hip's fused first conv): groups of 16 positions -- column j of rows 0 .. 15 for
    // j < cols, then the two halo rows (16, 17) as ceil(2 cols / 16) groups -- dealt to the four layer1 waves, TWO groups in flight per wave.
    // Lane (k-group g, position n): step s reads the 8 channel slots of the feature cell at tap 4 s + g of its position, 16 bytes per plane.
    auto layer1m = [&](const XTile& tc, int buf) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int n16 = lane_o & 15, g = lane_o >> 4;
        int tapoff[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int tap = 4 * ks + g, tp = tap < 9 ? tap : 0;   // taps 9 .. 11 meet zero weights: any cell of the (finite) tile will do
            tapoff[ks] = ((tp / 3) * YFW + tp % 3) * 16;
        }
        const char* fbytes = feat0 + buf * YFEAT;
        char* ob = smem + buf * YBUF;
        const int ngroups = cols + ((2 * cols + 15) >> 4);
        f32x4_t sh1m[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) sh1m[nt] = *reinterpret_cast<const f32x4_t*>(shtab + 32 + 16 * nt + 4 * g);
        for (int u0 = wq; u0 < ngroups; u0 += 8) {
            int pi[2], pj[2];
            bool pok[2];
            uint4 ph[2][3], pl[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = u0 + 4 * q;
                const int hp = (u - cols) * 16 + n16;                         // halo groups: position hp of rows 16, 17 laid end to end
                const bool main_g = u < cols;                                 // scalar
                pi[q] = main_g ? n16 : XT + (hp >= cols ? 1 : 0);
                pj[q] = main_g ? u : (hp >= cols ? hp - cols : hp);
                pok[q] = u < ngroups && (main_g || hp < 2 * cols);
                if (!pok[q]) { pi[q] = 0; pj[q] = 0; }
                const int src = (pi[q] * YFW + pj[q]) * 16;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    ph[q][ks] = *reinterpret_cast<const uint4*>(fbytes + src + tapoff[ks]);
                    pl[q][ks] = *reinterpret_cast<const uint4*>(fbytes + YFPLANE + src + tapoff[ks]);
                }
            }
            f32x4_t d[2][2];
            // per accumulator: steps 0, 1, 2, each hi.hi, hi.lo, lo.hi (convg.hip's order); fragment (tile nt, step ks, plane) = wreg[(nt * 3 + ks) * 2 + plane]
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4_t dd = sh1m[nt];
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        dd = mfma16(wreg[(nt * 3 + ks) * 2], ph[q][ks], dd);
                        dd = mfma16(wreg[(nt * 3 + ks) * 2], pl[q][ks], dd);
                        dd = mfma16(wreg[(nt * 3 + ks) * 2 + 1], ph[q][ks], dd);
                    }
                    d[q][nt] = dd;
                }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool inside = pok[q] && (unsigned)(tc.t0 - 1 + pi[q]) < (unsigned)a.T && (unsigned)(tc.f0 - 1 + pj[q]) < (unsigned)a.F;
                const float lim = inside ? __builtin_inff() : 0.f;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = __builtin_amdgcn_fmed3f(d[q][nt][r], 0.f, lim);   // ReLU and the map's zero padding
                    uint2 hi, lo;
                    split_bf16x2(o[0], o[1], hi.x, lo.x);
                    split_bf16x2(o[2], o[3], hi.y, lo.y);
                    // channels 16 nt + 4 g .. + 3 = half (g & 1) of 16-byte chunk 2 nt + (g >> 1) of the position
                    const int off = ((pi[q] * YP + pj[q]) * 4 + ((2 * nt + (g >> 1)) ^ yswz(pi[q]))) * 16 + (g & 1) * 8;
                    *reinterpret_cast<uint2*>(pok[q] ? ob + off : scratch) = hi;
                    *reinterpret_cast<uint2*>(pok[q] ? ob + YPLANE + off : scratch + 4096) = lo;
                }
            }
        }
    };

    // ---- layer2 of one tile (layer2 waves): pairs jp = wq, wq + 4, ... of this tile, all 32 channels, fragment rows through a two-row ring
    auto layer2 = [&](const XTile& tc, int buf) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int trow = lane_o & 15, g = lane_o >> 4;
        const int c0 = g * 8;
        int rbase[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) rbase[kh] = ((trow + kh) * YP * 4 + (g ^ yswz(trow + kh))) * 16;
        const char* tb = smem + buf * YBUF;
        const int t = tc.t0 + trow;
        // clip base (wave-uniform) + a 32-bit element offset per lane: one clip's map is T x F / 2 x 32 < 2^31 elements
        bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + (int64_t)grp * a.out_gs + (int64_t)tc.b * a.T * Fo * 32;
        bf16_t* out_lo = out + a.out_split;
        const unsigned orow = (unsigned)((t * Fo + (tc.f0 >> 1)) * 32 + c0);
        const int npairs = min(ft, ((a.F + 1) & ~1) - tc.f0) >> 1;
        const int np = wq < npairs ? (npairs - wq + 3) >> 2 : 0;
        uint4 x[2][4][2];
#define YLOAD(BUF, KH, JP)                                                                                 \
        _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                 \
            const int off = rbase[KH] + (2 * (JP) + cc) * 64;                                              \
            x[BUF][cc][0] = *reinterpret_cast<const uint4*>(tb + off);                                     \
            x[BUF][cc][1] = *reinterpret_cast<const uint4*>(tb + YPLANE + off);                            \
        }
        // fragment (tap, k, plane) = wreg[(tap * 2 + k) * 2 + plane].  Per accumulator the order is conv.hip's (taps kw = 0, 1, 2, each hi.hi,
        // hi.lo, lo.hi); the four accumulators (2 columns x 2 tiles) take turns and the order is pinned.
#define YROW(BUF, KH)                                                                                      \
        _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                                   \
            _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) {                                             \
                _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
                    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                          \
                        acc[e][k] = mfma16(wreg[(((KH) * 3 + kw) * 2 + k) * 2 + (pr == 2 ? 1 : 0)], x[BUF][kw + e][pr == 1 ? 1 : 0], acc[e][k]); \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }
#define YPAIR(B0, JP, JNEXT)                                                                               \
        {                                                                                                  \
            const f32x4_t s0_ = sh[0], s1_ = sh[1];                                                        \
            f32x4_t acc[2][2] = {{s0_, s1_}, {s0_, s1_}};                                                  \
            YLOAD((B0) ^ 1, 1, JP)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            YROW(B0, 0)                                                                                    \
            YLOAD(B0, 2, JP)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            YROW((B0) ^ 1, 1)                                                                              \
            YLOAD((B0) ^ 1, 0, JNEXT)                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            YROW(B0, 2)                                                                                    \
            const int fo = (tc.f0 >> 1) + (JP);                                                            \
            if (t < a.T && fo < Fo) {                                                                      \
                float y[8];                                                                                \
                _Pragma("unroll") for (int k = 0; k < 2; ++k)                                              \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) y[4 * k + r] = fmaxf(fmaxf(acc[0][k][r], acc[1][k][r]), 0.f); \
                uint4 hi, lo;                                                                              \
                xcvt8(y, hi, lo);                                                                          \
                const unsigned eo = orow + (unsigned)(JP) * 32u;                                           \
                *reinterpret_cast<uint4*>(out + eo) = hi;                                                  \
                *reinterpret_cast<uint4*>(out_lo + eo) = lo;                                               \
            }                                                                                              \
        }
        if (np > 0) {
            YLOAD(0, 0, wq)
            for (int pi = 0; pi < np; pi += 2) {
                const int ja = wq + 4 * pi;
                const int jb = min(ja + 4, npairs - 1), jc = min(ja + 8, npairs - 1);
                YPAIR(0, ja, jb)
                if (pi + 1 < np) {
                    YPAIR(1, jb, jc)
                }
            }
        }
#undef YPAIR
#undef YROW
#undef YLOAD
    };

    // ---- prologue: features of the first two tiles, layer1 of the first
    const int G = (int)gridDim.x;
    int tile = blockIdx.x;
    if (!l2role) {
        if (tile < ntiles) {
            feat_issue(xtile(tile, ntf, ntt, ft, ntiles));
            feat_store(0);
        }
        if (tile + G < ntiles) {
            feat_issue(xtile(tile + G, ntf, ntt, ft, ntiles));
            feat_store(1);
        }
    }
    __syncthreads();
    if (!l2role && tile < ntiles) {
        if constexpr (MC) layer1m(xtile(tile, ntf, ntt, ft, ntiles), 0);
        else layer1(xtile(tile, ntf, ntt, ft, ntiles), 0);
    }
    __syncthreads();

    int cur = 0;
    for (; tile < ntiles; tile += G) {
        if (l2role) {
            layer2(xtile(tile, ntf, ntt, ft, ntiles), cur);
        } else {
            const bool has1 = tile + G < ntiles, has2 = tile + 2 * G < ntiles;
            if (has2) feat_issue(xtile(tile + 2 * G, ntf, ntt, ft, ntiles));
            if (has1) {
                if constexpr (MC) layer1m(xtile(tile + G, ntf, ntt, ft, ntiles), cur ^ 1);
                else layer1(xtile(tile + G, ntf, ntt, ft, ntiles), cur ^ 1);
            }
            if (has2) feat_store(cur);                // tile k's features were consumed an iteration ago
        }
        lds_only_barrier();                           // LDS traffic only: the a2 stores stay in flight across it
        cur ^= 1;
    }
}

template <bool MC>
int launch_x12(const ConvArgs& a, hipStream_t stream) {
    constexpr int FT = Y12<MC>::FT, LDSB = Y12<MC>::LDS;
    const int fe = (a.F + 1) & ~1;
    const int ntf = (fe + FT - 1) / FT;
    const int ft = 2 * (((fe >> 1) + ntf - 1) / ntf);
    const int ntt = (a.T + XT - 1) / XT;
    const int64_t nblocks = (int64_t)ntf * ntt * a.B;
    AMTX_REQUIRE(nblocks < (1ll << 31), "convx12: grid too large");
    auto kern = convx12_kernel<MC>;
    AMTX_GRANT_LDS(kern, LDSB);
    int64_t gx = std::max<int64_t>(8, 256 / std::max(1, a.groups) / 8 * 8);
    if (gx > nblocks) gx = nblocks;
    const int fcols = ft + 4, inv_fcols = 65536 / fcols + 1;
    for (int it = 0; it < (MC ? Y12<MC>::FROWS * fcols : YFPRE * 256); ++it)
        if (((it * inv_fcols) >> 16) != it / fcols) {
            amtx_set_error("convx12: internal: reciprocal division inexact for fcols=%d", fcols);
            return AMTX_ERR_ARG;
        }
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)a.groups), dim3(512), LDSB, stream, a, ft, ntf, ntt, (int)nblocks, inv_fcols);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int NT>
int launch_x3(const ConvArgs& a, hipStream_t stream) {
    const int fe = (a.F + 1) & ~1;
    const int ntf = (fe + XFT - 1) / XFT;
    const int ft = 2 * (((fe >> 1) + ntf - 1) / ntf);
    const int ntt = (a.T + XT - 1) / XT;
    const int64_t nblocks = (int64_t)ntf * ntt * a.B;
    AMTX_REQUIRE(nblocks < (1ll << 31), "convx3: grid too large");
    AMTX_REQUIRE((int64_t)XROWS * a.F * CIN < (1ll << 31), "convx3: map too wide");
    const size_t lds = 2 * (size_t)XBUF;
    auto kern = convx3_kernel<NT>;
    AMTX_GRANT_LDS(kern, lds);
    // persistent grid: one block per CU (152 KiB of LDS each), blocks of one group a multiple of 8 so that a block's tiles stay on its XCD
    int64_t gx = std::max<int64_t>(8, 256 / std::max(1, a.groups) / 8 * 8);
    if (gx > nblocks) gx = nblocks;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)a.groups), dim3(512), lds, stream, a, ft, ntf, ntt, (int)nblocks);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

}  // namespace

// fused layer1 + layer2, two-plane weights, a2 as AMTX_T_SPLIT planes (ConvArgs as for conv.hip's fused kernel).  One input channel: fp32 features
// (or raw power) in `feats`, Toeplitz fragments in `w1frag`.  2 .. 8 input channels (round 6): the features as the two 16-bit planes themselves in
// `feats16` ([B][T][F][8] channels-last, the lo plane `in_split` elements behind the hi plane: amtx_cqt_forward16_split), `w1frag` =
// amtx_conv1g_pack_host's tap-major fragments, `wfrag` = amtx_conv3x3_pack_host's (conv.hip's order, NOT convg.hip's)
int amtx_launch_convx12(const ConvArgs& a, hipStream_t stream) {
    AMTX_REQUIRE(a.w1frag && a.shift1 && a.wfrag && a.shift && a.out && a.c_out == 32 && a.planes == 2 && a.out_type == AMTX_T_SPLIT,
                 "convx12: 32 -> 32 channels, two-plane weights and maps only");
    AMTX_REQUIRE(a.out_split > 0 && a.out_split % 8 == 0 && ((uintptr_t)a.out % 16) == 0, "convx12: planes must be 16-byte aligned");
    if (a.feats16) {
        AMTX_REQUIRE(a.c_in >= 2 && a.c_in <= 8 && !a.feats && !a.f_clip_max, "convx12: 16-bit channels-last features are the 2 .. 8-channel form");
        AMTX_REQUIRE(a.in_split >= (int64_t)a.B * a.T * a.F * 8 && a.in_split % 8 == 0 && ((uintptr_t)a.feats16 % 16) == 0 && (int64_t)a.T * a.F < (1ll << 31),
                     "convx12: feats16 needs its lo plane in_split elements (a multiple of 8, >= B T F 8) behind the hi plane, 16-byte aligned");
        return launch_x12<true>(a, stream);
    }
    AMTX_REQUIRE(a.feats && a.c_in == 1, "convx12: fp32 features are the one-channel form");
    return launch_x12<false>(a, stream);
}

// 32 -> c_out channels on AMTX_T_SPLIT maps; the caller (amtx_launch_conv3x3) has checked pointers, sizes and plane strides
int amtx_launch_convx3(const ConvArgs& a, hipStream_t stream) {
    AMTX_REQUIRE(a.in && a.wfrag && a.shift && a.out && a.planes == 2 && a.in_type == AMTX_T_SPLIT && a.out_type == AMTX_T_SPLIT, "convx3: two-plane maps and weights only");
    AMTX_REQUIRE(a.in_split > 0 && a.in_split % 8 == 0 && a.out_split > 0 && a.out_split % 8 == 0 && ((uintptr_t)a.in % 16) == 0 && ((uintptr_t)a.out % 16) == 0,
                 "convx3: planes must be 16-byte aligned");
    if (a.c_out == 64) return launch_x3<4>(a, stream);
    if (a.c_out == 32) return launch_x3<2>(a, stream);
    amtx_set_error("convx3: unsupported c_out=%d", a.c_out);
    return AMTX_ERR_UNSUPPORTED;
}
