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
#define XROW(BUF, KH)                                                                                      \
        _Pragma("unroll") for (int cc = 0; cc < 4; ++cc)                                                   \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
                const int kw = cc - e;                                                                     \
                if (kw < 0 || kw > 2) continue;                                                            \
                _Pragma("unroll") for (int k = 0; k < NW; ++k) {                                           \
                    acc[e][k] = mfma16(wf[(KH) * 3 + kw][k][0], x[BUF][cc][0], acc[e][k]);                 \
                    acc[e][k] = mfma16(wf[(KH) * 3 + kw][k][0], x[BUF][cc][1], acc[e][k]);                 \
                    acc[e][k] = mfma16(wf[(KH) * 3 + kw][k][1], x[BUF][cc][0], acc[e][k]);                 \
                }                                                                                          \
            }
        // one pair whose first row is in ring slot B0: + shift (the accumulators' initial value), ReLU, MaxPool(1,2) over the (f, f + 1)
        // pair, the two planes of the result
#define XPAIR(B0, JP, JNEXT)                                                                               \
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
                XPAIR(0, ja, jb)
                if (pi + 1 < np) {
                    XPAIR(1, jb, jc)
                }
            }
        }
#undef XPAIR
#undef XROW
#undef XLOAD
        // the next tile has landed (this wave's pieces: vmcnt; everybody's: the barrier) and everybody is done reading this one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
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
