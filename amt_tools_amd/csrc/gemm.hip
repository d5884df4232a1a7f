// MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T + bias   (fc1, LSTM input projections, LogisticBank heads)
//
// Replaces the nn.Linear / nn.LSTM input-projection matmuls of the reference
// (amt_tools/models/onsetsframes.py:422-427 fc1, :498-501 nn.LSTM W_ih, amt_tools/models/common.py:539 output_layer).
//
// * v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Computed "swapped" (D' = W . A^T) so that a lane ends up
//   holding 4 consecutive output columns of one row -> 8/16-byte stores.
// * AMTX_PREC_X3: operands are split hi+lo bf16 planes and every product is hi*hi + hi*lo + lo*hi
//   (3 MFMAs): fp32-class accuracy (~1e-5 relative) at 3/16 of the cost of f32 MFMA.
// * 128x128x32 block tile, 4 waves (2x2) of 64x64, register-staged double-buffered LDS with an XOR chunk
//   swizzle so the 16 rows of a fragment read hit 16 distinct 16-byte slots.

#include "amtx_f16_names.h"
#include "amtx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_BYTES = BM * BK * 2;   // one operand plane: 8 KiB

// 64-byte rows; a fragment read takes row = base16 + (lane & 15), chunk = lane >> 4.  XOR-ing the chunk with
// [0,2,3,1][(row >> 2) & 3] makes every ds_read_b128 lane group hit 16 distinct 16-byte slots
// (tools/lds_swizzle_search.py; the plain (row >> 2) & 3 swizzle is 2-way).
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return (row * 4 + (chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3))) * 16;
}

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
    return amtx_mfma_16x16x32(a, b, c);
}

template <int A_TYPE, int NS>
struct AStage {
    // one 8-element K chunk of one A row.  load() only ISSUES the loads (fp32 A: the raw bits of elements 0-3 in `hi`, 4-7 in `lo`);
    // finish() turns them into the bf16 planes and runs where the tile is stored, BEHIND the k-step's MFMAs (with the conversion inside
    // load() hipcc waited for a k-step's A loads before its first MFMA: fc1 of the two-plane mode 4.61 -> 4.43 ms per 512 clips).
    // A second register stage (loads two k-steps ahead) was measured slower: 4.81 ms.
    uint4 hi, lo;
    // vlo / vhi: bounds of the existing elements (vhi == 0: none), see GemmArgs::a_valid_lo
    __device__ __forceinline__ void load(const char* base, int64_t row_off_elems, bool valid, int64_t vlo = 0, int64_t vhi = 0, int64_t split = 0) {
        if (A_TYPE == AMTX_T_SPLIT) {
            // the planes as the producer wrote them: nothing to convert
            hi = valid ? *reinterpret_cast<const uint4*>(base + row_off_elems * 2) : make_uint4(0, 0, 0, 0);
            lo = valid ? *reinterpret_cast<const uint4*>(base + (split + row_off_elems) * 2) : make_uint4(0, 0, 0, 0);
        } else if (A_TYPE == AMTX_T_BF16) {
            hi = valid ? *reinterpret_cast<const uint4*>(base + row_off_elems * 2) : make_uint4(0, 0, 0, 0);
            lo = make_uint4(0, 0, 0, 0);
        } else {
            float4 x0 = make_float4(0, 0, 0, 0), x1 = make_float4(0, 0, 0, 0);
            if (valid && vhi > 0 && (row_off_elems < vlo || row_off_elems + 8 > vhi)) {
                // a chunk that hangs over an end of the data (or lies outside it): element by element, nothing outside is touched
                float e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int64_t idx = row_off_elems + i;
                    e[i] = (idx >= vlo && idx < vhi) ? reinterpret_cast<const float*>(base)[idx] : 0.f;
                }
                x0 = make_float4(e[0], e[1], e[2], e[3]);
                x1 = make_float4(e[4], e[5], e[6], e[7]);
            } else if (valid) {
                // rows of a strided-row A (the CQT products straight from a caller's audio) start at any 4-byte boundary: the type says so
                // (still one global_load_dwordx4 each; the hardware takes dword-aligned wide loads)
                struct __attribute__((packed, aligned(4))) f32x4_a4 { float x, y, z, w; };
                const f32x4_a4* p = reinterpret_cast<const f32x4_a4*>(base + row_off_elems * 4);
                const f32x4_a4 p0 = p[0], p1 = p[1];
                x0 = make_float4(p0.x, p0.y, p0.z, p0.w);
                x1 = make_float4(p1.x, p1.y, p1.z, p1.w);
            }
            hi = __builtin_bit_cast(uint4, x0);
            lo = __builtin_bit_cast(uint4, x1);
        }
    }
    __device__ __forceinline__ void finish() {
        if (A_TYPE == AMTX_T_F32) {
            const float4 x0 = __builtin_bit_cast(float4, hi), x1 = __builtin_bit_cast(float4, lo);
            const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            uint32_t h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                l[i] = 0;
                if (NS == 2) split_bf16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
                else h[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
            }
            hi = make_uint4(h[0], h[1], h[2], h[3]);
            lo = make_uint4(l[0], l[1], l[2], l[3]);
        }
    }
};

// one 128 x 128 output tile (bx, by) of group grp; the body of gemm_kernel (one problem) and gemm_multi_kernel (several)
template <int A_TYPE, int C_TYPE, int NS>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int grp, int bx, int by, char* smem) {
    // layout: [buf][A planes NS][W planes NS] x TILE_BYTES
    constexpr int BUF_BYTES = 2 * NS * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * BN;

    const char* Abase = reinterpret_cast<const char*>(g.A) + (int64_t)grp * g.a_gs * (A_TYPE == AMTX_T_F32 ? 4 : 2);
    const bf16_t* Wbase = g.W + (int64_t)grp * g.w_gs;
    const int64_t plane_elems = (int64_t)g.n_pad * g.k_pad;

    // per-thread staging slots: 2 chunks of A and 2 of W per k-step (idx = tid, tid + 256)
    const int srow0 = tid >> 2, srow1 = srow0 + 64, schunk = tid & 3;
    const bool arow_ok0 = (m0 + srow0) < g.M, arow_ok1 = (m0 + srow1) < g.M;
    const int64_t a_off0 = (m0 + srow0) * g.lda + schunk * 8, a_off1 = (m0 + srow1) * g.lda + schunk * 8;
    const int64_t w_off0 = (int64_t)(n0 + srow0) * g.k_pad + schunk * 8, w_off1 = (int64_t)(n0 + srow1) * g.k_pad + schunk * 8;
    const int soff0 = lds_off(srow0, schunk), soff1 = lds_off(srow1, schunk);

    AStage<A_TYPE, NS> sa0, sa1;
    uint4 sw0h, sw1h, sw0l = make_uint4(0, 0, 0, 0), sw1l = make_uint4(0, 0, 0, 0);
#define LOAD_TILE(k0)                                                                              \
    do {                                                                                           \
        const bool kok = ((k0) + schunk * 8) < g.K;                                                \
        sa0.load(Abase, a_off0 + (k0), arow_ok0 && kok, g.a_valid_lo, g.a_valid_hi, g.a_split);    \
        sa1.load(Abase, a_off1 + (k0), arow_ok1 && kok, g.a_valid_lo, g.a_valid_hi, g.a_split);    \
        sw0h = *reinterpret_cast<const uint4*>(Wbase + w_off0 + (k0));                             \
        sw1h = *reinterpret_cast<const uint4*>(Wbase + w_off1 + (k0));                             \
        if (NS == 2) {                                                                             \
            sw0l = *reinterpret_cast<const uint4*>(Wbase + plane_elems + w_off0 + (k0));           \
            sw1l = *reinterpret_cast<const uint4*>(Wbase + plane_elems + w_off1 + (k0));           \
        }                                                                                          \
    } while (0)
#define STORE_TILE(buf)                                                                            \
    do {                                                                                           \
        char* sb = smem + (buf) * BUF_BYTES;                                                       \
        sa0.finish();                                                                              \
        sa1.finish();                                                                              \
        *reinterpret_cast<uint4*>(sb + soff0) = sa0.hi;                                            \
        *reinterpret_cast<uint4*>(sb + soff1) = sa1.hi;                                            \
        if (NS == 2) {                                                                             \
            *reinterpret_cast<uint4*>(sb + TILE_BYTES + soff0) = sa0.lo;                           \
            *reinterpret_cast<uint4*>(sb + TILE_BYTES + soff1) = sa1.lo;                           \
        }                                                                                          \
        *reinterpret_cast<uint4*>(sb + NS * TILE_BYTES + soff0) = sw0h;                            \
        *reinterpret_cast<uint4*>(sb + NS * TILE_BYTES + soff1) = sw1h;                            \
        if (NS == 2) {                                                                             \
            *reinterpret_cast<uint4*>(sb + 3 * TILE_BYTES + soff0) = sw0l;                         \
            *reinterpret_cast<uint4*>(sb + 3 * TILE_BYTES + soff1) = sw1l;                         \
        }                                                                                          \
    } while (0)

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = g.k_pad / BK;
    LOAD_TILE(0);
    STORE_TILE(0);
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) LOAD_TILE((kt + 1) * BK);
        const char* b = smem + cur * BUF_BYTES;
        uint4 af[4][NS], wf[4][NS];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ar = wm * 64 + t * 16 + frow;
            const int wr = wn * 64 + t * 16 + frow;
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                af[t][p] = *reinterpret_cast<const uint4*>(b + p * TILE_BYTES + lds_off(ar, fchunk));
                wf[t][p] = *reinterpret_cast<const uint4*>(b + (NS + p) * TILE_BYTES + lds_off(wr, fchunk));
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[nt][mt] = mfma16(wf[nt][0], af[mt][0], acc[nt][mt]);
                if (NS == 2) {
                    acc[nt][mt] = mfma16(wf[nt][0], af[mt][1], acc[nt][mt]);
                    acc[nt][mt] = mfma16(wf[nt][1], af[mt][0], acc[nt][mt]);
                }
            }
        if (kt + 1 < nk) STORE_TILE(cur ^ 1);
        __syncthreads();
    }

#undef LOAD_TILE
#undef STORE_TILE
    // epilogue: lane holds D'[n = 4*(lane>>4) + r][m = lane&15] of every 16x16 tile
    const float* bias = g.bias ? g.bias + (int64_t)grp * g.bias_gs : nullptr;
    char* Cbase = reinterpret_cast<char*>(g.C) + (int64_t)grp * g.c_gs * (C_TYPE == AMTX_T_F32 ? 4 : 2);
    // magnitude epilogue with a running maximum per (group, harmonic): block-local maxima in LDS (the tile buffers are free after
    // the loop's last barrier), then one global atomic per harmonic and block.  Magnitudes are >= 0: uint order == float order.
    unsigned* lmax = reinterpret_cast<unsigned*>(smem);
    const bool want_max = C_TYPE == AMTX_T_F32 && g.pair_map && g.pair_max;
    if (want_max) {
        if (tid < 16) lmax[tid] = 0u;
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
        if (n >= g.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias[n + r];
        }
        if (C_TYPE == AMTX_T_F32 && g.pair_map) {
            // magnitude epilogue: this lane's four columns are two (re, im) pairs; 16 lanes = 16 consecutive rows m -> 64 contiguous
            // bytes of the transposed (pair-major) output
            const int2 pa = g.pair_map[n >> 1], pb = g.pair_map[(n >> 1) + (n + 2 < g.N ? 1 : 0)];
            float* po = g.pair_out + (int64_t)grp * g.pair_gs;
            float ma = 0.f, mb = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int64_t m = m0 + wm * 64 + mt * 16 + (lane & 15);
                const f32x4_t v = acc[nt][mt];
                if (m < g.pair_rows[pa.y & 15]) {
                    const float a = v[0] * v[0] + v[1] * v[1];       // POWER: the consumer (cqt.hip's scaling kernels) takes the logarithm of it, or the root
                    po[(int64_t)pa.x * g.pair_pitch + m] = a;
                    ma = fmaxf(ma, a);
                }
                if (n + 2 < g.N && m < g.pair_rows[pb.y & 15]) {
                    const float a = v[2] * v[2] + v[3] * v[3];
                    po[(int64_t)pb.x * g.pair_pitch + m] = a;
                    mb = fmaxf(mb, a);
                }
            }
            if (want_max) {
                if (ma > 0.f) atomicMax(lmax + (pa.y & 15), __float_as_uint(ma));
                if (mb > 0.f) atomicMax(lmax + (pb.y & 15), __float_as_uint(mb));
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int64_t m = m0 + wm * 64 + mt * 16 + (lane & 15);
            if (m >= g.M) continue;
            const f32x4_t v = acc[nt][mt];
            const float o0 = v[0] + bv[0], o1 = v[1] + bv[1], o2 = v[2] + bv[2], o3 = v[3] + bv[3];
            if (C_TYPE == AMTX_T_F32) {
                *reinterpret_cast<float4*>(Cbase + (m * g.ldc + n) * 4) = make_float4(o0, o1, o2, o3);
            } else if (C_TYPE == AMTX_T_SPLIT) {
                uint32_t h0, h1, l0, l1;
                split_bf16x2(o0, o1, h0, l0);
                split_bf16x2(o2, o3, h1, l1);
                *reinterpret_cast<uint2*>(Cbase + (m * g.ldc + n) * 2) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(Cbase + (g.c_split + m * g.ldc + n) * 2) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<uint2*>(Cbase + (m * g.ldc + n) * 2) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            }
        }
    }
    if (want_max) {
        __syncthreads();
        if (tid < 16 && lmax[tid] != 0u) atomicMax(reinterpret_cast<unsigned*>(g.pair_max) + (int64_t)grp * g.pair_nh + tid, lmax[tid]);
    }
}

template <int A_TYPE, int C_TYPE, int NS>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_tile<A_TYPE, C_TYPE, NS>(g, blockIdx.z, blockIdx.x, blockIdx.y, smem);
}

// Several problems of one kind in ONE launch (the CQT basis products of all pyramid levels: same types, same group count, their own
// A / W / sizes): blockIdx.z = problem * groups + group, the grid covers the largest problem and the other problems' surplus blocks leave.
template <int A_TYPE, int C_TYPE, int NS>
__global__ __launch_bounds__(256) void gemm_multi_kernel(GemmMulti mm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int groups = mm.p[0].groups;
    const int pi = blockIdx.z / groups, grp = blockIdx.z - pi * groups;
    const GemmArgs& g = mm.p[pi];
    if ((int64_t)blockIdx.x * BM >= g.M || (int)blockIdx.y * BN >= g.n_pad) return;
    gemm_tile<A_TYPE, C_TYPE, NS>(g, grp, blockIdx.x, blockIdx.y, smem);
}

// ------------------------------------------------------------------------------------------------
// bf16 x bf16 fast path: 128x128x64 tiles, operands DMA'd HBM/L2 -> LDS with global_load_lds_dwordx4
// (no VGPR round trip), double-buffered, one barrier per 64-deep k-tile.  global_load_lds writes a wave's
// 64 x 16 B linearly (M0 base + lane*16), so the bank swizzle lives on the SOURCE address: LDS chunk q of a
// 128-byte row holds logical chunk q ^ ((row >> 1) & 7); fragment reads apply the same involution and are
// conflict-free for every ds_read_b128 lane group (tools/lds_swizzle_search.py).
// Block ids are remapped so the n-tiles that share an A panel run back to back on one XCD (A is then read
// from HBM once and re-read from that XCD's L2).
constexpr int GBK = 64;
// Chunk swizzle of the W tile.  MFMA row tile t of a wave takes the W rows 16*(i >> 2) + 4*t + (i & 3), i = 0..15 (not 16
// consecutive rows), so that a lane ends up with 16 CONSECUTIVE output columns of one C row (acc[0..3][.][0..3]) and
// the epilogue writes 32/64 contiguous bytes per lane, 128/256 per row and wave, instead of 8-byte pieces.  For that row
// set the XOR below keeps every ds_read_b128 lane group on 16 distinct 16-byte slots.
__device__ __forceinline__ int wswz(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

// Measured and rejected on MI355X (fc1 shape, M = 320000, N = 512, K = 3648; this kernel: 1.20 ms = 994 TFLOP/s):
//   * v_mfma_f32_32x32x16_bf16 with a lane owning 32 consecutive C columns: 1.34 ms;
//   * spreading the 8 DMA instructions over the k-steps instead of issuing them at the top of the iteration: the time moves
//     from the issue phase into the vmcnt(0) wait (two buffers leave no slack for a later issue);
//   * the four-buffer two-group ring below for long K: 1.21 ms, and 1212 vs 1237 TFLOP/s with A cache-resident (it wins for
//     K <= 1024 only, where the epilogue and the cold start of a tile weigh more).
// With A fully cache-resident the same loop reaches 1211 TFLOP/s: ~20 % of its time is HBM/L2 latency, the rest is the
// one-barrier-per-k-tile structure (wave 0 of a block: 41 % fragment reads + MFMAs, 17 % DMA issue, 8 % vmcnt, 34 % barrier).
// TB = tile edge (128: 4 waves as 2x2, 256: 8 waves as 2x4); every wave owns (TB/2) x 64 outputs.
// A 256x256 tile moves 64 KiB per 4.2 M MACs (64 MAC/B); the 128x128 tile's 32 MAC/B sits right at the
// ~64 B/clk/CU the L2 can deliver, so the big tile is used whenever N is a multiple of 256.
template <int C_TYPE, int TB>
__global__ __launch_bounds__(TB * 2, (TB == 128 ? 2 : 1)) void gemm_glds_kernel(GemmArgs g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][A tile | W tile]
    constexpr int TILE = TB * GBK * 2;                            // bytes per operand per buffer
    constexpr int WAVES = TB / 32;                                // 4 or 8
    constexpr int WN = WAVES / 2;                                 // waves along N
    constexpr int MT = (TB / 2) / 16;                             // 16-row tiles per wave along M (4 or 8)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int grp = blockIdx.z;
    const unsigned nbn = g.n_pad / TB;

    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)grp * g.a_gs;
    const bf16_t* Wbase = g.W + (int64_t)grp * g.w_gs;
    const float* bias = g.bias ? g.bias + (int64_t)grp * g.bias_gs : nullptr;
    char* Cbase = reinterpret_cast<char*>(g.C) + (int64_t)grp * g.c_gs * (C_TYPE == AMTX_T_BF16 ? 2 : 4);

    // A row-major (row pitch lda, a k-tile 64 columns further) or in planes of 64 columns (GemmArgs::a_plane: row pitch 64, a k-tile one plane further)
    const int64_t a_pitch = g.a_plane ? GBK : g.lda, a_kstep = g.a_plane ? g.a_plane : GBK;
    // this wave DMAs rows [32*wave, 32*wave+32) of both tiles: 4 instructions x 8 rows each
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#define GLDS_SET_TILE(TILE_ID)                                                                                \
    do {                                                                                                      \
        const unsigned lg = xcd_remap((unsigned)(TILE_ID), (unsigned)ntiles);                                 \
        const int tn0 = (lg % nbn) * TB;                                                                      \
        const int64_t tm0 = (int64_t)(lg / nbn) * TB;                                                         \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                       \
            const int row = (wave * 4 + n) * 8 + (lane >> 3);                                                 \
            const int c = (lane & 7) ^ ((row >> 1) & 7);                                                      \
            const int cw = (lane & 7) ^ wswz(row);                                                            \
            int64_t mr = tm0 + row;                                                                           \
            if (mr >= g.M) mr = g.M - 1; /* rows past M are never stored; keep the read in bounds */          \
            a_src[n] = Abase + mr * a_pitch + c * 8;                                                          \
            w_src[n] = Wbase + (int64_t)(tn0 + row) * g.k_pad + cw * 8;                                       \
        }                                                                                                     \
    } while (0)
    // The DMA is issued from inline asm on purpose: with the builtin, hipcc treats every later ds_read as a possible
    // reader of the in-flight LDS write and puts `s_waitcnt vmcnt(0)` in front of the fragment reads, which
    // serialises load and compute.  Hidden from the compiler, the next tile streams in while this one is on the
    // matrix cores; ordering is restored by hand (vmcnt(0) + barrier at the end of each k-tile).
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + wave * 4096;
#define GLDS_ISSUE(k0, buf)                                                                                   \
    do {                                                                                                      \
        const unsigned sb = lds_base + (buf) * 2 * TILE;                                                      \
        const int64_t a_off = (int64_t)((k0) / GBK) * a_kstep;                                                \
        glds16x4(a_src[0] + a_off, a_src[1] + a_off, a_src[2] + a_off, a_src[3] + a_off, sb);                 \
        glds16x4(w_src[0] + (k0), w_src[1] + (k0), w_src[2] + (k0), w_src[3] + (k0), sb + TILE);              \
    } while (0)
#define GLDS_COMPUTE(buf)                                                                                     \
    do {                                                                                                      \
        const char* b = smem + (buf) * 2 * TILE;                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                    \
            uint4 af[MT], wf[4];                                                                              \
            _Pragma("unroll") for (int t = 0; t < MT; ++t) {                                                  \
                const int ar = wm * (TB / 2) + t * 16 + frow;                                                 \
                af[t] = *reinterpret_cast<const uint4*>(b + ar * 128 + (((kk * 4 + fchunk) ^ ((ar >> 1) & 7)) << 4)); \
            }                                                                                                 \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
                const int wr = wn * 64 + 16 * (frow >> 2) + 4 * t + (frow & 3);   /* see the epilogue */      \
                wf[t] = *reinterpret_cast<const uint4*>(b + TILE + wr * 128 + (((kk * 4 + fchunk) ^ wswz(wr)) << 4)); \
            }                                                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                  \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(wf[nt], af[mt], acc[nt][mt]); \
        }                                                                                                     \
    } while (0)
#define GLDS_SYNC()                                                                                           \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        __syncthreads();                                                                                      \
    } while (0)

    const int nk = g.k_pad / GBK;
    const int frow = lane & 15, fchunk = lane >> 4;

    // Persistent over output tiles (blockIdx.x, + gridDim.x, ...): the first k-tile of the NEXT output tile is already
    // streaming into the free LDS buffer while this tile's last k-tile is on the matrix cores and its C tile is
    // written, so short-K problems (the LSTM input projections, K = 512 / 192) do not pay a cold DMA round trip and an
    // un-overlapped epilogue per 128x128 of output.
    int tile = blockIdx.x;
    GLDS_SET_TILE(tile);
    GLDS_ISSUE(0, 0);
    GLDS_SYNC();
    int cur = 0;
    for (;;) {
        const unsigned lg = xcd_remap((unsigned)tile, (unsigned)ntiles);
        const int n0 = (lg % nbn) * TB;
        const int64_t m0 = (int64_t)(lg / nbn) * TB;
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;

        f32x4_t acc[4][MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt + 1 < nk; ++kt) {
            GLDS_ISSUE((kt + 1) * GBK, cur ^ 1);
            GLDS_COMPUTE(cur);
            GLDS_SYNC();
            cur ^= 1;
        }
        if (has_next) {
            GLDS_SET_TILE(next);
            GLDS_ISSUE(0, cur ^ 1);
        }
        GLDS_COMPUTE(cur);

        // ---- epilogue (the next tile's DMA is in flight underneath): lane (g, m) holds columns nb .. nb+15 of row m
        {
            const int nb = n0 + wn * 64 + 16 * (lane >> 4);
            float bv[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[nt][r] = (bias && nb + 4 * nt < g.N) ? bias[nb + 4 * nt + r] : 0.f;
            constexpr int ES = (C_TYPE == AMTX_T_BF16 ? 2 : 4);
            const bool wide = nb + 16 <= g.N && ((g.ldc * ES) % 16) == 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int64_t m = m0 + wm * (TB / 2) + mt * 16 + (lane & 15);
                if (m >= g.M) continue;
                char* dst = Cbase + (m * g.ldc + nb) * ES;
                float o[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[nt][r] = acc[nt][mt][r] + bv[nt][r];
                if (C_TYPE == AMTX_T_F32) {
                    if (g.C) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            if (wide || nb + 4 * nt < g.N) reinterpret_cast<float4*>(dst)[nt] = make_float4(o[nt][0], o[nt][1], o[nt][2], o[nt][3]);
                    }
                    if (g.copy16) {
                        // bf16 copy of the row into a second matrix (the refinement stage's K-padded input): columns
                        // copy16_col0 + grp * copy16_gs + n, plus copy16_pad zero columns behind the last valid one
                        bf16_t* cp = g.copy16 + m * g.copy16_ld + g.copy16_col0 + (int64_t)grp * g.copy16_gs + nb;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int n4 = nb + 4 * nt;
                            if (n4 < g.N) reinterpret_cast<uint2*>(cp)[nt] = make_uint2(pack_bf16x2(o[nt][0], o[nt][1]), pack_bf16x2(o[nt][2], o[nt][3]));
                            else if (n4 < g.N + g.copy16_pad) reinterpret_cast<uint2*>(cp)[nt] = make_uint2(0u, 0u);
                        }
                    }
                    if (g.roll_out && grp == g.roll_group) {
                        // piano-roll epilogue (LogisticBank.finalize_output, amt_tools/models/common.py:586-620): the row is frame
                        // t of clip b, its logits become sigmoid -> threshold in out[b][key][t]; the 16 lanes of a lane group hold 16
                        // consecutive frames of one key: 64-byte stores.  Same expression as pianoroll_kernel (head.hip): same bits.
                        const unsigned bclip = (unsigned)m / (unsigned)g.roll_T, tfrm = (unsigned)m - bclip * (unsigned)g.roll_T;
                        float* ro = g.roll_out + ((int64_t)bclip * g.N + nb) * g.roll_T + tfrm;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (nb + 4 * nt + r < g.N) {
                                    const float sg = 1.0f / (1.0f + expf(-o[nt][r]));
                                    ro[(int64_t)(4 * nt + r) * g.roll_T] = g.roll_thr < 0.f ? sg : (sg < g.roll_thr ? 0.f : 1.f);
                                }
                    }
                } else if (wide) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        reinterpret_cast<uint4*>(dst)[h] = make_uint4(pack_bf16x2(o[2 * h][0], o[2 * h][1]), pack_bf16x2(o[2 * h][2], o[2 * h][3]),
                                                                     pack_bf16x2(o[2 * h + 1][0], o[2 * h + 1][1]), pack_bf16x2(o[2 * h + 1][2], o[2 * h + 1][3]));
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (nb + 4 * nt < g.N) reinterpret_cast<uint2*>(dst)[nt] = make_uint2(pack_bf16x2(o[nt][0], o[nt][1]), pack_bf16x2(o[nt][2], o[nt][3]));
                }
            }
        }
        if (!has_next) break;
        GLDS_SYNC();
        cur ^= 1;
        tile = next;
    }
#undef GLDS_SET_TILE
#undef GLDS_ISSUE
#undef GLDS_COMPUTE
#undef GLDS_SYNC
}
// ------------------------------------------------------------------------------------------------
// Short-K problems (the LSTM input projections, K = 512 / 192): 256x256 tile, 32-deep k-stages in a FOUR-buffer LDS ring.
// With two 64-deep buffers only one stage is in flight while the other is on the matrix cores; here the waits are counted
// (`s_waitcnt vmcnt(4)`: the newest stage stays in flight), never 0.  The stage stream is flattened across this block's output
// tiles, so the ring never drains at a tile boundary; the C stores of a finished tile are counted exactly (every store
// instruction always issues: rows past M go to a scratch line) so that the vmcnt arithmetic stays valid while they are in
// flight.  bias sits in LDS: a global load in the epilogue would wait in order behind the whole ring.
constexpr int RBK = 32, RST = 4, RTB = 256;
constexpr int RING_MAX_NPAD = 4096;          // bias table: 16 KiB next to the 128 KiB ring
constexpr int ROP = RTB * RBK * 2;          // bytes per operand per stage (16 KiB)
constexpr int RSTAGE = 2 * ROP;
__device__ uint4 g_gemm_trash[4];

__device__ __forceinline__ int rswz(int q) { return (0x78 >> (q * 2)) & 3; }   // [0,2,3,1][q], see lds_off

// ------------------------------------------------------------------------------------------------
// The eight waves are split into two groups (wm = 0 / wm = 1) that run ONE barrier interval apart: while one
// group is on the matrix cores (16 MFMAs of one half of its 128 x 64 block) the other issues its fragment reads and its share
// of the DMA, then they swap -- every SIMD holds one wave of each group, so its matrix pipe always has a wave whose operands
// are already in registers.  A 32-deep stage is two such phases (rows 0-63, rows 64-127 of the wave's block):
//     L1: read W (4) + A rows 0-63 (4) of stage s            | barrier | lgkmcnt(0), 16 MFMA | barrier |
//     L2: vmcnt: my part of stage s+1 landed; read A rows 64-127 (4); DMA stage s+3 -> buffer of stage s-1
//                                                             | barrier | lgkmcnt(0), 16 MFMA [epilogue] | barrier |
// Hazards, by barrier count (group 1 lags group 0 by one interval; interval 4s is group 0's L1 of stage s):
//   * stage s+1 is first read in interval 4s+4; every wave's vmcnt for it sits in its L2 of stage s (intervals 4s+2 / 4s+3),
//     at least one barrier earlier;
//   * the buffer of stage s-1 is last read in group 1's L2 of stage s-1 (interval 4s-1, retired by its lgkmcnt(0) in interval
//     4s); the DMA that overwrites it is issued in L2 of stage s (intervals 4s+2 / 4s+3).
// In flight per wave at the vmcnt: stage s+2 (4 DMAs) [+ the counted C stores when the previous stage ended a tile].
#ifdef AMTX_GEMM_TIMING
// Debug build only (AMTX_EXTRA_FLAGS=-DAMTX_GEMM_TIMING): cycles waves 0 (group 0) and 4 (group 1) of every block spend per section of the
// two-group loop, summed over blocks; read with amtxdbg_gemm_prof().
__device__ unsigned long long g_gemm_prof[2][16];
#define PP_TICK(SLOT)                                                      \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        prof_acc[SLOT] += now_ - prof_t;                                   \
        prof_t = now_;                                                     \
    } while (0)
#else
#define PP_TICK(SLOT) do {} while (0)
#endif

template <int C_TYPE>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [RST][A 16K | W 16K] | bias[n_pad]
    constexpr int MT = 8;
    constexpr int ES = (C_TYPE == AMTX_T_BF16 ? 2 : 4);
    constexpr int NSTORE = MT * (C_TYPE == AMTX_T_BF16 ? 2 : 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int grp = blockIdx.z;
    const unsigned nbn = g.n_pad / RTB;
    const int nk = g.k_pad / RBK;

    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)grp * g.a_gs;
    const bf16_t* Wbase = g.W + (int64_t)grp * g.w_gs;
    char* Cbase = reinterpret_cast<char*>(g.C) + (int64_t)grp * g.c_gs * ES;
    float* bias_l = reinterpret_cast<float*>(smem + RST * RSTAGE);
    for (int i = tid; i < g.n_pad; i += 512) bias_l[i] = (g.bias && i < g.N) ? g.bias[(int64_t)grp * g.bias_gs + i] : 0.f;

    const bf16_t* a_src[2];
    const bf16_t* w_src[2];
#define PP_SET_TILE(TILE_ID)                                                                                  \
    do {                                                                                                      \
        const unsigned lg = xcd_remap((unsigned)(TILE_ID), (unsigned)ntiles);                                 \
        const int tn0 = (lg % nbn) * RTB;                                                                     \
        const int64_t tm0 = (int64_t)(lg / nbn) * RTB;                                                        \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                       \
            const int row = wave * 32 + n * 16 + (lane >> 2);                                                 \
            int64_t mr = tm0 + row;                                                                           \
            if (mr >= g.M) mr = g.M - 1;                                                                      \
            a_src[n] = Abase + mr * g.lda + ((lane & 3) ^ rswz((row >> 2) & 3)) * 8;                          \
            w_src[n] = Wbase + (int64_t)(tn0 + row) * g.k_pad + ((lane & 3) ^ rswz((row >> 4) & 3)) * 8;      \
        }                                                                                                     \
    } while (0)
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + wave * 2048;

    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * nk;
    int is = 0, ikt = 0, itile = blockIdx.x;
    bool pending_w = false;
// The DMA of a stage is issued in two halves so that both load phases carry the same VMEM work (the texture path moves 64 B/clk:
// four 1-KiB instructions per wave in one phase made that phase longer than the other group's 16 MFMAs): the A half of stage X
// in L2 of stage X-3, the W half in L1 of stage X-2.
#define PP_ISSUE_A()                                                                                          \
    do {                                                                                                      \
        if (is < S) {                                                                                         \
            const unsigned sb = lds_base + (is & (RST - 1)) * RSTAGE;                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) glds16(a_src[n] + ikt * RBK, sb + n * 1024);        \
            pending_w = true;                                                                                 \
        }                                                                                                     \
    } while (0)
#define PP_ISSUE_W()                                                                                          \
    do {                                                                                                      \
        if (pending_w) {                                                                                      \
            const unsigned sb = lds_base + (is & (RST - 1)) * RSTAGE;                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) glds16(w_src[n] + ikt * RBK, sb + ROP + n * 1024);  \
            pending_w = false;                                                                                \
            ++is;                                                                                             \
            if (++ikt == nk) {                                                                                \
                ikt = 0;                                                                                      \
                itile += (int)gridDim.x;                                                                      \
                if (is < S) PP_SET_TILE(itile);                                                               \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#define PP_BARRIER()                                                                                          \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        asm volatile("s_barrier" ::: "memory");                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)

    PP_SET_TILE(itile);
    PP_ISSUE_A(); PP_ISSUE_W();
    PP_ISSUE_A(); PP_ISSUE_W();
    PP_ISSUE_A();
    // stage 0 landed everywhere (stage 1 and the A half of stage 2 stay in flight) and bias_l visible
    if (S >= 3) wait_vm<6>(); else if (S == 2) wait_vm<4>(); else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) PP_BARRIER();                            // group 1 starts one interval late

    const int frow = lane & 15, fchunk = lane >> 4;
    const int a_off = (wm * 128 + frow) * 64 + ((fchunk ^ rswz((frow >> 2) & 3)) << 4);
    const int w_off = ROP + (wn * 64 + 16 * (frow >> 2) + (frow & 3)) * 64 + ((fchunk ^ rswz(frow >> 2)) << 4);

    f32x4_t acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#ifdef AMTX_GEMM_TIMING
    unsigned long long prof_acc[14] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif
    int ckt = 0, ctile = blockIdx.x;
    bool stores_recent = false;                           // the previous stage ended with an epilogue: its C stores are younger than
                                                          // stage cs+1's DMA but older than the W half of stage cs+2
    for (int cs = 0; cs < S; ++cs) {
        const char* b = smem + (cs & (RST - 1)) * RSTAGE;
        uint4 af[4], wf[4];
        // ---- phase 1
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const uint4*>(b + w_off + t * 256);
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const uint4*>(b + a_off + t * 1024);
        __builtin_amdgcn_sched_barrier(0);
        PP_ISSUE_W();                                     // W half of stage cs+2
        PP_TICK(0);
        PP_BARRIER();
        PP_TICK(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_TICK(2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mfma16(wf[nt], af[mt], acc[nt][mt]);
        __builtin_amdgcn_s_setprio(0);
        PP_TICK(3);
        PP_BARRIER();
        PP_TICK(4);
        // ---- phase 2
        {
            const int newer = min(S - 2 - cs, 1);         // stages newer than cs+1 already issued (cs+2), if they exist
            if (newer >= 0) {                             // stage cs+1 exists
                // issue order: ... W(cs+1) | A(cs+2) | [stores of stage cs-1's epilogue] | W(cs+2) | <- now
                if (!stores_recent) { if (newer == 1) wait_vm<4>(); else wait_vm<0>(); }
                else { if (newer == 1) wait_vm<4 + NSTORE>(); else wait_vm<NSTORE>(); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_TICK(5);
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const uint4*>(b + a_off + (4 + t) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        PP_ISSUE_A();                                     // A half of stage cs+3 -> the buffer stage cs-1 vacated two intervals ago
        stores_recent = false;
        PP_TICK(6);
        PP_BARRIER();
        PP_TICK(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_TICK(8);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][4 + mt] = mfma16(wf[nt], af[mt], acc[nt][4 + mt]);
        __builtin_amdgcn_s_setprio(0);
        PP_TICK(9);

        if (++ckt == nk) {
            // ---- epilogue: lane (g, m) holds columns nb .. nb+15 of row m; exactly NSTORE store instructions
            const unsigned lg = xcd_remap((unsigned)ctile, (unsigned)ntiles);
            const int n0 = (lg % nbn) * RTB;
            const int64_t m0 = (int64_t)(lg / nbn) * RTB;
            const int nb = n0 + wn * 64 + 16 * (lane >> 4);
            f32x4_t bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const f32x4_t*>(bias_l + nb + 4 * nt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int64_t m = m0 + wm * 128 + mt * 16 + (lane & 15);
#ifdef AMTX_PP_NOSTORE
                char* dst = reinterpret_cast<char*>(g_gemm_trash);
#else
                char* dst = m < g.M ? Cbase + (m * g.ldc + nb) * ES : reinterpret_cast<char*>(g_gemm_trash);
#endif
                float o[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[nt][r] = acc[nt][mt][r] + bv[nt][r];
                if (C_TYPE == AMTX_T_F32) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) reinterpret_cast<float4*>(dst)[nt] = make_float4(o[nt][0], o[nt][1], o[nt][2], o[nt][3]);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        reinterpret_cast<uint4*>(dst)[h] = make_uint4(pack_bf16x2(o[2 * h][0], o[2 * h][1]), pack_bf16x2(o[2 * h][2], o[2 * h][3]),
                                                                     pack_bf16x2(o[2 * h + 1][0], o[2 * h + 1][1]), pack_bf16x2(o[2 * h + 1][2], o[2 * h + 1][3]));
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
            ckt = 0;
            ctile += (int)gridDim.x;
            stores_recent = true;
            PP_TICK(13);
        }
        PP_TICK(10);
        PP_BARRIER();
        PP_TICK(11);
    }
#ifdef AMTX_GEMM_TIMING
    if (lane == 0 && (wave & 3) == 0) {
        for (int i = 0; i < 12; ++i) atomicAdd(&g_gemm_prof[wm][i], prof_acc[i]);
        atomicAdd(&g_gemm_prof[wm][13], prof_acc[13]);
        atomicAdd(&g_gemm_prof[wm][12], (unsigned long long)S);
    }
#endif
    if (wm == 0) PP_BARRIER();                            // group 0 meets group 1's last barrier
#undef PP_SET_TILE
#undef PP_ISSUE_A
#undef PP_ISSUE_W
#undef PP_BARRIER
}

// ------------------------------------------------------------------------------------------------
// Two-plane ("x3") problems whose A operand already IS two 16-bit planes (AMTX_T_SPLIT: the producing kernel's epilogue wrote
// hi = 16-bit(x), lo = 16-bit(x - hi)): all four operand planes travel HBM / L2 -> LDS by DMA, nothing is converted here.  The fp32-A
// path above (gemm_tile) spends ~60 conversion + 16 LDS-store instructions per thread and 32-deep step next to its 48 MFMAs and holds
// a 64 x 64 wave tile; this one is the direct-to-LDS kernel's structure (256 x 256 tile, eight waves of 128 x 64, two buffers, one barrier
// per stage) with 32-deep stages of 64 KiB = [A hi | A lo | W hi | W lo] x 256 rows x 64 bytes, the ring kernel's row swizzles, and
// three MFMAs per fragment pair: 96 MFMAs per wave and stage against 24 fragment reads and 8 DMA instructions -- three times the matrix
// work per staged byte of the one-plane kernels.  Same product order (hi.hi, hi.lo, lo.hi per 32-deep step, k ascending) as gemm_tile:
// the two kernels return the same bits for the same planes.
// C_TYPE: AMTX_T_F32 or AMTX_T_SPLIT (the next GEMM's A).
constexpr int SPL = 256 * RBK * 2;          // bytes per operand plane and stage (16 KiB)
constexpr int SSTAGE = 4 * SPL;

static __device__ __forceinline__ void glds16x2(const void* g0, const void* g1, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    const char* p1 = static_cast<const char*>(g1) - 1024;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(p1), "s"(lds_addr)
                 : "memory");
}

template <int C_TYPE>
__global__ __launch_bounds__(512) void gemm_split_kernel(GemmArgs g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A hi | A lo | W hi | W lo]
    constexpr int MT = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int grp = blockIdx.z;
    const unsigned nbn = g.n_pad / RTB;
    const int nk = g.k_pad / RBK;

    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)grp * g.a_gs;
    const bf16_t* Wbase = g.W + (int64_t)grp * g.w_gs;
    const int64_t w_plane = (int64_t)g.n_pad * g.k_pad;
    const float* bias = g.bias ? g.bias + (int64_t)grp * g.bias_gs : nullptr;

    // this wave DMAs rows [32 wave, 32 wave + 32) of every plane: two 1-KiB pieces (16 rows x 4 chunks) each
    const bf16_t* a_src[2];
    const bf16_t* w_src[2];
#define SP_SET_TILE(TILE_ID)                                                                                  \
    do {                                                                                                      \
        const unsigned lg = xcd_remap((unsigned)(TILE_ID), (unsigned)ntiles);                                 \
        const int tn0 = (lg % nbn) * RTB;                                                                     \
        const int64_t tm0 = (int64_t)(lg / nbn) * RTB;                                                        \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                       \
            const int row = wave * 32 + n * 16 + (lane >> 2);                                                 \
            int64_t mr = tm0 + row;                                                                           \
            if (mr >= g.M) mr = g.M - 1;                                                                      \
            a_src[n] = Abase + mr * g.lda + ((lane & 3) ^ rswz((row >> 2) & 3)) * 8;                          \
            w_src[n] = Wbase + (int64_t)(tn0 + row) * g.k_pad + ((lane & 3) ^ rswz((row >> 4) & 3)) * 8;      \
        }                                                                                                     \
    } while (0)
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + wave * 2048;
#define SP_ISSUE(k0, buf)                                                                                     \
    do {                                                                                                      \
        const unsigned sb = lds_base + (buf) * SSTAGE;                                                        \
        glds16x2(a_src[0] + (k0), a_src[1] + (k0), sb);                                                       \
        glds16x2(a_src[0] + g.a_split + (k0), a_src[1] + g.a_split + (k0), sb + SPL);                         \
        glds16x2(w_src[0] + (k0), w_src[1] + (k0), sb + 2 * SPL);                                             \
        glds16x2(w_src[0] + w_plane + (k0), w_src[1] + w_plane + (k0), sb + 3 * SPL);                         \
    } while (0)
    const int frow = lane & 15, fchunk = lane >> 4;
    const int a_off = (wm * 128 + frow) * 64 + ((fchunk ^ rswz((frow >> 2) & 3)) << 4);
    const int w_off = 2 * SPL + (wn * 64 + 16 * (frow >> 2) + (frow & 3)) * 64 + ((fchunk ^ rswz(frow >> 2)) << 4);
    // Per half (four 16-row tiles of A): the three products as three passes over the 16 accumulators, so that consecutive MFMAs never
    // share an accumulator; per accumulator the order stays hi.hi, hi.lo, lo.hi.
#define SP_COMPUTE(buf)                                                                                       \
    do {                                                                                                      \
        const char* b = smem + (buf) * SSTAGE;                                                                \
        uint4 wh[4], wl[4];                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                       \
            wh[t] = *reinterpret_cast<const uint4*>(b + w_off + t * 256);                                     \
            wl[t] = *reinterpret_cast<const uint4*>(b + w_off + SPL + t * 256);                               \
        }                                                                                                     \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                       \
            uint4 ah[4], al[4];                                                                               \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
                ah[t] = *reinterpret_cast<const uint4*>(b + a_off + (4 * h + t) * 1024);                      \
                al[t] = *reinterpret_cast<const uint4*>(b + a_off + SPL + (4 * h + t) * 1024);                \
            }                                                                                                 \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                  \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[nt][4 * h + mt] = mfma16(wh[nt], ah[mt], acc[nt][4 * h + mt]); \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                  \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[nt][4 * h + mt] = mfma16(wh[nt], al[mt], acc[nt][4 * h + mt]); \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                  \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[nt][4 * h + mt] = mfma16(wl[nt], ah[mt], acc[nt][4 * h + mt]); \
        }                                                                                                     \
    } while (0)
#define SP_SYNC()                                                                                             \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        __syncthreads();                                                                                      \
    } while (0)

    int tile = blockIdx.x;
    SP_SET_TILE(tile);
    SP_ISSUE(0, 0);
    SP_SYNC();
    int cur = 0;
    for (;;) {
        const unsigned lg = xcd_remap((unsigned)tile, (unsigned)ntiles);
        const int n0 = (lg % nbn) * RTB;
        const int64_t m0 = (int64_t)(lg / nbn) * RTB;
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;

        f32x4_t acc[4][MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt + 1 < nk; ++kt) {
            SP_ISSUE((kt + 1) * RBK, cur ^ 1);
            SP_COMPUTE(cur);
            SP_SYNC();
            cur ^= 1;
        }
        if (has_next) {
            SP_SET_TILE(next);
            SP_ISSUE(0, cur ^ 1);
        }
        SP_COMPUTE(cur);

        // ---- epilogue (the next tile's first stage is in flight underneath): lane (g, m) holds columns nb .. nb + 15 of row m
        {
            const int nb = n0 + wn * 64 + 16 * (lane >> 4);
            float bv[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[nt][r] = (bias && nb + 4 * nt < g.N) ? bias[nb + 4 * nt + r] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int64_t m = m0 + wm * 128 + mt * 16 + (lane & 15);
                if (m >= g.M || nb + 16 > g.N) continue;
                float o[16];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[4 * nt + r] = acc[nt][mt][r] + bv[nt][r];
                if (C_TYPE == AMTX_T_F32) {
                    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (int64_t)grp * g.c_gs + m * g.ldc + nb);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) dst[nt] = make_float4(o[4 * nt], o[4 * nt + 1], o[4 * nt + 2], o[4 * nt + 3]);
                } else {
                    bf16_t* dh = reinterpret_cast<bf16_t*>(g.C) + (int64_t)grp * g.c_gs + m * g.ldc + nb;
                    uint32_t hh[8], ll[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) split_bf16x2(o[2 * i], o[2 * i + 1], hh[i], ll[i]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        reinterpret_cast<uint4*>(dh)[q] = make_uint4(hh[4 * q], hh[4 * q + 1], hh[4 * q + 2], hh[4 * q + 3]);
                        reinterpret_cast<uint4*>(dh + g.c_split)[q] = make_uint4(ll[4 * q], ll[4 * q + 1], ll[4 * q + 2], ll[4 * q + 3]);
                    }
                }
            }
        }
        if (!has_next) break;
        SP_SYNC();
        cur ^= 1;
        tile = next;
    }
#undef SP_SET_TILE
#undef SP_ISSUE
#undef SP_COMPUTE
#undef SP_SYNC
}

template <int C_TYPE>
int launch_split(const GemmArgs& g, hipStream_t stream) {
    const int64_t ntiles = ((g.M + RTB - 1) / RTB) * (g.n_pad / RTB);
    AMTX_REQUIRE(ntiles < (1ll << 31), "gemm: too many output tiles");
    const size_t lds = 2 * (size_t)SSTAGE;
    auto kern = gemm_split_kernel<C_TYPE>;
    AMTX_GRANT_LDS(kern, lds);
    int64_t gx = 256 / std::max(1, g.groups);
    gx = std::max<int64_t>(8, gx / 8 * 8);
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1, (unsigned)g.groups), dim3(512), lds, stream, g, (int)ntiles);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int C_TYPE>
int launch_pp(const GemmArgs& g, hipStream_t stream) {
    const int64_t ntiles = ((g.M + RTB - 1) / RTB) * (g.n_pad / RTB);
    AMTX_REQUIRE(ntiles < (1ll << 31), "gemm: too many output tiles");
    const size_t lds = (size_t)RST * RSTAGE + (size_t)g.n_pad * sizeof(float);
    auto kern = gemm_pp_kernel<C_TYPE>;
    AMTX_GRANT_LDS(kern, RST * RSTAGE + RING_MAX_NPAD * sizeof(float));
    int64_t gx = 256 / std::max(1, g.groups);
    gx = std::max<int64_t>(8, gx / 8 * 8);
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1, (unsigned)g.groups), dim3(512), lds, stream, g, (int)ntiles);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Skinny N (n_pad == 128: the LogisticBanks and the folded pitch head, N = 88): these products are bound by streaming A from HBM -- the pitch
// head reads its whole conv3 map, 4.67 GB per 1024 clips -- and gemm_glds_kernel<., 128> keeps 32 KB of A in flight per CU (two blocks x one
// 16 KB stage), which at the loaded latency is the 4.7 - 4.9 TB/s it reaches; a kernel that only reads streams 6.1 TB/s on this chip with
// enough loads in flight (tools/hbm_read.hip).  Here ONE 512-thread block per CU owns 256 rows x all 128 columns: every wave DMAs the A
// rows of ITS 32-row slice into a private three-stage LDS ring (no barrier for A: a wave reads what it fetched itself) and its share of the W
// k-tile into a three-stage ring shared by the block (one barrier per k-tile), both TWO k-tiles ahead: 64 KB of A in flight per CU.  The
// stage stream runs across the block's row tiles, so the rings never drain; every iteration issues the same six DMA instructions per wave
// (past the end: a valid address again), which is what the counted wait relies on.  Same fragments, same k order, same accumulator layout and
// epilogue as gemm_glds_kernel: the same bits (AMTX_GEMM_NO_SKINNY=1 keeps that kernel).
constexpr int SKW = 128 * GBK * 2;            // bytes of a W stage (128 rows x 64 k)
constexpr int SKA = 32 * GBK * 2;             // bytes of a wave's A stage (32 rows x 64 k)
constexpr int SKD = 3;                        // ring depth of both
constexpr size_t SK_LDS = (size_t)SKD * SKW + 8 * (size_t)SKD * SKA + 128 * sizeof(float);

__global__ __launch_bounds__(512, 1) void gemm_skinny_kernel(GemmArgs g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [W ring][8 waves x A ring][bias]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.z;
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)grp * g.a_gs;
    const bf16_t* Wbase = g.W + (int64_t)grp * g.w_gs;
    char* Cbase = reinterpret_cast<char*>(g.C) + (int64_t)grp * g.c_gs * 4;
    float* bias_s = reinterpret_cast<float*>(smem + SKD * SKW + 8 * SKD * SKA);
    if (tid < 128) bias_s[tid] = (g.bias && tid < g.N) ? g.bias[(int64_t)grp * g.bias_gs + tid] : 0.f;     // (visible after the first stage's barrier)
    const int64_t a_pitch = g.a_plane ? GBK : g.lda, a_kstep = g.a_plane ? g.a_plane : GBK;
    const int nk = g.k_pad / GBK;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int stride = (int)gridDim.x;
    const int nmine = (ntiles - (int)blockIdx.x + stride - 1) / stride;       // row tiles of this block
    const int total = nmine * nk;                                             // its stages

    // W: this wave DMAs rows [16 wave, + 16) of a k-tile, 2 instructions x 8 rows
    const bf16_t* w_src[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = (wave * 2 + n) * 8 + (lane >> 3);
        w_src[n] = Wbase + (int64_t)row * g.k_pad + ((lane & 7) ^ wswz(row)) * 8;
    }
    const unsigned w_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + wave * 2048;
    const unsigned a_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + SKD * SKW + wave * (SKD * SKA);
    // A: 4 instructions x 8 rows of the wave's 32; row pointers of the row tile the prefetch stream is in
    const bf16_t* a_src[4];
    auto set_rows = [&](int j) {                                  // j-th row tile of this block (past the end: the last one again)
        const int64_t tm0 = (int64_t)((int)blockIdx.x + min(j, nmine - 1) * stride) * 256 + wave * 32;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = n * 8 + (lane >> 3);
            const int64_t mr = min(tm0 + row, g.M - 1);           // rows past M are never stored; keep the read in bounds
            a_src[n] = Abase + mr * a_pitch + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        }
    };
    int pj = 0, pkt = 0;                                          // the prefetch stream's row tile and k-tile
    auto issue = [&](int slot) {
        const int64_t a_off = (int64_t)pkt * a_kstep;
        glds16x4(a_src[0] + a_off, a_src[1] + a_off, a_src[2] + a_off, a_src[3] + a_off, a_lds + slot * SKA);
        glds16x2(w_src[0] + pkt * GBK, w_src[1] + pkt * GBK, w_lds + slot * SKW);
        if (++pkt == nk) { pkt = 0; ++pj; set_rows(pj); }
    };
    set_rows(0);
    issue(0);
    issue(1);

    f32x4_t acc[2][4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    int cj = 0, ckt = 0;                                          // the compute stream's row tile and k-tile
    for (int base = 0; base < total; base += SKD) {
#pragma unroll
        for (int u = 0; u < SKD; ++u) {
            if (base + u >= total) break;                         // (block-uniform)
            // stage base + u was issued two iterations ago; behind it in this wave's queue: one iteration's six instructions (+ stores)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            lds_only_barrier();                                   // every wave's share of the W stage has landed; last iteration's reads are done
                                                                  // (not __syncthreads: its fence would wait for the epilogue's stores AND the ring)
            issue((u + 2) % SKD);                                 // stage base + u + 2 -> the slot stage base + u - 1 was read from
            const char* wb = smem + u * SKW;
            const char* ab = smem + SKD * SKW + wave * (SKD * SKA) + u * SKA;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 af[2], wf[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ar = t * 16 + frow;
                    af[t] = *reinterpret_cast<const uint4*>(ab + ar * 128 + (((kk * 4 + fchunk) ^ ((ar >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int wr = h * 64 + 16 * (frow >> 2) + 4 * t + (frow & 3);   // see gemm_glds_kernel's epilogue
                        wf[h][t] = *reinterpret_cast<const uint4*>(wb + wr * 128 + (((kk * 4 + fchunk) ^ wswz(wr)) << 4));
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[h][nt][mt] = mfma16(wf[h][nt], af[mt], acc[h][nt][mt]);
            }
            if (++ckt < nk) continue;
            // ---- a row tile is complete: epilogue (gemm_glds_kernel's, for this wave's 32 rows and both 64-column halves)
            const int64_t m0 = (int64_t)((int)blockIdx.x + cj * stride) * 256 + wave * 32;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int nb = h * 64 + 16 * (lane >> 4);
                if (nb >= g.N + (g.copy16 ? g.copy16_pad : 0)) continue;
                float bv[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[nt][r] = bias_s[min(nb + 4 * nt + r, 127)];
                const bool wide = nb + 16 <= g.N && ((g.ldc * 4) % 16) == 0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int64_t m = m0 + mt * 16 + (lane & 15);
                    if (m >= g.M) continue;
                    char* dst = Cbase + (m * g.ldc + nb) * 4;
                    float o[4][4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[nt][r] = acc[h][nt][mt][r] + bv[nt][r];
                    if (g.C) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            if (wide || nb + 4 * nt < g.N) reinterpret_cast<float4*>(dst)[nt] = make_float4(o[nt][0], o[nt][1], o[nt][2], o[nt][3]);
                    }
                    if (g.copy16) {
                        bf16_t* cp = g.copy16 + m * g.copy16_ld + g.copy16_col0 + (int64_t)grp * g.copy16_gs + nb;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int n4 = nb + 4 * nt;
                            if (n4 < g.N) reinterpret_cast<uint2*>(cp)[nt] = make_uint2(pack_bf16x2(o[nt][0], o[nt][1]), pack_bf16x2(o[nt][2], o[nt][3]));
                            else if (n4 < g.N + g.copy16_pad) reinterpret_cast<uint2*>(cp)[nt] = make_uint2(0u, 0u);
                        }
                    }
                    if (g.roll_out && grp == g.roll_group) {
                        const unsigned bclip = (unsigned)m / (unsigned)g.roll_T, tfrm = (unsigned)m - bclip * (unsigned)g.roll_T;
                        float* ro = g.roll_out + ((int64_t)bclip * g.N + nb) * g.roll_T + tfrm;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (nb + 4 * nt + r < g.N) {
                                    const float sg = 1.0f / (1.0f + expf(-o[nt][r]));
                                    ro[(int64_t)(4 * nt + r) * g.roll_T] = g.roll_thr < 0.f ? sg : (sg < g.roll_thr ? 0.f : 1.f);
                                }
                    }
                }
            }
            zero_acc();
            ckt = 0;
            ++cj;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the two stages issued past the end
}

// The same for two-plane operands (x3: A in AMTX_T_SPLIT planes, two-plane weights): 32-deep stages in gemm_split_kernel's layout ([hi | lo] x rows
// x 64 bytes, its chunk swizzles), the same six DMA instructions per wave and stage (A: 2 planes x 2 pieces of 16 rows, W: 16 rows of each
// plane), three MFMAs per fragment pair in gemm_tile's order (hi.hi, hi.lo, lo.hi): the bits of the generic kernel these shapes ran on.
__global__ __launch_bounds__(512, 1) void gemm_skinny_split_kernel(GemmArgs g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [W ring][8 waves x A ring][bias]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.z;
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)grp * g.a_gs;      // hi plane; lo plane a_split elements further
    const bf16_t* Wbase = g.W + (int64_t)grp * g.w_gs;
    char* Cbase = reinterpret_cast<char*>(g.C) + (int64_t)grp * g.c_gs * 4;
    float* bias_s = reinterpret_cast<float*>(smem + SKD * SKW + 8 * SKD * SKA);
    if (tid < 128) bias_s[tid] = (g.bias && tid < g.N) ? g.bias[(int64_t)grp * g.bias_gs + tid] : 0.f;     // (visible after the first stage's barrier)
    const int64_t w_plane = (int64_t)g.n_pad * g.k_pad;
    const int nk = g.k_pad / RBK;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int stride = (int)gridDim.x;
    const int nmine = (ntiles - (int)blockIdx.x + stride - 1) / stride;       // row tiles of this block
    const int total = nmine * nk;                                             // its stages

    // W: this wave DMAs rows [16 wave, + 16) of both planes of a stage: one piece of 16 rows x 4 chunks each
    const bf16_t* w_src;
    {
        const int row = wave * 16 + (lane >> 2);
        w_src = Wbase + (int64_t)row * g.k_pad + ((lane & 3) ^ rswz((row >> 4) & 3)) * 8;
    }
    const unsigned w_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + wave * 1024;
    const unsigned a_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem) + SKD * SKW + wave * (SKD * SKA);
    // A: per plane 2 pieces x 16 rows of the wave's 32; row pointers of the row tile the prefetch stream is in
    const bf16_t* a_src[2];
    auto set_rows = [&](int j) {                                  // j-th row tile of this block (past the end: the last one again)
        const int64_t tm0 = (int64_t)((int)blockIdx.x + min(j, nmine - 1) * stride) * 256 + wave * 32;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int row = n * 16 + (lane >> 2);
            const int64_t mr = min(tm0 + row, g.M - 1);           // rows past M are never stored; keep the read in bounds
            a_src[n] = Abase + mr * g.lda + ((lane & 3) ^ rswz((row >> 2) & 3)) * 8;
        }
    };
    int pj = 0, pkt = 0;                                          // the prefetch stream's row tile and k-stage
    auto issue = [&](int slot) {
        const int k0 = pkt * RBK;
        glds16x2(a_src[0] + k0, a_src[1] + k0, a_lds + slot * SKA);
        glds16x2(a_src[0] + g.a_split + k0, a_src[1] + g.a_split + k0, a_lds + slot * SKA + SKA / 2);
        glds16(w_src + k0, w_lds + slot * SKW);
        glds16(w_src + w_plane + k0, w_lds + slot * SKW + SKW / 2);
        if (++pkt == nk) { pkt = 0; ++pj; set_rows(pj); }
    };
    set_rows(0);
    issue(0);
    issue(1);

    f32x4_t acc[2][4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    int cj = 0, ckt = 0;                                          // the compute stream's row tile and k-tile
    for (int base = 0; base < total; base += SKD) {
#pragma unroll
        for (int u = 0; u < SKD; ++u) {
            if (base + u >= total) break;                         // (block-uniform)
            // stage base + u was issued two iterations ago; behind it in this wave's queue: one iteration's six instructions (+ stores)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            lds_only_barrier();                                   // every wave's share of the W stage has landed; last iteration's reads are done
                                                                  // (not __syncthreads: its fence would wait for the epilogue's stores AND the ring)
            issue((u + 2) % SKD);                                 // stage base + u + 2 -> the slot stage base + u - 1 was read from
            const char* wb = smem + u * SKW;
            const char* ab = smem + SKD * SKW + wave * (SKD * SKA) + u * SKA;
            {
                uint4 ah[2], al[2], wh[2][4], wl[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int o = (t * 16 + frow) * 64 + ((fchunk ^ rswz((frow >> 2) & 3)) << 4);
                    ah[t] = *reinterpret_cast<const uint4*>(ab + o);
                    al[t] = *reinterpret_cast<const uint4*>(ab + SKA / 2 + o);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int o = (h * 64 + 16 * (frow >> 2) + 4 * t + (frow & 3)) * 64 + ((fchunk ^ rswz(frow >> 2)) << 4);
                        wh[h][t] = *reinterpret_cast<const uint4*>(wb + o);
                        wl[h][t] = *reinterpret_cast<const uint4*>(wb + SKW / 2 + o);
                    }
                // the three products as three passes over the 16 accumulators (consecutive MFMAs never share one); per accumulator hi.hi, hi.lo, lo.hi
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[h][nt][mt] = mfma16(wh[h][nt], ah[mt], acc[h][nt][mt]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[h][nt][mt] = mfma16(wh[h][nt], al[mt], acc[h][nt][mt]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[h][nt][mt] = mfma16(wl[h][nt], ah[mt], acc[h][nt][mt]);
            }
            if (++ckt < nk) continue;
            // ---- a row tile is complete: epilogue (gemm_glds_kernel's, for this wave's 32 rows and both 64-column halves)
            const int64_t m0 = (int64_t)((int)blockIdx.x + cj * stride) * 256 + wave * 32;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int nb = h * 64 + 16 * (lane >> 4);
                if (nb >= g.N + (g.copy16 ? g.copy16_pad : 0)) continue;
                float bv[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[nt][r] = bias_s[min(nb + 4 * nt + r, 127)];
                const bool wide = nb + 16 <= g.N && ((g.ldc * 4) % 16) == 0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int64_t m = m0 + mt * 16 + (lane & 15);
                    if (m >= g.M) continue;
                    char* dst = Cbase + (m * g.ldc + nb) * 4;
                    float o[4][4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[nt][r] = acc[h][nt][mt][r] + bv[nt][r];
                    if (g.C) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            if (wide || nb + 4 * nt < g.N) reinterpret_cast<float4*>(dst)[nt] = make_float4(o[nt][0], o[nt][1], o[nt][2], o[nt][3]);
                    }
                    if (g.copy16) {
                        bf16_t* cp = g.copy16 + m * g.copy16_ld + g.copy16_col0 + (int64_t)grp * g.copy16_gs + nb;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int n4 = nb + 4 * nt;
                            if (n4 < g.N) reinterpret_cast<uint2*>(cp)[nt] = make_uint2(pack_bf16x2(o[nt][0], o[nt][1]), pack_bf16x2(o[nt][2], o[nt][3]));
                            else if (n4 < g.N + g.copy16_pad) reinterpret_cast<uint2*>(cp)[nt] = make_uint2(0u, 0u);
                        }
                    }
                    if (g.roll_out && grp == g.roll_group) {
                        const unsigned bclip = (unsigned)m / (unsigned)g.roll_T, tfrm = (unsigned)m - bclip * (unsigned)g.roll_T;
                        float* ro = g.roll_out + ((int64_t)bclip * g.N + nb) * g.roll_T + tfrm;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (nb + 4 * nt + r < g.N) {
                                    const float sg = 1.0f / (1.0f + expf(-o[nt][r]));
                                    ro[(int64_t)(4 * nt + r) * g.roll_T] = g.roll_thr < 0.f ? sg : (sg < g.roll_thr ? 0.f : 1.f);
                                }
                    }
                }
            }
            zero_acc();
            ckt = 0;
            ++cj;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the two stages issued past the end
}

int launch_skinny(const GemmArgs& g, hipStream_t stream, bool split) {
    const int64_t ntiles = (g.M + 255) / 256;
    AMTX_REQUIRE(ntiles < (1ll << 31), "gemm: too many output tiles");
    int64_t gx = std::max<int64_t>(1, 256 / std::max(1, g.groups));
    if (gx > ntiles) gx = ntiles;
    if (split) {
        AMTX_GRANT_LDS(gemm_skinny_split_kernel, SK_LDS);
        hipLaunchKernelGGL(gemm_skinny_split_kernel, dim3((unsigned)gx, 1, (unsigned)g.groups), dim3(512), SK_LDS, stream, g, (int)ntiles);
    } else {
        AMTX_GRANT_LDS(gemm_skinny_kernel, SK_LDS);
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3((unsigned)gx, 1, (unsigned)g.groups), dim3(512), SK_LDS, stream, g, (int)ntiles);
    }
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int C_TYPE, int TB>
int launch_glds(const GemmArgs& g, hipStream_t stream) {
    const int64_t ntiles = ((g.M + TB - 1) / TB) * (g.n_pad / TB);
    AMTX_REQUIRE(ntiles < (1ll << 31), "gemm: too many output tiles");
    const size_t lds = 4 * (size_t)TB * GBK * 2;
    // persistent grid: every CU full (two 64 KiB blocks or one 128 KiB block), a multiple of 8 so that a block's tiles stay
    // on its XCD (tile ids advance by gridDim.x)
    int64_t gx = (TB == 128 ? 512 : 256) / std::max(1, g.groups);
    gx = std::max<int64_t>(8, gx / 8 * 8);
    if (gx > ntiles) gx = ntiles;
    auto kern = gemm_glds_kernel<C_TYPE, TB>;
    AMTX_GRANT_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1, (unsigned)g.groups), dim3(TB * 2), lds, stream, g, (int)ntiles);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int A_TYPE, int C_TYPE, int NS>
int launch(const GemmArgs& g, hipStream_t stream) {
    dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)(g.n_pad / BN), (unsigned)g.groups);
    const size_t lds = 2 * 2 * NS * TILE_BYTES;
    hipLaunchKernelGGL((gemm_kernel<A_TYPE, C_TYPE, NS>), grid, dim3(256), lds, stream, g);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

}  // namespace

#ifdef AMTX_GEMM_TIMING
extern "C" void amtxdbg_gemm_prof(unsigned long long* out, int reset) {
    unsigned long long h[2][16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_prof), sizeof(h));
    memcpy(out, h, sizeof(h));
    if (reset) { memset(h, 0, sizeof(h)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), h, sizeof(h)); }
}
#endif

void amtx_gemm_pack_dims(int N, int K, int* n_pad, int* k_pad) {
    *n_pad = ((N + BN - 1) / BN) * BN;
    *k_pad = ((K + GBK - 1) / GBK) * GBK;   // 64: a zero-padded K also fits the direct-to-LDS kernels' k-tile
}

void amtx_gemm_pack_host(const float* W, int64_t ldw, int N, int K, int planes, bf16_t* out) {
    int n_pad, k_pad;
    amtx_gemm_pack_dims(N, K, &n_pad, &k_pad);
    const size_t plane = (size_t)n_pad * k_pad;
    memset(out, 0, plane * planes * sizeof(bf16_t));
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float w = W[(int64_t)n * ldw + k];
            const bf16_t hi = f32_to_bf16_rn(w);
            out[(size_t)n * k_pad + k] = hi;
            if (planes == 2) out[plane + (size_t)n * k_pad + k] = f32_to_bf16_rn(w - bf16_to_f32(hi));
        }
}

static int check_generic(const GemmArgs& g) {
    AMTX_REQUIRE(g.A && g.W && (g.C || g.pair_map), "gemm: null pointer");
    AMTX_REQUIRE(!g.pair_map || (g.a_type == AMTX_T_F32 && g.c_type == AMTX_T_F32 && g.pair_out && g.N % 2 == 0), "gemm: the magnitude epilogue needs fp32 A / C, an output and an even N");
    AMTX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.groups > 0, "gemm: bad sizes");
    AMTX_REQUIRE(g.K % 8 == 0 && g.N % 4 == 0, "gemm: K must be a multiple of 8 and N of 4 (K=%d N=%d)", g.K, g.N);
    AMTX_REQUIRE(g.n_pad % BN == 0 && g.k_pad % BK == 0 && g.n_pad >= g.N && g.k_pad >= g.K, "gemm: bad packed dims");
    // bf16 A: 16-byte fragments straight from memory.  fp32 A goes through registers with dword-aligned wide loads: rows may start at any
    // 4-byte boundary (clips of an odd length in one buffer, strided rows of the CQT products)
    // bf16 / two-plane A: 16-byte fragments straight from memory.  fp32 A goes through registers; rows that start at ANY 4-byte boundary
    // (dword-aligned wide loads) are for the bounded strided-row problems only (a_valid_hi > 0: the CQT products straight from a caller's
    // audio) -- a plain fp32 A keeps 16-byte aligned rows (the relaxation had reached every caller)
    AMTX_REQUIRE(g.a_type == AMTX_T_F32 ? (g.a_valid_hi > 0 ? ((uintptr_t)g.A % 4) == 0 : (g.lda % 4 == 0 && ((uintptr_t)g.A % 16) == 0))
                                        : ((g.lda * 2) % 16 == 0 && ((uintptr_t)g.A % 16) == 0),
                 "gemm: A rows must be 16-byte aligned (4-byte aligned for a bounded fp32 A)");
    AMTX_REQUIRE(g.a_valid_hi == 0 || (g.a_type == AMTX_T_F32 && g.a_valid_lo >= 0 && g.a_valid_hi > g.a_valid_lo), "gemm: bad valid range of A");
    AMTX_REQUIRE(g.ldc % 4 == 0 && ((uintptr_t)g.C % 16) == 0, "gemm: C rows must be 16-byte aligned");
    AMTX_REQUIRE(g.planes == 1 || g.planes == 2, "gemm: planes must be 1 or 2");
    return AMTX_OK;
}

int amtx_launch_gemm_multi(const GemmArgs* gs, int n, hipStream_t stream) {
    AMTX_REQUIRE(gs && n >= 1 && n <= AMTX_GEMM_MULTI_MAX, "gemm_multi: 1 .. %d problems", AMTX_GEMM_MULTI_MAX);
    GemmMulti mm;
    unsigned gx = 0, gy = 0;
    for (int i = 0; i < n; ++i) {
        const GemmArgs& g = gs[i];
        int rc = check_generic(g);
        if (rc != AMTX_OK) return rc;
        AMTX_REQUIRE(g.a_type == AMTX_T_F32 && g.c_type == AMTX_T_F32 && g.planes == 2 && g.groups == gs[0].groups,
                     "gemm_multi: fp32 A / C, two weight planes and one group count for all problems");
        mm.p[i] = g;
        gx = std::max(gx, (unsigned)((g.M + BM - 1) / BM));
        gy = std::max(gy, (unsigned)(g.n_pad / BN));
    }
    mm.n = n;
    const size_t lds = 2 * 2 * 2 * TILE_BYTES;
    hipLaunchKernelGGL((gemm_multi_kernel<AMTX_T_F32, AMTX_T_F32, 2>), dim3(gx, gy, (unsigned)(n * gs[0].groups)), dim3(256), lds, stream, mm);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

// the shapes amtx_launch_gemm sends to gemm_glds_kernel<fp32 C> (not the two-group ring, whose epilogue has no roll code)
bool amtx_gemm_has_roll_epilogue(const GemmArgs& g) {
    return g.a_type == AMTX_T_BF16 && g.c_type == AMTX_T_F32 && g.planes == 1 && g.K % GBK == 0 && g.k_pad == g.K && (g.lda % 8) == 0 &&
           !(g.n_pad % 256 == 0 && g.N % 256 == 0 && g.M >= 256);
}

int amtx_launch_gemm(const GemmArgs& g, hipStream_t stream) {
    AMTX_REQUIRE(g.A && g.W, "gemm: null pointer");
    AMTX_REQUIRE(!g.pair_map || (g.a_type == AMTX_T_F32 && g.c_type == AMTX_T_F32 && g.pair_out && g.N % 2 == 0), "gemm: the magnitude epilogue needs fp32 A / C, an output and an even N");
    AMTX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.groups > 0, "gemm: bad sizes");
    AMTX_REQUIRE(g.K % 8 == 0 && g.N % 4 == 0, "gemm: K must be a multiple of 8 and N of 4 (K=%d N=%d)", g.K, g.N);
    AMTX_REQUIRE(g.n_pad % BN == 0 && g.k_pad % BK == 0 && g.n_pad >= g.N && g.k_pad >= g.K, "gemm: bad packed dims");
    // bf16 A: 16-byte fragments straight from memory.  fp32 A goes through registers with dword-aligned wide loads: rows may start at any
    // 4-byte boundary (clips of an odd length in one buffer, strided rows of the CQT products)
    // bf16 / two-plane A: 16-byte fragments straight from memory.  fp32 A goes through registers; rows that start at ANY 4-byte boundary
    // (dword-aligned wide loads) are for the bounded strided-row problems only (a_valid_hi > 0: the CQT products straight from a caller's
    // audio) -- a plain fp32 A keeps 16-byte aligned rows (the relaxation had reached every caller)
    AMTX_REQUIRE(g.a_type == AMTX_T_F32 ? (g.a_valid_hi > 0 ? ((uintptr_t)g.A % 4) == 0 : (g.lda % 4 == 0 && ((uintptr_t)g.A % 16) == 0))
                                        : ((g.lda * 2) % 16 == 0 && ((uintptr_t)g.A % 16) == 0),
                 "gemm: A rows must be 16-byte aligned (4-byte aligned for a bounded fp32 A)");
    AMTX_REQUIRE(g.a_valid_hi == 0 || (g.a_type == AMTX_T_F32 && g.a_valid_lo >= 0 && g.a_valid_hi > g.a_valid_lo), "gemm: bad valid range of A");
    AMTX_REQUIRE(g.ldc % 4 == 0 && ((uintptr_t)g.C % 16) == 0, "gemm: C rows must be 16-byte aligned");
    AMTX_REQUIRE(g.planes == 1 || g.planes == 2, "gemm: planes must be 1 or 2");
    AMTX_REQUIRE(g.a_type == AMTX_T_BF16 || g.a_type == AMTX_T_F32 || g.a_type == AMTX_T_SPLIT, "gemm: bad A type");
    AMTX_REQUIRE(!g.copy16 || (amtx_gemm_has_roll_epilogue(g) && g.copy16_pad % 4 == 0 && g.N + g.copy16_pad <= g.n_pad &&
                               (g.copy16_ld % 4) == 0 && ((g.copy16_col0 | g.copy16_gs) % 4) == 0 && ((uintptr_t)g.copy16 % 8) == 0),
                 "gemm: bad bf16-copy epilogue arguments");
    AMTX_REQUIRE(g.C || g.pair_map || g.roll_out || g.copy16, "gemm: no output");
    AMTX_REQUIRE(!g.roll_out || amtx_gemm_has_roll_epilogue(g), "gemm: the piano-roll epilogue exists on the bf16 direct-to-LDS path with fp32 C only");
    AMTX_REQUIRE(!g.roll_out || (g.roll_T > 0 && g.M % g.roll_T == 0 && g.M < (1ll << 31)), "gemm: piano-roll epilogue: M must be clips x frames");
    if (g.a_type == AMTX_T_BF16 && g.planes == 1 && g.K % GBK == 0 && g.k_pad == g.K && (g.lda % 8) == 0) {
        // skinny N: one 128-column tile, fp32 C -- the LogisticBanks and the pitch head (AMTX_GEMM_NO_SKINNY=1: the A/B switch)
        static const bool no_skinny = getenv("AMTX_GEMM_NO_SKINNY") != nullptr;
        if (!no_skinny && g.n_pad == 128 && g.c_type == AMTX_T_F32 && g.M >= 1024 && g.K >= 2 * GBK && g.M < (1ll << 31) * 256)
            return launch_skinny(g, stream, false);
        if (g.a_plane) {      // planar A: the two-buffer direct-to-LDS kernel only
            if (g.n_pad % 256 == 0 && g.N % 256 == 0 && g.M >= 256)
                return g.c_type == AMTX_T_BF16 ? launch_glds<AMTX_T_BF16, 256>(g, stream) : launch_glds<AMTX_T_F32, 256>(g, stream);
            return g.c_type == AMTX_T_BF16 ? launch_glds<AMTX_T_BF16, 128>(g, stream) : launch_glds<AMTX_T_F32, 128>(g, stream);
        }
        // A/B switches for tools/bench_gemm.py / tools/check_gemm_pp.py: AMTX_GEMM_PP=1 forces the two-group ring for any K,
        // AMTX_GEMM_NO_PP=1 disables it
        static const bool force_pp = getenv("AMTX_GEMM_PP") != nullptr, no_pp = getenv("AMTX_GEMM_NO_PP") != nullptr;
        // measured on MI355X (tools/bench_gemm.py, M = 320000, N = 1024): K = 512: 0.49 ms vs 0.53 (two 64-deep buffers);
        // K = 192: 0.255 vs 0.271 ms; K = 3648 (N = 512): 1.21 vs 1.20 ms.  (Round 5: 128 x 128 tiles, two blocks per CU, for the short K: 0.356 /
        // 0.611 ms -- worse.)
        if (!no_pp && (force_pp || g.K <= 1024) && g.n_pad % 256 == 0 && g.N % 256 == 0 && g.M >= 256 && g.n_pad <= RING_MAX_NPAD &&
            g.K >= 4 * RBK && (g.ldc * amtx_tsize(g.c_type)) % 16 == 0)
            return g.c_type == AMTX_T_BF16 ? launch_pp<AMTX_T_BF16>(g, stream) : launch_pp<AMTX_T_F32>(g, stream);
        if (g.n_pad % 256 == 0 && g.N % 256 == 0 && g.M >= 256)
            return g.c_type == AMTX_T_BF16 ? launch_glds<AMTX_T_BF16, 256>(g, stream) : launch_glds<AMTX_T_F32, 256>(g, stream);
        return g.c_type == AMTX_T_BF16 ? launch_glds<AMTX_T_BF16, 128>(g, stream) : launch_glds<AMTX_T_F32, 128>(g, stream);
    }
    AMTX_REQUIRE(!g.a_plane, "gemm: planar A exists on the bf16 direct-to-LDS path only");
    if (g.a_type == AMTX_T_SPLIT) {
        AMTX_REQUIRE(g.planes == 2 && g.a_split > 0 && (g.a_split % 8) == 0, "gemm: two-plane A needs two-plane weights and a plane stride that keeps 16-byte alignment");
        AMTX_REQUIRE(g.c_type != AMTX_T_SPLIT || (g.c_split > 0 && (g.c_split % 8) == 0), "gemm: two-plane C needs a plane stride");
        AMTX_REQUIRE(!g.pair_map && !g.roll_out && !g.copy16 && g.C, "gemm: two-plane A has the plain epilogues only");
        // the direct-to-LDS two-plane kernel: whole 256-column tiles, whole 32-deep stages of real A columns
        static const bool no_split_dma = getenv("AMTX_GEMM_NO_SPLIT_DMA") != nullptr;   // A/B switch: the generic kernel on the same planes
        // C rows are stored 16 bytes at a time: fp32 C needs ldc % 4 == 0, the 2-byte planes of a two-plane C need ldc % 8 == 0 and a plane
        // stride that keeps the second plane 16-byte aligned; anything else takes the generic kernel (8-byte stores)
        const bool c_aligned = g.c_type == AMTX_T_SPLIT ? ((g.ldc * 2) % 16 == 0 && (g.c_split % 8) == 0) : ((g.ldc * 4) % 16 == 0);
        if (!no_split_dma && g.n_pad % 256 == 0 && g.N % 256 == 0 && g.M >= 256 && g.K == g.k_pad && (g.lda % 8) == 0 && g.c_type != AMTX_T_BF16 &&
            c_aligned)
            return g.c_type == AMTX_T_SPLIT ? launch_split<AMTX_T_SPLIT>(g, stream) : launch_split<AMTX_T_F32>(g, stream);
        AMTX_REQUIRE(g.c_type != AMTX_T_BF16, "gemm: two-plane A writes fp32 or two-plane C");
        // skinny N (the x3 LogisticBanks and pitch head): the two-plane variant of gemm_skinny_kernel
        static const bool no_skinny2 = getenv("AMTX_GEMM_NO_SKINNY") != nullptr;
        if (!no_skinny2 && !no_split_dma && g.n_pad == 128 && g.c_type == AMTX_T_F32 && g.M >= 1024 && g.K == g.k_pad && g.K >= 2 * RBK && (g.lda % 8) == 0 &&
            g.M < (1ll << 31) * 256)
            return launch_skinny(g, stream, true);
        return g.c_type == AMTX_T_SPLIT ? launch<AMTX_T_SPLIT, AMTX_T_SPLIT, 2>(g, stream) : launch<AMTX_T_SPLIT, AMTX_T_F32, 2>(g, stream);
    }
    AMTX_REQUIRE(g.c_type != AMTX_T_SPLIT, "gemm: two-plane C needs two-plane A");
    const int key = (g.a_type << 2) | (g.c_type << 1) | (g.planes - 1);
    switch (key) {
        case 0: return launch<AMTX_T_BF16, AMTX_T_BF16, 1>(g, stream);
        case 1: return launch<AMTX_T_BF16, AMTX_T_BF16, 2>(g, stream);
        case 2: return launch<AMTX_T_BF16, AMTX_T_F32, 1>(g, stream);
        case 3: return launch<AMTX_T_BF16, AMTX_T_F32, 2>(g, stream);
        case 4: return launch<AMTX_T_F32, AMTX_T_BF16, 1>(g, stream);
        case 5: return launch<AMTX_T_F32, AMTX_T_BF16, 2>(g, stream);
        case 6: return launch<AMTX_T_F32, AMTX_T_F32, 1>(g, stream);
        case 7: return launch<AMTX_T_F32, AMTX_T_F32, 2>(g, stream);
    }
    amtx_set_error("gemm: unsupported type combination");
    return AMTX_ERR_UNSUPPORTED;
}
