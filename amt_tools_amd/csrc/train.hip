// Dense kernels of the TRAINING step for gfx950: nn.Linear and Conv2d(3x3, pad 1) forward, input gradient and weight gradient
// (amt_tools/train.py:126-141 drives them through autograd: AcousticModel amt_tools/models/onsetsframes.py:375-427, LogisticBank
// amt_tools/models/common.py:539, the nn.LSTM input projections onsetsframes.py:498-501).
//
// One tiled kernel, fp32 in / fp32 out:   C[m][n] = sum_k A(m, k) * B(n, k)  (+ bias[n])
// Every product runs on the matrix cores as split-bf16 ("x3": x = hi + lo, hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16, fp32
// accumulate): fp32-class accuracy (~2^-17 relative per product) at a third of the bf16 rate instead of the 1/16 of the fp32 MFMA.
// What differs between the nine GEMMs of a layer's forward / backward is only how an operand tile is FETCHED:
//     XK_ROWS       op(r, k) = p[r * ld + k]                       rows contiguous along the contraction   (x, dy in forward / dgrad)
//     XK_COLS       op(r, k) = p[k * ld + r]                       contraction strided                       (W in dgrad; x, dy in wgrad)
//     XK_CONV_ROWS  op(pos, (tap, c)) = x[pos + off(tap)][c] or 0  implicit im2col of a channels-last map   (conv forward / dgrad)
//     XK_CONV_COLS  op((tap, c), pos) = the same element            contraction over positions                (conv wgrad)
// A tile is 128 (or BN) rows x 32 k: global -> registers (the next tile is in flight during the MFMAs of the current one) -> hi / lo
// bf16 planes in LDS (row pitch 80 bytes: fragment reads conflict-free) -> 16-byte fragment reads.  Contraction-strided operands are
// loaded as 4-row x KPT-k register blocks and transposed on the way into LDS.  Long contractions with few output tiles (weight
// gradients: thousands to a million rows reduced into a few hundred outputs) are split over blockIdx.z into fp32 partial tiles that
// a second, deterministic pass sums in a fixed order (no atomics).
#include "amtx_kernels.h"

#include <algorithm>

namespace {

enum { XK_ROWS = 0, XK_COLS = 1, XK_CONV_ROWS = 2, XK_CONV_COLS = 3 };

struct XOp {
    const float* p;
    int64_t ld;
    int kind;
    int T, F, C;   // conv kinds: the map is [rows = clips x T][F][C] channels-last; positions = rows * F
};

struct XArgs {
    XOp A, B;
    const float* bias;
    float* C;
    int64_t ldc;
    int64_t M, N, K;
    int64_t kchunk;           // contraction elements per split (a multiple of 32)
    int64_t c_split_stride;   // elements between the partial results of two splits
};

constexpr int XBK = 32;
constexpr int XBM = 128;
constexpr int XPITCH = 80;   // bytes per LDS tile row: 32 bf16 + 16 bytes of padding

typedef __attribute__((ext_vector_type(8))) __bf16 x_bf16x8;
__device__ __forceinline__ f32x4_t xmfma(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x_bf16x8, a), __builtin_bit_cast(x_bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void tap_offsets(int tap, int& dy, int& dx) {
    dy = tap / 3 - 1;
    dx = tap - (tap / 3) * 3 - 1;
}

// One operand tile: ROWS x 32 k, 256 threads, ROWS / 32 float4 loads per thread.
template <int ROWS>
struct XTile {
    static constexpr int NL = ROWS / 32;     // float4 loads per thread = k's per thread of the strided kinds
    float4 r[NL];

    // geometry of the conv kinds, set up once per block
    int pos_t[NL], pos_f[NL];                // XK_CONV_ROWS: (t, f) of this thread's rows; XK_CONV_COLS: of its current k's
    int64_t pos_lin[NL];
    int cur_tap, cur_c;                      // XK_CONV_ROWS: (tap, channel) of this thread's four consecutive k of the current step

    __device__ __forceinline__ void init(const XOp& op, int64_t row0, int64_t nrows, int64_t kbeg) {
        const int tid = threadIdx.x;
        cur_tap = cur_c = 0;
        if (op.kind == XK_CONV_ROWS) {
            const int64_t k = kbeg + 4 * (tid & 7);
            cur_tap = (int)(k / op.C);
            cur_c = (int)(k - (int64_t)cur_tap * op.C);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int64_t pos = row0 + (tid >> 3) + 32 * i;
                pos_lin[i] = pos;
                const int64_t row = pos / op.F;
                pos_f[i] = (int)(pos - row * op.F);
                pos_t[i] = (int)(row % op.T);
            }
        } else if (op.kind == XK_CONV_COLS) {
            const int kg = tid / (ROWS / 4);
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int64_t pos = kbeg + kg * NL + j;
                pos_lin[j] = pos;
                const int64_t row = pos / op.F;
                pos_f[j] = (int)(pos - row * op.F);
                pos_t[j] = (int)(row % op.T);
            }
        }
    }

    // issue the global loads of the tile whose contraction range starts at k0 (rows row0 .. row0 + ROWS of `nrows`, k < kend)
    __device__ __forceinline__ void load(const XOp& op, int64_t row0, int64_t nrows, int64_t k0, int64_t kend) {
        const int tid = threadIdx.x;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        if (op.kind == XK_ROWS) {
            const int64_t k = k0 + 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int64_t row = row0 + (tid >> 3) + 32 * i;
                r[i] = zero;
                if (row < nrows && k < kend) r[i] = *reinterpret_cast<const float4*>(op.p + row * op.ld + k);
            }
        } else if (op.kind == XK_COLS) {
            const int rq = tid % (ROWS / 4), kg = tid / (ROWS / 4);
            const int64_t row = row0 + 4 * rq;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int64_t k = k0 + kg * NL + j;
                r[j] = zero;
                if (row < nrows && k < kend) r[j] = *reinterpret_cast<const float4*>(op.p + k * op.ld + row);
            }
        } else if (op.kind == XK_CONV_ROWS) {
            // k = tap * C + c; a thread's four consecutive k lie inside one tap (C is a multiple of 4); (cur_tap, cur_c) follow k0
            int dy, dx;
            tap_offsets(cur_tap, dy, dx);
            const int64_t shift = (int64_t)dy * op.F + dx;
            const bool kok = cur_tap < 9 && k0 + 4 * (tid & 7) < kend;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                r[i] = zero;
                const bool ok = kok && pos_lin[i] < nrows && (unsigned)(pos_t[i] + dy) < (unsigned)op.T && (unsigned)(pos_f[i] + dx) < (unsigned)op.F;
                if (ok) r[i] = *reinterpret_cast<const float4*>(op.p + (pos_lin[i] + shift) * op.C + cur_c);
            }
        } else {   // XK_CONV_COLS: rows = (tap, c), contraction = positions
            const int rq = tid % (ROWS / 4);
            const int64_t row = row0 + 4 * rq;
            if (op.C >= 4) {
                const int tap = (int)(row / op.C);
                const int c = (int)(row - (int64_t)tap * op.C);
                int dy, dx;
                tap_offsets(tap, dy, dx);
                const int64_t shift = (int64_t)dy * op.F + dx;
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    r[j] = zero;
                    const bool ok = row < nrows && pos_lin[j] < kend && (unsigned)(pos_t[j] + dy) < (unsigned)op.T && (unsigned)(pos_f[j] + dx) < (unsigned)op.F;
                    if (ok) r[j] = *reinterpret_cast<const float4*>(op.p + (pos_lin[j] + shift) * op.C + c);
                }
            } else {   // one channel (the first layer's weight gradient): the four rows of a quad are four taps
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int dy, dx;
                        tap_offsets((int)(row + q), dy, dx);
                        const bool ok = row + q < nrows && pos_lin[j] < kend && (unsigned)(pos_t[j] + dy) < (unsigned)op.T && (unsigned)(pos_f[j] + dx) < (unsigned)op.F;
                        v[q] = ok ? op.p[pos_lin[j] + (int64_t)dy * op.F + dx] : 0.f;
                    }
                    r[j] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }

    // contraction-over-positions tiles: advance this thread's positions by one 32-deep step
    __device__ __forceinline__ void advance(const XOp& op) {
        if (op.kind == XK_CONV_ROWS) {
            cur_c += XBK;
            while (cur_c >= op.C) {
                cur_c -= op.C;
                ++cur_tap;
            }
            return;
        }
        if (op.kind != XK_CONV_COLS) return;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            pos_lin[j] += XBK;
            pos_f[j] += XBK;
            while (pos_f[j] >= op.F) {
                pos_f[j] -= op.F;
                if (++pos_t[j] >= op.T) pos_t[j] = 0;
            }
        }
    }

    // registers -> hi / lo bf16 planes in LDS, [row][k] with a pitch of XPITCH bytes
    __device__ __forceinline__ void store(const XOp& op, char* hi, char* lo) const {
        const int tid = threadIdx.x;
        if (op.kind == XK_ROWS || op.kind == XK_CONV_ROWS) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int off = ((tid >> 3) + 32 * i) * XPITCH + (tid & 7) * 8;
                uint2 h, l;
                split_bf16x2(r[i].x, r[i].y, h.x, l.x);
                split_bf16x2(r[i].z, r[i].w, h.y, l.y);
                *reinterpret_cast<uint2*>(hi + off) = h;
                *reinterpret_cast<uint2*>(lo + off) = l;
            }
        } else {
            const int rq = tid % (ROWS / 4), kg = tid / (ROWS / 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[NL];
#pragma unroll
                for (int j = 0; j < NL; ++j) v[j] = q == 0 ? r[j].x : q == 1 ? r[j].y : q == 2 ? r[j].z : r[j].w;
                const int off = (4 * rq + q) * XPITCH + kg * NL * 2;
                if constexpr (NL == 4) {
                    uint2 h, l;
                    split_bf16x2(v[0], v[1], h.x, l.x);
                    split_bf16x2(v[2], v[3], h.y, l.y);
                    *reinterpret_cast<uint2*>(hi + off) = h;
                    *reinterpret_cast<uint2*>(lo + off) = l;
                } else if constexpr (NL == 2) {
                    uint32_t h, l;
                    split_bf16x2(v[0], v[1], h, l);
                    *reinterpret_cast<uint32_t*>(hi + off) = h;
                    *reinterpret_cast<uint32_t*>(lo + off) = l;
                } else {
                    uint32_t h, l;
                    split_bf16x2(v[0], 0.f, h, l);
                    *reinterpret_cast<uint16_t*>(hi + off) = (uint16_t)h;
                    *reinterpret_cast<uint16_t*>(lo + off) = (uint16_t)l;
                }
            }
        }
    }
};

template <int BN>
__global__ __launch_bounds__(256) void xgemm_kernel(XArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * XBM * XPITCH + 2 * BN * XPITCH];
    char* a_hi = smem;
    char* a_lo = smem + XBM * XPITCH;
    char* b_hi = smem + 2 * XBM * XPITCH;
    char* b_lo = b_hi + BN * XPITCH;
    constexpr int NT = BN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * XBM;
    const int64_t n0 = (int64_t)blockIdx.y * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
    const int64_t kend = std::min<int64_t>(g.K, kbeg + g.kchunk);

    XTile<XBM> ta;
    XTile<BN> tb;
    ta.init(g.A, m0, g.M, kbeg);
    tb.init(g.B, n0, g.N, kbeg);

    f32x4_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    ta.load(g.A, m0, g.M, kbeg, kend);
    tb.load(g.B, n0, g.N, kbeg, kend);
    const int frag = (lane & 15) * XPITCH + (lane >> 4) * 16;
    for (int64_t k0 = kbeg; k0 < kend; k0 += XBK) {
        ta.store(g.A, a_hi, a_lo);
        tb.store(g.B, b_hi, b_lo);
        __syncthreads();
        if (k0 + XBK < kend) {          // the next tile travels while this one is on the matrix cores
            ta.advance(g.A);
            tb.advance(g.B);
            ta.load(g.A, m0, g.M, k0 + XBK, kend);
            tb.load(g.B, n0, g.N, k0 + XBK, kend);
        }
        uint4 ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int off = (32 * wave + 16 * mt) * XPITCH + frag;
            ah[mt] = *reinterpret_cast<const uint4*>(a_hi + off);
            al[mt] = *reinterpret_cast<const uint4*>(a_lo + off);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint4 bh = *reinterpret_cast<const uint4*>(b_hi + 16 * nt * XPITCH + frag);
            const uint4 bl = *reinterpret_cast<const uint4*>(b_lo + 16 * nt * XPITCH + frag);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][nt] = xmfma(ah[mt], bh, acc[mt][nt]);
                acc[mt][nt] = xmfma(ah[mt], bl, acc[mt][nt]);
                acc[mt][nt] = xmfma(al[mt], bh, acc[mt][nt]);
            }
        }
        __syncthreads();
    }

    // D tile of a wave: lane holds column n = lane & 15 and rows 4 (lane >> 4) .. + 3 of every 16 x 16 fragment
    float* C = g.C + (int64_t)blockIdx.z * g.c_split_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t n = n0 + 16 * nt + (lane & 15);
        if (n >= g.N) continue;
        const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int64_t m = m0 + 32 * wave + 16 * mt + 4 * (lane >> 4) + rr;
                if (m < g.M) C[m * g.ldc + n] = acc[mt][nt][rr] + bv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Conv2d(3x3, pad 1) forward / input gradient with few output channels (C_in a multiple of 32, <= 64; N <= 64): in the generic kernel
// above every one of the nine taps re-fetches and re-splits its own 128 x C_in operand tile, and with N = 32 columns there are only 12
// MFMAs per wave to pay for it -- the conversion, not the matrix work, bounded those layers.  Here a block
//   * converts the whole (N x 9 C_in) weight matrix to hi / lo bf16 ONCE into LDS,
//   * per kernel row dy converts ONE strip of 130 consecutive input positions (the 128 of the tile shifted by dy rows, plus one on
//     either side) and serves the three taps dx = -1, 0, +1 from it as row-shifted fragment reads; positions whose column f + dx falls
//     off the band are zeroed on the fragment (per-lane mask), rows off the clip at strip load,
// so a tile costs 3 strip conversions instead of 9 tile conversions and no weight conversion per k-step.  Strips are double-buffered:
// one barrier per kernel row.
struct XConvArgs {
    const float* x;        // [positions][C] channels-last map
    const void* wimg;      // LDS image of the weights: [hi, lo][BN][9 C bf16 + 16 bytes], k = tap * C + c (xconv_wprep_kernel)
    const float* bias;
    float* y;              // [positions][N]
    int64_t M;             // positions
    int N, C, T, F;
};

// NS: float4 strip elements per thread (5 for C = 32, 9 for C = 64).  NS <= 5: all three strips of the tile are requested up front
// (60 registers) and the weight conversion runs underneath them -- one exposed memory round trip per tile instead of four.
// [N][9 C] fp32 (k = tap * C + c) -> the kernel's LDS image: plane p (0 = hi, 1 = lo), row n < bn, pitch 18 C + 16 bytes; rows >= N zero
__global__ __launch_bounds__(256) void xconv_wprep_kernel(const float* w, int N, int K9, int bn, char* img) {
    const int WP = K9 * 2 + 16;
    const int wq = K9 >> 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < bn * wq; idx += gridDim.x * 256) {
        const int n = idx / wq, kq = idx - n * wq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) v = *reinterpret_cast<const float4*>(w + (int64_t)n * K9 + 4 * kq);
        uint2 h, l;
        split_bf16x2(v.x, v.y, h.x, l.x);
        split_bf16x2(v.z, v.w, h.y, l.y);
        *reinterpret_cast<uint2*>(img + n * WP + kq * 8) = h;
        *reinterpret_cast<uint2*>(img + (size_t)bn * WP + n * WP + kq * 8) = l;
    }
    // the 16 padding bytes of every row are never read by a fragment (k < 9 C), they stay as they are
}

template <int BN, int NS>
__global__ __launch_bounds__(256) void xconv_kernel(XConvArgs g) {
    extern __shared__ __attribute__((aligned(16))) char xsm[];
    constexpr int NT = BN / 16;
    constexpr bool PRE = NS <= 5;
    constexpr int SROWS = XBM + 2;
    const int C = g.C, K9 = 9 * C;
    const int WP = K9 * 2 + 16, SP = C * 2 + 16;      // row pitches in bytes (16-byte padding: conflict-free 16-byte fragment reads)
    char* w_hi = xsm;
    char* w_lo = w_hi + BN * WP;
    char* strips = w_lo + BN * WP;                    // [2 buffers][hi, lo][SROWS][SP]
    const int SB = SROWS * SP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ntiles = (g.M + XBM - 1) / XBM;
    bool first = true;
    // persistent blocks: the weight image is copied once per block, not once per 128-position tile (it was 43 % of a tile's loads,
    // and the kernel sits at a CU's ~10 bytes / clock load path)
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * XBM;

    // ---- this thread's strip elements: (strip row i, channel quad cq), base position m0 - 1 + i and its frame index
    const int cq4 = C >> 2;
    const int ns = (SROWS * cq4 + 255) >> 8;
    int soff[NS], st[NS];
    int64_t spos[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int idx = tid + 256 * j;
        const int i = idx / cq4, cq = idx - i * cq4;
        const int64_t b = m0 - 1 + i;
        const bool ok = j < ns && i < SROWS && b >= 0 && b < g.M;
        soff[j] = i * SP + cq * 8;
        spos[j] = ok ? b * C + 4 * cq : -1;
        st[j] = ok ? (int)((b / g.F) % g.T) : 0;
    }
    // ---- this lane's output positions: column masks of the dx = -1 / +1 taps
    bool ok_m[2], ok_p[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int64_t p = m0 + 32 * wave + 16 * mt + (lane & 15);
        const int f = (int)(p % g.F);
        ok_m[mt] = f >= 1;
        ok_p[mt] = f + 1 < g.F;
    }

    f32x4_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // unconditional loads (rows off the clip re-read the base row) + a validity bit per load, applied when the strip is stored:
    // a branch around a load makes hipcc wait for that load at the join (one memory round trip per load instead of one per strip)
    float4 v[PRE ? 3 : 1][NS];
    unsigned vmask[PRE ? 3 : 1];
#define XCONV_LOAD(SLOT, DY)                                                                                  \
    vmask[SLOT] = 0;                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NS; ++j) {                                                          \
        const bool ok_ = spos[j] >= 0 && (unsigned)(st[j] + (DY)) < (unsigned)g.T;                              \
        vmask[SLOT] |= (unsigned)ok_ << j;                                                                    \
        v[SLOT][j] = *reinterpret_cast<const float4*>(g.x + (spos[j] >= 0 ? spos[j] : 0) + (ok_ ? (int64_t)(DY) * g.F * C : 0)); \
    }
#define XCONV_STORE(SLOT, BUF)                                                                                \
    _Pragma("unroll") for (int j = 0; j < NS; ++j) {                                                          \
        if (j < ns && tid + 256 * j < SROWS * cq4) {                                                          \
            if (!(vmask[SLOT] >> j & 1)) v[SLOT][j] = make_float4(0.f, 0.f, 0.f, 0.f);                        \
            uint2 h, l;                                                                                       \
            split_bf16x2(v[SLOT][j].x, v[SLOT][j].y, h.x, l.x);                                               \
            split_bf16x2(v[SLOT][j].z, v[SLOT][j].w, h.y, l.y);                                               \
            *reinterpret_cast<uint2*>(strips + (BUF) * 2 * SB + soff[j]) = h;                                 \
            *reinterpret_cast<uint2*>(strips + (BUF) * 2 * SB + SB + soff[j]) = l;                            \
        }                                                                                                     \
    }
    XCONV_LOAD(0, -1)
    if constexpr (PRE) {
        XCONV_LOAD(1, 0)
        XCONV_LOAD(2, 1)
    }
    if (first) {
    // ---- weights -> LDS: the hi / lo planes were converted ONCE per call into an LDS-shaped image (xconv_wprep_kernel); a block
    // copies it with batches of eight 16-byte loads in flight (a per-element convert loop here paid one memory round trip per
    // iteration: 9 of a block's 15 microseconds).  The strip loads above are in flight meanwhile.
    {
        const int nvec = (2 * BN * WP) >> 4;
        const uint4* img = reinterpret_cast<const uint4*>(g.wimg);
        for (int base = 0; base < nvec; base += 8 * 256) {
            uint4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * 256 + tid;
                t[j] = i < nvec ? img[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * 256 + tid;
                if (i < nvec) reinterpret_cast<uint4*>(w_hi)[i] = t[j];
            }
        }
    }
    first = false;
    }
    XCONV_STORE(0, 0)
    __syncthreads();
    const int arow = (32 * wave + (lane & 15)) * SP + (lane >> 4) * 16;      // + 16 mt rows, + (1 + dx) rows, + 64 kc
    const int brow = (lane & 15) * WP + (lane >> 4) * 16;
    const int nkc = C >> 5;
    for (int dyi = 0; dyi < 3; ++dyi) {
        if constexpr (!PRE) {
            if (dyi < 2) { XCONV_LOAD(0, dyi) }             // dy = dyi: the next kernel row's strip travels during this row's MFMAs
        }
        const char* sh = strips + (dyi & 1) * 2 * SB;
        const char* sl = sh + SB;
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int tap = dyi * 3 + dxi;
            for (int kc = 0; kc < nkc; ++kc) {
                uint4 ah[2], al[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int off = arow + (16 * mt + dxi) * SP + kc * 64;
                    ah[mt] = *reinterpret_cast<const uint4*>(sh + off);
                    al[mt] = *reinterpret_cast<const uint4*>(sl + off);
                    const bool keep = dxi == 1 || (dxi == 0 ? ok_m[mt] : ok_p[mt]);
                    if (!keep) {
                        ah[mt] = make_uint4(0, 0, 0, 0);
                        al[mt] = make_uint4(0, 0, 0, 0);
                    }
                }
                const int boff = brow + (tap * C + kc * 32) * 2;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uint4 bh = *reinterpret_cast<const uint4*>(w_hi + 16 * nt * WP + boff);
                    const uint4 bl = *reinterpret_cast<const uint4*>(w_lo + 16 * nt * WP + boff);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt][nt] = xmfma(ah[mt], bh, acc[mt][nt]);
                        acc[mt][nt] = xmfma(ah[mt], bl, acc[mt][nt]);
                        acc[mt][nt] = xmfma(al[mt], bh, acc[mt][nt]);
                    }
                }
            }
        }
        if (dyi < 2) {
            if constexpr (PRE) {
                if (dyi == 0) { XCONV_STORE(1, 1) } else { XCONV_STORE(2, 0) }
            } else {
                XCONV_STORE(0, (dyi + 1) & 1)
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + (lane & 15);
        if (n >= g.N) continue;
        const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int64_t m = m0 + 32 * wave + 16 * mt + 4 * (lane >> 4) + rr;
                if (m < g.M) g.y[m * g.N + n] = acc[mt][nt][rr] + bv;
            }
    }
    __syncthreads();        // every wave is done with strip buffer 0 before the next tile's first strip is stored
    }
#undef XCONV_LOAD
#undef XCONV_STORE
}

// ---------------------------------------------------------------------------------------------------------------------------
// Conv2d(3x3) WEIGHT gradient for C_in = 32, C_out = 32 / 64:  G[(tap, ci)][co] = sum_p x[p + off(tap)][ci] * dy[p][co].
// In the generic kernel the contraction runs over a million positions and every tap re-fetches and re-converts its own shifted copy
// of x (9 x) and every 128-row tile its own copy of dy (3 x).  Here ONE block keeps the whole 288 x C_out gradient in registers
// (36 / 72 per thread) and streams over its share of the positions, 32 at a time:
//   * per kernel row dy one strip of 48 positions x 32 channels of x is fetched, split into hi / lo bf16 ONCE and stored TRANSPOSED
//     ([channel][position]: the contraction index must be contiguous in an MFMA operand); the three taps dx = -1, 0, +1 read it at a
//     2-byte offset: the dx = 0 fragment is one aligned 16-byte read, the other two are cut out of a 32-byte window with
//     v_alignbyte (LDS b128 reads want 16-byte alignment);
//   * the dy tile (32 positions x C_out) is converted once and stored three times, once per dx, with the positions whose column
//     f + dx falls off the band zeroed (the product must vanish there; masking the shared x strip per tap would cost a select per
//     element); rows off the clip (t + dy) are zeroed in the x strip at load time.
// 18 (tap, 16-channel) row fragments are dealt to the four waves (5, 5, 4, 4).  Partial gradients go to [block][288][C_out] and are
// summed by xreduce_kernel in a fixed order.
struct XWgradArgs {
    const float* x;        // [positions][32]
    const float* dy;       // [positions][CO]
    float* partial;        // [gridDim.x][288][CO]
    int64_t M;             // positions
    int T, F;
    int64_t steps_per_block;   // 32-position steps per block
};

constexpr int WG_C = 32;
constexpr int WG_XP = 112;             // bytes per channel row of a strip: 48 positions x 2 B + 16 B pad
constexpr int WG_DP = 80;              // bytes per output-channel row of a dy tile: 32 positions x 2 B + 16 B pad

template <int NT>
__global__ __launch_bounds__(256) void xwgrad_kernel(XWgradArgs g) {
    constexpr int CO = 16 * NT;
    __shared__ __attribute__((aligned(16))) char wsm[3 * 2 * WG_C * WG_XP + 3 * 2 * CO * WG_DP];
    char* xs = wsm;                                  // [dy 3][plane 2][ci 32][WG_XP]
    char* ds = wsm + 3 * 2 * WG_C * WG_XP;           // [dx 3][plane 2][co CO][WG_DP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t step0 = (int64_t)blockIdx.x * g.steps_per_block;
    const int64_t nsteps_all = (g.M + 31) / 32;
    const int64_t step1 = step0 + g.steps_per_block < nsteps_all ? step0 + g.steps_per_block : nsteps_all;

    // this wave's row fragments: j = wave, wave + 4, ... < 18; fragment j = (tap = j / 2, channel half = j % 2)
    constexpr int MAXF = 5;
    f32x4_t acc[MAXF][NT];
#pragma unroll
    for (int a = 0; a < MAXF; ++a)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[a][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // x strip work items: 3 strips x (12 position quads x 8 channel quads) = 288 4x4 blocks, two rounds of 256 threads
    // dy work items: 8 position quads x CO / 4 channel quads
    constexpr int DYB = 8 * (CO / 4);                // 64 or 128 4x4 blocks of the dy tile
    constexpr int DYR = (3 * DYB + 255) / 256;       // rounds of (block, dx variant) work items: 1 (C_out 32) or 2 (64)
    float4 xv[2][4], dv[DYR][4];
    unsigned xmask = 0, dmask = 0;                   // validity bits of the loads in flight (unconditional loads, mask at store)
    // work distribution: x items 0..255 -> all threads, 256..287 -> threads 0..31; dy items (block, dx variant) are dealt from the
    // LAST thread downwards (item = 255 - tid) so that the first wave, which has the second x round, gets none of them
    auto dy_item = [&](int r) __attribute__((always_inline)) { return r == 0 ? 255 - tid : 256 + tid; };
    // (column, frame) of this thread's first strip position of each round and column of its first dy position, advanced by 32
    // positions per step (no division in the loop)
    int xf[2], xt[2], df[DYR];
    {
        const int64_t tf = (int64_t)g.T * g.F;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + 256 * r, rem = idx % 96, pq = rem >> 3;
            const int64_t bb = step0 * 32 - 8 + 4 * pq + 8 * tf;               // + a multiple of a clip: non-negative, same (t, f)
            xf[r] = (int)(bb % g.F);
            xt[r] = (int)((bb / g.F) % g.T);
        }
#pragma unroll
        for (int r = 0; r < DYR; ++r) {
            const int blk = min(dy_item(r), 3 * DYB - 1) % DYB;
            df[r] = (int)((step0 * 32 + 4 * (blk / (CO / 4))) % g.F);
        }
    }
    auto load_step = [&](int64_t step) __attribute__((always_inline)) {
        const int64_t p0 = step * 32;
        xmask = dmask = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = min(tid + 256 * r, 287);                               // threads past the 288 work items repeat the last one (never stored)
            const int dyi = idx / 96, rem = idx - dyi * 96, pq = rem >> 3, cq = rem & 7;
            int f = xf[r], t = xt[r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t b = p0 - 8 + 4 * pq + j;                             // base position (same row as the outputs that use it)
                const bool inside = b >= 0 && b < g.M;
                const bool ok = inside && (unsigned)(t + dyi - 1) < (unsigned)g.T;
                xmask |= (unsigned)ok << (4 * r + j);
                const int64_t bc = inside ? b : (b < 0 ? 0 : g.M - 1);
                xv[r][j] = *reinterpret_cast<const float4*>(g.x + (bc + (ok ? (int64_t)(dyi - 1) * g.F : 0)) * WG_C + 4 * cq);
                if (++f == g.F) { f = 0; if (++t == g.T) t = 0; }
            }
            xf[r] += 32;
            while (xf[r] >= g.F) { xf[r] -= g.F; if (++xt[r] == g.T) xt[r] = 0; }
        }
#pragma unroll
        for (int r = 0; r < DYR; ++r) {
            const int blk = min(dy_item(r), 3 * DYB - 1) % DYB;
            const int pq = blk / (CO / 4), cq = blk - pq * (CO / 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t p = p0 + 4 * pq + j;
                dmask |= (unsigned)(p < g.M) << (4 * r + j);
                dv[r][j] = *reinterpret_cast<const float4*>(g.dy + (p < g.M ? p : g.M - 1) * CO + 4 * cq);
            }
        }
    };
    auto comp = [](const float4& v, int e) __attribute__((always_inline)) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
    auto store_step = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(xmask >> (4 * r + j) & 1)) xv[r][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < DYR; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(dmask >> (4 * r + j) & 1)) dv[r][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + 256 * r;
            if (idx < 288) {
                const int dyi = idx / 96, rem = idx - dyi * 96, pq = rem >> 3, cq = rem & 7;
#pragma unroll
                for (int e = 0; e < 4; ++e) {                                       // channel 4 cq + e: four consecutive positions
                    uint2 h, l;
                    split_bf16x2(comp(xv[r][0], e), comp(xv[r][1], e), h.x, l.x);
                    split_bf16x2(comp(xv[r][2], e), comp(xv[r][3], e), h.y, l.y);
                    const int off = ((dyi * 2) * WG_C + 4 * cq + e) * WG_XP + pq * 8;
                    *reinterpret_cast<uint2*>(xs + off) = h;
                    *reinterpret_cast<uint2*>(xs + off + WG_C * WG_XP) = l;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < DYR; ++r) {
            const int item = dy_item(r);
            bool zero[4];                                                            // dx = -1: column 0, dx = +1: column F - 1 of the band
            {
                const int dxi = min(item, 3 * DYB - 1) / DYB;
                int f = df[r];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    zero[j] = (dxi == 0 && f == 0) || (dxi == 2 && f == g.F - 1);
                    if (++f == g.F) f = 0;
                }
                df[r] += 32;
                while (df[r] >= g.F) df[r] -= g.F;
            }
            if (item >= 0 && item < 3 * DYB) {
                const int dxi = item / DYB, blk = item - dxi * DYB;
                const int pq = blk / (CO / 4), cq = blk - pq * (CO / 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = zero[j] ? 0.f : comp(dv[r][j], e);
                    uint2 h, l;
                    split_bf16x2(u[0], u[1], h.x, l.x);
                    split_bf16x2(u[2], u[3], h.y, l.y);
                    const int off = ((dxi * 2) * CO + 4 * cq + e) * WG_DP + pq * 8;
                    *reinterpret_cast<uint2*>(ds + off) = h;
                    *reinterpret_cast<uint2*>(ds + off + CO * WG_DP) = l;
                }
            }
        }
    };
    // 16-byte fragment of 8 consecutive positions starting `shift` bf16 into an aligned 32-byte window
    auto frag = [](const char* base, int dxi) __attribute__((always_inline)) -> uint4 {
        // base = address of strip position 8 c + 8 (16-byte aligned); dx = dxi - 1 moves the start by 2 dx bytes
        if (dxi == 1) return *reinterpret_cast<const uint4*>(base);
        if (dxi == 0) {   // bytes [base - 2, base + 14): window [base - 16, base + 16), byte offset 14 = dword 3 + 2 bytes
            const uint4 w0 = *reinterpret_cast<const uint4*>(base - 16), w1 = *reinterpret_cast<const uint4*>(base);
            return make_uint4(__builtin_amdgcn_alignbyte(w1.x, w0.w, 2), __builtin_amdgcn_alignbyte(w1.y, w1.x, 2),
                              __builtin_amdgcn_alignbyte(w1.z, w1.y, 2), __builtin_amdgcn_alignbyte(w1.w, w1.z, 2));
        }
        const uint4 w0 = *reinterpret_cast<const uint4*>(base), w1 = *reinterpret_cast<const uint4*>(base + 16);   // bytes [base + 2, base + 18)
        return make_uint4(__builtin_amdgcn_alignbyte(w0.y, w0.x, 2), __builtin_amdgcn_alignbyte(w0.z, w0.y, 2),
                          __builtin_amdgcn_alignbyte(w0.w, w0.z, 2), __builtin_amdgcn_alignbyte(w1.x, w0.w, 2));
    };

    if (step0 < step1) load_step(step0);
    for (int64_t step = step0; step < step1; ++step) {
        store_step();
        __syncthreads();
        if (step + 1 < step1) load_step(step + 1);
        const int arow = (lane & 15) * WG_XP + (lane >> 4) * 16 + 16;      // strip position 8 (lane >> 4) + 8 of channel row lane & 15
        const int brow = (lane & 15) * WG_DP + (lane >> 4) * 16;
#pragma unroll
        for (int a = 0; a < MAXF; ++a) {
            const int j = wave + 4 * a;
            if (j < 18) {
                const int tap = j >> 1, half = j & 1, dyi = tap / 3, dxi = tap - 3 * dyi;
                const char* xb = xs + ((dyi * 2) * WG_C + 16 * half) * WG_XP + arow;
                const uint4 ah = frag(xb, dxi), al = frag(xb + WG_C * WG_XP, dxi);
                const char* db = ds + (dxi * 2) * CO * WG_DP + brow;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint4 bh = *reinterpret_cast<const uint4*>(db + 16 * n * WG_DP);
                    const uint4 bl = *reinterpret_cast<const uint4*>(db + CO * WG_DP + 16 * n * WG_DP);
                    acc[a][n] = xmfma(ah, bh, acc[a][n]);
                    acc[a][n] = xmfma(ah, bl, acc[a][n]);
                    acc[a][n] = xmfma(al, bh, acc[a][n]);
                }
            }
        }
        __syncthreads();
    }
    float* out = g.partial + (int64_t)blockIdx.x * 288 * CO;
#pragma unroll
    for (int a = 0; a < MAXF; ++a) {
        const int j = wave + 4 * a;
        if (j >= 18) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) out[(16 * j + 4 * (lane >> 4) + rr) * CO + 16 * n + (lane & 15)] = acc[a][n][rr];
    }
}

// out[i] = sum_s partial[s * stride + i] (+ bias[i % ncols]) in a fixed order: 64 outputs per block, four lanes per output walk the
// splits (k = lane, lane + 4, ...), then ((s0 + s1) + (s2 + s3))
__global__ __launch_bounds__(256) void xreduce_kernel(const float* partial, int splits, int64_t stride, int64_t count, const float* bias, int ncols,
                                                     float* out) {
    __shared__ float red[4][64];
    const int sub = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    float s = 0.f;
    if (i < count)
        for (int k = sub; k < splits; k += 4) s += partial[(int64_t)k * stride + i];
    red[sub][l] = s;
    __syncthreads();
    if (sub == 0 && i < count) {
        float v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        if (bias) v += bias[i % ncols];
        out[i] = v;
    }
}

// column sums of a (rows x ncols) matrix with leading dimension ld, split over blockIdx.y: partial[y][n]
__global__ __launch_bounds__(256) void xcolsum_kernel(const float* x, int64_t ld, int64_t rows, int ncols, int64_t rows_per_split, float* partial) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int64_t rbeg = (int64_t)blockIdx.y * rows_per_split, rend = std::min<int64_t>(rows, rbeg + rows_per_split);
    float s = 0.f;
    if (col < ncols)
        for (int64_t r = rbeg + sub; r < rend; r += 4) s += x[r * ld + col];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && col < ncols) partial[(int64_t)blockIdx.y * ncols + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the same for a CONTIGUOUS narrow matrix (ld == ncols, ncols divides 1024: the channels-last gradient maps, a million rows of 32 / 64
// channels): the matrix is one flat stream of float4, a thread's four columns are the same for every float4 it visits (stride 1024
// floats), so the loop is pure coalesced loads + adds; threads that own the same columns meet in LDS at the end.  partial[block][n]
__global__ __launch_bounds__(256) void xcolsum_flat_kernel(const float* x, int64_t total, int ncols, float* partial) {
    __shared__ float red[1024];
    const int64_t n4 = total >> 2;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(red)[threadIdx.x] = s;
    __syncthreads();
    for (int c = threadIdx.x; c < ncols; c += 256) {
        float t = 0.f;
        for (int j = c; j < 1024; j += ncols) t += red[j];
        partial[(int64_t)blockIdx.x * ncols + c] = t;
    }
}

// weight layouts of the convolution GEMMs.  w is the reference's (c_out, c_in, 3, 3).
//   mode 0  forward : out[co][tap][ci]      = w[co][ci][tap]
//   mode 1  dgrad   : out[ci][tap][co]      = w[co][ci][8 - tap]      (correlation with the flipped kernel)
//   mode 2  wgrad   : out = dw[co][ci][tap] = in[(tap * c_in + ci) * c_out + co]     (in = the GEMM's [(tap, ci)][co] result)
__global__ __launch_bounds__(256) void conv_w_permute_kernel(const float* in, float* out, int c_out, int c_in, int mode) {
    const int total = c_out * c_in * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        if (mode == 0) {
            const int ci = i % c_in, tap = (i / c_in) % 9, co = i / (9 * c_in);
            out[i] = in[(co * c_in + ci) * 9 + tap];
        } else if (mode == 1) {
            const int co = i % c_out, tap = (i / c_out) % 9, ci = i / (9 * c_out);
            out[i] = in[(co * c_in + ci) * 9 + (8 - tap)];
        } else {
            const int tap = i % 9, ci = (i / 9) % c_in, co = i / (9 * c_in);
            out[i] = in[(tap * c_in + ci) * c_out + co];
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_xgemm(XArgs g, int splits, hipStream_t stream) {
    AMTX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "matmul: empty problem");
    const int bn = g.N <= 32 ? 32 : g.N <= 64 ? 64 : 128;
    const int64_t ksteps = (g.K + XBK - 1) / XBK;
    splits = (int)std::max<int64_t>(1, std::min<int64_t>(splits, ksteps));
    g.kchunk = (ksteps + splits - 1) / splits * XBK;
    splits = (int)((g.K + g.kchunk - 1) / g.kchunk);
    const int64_t gm = (g.M + XBM - 1) / XBM, gn = (g.N + bn - 1) / bn;
    AMTX_REQUIRE(gm < (1ll << 31) && gn < 65536 && splits < 65536, "matmul: grid too large");
    dim3 grid((unsigned)gm, (unsigned)gn, (unsigned)splits);
    if (bn == 32) hipLaunchKernelGGL(xgemm_kernel<32>, grid, dim3(256), 0, stream, g);
    else if (bn == 64) hipLaunchKernelGGL(xgemm_kernel<64>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(xgemm_kernel<128>, grid, dim3(256), 0, stream, g);
    AMTX_CHECK_LAUNCH();
    return splits;
}

// number of contraction splits: enough blocks to fill the chip when the output has few tiles, at least 8 k-steps per split
int pick_splits(int64_t M, int64_t N, int64_t K) {
    const int bn = N <= 32 ? 32 : N <= 64 ? 64 : 128;
    const int64_t tiles = ((M + XBM - 1) / XBM) * ((N + bn - 1) / bn);
    const int64_t ksteps = (K + XBK - 1) / XBK;
    if (tiles >= 192 || ksteps < 32) return 1;
    // ~4 blocks per CU: a block is latency-bound (two barriers and a dependent load per 32-deep step), co-resident blocks overlap
    return (int)std::max<int64_t>(1, std::min<int64_t>((1024 + tiles - 1) / tiles, ksteps / 16));
}

// C = op(A) op(B)^T through (optional) split partials in `ws`
int matmul(const XOp& A, const XOp& B, const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* ws, size_t ws_bytes,
           hipStream_t stream) {
    XArgs g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K;
    int splits = pick_splits(M, N, K);
    if (splits > 1 && (ldc != N || !ws || ws_bytes < (size_t)splits * M * N * sizeof(float))) splits = 1;
    if (splits == 1) {
        g.bias = bias; g.C = C; g.ldc = ldc; g.c_split_stride = 0;
        const int rc = launch_xgemm(g, 1, stream);
        return rc < 0 ? rc : AMTX_OK;
    }
    g.bias = nullptr; g.C = static_cast<float*>(ws); g.ldc = N; g.c_split_stride = M * N;
    const int used = launch_xgemm(g, splits, stream);
    if (used < 0) return used;
    const int64_t count = M * N;
    hipLaunchKernelGGL(xreduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, (const float*)ws, used, count, count, bias, (int)N, C);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int colsum(const float* x, int64_t ld, int64_t rows, int ncols, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (ld == ncols && 1024 % ncols == 0 && rows * ncols >= (1 << 16) && aligned16(x) && ws && ws_bytes >= (size_t)256 * ncols * sizeof(float)) {
        const int blocks = 256;
        hipLaunchKernelGGL(xcolsum_flat_kernel, dim3(blocks), dim3(256), 0, stream, x, rows * ncols, ncols, static_cast<float*>(ws));
        AMTX_CHECK_LAUNCH();
        hipLaunchKernelGGL(xreduce_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(256), 0, stream, (const float*)ws, blocks, (int64_t)ncols, (int64_t)ncols,
                           (const float*)nullptr, ncols, out);
        AMTX_CHECK_LAUNCH();
        return AMTX_OK;
    }
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(256, rows / 256));
    while (splits > 1 && (size_t)splits * ncols * sizeof(float) > ws_bytes) splits /= 2;
    const int64_t per = (rows + splits - 1) / splits;
    float* partial = splits == 1 ? out : static_cast<float*>(ws);
    hipLaunchKernelGGL(xcolsum_kernel, dim3((unsigned)((ncols + 63) / 64), (unsigned)splits), dim3(256), 0, stream, x, ld, rows, ncols, per, partial);
    AMTX_CHECK_LAUNCH();
    if (splits > 1) {
        hipLaunchKernelGGL(xreduce_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(256), 0, stream, (const float*)partial, splits, (int64_t)ncols, (int64_t)ncols,
                           (const float*)nullptr, ncols, out);
        AMTX_CHECK_LAUNCH();
    }
    return AMTX_OK;
}

// strip kernel when it applies (see xconv_kernel), else the generic implicit GEMM
int conv_rows_matmul(const float* x, const float* w_tapmajor, const float* bias, float* y, int64_t positions, int T, int F, int C, int N,
                     void* img_ws, size_t img_bytes, hipStream_t stream) {
    const int bn = N <= 32 ? 32 : 64;
    const size_t wimg = (size_t)2 * bn * (18 * C + 16);
    const size_t lds = wimg + (size_t)4 * (XBM + 2) * (2 * C + 16);
    if (C % 32 == 0 && C <= 64 && N <= 64 && lds <= 156 * 1024 && img_ws && img_bytes >= wimg) {
        hipLaunchKernelGGL(xconv_wprep_kernel, dim3(16), dim3(256), 0, stream, w_tapmajor, N, 9 * C, bn, static_cast<char*>(img_ws));
        AMTX_CHECK_LAUNCH();
        XConvArgs g{x, img_ws, bias, y, positions, N, C, T, F};
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / lds));
        const int64_t blocks = std::min<int64_t>((positions + XBM - 1) / XBM, 256 * per_cu);
#define XCONV_LAUNCH(BN_, NS_)                                                                             \
        do {                                                                                                  \
            AMTX_GRANT_LDS((xconv_kernel<BN_, NS_>), lds);                                                    \
            hipLaunchKernelGGL((xconv_kernel<BN_, NS_>), dim3((unsigned)blocks), dim3(256), lds, stream, g);  \
        } while (0)
        if (bn == 32 && C == 32) XCONV_LAUNCH(32, 5);
        else if (bn == 32) XCONV_LAUNCH(32, 9);
        else if (C == 32) XCONV_LAUNCH(64, 5);
        else XCONV_LAUNCH(64, 9);
#undef XCONV_LAUNCH
        AMTX_CHECK_LAUNCH();
        return AMTX_OK;
    }
    XOp A{x, 0, XK_CONV_ROWS, T, F, C};
    return matmul(A, XOp{w_tapmajor, (int64_t)9 * C, XK_ROWS, 0, 0, 0}, bias, y, N, positions, N, (int64_t)9 * C, nullptr, 0, stream);
}

XOp rows_op(const float* p, int64_t ld) { return XOp{p, ld, XK_ROWS, 0, 0, 0}; }
XOp cols_op(const float* p, int64_t ld) { return XOp{p, ld, XK_COLS, 0, 0, 0}; }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------ C ABI
extern "C" size_t amtx_matmul_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    const int s = pick_splits(m, n, k);
    return s > 1 ? (size_t)s * m * n * sizeof(float) : 0;
}

extern "C" int amtx_matmul_f32(const float* a, int64_t lda, int a_trans, const float* b, int64_t ldb, int b_trans, const float* bias, float* c,
                               int64_t ldc, int64_t m, int64_t n, int64_t k, void* workspace, size_t workspace_bytes, void* stream) {
    AMTX_REQUIRE(a && b && c && m > 0 && n > 0 && k > 0, "amtx_matmul_f32: null pointer or empty problem");
    AMTX_REQUIRE(aligned16(a) && aligned16(b) && lda % 4 == 0 && ldb % 4 == 0, "amtx_matmul_f32: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    AMTX_REQUIRE((a_trans ? m : k) % 4 == 0 && (b_trans ? n : k) % 4 == 0, "amtx_matmul_f32: the contiguous extent of each operand must be a multiple of 4 (m %lld n %lld k %lld)", (long long)m, (long long)n, (long long)k);
    return matmul(a_trans ? cols_op(a, lda) : rows_op(a, lda), b_trans ? cols_op(b, ldb) : rows_op(b, ldb), bias, c, ldc, m, n, k, workspace,
                  workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int amtx_linear_train_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy, int64_t m,
                                     int n, int k, void* stream) {
    return amtx_matmul_f32(x, ldx, 0, w, ldw, 0, bias, y, ldy, m, n, k, nullptr, 0, stream);
}

extern "C" size_t amtx_linear_bwd_workspace_bytes(int64_t m, int n, int k) {
    return std::max<size_t>(amtx_matmul_workspace_bytes(n, k, m), (size_t)256 * n * sizeof(float)) + 256;
}

extern "C" int amtx_linear_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, int64_t ldw, float* dx, int64_t lddx,
                               float* dw, float* db, int64_t m, int n, int k, void* workspace, size_t workspace_bytes, void* stream) {
    AMTX_REQUIRE(dy && m > 0 && n > 0 && k > 0, "amtx_linear_bwd: null pointer or empty problem");
    int rc;
    if (dx) {   // dx[m][k] = sum_n dy[m][n] w[n][k]
        AMTX_REQUIRE(w, "amtx_linear_bwd: dx needs w");
        if ((rc = amtx_matmul_f32(dy, lddy, 0, w, ldw, 1, nullptr, dx, lddx, m, k, n, nullptr, 0, stream)) != AMTX_OK) return rc;
    }
    if (dw) {   // dw[n][k] = sum_m dy[m][n] x[m][k]
        AMTX_REQUIRE(x, "amtx_linear_bwd: dw needs x");
        if ((rc = amtx_matmul_f32(dy, lddy, 1, x, ldx, 1, nullptr, dw, k, n, k, m, workspace, workspace_bytes, stream)) != AMTX_OK) return rc;
    }
    if (db) return colsum(dy, lddy, m, n, db, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    return AMTX_OK;
}

namespace {
size_t conv_ws_w(int c_in, int c_out) { return ((size_t)9 * c_in * c_out * sizeof(float) + 255) / 256 * 256; }
}

extern "C" size_t amtx_conv3x3_train_workspace_bytes(int64_t rows, int num_bins, int c_in, int c_out) {
    const int64_t positions = rows * num_bins;
    const size_t split = amtx_matmul_workspace_bytes((int64_t)9 * c_in, c_out, positions);
    const size_t image = std::max<size_t>((size_t)2 * 64 * (18 * std::max(c_in, c_out) + 16),    // weight image of the strip kernel
                                          (size_t)768 * 288 * c_out * sizeof(float));           // partial gradients of xwgrad_kernel
    return 2 * conv_ws_w(c_in, c_out) + std::max(std::max<size_t>(split, (size_t)256 * c_out * sizeof(float)), image) + 256;
}

extern "C" int amtx_conv3x3_train_fwd(const float* x, const float* w, const float* bias, float* y, int64_t rows, int frames_per_clip, int num_bins,
                                      int c_in, int c_out, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AMTX_REQUIRE(x && w && y && rows > 0 && frames_per_clip > 0 && num_bins > 0 && rows % frames_per_clip == 0, "amtx_conv3x3_train_fwd: bad arguments");
    AMTX_REQUIRE(rows * num_bins < (1ll << 31), "amtx_conv3x3_train_fwd: too many positions");
    if (c_in % 4 != 0) {
        // the first layer (1 or a few input channels): direct fp32 kernel, weights in the reference layout
        AMTX_REQUIRE(c_out % 8 == 0 && bias, "amtx_conv3x3_train_fwd: c_in %d needs c_out %% 8 == 0 and a bias", c_in);
        Conv1Args c{};
        c.in = x;   // channels-last (rows, bins, c_in)
        c.stride_b = (int64_t)frames_per_clip * num_bins * c_in; c.stride_c = 1; c.stride_t = (int64_t)num_bins * c_in; c.stride_f = c_in;
        c.w = w; c.shift = bias; c.out = y; c.out_type = AMTX_T_F32;
        c.B = (int)(rows / frames_per_clip); c.T = frames_per_clip; c.F = num_bins; c.c_in = c_in; c.c_out = c_out;
        c.groups = 1; c.w_gs = 0; c.shift_gs = 0; c.out_gs = 0; c.relu = 0;
        return amtx_launch_conv1(c, stream);
    }
    AMTX_REQUIRE(workspace && workspace_bytes >= conv_ws_w(c_in, c_out), "amtx_conv3x3_train_fwd: workspace too small");
    float* wf = static_cast<float*>(workspace);
    hipLaunchKernelGGL(conv_w_permute_kernel, dim3((unsigned)((9 * c_in * c_out + 255) / 256)), dim3(256), 0, stream, w, wf, c_out, c_in, 0);
    AMTX_CHECK_LAUNCH();
    char* img = static_cast<char*>(workspace) + 2 * conv_ws_w(c_in, c_out);
    const size_t img_bytes = workspace_bytes > 2 * conv_ws_w(c_in, c_out) ? workspace_bytes - 2 * conv_ws_w(c_in, c_out) : 0;
    return conv_rows_matmul(x, wf, bias, y, rows * num_bins, frames_per_clip, num_bins, c_in, c_out, img, img_bytes, stream);
}

extern "C" int amtx_conv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int64_t rows, int frames_per_clip,
                                int num_bins, int c_in, int c_out, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    AMTX_REQUIRE(dy && rows > 0 && frames_per_clip > 0 && num_bins > 0 && rows % frames_per_clip == 0, "amtx_conv3x3_bwd: bad arguments");
    AMTX_REQUIRE(c_out % 4 == 0, "amtx_conv3x3_bwd: c_out must be a multiple of 4 (got %d)", c_out);
    AMTX_REQUIRE(workspace && workspace_bytes >= amtx_conv3x3_train_workspace_bytes(rows, num_bins, c_in, c_out), "amtx_conv3x3_bwd: workspace too small");
    const int64_t positions = rows * num_bins;
    AMTX_REQUIRE(positions < (1ll << 31), "amtx_conv3x3_bwd: too many positions");
    char* ws = static_cast<char*>(workspace);
    float* wt = reinterpret_cast<float*>(ws);                                   // permuted weights / [(tap, ci)][co] gradient
    float* gt = reinterpret_cast<float*>(ws + conv_ws_w(c_in, c_out));
    void* rest = ws + 2 * conv_ws_w(c_in, c_out);
    const size_t rest_bytes = workspace_bytes - 2 * conv_ws_w(c_in, c_out);
    int rc;
    if (dx) {   // correlation of dy with the flipped kernel: dx[pos][ci] = sum_{tap, co} dy[pos + off(tap)][co] w[co][ci][8 - tap]
        AMTX_REQUIRE(w && c_in % 4 == 0, "amtx_conv3x3_bwd: dx needs w and c_in %% 4 == 0");
        hipLaunchKernelGGL(conv_w_permute_kernel, dim3((unsigned)((9 * c_in * c_out + 255) / 256)), dim3(256), 0, stream, w, wt, c_out, c_in, 1);
        AMTX_CHECK_LAUNCH();
        // the weight image lives at the start of the scratch area; the split partials of dw reuse it afterwards (stream-ordered)
        if ((rc = conv_rows_matmul(dy, wt, nullptr, dx, positions, frames_per_clip, num_bins, c_out, c_in, rest, rest_bytes, stream)) != AMTX_OK) return rc;
    }
    if (dw) {   // G[(tap, ci)][co] = sum_pos x[pos + off(tap)][ci] dy[pos][co], then back to (c_out, c_in, 3, 3)
        AMTX_REQUIRE(x && (c_in == 1 || c_in % 4 == 0), "amtx_conv3x3_bwd: dw needs x and c_in = 1 or a multiple of 4");
        const int wg_blocks = (int)std::min<int64_t>(768, (positions + 31) / 32);
        if (c_in == WG_C && (c_out == 32 || c_out == 64) && rest_bytes >= (size_t)wg_blocks * 288 * c_out * sizeof(float)) {
            XWgradArgs wa{x, dy, static_cast<float*>(rest), positions, frames_per_clip, num_bins, 0};
            const int64_t nsteps = (positions + 31) / 32;
            wa.steps_per_block = (nsteps + wg_blocks - 1) / wg_blocks;
            const int used = (int)((nsteps + wa.steps_per_block - 1) / wa.steps_per_block);
            if (c_out == 32) hipLaunchKernelGGL(xwgrad_kernel<2>, dim3(used), dim3(256), 0, stream, wa);
            else hipLaunchKernelGGL(xwgrad_kernel<4>, dim3(used), dim3(256), 0, stream, wa);
            AMTX_CHECK_LAUNCH();
            const int64_t count = (int64_t)288 * c_out;
            hipLaunchKernelGGL(xreduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, (const float*)rest, used, count, count,
                               (const float*)nullptr, c_out, gt);
            AMTX_CHECK_LAUNCH();
        } else {
            XOp A{x, 0, XK_CONV_COLS, frames_per_clip, num_bins, c_in};
            if ((rc = matmul(A, cols_op(dy, c_out), nullptr, gt, c_out, (int64_t)9 * c_in, c_out, positions, rest, rest_bytes, stream)) != AMTX_OK) return rc;
        }
        hipLaunchKernelGGL(conv_w_permute_kernel, dim3((unsigned)((9 * c_in * c_out + 255) / 256)), dim3(256), 0, stream, (const float*)gt, dw, c_out, c_in, 2);
        AMTX_CHECK_LAUNCH();
    }
    if (db) return colsum(dy, c_out, positions, c_out, db, rest, rest_bytes, stream);
    return AMTX_OK;
}
