// Persistent BiLSTM recurrence for gfx950 (hidden = 128 per direction).
//
// Replaces the time loop of nn.LSTM(batch_first, bidirectional) inside LanguageModel.forward
// (amt_tools/models/onsetsframes.py:466-575; the reference's 512-frame eval chunking is a numerical
// no-op, SURVEY finding F9).  The input projection W_ih x + b_ih + b_hh is a separate GEMM (gemm.hip);
// this kernel runs the T dependent steps
//     gates = xproj_t + W_hh h_{t-1};  c = sig(f) c + sig(i) tanh(g);  h = sig(o) tanh(c)   (gate order i,f,g,o)
//
// One 512-thread block (8 waves, two per SIMD) = one direction x 16 clips, resident for the whole sequence:
//   * W_hh (512x128) lives in registers for all T steps: wave w owns hidden units [16w, 16w+16) and their
//     four gates = 4 MFMA row tiles x 4 k-steps (64 VGPRs in bf16, 128 in x3); the two waves of a SIMD
//     overlap one's MFMAs with the other's gate math, which shortens the per-step dependency chain,
//   * h_{t-1} (16 clips x 128) is the MFMA B operand, exchanged through a double-buffered 4 KiB LDS tile
//     (row pitch 272 B -> conflict-free 16-byte fragment reads), one barrier per step,
//   * computed swapped (D' = W_hh . h^T) so i,f,g,o of one (clip, unit) land in the same lane and the
//     cell update is lane-local; c stays in registers,
//   * xproj of step t+1 is prefetched into registers while step t's MFMAs run and is used as the MFMA
//     accumulator init (no separate add).
// The recurrence is a dependency chain (latency-bound, not roofline-bound): throughput comes from running
// ceil(B/16) x 2 directions x groups blocks concurrently.

#include "amtx_f16_names.h"
#include "amtx_kernels.h"

#include <type_traits>

namespace {

constexpr int H = 128;
constexpr int UB = 1;                        // 16-unit blocks per wave
constexpr int LWAVES = H / (16 * UB);        // waves per block (8): wave w owns hidden units [16*UB*w, 16*UB*(w+1))
constexpr int LTHREADS = 64 * LWAVES;
constexpr int HP = H + 8;                    // bf16 elements per LDS row
constexpr int HBUF_BYTES = 16 * HP * 2;      // one plane of one buffer

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
    return amtx_mfma_16x16x32(a, b, c);
}

// FAST (bf16 mode): v_exp_f32 + v_rcp_f32 (1 ulp each) -- far below the bf16 rounding of h.
template <bool FAST>
__device__ __forceinline__ float sigmoid_f(float x) {
    if (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return 1.0f / (1.0f + expf(-x));
}
template <bool FAST>
__device__ __forceinline__ float tanh_f(float x) {
    if (FAST) return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.0f);
    return tanhf(x);
}

// raw (still packed) x-projection registers of one step: converting bf16 -> fp32 right after the load would
// put the wait for the prefetch at the load instead of at the first use one step later.
template <int X_TYPE> struct XRaw { typedef float4 type; };
template <> struct XRaw<AMTX_T_BF16> { typedef uint2 type; };

template <int X_TYPE>
__device__ __forceinline__ void load_x(const char* xbase, int64_t row_off, typename XRaw<X_TYPE>::type (&dst)[UB][4]) {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t e = row_off + q * 128 + 16 * ub;
            dst[ub][q] = *reinterpret_cast<const typename XRaw<X_TYPE>::type*>(xbase + e * (X_TYPE == AMTX_T_F32 ? 4 : 2));
        }
}
__device__ __forceinline__ f32x4_t unpack_x(float4 v) { return (f32x4_t){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ f32x4_t unpack_x(uint2 v) {
    return (f32x4_t){unpack16_lo(v.x), unpack16_hi(v.x), unpack16_lo(v.y), unpack16_hi(v.y)};
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for its global loads/stores (the
// x-projection prefetch and the h stores stay in flight across the barrier).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // keep the next step's register-only work (e.g. unpacking the prefetched x-projection) below this point:
    // hoisted above, it would drag the wait for the prefetch to right behind its issue
    __builtin_amdgcn_sched_barrier(0);
}

template <int NS, int X_TYPE, int OUT_TYPE, bool FAST>
__device__ __forceinline__ void lstm_step(char* smem, int cur, const uint4 (&wf)[UB][4][4][NS], const typename XRaw<X_TYPE>::type (&xcur)[UB][4],
                                          typename XRaw<X_TYPE>::type (&xnext)[UB][4], float (&c)[UB][4], const char* xbase, char* obase, int t, int tnext,
                                          int clip, int g, int wave, bool clip_ok) {
    const char* hb = smem + cur * NS * HBUF_BYTES;
    uint4 hf[4][NS];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < NS; ++p)
            hf[ks][p] = *reinterpret_cast<const uint4*>(hb + p * HBUF_BYTES + (clip * HP + 32 * ks + 8 * g) * 2);

    // unconditional (steps past the end re-read the last row): a branch here would make the compiler's waitcnt pass
    // assume the prefetch may not have been issued and wait for it right away
    load_x<X_TYPE>(xbase, (int64_t)tnext * 1024, xnext);

    f32x4_t acc[UB][4];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[ub][q] = unpack_x(xcur[ub][q]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[ub][q] = mfma16(wf[ub][q][ks][0], hf[ks][0], acc[ub][q]);
                if (NS == 2) {
                    acc[ub][q] = mfma16(wf[ub][q][ks][0], hf[ks][1], acc[ub][q]);
                    acc[ub][q] = mfma16(wf[ub][q][ks][1], hf[ks][0], acc[ub][q]);
                }
            }

    char* hn = smem + (cur ^ 1) * NS * HBUF_BYTES;
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoid_f<FAST>(acc[ub][0][r]);
            const float fg = sigmoid_f<FAST>(acc[ub][1][r]);
            const float gg = tanh_f<FAST>(acc[ub][2][r]);
            const float og = sigmoid_f<FAST>(acc[ub][3][r]);
            c[ub][r] = fg * c[ub][r] + ig * gg;
            h[r] = og * tanh_f<FAST>(c[ub][r]);
        }
        uint2 hiw, low = make_uint2(0, 0);
        if (NS == 2) {
            split_bf16x2(h[0], h[1], hiw.x, low.x);
            split_bf16x2(h[2], h[3], hiw.y, low.y);
        } else {
            hiw = make_uint2(pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]));
        }
        const int hoff = (clip * HP + 16 * UB * wave + 16 * ub + 4 * g) * 2;
        *reinterpret_cast<uint2*>(hn + hoff) = hiw;
        if (NS == 2) *reinterpret_cast<uint2*>(hn + HBUF_BYTES + hoff) = low;
        if (clip_ok) {
            const int64_t e = (int64_t)t * 256 + 16 * ub;
            if (OUT_TYPE == AMTX_T_BF16) *reinterpret_cast<uint2*>(obase + e * 2) = hiw;
            else *reinterpret_cast<float4*>(obase + e * 4) = make_float4(h[0], h[1], h[2], h[3]);
        }
    }
    lds_barrier();
}

template <int NS, int X_TYPE, int OUT_TYPE>
__global__ __launch_bounds__(LTHREADS) void bilstm_kernel(LstmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][NS planes][16][HP] bf16
    constexpr bool FAST = true;   // v_exp_f32 / v_rcp_f32 are ~1 ulp: also fine for the fp32-class (two-plane) mode, checked at 1e-4 / 2e-4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 16, dir = blockIdx.y, grp = blockIdx.z;
    const int b = b0 + clip;
    const bool clip_ok = b < a.B;
    const int T = a.T;

    // ---- stationary recurrent weights
    uint4 wf[UB][4][4][NS];
    {
        const uint4* w = reinterpret_cast<const uint4*>(a.whh + (int64_t)grp * a.w_gs) + lane;
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        wf[ub][q][ks][p] = w[((((((dir * LWAVES + wave) * UB + ub) * 4 + q) * 4 + ks) * NS) + p) * 64];
    }

    for (int i = tid; i < 2 * NS * HBUF_BYTES / 16; i += LTHREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    float c[UB][4];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[ub][r] = 0.f;

    const char* xbase = reinterpret_cast<const char*>(a.xproj) +
                        ((int64_t)grp * a.x_gs + (int64_t)(clip_ok ? b : 0) * T * 1024 + dir * 512 + 16 * UB * wave + 4 * g) * (X_TYPE == AMTX_T_BF16 ? 2 : 4);
    char* obase = reinterpret_cast<char*>(a.out) +
                  ((int64_t)grp * a.out_gs + (int64_t)(clip_ok ? b : 0) * T * 256 + dir * 128 + 16 * UB * wave + 4 * g) * (OUT_TYPE == AMTX_T_BF16 ? 2 : 4);

    // x-projection ring: the rows of steps s+1..s+3 are in flight while step s runs (HBM latency is about as long as
    // one step, so a one-step lookahead stalls every step).  Four steps per iteration so the ring slots are named
    // registers (no copies, no dynamic indexing).
    typename XRaw<X_TYPE>::type x0[UB][4], x1[UB][4], x2[UB][4], x3[UB][4];
    auto tidx = [&](int s) { s = s < T ? s : T - 1; return (int64_t)(dir == 0 ? s : T - 1 - s); };
    load_x<X_TYPE>(xbase, tidx(0) * 1024, x0);
    load_x<X_TYPE>(xbase, tidx(1) * 1024, x1);
    load_x<X_TYPE>(xbase, tidx(2) * 1024, x2);
    __syncthreads();

    for (int s = 0; s < T; s += 4) {
        // straight-line body: the (up to three) steps past T run on clamped rows with their stores masked.  Branching
        // around them would give the loop header several predecessor states and hipcc answers that with a full
        // s_waitcnt vmcnt(0) per iteration, which drains the x-projection ring.
        lstm_step<NS, X_TYPE, OUT_TYPE, FAST>(smem, 0, wf, x0, x3, c, xbase, obase, (int)tidx(s), (int)tidx(s + 3), clip, g, wave, clip_ok);
        lstm_step<NS, X_TYPE, OUT_TYPE, FAST>(smem, 1, wf, x1, x0, c, xbase, obase, (int)tidx(s + 1), (int)tidx(s + 4), clip, g, wave, clip_ok && s + 1 < T);
        lstm_step<NS, X_TYPE, OUT_TYPE, FAST>(smem, 0, wf, x2, x1, c, xbase, obase, (int)tidx(s + 2), (int)tidx(s + 5), clip, g, wave, clip_ok && s + 2 < T);
        lstm_step<NS, X_TYPE, OUT_TYPE, FAST>(smem, 1, wf, x3, x2, c, xbase, obase, (int)tidx(s + 3), (int)tidx(s + 6), clip, g, wave, clip_ok && s + 3 < T);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Four clips per block.  With 16 clips per block a batch of 512 clips is only 64 blocks: a quarter of the chip
// works, and each of those CUs is bound by the quarter-rate v_exp/v_rcp of 2048 cell updates per step.  Here the
// product is NOT swapped (D = h . W_hh^T): the four clips sit in rows 0, 4, 8, 12 of the 16-row h tile (the other
// rows stay zero), so accumulator register 0 of lane (unit = lane & 15, clip = lane >> 4) holds that cell's gate
// and every lane updates exactly ONE cell per step -- a quarter of the vector work per block and four times as
// many blocks.  Operand packing (W_hh fragments, h tile) and the k order are those of the 16-clip kernel, so
// both produce the same bits.
template <int X_TYPE> struct XScalar { typedef float type; };
// bf16: the aligned dword that contains the element, untouched until its use one step later (a 16-bit load gets a
// zero-extension / pair-packing instruction right behind it, which drags the wait for the prefetch to the load)
template <> struct XScalar<AMTX_T_BF16> { typedef unsigned int type; };
__device__ __forceinline__ float unpack_xs(float v, bool) { return v; }
__device__ __forceinline__ float unpack_xs(unsigned int v, bool odd) { return odd ? unpack16_hi(v) : unpack16_lo(v); }

template <int X_TYPE>
__device__ __forceinline__ void load_x4(const char* xbase, int64_t row_off, typename XScalar<X_TYPE>::type (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        if (X_TYPE == AMTX_T_F32) dst[q] = *reinterpret_cast<const typename XScalar<X_TYPE>::type*>(xbase + (row_off + q * 128) * 4);
        else dst[q] = *reinterpret_cast<const unsigned int*>(xbase + (row_off + q * 128) * 2);   // xbase is rounded down to the dword
    }
}

// NC = clips per lane: 1 -> four clips per block in rows 0 / 4 / 8 / 12 of the h tile (accumulator register 0), 2 -> eight clips in
// rows 0 / 2 / ... / 14 (registers 0 and 2): the same 16 MFMAs per wave and step serve twice the clips.
template <int NS, int X_TYPE, int OUT_TYPE, bool FAST, int NC>
__device__ __forceinline__ void lstm4_step(char* smem, int cur, const uint4 (&wf)[4][4][NS], const typename XScalar<X_TYPE>::type (&xcur)[NC][4],
                                           typename XScalar<X_TYPE>::type (&xnext)[NC][4], float (&c)[NC], const char* const (&xbase)[NC],
                                           char* const (&obase)[NC], int t, int tnext, int lane, const int (&hwoff)[NC], const bool (&clip_ok)[NC],
                                           float* save) {
    constexpr int RSTEP = 4 / NC;                 // accumulator register (= h tile row inside the lane group) of clip j: RSTEP * j
    const char* hb = smem + cur * NS * HBUF_BYTES;
    uint4 hf[4][NS];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < NS; ++p)
            hf[ks][p] = *reinterpret_cast<const uint4*>(hb + p * HBUF_BYTES + ((lane & 15) * HP + 32 * ks + 8 * (lane >> 4)) * 2);

#pragma unroll
    for (int j = 0; j < NC; ++j) load_x4<X_TYPE>(xbase[j], (int64_t)tnext * 1024, xnext[j]);     // unconditional, see lstm_step

    f32x4_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[q][RSTEP * j] = unpack_xs(xcur[j][q], lane & 1);
    }
    // (two planes: the three products of a k-step one gate after the other, not one accumulator three times in a row -- the same sums in
    // the same order per accumulator, without back-to-back dependent MFMAs)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = mfma16(hf[ks][0], wf[q][ks][0], acc[q]);
        if (NS == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(hf[ks][NS - 1], wf[q][ks][0], acc[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(hf[ks][0], wf[q][ks][NS - 1], acc[q]);
        }
    }

    char* hn = smem + (cur ^ 1) * NS * HBUF_BYTES;
    // The cells of a lane are independent chains of ~12 dependent operations (five of them exp / rcp pairs): all arithmetic first, the
    // (per-lane predicated) stores behind it -- with a predicated store between the cells hipcc runs the chains one after the other
    // (a branch around each store), which is most of what a second clip per lane costs.
    float ig[NC], fg[NC], gg[NC], og[NC], h[NC];
    uint32_t hiw[NC], low[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        ig[j] = sigmoid_f<FAST>(acc[0][RSTEP * j]);
        fg[j] = sigmoid_f<FAST>(acc[1][RSTEP * j]);
        gg[j] = tanh_f<FAST>(acc[2][RSTEP * j]);
        og[j] = sigmoid_f<FAST>(acc[3][RSTEP * j]);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        c[j] = fg[j] * c[j] + ig[j] * gg[j];
        h[j] = og[j] * tanh_f<FAST>(c[j]);
        low[j] = 0;
        if (NS == 2) split_bf16x2(h[j], 0.f, hiw[j], low[j]);
        else hiw[j] = pack_bf16x2(h[j], 0.f);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        *reinterpret_cast<unsigned short*>(hn + hwoff[j]) = (unsigned short)hiw[j];
        if (NS == 2) *reinterpret_cast<unsigned short*>(hn + HBUF_BYTES + hwoff[j]) = (unsigned short)low[j];
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        if (NC == 1 && save && clip_ok[j]) {     // training: post-activation gates and the new cell state, [b][t][dir][5][128] (save points at [b][0][dir][0][unit])
            float* sv = save + (int64_t)t * (2 * 5 * H);
            sv[0] = ig[j]; sv[H] = fg[j]; sv[2 * H] = gg[j]; sv[3 * H] = og[j]; sv[4 * H] = c[j];
        }
        if (clip_ok[j]) {
            if (OUT_TYPE == AMTX_T_BF16) *reinterpret_cast<unsigned short*>(obase[j] + (int64_t)t * 256 * 2) = (unsigned short)hiw[j];
            else *reinterpret_cast<float*>(obase[j] + (int64_t)t * 256 * 4) = h[j];
        }
    }
    lds_barrier();
}

template <int NS, int X_TYPE, int OUT_TYPE, int NC = 1>
__global__ __launch_bounds__(LTHREADS) void bilstm4_kernel(LstmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][NS planes][16][HP] bf16, rows 0/4/8/12 (NC = 2: every second row) used
    constexpr bool FAST = true;   // v_exp_f32 / v_rcp_f32 are ~1 ulp: also fine for the fp32-class (two-plane) mode, checked at 1e-4 / 2e-4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = 16 * wave + (lane & 15), cg = lane >> 4;
    const int dir = blockIdx.y, grp = blockIdx.z;
    const int T = a.T;

    uint4 wf[4][4][NS];
    {
        const uint4* w = reinterpret_cast<const uint4*>(a.whh + (int64_t)grp * a.w_gs) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int p = 0; p < NS; ++p) wf[q][ks][p] = w[(((((dir * LWAVES + wave) * 4 + q) * 4 + ks) * NS) + p) * 64];
    }
    for (int i = tid; i < 2 * NS * HBUF_BYTES / 16; i += LTHREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    float c[NC];
    bool clip_ok[NC];
    const char* xbase[NC];
    char* obase[NC];
    int hwoff[NC];
    float* save = nullptr;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int b = (blockIdx.x * 4 + cg) * NC + j;
        clip_ok[j] = b < a.B;
        c[j] = 0.f;
        xbase[j] = reinterpret_cast<const char*>(a.xproj) +
                   ((int64_t)grp * a.x_gs + (int64_t)(clip_ok[j] ? b : 0) * T * 1024 + dir * 512 + (X_TYPE == AMTX_T_BF16 ? (unit & ~1) : unit)) *
                       (X_TYPE == AMTX_T_BF16 ? 2 : 4);
        obase[j] = reinterpret_cast<char*>(a.out) +
                   ((int64_t)grp * a.out_gs + (int64_t)(clip_ok[j] ? b : 0) * T * 256 + dir * 128 + unit) * (OUT_TYPE == AMTX_T_BF16 ? 2 : 4);
        hwoff[j] = ((4 * cg + (4 / NC) * j) * HP + unit) * 2;
        if (NC == 1) save = a.save ? a.save + (((int64_t)grp * a.B + (clip_ok[j] ? b : 0)) * T * 2 + dir) * (5 * H) + unit : nullptr;
    }

    typename XScalar<X_TYPE>::type x0[NC][4], x1[NC][4], x2[NC][4], x3[NC][4];
    auto tidx = [&](int s) { s = s < T ? s : T - 1; return (int64_t)(dir == 0 ? s : T - 1 - s); };
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        load_x4<X_TYPE>(xbase[j], tidx(0) * 1024, x0[j]);
        load_x4<X_TYPE>(xbase[j], tidx(1) * 1024, x1[j]);
        load_x4<X_TYPE>(xbase[j], tidx(2) * 1024, x2[j]);
    }
    __syncthreads();

    for (int s = 0; s < T; s += 4) {
        // straight-line body, stores of the steps past T masked (see bilstm_kernel)
        bool ok1[NC], ok2[NC], ok3[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) { ok1[j] = clip_ok[j] && s + 1 < T; ok2[j] = clip_ok[j] && s + 2 < T; ok3[j] = clip_ok[j] && s + 3 < T; }
        lstm4_step<NS, X_TYPE, OUT_TYPE, FAST, NC>(smem, 0, wf, x0, x3, c, xbase, obase, (int)tidx(s), (int)tidx(s + 3), lane, hwoff, clip_ok, save);
        lstm4_step<NS, X_TYPE, OUT_TYPE, FAST, NC>(smem, 1, wf, x1, x0, c, xbase, obase, (int)tidx(s + 1), (int)tidx(s + 4), lane, hwoff, ok1, save);
        lstm4_step<NS, X_TYPE, OUT_TYPE, FAST, NC>(smem, 0, wf, x2, x1, c, xbase, obase, (int)tidx(s + 2), (int)tidx(s + 5), lane, hwoff, ok2, save);
        lstm4_step<NS, X_TYPE, OUT_TYPE, FAST, NC>(smem, 1, wf, x3, x2, c, xbase, obase, (int)tidx(s + 3), (int)tidx(s + 6), lane, hwoff, ok3, save);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Training: backward recurrence of the BiLSTM (the reference trains with nn.LSTM, amt_tools/train.py:126-141; MIOpen's
// LSTM takes 46 of the 61 ms of GPU time of a training step at 8 clips x 625 frames and keeps the host launch-bound).
// Given dL/dh for every step and the saved post-activation gates / cell states of the forward pass, one persistent
// block (1 direction x 4 clips, same lane = one cell mapping as bilstm4_kernel) walks the steps in reverse:
//     dh   = dout_t + W_hh^T dgates_{next}          (MFMA: dgates tile (4 clips in rows 0/4/8/12) x W_hh as the B operand)
//     do   = dh tanh(c) o(1-o);  dc = dc_next f_next + dh o (1 - tanh(c)^2)
//     di   = dc g i(1-i);  dg = dc i (1-g^2);  df = dc c_prev f(1-f)
// and writes dgates = dL/d(xproj).  The parameter gradients are plain GEMMs over (B*T) on the host side
// (dW_ih = dG^T X, dW_hh = dG^T H_prev, db = sum dG, dX = dG W_ih) -- amt_tools_amd/autograd.py.
constexpr int GP = 4 * H + 8;                     // bf16 elements per row of the dgates LDS tile
constexpr int GBUF_BYTES = 16 * GP * 2;           // one plane of one buffer

struct LstmBwdArgs {
    const float* dout;        // [B][T][256]
    const float* save;        // [B][T][2][5][128]: i, f, g, o (post-activation), c
    const bf16_t* whh_t;      // transposed fragments, see bilstm_pack_dev_kernel
    int planes;
    float* dxproj;            // [B][T][2][512]
    int B, T;
};

#ifdef AMTX_LSTM_TIMING
// Debug build only: cycles wave 0 of block 0 spends per phase of a backward step: [0] saved-value wait + elementwise + LDS writes,
// [1] barrier, [2] fragment reads + MFMAs, [3] steps.  Read with amtxdbg_lstm_prof().
__device__ unsigned long long g_lstm_prof[4];
#define LT_TICK(SLOT) do { const unsigned long long n_ = __builtin_readcyclecounter(); lt_acc[SLOT] += n_ - lt_t; lt_t = n_; } while (0)
#else
#define LT_TICK(SLOT) do {} while (0)
#endif
template <int NS>
__global__ __launch_bounds__(LTHREADS) void bilstm4_bwd_kernel(LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][NS planes][16][GP] bf16, rows 0/4/8/12 used
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = 16 * wave + (lane & 15), cg = lane >> 4;
    const int dir = blockIdx.y;
    const int b = blockIdx.x * 4 + cg;
    const bool clip_ok = b < a.B;
    const int T = a.T;

    // W_hh as the MFMA B operand: column = this wave's unit 16 w + (lane & 15), k = gate row 32 ks + 8 (lane >> 4) + j
    uint4 wt[16][NS];
    {
        const uint4* w = reinterpret_cast<const uint4*>(a.whh_t) + lane;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int p = 0; p < NS; ++p) wt[ks][p] = w[((((dir * LWAVES + wave) * 16 + ks) * NS) + p) * 64];
    }
    for (int i = tid; i < 2 * NS * GBUF_BYTES / 16; i += LTHREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    const int64_t bb = clip_ok ? b : 0;
    const float* sv0 = a.save + ((bb * T) * 2 + dir) * (5 * H) + unit;          // + t * (2*5*H)
    const float* do0 = a.dout + (bb * T) * 256 + dir * H + unit;                 // + t * 256
    float* dx0 = a.dxproj + (bb * T) * 1024 + dir * 512 + unit;                  // + t * 1024 + q * 128
    const int gwoff = (4 * cg * GP + unit) * 2;                                   // byte offset of (slot, gate 0) in a tile plane

    float dh_rec = 0.f, dc_rec = 0.f;
    // the forward pass of direction 0 ran t = 0..T-1, of direction 1 t = T-1..0: walk them backwards.  Step s reads the saved
    // values of frame t(s) and the cell state of the step BEFORE it in forward order, which is frame t(s+1): the seven values of
    // step s+1 are requested (unconditionally, clamped) before step s's mat-vec, so their latency is off the dependency chain.
    auto frame = [&](int s_) { s_ = s_ < T ? s_ : T - 1; return dir == 0 ? T - 1 - s_ : s_; };
    float n_i, n_f, n_g, n_o, n_c, n_do;
    {
        const float* sv = sv0 + (int64_t)frame(0) * (2 * 5 * H);
        n_i = sv[0]; n_f = sv[H]; n_g = sv[2 * H]; n_o = sv[3 * H]; n_c = sv[4 * H];
        n_do = do0[(int64_t)frame(0) * 256];
    }
#ifdef AMTX_LSTM_TIMING
    unsigned long long lt_acc[4] = {0, 0, 0, 0}, lt_t = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < T; ++s) {
        const int t = frame(s);
        const float ig = n_i, fg = n_f, gg = n_g, og = n_o, ct = n_c, dout_t = n_do;
        {
            const int tn = frame(s + 1);
            const float* sv = sv0 + (int64_t)tn * (2 * 5 * H);
            n_i = sv[0]; n_f = sv[H]; n_g = sv[2 * H]; n_o = sv[3 * H]; n_c = sv[4 * H];
            n_do = do0[(int64_t)tn * 256];
        }
        asm volatile("" : "+v"(n_c));                      // n_c is consumed below as c_prev: keep it a plain register read
        const float cp = s + 1 < T ? n_c : 0.f;             // cell state before step t in forward order (zero initial state)
        const float dh = dout_t + dh_rec;
        const float tc = tanh_f<true>(ct);
        const float d_o = dh * tc * og * (1.f - og);
        const float dc = dc_rec + dh * og * (1.f - tc * tc);
        const float d_i = dc * gg * ig * (1.f - ig);
        const float d_g = dc * ig * (1.f - gg * gg);
        const float d_f = dc * cp * fg * (1.f - fg);
        dc_rec = dc * fg;
        const float dgv[4] = {d_i, d_f, d_g, d_o};
        char* gt = smem + (s & 1) * NS * GBUF_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (clip_ok) dx0[(int64_t)t * 1024 + q * H] = dgv[q];
            uint32_t hiw, low = 0;
            if (NS == 2) split_bf16x2(dgv[q], 0.f, hiw, low);
            else hiw = pack_bf16x2(dgv[q], 0.f);
            *reinterpret_cast<unsigned short*>(gt + gwoff + q * H * 2) = (unsigned short)hiw;
            if (NS == 2) *reinterpret_cast<unsigned short*>(gt + GBUF_BYTES + gwoff + q * H * 2) = (unsigned short)low;
        }
        LT_TICK(0);
        lds_barrier();
        LT_TICK(1);
        // dh_{prev} = dgates . W_hh: 16 k-steps over the 512 gate rows, in FOUR accumulators (k-step ks -> accumulator ks & 3, summed at
        // the end): one accumulator is a chain of 48 dependent MFMAs (two-plane mode), each waiting for the one before it -- most of
        // a backward step
        f32x4_t acc4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc4[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4) {
            uint4 gf[4][NS];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    gf[i][p] = *reinterpret_cast<const uint4*>(gt + p * GBUF_BYTES + ((lane & 15) * GP + 32 * (k4 + i) + 8 * (lane >> 4)) * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc4[i] = mfma16(gf[i][0], wt[k4 + i][0], acc4[i]);
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc4[i] = mfma16(gf[i][NS - 1], wt[k4 + i][0], acc4[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc4[i] = mfma16(gf[i][0], wt[k4 + i][NS - 1], acc4[i]);
            }
        }
        dh_rec = (acc4[0][0] + acc4[1][0]) + (acc4[2][0] + acc4[3][0]);
#ifdef AMTX_LSTM_TIMING
        asm volatile("" ::"v"(dh_rec));
        LT_TICK(2);
        lt_acc[3] += 1;
#endif
    }
#ifdef AMTX_LSTM_TIMING
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&g_lstm_prof[i], lt_acc[i]);
#endif
}
#ifdef AMTX_LSTM_TIMING
extern "C" int amtxdbg_lstm_prof(unsigned long long* out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_lstm_prof), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// fp32 W_hh (512 x 128, both directions) on the DEVICE -> forward fragments (amtx_bilstm_pack_host's layout) and transposed
// fragments for the backward kernel, hi/lo planes: training repacks after every optimizer step without a host round trip.
__global__ void bilstm_pack_dev_kernel(const float* __restrict__ whh_fwd, const float* __restrict__ whh_bwd, int planes,
                                       bf16_t* __restrict__ frag_fwd, bf16_t* __restrict__ frag_bwd) {
    const int n = 2 * 4 * H * H;                                   // elements per plane set (both directions)
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        // forward fragment element: [dir][w][q][ks][l][j]
        {
            int r = idx;
            const int j = r & 7; r >>= 3;
            const int l = r & 63; r >>= 6;
            const int ks = r & 3; r >>= 2;
            const int q = r & 3; r >>= 2;
            const int w = r & 7; r >>= 3;
            const int dir = r;
            const float* W = dir == 0 ? whh_fwd : whh_bwd;
            const float v = W[(q * H + 16 * w + (l & 15)) * H + 32 * ks + 8 * (l >> 4) + j];
            const bf16_t hi = f32_to_bf16_rn(v);
            const size_t base = ((size_t)((((dir * LWAVES + w) * 4 + q) * 4 + ks) * planes)) * 512 + (size_t)l * 8 + j;
            frag_fwd[base] = hi;
            if (planes == 2) frag_fwd[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
        }
        // transposed fragment element: [dir][w][ks(16)][l][j]: W[row k = 32 ks + 8 (l >> 4) + j][unit 16 w + (l & 15)]
        {
            int r = idx;
            const int j = r & 7; r >>= 3;
            const int l = r & 63; r >>= 6;
            const int ks = r & 15; r >>= 4;
            const int w = r & 7; r >>= 3;
            const int dir = r;
            const float* W = dir == 0 ? whh_fwd : whh_bwd;
            const float v = W[(32 * ks + 8 * (l >> 4) + j) * H + 16 * w + (l & 15)];
            const bf16_t hi = f32_to_bf16_rn(v);
            const size_t base = ((size_t)(((dir * LWAVES + w) * 16 + ks) * planes)) * 512 + (size_t)l * 8 + j;
            frag_bwd[base] = hi;
            if (planes == 2) frag_bwd[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Any hidden size HH = 128 k (OnsetsFrames at model_complexity 3: 256, onsetsframes.py:57-58 dim_lm = 256 (mc - 1)).
// 4 HH x HH bf16 recurrent weights of one direction (512 KiB at HH = 256) are as large as a CU's whole register file, so
// they cannot all be stationary: the fragments of a wave's unit tiles form 4-KiB groups (16 VGPRs); in the bf16 mode every
// fourth group is pinned in registers, every fourth in LDS, the other half is streamed from L2 every step (all blocks of a
// launch read the same <= 2 MiB, which stays cache-resident) through a four-slot register ring, three groups ahead of the MFMAs,
// the first groups of the next step already in flight across the step barrier (two-plane mode: everything streamed).  One
// block = one direction x 16 clips x all T steps, eight waves; product swapped (D' = W_hh . h^T) so the four gates of a
// (clip, unit) meet in one lane; h_{t-1} exchanged through a double-buffered LDS tile; c in registers.  With `save` set it is
// also the training forward (post-activation gates and cell states written per step).  Step time at HH = 256: 5.3 us bf16
// (7.3 us all-streamed), 10 us two-plane: T dependent steps per launch, latency-bound like the register-stationary kernels,
// only with a longer step.
// compile-time loop: the body sees its index as a constant expression (register arrays indexed through it stay in registers;
// with a plain unrolled loop and computed indices hipcc left the weight arrays in scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int HH, int NS, int X_TYPE, int OUT_TYPE>
__global__ __launch_bounds__(512) void bilstm_stream_kernel(LstmArgs a) {
    constexpr int NU = HH / 16, UPW = NU / 8, KSN = HH / 32;
    constexpr int SGK = NS == 2 ? 2 : 4;          // k-steps per streamed group: 16 VGPRs per group in either precision
    constexpr int NSUB = KSN / SGK;               // groups per (unit tile, gate)
    constexpr int NG = UPW * 4 * NSUB;            // groups per step and wave
    constexpr int HPG = HH + 8;
    constexpr int HBG = 16 * HPG * 2;
    constexpr bool FAST = true;
    constexpr int RS = 4;                         // ring slots: RS - 1 streamed groups (4 KiB each per wave) in flight (8 slots: no faster,
                                                  // the stream is bound by the 64 B/clk a CU's vector memory path takes, not by latency)
    // HH = 256, bf16: of every four consecutive groups of a wave, group 0 is held in registers and group 1 in LDS for the whole launch
    // (64 spare VGPRs, 128 KiB of LDS per block), the other two are streamed: half the bytes per step.  HH = 384, bf16: 36 groups per
    // wave, no spare registers; every ninth group (4 per wave, again 128 KiB per block) sits in LDS, 32 are streamed.
    constexpr bool PIN = NS == 1 && HH == 256;    // register + LDS pinning, period 4
    constexpr bool PINL = NS == 1 && HH == 384;   // LDS pinning only: four groups per wave, period PER = 9 (HH = 512: the 128 KiB of pinned groups
                                                  // and the two h tiles are 512 bytes more than a CU's LDS: everything streamed)
    constexpr int PER = NG / 4;
    constexpr int NPIN = PIN ? NG / 4 : (PINL ? 4 : 0);   // groups pinned per kind and wave
    constexpr int NSG = NG - (PIN ? 2 : (PINL ? 1 : 0)) * NPIN;   // streamed groups per step and wave
    static_assert(NU % 8 == 0 && KSN % SGK == 0 && NG % 4 == 0 && NSG % RS == 0, "hidden size must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][NS planes][16][HPG] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 16, dir = blockIdx.y, grp = blockIdx.z;
    const int b = b0 + clip;
    const bool clip_ok = b < a.B;
    const int T = a.T;
    constexpr int XES = X_TYPE == AMTX_T_BF16 ? 2 : 4, OES = OUT_TYPE == AMTX_T_BF16 ? 2 : 4;

    for (int i = tid; i < 2 * NS * HBG / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    // fragment (u, q, ks, p) of direction dir: uint4 index ((((dir * NU + u) * 4 + q) * KSN + ks) * NS + p) * 64 + lane;
    // a wave's groups of one step are consecutive in memory
    // wave-uniform base (scalar registers) + one per-lane byte offset: every fragment load is `global_load saddr + voffset`, without
    // a 64-bit address pair per load held in vector registers across the loop
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* wbase = reinterpret_cast<const char*>(a.whh + (int64_t)grp * a.w_gs) + ((int64_t)(dir * NU + wave_u * UPW) * 4 * KSN * NS) * 1024;
    const unsigned wlane = lane * 16;
    const char* xbase = reinterpret_cast<const char*>(a.xproj) +
                        ((int64_t)grp * a.x_gs + (int64_t)(clip_ok ? b : 0) * T * 8 * HH + dir * 4 * HH + 16 * UPW * wave + 4 * g) * XES;
    char* obase = reinterpret_cast<char*>(a.out) +
                  ((int64_t)grp * a.out_gs + (int64_t)(clip_ok ? b : 0) * T * 2 * HH + dir * HH + 16 * UPW * wave + 4 * g) * OES;

    float c[UPW][4];
#pragma unroll
    for (int ub = 0; ub < UPW; ++ub)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[ub][r] = 0.f;

    typedef typename XRaw<X_TYPE>::type xraw_t;
    auto tidx = [&](int s) { s = s < T ? s : T - 1; return (int64_t)(dir == 0 ? s : T - 1 - s); };
    auto load_xrow = [&](int64_t t, xraw_t (&dst)[UPW][4]) {
#pragma unroll
        for (int ub = 0; ub < UPW; ++ub)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[ub][q] = *reinterpret_cast<const xraw_t*>(xbase + (t * 8 * HH + q * HH + 16 * ub) * XES);
    };
    uint4 w[RS][SGK][NS];                         // ring: streamed group si lives in slot si % RS, RS - 1 groups in flight
    uint4 wpin[PIN ? NPIN : 1][SGK][NS];          // register-resident groups
    // group gi -> kind (0 registers, 1 LDS, 2 streamed) and index within its kind: see kind_of / sidx_of / gi_of_sidx below
    char* wlds = smem + 2 * NS * HBG + wave * (NPIN * SGK * NS * 1024);      // this wave's LDS-resident groups
    typedef const __attribute__((address_space(1))) char* gchar_p;      // explicitly global: the asm below must not turn the loads into flat ones
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    typedef __attribute__((address_space(1))) u32x4_t gu4_t;
    gchar_p wb = (gchar_p)wbase;
    auto load_frag = [&](int gi, int k, int p) {
        return __builtin_bit_cast(uint4, *reinterpret_cast<const gu4_t*>(wb + ((gi * SGK + k) * NS + p) * 1024 + wlane));
    };
    if constexpr (PIN || PINL) {
        static_for<0, NG>([&](auto ic) {
            constexpr int gi = decltype(ic)::value;
            constexpr int kind = PIN ? ((gi & 3) == 0 ? 0 : ((gi & 3) == 1 ? 1 : 2)) : (gi % PER == 0 ? 1 : 2);
            constexpr int pidx = PIN ? gi >> 2 : gi / PER;
            if constexpr (kind < 2) {
                static_for<0, SGK>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    static_for<0, NS>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        const uint4 v = load_frag(gi, k, p);
                        if constexpr (kind == 0) wpin[pidx][k][p] = v;
                        else *reinterpret_cast<uint4*>(wlds + ((pidx * SGK + k) * NS + p) * 1024 + wlane) = v;
                    });
                });
            }
        });
    }

    xraw_t xn[UPW][4];
    load_xrow(tidx(0), xn);
    static_for<0, RS - 1>([&](auto sc) {
        constexpr int si = decltype(sc)::value;
        constexpr int gi = PIN ? (si >> 1) * 4 + 2 + (si & 1) : (PINL ? si + si / (PER - 1) + 1 : si);
        static_for<0, SGK>([&](auto kc) {
            static_for<0, NS>([&](auto pc) { w[si % RS][decltype(kc)::value][decltype(pc)::value] = load_frag(gi, decltype(kc)::value, decltype(pc)::value); });
        });
    });
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < T; ++s) {
        // opaque to the optimiser: otherwise it hoists one 64-bit vector address per fragment load out of the loop (128 VGPRs)
        wb = (gchar_p)wbase;
        asm volatile("" : "+s"(wb));
        const char* hb = smem + cur * NS * HBG;
        // the h fragments of a step: in registers while they are at most 64 (HH = 384 in two planes would be 96: read at their use)
        constexpr bool HF_REG = KSN * NS <= 16;
        uint4 hf[HF_REG ? KSN : 1][NS];
        if constexpr (HF_REG) {
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                for (int p = 0; p < NS; ++p) hf[ks][p] = *reinterpret_cast<const uint4*>(hb + p * HBG + (clip * HPG + 32 * ks + 8 * g) * 2);
        }

        f32x4_t acc[UPW][4];
#pragma unroll
        for (int ub = 0; ub < UPW; ++ub)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[ub][q] = unpack_x(xn[ub][q]);
        load_xrow(tidx(s + 1), xn);
        __builtin_amdgcn_sched_barrier(0);

        static_for<0, NG>([&](auto ic) {
            constexpr int gi = decltype(ic)::value;
            constexpr int ub = gi / (4 * NSUB), q = (gi / NSUB) & 3, sub = gi % NSUB;
            constexpr int kind = PIN ? ((gi & 3) == 0 ? 0 : ((gi & 3) == 1 ? 1 : 2)) : (PINL ? (gi % PER == 0 ? 1 : 2) : 2);     // 0 registers, 1 LDS, 2 streamed
            constexpr int pidx = PIN ? gi >> 2 : gi / PER;                                      // index among the pinned groups of its kind
            constexpr int si = PIN ? (gi >> 2) * 2 + (gi & 3) - 2 : (PINL ? gi - gi / PER - 1 : gi);   // index among the streamed groups
            if constexpr (kind == 2) {
                // behind the last RS - 1 streamed groups: the first ones of the next step
                constexpr int sn = (si + RS - 1) % NSG;
                constexpr int gn = PIN ? (sn >> 1) * 4 + 2 + (sn & 1) : (PINL ? sn + sn / (PER - 1) + 1 : sn);
                static_for<0, SGK>([&](auto kc) {
                    static_for<0, NS>([&](auto pc) { w[sn % RS][decltype(kc)::value][decltype(pc)::value] = load_frag(gn, decltype(kc)::value, decltype(pc)::value); });
                });
            }
            f32x4_t d = acc[ub][q];
            static_for<0, SGK>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int ks = sub * SGK + k;
                uint4 w0, w1;
                if constexpr (kind == 0) {
                    w0 = wpin[pidx][k][0]; w1 = wpin[pidx][k][NS - 1];
                } else if constexpr (kind == 1) {
                    w0 = *reinterpret_cast<const uint4*>(wlds + ((pidx * SGK + k) * NS + 0) * 1024 + wlane);
                    w1 = w0;
                } else {
                    w0 = w[si % RS][k][0]; w1 = w[si % RS][k][NS - 1];
                }
                uint4 h0, h1;
                if constexpr (HF_REG) {
                    h0 = hf[ks][0]; h1 = hf[ks][NS - 1];
                } else {
                    h0 = *reinterpret_cast<const uint4*>(hb + (clip * HPG + 32 * ks + 8 * g) * 2);
                    h1 = *reinterpret_cast<const uint4*>(hb + (NS - 1) * HBG + (clip * HPG + 32 * ks + 8 * g) * 2);
                }
                d = mfma16(w0, h0, d);
                if constexpr (NS == 2) {
                    d = mfma16(w0, h1, d);
                    d = mfma16(w1, h0, d);
                }
            });
            acc[ub][q] = d;
            __builtin_amdgcn_sched_barrier(0);   // keep the ring as deep as written: no hoisting of later loads above this group
        });

        char* hn = smem + (cur ^ 1) * NS * HBG;
        const int64_t t = tidx(s);
#pragma unroll
        for (int ub = 0; ub < UPW; ++ub) {
            float h[4], sv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = sigmoid_f<FAST>(acc[ub][0][r]);
                const float fg = sigmoid_f<FAST>(acc[ub][1][r]);
                const float gg = tanh_f<FAST>(acc[ub][2][r]);
                const float og = sigmoid_f<FAST>(acc[ub][3][r]);
                c[ub][r] = fg * c[ub][r] + ig * gg;
                h[r] = og * tanh_f<FAST>(c[ub][r]);
                sv[0][r] = ig; sv[1][r] = fg; sv[2][r] = gg; sv[3][r] = og;
            }
            if (a.save && clip_ok) {                          // training: post-activation gates and the cell state, [B][T][2][5][HH]
                float* sp = a.save + ((((int64_t)grp * a.B + b) * T + t) * 2 + dir) * (5 * HH) + 16 * UPW * wave + 16 * ub + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sp + q * HH) = make_float4(sv[q][0], sv[q][1], sv[q][2], sv[q][3]);
                *reinterpret_cast<float4*>(sp + 4 * HH) = make_float4(c[ub][0], c[ub][1], c[ub][2], c[ub][3]);
            }
            uint2 hiw, low = make_uint2(0, 0);
            if (NS == 2) {
                split_bf16x2(h[0], h[1], hiw.x, low.x);
                split_bf16x2(h[2], h[3], hiw.y, low.y);
            } else {
                hiw = make_uint2(pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]));
            }
            const int hoff = (clip * HPG + 16 * UPW * wave + 16 * ub + 4 * g) * 2;
            *reinterpret_cast<uint2*>(hn + hoff) = hiw;
            if (NS == 2) *reinterpret_cast<uint2*>(hn + HBG + hoff) = low;
            if (clip_ok) {
                const int64_t e = t * 2 * HH + 16 * ub;
                if (OUT_TYPE == AMTX_T_BF16) *reinterpret_cast<uint2*>(obase + e * 2) = hiw;
                else *reinterpret_cast<float4*>(obase + e * 4) = make_float4(h[0], h[1], h[2], h[3]);
            }
        }
        cur ^= 1;
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward recurrence for any hidden size HH = 128 k (training at model_complexity 3): dout [B][T][2 HH], the forward's saved
// gates / cell states -> dxproj [B][T][2][4 HH].  Lane mapping of bilstm4_bwd_kernel (four clips per block in rows 0/4/8/12 of the
// dgates tile, lane = (unit, clip), product not swapped), but the 4 HH x HH recurrent weights (1 MiB in two planes at HH = 256)
// are streamed from L2 every step through the same four-slot register ring as the forward stream kernel; every wave owns
// HH / 8 hidden units = UT unit tiles.
struct LstmBwdHArgs {
    const float* dout; const float* save; const bf16_t* whh_t; float* dxproj; int B, T;
    int64_t w_gs;      // fragment elements per LSTM: blockIdx.z walks independent LSTMs of the same (B, T) (dout / save / dxproj are [groups][B]...)
};

template <int HH, int NS>
__global__ __launch_bounds__(512) void bilstm_stream_bwd_kernel(LstmBwdHArgs a) {
    constexpr int UT = HH / 128;                   // unit tiles per wave
    constexpr int KSN = 4 * HH / 32;               // k-steps over the gate rows
    constexpr int SGK = NS == 2 ? 2 : 4;
    constexpr int NG = UT * KSN / SGK;             // fragment groups (16 VGPRs each) per step and wave
    constexpr int RS = 4;
    // every fourth group is pinned in registers for the whole launch (the kernel needs ~106 VGPRs of the 256 it may use), the
    // other three are streamed: a quarter fewer bytes per step
    constexpr bool PINB = HH == 256;               // (HH = 384: 18 pinned groups would be 288 VGPRs: all streamed)
    constexpr int NPINB = PINB ? NG / 4 : 0, NSGB = NG - NPINB;
    constexpr int GPH = 4 * HH + 8;                // bf16 elements per row of the dgates tile
    // one plane of one buffer: the four clips' rows + ONE zero row that stands for the twelve unused rows of the 16-row MFMA operand
    // (all 16 rows: 197 KiB at HH = 384 in two planes)
    constexpr int GBH = 5 * GPH * 2;
    static_assert(NG % 4 == 0 && NSGB % RS == 0, "hidden size must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][NS planes][5][GPH] bf16: clip rows 0..3, zero row 4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane >> 4;
    const int dir = blockIdx.y;
    const int b = blockIdx.x * 4 + cg;
    const bool clip_ok = b < a.B;
    const int T = a.T;
    const int unit0 = 16 * UT * wave + (lane & 15);               // + 16 ut

    for (int i = tid; i < 2 * NS * GBH / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    // transposed fragments: uint4 index ((((dir * 8 + wave) * UT + ut) * KSN + ks) * NS + p) * 64 + lane
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int grp = blockIdx.z;
    const char* wbase = reinterpret_cast<const char*>(a.whh_t + (int64_t)grp * a.w_gs) + ((int64_t)(dir * 8 + wave_u) * UT * KSN * NS) * 1024;
    const unsigned wlane = lane * 16;
    typedef const __attribute__((address_space(1))) char* gchar_p;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    typedef __attribute__((address_space(1))) u32x4_t gu4_t;
    gchar_p wb = (gchar_p)wbase;
    uint4 w[RS][SGK][NS];
    auto load_frag = [&](int gi, int k, int p) {
        return __builtin_bit_cast(uint4, *reinterpret_cast<const gu4_t*>(wb + ((gi * SGK + k) * NS + p) * 1024 + wlane));
    };

    const int64_t bb = (int64_t)grp * a.B + (clip_ok ? b : 0);
    const float* sv0 = a.save + ((bb * T) * 2 + dir) * (5 * HH) + unit0;          // + t * (2*5*HH) + 16 ut + q * HH
    const float* do0 = a.dout + (bb * T) * (2 * HH) + dir * HH + unit0;            // + t * 2 HH + 16 ut
    float* dx0 = a.dxproj + (bb * T) * (8 * HH) + dir * 4 * HH + unit0;            // + t * 8 HH + q * HH + 16 ut
    const int gwoff = (cg * GPH + unit0) * 2;                                       // byte offset of (clip row, gate 0, ut 0) in a tile plane
    // fragment row (lane & 15) of the operand: rows 0 / 4 / 8 / 12 are the clips, every other row reads the zero row
    const int groff = ((((lane & 15) & 3) == 0 ? (lane & 15) >> 2 : 4) * GPH + 8 * (lane >> 4)) * 2;

    float dh_rec[UT], dc_rec[UT];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) { dh_rec[ut] = 0.f; dc_rec[ut] = 0.f; }
    auto frame = [&](int s_) { s_ = s_ < T ? s_ : T - 1; return dir == 0 ? T - 1 - s_ : s_; };
    float nv[UT][6];                                                              // i, f, g, o, c, dout of the next step
    auto load_next = [&](int tn) {
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            const float* sv = sv0 + (int64_t)tn * (2 * 5 * HH) + 16 * ut;
#pragma unroll
            for (int q = 0; q < 5; ++q) nv[ut][q] = sv[q * HH];
            nv[ut][5] = do0[(int64_t)tn * (2 * HH) + 16 * ut];
        }
    };
    load_next(frame(0));
    uint4 wpin[PINB ? NPINB : 1][SGK][NS];
    static_for<0, NPINB>([&](auto gc) {
        static_for<0, SGK>([&](auto kc) {
            static_for<0, NS>([&](auto pc) { wpin[decltype(gc)::value][decltype(kc)::value][decltype(pc)::value] = load_frag(4 * decltype(gc)::value, decltype(kc)::value, decltype(pc)::value); });
        });
    });
    static_for<0, RS - 1>([&](auto sc) {
        constexpr int si = decltype(sc)::value;
        constexpr int gi = PINB ? (si / 3) * 4 + 1 + si % 3 : si;        // streamed group si -> group index
        static_for<0, SGK>([&](auto kc) {
            static_for<0, NS>([&](auto pc) { w[si % RS][decltype(kc)::value][decltype(pc)::value] = load_frag(gi, decltype(kc)::value, decltype(pc)::value); });
        });
    });
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        wb = (gchar_p)wbase;
        asm volatile("" : "+s"(wb));                          // see bilstm_stream_kernel: keeps the fragment addresses scalar
        const int t = frame(s);
        float cur[UT][6];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int q = 0; q < 6; ++q) cur[ut][q] = nv[ut][q];
        load_next(frame(s + 1));
        char* gt = smem + (s & 1) * NS * GBH;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            asm volatile("" : "+v"(nv[ut][4]));
            const float cp = s + 1 < T ? nv[ut][4] : 0.f;     // cell state before step t in forward order (zero initial state)
            const float ig = cur[ut][0], fg = cur[ut][1], gg = cur[ut][2], og = cur[ut][3], ct = cur[ut][4];
            const float dh = cur[ut][5] + dh_rec[ut];
            const float tc = tanh_f<true>(ct);
            const float d_o = dh * tc * og * (1.f - og);
            const float dc = dc_rec[ut] + dh * og * (1.f - tc * tc);
            const float d_i = dc * gg * ig * (1.f - ig);
            const float d_g = dc * ig * (1.f - gg * gg);
            const float d_f = dc * cp * fg * (1.f - fg);
            dc_rec[ut] = dc * fg;
            const float dgv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (clip_ok) dx0[(int64_t)t * (8 * HH) + q * HH + 16 * ut] = dgv[q];
                uint32_t hiw, low = 0;
                if (NS == 2) split_bf16x2(dgv[q], 0.f, hiw, low);
                else hiw = pack_bf16x2(dgv[q], 0.f);
                *reinterpret_cast<unsigned short*>(gt + gwoff + (q * HH + 16 * ut) * 2) = (unsigned short)hiw;
                if (NS == 2) *reinterpret_cast<unsigned short*>(gt + GBH + gwoff + (q * HH + 16 * ut) * 2) = (unsigned short)low;
            }
        }
        lds_barrier();
        // dh_prev[unit] = sum over the 4 HH gate rows of dgates[clip][row] * W_hh[row][unit]
        f32x4_t acc[UT];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) acc[ut] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        static_for<0, NG>([&](auto ic) {
            constexpr int gi = decltype(ic)::value;
            constexpr int ut = gi / (KSN / SGK), sub = gi % (KSN / SGK);
            constexpr bool pinned = PINB && (gi & 3) == 0;
            constexpr int si = PINB ? (gi >> 2) * 3 + (gi & 3) - 1 : gi;         // index among the streamed groups
            if constexpr (!pinned) {
                constexpr int sn = (si + RS - 1) % NSGB;                         // behind the last RS - 1 streamed groups
                constexpr int gn = PINB ? (sn / 3) * 4 + 1 + sn % 3 : sn;
                static_for<0, SGK>([&](auto kc) {
                    static_for<0, NS>([&](auto pc) { w[sn % RS][decltype(kc)::value][decltype(pc)::value] = load_frag(gn, decltype(kc)::value, decltype(pc)::value); });
                });
            }
            static_for<0, SGK>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int ks = sub * SGK + k;
                uint4 gf[NS];
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    gf[p] = *reinterpret_cast<const uint4*>(gt + p * GBH + groff + 32 * ks * 2);
                uint4 w0, w1;
                if constexpr (pinned) { w0 = wpin[gi >> 2][k][0]; w1 = wpin[gi >> 2][k][NS - 1]; }
                else { w0 = w[si % RS][k][0]; w1 = w[si % RS][k][NS - 1]; }
                acc[ut] = mfma16(gf[0], w0, acc[ut]);
                if constexpr (NS == 2) {
                    acc[ut] = mfma16(gf[NS - 1], w0, acc[ut]);
                    acc[ut] = mfma16(gf[0], w1, acc[ut]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) dh_rec[ut] = acc[ut][0];
    }
}

// fp32 W_hh (4 hidden x hidden, both directions) on the DEVICE -> forward fragments of bilstm_stream_kernel and transposed
// fragments of bilstm_stream_bwd_kernel, hi/lo planes
__global__ void bilstm_pack_dev_h_kernel(const float* __restrict__ whh_fwd, const float* __restrict__ whh_bwd, int hidden, int planes,
                                         bf16_t* __restrict__ frag_fwd, bf16_t* __restrict__ frag_bwd) {
    const int HHr = hidden, nu = HHr / 16, ksn = HHr / 32, ut_n = HHr / 128, ksn_b = 4 * HHr / 32;
    const int n = 2 * 4 * HHr * HHr;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        {   // forward: [dir][u][q][ks][p][l][j]
            int r = idx;
            const int j = r & 7; r >>= 3;
            const int l = r & 63; r >>= 6;
            const int ks = r % ksn; r /= ksn;
            const int q = r & 3; r >>= 2;
            const int u = r % nu; r /= nu;
            const int dir = r;
            const float* W = dir == 0 ? whh_fwd : whh_bwd;
            const float v = W[(size_t)(q * HHr + 16 * u + (l & 15)) * HHr + 32 * ks + 8 * (l >> 4) + j];
            const bf16_t hi = f32_to_bf16_rn(v);
            const size_t base = ((((size_t)(dir * nu + u) * 4 + q) * ksn + ks) * planes) * 512 + (size_t)l * 8 + j;
            frag_fwd[base] = hi;
            if (planes == 2) frag_fwd[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
        }
        {   // transposed: [dir][wave][ut][ks][p][l][j] = W[row 32 ks + 8 (l >> 4) + j][unit 16 ut_n wave + 16 ut + (l & 15)]
            int r = idx;
            const int j = r & 7; r >>= 3;
            const int l = r & 63; r >>= 6;
            const int ks = r % ksn_b; r /= ksn_b;
            const int ut = r % ut_n; r /= ut_n;
            const int wv = r & 7; r >>= 3;
            const int dir = r;
            const float* W = dir == 0 ? whh_fwd : whh_bwd;
            const float v = W[(size_t)(32 * ks + 8 * (l >> 4) + j) * HHr + 16 * ut_n * wv + 16 * ut + (l & 15)];
            const bf16_t hi = f32_to_bf16_rn(v);
            const size_t base = ((((size_t)(dir * 8 + wv) * ut_n + ut) * ksn_b + ks) * planes) * 512 + (size_t)l * 8 + j;
            frag_bwd[base] = hi;
            if (planes == 2) frag_bwd[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
        }
    }
}

template <int HH, int NS, int X_TYPE, int OUT_TYPE>
int launch_stream(const LstmArgs& a, hipStream_t stream) {
    // h tiles + (bf16 mode) the LDS-resident quarter of W_hh: 8 waves x (groups / 4) x 4 KiB
    // h tiles + (bf16 mode) the LDS-resident groups of W_hh: 8 waves x 4 groups x 4 KiB at either hidden size
    const size_t lds = 2 * (size_t)NS * 16 * (HH + 8) * 2 + (NS == 1 && HH <= 384 ? (size_t)8 * 4 * 4096 : 0);
    auto kern = bilstm_stream_kernel<HH, NS, X_TYPE, OUT_TYPE>;
    AMTX_GRANT_LDS(kern, lds);
    dim3 grid((unsigned)((a.B + 15) / 16), 2, (unsigned)a.groups);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

template <int HH>
int dispatch_stream(const LstmArgs& a, hipStream_t stream) {
    if (a.planes == 1 && a.x_type == AMTX_T_BF16 && a.out_type == AMTX_T_BF16) return launch_stream<HH, 1, AMTX_T_BF16, AMTX_T_BF16>(a, stream);
    if (a.planes == 2 && a.x_type == AMTX_T_F32 && a.out_type == AMTX_T_F32) return launch_stream<HH, 2, AMTX_T_F32, AMTX_T_F32>(a, stream);
    amtx_set_error("bilstm (hidden %d): unsupported precision/type combination", HH);
    return AMTX_ERR_UNSUPPORTED;
}

// clips per block: 4 while that still leaves fewer blocks than ~4 per CU, else 16 (a quarter of the total wave-steps)
inline bool use_four_clip_blocks(const LstmArgs& a) { return (int64_t)((a.B + 3) / 4) * 2 * a.groups <= 1024; }

template <int NS, int X_TYPE, int OUT_TYPE>
int launch(const LstmArgs& a, hipStream_t stream) {
    const size_t lds = 2 * NS * HBUF_BYTES;
    // Eight clips per block once four-clip blocks would outnumber the CUs: two co-resident blocks share a SIMD's matrix pipe and issue
    // slots and a step takes 1650 cycles instead of the 1070 a block has to itself (0.54 vs 0.35 ms per 625 steps, measured at 1024 and
    // 512 clips); the eight-clip block does the same 16 MFMAs per wave and step for twice the clips, one block per CU.
    static const bool no8 = getenv("AMTX_LSTM_NO8") != nullptr;       // A/B switch
    const int64_t blocks4 = (int64_t)((a.B + 3) / 4) * 2 * a.groups;
    // (one plane only: in the two-plane mode a step is 48 MFMAs per wave and the eight-clip block's doubled gate arithmetic sits on top of
    // them -- 1.64 ms per 625 steps at 1024 clips against 1.25 for two co-resident four-clip blocks, round 5)
    if (!a.save && !no8 && NS == 1 && blocks4 > 256 && use_four_clip_blocks(a)) {
        dim3 grid((unsigned)((a.B + 7) / 8), 2, (unsigned)a.groups);
        hipLaunchKernelGGL((bilstm4_kernel<NS, X_TYPE, OUT_TYPE, 2>), grid, dim3(LTHREADS), lds, stream, a);
    } else if (use_four_clip_blocks(a) || a.save) {     // the training forward (save != null) exists for the 4-clip mapping only
        dim3 grid((unsigned)((a.B + 3) / 4), 2, (unsigned)a.groups);
        hipLaunchKernelGGL((bilstm4_kernel<NS, X_TYPE, OUT_TYPE>), grid, dim3(LTHREADS), lds, stream, a);
    } else {
        dim3 grid((unsigned)((a.B + 15) / 16), 2, (unsigned)a.groups);
        hipLaunchKernelGGL((bilstm_kernel<NS, X_TYPE, OUT_TYPE>), grid, dim3(LTHREADS), lds, stream, a);
    }
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

}  // namespace

size_t amtx_bilstm_wfrag_elems(int planes) { return (size_t)2 * 512 * 128 * planes; }

size_t amtx_bilstm_wfrag_elems_h(int hidden, int planes) { return (size_t)2 * 4 * hidden * hidden * planes; }

// hidden != 128: fragment order of bilstm_stream_kernel, [dir][unit tile][gate][k-step][plane][lane][8]
void amtx_bilstm_pack_host_h(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, bf16_t* out) {
    if (hidden == H) { amtx_bilstm_pack_host(whh_fwd, whh_bwd, planes, out); return; }
    const int nu = hidden / 16, ksn = hidden / 32;
    for (int dir = 0; dir < 2; ++dir) {
        const float* W = dir == 0 ? whh_fwd : whh_bwd;   // (4 hidden, hidden) row-major, gate-major rows i,f,g,o
        for (int u = 0; u < nu; ++u)
            for (int q = 0; q < 4; ++q)
                for (int ks = 0; ks < ksn; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int row = q * hidden + 16 * u + (l & 15);
                            const int k = 32 * ks + 8 * (l >> 4) + j;
                            const float v = W[(size_t)row * hidden + k];
                            const bf16_t hi = f32_to_bf16_rn(v);
                            const size_t base = (((((size_t)dir * nu + u) * 4 + q) * ksn + ks) * planes) * 512 + (size_t)l * 8 + j;
                            out[base] = hi;
                            if (planes == 2) out[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                        }
    }
}

void amtx_bilstm_pack_host(const float* whh_fwd, const float* whh_bwd, int planes, bf16_t* out) {
    for (int dir = 0; dir < 2; ++dir) {
        const float* W = dir == 0 ? whh_fwd : whh_bwd;   // (512, 128) row-major, gate-major rows i,f,g,o
        for (int w = 0; w < LWAVES; ++w)
            for (int ub = 0; ub < UB; ++ub)
                for (int q = 0; q < 4; ++q)
                    for (int ks = 0; ks < 4; ++ks)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int row = q * 128 + 16 * UB * w + 16 * ub + (l & 15);
                                const int k = 32 * ks + 8 * (l >> 4) + j;
                                const float v = W[(size_t)row * H + k];
                                const bf16_t hi = f32_to_bf16_rn(v);
                                const size_t base = ((((((size_t)dir * LWAVES + w) * UB + ub) * 4 + q) * 4 + ks) * planes) * 512 + (size_t)l * 8 + j;
                                out[base] = hi;
                                if (planes == 2) out[base + 512] = f32_to_bf16_rn(v - bf16_to_f32(hi));
                            }
    }
}

int amtx_launch_bilstm(const LstmArgs& a, hipStream_t stream) {
    AMTX_REQUIRE(a.xproj && a.whh && a.out, "bilstm: null pointer");
    AMTX_REQUIRE(a.B > 0 && a.T > 0 && a.groups > 0, "bilstm: bad sizes");
    AMTX_REQUIRE(a.planes == 1 || a.planes == 2, "bilstm: planes must be 1 or 2");
    if (a.hidden != H) {
        AMTX_REQUIRE(!a.save || a.planes == 2, "bilstm: the training forward (save) is built for the two-plane precision");
        if (a.hidden == 256) return dispatch_stream<256>(a, stream);
        if (a.hidden == 384) return dispatch_stream<384>(a, stream);
        if (a.hidden == 512) return dispatch_stream<512>(a, stream);
        amtx_set_error("bilstm: unsupported hidden size %d (128, 256, 384 and 512 are built)", a.hidden);
        return AMTX_ERR_UNSUPPORTED;
    }
    if (a.planes == 1 && a.x_type == AMTX_T_BF16 && a.out_type == AMTX_T_BF16) return launch<1, AMTX_T_BF16, AMTX_T_BF16>(a, stream);
    if (a.planes == 1 && a.x_type == AMTX_T_F32 && a.out_type == AMTX_T_F32) return launch<1, AMTX_T_F32, AMTX_T_F32>(a, stream);
    if (a.planes == 2 && a.x_type == AMTX_T_F32 && a.out_type == AMTX_T_F32) return launch<2, AMTX_T_F32, AMTX_T_F32>(a, stream);
    amtx_set_error("bilstm: unsupported precision/type combination");
    return AMTX_ERR_UNSUPPORTED;
}

int amtx_launch_bilstm_pack_dev(const float* whh_fwd, const float* whh_bwd, int planes, bf16_t* frag_fwd, bf16_t* frag_bwd, hipStream_t stream) {
    AMTX_REQUIRE(whh_fwd && whh_bwd && frag_fwd && frag_bwd && (planes == 1 || planes == 2), "bilstm pack: bad argument");
    hipLaunchKernelGGL(bilstm_pack_dev_kernel, dim3(128), dim3(256), 0, stream, whh_fwd, whh_bwd, planes, frag_fwd, frag_bwd);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_launch_bilstm_bwd(const float* dout, const float* save, const bf16_t* whh_t, int planes, float* dxproj, int B, int T, hipStream_t stream) {
    AMTX_REQUIRE(dout && save && whh_t && dxproj, "bilstm backward: null pointer");
    AMTX_REQUIRE(B > 0 && T > 0 && (planes == 1 || planes == 2), "bilstm backward: bad sizes");
    LstmBwdArgs a;
    a.dout = dout; a.save = save; a.whh_t = whh_t; a.planes = planes; a.dxproj = dxproj; a.B = B; a.T = T;
    dim3 grid((unsigned)((B + 3) / 4), 2);
    const size_t lds = 2 * (size_t)planes * GBUF_BYTES;
    if (planes == 2) {
        AMTX_GRANT_LDS(bilstm4_bwd_kernel<2>, lds);
        hipLaunchKernelGGL(bilstm4_bwd_kernel<2>, grid, dim3(LTHREADS), lds, stream, a);
    } else {
        hipLaunchKernelGGL(bilstm4_bwd_kernel<1>, grid, dim3(LTHREADS), lds, stream, a);
    }
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_launch_bilstm_pack_dev_h(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, bf16_t* frag_fwd, bf16_t* frag_bwd, hipStream_t stream) {
    AMTX_REQUIRE(whh_fwd && whh_bwd && frag_fwd && frag_bwd && (planes == 1 || planes == 2), "bilstm pack: bad argument");
    if (hidden == H) return amtx_launch_bilstm_pack_dev(whh_fwd, whh_bwd, planes, frag_fwd, frag_bwd, stream);
    AMTX_REQUIRE(hidden == 256 || hidden == 384 || hidden == 512, "bilstm pack: hidden size %d is not built (128, 256, 384, 512)", hidden);
    hipLaunchKernelGGL(bilstm_pack_dev_h_kernel, dim3(256), dim3(256), 0, stream, whh_fwd, whh_bwd, hidden, planes, frag_fwd, frag_bwd);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_launch_bilstm_bwd_h(const float* dout, const float* save, const bf16_t* whh_t, int hidden, int planes, float* dxproj, int B, int T, int groups,
                             hipStream_t stream) {
    AMTX_REQUIRE(groups >= 1, "bilstm backward: bad group count");
    if (hidden == H) {
        for (int g = 0; g < groups; ++g) {        // the register-stationary kernel takes one LSTM per launch
            const int64_t bt = (int64_t)B * T;
            int rc = amtx_launch_bilstm_bwd(dout + g * bt * 2 * H, save + g * bt * 2 * 5 * H, whh_t + (size_t)g * amtx_bilstm_wfrag_elems(planes), planes,
                                            dxproj + g * bt * 8 * H, B, T, stream);
            if (rc != AMTX_OK) return rc;
        }
        return AMTX_OK;
    }
    AMTX_REQUIRE(dout && save && whh_t && dxproj, "bilstm backward: null pointer");
    AMTX_REQUIRE(B > 0 && T > 0 && planes == 2 && (hidden == 256 || hidden == 384 || hidden == 512),
                 "bilstm backward: hidden 256 / 384 / 512 are built for the two-plane precision only (got hidden %d, planes %d)", hidden, planes);
    LstmBwdHArgs a{dout, save, whh_t, dxproj, B, T, (int64_t)amtx_bilstm_wfrag_elems_h(hidden, planes)};
    dim3 grid((unsigned)((B + 3) / 4), 2, (unsigned)groups);
    const size_t lds = 2 * (size_t)planes * 5 * (4 * hidden + 8) * 2;
    if (hidden == 256) {
        AMTX_GRANT_LDS((bilstm_stream_bwd_kernel<256, 2>), lds);
        hipLaunchKernelGGL((bilstm_stream_bwd_kernel<256, 2>), grid, dim3(512), lds, stream, a);
    } else if (hidden == 384) {
        AMTX_GRANT_LDS((bilstm_stream_bwd_kernel<384, 2>), lds);
        hipLaunchKernelGGL((bilstm_stream_bwd_kernel<384, 2>), grid, dim3(512), lds, stream, a);
    } else {
        AMTX_GRANT_LDS((bilstm_stream_bwd_kernel<512, 2>), lds);
        hipLaunchKernelGGL((bilstm_stream_bwd_kernel<512, 2>), grid, dim3(512), lds, stream, a);
    }
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}
