// Spectral front-end for gfx950: framed real FFT (n_fft = 2048 tuned; any other power of two 128 .. 4096 through a generic kernel) with LDS butterfly staging ->
// power spectrum -> sparse triangular mel filterbank (wave-parallel rows) -> per-clip max [K1],
// then dB / scale / layout [K2].
//
// Replaces librosa.stft + librosa.feature.melspectrogram + power_to_db / amplitude_to_db as the
// reference calls them from amt_tools/features/stft.py:66-72, mel.py:64-71,94, common.py:199,218-228.
//
// K1 mapping (one wave64 = one frame at a time; one 256-thread block = FPB consecutive frames of a clip):
//   * the 2048 real samples of a frame are read straight from HBM/L2 as 1024 complex z[m] = x[2m] + i x[2m+1]
//     (lane j reads m = 64*n1 + j: consecutive lanes -> consecutive addresses; the four waves of a block
//     work on adjacent frames, so the 4x frame overlap is served by L1/L2, HBM sees each sample ~once),
//   * 1024-point complex FFT = radix-16 (in registers) x radix-16 (in registers) x radix-4, with two
//     wave-private LDS exchanges (no workgroup barrier: a wave's LDS operations retire in order),
//   * the radix-4 tail is merged into the real-FFT untangling pass that produces |X[k]|^2 for k and
//     n_fft/2-k together, written to a wave-private LDS power row,
//   * mel rows are contiguous bin ranges (<= ~30 taps): lane-per-row gather from that LDS row,
//     4 rounds of 64 rows, coalesced store of the [T][n_mels] row, running max for the dB reference.
// Algorithmic HBM bytes per frame: hop*4 read + n_out*4 written (2048 + 916 B for mel-229).

#include "amtx_common.h"

#include <math.h>
#include <algorithm>
#include <functional>
#include <vector>

namespace {

constexpr int NFFT = 2048;
constexpr int M = NFFT / 2;          // complex FFT length
constexpr int XB_PITCH = 68;         // LDS row pitch (complex) of the exchange buffer: conflict-free for both passes
constexpr int XB_ELEMS = 16 * XB_PITCH;
constexpr int PB_BINS = 1025 + 128;  // bins a padded mel row may touch: the last real bin is 1024, the rest stays zero
constexpr int PB_ELEMS = 1232;       // power row: bin k lives at k + (k >> 4) (PB_BINS bins -> 1225 slots), see pidx()
constexpr int MEL_UNROLL = 4;        // taps per batch of the mel gather (loads in flight per lane); round tap counts are padded to it
constexpr int WAVES = 4;
constexpr int MAX_MEL_ROUNDS = 8;    // up to 512 output rows

struct SpecDev {
    const float* window;     // [NFFT]
    const float2* tw_fft;    // [M]   exp(-2 pi i k / M)
    const float2* tw_post;   // [M]   exp(-2 pi i k / NFFT), k < M
    const int* mel_start;    // [64 * rounds] slot (round r, lane l): first bin its taps read (the row's first non-zero bin minus an even number of zero taps)
    const int* mel_row;      // [64 * rounds] slot -> mel row it computes (-1: none).  Rows are dealt to slots so that the 32 lanes of a half-wave start
                             // on 32 different LDS banks (plan creation: mel_assign_slots), not in row order
    const float* mel_wt;     // tap-major weights of round r at round_off[r]*64: [tap j][lane] = weight j of row 64r + lane, zero padded
    int round_max[MAX_MEL_ROUNDS];   // taps of round r = max tap count of rows [64r, 64r+64), rounded up to a multiple of MEL_UNROLL
    int round_off[MAX_MEL_ROUNDS];   // first tap slot of round r in mel_wt
    int hop, n_out, n_mels, center, pad_mode;
    int n_fft;               // frame length; the tuned kernel below is n_fft = 2048 only, spec_power_pow2_kernel takes 128 .. 4096
};

struct __attribute__((packed, aligned(4))) f32pair_a4 { float x, y; };   // two floats at a 4-byte aligned address: one global_load_dwordx2

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// forward DFT-4 in place: X[k] = sum_n x[n] exp(-2 pi i n k / 4)
__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    float2 t0 = make_float2(x0.x + x2.x, x0.y + x2.y);
    float2 t1 = make_float2(x0.x - x2.x, x0.y - x2.y);
    float2 t2 = make_float2(x1.x + x3.x, x1.y + x3.y);
    float2 t3 = make_float2(x1.x - x3.x, x1.y - x3.y);
    x0 = make_float2(t0.x + t2.x, t0.y + t2.y);
    x2 = make_float2(t0.x - t2.x, t0.y - t2.y);
    x1 = make_float2(t1.x + t3.y, t1.y - t3.x);
    x3 = make_float2(t1.x - t3.y, t1.y + t3.x);
}

// forward DFT-16 in registers.  Input v[n] natural order; output X[k] lands in v[4*(k&3) + (k>>2)].
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508977f;   // sin(pi/8)
    constexpr float R2 = 0.70710678118654752f;   // sqrt(1/2)
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // v[4*k1 + n2] *= W16^(n2*k1),  W16^m = (cos(pi m/8), -sin(pi m/8))
    v[5] = cmul(v[5], make_float2(C1, -S1));     // m = 1
    v[6] = cmul(v[6], make_float2(R2, -R2));     // m = 2
    v[7] = cmul(v[7], make_float2(S1, -C1));     // m = 3
    v[9] = cmul(v[9], make_float2(R2, -R2));     // m = 2
    v[10] = make_float2(v[10].y, -v[10].x);      // m = 4: -i
    v[11] = cmul(v[11], make_float2(-R2, -R2));  // m = 6
    v[13] = cmul(v[13], make_float2(S1, -C1));   // m = 3
    v[14] = cmul(v[14], make_float2(-R2, -R2));  // m = 6
    v[15] = cmul(v[15], make_float2(-C1, S1));   // m = 9: (cos(9pi/8), -sin(9pi/8)) = (-C1, +S1)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

// power-row slot of FFT bin k: one pad word per 16 bins, so the untangling pass (lanes stride 16 bins) and the
// mel gather (lanes start at arbitrary bins) both spread over all LDS banks instead of two.
__device__ __forceinline__ int pidx(int k) { return k; }

__device__ __forceinline__ void wave_lds_sync() {
    // same-wave LDS traffic retires in order; this only stops the compiler from moving LDS accesses
    // of different lanes' data across the exchange point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float fetch_padded(const float* clip, int64_t idx, int64_t n, int pad_mode) {
    if (idx >= 0 && idx < n) return clip[idx];
    if (pad_mode == AMTX_PAD_REFLECT) {
        if (idx < 0) idx = -idx;
        if (idx >= n) idx = 2 * (n - 1) - idx;
        if (idx >= 0 && idx < n) return clip[idx];
    }
    return 0.0f;
}

// |E + W O|^2 and |E - W O|^2 for the real-FFT untangling of the pair (Z[k], Z[M-k]).
__device__ __forceinline__ void untangle_pair(float2 zk, float2 zp, float2 w, float& pk, float& pmk) {
    // E = (Zk + conj(Zp))/2 ; O = (Zk - conj(Zp))/(2i)
    float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));
    float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));
    float2 o = make_float2(d.y, -d.x);                    // d / i
    float2 wo = cmul(w, o);
    float ar = e.x + wo.x, ai = e.y + wo.y;
    float br = e.x - wo.x, bi = e.y - wo.y;
    pk = ar * ar + ai * ai;
    pmk = br * br + bi * bi;
}

// 1024-point complex FFT of the windowed frame (v[n1] = z[64 n1 + lane]) + real-FFT untangling -> |X[k]|^2, k = 0 .. 1024, in the
// wave-private LDS power row pb.  xb: the wave's exchange buffer; twp / tw2l: block-shared twiddle tables; tw1: per-lane pass-A twiddles.
__device__ __forceinline__ void fft_power_row(float2 (&v)[16], float2* __restrict__ xb, float* __restrict__ pb, const float2* __restrict__ twp,
                                              const float2* __restrict__ tw2l, const float2 (&tw1)[16], int lane) {
    // ---- pass A: DFT-16 over n1, twiddle W_1024^(lane*k1), exchange 1
    dft16(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const float2 y = cmul(v[4 * (k1 & 3) + (k1 >> 2)], tw1[k1]);
        xb[k1 * XB_PITCH + lane] = y;
    }
    wave_lds_sync();
    // lane = (k1 = lane>>2, b = lane&3) gathers B[k1][4a + b], a = 0..15
    {
        const float2* row = xb + (lane >> 2) * XB_PITCH + (lane & 3);
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = row[4 * a];
    }
    wave_lds_sync();
    // ---- pass B: DFT-16 over a, twiddle W_64^(b*c), exchange 2 into [k1][c][b]
    dft16(v);
    {
        float2* row = xb + (lane >> 2) * XB_PITCH + (lane & 3);
        const float2* t2 = tw2l + (lane & 3) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) row[4 * c] = cmul(v[4 * (c & 3) + (c >> 2)], t2[c]);
    }
    wave_lds_sync();

    // ---- radix-4 tail merged with the real-FFT untangling: 128 tasks = exactly two full-wave iterations.
    // A task takes a 4-point group (k1, c) and its partner group ((16 - k1) & 15, 15 - c), runs the two DFT-4s and
    // untangles the four bin pairs (k, M - k), k = k1 + 16 c + 256 d.  The two self-paired groups (0,0) and (0,8) would be
    // a 129th task (and a third, one-lane iteration): they are merged into ONE task (q = 127) whose four untangle slots
    // are (Z0,Z0) -> bins 0 / 1024, (Z1,Z3) -> 256 / 768 of group (0,0) and (Z0,Z3) -> 128 / 896, (Z1,Z2) -> 384 / 640 of
    // group (0,8); the one bin left over, 512, is its own partner: X[512] = conj(Z[512]), power |Z2|^2.
#pragma unroll
    for (int q = lane; q < 128; q += 64) {
        int k1, c, pk1, pc;
        const bool sp = q == 127;
        if (q < 112) { c = (q * 9363) >> 16; k1 = 1 + q - 7 * c; pk1 = 16 - k1; pc = 15 - c; }   // c = q / 7, k1 = 1 + q % 7: k1-major lanes
        else if (q < 120) { k1 = 8; c = q - 112; pk1 = 8; pc = 15 - c; }
        else if (q < 127) { k1 = 0; c = q - 119; pk1 = 0; pc = 16 - c; }                          // c = 1..7 with 15..9
        else { k1 = 0; c = 0; pk1 = 0; pc = 8; }
        const float4* ga = reinterpret_cast<const float4*>(xb + k1 * XB_PITCH + 4 * c);
        const float4* gb = reinterpret_cast<const float4*>(xb + pk1 * XB_PITCH + 4 * pc);
        float4 a01 = ga[0], a23 = ga[1], b01 = gb[0], b23 = gb[1];
        float2 za[4] = {make_float2(a01.x, a01.y), make_float2(a01.z, a01.w), make_float2(a23.x, a23.y), make_float2(a23.z, a23.w)};
        float2 zb[4] = {make_float2(b01.x, b01.y), make_float2(b01.z, b01.w), make_float2(b23.x, b23.y), make_float2(b23.z, b23.w)};
        dft4(za[0], za[1], za[2], za[3]);
        dft4(zb[0], zb[1], zb[2], zb[3]);
        const int kbase = k1 + 16 * c;
        // slot d untangles (A[d], P[d]) into bins (ks[d], M - ks[d]); the merged task re-routes slots by select
        const float2 A[4] = {za[0], za[1], sp ? zb[0] : za[2], sp ? zb[1] : za[3]};
        const float2 P[4] = {sp ? za[0] : zb[3], sp ? za[3] : zb[2], sp ? zb[3] : zb[1], sp ? zb[2] : zb[0]};
        const int ks[4] = {kbase, kbase + 256, sp ? 128 : kbase + 512, sp ? 384 : kbase + 768};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float pk, pmk;
            float2 w = twp[ks[d] & (M / 2 - 1)];
            if (d >= 2) {                                  // ks >= 512 unless this is the merged task (128, 384)
                const float2 wr = make_float2(w.y, -w.x);  // times -i
                w = sp ? w : wr;
            }
            untangle_pair(A[d], P[d], w, pk, pmk);
            pb[pidx(ks[d])] = pk;
            pb[pidx(M - ks[d])] = pmk;
        }
        if (sp) pb[pidx(M / 2)] = za[2].x * za[2].x + za[2].y * za[2].y;
    }
    wave_lds_sync();

}

// MELLDS: the tap-major mel table (slots x 64 lanes) is copied into LDS once per block and the gather reads its weights from
// there: per frame 72 global loads per lane (L1/L2 hits, but nine dependent batches of load latency with only two waves per SIMD
// to cover them) become conflict-free LDS reads.  Plans whose table does not leave room for two blocks per CU keep it in memory.
constexpr int MEL_LDS_MAX_SLOTS = 80;

#ifdef AMTX_SPEC_TIMING
// debug build only (general kernel): cycles wave 0 of every block spends per section of a frame, summed: [0] window + next-frame load
// issue, [3] FFT + untangling (fft_power_row), [4] mel gather + stores, [5] frames
__device__ unsigned long long g_spec_prof[8];
#define SPEC_TICK(SLOT)                                                    \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        prof_acc[SLOT] += now_ - prof_t;                                   \
        prof_t = now_;                                                     \
    } while (0)
#else
#define SPEC_TICK(SLOT) do {} while (0)
#endif

template <int FPW, bool MEL, bool MELLDS>
__global__ __launch_bounds__(256, 2) void spec_power_kernel(SpecDev p, const float* __restrict__ audio, int64_t num_samples,
                                                         int64_t audio_stride, int64_t num_frames,
                                                         float* __restrict__ power, unsigned* __restrict__ clip_max) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float2* xb = reinterpret_cast<float2*>(smem) + wave * XB_ELEMS;
    float* pb = reinterpret_cast<float*>(smem + WAVES * XB_ELEMS * sizeof(float2)) + wave * PB_ELEMS;
    float2* twp = reinterpret_cast<float2*>(smem + WAVES * XB_ELEMS * sizeof(float2) + WAVES * PB_ELEMS * sizeof(float));
    float2* tw2l = twp + M / 2;                       // W_64^(b*c), [b][c]: 64 entries
    int* mstart = reinterpret_cast<int*>(tw2l + 64);  // first bin of every mel slot, [64 * MAX_MEL_ROUNDS]
    int* mrow = mstart + 64 * MAX_MEL_ROUNDS;         // mel row of every slot (-1: none), [64 * MAX_MEL_ROUNDS]
    float* melw = reinterpret_cast<float*>(mrow + 64 * MAX_MEL_ROUNDS);     // MELLDS: [slot][lane] weights

    constexpr int FPB = FPW * WAVES;
    const unsigned chunks = (unsigned)((num_frames + FPB - 1) / FPB);
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned clip_idx = logical / chunks;
    const unsigned chunk = logical % chunks;
    const float* clip = audio + (int64_t)clip_idx * audio_stride;

    // block-shared tables: untangling twiddles exp(-2 pi i k / 2048) for k < 512 (k + 512 is the same value times -i: a swap and
    // a sign, applied where it is used) and the pass-B twiddles
    for (int i = threadIdx.x; i < M / 2; i += 256) twp[i] = p.tw_post[i];
    if (threadIdx.x < 64) tw2l[threadIdx.x] = p.tw_fft[(16 * (threadIdx.x >> 4) * (threadIdx.x & 15)) & (M - 1)];

    // per-lane constants, reused by every frame this wave transforms
    float wre[16], wim[16];
    float2 tw1[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int m = n1 * 64 + lane;
        wre[n1] = p.window[2 * m];
        wim[n1] = p.window[2 * m + 1];
        tw1[n1] = p.tw_fft[(lane * n1) & (M - 1)];
    }
    const int rounds = MEL ? (p.n_mels + 63) / 64 : 0;
    for (int i = threadIdx.x; i < 64 * rounds; i += 256) { mstart[i] = p.mel_start[i]; mrow[i] = p.mel_row[i]; }
    if constexpr (MELLDS) {
        const int nw = 64 * (p.round_off[rounds - 1] + p.round_max[rounds - 1]);
        for (int i = threadIdx.x; i < nw; i += 256) melw[i] = p.mel_wt[i];
    }
    for (int i = pidx(M + 1) + lane; i < PB_ELEMS; i += 64) pb[i] = 0.0f;   // bins past 1024: read (times 0) by padded mel taps, never written
    __syncthreads();

    float run_max = 0.0f;
    const int64_t half = p.center ? NFFT / 2 : 0;

    // raw samples of the frame about to be transformed; the NEXT frame's samples are requested right after these are
    // consumed, so their HBM/L2 latency is covered by a whole FFT + mel pass instead of stalling every frame
    float xr[16], xi[16];
#define SPEC_LOAD_FRAME(T_)                                                                              \
    do {                                                                                                 \
        const int64_t s0_ = (T_) * p.hop - half;                                                         \
        if (s0_ >= 0 && s0_ + NFFT <= num_samples) {                                                     \
            const float* src_ = clip + s0_ + 2 * lane;                                                   \
            _Pragma("unroll") for (int n1 = 0; n1 < 16; ++n1) { xr[n1] = src_[128 * n1]; xi[n1] = src_[128 * n1 + 1]; } \
        } else {                                                                                         \
            _Pragma("unroll") for (int n1 = 0; n1 < 16; ++n1) {                                          \
                const int64_t idx_ = s0_ + 2 * (n1 * 64 + lane);                                         \
                xr[n1] = fetch_padded(clip, idx_, num_samples, p.pad_mode);                              \
                xi[n1] = fetch_padded(clip, idx_ + 1, num_samples, p.pad_mode);                          \
            }                                                                                            \
        }                                                                                                \
    } while (0)

    {
        const int64_t tfirst = (int64_t)chunk * FPB + wave;
        if (tfirst < num_frames) SPEC_LOAD_FRAME(tfirst);
    }

#ifdef AMTX_SPEC_TIMING
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif
#pragma unroll 1
    for (int i = 0; i < FPW; ++i) {
        const int64_t t = (int64_t)chunk * FPB + i * WAVES + wave;
        if (t >= num_frames) break;
#ifdef AMTX_SPEC_TIMING
        prof_t = __builtin_readcyclecounter();
        prof_acc[5] += 1;
#endif

        // ---- window: v[n1] = z[64 n1 + lane]
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = make_float2(xr[n1] * wre[n1], xi[n1] * wim[n1]);
        {
            const int64_t tnext = t + WAVES;
            if (i + 1 < FPW && tnext < num_frames) SPEC_LOAD_FRAME(tnext);
        }

        SPEC_TICK(0);
        fft_power_row(v, xb, pb, twp, tw2l, tw1, lane);
        SPEC_TICK(3);
        float* out_row = power + ((int64_t)clip_idx * num_frames + t) * p.n_out;
        if (MEL) {
            // ---- sparse mel: lane-per-row gather over contiguous bin ranges.  Weights are tap-major and zero padded to
            // the round's tap count, so every access is unconditional: MEL_UNROLL coalesced weight loads (L1/L2 hits, the
            // table is shared by all waves) and LDS power reads are in flight per batch instead of one waited-for load per
            // tap.  Rows past their support multiply finite power values (the row is zero beyond bin 1024) by 0.
            // The stores wait until every round is done: vmcnt retires in order, a store between two rounds would make
            // the next round's first weight wait for the store's acknowledgement.
            float res[MAX_MEL_ROUNDS];
#pragma unroll
            for (int r = 0; r < MAX_MEL_ROUNDS; ++r) {
                res[r] = 0.0f;
                if (r < rounds) {
                    const int start = mstart[r * 64 + lane];
                    const float* wt = (MELLDS ? melw : p.mel_wt) + p.round_off[r] * 64 + lane;
                    const int nmax = p.round_max[r];
                    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll 1
                    for (int j = 0; j < nmax; j += MEL_UNROLL) {
                        float wv[MEL_UNROLL], xv[MEL_UNROLL];
#pragma unroll
                        for (int u = 0; u < MEL_UNROLL; ++u) wv[u] = wt[(j + u) * 64];
#pragma unroll
                        for (int u = 0; u < MEL_UNROLL; ++u) xv[u] = pb[pidx(start + j + u)];
#pragma unroll
                        for (int u = 0; u < MEL_UNROLL; u += 2) {
                            acc0 = fmaf(wv[u], xv[u], acc0);
                            acc1 = fmaf(wv[u + 1], xv[u + 1], acc1);
                        }
                    }
                    res[r] = acc0 + acc1;
                }
            }
#pragma unroll
            for (int r = 0; r < MAX_MEL_ROUNDS; ++r) {
                const int row = r < rounds ? mrow[r * 64 + lane] : -1;
                if (row >= 0) {
                    out_row[row] = res[r];
                    run_max = fmaxf(run_max, res[r]);
                }
            }
        } else {
            for (int k = lane; k < p.n_out; k += 64) {
                const float val = pb[pidx(k)];
                out_row[k] = val;
                run_max = fmaxf(run_max, val);
            }
        }
        wave_lds_sync();   // pb / xb are rewritten by the next frame
        SPEC_TICK(4);
    }
#undef SPEC_LOAD_FRAME
#ifdef AMTX_SPEC_TIMING
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&g_spec_prof[i], prof_acc[i]);
#endif

    run_max = wave_max_f32(run_max);
    if (lane == 0 && run_max > 0.0f) atomicMax(clip_max + clip_idx, __float_as_uint(run_max));   // values >= 0: uint order == float order
}

// ------------------------------------------------------------------------------------------------------------------
// The same frame pipeline specialised for the BASELINE log-mel front-end (n_fft 2048, hop 512, a mel table whose four rounds of 64
// rows need S0 / S1 / S2 / S3 tap slots): measured on the general kernel, a quarter of its time went into fetching every sample
// four times (each wave loads the whole 2048-sample frame, the four frames of a block overlap by 75 %: 0.40 of 2.00 ms with the
// loads compiled out) and another quarter into the mel gather's two LDS reads per tap.  Here
//   * the block keeps a 4096-sample RING of the clip in LDS: per iteration (four adjacent frames, one per wave) the block fetches
//     only the 2048 NEW samples (8 per thread, prefetched a whole iteration ahead in registers) and every wave reads its frame from
//     the ring -- 3.7 x fewer global load instructions, two block barriers per four frames;
//   * the mel weights are STATIONARY IN REGISTERS (slot q of lane l = weight q of row 64 r + l, exactly the tap-major table): the
//     gather is one LDS read + one FMA per tap, and the 18 KB the table took in LDS pay for the ring.
// Arithmetic and summation order are those of spec_power_kernel: the two kernels return the same bits (tests/test_gpu_frontend.py).
constexpr int RING_HOP = 512;
constexpr int RING_ELEMS = 4096;                 // floats; window of an iteration: 3 hops + 2048 = 3584, new per iteration: 2048

template <int FPW, int S0, int S1, int S2, int S3>
__global__ __launch_bounds__(256, 2) void spec_power_ring_kernel(SpecDev p, const float* __restrict__ audio, int64_t num_samples,
                                                              int64_t audio_stride, int64_t num_frames, float* __restrict__ power,
                                                              unsigned* __restrict__ clip_max) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NSLOT = S0 + S1 + S2 + S3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* xb = reinterpret_cast<float2*>(smem) + wave * XB_ELEMS;
    float* pb = reinterpret_cast<float*>(smem + WAVES * XB_ELEMS * sizeof(float2)) + wave * PB_ELEMS;
    float2* twp = reinterpret_cast<float2*>(smem + WAVES * XB_ELEMS * sizeof(float2) + WAVES * PB_ELEMS * sizeof(float));
    float2* tw2l = twp + M / 2;
    int* mstart = reinterpret_cast<int*>(tw2l + 64);                       // [4][64]
    float* ring = reinterpret_cast<float*>(mstart + 64 * 4);               // [RING_ELEMS]

    constexpr int FPB = FPW * WAVES;
    const unsigned chunks = (unsigned)((num_frames + FPB - 1) / FPB);
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned clip_idx = logical / chunks, chunk = logical % chunks;
    const float* clip = audio + (int64_t)clip_idx * audio_stride;

    for (int i = tid; i < M / 2; i += 256) twp[i] = p.tw_post[i];
    if (tid < 64) tw2l[tid] = p.tw_fft[(16 * (tid >> 4) * (tid & 15)) & (M - 1)];
    for (int i = tid; i < 64 * 4; i += 256) mstart[i] = p.mel_start[i];
    float wre[16], wim[16];
    float2 tw1[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int m = n1 * 64 + lane;
        wre[n1] = p.window[2 * m];
        wim[n1] = p.window[2 * m + 1];
        tw1[n1] = p.tw_fft[(lane * n1) & (M - 1)];
    }
    float mw[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) mw[q] = p.mel_wt[q * 64 + lane];
    // mel row of this lane's slot in each round (-1: none).  FOUR rounds exactly: the dispatcher (amtx_spec_power) sends only plans with
    // mel_rounds == 4 here -- with fewer, entries r >= rounds of mel_row would be weight bits, not row indices
    static_assert(NSLOT == 4 + 8 + 20 + 32, "spec_power_ring_kernel is built for the four mel rounds of the BASELINE table");
    int orow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) orow[r] = p.mel_row[r * 64 + lane];
    for (int i = pidx(M + 1) + lane; i < PB_ELEMS; i += 64) pb[i] = 0.0f;

    // ---- sample staging.  sb = first sample of the block's first frame; sample s lives at ring[(s - sb) & (RING_ELEMS - 1)].
    const int64_t half = p.center ? NFFT / 2 : 0;
    const int64_t sb = (int64_t)chunk * FPB * RING_HOP - half;
    // 2048 samples starting at s0: thread tid takes the pairs (2 tid + 512 m, + 1), m = 0..3 (consecutive lanes, consecutive 8 bytes)
    float2 pre[4];
#define RING_FETCH(S0_)                                                                                  \
    do {                                                                                                 \
        const int64_t s0_ = (S0_);                                                                       \
        if (s0_ >= 0 && s0_ + 2048 <= num_samples) {                                                     \
            const f32pair_a4* src_ = reinterpret_cast<const f32pair_a4*>(clip + s0_ + 2 * tid);          \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) { const f32pair_a4 v_ = src_[256 * m]; pre[m] = make_float2(v_.x, v_.y); } \
        } else {                                                                                         \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                              \
                const int64_t idx_ = s0_ + 2 * tid + 512 * m;                                            \
                pre[m] = make_float2(fetch_padded(clip, idx_, num_samples, p.pad_mode), fetch_padded(clip, idx_ + 1, num_samples, p.pad_mode)); \
            }                                                                                            \
        }                                                                                                \
    } while (0)
#define RING_STORE(S0_)                                                                                  \
    do {                                                                                                 \
        const int r0_ = (int)(((S0_) - sb) & (RING_ELEMS - 1));                                          \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                    \
            *reinterpret_cast<float2*>(ring + ((r0_ + 2 * tid + 512 * m) & (RING_ELEMS - 1))) = pre[m];  \
    } while (0)
    // window 0 = [sb, sb + 3584): 2048 + 1536 samples (the second fetch brings 2048, of which the last 512 belong to window 1)
    RING_FETCH(sb);
    RING_STORE(sb);
    RING_FETCH(sb + 2048);
    RING_STORE(sb + 2048);
    if (FPW > 1) RING_FETCH(sb + 4096);                       // the rest of window 1's new samples: [sb + 4096, sb + 6144)

    float run_max = 0.0f;
#pragma unroll 1
    for (int i = 0; i < FPW; ++i) {
        const int64_t t = (int64_t)chunk * FPB + i * WAVES + wave;
        const bool active = t < num_frames;
        __syncthreads();                                      // window i is in the ring (and the tables, first time round)
        float2 v[16];
        {
            const int r0 = (2048 * i + RING_HOP * wave + 2 * lane) & (RING_ELEMS - 1);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 x = *reinterpret_cast<const float2*>(ring + ((r0 + 128 * n1) & (RING_ELEMS - 1)));
                v[n1] = make_float2(x.x * wre[n1], x.y * wim[n1]);
            }
        }
        __syncthreads();                                      // every wave has its frame: the oldest 2048 ring slots are free
        if (i + 1 < FPW) {
            // window i + 1 = [sb + 2048 (i + 1), + 3584): its last 1536 ... the ring holds up to sb + 2048 i + 4096, `pre` brings
            // [sb + 2048 i + 4096, + 2048)
            RING_STORE(sb + 2048 * (int64_t)i + 4096);
            if (i + 2 < FPW) RING_FETCH(sb + 2048 * (int64_t)(i + 1) + 4096);
        }
        if (!active) continue;

        fft_power_row(v, xb, pb, twp, tw2l, tw1, lane);

        // ---- sparse mel, weights in registers; same accumulation order as the general kernel (even taps -> acc0, odd -> acc1)
        float res[4];
#define RING_MEL_ROUND(R, OFF, N)                                                                        \
        {                                                                                                \
            const int start_ = mstart[(R) * 64 + lane];                                                  \
            float acc0 = 0.0f, acc1 = 0.0f;                                                              \
            _Pragma("unroll") for (int j = 0; j < (N); j += 2) {                                         \
                acc0 = fmaf(mw[(OFF) + j], pb[pidx(start_ + j)], acc0);                                  \
                acc1 = fmaf(mw[(OFF) + j + 1], pb[pidx(start_ + j + 1)], acc1);                          \
            }                                                                                            \
            res[R] = acc0 + acc1;                                                                        \
        }
        RING_MEL_ROUND(0, 0, S0)
        RING_MEL_ROUND(1, S0, S1)
        RING_MEL_ROUND(2, S0 + S1, S2)
        RING_MEL_ROUND(3, S0 + S1 + S2, S3)
#undef RING_MEL_ROUND
        float* out_row = power + ((int64_t)clip_idx * num_frames + t) * p.n_out;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (orow[r] >= 0) {
                out_row[orow[r]] = res[r];
                run_max = fmaxf(run_max, res[r]);
            }
        }
        wave_lds_sync();   // pb / xb are rewritten by the next frame
    }
#undef RING_FETCH
#undef RING_STORE
    run_max = wave_max_f32(run_max);
    if (lane == 0 && run_max > 0.0f) atomicMax(clip_max + clip_idx, __float_as_uint(run_max));
}

// Any other power-of-two frame length (128 .. 4096; amt_tools/features/stft.py:15-40 and mel.py:15-38 take any n_fft): one wave per
// frame, the n_fft real samples as m = n_fft / 2 complex points in a wave-private LDS buffer, in-place radix-2 decimation-in-time
// (bit-reversed on the way in, log2(m) butterfly passes separated by wave-level LDS ordering only), the same real-FFT untangling and
// the same lane-per-row mel gather as the tuned 2048 kernel.  Not tuned: ~log2(m) LDS round trips per frame instead of two.
__global__ __launch_bounds__(256) void spec_power_pow2_kernel(SpecDev p, int log2m, int fpw, const float* __restrict__ audio, int64_t num_samples,
                                                           int64_t audio_stride, int64_t num_frames, float* __restrict__ power,
                                                           unsigned* __restrict__ clip_max, int mel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = 1 << log2m, nfft = 2 * m;
    const int pb_elems = m + 1 + 128 + 3;                       // bins a zero-padded mel tap batch may touch
    float2* z = reinterpret_cast<float2*>(smem) + (size_t)wave * m;
    float* pb = reinterpret_cast<float*>(smem + (size_t)WAVES * m * sizeof(float2)) + (size_t)wave * pb_elems;
    const int fpb = fpw * WAVES;
    const unsigned chunks = (unsigned)((num_frames + fpb - 1) / fpb);
    const unsigned clip_idx = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const float* clip = audio + (int64_t)clip_idx * audio_stride;
    const int rounds = mel ? (p.n_mels + 63) / 64 : 0;
    for (int i = m + 1 + lane; i < pb_elems; i += 64) pb[i] = 0.0f;
    float run_max = 0.0f;
    const int64_t half = p.center ? nfft / 2 : 0;
    for (int i = 0; i < fpw; ++i) {
        const int64_t t = (int64_t)chunk * fpb + i * WAVES + wave;
        if (t >= num_frames) break;
        const int64_t s0 = t * p.hop - half;
        // windowed samples -> bit-reversed complex points
        for (int j = lane; j < m; j += 64) {
            const float xr = fetch_padded(clip, s0 + 2 * j, num_samples, p.pad_mode) * p.window[2 * j];
            const float xi = fetch_padded(clip, s0 + 2 * j + 1, num_samples, p.pad_mode) * p.window[2 * j + 1];
            z[__brev((unsigned)j) >> (32 - log2m)] = make_float2(xr, xi);
        }
        wave_lds_sync();
        for (int s = 0; s < log2m; ++s) {
            const int h = 1 << s;
            for (int b = lane; b < m / 2; b += 64) {
                const int j = b & (h - 1), i0 = ((b >> s) << (s + 1)) + j;
                const float2 w = p.tw_fft[j << (log2m - 1 - s)];          // exp(-2 pi i j / (2 h))
                const float2 a = z[i0], c = cmul(z[i0 + h], w);
                z[i0] = make_float2(a.x + c.x, a.y + c.y);
                z[i0 + h] = make_float2(a.x - c.x, a.y - c.y);
            }
            wave_lds_sync();
        }
        // real-FFT untangling: bins k and m - k from Z[k], Z[m - k]; k = 0 gives DC and Nyquist, k = m / 2 is its own partner
        for (int k = lane; k <= m / 2; k += 64) {
            const float2 zk = z[k], zp = z[(m - k) & (m - 1)];
            float pk, pmk;
            untangle_pair(zk, zp, p.tw_post[k], pk, pmk);
            pb[k] = pk;
            pb[m - k] = pmk;
        }
        wave_lds_sync();
        float* out_row = power + ((int64_t)clip_idx * num_frames + t) * p.n_out;
        if (mel) {
            for (int r = 0; r < rounds; ++r) {
                const int row = p.mel_row[r * 64 + lane];
                const int start = p.mel_start[r * 64 + lane];
                const float* wt = p.mel_wt + p.round_off[r] * 64 + lane;
                float acc = 0.0f;
                for (int j = 0; j < p.round_max[r]; ++j) acc = fmaf(wt[j * 64], pb[min(start + j, pb_elems - 1)], acc);
                if (row >= 0) {
                    out_row[row] = acc;
                    run_max = fmaxf(run_max, acc);
                }
            }
        } else {
            for (int k = lane; k < p.n_out; k += 64) {
                const float val = pb[k];
                out_row[k] = val;
                run_max = fmaxf(run_max, val);
            }
        }
        wave_lds_sync();
    }
    run_max = wave_max_f32(run_max);
    if (lane == 0 && run_max > 0.0f) atomicMax(clip_max + clip_idx, __float_as_uint(run_max));
}

// K2: dB conversion / scaling / layout.  One block = 32 frames x 32 bins tile.
__global__ __launch_bounds__(256) void spec_scale_kernel(const float* __restrict__ power, const float* __restrict__ clip_max,
                                                         const float* __restrict__ ref, int64_t num_frames, int n_bins,
                                                         int transform, int layout, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int f0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8

    DbScale dbs = {0.0f, 0.0f};
    if (transform == AMTX_SCALE_DB) {
        const float own = clip_max[b];
        dbs = db_scale_make(own, ref ? ref[b] : own);
    }
    const float* src = power + (int64_t)b * num_frames * n_bins;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = t0 + ty + 8 * i;
        const int f = f0 + tx;
        float v = 0.0f;
        if (t < num_frames && f < n_bins) {
            const float s = src[t * n_bins + f];
            if (transform == AMTX_SCALE_DB) {
                v = db_scale_apply(s, dbs);
            } else if (transform == AMTX_SCALE_MAGNITUDE) {
                v = sqrtf(s);
            } else {
                v = s;
            }
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
    if (layout == AMTX_LAYOUT_BTF_F32) {
        float* dst = out + (int64_t)b * num_frames * n_bins;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t t = t0 + ty + 8 * i;
            const int f = f0 + tx;
            if (t < num_frames && f < n_bins) dst[t * n_bins + f] = tile[ty + 8 * i][tx];
        }
    } else {
        float* dst = out + (int64_t)b * n_bins * num_frames;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = f0 + ty + 8 * i;
            const int64_t t = t0 + tx;
            if (t < num_frames && f < n_bins) dst[(int64_t)f * num_frames + t] = tile[tx][ty + 8 * i];
        }
    }
}

// ------------------------------------------------------------------ host-side table construction
double hz_to_mel(double f, bool htk) {
    if (htk) return 2595.0 * log10(1.0 + f / 700.0);
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m, bool htk) {
    if (htk) return 700.0 * (pow(10.0, m / 2595.0) - 1.0);
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// Which lane computes which mel row.  The gather reads pb[start + j] for tap j on every lane at once: two lanes of a half-wave whose
// first bins are congruent mod 32 hit one LDS bank at EVERY tap (dealt in row order, the 229-row Slaney table cost 304 LDS cycles per
// frame where a conflict-free gather costs 128: 44 % of the kernel's LDS cycles were conflicts, PMC round 3).  Two degrees of freedom
// remove them without touching any row's arithmetic: a row may sit in ANY round whose tap count covers it (the padding taps are
// zero), and its first tap may start an EVEN number of bins early (zero weights in front: the even / odd accumulator split of the
// gather, and with it every sum's rounding, is unchanged).  A slot = (round r, half-wave h, bank b); a maximum bipartite matching
// rows -> slots (Kuhn's augmenting paths, options tried nearest-to-natural first) gives every row of the BASELINE tables its own
// bank: 128 cycles.  Rows the matching leaves over take any free lane (they then share a bank with one other row).
void mel_assign_slots(const std::vector<int>& start, const std::vector<int>& count, int n_mels, int rounds, const int* round_max, bool natural,
                      std::vector<int>& slot_row, std::vector<int>& slot_start) {
    const int G = 2 * rounds;
    std::fill(slot_row.begin(), slot_row.end(), -1);
    std::fill(slot_start.begin(), slot_start.end(), 0);
    auto natural_fill = [&]() {
        std::fill(slot_row.begin(), slot_row.end(), -1);
        for (int i = 0; i < n_mels; ++i) { slot_row[i] = i; slot_start[i] = start[i]; }
    };
    if (natural || rounds == 0) { natural_fill(); return; }
    struct Opt { int cell, delta; };
    std::vector<std::vector<Opt>> opts(n_mels);
    for (int i = 0; i < n_mels; ++i) {
        std::vector<int> order;                                   // groups by distance from the row's natural round, halves of a round together
        const int nat = i / 64;
        for (int dist = 0; dist < rounds; ++dist)
            for (int sgn = 0; sgn < 2; ++sgn) {
                const int r = sgn ? nat - dist : nat + dist;
                if ((dist == 0 && sgn) || r < 0 || r >= rounds) continue;
                order.push_back(2 * r + ((i / 32) & 1));
                order.push_back(2 * r + 1 - ((i / 32) & 1));
            }
        for (int g : order) {
            const int S = round_max[g / 2];
            for (int d = 0; d + count[i] <= S && start[i] - d >= 0; d += 2) opts[i].push_back({g * 32 + ((start[i] - d) & 31), d});
        }
    }
    std::vector<int> owner((size_t)G * 32, -1), pick(n_mels, -1);
    std::vector<char> seen;
    std::function<bool(int)> place = [&](int i) -> bool {
        for (int k = 0; k < (int)opts[i].size(); ++k) {
            const int cell = opts[i][k].cell;
            if (seen[cell]) continue;
            seen[cell] = 1;
            if (owner[cell] < 0 || place(owner[cell])) { owner[cell] = i; pick[i] = k; return true; }
        }
        return false;
    };
    std::vector<int> rows(n_mels);
    for (int i = 0; i < n_mels; ++i) rows[i] = i;
    std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return opts[a].size() < opts[b].size(); });
    for (int i : rows) {
        seen.assign((size_t)G * 32, 0);
        (void)place(i);
    }
    // cells -> lanes (the lanes of a half-wave are interchangeable: dealt in bank order), then the leftovers
    std::vector<int> used(G, 0);
    for (int g = 0; g < G; ++g)
        for (int b = 0; b < 32; ++b) {
            const int i = owner[(size_t)g * 32 + b];
            if (i < 0) continue;
            const int sl = (g / 2) * 64 + (g & 1) * 32 + used[g]++;
            slot_row[sl] = i;
            slot_start[sl] = start[i] - opts[i][pick[i]].delta;
        }
    for (int i = 0; i < n_mels; ++i) {
        if (pick[i] >= 0 && owner[opts[i][pick[i]].cell] == i) continue;
        bool done = false;
        for (int g = 0; g < G && !done; ++g)
            if (round_max[g / 2] >= count[i] && used[g] < 32) {
                const int sl = (g / 2) * 64 + (g & 1) * 32 + used[g]++;
                slot_row[sl] = i; slot_start[sl] = start[i];
                done = true;
            }
        if (!done) { natural_fill(); return; }                    // cannot happen while every row fits its natural round; keep the plan valid anyway
    }
}

// Host tables of the mel stage: librosa.filters.mel (norm='slaney', fmin=0, fmax=sr/2; float32 triangle scaled by the float64 area norm,
// as librosa does) and the gather's slot layout (mel_assign_slots).  No device involved: amtx_spec_mel_layout exposes it to CPU tests.
struct MelHost {
    int rounds = 0;
    int round_max[MAX_MEL_ROUNDS], round_off[MAX_MEL_ROUNDS];
    std::vector<float> fb_dense;            // n_mels x (n_fft / 2 + 1)
    std::vector<int> start, count;          // per row: first non-zero bin, taps
    std::vector<int> slot_row, slot_start;  // per slot (64 per round)
    std::vector<float> w;                   // tap-major weights, [slot tap][lane]
};

int build_mel_tables(int sample_rate, int n_fft, int n_mels, int htk, MelHost& mh) {
    const int nb = n_fft / 2 + 1;
    const int mel_rounds = n_mels > 0 ? (n_mels + 63) / 64 : 0;
    mh.rounds = mel_rounds;
    memset(mh.round_max, 0, sizeof(mh.round_max));
    memset(mh.round_off, 0, sizeof(mh.round_off));
    mh.start.assign(n_mels > 0 ? n_mels : 1, 0);
    mh.count.assign(n_mels > 0 ? n_mels : 1, 0);
    mh.slot_row.assign(mel_rounds > 0 ? 64 * mel_rounds : 1, -1);
    mh.slot_start.assign(mel_rounds > 0 ? 64 * mel_rounds : 1, 0);
    mh.w.clear();
    mh.fb_dense.clear();
    if (n_mels > 0) {
        std::vector<int>& m_start = mh.start;
        std::vector<int>& m_count = mh.count;
        mh.fb_dense.assign((size_t)n_mels * nb, 0.0f);
        const double fmax = sample_rate / 2.0;
        std::vector<double> mel_f(n_mels + 2);
        const double mlo = hz_to_mel(0.0, htk), mhi = hz_to_mel(fmax, htk);
        for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(mlo + (mhi - mlo) * i / (n_mels + 1), htk);
        for (int i = 0; i < n_mels; ++i) {
            const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
            int first = -1, last = -1;
            const double fstep = fmax / (nb - 1);
            for (int k = 0; k < nb; ++k) {
                const double f = k * fstep;
                const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
                const double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
                const double w = lower < upper ? lower : upper;
                // librosa stores the triangle in float32, then scales in place by the float64 area norm
                const float w32 = (float)(w > 0 ? w : 0);
                const float wf = (float)((double)w32 * enorm);
                mh.fb_dense[(size_t)i * nb + k] = wf;
                if (wf != 0.0f) { if (first < 0) first = k; last = k; }
            }
            if (first >= 0) { m_start[i] = first; m_count[i] = last - first + 1; }
            const int r = i / 64;
            if (m_count[i] > mh.round_max[r]) mh.round_max[r] = m_count[i];
        }
        // tap-major, zero-padded weight table per round of 64 slots
        int slots = 0;
        for (int r = 0; r < mel_rounds; ++r) {
            mh.round_max[r] = (mh.round_max[r] + MEL_UNROLL - 1) / MEL_UNROLL * MEL_UNROLL;
            mh.round_off[r] = slots;
            slots += mh.round_max[r];
        }
        // The bank-conflict-free matching may move a row into a round with more tap slots or start it early behind zero weights; should that
        // ever push a slot past the padded power row, the natural row-order layout (which fits whenever the rows do) is used instead of
        // failing the plan.
        bool natural = getenv("AMTX_SPEC_NATURAL_ROWS") != nullptr;
        for (int attempt = 0; attempt < 2; ++attempt) {
            mel_assign_slots(m_start, m_count, n_mels, mel_rounds, mh.round_max, natural, mh.slot_row, mh.slot_start);
            mh.w.assign((size_t)slots * 64, 0.0f);
            int bad = -1;
            for (int sl = 0; sl < 64 * mel_rounds && bad < 0; ++sl) {
                const int i = mh.slot_row[sl];
                if (i < 0) continue;
                const int r = sl / 64, l = sl % 64;
                const int lead = m_start[i] - mh.slot_start[sl];          // zero taps in front of the row's first weight (even, >= 0)
                if (lead < 0 || lead + m_count[i] > mh.round_max[r] || mh.slot_start[sl] + mh.round_max[r] > nb + 128) { bad = i; break; }
                for (int j = 0; j < m_count[i]; ++j) mh.w[((size_t)mh.round_off[r] + lead + j) * 64 + l] = mh.fb_dense[(size_t)i * nb + m_start[i] + j];
            }
            if (bad < 0) break;
            if (natural) {
                amtx_set_error("amtx_spec_plan_create: mel row %d (%d taps from bin %d) does not fit its slot / the padded power row", bad, m_count[bad], m_start[bad]);
                return AMTX_ERR_UNSUPPORTED;
            }
            natural = true;                                              // second attempt: rows in order
        }
    }
    return AMTX_OK;
}

}  // namespace

struct amtx_spec_plan {
    int sample_rate, n_fft, hop, win, n_mels, htk, center, pad_mode;
    int n_bins_fft;   // n_fft/2 + 1
    int n_out;        // n_mels, or n_bins_fft for an STFT plan
    std::vector<float> fb_dense;   // host copy, n_mels x n_bins_fft
    SpecDev dev;
    void* d_blob;
};

extern "C" int amtx_spec_plan_create(amtx_spec_plan** out, int sample_rate, int n_fft, int hop_length, int win_length,
                                     int n_mels, int htk, int center, int pad_mode) {
    AMTX_REQUIRE(out != nullptr, "amtx_spec_plan_create: null plan pointer");
    *out = nullptr;
    if (n_fft < 128 || n_fft > 4096 || (n_fft & (n_fft - 1)) != 0) {
        amtx_set_error("amtx_spec_plan_create: n_fft must be a power of two in 128 .. 4096 (got %d)", n_fft);
        return AMTX_ERR_UNSUPPORTED;
    }
    const int M = n_fft / 2;   // (shadows the tuned kernel's compile-time constant: tables are sized by THIS plan's frame length)
    if (win_length <= 0) win_length = n_fft;
    AMTX_REQUIRE(win_length <= n_fft && hop_length > 0 && sample_rate > 0, "amtx_spec_plan_create: bad win/hop/sr");
    AMTX_REQUIRE(n_mels >= 0 && n_mels <= 64 * MAX_MEL_ROUNDS, "amtx_spec_plan_create: n_mels out of range (0..%d)", 64 * MAX_MEL_ROUNDS);
    AMTX_REQUIRE(pad_mode == AMTX_PAD_CONSTANT || pad_mode == AMTX_PAD_REFLECT, "amtx_spec_plan_create: bad pad_mode");

    amtx_spec_plan* pl = new amtx_spec_plan();
    pl->sample_rate = sample_rate; pl->n_fft = n_fft; pl->hop = hop_length; pl->win = win_length;
    pl->n_mels = n_mels; pl->htk = htk; pl->center = center; pl->pad_mode = pad_mode;
    pl->n_bins_fft = n_fft / 2 + 1;
    pl->n_out = n_mels > 0 ? n_mels : pl->n_bins_fft;
    pl->d_blob = nullptr;

    const double PI = 3.14159265358979323846;
    // periodic Hann of win_length, centre-padded to n_fft (librosa.stft / util.pad_center)
    std::vector<float> window(n_fft, 0.0f);
    const int lpad = (n_fft - win_length) / 2;
    for (int n = 0; n < win_length; ++n) window[lpad + n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / win_length));
    std::vector<float2> tw_fft(M), tw_post(M);
    for (int k = 0; k < M; ++k) {
        tw_fft[k] = make_float2((float)cos(2.0 * PI * k / M), (float)(-sin(2.0 * PI * k / M)));
        tw_post[k] = make_float2((float)cos(2.0 * PI * k / n_fft), (float)(-sin(2.0 * PI * k / n_fft)));
    }
    // librosa.filters.mel + the slot layout of the gather (build_mel_tables)
    MelHost mh;
    {
        const int mrc = build_mel_tables(sample_rate, n_fft, n_mels, htk, mh);
        if (mrc != AMTX_OK) { delete pl; return mrc; }
    }
    pl->fb_dense = mh.fb_dense;
    memcpy(pl->dev.round_max, mh.round_max, sizeof(pl->dev.round_max));
    memcpy(pl->dev.round_off, mh.round_off, sizeof(pl->dev.round_off));
    std::vector<int>& m_slot_row = mh.slot_row;
    std::vector<int>& m_slot_start = mh.slot_start;
    std::vector<float>& m_w = mh.w;
    if (m_w.empty()) m_w.push_back(0.0f);

    // one device blob for all tables
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t o_win = 0, o_twf = align16(o_win + window.size() * 4), o_twp = align16(o_twf + tw_fft.size() * 8);
    size_t o_ms = align16(o_twp + tw_post.size() * 8), o_mr = align16(o_ms + m_slot_start.size() * 4), o_mw = align16(o_mr + m_slot_row.size() * 4);
    size_t total = align16(o_mw + m_w.size() * 4);
    std::vector<char> host(total, 0);
    memcpy(host.data() + o_win, window.data(), window.size() * 4);
    memcpy(host.data() + o_twf, tw_fft.data(), tw_fft.size() * 8);
    memcpy(host.data() + o_twp, tw_post.data(), tw_post.size() * 8);
    memcpy(host.data() + o_ms, m_slot_start.data(), m_slot_start.size() * 4);
    memcpy(host.data() + o_mr, m_slot_row.data(), m_slot_row.size() * 4);
    memcpy(host.data() + o_mw, m_w.data(), m_w.size() * 4);
    hipError_t e = hipMalloc(&pl->d_blob, total);
    if (e == hipSuccess) e = hipMemcpy(pl->d_blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        amtx_set_error("amtx_spec_plan_create: device table upload failed: %s", hipGetErrorString(e));
        if (pl->d_blob) (void)hipFree(pl->d_blob);
        delete pl;
        return AMTX_ERR_HIP;
    }
    char* d = (char*)pl->d_blob;
    pl->dev.window = (const float*)(d + o_win);
    pl->dev.tw_fft = (const float2*)(d + o_twf);
    pl->dev.tw_post = (const float2*)(d + o_twp);
    pl->dev.mel_start = (const int*)(d + o_ms);
    pl->dev.mel_row = (const int*)(d + o_mr);
    pl->dev.mel_wt = (const float*)(d + o_mw);
    pl->dev.hop = hop_length; pl->dev.n_out = pl->n_out; pl->dev.n_mels = n_mels;
    pl->dev.center = center; pl->dev.pad_mode = pad_mode; pl->dev.n_fft = n_fft;
    *out = pl;
    return AMTX_OK;
}

// Host-only: the mel stage's slot layout for a (sample_rate, n_fft, n_mels, htk) plan, without creating one (no device needed).
// slot_row / slot_start: [64 * rounds] (capacity 64 * 8), round_max: [rounds] tap slots per round.  Returns the number of rounds.
extern "C" int amtx_spec_mel_layout(int sample_rate, int n_fft, int n_mels, int htk, int32_t* slot_row, int32_t* slot_start, int32_t* round_max) {
    AMTX_REQUIRE(slot_row && slot_start && round_max, "amtx_spec_mel_layout: null pointer");
    AMTX_REQUIRE(n_mels > 0 && n_mels <= 64 * MAX_MEL_ROUNDS && n_fft >= 128 && sample_rate > 0, "amtx_spec_mel_layout: bad sizes");
    MelHost mh;
    const int rc = build_mel_tables(sample_rate, n_fft, n_mels, htk, mh);
    if (rc != AMTX_OK) return rc;
    for (int i = 0; i < 64 * mh.rounds; ++i) { slot_row[i] = mh.slot_row[i]; slot_start[i] = mh.slot_start[i]; }
    for (int r = 0; r < mh.rounds; ++r) round_max[r] = mh.round_max[r];
    return mh.rounds;
}

extern "C" int amtx_spec_plan_destroy(amtx_spec_plan* plan) {
    if (!plan) return AMTX_OK;
    if (plan->d_blob) (void)hipFree(plan->d_blob);
    delete plan;
    return AMTX_OK;
}

extern "C" int amtx_spec_num_bins(const amtx_spec_plan* plan) { return plan ? plan->n_out : AMTX_ERR_ARG; }

extern "C" int64_t amtx_spec_num_frames(const amtx_spec_plan* plan, int64_t n) {
    if (!plan || n < 0) return AMTX_ERR_ARG;
    if (n == 0) return 0;
    if (plan->center) return 1 + n / plan->hop;                       // features/common.py:64
    // not centred: the reference zero-pads to a whole number of hops first (features/common.py:137-166)
    int64_t divisor = plan->win;
    if (n > divisor) divisor = plan->hop;
    const int64_t padded = ((n + divisor - 1) / divisor) * divisor;
    if (padded < plan->n_fft) return 0;
    return 1 + (padded - plan->n_fft) / plan->hop;
}

extern "C" int amtx_spec_filterbank(const amtx_spec_plan* plan, float* host_out) {
    AMTX_REQUIRE(plan && host_out, "amtx_spec_filterbank: null argument");
    AMTX_REQUIRE(plan->n_mels > 0, "amtx_spec_filterbank: plan has no mel filterbank");
    memcpy(host_out, plan->fb_dense.data(), plan->fb_dense.size() * sizeof(float));
    return AMTX_OK;
}

extern "C" int amtx_spec_power(const amtx_spec_plan* plan, const float* audio, int64_t num_samples, int64_t audio_stride,
                               int batch, float* power, float* clip_max, void* stream_) {
    AMTX_REQUIRE(plan && audio && power && clip_max, "amtx_spec_power: null argument");
    AMTX_REQUIRE(batch > 0 && num_samples > 0 && audio_stride >= num_samples, "amtx_spec_power: bad batch/num_samples/stride");
    if (plan->pad_mode == AMTX_PAD_REFLECT)
        AMTX_REQUIRE(num_samples > plan->n_fft / 2, "amtx_spec_power: reflect padding needs more than n_fft/2 samples");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t T = amtx_spec_num_frames(plan, num_samples);
    AMTX_REQUIRE(T > 0, "amtx_spec_power: clip too short for one frame");
    AMTX_CHECK_HIP(hipMemsetAsync(clip_max, 0, sizeof(float) * batch, stream));
    if (plan->n_fft != NFFT) {
        int log2m = 0;
        while ((2 << log2m) < plan->n_fft) ++log2m;
        const int m = plan->n_fft / 2, fpw = 8;
        const int64_t chunks_g = (T + fpw * WAVES - 1) / (fpw * WAVES);
        const int64_t nblocks_g = chunks_g * batch;
        AMTX_REQUIRE(nblocks_g < (1ll << 31), "amtx_spec_power: grid too large");
        const size_t lds_g = (size_t)WAVES * m * sizeof(float2) + (size_t)WAVES * (m + 1 + 128 + 3) * sizeof(float);
        AMTX_GRANT_LDS(spec_power_pow2_kernel, lds_g);
        hipLaunchKernelGGL(spec_power_pow2_kernel, dim3((unsigned)nblocks_g), dim3(256), lds_g, stream, plan->dev, log2m, fpw, audio, num_samples,
                           audio_stride, T, power, (unsigned*)clip_max, plan->n_mels > 0 ? 1 : 0);
        AMTX_CHECK_LAUNCH();
        return AMTX_OK;
    }
    constexpr int FPW = 8;
    constexpr int FPB = FPW * WAVES;
    const int64_t chunks = (T + FPB - 1) / FPB;
    const int64_t nblocks = chunks * batch;
    AMTX_REQUIRE(nblocks < (1ll << 31), "amtx_spec_power: grid too large");
    const int mel_rounds = plan->n_mels > 0 ? (plan->n_mels + 63) / 64 : 0;
    const int mel_slots = mel_rounds > 0 ? plan->dev.round_off[mel_rounds - 1] + plan->dev.round_max[mel_rounds - 1] : 0;
    const bool mel_lds = mel_rounds > 0 && mel_slots <= MEL_LDS_MAX_SLOTS;
    const size_t lds = WAVES * XB_ELEMS * sizeof(float2) + WAVES * PB_ELEMS * sizeof(float) + (M / 2 + 64) * sizeof(float2) +
                       2 * 64 * MAX_MEL_ROUNDS * sizeof(int) + (mel_lds ? (size_t)mel_slots * 64 * sizeof(float) : 0);
    auto launch = [&](auto kern) -> int {
        AMTX_GRANT_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), lds, stream, plan->dev, audio, num_samples, audio_stride, T, power,
                           (unsigned*)clip_max);
        return AMTX_OK;
    };
    // the BASELINE log-mel shape (hop 512, four mel rounds of 4 / 8 / 20 / 32 tap slots = librosa's 229 Slaney rows at 22.05 kHz) has its own
    // kernel: clip ring in LDS + mel weights in registers; AMTX_SPEC_NO_RING=1 keeps the general kernel (A/B and the bit-equality test)
    static const bool no_ring = getenv("AMTX_SPEC_NO_RING") != nullptr;
    const auto& dv = plan->dev;
    if (!no_ring && plan->hop == RING_HOP && mel_rounds == 4 && dv.round_max[0] == 4 && dv.round_max[1] == 8 && dv.round_max[2] == 20 &&
        dv.round_max[3] == 32 && dv.round_off[1] == 4 && dv.round_off[2] == 12 && dv.round_off[3] == 32) {
        auto kern = spec_power_ring_kernel<FPW, 4, 8, 20, 32>;
        const size_t lds_r = WAVES * XB_ELEMS * sizeof(float2) + WAVES * PB_ELEMS * sizeof(float) + (M / 2 + 64) * sizeof(float2) +
                             64 * 4 * sizeof(int) + RING_ELEMS * sizeof(float);
        AMTX_GRANT_LDS(kern, lds_r);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), lds_r, stream, plan->dev, audio, num_samples, audio_stride, T, power,
                           (unsigned*)clip_max);
        AMTX_CHECK_LAUNCH();
        return AMTX_OK;
    }
    int rc;
    if (mel_lds) rc = launch(spec_power_kernel<FPW, true, true>);
    else if (plan->n_mels > 0) rc = launch(spec_power_kernel<FPW, true, false>);
    else rc = launch(spec_power_kernel<FPW, false, false>);
    if (rc != AMTX_OK) return rc;
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

extern "C" int amtx_spec_scale(const amtx_spec_plan* plan, const float* power, const float* clip_max, const float* ref,
                               int batch, int64_t num_frames, int transform, int layout, float* out, void* stream_) {
    AMTX_REQUIRE(plan && power && out, "amtx_spec_scale: null argument");
    AMTX_REQUIRE(transform != AMTX_SCALE_DB || clip_max, "amtx_spec_scale: dB scaling needs clip_max");
    AMTX_REQUIRE(batch > 0 && batch < 65536 && num_frames > 0, "amtx_spec_scale: bad batch/num_frames");
    AMTX_REQUIRE(layout == AMTX_LAYOUT_BFT_F32 || layout == AMTX_LAYOUT_BTF_F32, "amtx_spec_scale: bad layout");
    hipStream_t stream = (hipStream_t)stream_;
    dim3 grid((unsigned)((num_frames + 31) / 32), (unsigned)((plan->n_out + 31) / 32), (unsigned)batch);
    hipLaunchKernelGGL(spec_scale_kernel, grid, dim3(256), 0, stream, power, clip_max, ref, num_frames, plan->n_out, transform,
                       layout, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

#ifdef AMTX_SPEC_TIMING
extern "C" int amtxdbg_spec_prof(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_spec_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_spec_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
