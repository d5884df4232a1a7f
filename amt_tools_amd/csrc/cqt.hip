// Constant-Q / Variable-Q / Harmonic CQT front-end for gfx950.
//
// Replaces librosa.vqt as called from amt_tools/features/vqt.py:183-193 (CQT: cqt.py:22; HVQT/HCQT: one VQT per
// harmonic, hvqt.py:45-58,107-133) plus abs -> amplitude_to_db(ref=max) -> /80+1 (features/common.py:199,218-228).
//
// librosa evaluates, per octave, a sparse FFT-domain wavelet basis against a "ones"-window STFT of the
// progressively 2x-decimated signal.  Here the same linear map is applied in the time domain: for every bank
// (harmonic, octave) the sparsified one-sided FFT basis is folded back on the host (fp64) into its equivalent
// n_fft-tap complex kernel, all banks that live on the same decimation level are stacked into one matrix, and the
// response of a level is ONE strided-row GEMM on the matrix cores
//        R_d[t, :] = W_d . y_d[t*hop_d - n_fft/2 : t*hop_d + n_fft/2]          (A rows overlap: lda = hop_d)
// in split-bf16 "x3" arithmetic (fp32-class accuracy: a bf16-only product would leak a strong partial into bins
// 50 dB below it).  Harmonics that are octaves apart share the pyramid level and the launch.
// The pyramid itself is a 301-tap Kaiser half-band decimator (the project's documented stand-in for soxr, which
// is not bit-reproducible anyway), then |.|, per-(clip, harmonic) max, dB, /80+1.

#include "amtx_kernels.h"

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace {

constexpr double DEC_CUTOFF = 0.239;
constexpr double DEC_BETA = 10.0;
constexpr int MAX_BANKS = 16;          // banks per pyramid level
constexpr double PI = 3.14159265358979323846;
constexpr double HANN_BW = 1.50018310546875;

struct BankDev { int col0, nf, harm, bin0, frames; };
struct LevelDev { int nbanks; int ncols; BankDev b[MAX_BANKS]; };


double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 200; ++k) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

// ---------------------------------------------------------------- kernels
// maximum over the 16 lanes of a DPP row, in every lane of it
__device__ __forceinline__ float row_max_f32(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));   // row_half_mirror
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));   // row_mirror
    return v;
}
typedef __attribute__((ext_vector_type(8))) __bf16 cq_bf16x8;
__device__ __forceinline__ f32x4_t cq_mfma(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cq_bf16x8, a), __builtin_bit_cast(cq_bf16x8, b), c, 0, 0, 0);
}

// level 0 of the pyramid: the clip copied between its centre paddings.  `zero_pads`: the paddings are written here too (zeros, librosa
// >= 0.10; the reflecting pad of 0.9 keeps its own kernel); the per-(clip, harmonic) maxima the basis products accumulate are reset.
__global__ __launch_bounds__(256) void cqt_level0_kernel(const float* __restrict__ audio, int64_t n, int64_t astride, float* __restrict__ pyr,
                                                         int64_t pstride, int pad, int zero_pads, float* __restrict__ maxbuf, int n_harm) {
    const int b = blockIdx.y;
    const float* src = audio + (int64_t)b * astride;
    float* row = pyr + (int64_t)b * pstride;
    if (blockIdx.x == 0 && (int)threadIdx.x < n_harm) maxbuf[b * n_harm + threadIdx.x] = 0.f;
    if (zero_pads) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pstride; i += (int64_t)gridDim.x * 256)
            row[i] = (i >= pad && i < pad + n) ? src[i - pad] : 0.f;
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) row[pad + i] = src[i];
    }
}

// centre padding of one pyramid level: zeros (librosa >= 0.10) or reflection (librosa 0.9)
__global__ __launch_bounds__(256) void cqt_pad_kernel(float* __restrict__ pyr, int64_t n, int64_t stride, int pad, int reflect) {
    const int b = blockIdx.y;
    float* row = pyr + (int64_t)b * stride;
    const int64_t tail = stride - pad - n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pad + tail; i += (int64_t)gridDim.x * 256) {
        if (i < pad) {                                      // left pad, position pad-1-i  <->  sample index i+1
            float v = 0.f;
            if (reflect && i + 1 < n) v = row[pad + i + 1];
            row[pad - 1 - i] = v;
        } else {                                            // right region, position pad+n+j  <->  sample n-2-j
            const int64_t j = i - pad;
            float v = 0.f;
            if (reflect && j < pad && n - 2 - j >= 0) v = row[pad + n - 2 - j];
            row[pad + n + j] = v;
        }
    }
}

// amplitude_to_db(ref = max) -> clamp -80 -> /80 + 1 (or plain magnitude), truncated to t_out frames; from the POWER map
__global__ __launch_bounds__(256) void cqt_scale_kernel(const float* __restrict__ mag, const float* __restrict__ maxbuf, int n_bins, int64_t t_buf,
                                                        int64_t t_out, int decibels, float* __restrict__ out) {
    const int bh = blockIdx.y;
    const float ref = maxbuf[bh];                            // `mag` and `maxbuf` hold POWERS (re^2 + im^2 as the basis products' epilogue leaves them:
                                                             // no root there, no square here)
    const float amin2 = 1e-10f;                              // amin = 1e-5 on magnitude
    const float offs = 10.0f * log10f(fmaxf(amin2, ref));
    const float floor_db = (10.0f * log10f(fmaxf(amin2, ref)) - offs) - 80.0f;
    const int64_t total = (int64_t)n_bins * t_out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i / t_out);
        const int64_t t = i - (int64_t)k * t_out;
        const float a = mag[((int64_t)bh * n_bins + k) * t_buf + t];
        float v;
        if (decibels) {
            float db = 10.0f * log10f(fmaxf(amin2, a)) - offs;
            db = fmaxf(db, floor_db);
            v = db / 80.0f + 1.0f;
        } else {
            v = sqrtf(a);
        }
        out[((int64_t)bh * n_bins + k) * t_out + t] = v;
    }
}

// The same scaling written the way the engine's fused first conv stages its features (convg.hip, tap-major: ConvArgs.feats16):
// out16[b][t][bin][8] bf16, the harmonics of a (frame, bin) position side by side in one 16-byte slot (slots n_harm .. 7 zero) -- that kernel
// then fetches a position with one 16-byte load instead of n_harm strided fp32 loads, conversions and 2-byte LDS stores (29 % of its tile),
// and the map is written once in 4 bytes less per value.  The values are the fp32 ones of cqt_scale_kernel rounded to bf16 by the conversion
// the conv kernel applies itself: the same bits reach the matrix cores.  A block transposes 32 frames x (a chunk of) the bins of one clip
// through LDS: reads run along t (mag's inner axis), writes along (t, bin).
constexpr int SC16_TT = 32;
constexpr int SC16_U = 9;          // 72 bins = one round of 8 rows x 9
__global__ __launch_bounds__(256) void cqt_scale16_kernel(const float* __restrict__ mag, const float* __restrict__ maxbuf, int n_harm, int n_bins, int nbc,
                                                          int64_t t_buf, int64_t t_out, int decibels, uint4* __restrict__ out16, int64_t lo_plane) {
    // lo_plane != 0 (amtx_cqt_forward16_split, the x3 engine's hand-over): a second map, lo = 16-bit(v - 16-bit(v)), lo_plane 16-byte slots
    // behind the first -- the two planes split_bf16x2 makes of the fp32 feature, i.e. what the two-plane conv kernels made of it themselves
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [planes][SC16_TT][nbc + 1] slots of 16 bytes, then 8 x 2 floats
    const int tid = threadIdx.x, b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * SC16_TT;
    const int rs = nbc + 1;                                   // slots per frame row (+ 1: rows start 4 banks apart)
    const int npl = lo_plane ? 2 : 1;
    uint4* tile = reinterpret_cast<uint4*>(smem);
    uint4* tile_lo = tile + SC16_TT * rs;
    float* refs = reinterpret_cast<float*>(smem + (size_t)npl * SC16_TT * rs * 16);
    const float amin2 = 1e-10f;                              // amin = 1e-5 on magnitude
    if (tid < n_harm) {
        const float ref = maxbuf[b * n_harm + tid];            // a power, like `mag` (cqt_scale_kernel)
        const float offs = 10.0f * log10f(fmaxf(amin2, ref));
        refs[2 * tid] = offs;
        refs[2 * tid + 1] = (10.0f * log10f(fmaxf(amin2, ref)) - offs) - 80.0f;
    }
    for (int i = tid; i < npl * SC16_TT * rs; i += 256) tile[i] = make_uint4(0, 0, 0, 0);
    const int tt = tid & 31, r = tid >> 5;
    const bool t_ok = t0 + tt < t_out;
    for (int bin0 = 0; bin0 < n_bins; bin0 += nbc) {
        const int nb = min(nbc, n_bins - bin0);
        __syncthreads();                                      // tile zeroed / written out; refs visible
        // SC16_U loads in flight per thread before the first is used (one at a time -- load, log, store -- a block spent its life waiting:
        // 0.54 ms for the kernel instead of the 0.27 ms of cqt_scale_kernel)
        for (int h = 0; h < n_harm; ++h) {
            const float* row = mag + ((int64_t)(b * n_harm + h) * n_bins + bin0) * t_buf + t0 + tt;
            const float offs = refs[2 * h], floor_db = refs[2 * h + 1];
            for (int k0 = r; k0 < nb; k0 += 8 * SC16_U) {
                float a[SC16_U];
#pragma unroll
                for (int u = 0; u < SC16_U; ++u) a[u] = t_ok ? row[(int64_t)min(k0 + 8 * u, nb - 1) * t_buf] : 0.f;
#pragma unroll
                for (int u = 0; u < SC16_U; ++u) {
                    const int k = k0 + 8 * u;
                    float v;
                    if (decibels) {
                        float db = 10.0f * log10f(fmaxf(amin2, a[u])) - offs;
                        db = fmaxf(db, floor_db);
                        v = db / 80.0f + 1.0f;
                    } else {
                        v = sqrtf(a[u]);
                    }
                    if (k < nb) {
                        uint32_t hi, lo;
                        split_bf16x2(v, 0.f, hi, lo);         // hi = the rounding pack_bf16x2 applies
                        reinterpret_cast<unsigned short*>(tile + tt * rs + k)[h] = (unsigned short)hi;
                        if (lo_plane) reinterpret_cast<unsigned short*>(tile_lo + tt * rs + k)[h] = (unsigned short)lo;
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < SC16_TT * nb; i += 256) {
            const int ti = i / nb, k = i - ti * nb;
            if (t0 + ti < t_out) {
                const int64_t o = ((int64_t)b * t_out + t0 + ti) * n_bins + bin0 + k;
                out16[o] = tile[ti * rs + k];
                if (lo_plane) out16[lo_plane + o] = tile_lo[ti * rs + k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The basis products of every pyramid level, one launch (round 5; until then amtx_launch_gemm_multi ran them on the generic fp32-A
// two-plane GEMM).  A row of a level's product is a WINDOW of its signal: row t = y[t hop - K / 2 .. t hop + K / 2), K = n_fft = 128 / 256,
// hop = 512 .. 4 -- consecutive rows overlap by K - hop samples, up to 63 / 64 of a row.  The generic GEMM loaded and converted (fp32 ->
// hi / lo planes, ~3 vector instructions per value) every row on its own: each sample of a deep level up to 64 times.  Here a block
// stages the signal of FT consecutive frames of one clip ONCE (coalesced fp32 loads, one split per sample, two 16-bit planes in LDS) and
// every row's MFMA operand is a 16-byte window of that copy; the level's basis (<= 256 columns x K, two planes) sits in the waves'
// registers for the whole launch.  Frames run down the lanes (lane & 15), so a frame's window starts hop samples after its neighbour's:
// the copy is stored with 16 pad samples behind every hop samples (hop >= 32) -- a frame stride of 2 (mod 4) 16-byte slots, see bas_pad --
// which makes every ds_read_b128 lane group conflict-free; hop = 16 and 8 need none, hop = 4 reads two 8-byte halves.
// Same planes, same product order (hi.hi, hi.lo, lo.hi per 32-deep step, k ascending) and the same epilogue (|re + i im|^2 transposed into
// mag[b][harmonic][bin][t], per-(clip, harmonic) maxima) as gemm_tile: the same bits.
#ifdef AMTX_CQT_TIMING
// debug build only (tools/build_dbg.sh cqtbasis cqt.hip -DAMTX_CQT_TIMING; tools/cqt_basis_prof.py): cycles wave 0 of every block of
// cqt_basis_kernel spends per phase: [0] tile setup, [1] staging (loads, split, LDS stores), [2] barrier, [3] matrix loop + magnitude stores,
// [4] maxima + barrier + global atomics, [7] tiles
__device__ unsigned long long g_bas_prof[8];
#define BQ_TICK(SLOT) do { const unsigned long long now_ = __builtin_readcyclecounter(); bq_acc[SLOT] += now_ - bq_t; bq_t = now_; } while (0)
extern "C" int amtxdbg_bas_prof(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bas_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bas_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define BQ_TICK(SLOT) do {} while (0)
#endif
// pad samples behind every `hop` samples of the staged copy: the frame stride must be 2 (mod 4) 16-byte slots for the operand reads to be
// conflict-free -- ds_read_b128 serves lanes {0-3, 12-15} of one k-group together with lanes {4-11} of the NEXT (one slot further), so an odd
// stride (the 8 samples of the first version: 9, 5, 3 slots) put seven of every eight of them on a slot twice: 48 % of the kernel's LDS
// cycles were conflicts, and its operand reads as many cycles as its matrix instructions (tools/lds_swizzle_check_basis.py)
__host__ __device__ constexpr int bas_pad(int hop) { return hop >= 32 ? 16 : 0; }
constexpr int BAS_MAXLEV = 10;
constexpr int BAS_AHEAD = 1;            // k-steps the operand reads of cqt_basis_kernel run ahead
constexpr int BAS_MAXWIN = 16384;         // samples of a tile's window at most (66 KB of LDS with the pads: two blocks per CU)
struct BasisLevel {
    const float* sig; int64_t sig_gs;      // level signal of clip 0 (sample 0 = first sample of the clip at this level), elements per clip
    int64_t len;                           // samples that exist: indices [0, len) of `sig`; everything else reads as zero
    int off;                               // index of the level's sample 0 in `sig` (librosa 0.9: the reflecting pad in front of it exists)
    const bf16_t* w; const int2* map;      // packed basis [2][n_pad][k_pad], pair map
    int n_pad, k_pad, ncols, hop, frames, ft, tiles_per_clip;
    int tile0;                             // first tile id of this level in the launch
};
struct BasisArgs {
    BasisLevel lev[BAS_MAXLEV];
    int nlev, ntiles, B;
    float* mag; int64_t mag_gs, mag_pitch; // mag[b * mag_gs + row * mag_pitch + t]
    float* maxbuf; int n_harm;
    int rows[16];                          // frames of harmonic h
};

struct BasisLevs { int n; int lev[BAS_MAXLEV]; int b0[BAS_MAXLEV + 1]; };   // levels of a launch; blocks [b0[i], b0[i + 1]) work on level lev[i]

// grid: (blocks of all levels of this K, column slices of 128): all levels of a call run side by side, the small deep ones under the large
// shallow ones.  256-thread blocks, two per CU: a wave keeps the basis of ITS two column tiles in registers and walks the tile's frames on
// its own (a dependent read -> MFMA chain per 16 frames), so what fills the matrix pipe is the other waves of the CU; a level with more than
// 128 columns (HCQT's middle levels stack 144) runs a second slice of blocks for the rest, which stages the same signal again (an L2 hit).
template <int KS>
__global__ __launch_bounds__(256, 2) void cqt_basis_kernel(BasisArgs a, BasisLevs ls) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [hi plane | lo plane] of the padded window, then 16 maxima
    constexpr int K = 32 * KS;
    // this block's level: the launch deals its blocks to the levels by their work (staged samples + a per-tile charge)
    int li = 0;
    while (li + 1 < ls.n && (int)blockIdx.x >= ls.b0[li + 1]) ++li;
    const BasisLevel& L = a.lev[ls.lev[li]];
    const int bid = (int)blockIdx.x - ls.b0[li], nblk = ls.b0[li + 1] - ls.b0[li];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // `hop` below is the frame stride of the STAGED copy: the level's hop, or K where the hop is larger (level 0, hop 512 > K = 256: rows do
    // not overlap and only the K samples of every frame are staged, back to back)
    const int ghop = L.hop, ft = L.ft;
    const int hop = min(ghop, K), lh = __builtin_ctz(hop);        // powers of two
    const int pad = bas_pad(hop);                                  // pad samples behind every `hop`
    const int ws = (ft - 1) * hop + K;                             // samples of a tile's window
    const int wsp = ws + pad * ((ws + hop - 1) / hop);             // ... in LDS
    const int plane = ((wsp + 7) & ~7) * 2;                        // bytes per plane
    unsigned* lmax = reinterpret_cast<unsigned*>(smem + 2 * plane);

    // ---- this wave's columns: two 16-column tiles per wave, four waves = one slice of 128 columns; waves beyond the slice's columns share
    // the frame tiles of a column group
    const int col0 = 128 * blockIdx.y;                             // first column of this slice
    if (col0 >= L.ncols) return;
    const int nct = (min(L.ncols - col0, 128) + 15) >> 4;          // 16-column tiles with real columns in the slice (<= 8)
    const int ncg = (nct + 1) >> 1;                                // column groups of two tiles (1 .. 4)
    const int wpg = 4 / ncg;                                       // waves per group (4, 2, 1, 1)
    const bool idle = wave >= ncg * wpg;
    const int cg = (idle ? 0 : wave % ncg) + 4 * blockIdx.y, sub = wave / ncg;
    const int g = lane >> 4, fl = lane & 15;
    uint4 wh[2][KS], wl[2][KS];
    {
        const int64_t wplane = (int64_t)L.n_pad * L.k_pad;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = min((2 * cg + c) * 16 + fl, L.n_pad - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wh[c][ks] = *reinterpret_cast<const uint4*>(L.w + (int64_t)n * L.k_pad + ks * 32 + 8 * g);
                wl[c][ks] = *reinterpret_cast<const uint4*>(L.w + wplane + (int64_t)n * L.k_pad + ks * 32 + 8 * g);
            }
        }
    }
    // pair maps of this lane's four columns per tile: columns n0 + 4 g + {0, 1} = (re, im) of one filter, + {2, 3} of the next
    int2 pm[2][2];
    bool pv[2][2];
    int rl[2][2], ro[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = (2 * cg + c) * 16 + 4 * g + 2 * h;
            pv[c][h] = !idle && n < L.ncols;
            pm[c][h] = L.map[min(n, L.ncols - 2) >> 1];
            // frames of this filter's harmonic (0: the lane's columns do not exist): read ONCE -- indexed per lane inside the frame loop it was
            // a memory round trip per filter and 16 frames
            rl[c][h] = pv[c][h] ? a.rows[pm[c][h].y & 15] : 0;
            ro[c][h] = pm[c][h].x * (int)a.mag_pitch;            // row offset inside the clip's map
        }
    // byte offset of this lane's operand inside a frame's window: sample 8 g of k-step 0 (+ 32 samples per k-step), pads included
    auto soff = [&](int s) { return (s + pad * (s >> lh)) * 2; };
    // ... split for the matrix loop: frame f starts at byte f * fstride, k-step ks of this lane koff[ks] bytes further
    const int fstride = 2 * (hop + pad);
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = soff(ks * 32 + 8 * g);

    // ---- tile loop.  Staging: four samples per thread and step (a 16-byte load at any 4-byte boundary: clips of odd length in one buffer), in
    // ROUNDS of six steps whose loads are issued back to back from clamped addresses (a load -> test -> store loop pays one memory round
    // trip per step); what lies outside the signal is zeroed at the split.  Staged indices are multiples of 4 and so are the window origins,
    // so a group hangs over the signal's END only (by 1 - 3 samples when the length is not a multiple of 4): its values are picked out of
    // the clamped load by the distance it was moved.  The block's barriers wait for LDS traffic only (the stores of a tile's magnitudes
    // and the maxima's atomics stay in flight), and the other block of the CU covers a tile's load round trips.
    struct __attribute__((packed, aligned(4))) f32x4_a4 { float x, y, z, w; };
    constexpr int NST = 6, NSTF = 9;             // loads per round: the general path / a tile inside the signal (fewer temporaries: 36 registers of loads fit)
    const int ntile = L.tiles_per_clip * a.B;
    if (tid < 16) lmax[tid] = 0u;
#ifdef AMTX_CQT_TIMING
    unsigned long long bq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bq_t = __builtin_readcyclecounter();
#endif
    for (int tile = bid; tile < ntile; tile += nblk) {
        BQ_TICK(0);
        const int b = tile / L.tiles_per_clip, tt = tile - b * L.tiles_per_clip;
        const int f0 = tt * ft;                                    // first frame of the tile
        const float* sig = L.sig + (int64_t)b * L.sig_gs;
        const int64_t s0 = (int64_t)f0 * ghop - K / 2 + L.off;     // first sample of the window (index into sig)
        // staged index i -> signal index: identity while the windows overlap, frame-wise (hop > K) otherwise
        auto sidx = [&](int i) { return s0 + (ghop > K ? (int64_t)(i >> lh) * ghop + (i & (hop - 1)) : (int64_t)i); };
        // a tile whose whole window lies inside the signal (all but a clip's first and last tiles): no clamping, no picking -- a third of the
        // instructions of the general case below (which was 37 % of the kernel's vector instructions)
        const int64_t s_end = s0 + (ghop > K ? (int64_t)(ft - 1) * ghop + K : (int64_t)ws);
        if (s0 >= 0 && s_end <= L.len) {
            const float* st = sig + s0;                            // wave-uniform base, 32-bit offsets per lane
            for (int base = 0; base < ws; base += 1024 * NSTF) {
                f32x4_a4 rr[NSTF];
#pragma unroll
                for (int n = 0; n < NSTF; ++n) {
                    const int i = min(base + 4 * tid + 1024 * n, ws - 4);
                    rr[n] = *reinterpret_cast<const f32x4_a4*>(st + (unsigned)(ghop > K ? (i >> lh) * ghop + (i & (hop - 1)) : i));
                }
#pragma unroll
                for (int n = 0; n < NSTF; ++n) {
                    const int i = base + 4 * tid + 1024 * n;
                    uint32_t h0, h1, l0, l1;
                    split_bf16x2(rr[n].x, rr[n].y, h0, l0);
                    split_bf16x2(rr[n].z, rr[n].w, h1, l1);
                    if (i < ws) {
                        const int o = soff(i);
                        *reinterpret_cast<uint2*>(smem + o) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(smem + plane + o) = make_uint2(l0, l1);
                    }
                }
            }
        } else
        for (int base = 0; base < ws; base += 1024 * NST) {
            f32x4_a4 rr[NST];
#pragma unroll
            for (int n = 0; n < NST; ++n) {
                const int64_t q = sidx(min(base + 4 * tid + 1024 * n, ws - 4));
                rr[n] = *reinterpret_cast<const f32x4_a4*>(sig + min(max(q, (int64_t)0), L.len - 4));
            }
#pragma unroll
            for (int n = 0; n < NST; ++n) {
                const int i = base + 4 * tid + 1024 * n;
                const int64_t q = sidx(min(i, ws - 4));
                // 0: as loaded; 1 - 3: the load was moved back by that much; 4: nothing of the group exists
                const int d = (q < 0 || q >= L.len) ? 4 : (int)(q - min(q, L.len - 4));
                const float r[4] = {rr[n].x, rr[n].y, rr[n].z, rr[n].w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // element e of the group is element e + d of the load, if that exists
                    float x = r[e];
                    if (e + 1 < 4) x = d == 1 ? r[e + 1] : x;
                    if (e + 2 < 4) x = d == 2 ? r[e + 2] : x;
                    if (e + 3 < 4) x = d == 3 ? r[e + 3] : x;
                    v[e] = e + d < 4 ? x : 0.f;
                }
                uint32_t h0, h1, l0, l1;
                split_bf16x2(v[0], v[1], h0, l0);
                split_bf16x2(v[2], v[3], h1, l1);
                if (i < ws) {
                    const int o = soff(i);                         // four samples never straddle a pad (blocks are multiples of 4)
                    *reinterpret_cast<uint2*>(smem + o) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(smem + plane + o) = make_uint2(l0, l1);
                }
            }
        }
        BQ_TICK(1);
        lds_only_barrier();
        BQ_TICK(2);
        float mx[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        const int nft = idle ? 0 : min(ft, L.frames - f0 + 15) >> 4;   // 16-frame tiles with real frames (a wave without columns has none)
        float* magb = a.mag + (int64_t)b * a.mag_gs;               // wave-uniform base, 32-bit offsets per lane (a clip's map is < 2^31 elements)
        // TWO 16-frame tiles per step: four independent accumulator chains and twice the reads in flight (one tile at a time, a wave waited for
        // every k-step's reads in front of its six dependent MFMAs)
        for (int q = sub; q < nft; q += 2 * wpg) {
            const int q2 = q + wpg < nft ? q + wpg : q;            // the second tile of the step (the first again when there is none: not stored)
            const int fr[2] = {q * 16 + fl, q2 * 16 + fl};         // this lane's frames inside the tile
            f32x4_t acc[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[u][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            // operand reads run BAS_AHEAD k-steps ahead of the matrix instructions that use them, both pinned: left to itself hipcc reads a
            // k-step's four operands right in front of its twelve instructions (an LDS round trip per k-step, ~45 % on top of the matrix time)
            const int fbase[2] = {fr[0] * fstride, fr[1] * fstride};
            constexpr int NB = BAS_AHEAD + 1;              // operand sets in flight (a ring)
            uint4 ah[NB][2], al[NB][2];
            auto fetch = [&](int ks) {
                const int sl = ks % NB;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int o = fbase[u] + koff[ks];
                    if (hop >= 8) {
                        ah[sl][u] = *reinterpret_cast<const uint4*>(smem + o);
                        al[sl][u] = *reinterpret_cast<const uint4*>(smem + plane + o);
                    } else {                                       // hop 4: windows start at 8-byte boundaries
                        const uint2 h0 = *reinterpret_cast<const uint2*>(smem + o), h1 = *reinterpret_cast<const uint2*>(smem + o + 8);
                        const uint2 l0 = *reinterpret_cast<const uint2*>(smem + plane + o), l1 = *reinterpret_cast<const uint2*>(smem + plane + o + 8);
                        ah[sl][u] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                        al[sl][u] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                    }
                }
            };
#pragma unroll
            for (int ks = 0; ks < BAS_AHEAD; ++ks) fetch(ks);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + BAS_AHEAD < KS) fetch(ks + BAS_AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                // per accumulator hi.hi, hi.lo, lo.hi (gemm_tile's order), the four accumulators taking turns
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            acc[u][c] = cq_mfma(pr == 2 ? wl[c][ks] : wh[c][ks], pr == 1 ? al[ks % NB][u] : ah[ks % NB][u], acc[u][c]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && q2 == q) break;                      // wave-uniform
                const int m = f0 + fr[u];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (m < rl[c][h]) {
                            const float re = acc[u][c][2 * h], im = acc[u][c][2 * h + 1];
                            const float v = re * re + im * im;           // power: the scaling kernels take its logarithm (or its root)
                            magb[(unsigned)(ro[c][h] + m)] = v;
                            mx[c][h] = fmaxf(mx[c][h], v);
                        }
                    }
            }
        }
        BQ_TICK(3);
        // per-(clip, harmonic) maxima: magnitudes are >= 0, so uint order == float order
        // the 16 lanes of a row (one lane group g) hold the same four filters: their maximum by DPP, one LDS atomic per row and filter (64
        // lanes on <= 6 addresses each were half of the kernel's LDS bank-conflict cycles)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float m = row_max_f32(mx[c][h]);
                if (fl == 0 && m > 0.f) atomicMax(lmax + (pm[c][h].y & 15), __float_as_uint(m));
            }
        lds_only_barrier();                                        // everybody is done with the staged copy, the maxima are in
        if (tid < 16) {
            const unsigned v = lmax[tid];
            lmax[tid] = 0u;                                        // for the next tile (ordered behind its barrier)
            if (v != 0u) atomicMax(reinterpret_cast<unsigned*>(a.maxbuf) + (int64_t)b * a.n_harm + tid, v);
        }
        BQ_TICK(4);
#ifdef AMTX_CQT_TIMING
        bq_acc[7] += 1;
#endif
    }
#ifdef AMTX_CQT_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_bas_prof[i], bq_acc[i]);
#endif
}

struct Level {
    int nfft = 0, ncols = 0, hop = 0;
    std::vector<BankDev> banks;
    std::vector<float> W;          // [ncols][nfft] fp32 row-major (built incrementally per bank at bank nfft, re-centred later)
    std::vector<std::pair<int, std::vector<std::complex<double>>>> raw;   // (bank nfft, kernels nf x nfft)
    bf16_t* d_w = nullptr;
    int2* d_map = nullptr;          // per output pair (filter): {row of the (harmonic, bin) in the magnitude map, harmonic}
    int n_pad = 0, k_pad = 0;
};

}  // namespace

struct amtx_cqt_plan {
    int sr, hop, n_bins, bpo, n_harm, lib09, truncate;
    double fmin, gamma, alpha;
    std::vector<double> harmonics;
    std::vector<int> early;           // early-downsample count per harmonic
    std::vector<Level> levels;
    int pad = 0;                      // centre padding of every pyramid level (max n_fft / 2)
    void* d_tfrag = nullptr;                       // Toeplitz fragments of the decimator for cqt_decimate_mfma_kernel
};

namespace {

double alpha_of(int bpo, int lib09) {
    const double r = std::pow(2.0, 1.0 / bpo);
    return lib09 ? (r - 1.0) : (r * r - 1.0) / (r * r + 1.0);
}

int num_two_factors(int x) {
    int n = 0;
    while (x > 0 && x % 2 == 0) { ++n; x /= 2; }
    return n;
}

int early_count(double nyquist, double cutoff, int hop, int n_oct) {
    const int c1 = std::max(0, (int)std::ceil(std::log2(nyquist / cutoff)) - 1 - 1);
    const int c2 = std::max(0, num_two_factors(hop) - n_oct + 1);
    return std::min(c1, c2);
}

// time-domain equivalent of librosa's sparsified one-sided FFT basis of one octave (see header comment)
void build_bank(const std::vector<double>& freqs, double sr_i, double sr_base, double gamma, double alpha,
                const std::vector<double>& len_base, int& nfft, std::vector<std::complex<double>>& kern) {
    const int nf = (int)freqs.size();
    const double Q = 1.0 / alpha;
    std::vector<double> ilen(nf);
    double maxlen = 0;
    for (int k = 0; k < nf; ++k) { ilen[k] = Q * sr_i / (freqs[k] + gamma / alpha); maxlen = std::max(maxlen, ilen[k]); }
    nfft = (int)std::llround(std::pow(2.0, std::ceil(std::log2(maxlen))));
    kern.assign((size_t)nf * nfft, 0.0);
    const int nb = nfft / 2 + 1;
    std::vector<std::complex<double>> tw(nfft);
    for (int i = 0; i < nfft; ++i) tw[i] = std::polar(1.0, -2.0 * PI * i / nfft);
    std::vector<std::complex<double>> basis(nfft), spec(nb);
    std::vector<double> mags(nb), sorted(nb);
    for (int k = 0; k < nf; ++k) {
        const long n0 = (long)std::floor(-ilen[k] / 2.0), n1 = (long)std::floor(ilen[k] / 2.0);
        const int cnt = (int)(n1 - n0);
        std::fill(basis.begin(), basis.end(), 0.0);
        const int lpad = (nfft - cnt) / 2;
        double l1 = 0;
        for (int n = 0; n < cnt; ++n) l1 += 0.5 - 0.5 * std::cos(2.0 * PI * n / cnt);
        for (int n = 0; n < cnt; ++n) {
            const double w = (0.5 - 0.5 * std::cos(2.0 * PI * n / cnt)) / l1;
            basis[lpad + n] = std::polar(w, 2.0 * PI * freqs[k] * (double)(n0 + n) / sr_i) * (ilen[k] / nfft);
        }
        double norm = 0;
        for (int f = 0; f < nb; ++f) {
            std::complex<double> s = 0;
            for (int n = lpad; n < lpad + cnt; ++n) s += basis[n] * tw[(int)(((long long)f * n) % nfft)];
            spec[f] = s;
            mags[f] = std::abs(s);
            norm += mags[f];
        }
        sorted = mags;
        std::sort(sorted.begin(), sorted.end());
        double cum = 0, thr = sorted[0];
        for (int f = 0; f < nb; ++f) {           // first index whose cumulative normalised magnitude reaches the 1 % quantile
            cum += sorted[f] / norm;
            if (!(cum < 0.01)) { thr = sorted[f]; break; }
        }
        const double scale = std::sqrt(sr_base / sr_i) / std::sqrt(len_base[k]);
        for (int n = 0; n < nfft; ++n) {
            std::complex<double> s = 0;
            for (int f = 0; f < nb; ++f)
                if (mags[f] >= thr) s += spec[f] * tw[(int)(((long long)f * n) % nfft)];
            kern[(size_t)k * nfft + n] = s * scale;
        }
    }
}

}  // namespace

extern "C" int amtx_cqt_plan_destroy(amtx_cqt_plan* p) {
    if (!p) return AMTX_OK;
    for (auto& l : p->levels) {
        if (l.d_w) (void)hipFree(l.d_w);
        if (l.d_map) (void)hipFree(l.d_map);
    }
    if (p->d_tfrag) (void)hipFree(p->d_tfrag);
    delete p;
    return AMTX_OK;
}

extern "C" int amtx_cqt_plan_create(amtx_cqt_plan** out, int sample_rate, int hop_length, double fmin, int n_bins, int bins_per_octave,
                                    double gamma, const double* harmonics, int n_harmonics, int truncate_to_expected, int librosa09) {
    AMTX_REQUIRE(out, "amtx_cqt_plan_create: null plan pointer");
    *out = nullptr;
    AMTX_REQUIRE(sample_rate > 0 && hop_length > 0 && fmin > 0 && n_bins > 0 && bins_per_octave > 0 && gamma >= 0 && n_harmonics > 0 &&
                     n_harmonics <= MAX_BANKS && harmonics, "amtx_cqt_plan_create: bad argument");
    amtx_cqt_plan* p = new amtx_cqt_plan();
    p->sr = sample_rate; p->hop = hop_length; p->n_bins = n_bins; p->bpo = bins_per_octave; p->n_harm = n_harmonics;
    p->lib09 = librosa09; p->truncate = truncate_to_expected; p->fmin = fmin; p->gamma = gamma;
    p->alpha = alpha_of(bins_per_octave, librosa09);
    p->harmonics.assign(harmonics, harmonics + n_harmonics);
    const int n_oct = (n_bins + bins_per_octave - 1) / bins_per_octave;
    const int nf_full = std::min(bins_per_octave, n_bins);
    const double Q = 1.0 / p->alpha;
    for (int h = 0; h < n_harmonics; ++h) {
        const double f0 = harmonics[h] * fmin;
        std::vector<double> freqs(n_bins);
        double cutoff = 0;
        for (int k = 0; k < n_bins; ++k) {
            freqs[k] = f0 * std::pow(2.0, (double)k / bins_per_octave);
            cutoff = std::max(cutoff, freqs[k] * (1 + 0.5 * HANN_BW / Q) + 0.5 * gamma);
        }
        if (cutoff > sample_rate / 2.0) {
            amtx_set_error("amtx_cqt_plan_create: wavelet basis with max frequency=%g Hz (harmonic %g) would exceed the Nyquist frequency=%g",
                           freqs[n_bins - 1], harmonics[h], sample_rate / 2.0);
            amtx_cqt_plan_destroy(p);
            return AMTX_ERR_ARG;
        }
        const int c = early_count(sample_rate / 2.0, cutoff, hop_length, n_oct);
        p->early.push_back(c);
        const double sr_base = sample_rate / std::pow(2.0, c);
        std::vector<double> len_all(n_bins);
        for (int k = 0; k < n_bins; ++k) len_all[k] = Q * sr_base / (freqs[k] + gamma / p->alpha);
        for (int j = 0; j < n_oct; ++j) {
            const int hi = n_bins - nf_full * j, lo = std::max(0, n_bins - nf_full * (j + 1));
            if (hi <= 0) break;
            const int level = c + j;
            if ((hop_length >> level) << level != hop_length || (hop_length >> level) < 4) {
                amtx_set_error("amtx_cqt_plan_create: hop_length=%d is not divisible far enough for %d pyramid levels (level hop must be a "
                               "multiple of 4)", hop_length, level + 1);
                amtx_cqt_plan_destroy(p);
                return AMTX_ERR_UNSUPPORTED;
            }
            if ((int)p->levels.size() <= level) p->levels.resize(level + 1);
            Level& L = p->levels[level];
            L.hop = hop_length >> level;
            std::vector<double> fr(freqs.begin() + lo, freqs.begin() + hi), lb(len_all.begin() + lo, len_all.begin() + hi);
            int nfft = 0;
            std::vector<std::complex<double>> kern;
            build_bank(fr, sample_rate / std::pow(2.0, level), sr_base, gamma, p->alpha, lb, nfft, kern);
            if ((int)L.banks.size() >= MAX_BANKS) {
                amtx_set_error("amtx_cqt_plan_create: more than %d banks on one pyramid level", MAX_BANKS);
                amtx_cqt_plan_destroy(p);
                return AMTX_ERR_UNSUPPORTED;
            }
            BankDev bd; bd.col0 = L.ncols; bd.nf = hi - lo; bd.harm = h; bd.bin0 = lo; bd.frames = 0;
            L.banks.push_back(bd);
            L.ncols += 2 * (hi - lo);
            L.nfft = std::max(L.nfft, nfft);
            L.raw.emplace_back(nfft, std::move(kern));
        }
    }
    // stack the banks of every level (kernels centred in the level's n_fft window), pack hi/lo planes, upload
    for (auto& L : p->levels) {
        if (L.banks.empty()) continue;
        p->pad = std::max(p->pad, L.nfft / 2);
        std::vector<float> W((size_t)L.ncols * L.nfft, 0.f);
        for (size_t bi = 0; bi < L.banks.size(); ++bi) {
            const BankDev& bd = L.banks[bi];
            const int nfft = L.raw[bi].first, off = (L.nfft - nfft) / 2;
            for (int k = 0; k < bd.nf; ++k)
                for (int n = 0; n < nfft; ++n) {
                    const std::complex<double> v = L.raw[bi].second[(size_t)k * nfft + n];
                    // (re, im) of a filter in neighbouring columns: the GEMM's magnitude epilogue combines them in registers
                    W[(size_t)(bd.col0 + 2 * k) * L.nfft + off + n] = (float)v.real();
                    W[(size_t)(bd.col0 + 2 * k + 1) * L.nfft + off + n] = (float)v.imag();
                }
        }
        L.raw.clear();
        amtx_gemm_pack_dims(L.ncols, L.nfft, &L.n_pad, &L.k_pad);
        std::vector<bf16_t> packed((size_t)L.n_pad * L.k_pad * 2);
        amtx_gemm_pack_host(W.data(), L.nfft, L.ncols, L.nfft, 2, packed.data());
        hipError_t e = hipMalloc(&L.d_w, packed.size() * 2);
        if (e == hipSuccess) e = hipMemcpy(L.d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            std::vector<int2> map((size_t)L.ncols / 2);
            for (const BankDev& bd : L.banks)
                for (int k = 0; k < bd.nf; ++k) map[(size_t)bd.col0 / 2 + k] = make_int2(bd.harm * n_bins + bd.bin0 + k, bd.harm);
            e = hipMalloc(&L.d_map, map.size() * sizeof(int2));
            if (e == hipSuccess) e = hipMemcpy(L.d_map, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            amtx_set_error("amtx_cqt_plan_create: weight upload failed: %s", hipGetErrorString(e));
            amtx_cqt_plan_destroy(p);
            return AMTX_ERR_HIP;
        }
    }
    p->pad = (p->pad + 3) & ~3;
    // decimator taps
    std::vector<float> taps(DEC_TAPS, 0.0f);
    {
        std::vector<double> h(DEC_TAPS);
        double sum = 0;
        for (int i = 0; i < DEC_TAPS; ++i) {
            const double n = i - DEC_HALF, x = 2.0 * DEC_CUTOFF * n;
            const double sinc = (n == 0) ? 1.0 : std::sin(PI * x) / (PI * x);
            const double r = 2.0 * i / (DEC_TAPS - 1) - 1.0;
            h[i] = 2.0 * DEC_CUTOFF * sinc * bessel_i0(DEC_BETA * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(DEC_BETA);
            sum += h[i];
        }
        for (int i = 0; i < DEC_TAPS; ++i) taps[i] = (float)(h[i] / sum);
    }
    std::vector<bf16_t> tfrag((size_t)DEC_NKS * 3 * 64 * 8);
    for (int ks = 0; ks < DEC_NKS; ++ks)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int i = l & 15, k = 32 * ks + 8 * (l >> 4) + j, tap = k - 2 * i;
                const float v = (tap >= 0 && tap < DEC_TAPS) ? taps[tap] : 0.0f;
                const bf16_t hi = f32_to_bf16_rn(v);
                const float r1 = v - bf16_to_f32(hi);
                const bf16_t mid = f32_to_bf16_rn(r1);
                tfrag[((size_t)(ks * 3 + 0) * 64 + l) * 8 + j] = hi;
                tfrag[((size_t)(ks * 3 + 1) * 64 + l) * 8 + j] = mid;
                tfrag[((size_t)(ks * 3 + 2) * 64 + l) * 8 + j] = f32_to_bf16_rn(r1 - bf16_to_f32(mid));
            }
    hipError_t e = hipMalloc(&p->d_tfrag, tfrag.size() * sizeof(bf16_t));
    if (e == hipSuccess) e = hipMemcpy(p->d_tfrag, tfrag.data(), tfrag.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        amtx_set_error("amtx_cqt_plan_create: device allocation failed: %s", hipGetErrorString(e));
        amtx_cqt_plan_destroy(p);
        return AMTX_ERR_HIP;
    }
    *out = p;
    return AMTX_OK;
}

namespace {

struct CqtDims {
    std::vector<int64_t> len, stride, frames;     // per level
    std::vector<int> frames_h;                    // per harmonic: librosa's own output length
    int64_t t_buf = 0, t_out = 0;
    size_t off_pyr = 0, off_r = 0, off_mag = 0, off_max = 0, total = 0;
    std::vector<size_t> pyr_off, r_off;
};

// VQT.get_expected_frames (features/vqt.py:102-134) with the reference's own early-downsample estimate (librosa-0.9 Q)
int64_t reference_expected_frames(const amtx_cqt_plan* p, double f0, int64_t n) {
    const double fmax = f0 * std::pow(2.0, (double)(p->n_bins - 1) / p->bpo);
    const double cQ = 1.0 / (std::pow(2.0, 1.0 / p->bpo) - 1.0);
    const double cutoff = fmax * (1 + 0.5 * HANN_BW / cQ) + 0.5 * p->gamma;
    const int n_oct = (p->n_bins + p->bpo - 1) / p->bpo;
    const int c = early_count(p->sr / 2.0, cutoff, p->hop, n_oct);
    int64_t best = -1;
    for (int k = c; k < c + n_oct; ++k) {
        const int64_t sig = (int64_t)std::ceil((double)n / std::pow(2.0, k));
        const int64_t hop = p->hop >> k;
        if (hop <= 0) continue;
        const int64_t f = sig / hop + 1;
        if (best < 0 || f < best) best = f;
    }
    return best;
}

CqtDims dims(const amtx_cqt_plan* p, int B, int64_t n) {
    CqtDims d;
    const int nl = (int)p->levels.size();
    d.len.resize(nl); d.stride.resize(nl); d.frames.resize(nl); d.pyr_off.resize(nl); d.r_off.resize(nl);
    int64_t len = n;
    size_t off = 0;
    for (int l = 0; l < nl; ++l) {
        if (l > 0) len = (len + 1) / 2;
        d.len[l] = len;
        d.stride[l] = ((2 * (int64_t)p->pad + len + 3) / 4) * 4 + 4;
        d.frames[l] = p->levels[l].hop > 0 ? 1 + len / p->levels[l].hop : 0;
        d.pyr_off[l] = off;
        off += ((size_t)B * d.stride[l] * 4 + 255) & ~(size_t)255;
    }
    d.off_r = off;          // (the complex responses used to live here; the GEMM's magnitude epilogue writes the map directly)
    for (int l = 0; l < nl; ++l) d.r_off[l] = off;
    const int n_oct = (p->n_bins + p->bpo - 1) / p->bpo;
    d.frames_h.resize(p->n_harm);
    int64_t tmin = -1, tref = -1;
    for (int h = 0; h < p->n_harm; ++h) {
        int64_t t = -1;
        for (int j = 0; j < n_oct; ++j) {
            const int l = p->early[h] + j;
            if (l < nl && !p->levels[l].banks.empty()) t = (t < 0) ? d.frames[l] : std::min(t, d.frames[l]);
        }
        d.frames_h[h] = (int)t;
        d.t_buf = std::max<int64_t>(d.t_buf, t);
        tmin = (tmin < 0) ? t : std::min(tmin, t);
        const int64_t r = reference_expected_frames(p, p->harmonics[h] * p->fmin, n);
        tref = (tref < 0) ? r : std::min(tref, r);
    }
    d.t_out = p->truncate ? std::min(tmin, tref) : tmin;
    d.off_mag = off;
    off += ((size_t)B * p->n_harm * p->n_bins * d.t_buf * 4 + 255) & ~(size_t)255;
    d.off_max = off;
    off += ((size_t)B * p->n_harm * 4 + 255) & ~(size_t)255;
    d.total = off;
    return d;
}

}  // namespace

extern "C" int amtx_cqt_num_harmonics(const amtx_cqt_plan* p) { return p ? p->n_harm : AMTX_ERR_ARG; }

extern "C" int64_t amtx_cqt_num_frames(const amtx_cqt_plan* p, int64_t num_samples) {
    if (!p || num_samples <= 0) return p ? 0 : AMTX_ERR_ARG;
    return dims(p, 1, num_samples).t_out;
}

extern "C" size_t amtx_cqt_workspace_bytes(const amtx_cqt_plan* p, int batch, int64_t num_samples) {
    if (!p || batch <= 0 || num_samples <= 0) return 0;
    return dims(p, batch, num_samples).total;
}

// out16 != null: the map as [B][T][n_bins][8] bf16 (amtx_cqt_forward16) instead of `out`
static int cqt_forward_impl(const amtx_cqt_plan* p, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                            void* workspace, size_t workspace_bytes, float* out, void* out16, void* stream_, int64_t lo_plane = 0) {
    AMTX_REQUIRE(p && audio && workspace && (out || out16), "amtx_cqt_forward: null pointer");
    AMTX_REQUIRE(batch > 0 && batch < 65536 && num_samples > 0 && audio_stride >= num_samples, "amtx_cqt_forward: bad batch/num_samples");
    AMTX_REQUIRE(((uintptr_t)workspace % 256) == 0, "amtx_cqt_forward: workspace must be 256-byte aligned");
    const CqtDims d = dims(p, batch, num_samples);
    AMTX_REQUIRE(workspace_bytes >= d.total, "amtx_cqt_forward: workspace too small (%zu < %zu)", workspace_bytes, d.total);
    AMTX_REQUIRE(d.t_out > 0, "amtx_cqt_forward: clip too short");
    if (p->lib09) AMTX_REQUIRE(d.len.back() > p->pad, "amtx_cqt_forward: reflect padding needs a longer clip");
    hipStream_t s = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const int B = batch, nl = (int)p->levels.size();
    float* mag = (float*)(ws + d.off_mag);
    float* maxbuf = (float*)(ws + d.off_max);
    const int zero_pads = p->lib09 ? 0 : 1;

    // librosa >= 0.10 (zero centre padding): level 0 of the pyramid IS the caller's audio -- the first decimation and the level-0 basis
    // product read it where it lies (out-of-clip samples read as zero: the decimator clamps, the GEMM's A loader takes a valid range), so
    // the clip is not copied between two paddings first (0.45 ms and 1.3 GB per 512 clips).  librosa 0.9's reflecting pad keeps the copy.
    static const bool copy_level0 = getenv("AMTX_CQT_COPY_LEVEL0") != nullptr;        // A/B switch
    const bool direct0 = zero_pads && !copy_level0;
    if (direct0 && nl == 1) AMTX_CHECK_HIP(hipMemsetAsync(maxbuf, 0, sizeof(float) * B * p->n_harm, s));
    for (int l = 0; l < nl; ++l) {
        float* pyr = (float*)(ws + d.pyr_off[l]);
        if (l == 0) {
            if (direct0) continue;
            const unsigned nb = (unsigned)std::min<int64_t>((num_samples + 255) / 256, 4096);
            hipLaunchKernelGGL(cqt_level0_kernel, dim3(nb, B), dim3(256), 0, s, audio, num_samples, audio_stride, pyr, d.stride[0], p->pad,
                               zero_pads, maxbuf, p->n_harm);
        } else {
            const bool from_audio = direct0 && l == 1;
            int rc = amtx_launch_cqt_decimate(from_audio ? audio : (const float*)(ws + d.pyr_off[l - 1]), d.len[l - 1], from_audio ? audio_stride : d.stride[l - 1],
                                              from_audio ? 0 : p->pad, pyr, d.len[l], d.stride[l], p->pad, p->d_tfrag, zero_pads,
                                              from_audio ? maxbuf : (float*)nullptr, p->n_harm, B, s);
            if (rc != AMTX_OK) return rc;
        }
        AMTX_CHECK_LAUNCH();
        if (!zero_pads) {      // librosa 0.9: reflecting centre pad, from the level's own samples
            hipLaunchKernelGGL(cqt_pad_kernel, dim3(8, B), dim3(256), 0, s, pyr, d.len[l], d.stride[l], p->pad, p->lib09);
            AMTX_CHECK_LAUNCH();
        }
    }
    // the basis products of all levels: the windowed kernel (cqt_basis_kernel), one launch per level on the same stream; AMTX_CQT_GEMM_BASIS=1
    // keeps round 4's path (all levels in ONE launch of the generic fp32-A two-plane GEMM) for the A/B
    static const bool gemm_basis = getenv("AMTX_CQT_GEMM_BASIS") != nullptr;
    bool windowed = !gemm_basis && nl <= BAS_MAXLEV && p->n_harm <= 16 && (int64_t)p->n_harm * p->n_bins * d.t_buf < (1ll << 31);
    for (int l = 0; l < nl && windowed; ++l) {
        const Level& L = p->levels[l];
        if (L.banks.empty()) continue;
        windowed = (L.nfft == 128 || L.nfft == 256) && L.k_pad >= L.nfft && L.ncols <= 256 && L.ncols % 2 == 0 && L.hop >= 4 && (L.hop & (L.hop - 1)) == 0 &&
                   (L.hop <= 512) && d.len[l] >= 4;
    }
    if (windowed) {
        BasisArgs ba;
        ba.nlev = nl; ba.B = B; ba.ntiles = 0;
        ba.mag = mag; ba.mag_gs = (int64_t)p->n_harm * p->n_bins * d.t_buf; ba.mag_pitch = d.t_buf;
        ba.maxbuf = maxbuf; ba.n_harm = p->n_harm;
        for (int h = 0; h < 16; ++h) ba.rows[h] = h < p->n_harm ? d.frames_h[h] : 0;
        for (int l = 0; l < nl; ++l) {
            const Level& L = p->levels[l];
            BasisLevel& bl = ba.lev[l];
            bl = BasisLevel();
            if (L.banks.empty()) { bl.frames = 0; continue; }
            if (direct0 && l == 0) { bl.sig = audio; bl.sig_gs = audio_stride; }
            else { bl.sig = (const float*)(ws + d.pyr_off[l]) + p->pad; bl.sig_gs = d.stride[l]; }
            bl.len = d.len[l];
            // librosa 0.9's reflecting pads live in the pyramid buffer on both sides of the signal: there every sample the windows touch exists
            bl.off = 0;
            if (!zero_pads) { bl.sig -= p->pad; bl.off = p->pad; bl.len = d.stride[l]; }
            bl.w = L.d_w; bl.map = L.d_map; bl.n_pad = L.n_pad; bl.k_pad = L.k_pad; bl.ncols = L.ncols; bl.hop = L.hop;
            bl.frames = (int)d.frames[l];
            const int hs0 = std::min(L.hop, L.nfft);
            int ftl = std::max(16, std::min(128, ((BAS_MAXWIN - L.nfft) / hs0 + 1) / 16 * 16));
            ftl = std::min(ftl, (bl.frames + 15) / 16 * 16);
            bl.ft = ftl;
            bl.tiles_per_clip = (bl.frames + ftl - 1) / ftl;
        }
        for (int kk = 128; kk <= 256; kk *= 2) {
            BasisLevs ls;
            ls.n = 0;
            size_t lds = 0;
            for (int l = 0; l < nl; ++l) {
                const BasisLevel& bl = ba.lev[l];
                if (bl.frames <= 0 || p->levels[l].nfft != kk) continue;
                const int hs = std::min(bl.hop, kk);               // frame stride of the staged copy (cqt_basis_kernel)
                const int ws_s = (bl.ft - 1) * hs + kk;
                const int wsp = ws_s + bas_pad(hs) * ((ws_s + hs - 1) / hs);
                lds = std::max(lds, (size_t)((wsp + 7) & ~7) * 2 * 2 + 64);
                ls.lev[ls.n++] = l;
            }
            if (!ls.n) continue;
            // two 256-thread blocks per CU; the levels (and the second column slice of the wide ones) share the chip: 512 blocks dealt EVENLY to the
            // levels (dealing them by staged samples, by matrix work or by a mix was measured slower, 2.2 - 2.6 against 1.97 ms for the
            // front-end: the deep levels' tiles are as many as the shallow ones' and a tile's fixed costs -- barriers, a load round trip, the
            // maxima's atomics -- weigh as much as its size)
            int slices = 1;
            for (int i = 0; i < ls.n; ++i) slices = std::max(slices, (ba.lev[ls.lev[i]].ncols + 127) / 128);
            ls.b0[0] = 0;
            for (int i = 0; i < ls.n; ++i) {
                const int64_t tiles = (int64_t)ba.lev[ls.lev[i]].tiles_per_clip * B;
                ls.b0[i + 1] = ls.b0[i] + (int)std::min<int64_t>(std::max(8, 512 / ls.n), tiles);
            }
            const unsigned gx = (unsigned)ls.b0[ls.n];
            if (kk == 256) {
                AMTX_GRANT_LDS(cqt_basis_kernel<8>, lds);
                hipLaunchKernelGGL(cqt_basis_kernel<8>, dim3(gx, slices), dim3(256), lds, s, ba, ls);
            } else {
                AMTX_GRANT_LDS(cqt_basis_kernel<4>, lds);
                hipLaunchKernelGGL(cqt_basis_kernel<4>, dim3(gx, slices), dim3(256), lds, s, ba, ls);
            }
            AMTX_CHECK_LAUNCH();
        }
    } else {
    // the basis products of ALL levels in one launch (same kernel, per-level A / W / sizes; 8 launches before)
    GemmArgs gs[AMTX_GEMM_MULTI_MAX];
    int ng = 0;
    for (int l = 0; l < nl; ++l) {
        const Level& L = p->levels[l];
        if (L.banks.empty()) continue;
        // the basis product with the power epilogue: |re + i im|^2 of every filter goes straight into mag[b][harmonic][bin][t]
        // (transposed, truncated to the harmonic's frame count); the complex response never touches HBM (it was written by the GEMM
        // and read back by a magnitude / transpose kernel: 0.4 GB per level and 512 clips, a third of the front-end's traffic)
        GemmArgs g;
        g.A = (const float*)(ws + d.pyr_off[l]) + (p->pad - L.nfft / 2); g.lda = L.hop; g.a_type = AMTX_T_F32;
        if (direct0 && l == 0) {   // rows start n_fft / 2 before the clip and run past its end: bounded reads straight from the audio
            // a pointer n_fft / 2 elements BEFORE the caller's allocation: nothing below a_valid_lo is ever dereferenced (AStage::load), and
            // the range starts exactly where the clip does
            g.A = audio - L.nfft / 2;
            g.a_valid_lo = L.nfft / 2; g.a_valid_hi = (int64_t)L.nfft / 2 + num_samples;
            AMTX_REQUIRE(g.a_valid_lo >= L.nfft / 2, "amtx_cqt_forward: internal: level-0 rows would reach below the clip");
        }
        g.W = L.d_w; g.n_pad = L.n_pad; g.k_pad = L.k_pad; g.planes = 2; g.bias = nullptr;
        g.C = nullptr; g.ldc = L.ncols; g.c_type = AMTX_T_F32;
        g.M = d.frames[l]; g.N = L.ncols; g.K = L.nfft;
        g.groups = B; g.a_gs = (direct0 && l == 0) ? audio_stride : d.stride[l]; g.w_gs = 0; g.bias_gs = 0; g.c_gs = 0;
        g.pair_map = L.d_map; g.pair_out = mag; g.pair_gs = (int64_t)p->n_harm * p->n_bins * d.t_buf; g.pair_pitch = d.t_buf;
        for (int h = 0; h < p->n_harm && h < 16; ++h) g.pair_rows[h] = d.frames_h[h];
        g.pair_max = maxbuf; g.pair_nh = p->n_harm;       // the per-(clip, harmonic) maxima of the dB reference, kept by the epilogue
        if (ng == AMTX_GEMM_MULTI_MAX) {
            int rc = amtx_launch_gemm_multi(gs, ng, s);
            if (rc != AMTX_OK) return rc;
            ng = 0;
        }
        gs[ng++] = g;
    }
    if (ng) {
        int rc = amtx_launch_gemm_multi(gs, ng, s);
        if (rc != AMTX_OK) return rc;
    }
    }
    if (out16) {
        AMTX_REQUIRE(p->n_harm <= 8 && ((uintptr_t)out16 % 16) == 0, "amtx_cqt_forward16: at most 8 harmonics, output 16-byte aligned");
        const int nbc = std::min(p->n_bins, 120);            // bins per pass: 32 x 121 x 16 bytes of LDS
        const size_t lds = (size_t)(lo_plane ? 2 : 1) * SC16_TT * (nbc + 1) * 16 + 64;
        AMTX_GRANT_LDS(cqt_scale16_kernel, lds);
        hipLaunchKernelGGL(cqt_scale16_kernel, dim3((unsigned)((d.t_out + SC16_TT - 1) / SC16_TT), B), dim3(256), lds, s, (const float*)mag,
                           (const float*)maxbuf, p->n_harm, p->n_bins, nbc, d.t_buf, d.t_out, decibels, (uint4*)out16, lo_plane);
        AMTX_CHECK_LAUNCH();
        return AMTX_OK;
    }
    hipLaunchKernelGGL(cqt_scale_kernel, dim3(8, B * p->n_harm), dim3(256), 0, s, (const float*)mag, (const float*)maxbuf, p->n_bins, d.t_buf,
                       d.t_out, decibels, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

extern "C" int amtx_cqt_forward(const amtx_cqt_plan* p, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                                void* workspace, size_t workspace_bytes, float* out, void* stream_) {
    AMTX_REQUIRE(out, "amtx_cqt_forward: null pointer");
    return cqt_forward_impl(p, audio, num_samples, audio_stride, batch, decibels, workspace, workspace_bytes, out, nullptr, stream_);
}

extern "C" int amtx_cqt_forward16(const amtx_cqt_plan* p, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                                  void* workspace, size_t workspace_bytes, void* out16, void* stream_) {
    AMTX_REQUIRE(out16, "amtx_cqt_forward16: null pointer");
    return cqt_forward_impl(p, audio, num_samples, audio_stride, batch, decibels, workspace, workspace_bytes, nullptr, out16, stream_);
}

// The same map as TWO 16-bit planes for the two-plane (x3) engine: hi = bf16(v) at out16 (exactly amtx_cqt_forward16's map), lo = bf16(v - hi)
// `plane_elems` 16-bit elements behind it (>= B T n_bins 8, a multiple of 8) -- split_bf16x2 of the fp32 feature amtx_cqt_forward writes.
extern "C" int amtx_cqt_forward16_split(const amtx_cqt_plan* p, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                                        void* workspace, size_t workspace_bytes, void* out16, int64_t plane_elems, void* stream_) {
    AMTX_REQUIRE(out16 && plane_elems > 0 && plane_elems % 8 == 0, "amtx_cqt_forward16_split: null pointer / plane stride not a multiple of 8 elements");
    return cqt_forward_impl(p, audio, num_samples, audio_stride, batch, decibels, workspace, workspace_bytes, nullptr, out16, stream_, plane_elems / 8);
}
