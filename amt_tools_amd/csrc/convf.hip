// The whole convolution stack of the Onsets & Frames acoustic model in ONE kernel for gfx950
// (amt_tools/models/onsetsframes.py:375-412: layer1 Conv(1->32)+BN+ReLU, layer2 Conv(32->32)+BN+ReLU+MaxPool(1,2),
// layer3 Conv(32->64)+BN+ReLU+MaxPool(1,2); Dropout is the identity in eval mode), bf16 operands, fp32 accumulation.
//
// Why: as two kernels (conv.hip) the 32-channel map behind layer2 -- 14.6 kB per frame, 9.3 GB per 1024 clips -- is written to
// HBM by one kernel and read back by the next: 37 % of all bytes a forward pass moves, and the kernels run power-managed at
// 1.8 GHz under that traffic.  Here neither intermediate map exists outside LDS.
//
// Shape of the kernel.  One 512-thread block per CU, wave-specialised: waves 0-3 ("producers") run layer2, waves 4-7 ("consumers")
// run layer3, so that each SIMD hosts one wave of either kind and their (equal) matrix work shares the SIMD's matrix pipe; every wave
// keeps ITS layer's folded weights stationary in registers (producer 72 + layer1's 32, consumer 144 VGPRs), which is why the split is
// by layer: one wave cannot hold both sets.
// A block owns "strips" = (head, clip, 62 or 60 consecutive frames: HALO below) and STREAMS each strip along the frequency axis in steps of 8
// layer2 columns (= 4 pooled columns = 2 output columns), ONE barrier per step:
//     step k:   producers   layer2 of step k (a1 ring, 66 rows x 18 columns x 32 ch) -> a2 ring (64 rows x 10 pooled columns),
//                           then the feature staging of step k + 2 and their layer1 unit(s) of step k + 1
//               consumers   their layer1 unit of step k + 1, then layer3 on the pooled columns of step k - 1 -> 64-byte stores
//               barrier
// layer1 is a latency chain (LDS read -> 8 small MFMAs -> convert / clamp -> LDS write) with 3 % of the flops.  It runs a step AHEAD,
// dealt in 16-row x 4-column units to all eight waves (the consumers read its Toeplitz fragments from LDS; the ninth unit, rows 64 - 65,
// goes to a producer wave in turn), and each role does its vector work while the OTHER role's wave on the same SIMD is in its matrix loop
// (as a phase of its own, behind a barrier, layer1 was a third of the step with the matrix pipe idle).
// Rings are indexed by a running column counter (mod 18 / mod 10), so the two columns a 3x3 window needs from the previous step are
// simply still there, and a strip's last step flows into the next strip's first one without a drain.  Streaming along frequency means
// there is NO halo in frequency at all (the zero padding at both ends is the real padding); in time a strip computes 64 layer2
// rows for 62 layer3 rows (+3 %) and 66 layer1 rows.  MFMA N-tile = 16 frames at one frequency column, "swapped" product
// D = W . X^T exactly as in conv.hip (a lane ends up with consecutive channels of one position: MaxPool over the frequency pair,
// ReLU and the channels-last store are lane-local); the accumulation order of every output equals conv.hip's, so the result is
// BIT-IDENTICAL to the two-kernel path (tests/test_gpu_model.py::test_fused_conv_stack_is_bit_identical_to_the_two_kernel_path).
// HISTORY.md ("Measured (round 3)") has the history of the structure, the ablation numbers and what was tried and rejected.
//
// LDS (137 KB of the CU's 160): a1 ring 4 chunk planes x 66 x 19 x 16 B, a2 ring 4 x 64 x 11 x 16 B (chunk-major: 16-byte chunk c of
// position (row, slot) at c * PLANE + (row * pitch + slot) * 16; both row pitches are odd, so the 16 rows of a ds_read_b128 lane group
// hit 16 distinct 16-byte bank groups, and the 8 rows of a ds_write_b128 lane group 8 distinct ones), two bf16 feature slabs (68 rows x
// 12 columns, dB-scaled while they are staged: amtx_of_forward_power) + scratch lines, layer1's Toeplitz fragments, the shift tables.

#include "amtx_f16_names.h"
#include "amtx_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int RT = 4;                 // 16-row tiles per strip
constexpr int R2 = 16 * RT;           // layer2 rows per strip (64): strip row i2 <-> frame r0 - 1 + i2
// HALO = true: 66 layer1 rows (a ninth layer1 unit for rows 64, 65) -> all 64 layer2 rows valid -> 62 output rows per strip.
// HALO = false: 64 layer1 rows in eight units -> layer2 rows 62, 63 are garbage (they read the two never-written, zeroed ring rows) and are
// never used -> 60 output rows per strip.  The ninth unit is two rows of work but a whole unit's latency chain (~1000 cycles) on ONE wave
// of every step, and the step's barrier waits for that wave: every wave stood ~700 cycles at the barrier.  Without it a 625-frame clip is
// still 11 strips (11 x 60 = 660), i.e. the same matrix work.
// Both variants are compiled (template parameter of the kernel); the launcher takes HALO = true only where the two extra rows save
// a strip per clip (ceil(T / 62) < ceil(T / 60)).  The LDS layout is the one of HALO = true for both.
constexpr int r3_of(bool halo) { return halo ? R2 - 2 : R2 - 4; }   // layer3 (output) rows per strip: o <-> frame r0 + o
constexpr int r1_of(bool halo) { return halo ? R2 + 2 : R2; }       // layer1 rows computed: i1 <-> frame r0 - 2 + i1
constexpr int R1A = R2 + 2;                  // layer1 rows the ring holds (layer2's fragment reads reach row 65)
constexpr int RFA = R1A + 2;                 // feature rows a slab holds (68): fi <-> frame r0 - 3 + fi
constexpr int CS = 8;                 // layer2 columns per step
constexpr int CP = CS / 2;            // pooled columns per step (one per producer wave)
constexpr int RC1 = 2 * CS + 2;       // a1 ring columns: this step's 8 + 2 carried + the next step's 8 (layer1 runs a step ahead)
constexpr int RC2 = 2 * CP + 2;       // a2 ring columns: the step being consumed (4 + 2 carried) + the step being produced (4)
constexpr int PITCH1 = RC1 + 1;       // 19 slots per a1 ring row (odd)
constexpr int PITCH2 = RC2 + 1;       // 11 (odd)
constexpr int ROWB1 = PITCH1 * 16, ROWB2 = PITCH2 * 16;
constexpr int PLANE1 = (R1A * ROWB1 + 255) / 256 * 256;
constexpr int PLANE2 = (R2 * ROWB2 + 255) / 256 * 256;
constexpr int A1_OFF = 0;
constexpr int A2_OFF = 4 * PLANE1;
constexpr int SLAB_COLS = CS + 4;     // 12: feature columns 8 j - 2 .. 8 j + 9 of step j (the last two only ever meet zero weights)
constexpr int SLAB_LOAD = CS + 2;     // 10 of them are loaded
constexpr int SLABP = SLAB_COLS * 2;  // 24 bytes per slab row = 8 x odd: the 17 rows of a ds_read_b64 lane group on distinct banks
constexpr int SLAB_BYTES = (RFA * SLABP + 15) / 16 * 16;
constexpr int SLAB_OFF = A2_OFF + 4 * PLANE2;
constexpr int SH3_OFF = SLAB_OFF + 3 * SLAB_BYTES + 128 + 1024;   // + slack: the halo unit's discarded lanes read past their slab; staging's scratch slots
constexpr int W1_OFF = SH3_OFF + 64 * 4;                   // layer1's Toeplitz fragments (8 x 1 KiB) + shift for the waves that have no registers for them
constexpr int SH1_OFF = W1_OFF + 8 * 1024;
constexpr int LDS_BYTES = SH1_OFF + 32 * 4;
constexpr int FPRE = (RFA * SLAB_LOAD + 255) / 256;         // feature values per producer thread and step (3: 660 or 680 values)
constexpr int XB = CS / 4;                                 // 4-column blocks per step (2)
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(CP == 4 && XB == 2, "wave roles below are written for 8 layer2 columns per step");

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
    return amtx_mfma_16x16x32(a, b, c);
}
__device__ __forceinline__ void settle(const uint4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void settle(float v) { asm volatile("" ::"v"(v)); }

#ifdef AMTX_CONVF_TIMING
// Debug build only (AMTX_EXTRA_FLAGS=-DAMTX_CONVF_TIMING): cycles wave 0 (producer) and wave 4 (consumer) of every block spend per
// phase, summed over blocks: [0] layer2, [1] slab write + load issue + layer1 unit, [2] wait at the barrier, [3] steps;
// [8] layer1 unit(s), [9] layer3, [10] wait at the barrier, [11] steps.  Read with amtxdbg_convf_prof().  AMTX_CONVF_DBG (bit mask, read per launch) skips
// work for timing experiments (the output is garbage then): 1 layer1, 2 layer2, 4 layer3, 8 feature staging.
__device__ unsigned long long g_convf_prof[16];
#define CONVF_TICK(SLOT)                                                   \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        prof_acc[SLOT] += now_ - prof_t;                                   \
        prof_t = now_;                                                     \
    } while (0)
#else
#define CONVF_TICK(SLOT) do {} while (0)
#endif
#if defined(AMTX_CONVF_TIMING) || defined(AMTX_CONVF_ABLATE)
#define CONVF_DBG(BIT) ((dbg & (BIT)) != 0)
#else
#define CONVF_DBG(BIT) false
#endif

struct ConvFArgs {
    const float* feats; int64_t f_stride_b, f_stride_t, f_stride_f;   // (B, T, F) view of the one-channel features (or raw power)
    const float* f_clip_max; const float* f_ref;                      // raw power: dB-scaled while staged (null: features as is)
    const bf16_t* w1frag; int64_t w1_gs; const float* shift1;         // amtx_conv1_pack_host(c_in = 1), [groups][32]
    const bf16_t* w2frag; int64_t w2_gs; const float* shift2;         // amtx_conv3x3_pack_host(32), [groups][32]
    const bf16_t* w3frag; int64_t w3_gs; const float* shift3;         // amtx_conv3x3_pack_host(64), [groups][64]
    bf16_t* out; int64_t out_gs;                                      // [groups][B][T][F / 4][64]
    int64_t out_plane;                                                // != 0: [groups][F / 4][B T][64], out_plane = B T 64 (GemmArgs::a_plane)
    int B, T, F;
};

// position of one of the block's streams (feature loads, layer1, layer2, layer3) in its sequence of steps: all scalar
struct Pos { int kk, j, b, r0; };
template <int R3_>
__device__ __forceinline__ Pos pos_first_t(int first, int ntt) {
    Pos p;
    p.kk = 0; p.j = 0; p.b = first / ntt; p.r0 = (first - p.b * ntt) * R3_;
    return p;
}
template <int R3_>
__device__ __forceinline__ void pos_next_t(Pos& p, int nstep, int ntt) {
    ++p.kk;
    if (++p.j == nstep) {
        p.j = 0;
        p.r0 += R3_;
        if (p.r0 >= ntt * R3_) { p.r0 = 0; ++p.b; }
    }
}

template <bool HALO>
__global__ __launch_bounds__(512, 2) void convf_kernel(ConvFArgs a, int nstep, int ntt, int nstrips, int per_block, int dbg) {
    constexpr int R3 = r3_of(HALO), R1 = r1_of(HALO), RF = R1 + 2;
    constexpr int FITEMS = RF * SLAB_LOAD;                     // feature values per step (660 / 680)
    constexpr int NUNITS = RT * XB + (HALO ? 1 : 0);           // layer1 units per step: 16 rows x 4 columns each (+ one unit for rows 64, 65)
    auto pos_first = [](int first_, int ntt_) { return pos_first_t<R3>(first_, ntt_); };
    auto pos_next = [](Pos& p_, int nstep_, int ntt_) { pos_next_t<R3>(p_, nstep_, ntt_); };
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

    // ---------------------------------------------------------------- layer1 (Toeplitz product, see conv.hip), shared by both roles
    // A step's a1 columns are 9 units: unit u < 8 = rows 16 rt + n16, columns 8 j - 1 + 4 xb + q (rt = u / 2, xb = u % 2) -> wave u;
    // unit 8 = rows 64 + (n16 & 1), column block n16 >> 1 (lanes with block >= 2 compute garbage that is not stored) -> a consumer wave
    // in turn.  One unit per wave: as a phase of the producer waves alone (three units in a row on wave 0) this latency chain was the
    // longest part of a step.
    // (the macros below read n16 = lane & 15, g1 = lane >> 4, gg = min(g1, 2): k-group 3 has zero weights and re-reads group 2's row,
    // lane_a1 = the lane part of an a1 address -- each role defines them where it can afford the registers)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    // one unit U of the step at P: CONVF_L1_BEGIN reads the unit's B fragment (8 consecutive feature values of row n + k-group) and
    // sets up the addresses; CONVF_L1_STORE(Q, A0, A1) finishes column Q of the unit's four from its two accumulators
#define CONVF_L1_BEGIN(P, U)                                                                                \
        const char* slab = smem + SLAB_OFF + ((P).kk & 1) * SLAB_BYTES;                                     \
        const int base1 = ((P).kk * CS) % RC1;      /* ring slot of the step's first new a1 column (8 j - 1) */ \
        const int u = (U);                                                                                  \
        const bool mainu = !HALO || u < NUNITS - 1;             /* scalar */                                \
        const int row1 = mainu ? 16 * (u >> 1) + n16 : R2 + (n16 & 1);                                      \
        const int xb = mainu ? (u & 1) : (n16 >> 1);                                                        \
        const char* fp = slab + (row1 + gg) * SLABP + 8 * xb;                                               \
        const uint2 b0 = *reinterpret_cast<const uint2*>(fp);                                               \
        const uint2 b1 = *reinterpret_cast<const uint2*>(fp + 8);                                           \
        const uint4 bh = make_uint4(b0.x, b0.y, b1.x, b1.y);                                                \
        /* straight-line epilogue: the zero padding of the map is a mask, the ring slot a per-lane value, and lanes of the last unit */ \
        /* that have no position (column block >= 2) store to a scratch line: no branch, no exec mask */    \
        const uint32_t rowmask = ((unsigned)((P).r0 - 2 + row1) < (unsigned)a.T && xb < XB) ? 0xffffffffu : 0u; \
        const int c1l = CS * (P).j - 1 + 4 * xb;                /* first a1 column of this lane's four positions */ \
        const int slotl = base1 + 4 * xb;                                                                   \
        const int dst_row = A1_OFF + g1 * PLANE1 + row1 * ROWB1;                                            \
        const int dst_scratch = SLAB_OFF + 2 * SLAB_BYTES + 128 + (lane_o & 63) * 16;                       \
        /* scalar: no position of the unit is padding -> the lean epilogue (a unit is ~50 instead of ~140 vector instructions, and */ \
        /* a wave's instruction stream, not the matrix pipe, is what a step of this kernel waits for) */    \
        const int tu_ = (P).r0 - 2 + 16 * (u >> 1), cu_ = CS * (P).j - 1 + 4 * (u & 1);                     \
        const bool lean = mainu && tu_ >= 0 && tu_ + 15 < a.T && cu_ >= 0 && cu_ + 3 < a.F;                 \
        const int slots_ = base1 + 4 * (u & 1);                 /* scalar ring slot of a main unit's first column */
#define CONVF_L1_STORE_LEAN(Q, ACC0, ACC1)                                                                  \
        do {                                                                                                \
            uint32_t pk[4];                                                                                 \
            pk[0] = pack_bf16x2((ACC0)[0], (ACC0)[1]); pk[1] = pack_bf16x2((ACC0)[2], (ACC0)[3]);           \
            pk[2] = pack_bf16x2((ACC1)[0], (ACC1)[1]); pk[3] = pack_bf16x2((ACC1)[2], (ACC1)[3]);           \
            _Pragma("unroll") for (int h = 0; h < 4; ++h)                                                   \
                pk[h] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[h]), (s16x2){0, 0})); \
            int sq = slots_ + (Q);                                                                          \
            sq = sq >= RC1 ? sq - RC1 : sq;                                                                 \
            *reinterpret_cast<uint4*>(smem + dst_row + sq * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);   \
        } while (0)
#define CONVF_L1_STORE(Q, ACC0, ACC1)                                                                       \
        do {                                                                                                \
            /* round first, then ReLU on the packed pairs as a signed 16-bit max with 0 (a negative bf16 is a negative int16), then the padding */ \
            uint32_t pk[4];                                                                                 \
            pk[0] = pack_bf16x2((ACC0)[0], (ACC0)[1]); pk[1] = pack_bf16x2((ACC0)[2], (ACC0)[3]);           \
            pk[2] = pack_bf16x2((ACC1)[0], (ACC1)[1]); pk[3] = pack_bf16x2((ACC1)[2], (ACC1)[3]);           \
            const uint32_t keep = (unsigned)(c1l + (Q)) < (unsigned)a.F ? rowmask : 0u;                     \
            _Pragma("unroll") for (int h = 0; h < 4; ++h)                                                   \
                pk[h] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[h]), (s16x2){0, 0})) & keep; \
            int slot = slotl + (Q);                             /* < 18 + 8 */                              \
            slot = slot >= RC1 ? slot - RC1 : slot;                                                         \
            const int dsto = xb < XB ? dst_row + slot * 16 : dst_scratch;                                   \
            *reinterpret_cast<uint4*>(smem + dsto) = make_uint4(pk[0], pk[1], pk[2], pk[3]);                \
        } while (0)

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
        // a1 ring rows no unit ever writes (HALO = false: rows 64, 65; layer2's last row tile reads them): zero once, all chunk planes
        if constexpr (!HALO) {
            constexpr int per = (R1A - R1) * ROWB1 / 16;
            for (int i = tid; i < 4 * per; i += 256)
                reinterpret_cast<uint4*>(smem + A1_OFF + (i / per) * PLANE1 + R1 * ROWB1)[i % per] = make_uint4(0, 0, 0, 0);
        }
        // the two slab columns no step ever stages (they meet zero weights, but NaN x 0 is NaN): zero once, both slabs
        for (int i = tid; i < 2 * RFA * 2; i += 256) {
            const int s = i / (RFA * 2), r = i % (RFA * 2);
            *reinterpret_cast<uint16_t*>(smem + SLAB_OFF + s * SLAB_BYTES + (r >> 1) * SLABP + (SLAB_LOAD + (r & 1)) * 2) = 0;
        }

        // feature staging: thread -> FPRE cells (row fi, column fc) of a step's 68 x 10 slab, fixed for the whole kernel.  Straight-line code:
        // loads are UNCONDITIONAL (a padding cell re-reads the clip's first value; hipcc waits for a guarded load at the join of its
        // branch: three guarded loads were three memory round trips in the producer's critical path) and so are the slab stores (a thread
        // without an n-th cell writes to a scratch slot); the validity bit is applied when the value is written to the slab two steps later.
        float fpre[FPRE];
        float fown = 0.f, fref = 0.f;
        unsigned fvalid = 0;                                       // bit n: fpre[n] is a real feature value (not padding)
        int fcell[FPRE], foff[FPRE], loff[FPRE];                   // fi << 8 | fc (-1 << 8: no cell), element offset inside the step's window, slab byte offset
#pragma unroll
        for (int n = 0; n < FPRE; ++n) {
            const int it = tid + 256 * n;
            const int fi = it / SLAB_LOAD, fc = it % SLAB_LOAD;
            fcell[n] = it < FITEMS ? (fi << 8 | fc) : -256;        // row -1: never inside the clip
            foff[n] = it < FITEMS ? fi * (int)a.f_stride_t + fc * (int)a.f_stride_f : 0;
            loff[n] = it < FITEMS ? fi * SLABP + fc * 2 : 2 * SLAB_BYTES + 64 + (tid & 31) * 2;   // scratch: the slack behind the slabs
        }
        const int fst = (int)a.f_stride_t, fsf = (int)a.f_stride_f;   // per-clip offsets fit 32 bits (checked at launch)
        // feature values of the step at P -> registers (written to a slab two steps later)
#define CONVF_ISSUE_LOADS(P)                                                                                \
        do {                                                                                                \
            const float* fb = a.feats + (int64_t)(P).b * a.f_stride_b;                                      \
            const int tb = (P).r0 - 3, cb = CS * (P).j - 2;                                                 \
            const int wb = tb * fst + cb * fsf;                 /* window origin; may be negative: only used when valid */ \
            fvalid = 0;                                                                                     \
            _Pragma("unroll") for (int n = 0; n < FPRE; ++n) {                                              \
                const bool ok = (unsigned)(tb + (fcell[n] >> 8)) < (unsigned)a.T && (unsigned)(cb + (fcell[n] & 255)) < (unsigned)a.F; \
                fvalid |= ok ? (1u << n) : 0u;                                                              \
                fpre[n] = fb[(unsigned)(ok ? wb + foff[n] : 0)];                                            \
            }                                                                                               \
            if (db) {   /* two independent loads (f_ref = f_clip_max when the caller gave none): a select on a just-loaded value waits for it */ \
                fown = a.f_clip_max[(P).b];                                                                 \
                fref = a.f_ref[(P).b];                                                                      \
            }                                                                                               \
        } while (0)
        // registers -> slab (KK & 1) as bf16, dB-scaled on the way when the input is raw power (db_scale_apply: the bits
        // amtx_spec_scale would have written, then the round-to-nearest-even conv.hip applies when it reads its fp32 tile)
#define CONVF_WRITE_SLAB(KK)                                                                                \
        do {                                                                                                \
            DbScale dbs = {0.f, 0.f};                                                                       \
            if (db) dbs = db_scale_make(fown, fref);                                                        \
            char* slab = smem + SLAB_OFF + ((KK) & 1) * SLAB_BYTES;                                         \
            _Pragma("unroll") for (int n = 0; n < FPRE; ++n) {                                              \
                float v = fpre[n];                                                                          \
                if (db) v = db_scale_apply(v, dbs);                                                         \
                v = (fvalid >> n) & 1u ? v : 0.f;                                                           \
                *reinterpret_cast<uint16_t*>(slab + loff[n]) = (uint16_t)pack_bf16x2(v, 0.f);               \
            }                                                                                               \
        } while (0)

        const int lane_a2 = A2_OFF + g * PLANE2 + trow * ROWB2;     // lane part of an a2 store address
        const int n16 = trow, g1 = g, gg = min(g, 2), lane_o = lane;
        const int lane_a1 = A1_OFF + g * PLANE1 + trow * ROWB1;     // lane part of an a1 address (fragment reads and layer1 stores)
#define CONVF_L1_PUNIT(P, U)                                                                                \
        do {                                                                                                \
            CONVF_L1_BEGIN(P, U)                                                                            \
            f32x4_t acc1[4][2];                                                                             \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) acc1[q][nt] = mfma16(w1t[q][nt], bh, sh1[nt]); \
            if (lean) {                                                                                     \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) CONVF_L1_STORE_LEAN(q, acc1[q][0], acc1[q][1]); \
            } else {                                                                                        \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) CONVF_L1_STORE(q, acc1[q][0], acc1[q][1]);    \
            }                                                                                               \
        } while (0)
        // a producer wave's units: u = wave (0 .. 3) of every step, and unit 8 when it is its turn (weights in registers: no LDS round trip)
#define CONVF_L1_PRODUCER(P)                                                                                \
        do {                                                                                                \
            CONVF_L1_PUNIT(P, wave);                                                                        \
            if (HALO && ((P).kk & 3) == wave) CONVF_L1_PUNIT(P, NUNITS - 1);                                \
        } while (0)

        Pos pl = pos_first(first, ntt), p1 = pl, p0 = pl;
        CONVF_ISSUE_LOADS(pl);
        CONVF_WRITE_SLAB(0);
        pos_next(pl, nstep, ntt);
        if (K > 1) {
            CONVF_ISSUE_LOADS(pl);
            CONVF_WRITE_SLAB(1);
            pos_next(pl, nstep, ntt);
        }
        if (K > 2) {
            CONVF_ISSUE_LOADS(pl);
            pos_next(pl, nstep, ntt);
        }
        __syncthreads();
        CONVF_L1_PRODUCER(p1);
        pos_next(p1, nstep, ntt);
        lds_only_barrier();

#ifdef AMTX_CONVF_TIMING
        unsigned long long prof_acc[4] = {0, 0, 0, 0};
        unsigned long long prof_t = __builtin_readcyclecounter();
#endif
        // static priority for the producer waves' matrix loop: their path through a step (layer2, staging, one or two layer1 units) is the
        // longer one, the consumer waves' layer3 takes the issue slots that are left (8.9 -> 8.6 ms per 1024 clips; the other way round: 9.3)
        __builtin_amdgcn_s_setprio(2);
        for (int kk = 0; kk <= K; ++kk) {
            if (kk < K) {
                // ---------------------------------------------------------------- layer2 + MaxPool(1,2): a1 ring -> a2 ring
                // this wave's pooled column p = 4 j - 1 + wave, its four a1 columns at ring offsets 2 wave - 2 + cc behind base1
                if (!CONVF_DBG(2)) {
                    const int base1 = (p0.kk * CS) % RC1;
                    const int base2 = (p0.kk * CP) % RC2;           // ring slot of this step's first pooled column (4 j - 1)
                    const int p = CP * p0.j - 1 + wave;
                    const bool pvalid = p >= 0 && p < F2;           // scalar
                    int addr[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) addr[cc] = lane_a1 + ((base1 + RC1 + 2 * wave - 2 + cc) % RC1) * 16;
                    int slot2 = base2 + wave;
                    slot2 = slot2 >= RC2 ? slot2 - RC2 : slot2;
                    char* dst2 = smem + lane_a2 + slot2 * 16;
                    // row tiles with a frame below T: r0 - 1 + 16 rt < T
                    const int nrt = pvalid ? min(RT, (a.T - p0.r0 + 16) >> 4) : 0;
                    uint4 x[3][4];
#define CONVF_LOAD_ROW2(KH, RTI)                                                                            \
                    _Pragma("unroll") for (int cc = 0; cc < 4; ++cc)                                        \
                        x[KH][cc] = *reinterpret_cast<const uint4*>(smem + addr[cc] + (16 * (RTI) + (KH)) * ROWB1);
                    if (nrt > 0) {
                        CONVF_LOAD_ROW2(0, 0)
                        CONVF_LOAD_ROW2(1, 0)
                        CONVF_LOAD_ROW2(2, 0)
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        uint4 o = make_uint4(0, 0, 0, 0);
                        if (rt < nrt) {
                            f32x4_t acc[2][2];
#pragma unroll
                            for (int e = 0; e < 2; ++e)
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt) acc[e][nt] = sh2[nt];
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        const int kw = cc - e;
                                        if (kw < 0 || kw > 2) continue;
#pragma unroll
                                        for (int nt = 0; nt < 2; ++nt) acc[e][nt] = mfma16(wf[kh * 3 + kw][nt], x[kh][cc], acc[e][nt]);
                                    }
                                // the fragment reads of the next row tile go out right behind the last use of this tap row
                                if (rt + 1 < RT) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (kh == 0) { CONVF_LOAD_ROW2(0, rt + 1) } else if (kh == 1) { CONVF_LOAD_ROW2(1, rt + 1) } else { CONVF_LOAD_ROW2(2, rt + 1) }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            float v[8];
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = fmaxf(fmaxf(acc[0][nt][r], acc[1][nt][r]), 0.f);
                            const int t2 = p0.r0 - 1 + 16 * rt + trow;
                            const uint32_t rowmask = (unsigned)t2 < (unsigned)a.T ? 0xffffffffu : 0u;
                            o = make_uint4(pack_bf16x2(v[0], v[1]) & rowmask, pack_bf16x2(v[2], v[3]) & rowmask,
                                           pack_bf16x2(v[4], v[5]) & rowmask, pack_bf16x2(v[6], v[7]) & rowmask);
                        }
                        *reinterpret_cast<uint4*>(dst2 + (16 * rt) * ROWB2) = o;
                    }
#undef CONVF_LOAD_ROW2
                }
                pos_next(p0, nstep, ntt);
                CONVF_TICK(0);
                // feature staging and this wave's layer1 unit of the NEXT step come last: the consumer wave of this SIMD starts its step
                // with its layer1 unit and runs its matrix work now
                // these are latency chains (load -> log2 -> LDS; LDS -> 8 small MFMAs -> convert -> LDS) beside the other wave's dense MFMA
                // stream: at raised priority every instruction that is ready goes first
                __builtin_amdgcn_s_setprio(3);
                if (!CONVF_DBG(8)) {
                    if (kk + 2 < K) CONVF_WRITE_SLAB(kk + 2);
                    if (kk + 3 < K) {
                        CONVF_ISSUE_LOADS(pl);
                        pos_next(pl, nstep, ntt);
                    }
                }
                if (kk + 1 < K) {
                    if (!CONVF_DBG(1)) CONVF_L1_PRODUCER(p1);
                    pos_next(p1, nstep, ntt);
                }
                __builtin_amdgcn_s_setprio(2);
                CONVF_TICK(1);
            }
            lds_only_barrier();
            CONVF_TICK(2);
        }
#ifdef AMTX_CONVF_TIMING
        if (tid == 0) {
            for (int i = 0; i < 3; ++i) atomicAdd(&g_convf_prof[i], prof_acc[i]);
            atomicAdd(&g_convf_prof[3], (unsigned long long)(K + 1));
        }
#endif
#undef CONVF_ISSUE_LOADS
#undef CONVF_WRITE_SLAB
#undef CONVF_L1_PRODUCER
#undef CONVF_L1_PUNIT
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
        if (tid - 256 < 32) reinterpret_cast<float*>(smem + SH1_OFF)[tid - 256] = a.shift1[(int64_t)grp * 32 + tid - 256];
        for (int i = tid - 256; i < 512; i += 256)
            reinterpret_cast<uint4*>(smem + W1_OFF)[i] = reinterpret_cast<const uint4*>(a.w1frag + (int64_t)grp * a.w1_gs)[i];
        const f32x4_t* shl0 = reinterpret_cast<const f32x4_t*>(smem + SH3_OFF);
        __syncthreads();
        // this wave's layer1 unit: u = wave (4 .. 7) of every step
        // (fragments from LDS, two columns in flight: the registers next to layer3's 144 weight registers do not hold all eight)
#define CONVF_L1_CUNIT(P, U)                                                                                \
        do {                                                                                                \
            int lane_o = lane;                                                                              \
            asm volatile("" : "+v"(lane_o)); /* lane constants recomputed here, not kept live through layer3 */ \
            const int n16 = lane_o & 15, g1 = lane_o >> 4, gg = min(g1, 2);                                 \
            CONVF_L1_BEGIN(P, U)                                                                            \
            const char* wl = smem + W1_OFF + lane_o * 16;                                                   \
            f32x4_t sh1l[2];                                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) sh1l[nt] = *reinterpret_cast<const f32x4_t*>(smem + SH1_OFF + (g1 * 8 + 4 * nt) * 4); \
            uint4 w1l[4][2];                                                                                \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) w1l[q][nt] = *reinterpret_cast<const uint4*>(wl + (2 * q + nt) * 1024); \
            f32x4_t acc1[4][2];                                                                             \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) acc1[q][nt] = mfma16(w1l[q][nt], bh, sh1l[nt]); \
            if (lean) {                                                                                     \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) CONVF_L1_STORE_LEAN(q, acc1[q][0], acc1[q][1]); \
            } else {                                                                                        \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) CONVF_L1_STORE(q, acc1[q][0], acc1[q][1]);    \
            }                                                                                               \
        } while (0)
#define CONVF_L1_CONSUMER(P)                                                                                \
        do {                                                                                                \
            CONVF_L1_CUNIT(P, wave);                                                                        \
        } while (0)
        Pos pc1 = pos_first(first, ntt);
        CONVF_L1_CONSUMER(pc1);
        pos_next(pc1, nstep, ntt);
        lds_only_barrier();

        // this wave's share of a step: output column mi of the two, row tiles rt0, rt0 + 1 of the four
        const int mi = (wave - 4) & 1;
        const int rt0 = ((wave - 4) >> 1) * 2;
        const int lane_a2 = A2_OFF + g * PLANE2 + (16 * rt0 + trow) * ROWB2;
        Pos pc = pos_first(first, ntt);
#ifdef AMTX_CONVF_TIMING
        unsigned long long prof_acc[4] = {0, 0, 0, 0};
        unsigned long long prof_t = __builtin_readcyclecounter();
#endif
        for (int kk = 0; kk <= K; ++kk) {
            // layer1 unit(s) of the step after next first: vector-ALU work in the shadow of the producer wave's layer2
            if (kk + 1 < K) {
                __builtin_amdgcn_s_setprio(3);
                if (!CONVF_DBG(1)) CONVF_L1_CONSUMER(pc1);
                __builtin_amdgcn_s_setprio(0);
                pos_next(pc1, nstep, ntt);
            }
            CONVF_TICK(0);
            if (kk >= 1) {
                if (!CONVF_DBG(4)) {
                    const int base2 = (pc.kk * CP) % RC2;           // ring slot of pooled column 4 j - 1 of the step consumed now
                    const int m = (CP / 2) * pc.j - 1 + mi;         // output column; its a2 columns 2 m - 1 + cc at ring offsets 2 mi - 2 + cc
                    const bool mvalid = m >= 0 && m < F4;
                    // this wave's row tiles with an output frame below T: r0 + 16 rt < T
                    const int nrt = mvalid ? min(2, (a.T - pc.r0 + 15 - 16 * rt0) >> 4) : 0;
                    int addr[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) addr[cc] = lane_a2 + ((base2 + RC2 + 2 * mi - 2 + cc) % RC2) * 16;
                    uint4 x[3][4];
#define CONVF_LOAD_ROW3(KH, RI)                                                                             \
                    _Pragma("unroll") for (int cc = 0; cc < 4; ++cc)                                        \
                        x[KH][cc] = *reinterpret_cast<const uint4*>(smem + addr[cc] + (16 * (RI) + (KH)) * ROWB2);
                    if (nrt > 0) {
                        CONVF_LOAD_ROW3(0, 0)
                        CONVF_LOAD_ROW3(1, 0)
                        CONVF_LOAD_ROW3(2, 0)
                    }
#pragma unroll
                    for (int ri = 0; ri < 2; ++ri) {
                        if (ri < nrt) {
                            f32x4_t acc[2][4];
#pragma unroll
                            for (int e = 0; e < 2; ++e)
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt) acc[e][nt] = shl0[(nt >> 1) * 8 + g * 2 + (nt & 1)];
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        const int kw = cc - e;
                                        if (kw < 0 || kw > 2) continue;
#pragma unroll
                                        for (int nt = 0; nt < 4; ++nt) acc[e][nt] = mfma16(wf[kh * 3 + kw][nt], x[kh][cc], acc[e][nt]);
                                    }
                                if (ri == 0) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (kh == 0) { CONVF_LOAD_ROW3(0, 1) } else if (kh == 1) { CONVF_LOAD_ROW3(1, 1) } else { CONVF_LOAD_ROW3(2, 1) }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                            const int o = 16 * (rt0 + ri) + trow;
                            const int t = pc.r0 + o;
                            if (o < R3 && t < a.T) {
                                float v[16];
#pragma unroll
                                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[nt * 4 + r] = fmaxf(fmaxf(acc[0][nt][r], acc[1][nt][r]), 0.f);
                                const int64_t orow = (int64_t)pc.b * a.T + t;
                                uint4* dst = reinterpret_cast<uint4*>(a.out + (int64_t)grp * a.out_gs + (a.out_plane ? m * a.out_plane + orow * 64 : (orow * F4 + m) * 64) + g * 8);
#pragma unroll
                                for (int q = 0; q < 2; ++q)
                                    dst[4 * q] = make_uint4(pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                                                            pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7]));
                            }
                        }
                    }
#undef CONVF_LOAD_ROW3
                }
                pos_next(pc, nstep, ntt);
            }
            CONVF_TICK(1);
            lds_only_barrier();
            CONVF_TICK(2);
        }
#undef CONVF_L1_CUNIT
#undef CONVF_L1_CONSUMER
#ifdef AMTX_CONVF_TIMING
        if (tid == 256) {
            for (int i = 0; i < 3; ++i) atomicAdd(&g_convf_prof[8 + i], prof_acc[i]);
            atomicAdd(&g_convf_prof[11], (unsigned long long)(K + 1));
        }
#endif
    }
}

#undef CONVF_L1_BEGIN
#undef CONVF_L1_STORE
#undef CONVF_L1_STORE_LEAN

}  // namespace

// Steps per strip: step j finishes output columns 2 j - 1 and 2 j, so the smallest n with 2 (n - 1) >= F / 4 - 1 (the zero column
// behind the pooled map, 2 (F / 4) at most, is then inside that step's four pooled columns 4 (n - 1) - 1 .. 4 (n - 1) + 2 as well)
static int convf_steps(int F) {
    const int F4 = (F >> 1) >> 1;
    int n = 1;
    while (2 * (n - 1) < F4 - 1) ++n;
    return n;
}

// One launch handles `groups` heads x B clips x T frames.  The caller decides WHEN (amtx_conv_stack_fused_ok): below ~one strip per
// CU the two-kernel path's finer tiles fill the chip better.
// 62 output rows per strip (ninth layer1 unit) only where that saves a strip per clip, else 60
static bool convf_halo(int T) { return (T + r3_of(true) - 1) / r3_of(true) < (T + r3_of(false) - 1) / r3_of(false); }

bool amtx_conv_stack_fused_ok(int B, int T, int F, int groups) {
    if (F < 4 || B <= 0 || T <= 0 || groups <= 0) return false;
    const int r3 = r3_of(convf_halo(T));
    const int64_t strips = (int64_t)B * ((T + r3 - 1) / r3) * groups;
    return strips >= 256 && (int64_t)B * ((T + r3 - 1) / r3) < (1ll << 30);
}

int amtx_launch_conv_stack(const ConvArgs& c2, const bf16_t* w3frag, int64_t w3_gs, const float* shift3, void* out, int64_t out_gs,
                           int64_t out_plane, hipStream_t stream) {
    AMTX_REQUIRE(c2.feats && c2.c_in == 1 && c2.w1frag && c2.shift1 && c2.wfrag && c2.shift && w3frag && shift3 && out,
                 "conv_stack: null pointer / not a one-channel input");
    AMTX_REQUIRE(c2.planes == 1 && c2.c_out == 32 && c2.out_type == AMTX_T_BF16, "conv_stack: bf16, 32 -> 32 -> 64 channels only");
    AMTX_REQUIRE(c2.B > 0 && c2.T > 0 && c2.F >= 4 && c2.groups > 0, "conv_stack: bad sizes");
    AMTX_REQUIRE(c2.f_stride_t >= 0 && c2.f_stride_f >= 0 && (int64_t)c2.T * c2.f_stride_t + (int64_t)c2.F * c2.f_stride_f < (1ll << 31),
                 "conv_stack: a clip's feature view must span fewer than 2^31 elements with non-negative strides");
    ConvFArgs a;
    a.feats = c2.feats; a.f_stride_b = c2.f_stride_b; a.f_stride_t = c2.f_stride_t; a.f_stride_f = c2.f_stride_f;
    a.f_clip_max = c2.f_clip_max; a.f_ref = c2.f_ref ? c2.f_ref : c2.f_clip_max;
    a.w1frag = c2.w1frag; a.w1_gs = c2.w1_gs; a.shift1 = c2.shift1;
    a.w2frag = c2.wfrag; a.w2_gs = c2.w_gs; a.shift2 = c2.shift;
    a.w3frag = w3frag; a.w3_gs = w3_gs; a.shift3 = shift3;
    a.out = (bf16_t*)out; a.out_gs = out_gs; a.out_plane = out_plane;
    a.B = c2.B; a.T = c2.T; a.F = c2.F;
    const bool halo = convf_halo(c2.T);
    const int r3 = r3_of(halo);
    const int ntt = (c2.T + r3 - 1) / r3;
    const int64_t nstrips = (int64_t)c2.B * ntt;
    AMTX_REQUIRE(nstrips < (1ll << 30), "conv_stack: grid too large");
    const int nstep = convf_steps(c2.F);
    AMTX_REQUIRE((int64_t)nstep * (nstrips + 1) < (1ll << 31), "conv_stack: too many steps");
    int gx = std::max(1, 256 / c2.groups);             // one resident block per CU in total
    if (gx > nstrips) gx = (int)nstrips;
    const int per_block = (int)((nstrips + gx - 1) / gx);
    gx = (int)((nstrips + per_block - 1) / per_block);
    auto kern = halo ? convf_kernel<true> : convf_kernel<false>;
    AMTX_GRANT_LDS(convf_kernel<true>, LDS_BYTES);
    AMTX_GRANT_LDS(convf_kernel<false>, LDS_BYTES);
    int dbg = 0;
#if defined(AMTX_CONVF_TIMING) || defined(AMTX_CONVF_ABLATE)
    if (const char* e = getenv("AMTX_CONVF_DBG")) dbg = atoi(e);
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)c2.groups), dim3(512), LDS_BYTES, stream, a, nstep, ntt, (int)nstrips, per_block, dbg);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

#ifdef AMTX_CONVF_TIMING
extern "C" int amtxdbg_convf_prof(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_convf_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_convf_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
