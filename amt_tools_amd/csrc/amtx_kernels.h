// Internal launch interface of the dense kernels (shared by the op-level C ABI and the model engine).
#pragma once
#include "amtx_common.h"

// element types of activations held in HBM.  AMTX_T_SPLIT (round 5, the x3 precision's own activation format): TWO 16-bit planes,
// hi = 16-bit(x) at the tensor's base pointer and lo = 16-bit(x - hi) `*_split` elements behind it -- the operand planes of the
// three-MFMA product, written once by the producing kernel instead of being re-derived from fp32 by every consumer (and therefore
// loadable HBM -> LDS by DMA).  Four bytes per element like fp32.
enum { AMTX_T_BF16 = 0, AMTX_T_F32 = 1, AMTX_T_SPLIT = 2 };

static inline size_t amtx_tsize(int t) { return t == AMTX_T_BF16 ? 2 : 4; }

// ---------------------------------------------------------------- GEMM  C[M,N] = A[M,K] . W[N,K]^T + bias
// W is pre-packed: [planes][n_pad][k_pad] bf16 (planes = 1 for bf16, 2 = hi/lo for x3), zero padded.
struct GemmArgs {
    const void* A; int64_t lda; int a_type;            // row-major, K contiguous, rows 16-byte aligned
    const bf16_t* W; int n_pad, k_pad, planes;
    const float* bias;                                   // [N] or null
    void* C; int64_t ldc; int c_type;
    int64_t M; int N, K;
    int groups; int64_t a_gs, w_gs, bias_gs, c_gs;       // per-group strides in elements (grid.z = groups)
    // optional magnitude epilogue (fp32-A / fp32-C kernel only; the CQT basis products): columns (2p, 2p+1) are (re, im) of output
    // pair p and what is stored is the POWER re^2 + im^2 (round 5; the magnitude until then), TRANSPOSED: pair_out[grp * pair_gs + pair_map[p].x * pair_pitch + m] for rows
    // m < pair_rows[pair_map[p].y]; C is not written.
    const int2* pair_map = nullptr; float* pair_out = nullptr; int64_t pair_gs = 0, pair_pitch = 0;
    int pair_rows[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // optional with the magnitude epilogue: running maximum of the stored powers per (group, pair_map[p].y): pair_max[grp * pair_nh + y]
    // (atomic max on the float bits; the caller zeroes it first)
    float* pair_max = nullptr; int pair_nh = 0;
    // optional piano-roll epilogue (bf16-A / fp32-C direct-to-LDS kernel, amtx_gemm_has_roll_epilogue): rows are (clip, frame) with roll_T
    // frames per clip; group roll_group also writes roll_out[clip][n][frame] = sigmoid(c) (roll_thr < 0) or sigmoid(c) < roll_thr ? 0 : 1.
    // C may then be null (logits not wanted).
    float* roll_out = nullptr; int roll_T = 0; float roll_thr = 0.5f; int roll_group = 0;
    // optional, same kernel: a bf16 copy of the output rows into copy16[m * copy16_ld + copy16_col0 + grp * copy16_gs + n] with copy16_pad zero
    // columns behind column N - 1 (N + copy16_pad <= n_pad); C may be null
    bf16_t* copy16 = nullptr; int64_t copy16_ld = 0; int copy16_col0 = 0, copy16_gs = 0, copy16_pad = 0;
    // A stored in PLANES of 64 columns (direct-to-LDS kernels only): element (m, k) at A[(k / 64) * a_plane + m * 64 + k % 64] -- the layout the
    // fused convolution stack writes (one plane per pooled frequency column: a 64-deep k-tile of 256 rows is 32 KiB of CONTIGUOUS memory
    // instead of 256 pieces of 128 bytes 7296 bytes apart).  0: row-major with lda.
    int64_t a_plane = 0;
    // optional (fp32-A kernels): only elements [a_valid_lo, a_valid_hi) of a group's A, counted from its base pointer A + grp * a_gs, exist;
    // everything else reads as zero and is never dereferenced (strided rows hanging over both ends of a clip: the CQT basis product
    // straight from the caller's audio, whose centre padding is then implicit).  a_valid_hi == 0: no bounds.
    int64_t a_valid_lo = 0, a_valid_hi = 0;
    // AMTX_T_SPLIT operands: elements between the hi and the lo plane of A / of C
    int64_t a_split = 0, c_split = 0;
};
bool amtx_gemm_has_roll_epilogue(const GemmArgs& g);
int amtx_launch_gemm(const GemmArgs& g, hipStream_t stream);
// several fp32-A / fp32-C / two-plane problems with one group count in one launch (generic 128 x 128 kernel)
constexpr int AMTX_GEMM_MULTI_MAX = 10;
struct GemmMulti { GemmArgs p[AMTX_GEMM_MULTI_MAX]; int n; };
int amtx_launch_gemm_multi(const GemmArgs* gs, int n, hipStream_t stream);
void amtx_gemm_pack_dims(int N, int K, int* n_pad, int* k_pad);
// host packing: W (N x K fp32 row-major, leading dim ldw) -> [planes][n_pad][k_pad] bf16
void amtx_gemm_pack_host(const float* W, int64_t ldw, int N, int K, int planes, bf16_t* out);

// ---------------------------------------------------------------- conv3x3 (C_in = 32) + folded BN + ReLU + MaxPool(1,2)
struct ConvArgs {
    const void* in; int in_type;                         // [B][T][F][32] channels-last
    const bf16_t* wfrag; int planes;                     // packed fragments, see conv.hip
    const float* shift;                                  // [c_out] folded BN shift (+ conv bias)
    void* out; int out_type;                             // [B][T][F/2][c_out]
    int B, T, F, c_out;
    int groups; int64_t in_gs, w_gs, shift_gs, out_gs;   // per-group strides in elements
    // optional fused first conv (C_in*9 <= 64 -> 32 channels): when `feats` is set, `in` is ignored and the
    // kernel computes Conv(c_in->32)+BN+ReLU from the fp32 features straight into its LDS input tile
    const float* feats = nullptr; int64_t f_stride_b = 0, f_stride_c = 0, f_stride_t = 0, f_stride_f = 0;
    int c_in = 0;
    const bf16_t* w1frag = nullptr; const float* shift1 = nullptr;   // per-group strides: w1_gs, 32
    int64_t w1_gs = 0;
    // c_in = 1 only: `feats` holds raw POWER values and the kernel applies the dB scaling (db_scale_apply, amtx_common.h) while it stages
    // them: f_clip_max[b] = the clip's own maximum, f_ref[b] = the reference power (null: the own maximum).  Null f_clip_max: features as is.
    const float* f_clip_max = nullptr; const float* f_ref = nullptr;
    // convg.hip's tap-major fused first conv only: the features as [B][T][F][8] 16-bit channels-last (channel slots c_in .. 7 zero) instead of
    // `feats` (amtx_cqt_forward16 writes them so): a position is one 16-byte load
    const void* feats16 = nullptr;
    int64_t out_ts = 0;                                  // convg.hip only: elements between consecutive (b, t) rows of `out`; 0 = (F/2) * c_out
    int64_t in_split = 0, out_split = 0;                 // AMTX_T_SPLIT maps (conv.hip, two-plane mode): elements between the hi and the lo plane
};
int amtx_launch_conv3x3(const ConvArgs& c, hipStream_t stream);
// convx.hip: the same layer (no fused first conv) on AMTX_T_SPLIT maps, two-plane weights: tiles DMA'd into a second LDS buffer under the matrix work
int amtx_launch_convx3(const ConvArgs& c, hipStream_t stream);
// convx.hip: conv.hip's fused first conv + 32 -> 32 layer (c_in = 1, two-plane weights) with a2 written as AMTX_T_SPLIT planes, two a1 tiles in LDS
int amtx_launch_convx12(const ConvArgs& c, hipStream_t stream);
size_t amtx_conv1_wfrag_elems(int c_in, int planes);
// host packing of the fused first conv: weight (32, c_in, 3, 3) fp32 * scale[32] -> fragment order
void amtx_conv1_pack_host(const float* w, const float* scale, int c_in, int planes, bf16_t* out);
size_t amtx_conv3x3_wfrag_elems(int c_out, int planes);
// host packing: weight (c_out, 32, 3, 3) fp32 * scale[c_out] -> fragment order
void amtx_conv3x3_pack_host(const float* w, const float* scale, int c_out, int planes, bf16_t* out);

// the whole stack layer1 -> layer2 -> layer3 of a one-channel, 32/32/64-channel, bf16 model in one kernel (convf.hip): `c2` as for
// amtx_launch_conv3x3 with the fused first conv (feats, w1frag, shift1, wfrag, shift; `out` ignored), plus layer3's packed weights and
// shift; out = [groups][B][T][F / 4][64] bf16.  amtx_conv_stack_fused_ok: the batch is large enough for its one-strip-per-CU granularity.
bool amtx_conv_stack_fused_ok(int B, int T, int F, int groups);
int amtx_launch_conv_stack(const ConvArgs& c2, const bf16_t* w3frag, int64_t w3_gs, const float* shift3, void* out, int64_t out_gs,
                           int64_t out_plane, hipStream_t stream);

// general channel counts (convg.hip): C_in a multiple of 16, weights staged in LDS per C_out chunk; `a.in` is [B][T][F][c_in]
int amtx_conv3x3_gen_ntc(int c_in, int c_out);           // 0 = this pair of channel counts is not built
size_t amtx_conv3x3_gen_wfrag_elems(int c_in, int c_out, int planes);
void amtx_conv3x3_gen_pack_host(const float* w /*(c_out,c_in,3,3)*/, const float* scale, int c_in, int c_out, int planes, bf16_t* out);
int amtx_launch_conv3x3_gen(const ConvArgs& c, int c_in, hipStream_t stream);
// fused first conv of the general kernel (ConvArgs.feats / c_in / w1frag / shift1 as for conv.hip; `c_in` above = its output channels)
// Two K orders: the weight tensor's own (ci, kh, kw) padded to whole 32-deep steps (one-channel inputs, and the two-plane x3 mode), and -- for
// 2 .. 8 input channels (round 5: in the two-plane mode too, both planes of the features staged once) -- TAP-MAJOR with the channels padded to 8 (k = 8 tap + ci, 9 taps -> 3 steps): the kernel then
// stages its feature tile channels-last in bf16 and a lane's 8 K values of a step are ONE 16-byte LDS read (round 4; before, 16 scalar
// gathers + 8 conversions per 16 positions made the first conv 41 % of the HCQT model's conv2 kernel).
static inline __host__ __device__ bool amtx_conv1g_tapk(int c_in, int planes) { return (planes == 1 || planes == 2) && c_in >= 2 && c_in <= 8; }
size_t amtx_conv1g_wfrag_elems(int c_in, int c_mid, int planes);
void amtx_conv1g_pack_host(const float* w /*(c_mid,c_in,3,3)*/, const float* scale, int c_in, int c_mid, int planes, bf16_t* out);
bool amtx_conv3x3_gen_can_fuse1(int c_in, int c_mid, int c_out, int planes);

// ---------------------------------------------------------------- first conv (small C_in) + folded BN + ReLU, direct
struct Conv1Args {
    const float* in; int64_t stride_b, stride_c, stride_t, stride_f;   // fp32 features, arbitrary strides (elements)
    const float* w;                                      // [c_out][c_in][3][3] fp32, BN scale folded in
    const float* shift;                                  // [c_out]
    void* out; int out_type;                             // [B][T][F][c_out] channels-last
    int B, T, F, c_in, c_out;
    int groups; int64_t w_gs, shift_gs, out_gs;
    int relu = 1;                                        // 0: plain convolution + shift (the training path applies BatchNorm first)
};
int amtx_launch_conv1(const Conv1Args& c, hipStream_t stream);

// ---------------------------------------------------------------- BiLSTM recurrence (hidden = 128 per direction)
struct LstmArgs {
    const void* xproj; int x_type;                       // [B][T][2][512]: W_ih x + b_ih + b_hh, gate order i,f,g,o
    const bf16_t* whh; int planes;                       // packed fragments [dir][...], see lstm.hip
    void* out; int out_type;                             // [B][T][256] = h_fwd | h_bwd
    int B, T;
    int groups; int64_t x_gs, w_gs, out_gs;
    int hidden = 128;                                    // per direction; != 128: xproj [B][T][2][4 hidden], out [B][T][2 hidden]
    float* save = nullptr;                               // training only: [groups][B][T][2][5][128] post-activation i,f,g,o and c (4-clip kernel)
};
int amtx_launch_bilstm(const LstmArgs& l, hipStream_t stream);
// training: device-side packing of fp32 W_hh into forward + transposed (backward) fragments; backward recurrence -> dL/d(xproj)
int amtx_launch_bilstm_pack_dev(const float* whh_fwd, const float* whh_bwd, int planes, bf16_t* frag_fwd, bf16_t* frag_bwd, hipStream_t stream);
int amtx_launch_bilstm_bwd(const float* dout, const float* save, const bf16_t* whh_t, int planes, float* dxproj, int B, int T, hipStream_t stream);
size_t amtx_bilstm_wfrag_elems(int planes);              // per LSTM (both directions)
void amtx_bilstm_pack_host(const float* whh_fwd, const float* whh_bwd, int planes, bf16_t* out);   // each (512,128)
int amtx_launch_bilstm_pack_dev_h(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, bf16_t* frag_fwd, bf16_t* frag_bwd, hipStream_t stream);
int amtx_launch_bilstm_bwd_h(const float* dout, const float* save, const bf16_t* whh_t, int hidden, int planes, float* dxproj, int B, int T, int groups,
                             hipStream_t stream);
size_t amtx_bilstm_wfrag_elems_h(int hidden, int planes);
void amtx_bilstm_pack_host_h(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, bf16_t* out);   // each (4 hidden, hidden)

// ---------------------------------------------------------------- logits -> piano roll
// out[b][k][t] = threshold < 0 ? sigmoid(x) : (sigmoid(x) < threshold ? 0 : 1), x = logits[(b*T+t)*ld + col0 + k]
size_t amtx_bce_loss_partials(int B, int T, int keys);
int amtx_launch_bce_loss(const float* logits, int64_t ld, const float* labels, const float* weight, int B, int T, int keys, float* loss,
                         float* grad, float* partial, hipStream_t stream);
int amtx_launch_pianoroll(const float* logits, int64_t ld, int col0, int B, int T, int keys, float threshold, float* out,
                          hipStream_t stream);

int amtx_launch_cvt_pad_bf16(const float* src, int64_t ld_src, int n_src, bf16_t* dst, int ld_dst, int64_t rows, hipStream_t stream,
                             bool f16 = false /* IEEE half instead of bf16 (AMTX_PREC_F16) */);
// fp32 rows -> AMTX_T_SPLIT rows (hi / lo planes `split` elements apart), columns n_src .. ld_dst zero
int amtx_launch_cvt_split(const float* src, int64_t ld_src, int n_src, bf16_t* dst, int ld_dst, int64_t split, int64_t rows, hipStream_t stream);
int amtx_launch_zero_cols(void* base, int64_t pitch_bytes, int width_bytes, int64_t rows, hipStream_t stream);

// ---------------------------------------------------------------- device-side weight packing (pack.hip): the host packers' layouts and
// arithmetic as kernels, for a weight re-sync that does not leave the GPU (amtx_of_model_finalize_device)
int amtx_pack_bn_fold_dev(const float* conv_bias, const float* gamma, const float* beta, const float* mean, const float* var, int c_out, float* scale,
                          float* shift, hipStream_t s);
int amtx_pack_conv3x3_dev(const float* w, const float* scale, int c_out, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_conv1_dev(const float* w, const float* scale, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_conv_gen_dev(const float* w, const float* scale, int c_in, int c_out, int ntc, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_conv1g_dev(const float* w, const float* scale, int c_in, int c_mid, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_scale_rows_dev(const float* w, const float* scale, int rows, int cols, float* out, hipStream_t s);
int amtx_pack_linear_dev(const float* W, int64_t ldw, int N, int K, int planes, int n_pad, int k_pad, int row0, int rows_owned, int perm_c, int perm_f,
                         bf16_t* out, hipStream_t s);
int amtx_pack_head_fold_dev(const float* w_out, const float* w_fc1, const float* b_fc1, const float* b_out, int n_out, int dim_am, int kfc, int kfc_pad,
                            int nf3, int fq, float* wfold, float* bfold, hipStream_t s);
int amtx_pack_vec_add_dev(const float* a, const float* b, int n, float* out, hipStream_t s);

// ---- cqt_dec.hip: the half-band decimator of the CQT pyramid (cqt.hip builds its Toeplitz fragments and calls it per level)
constexpr int DEC_HALF = 150;                     // 301-tap Kaiser half-band filter
constexpr int DEC_TAPS = 2 * DEC_HALF + 1;
constexpr int DEC_KW = 352;                       // >= 30 + DEC_TAPS, a multiple of 32: columns of the 16-row Toeplitz matrix
constexpr int DEC_NKS = DEC_KW / 32;
// out[b][pad + m] = sqrt(2) sum_k h[k] in[b][in_pad + 2 m + k - DEC_HALF] (zero outside [0, n_in)), m < n_out; tfrag = DEC_NKS x 3 planes x 64 lanes
// x 16 bytes (cqt.hip, plan creation); zero_pads: the [0, pad) and [pad + n_out, out_stride) ranges of every output row are zeroed as well;
// maxbuf (or null): batch x n_harm floats reset to zero
int amtx_launch_cqt_decimate(const float* in, int64_t n_in, int64_t in_stride, int in_pad, float* out, int64_t n_out, int64_t out_stride, int pad,
                             const void* tfrag, int zero_pads, float* maxbuf, int n_harm, int batch, hipStream_t stream);
