/*
 * amtx -- C ABI of the MI355X-native (gfx950) hot path of amt-tools:
 *         spectral front-end (STFT / MelSpec / CQT) -> Onsets & Frames CNN + BiLSTM -> piano-roll.
 *
 * The reference (cwitkowitz/amt-tools) is pure Python; its boundary for this path is the duck-typed
 * plugin API (FeatureModule.process_audio, TranscriptionModel.run_on_batch).  The functions below are
 * what a binding of that API lands on: extern "C", plain pointers and sizes, explicit hipStream_t
 * (passed as void*), caller-owned DEVICE pointers unless a parameter says "host", no allocation
 * inside any *_forward / *_fwd call (plans own their read-only tables; the caller passes a
 * workspace sized by the matching *_workspace_bytes query), int return (0 = ok, < 0 = error, message
 * via amtx_last_error()).  Thread-safe for distinct (plan, stream) pairs.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 */
#ifndef AMTX_H
#define AMTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMTX_OK 0
#define AMTX_ERR_ARG (-1)
#define AMTX_ERR_HIP (-2)
#define AMTX_ERR_UNSUPPORTED (-3)

/* precision of the dense (MFMA) stages */
#define AMTX_PREC_BF16 0   /* bf16 operands, fp32 accumulate (headline mode)                     */
#define AMTX_PREC_X3 1     /* split-bf16 (hi+lo) operands, 3 MFMAs per product: fp32-class parity */
#define AMTX_PREC_F16 2    /* IEEE half operands, fp32 accumulate: the bf16 mode's speed, 3 more mantissa bits (engine only; values beyond
                            * +-65504 would overflow -- log-mel / CQT features in [0, 1] and BatchNorm'd maps do not) */

const char* amtx_last_error(void);   /* thread-local message of the last failing call */
int amtx_version(void);

/* ------------------------------------------------------------------------------------------------
 * Spectral front-end: STFT power -> (mel filterbank) -> per-clip max  [K1]  and dB scaling [K2].
 * Replaces librosa.stft / librosa.feature.melspectrogram / power_to_db / amplitude_to_db as called
 * from amt_tools/features/stft.py:66-72, amt_tools/features/mel.py:64-71,94 and
 * amt_tools/features/common.py:199,218-228.
 * ------------------------------------------------------------------------------------------------ */
typedef struct amtx_spec_plan amtx_spec_plan;

#define AMTX_PAD_CONSTANT 0 /* librosa >= 0.10 centre padding (zeros) */
#define AMTX_PAD_REFLECT 1  /* librosa 0.9 centre padding (reflect)   */

/* n_mels > 0: mel-power plan (MelSpec); n_mels == 0: plain STFT power plan (STFT), n_bins = n_fft/2+1.
 * Builds the window / twiddle / sparse-filterbank tables on the host in fp64 and uploads them. */
int amtx_spec_plan_create(amtx_spec_plan** plan, int sample_rate, int n_fft, int hop_length, int win_length,
                          int n_mels, int htk, int center, int pad_mode);
/* Host-only (no device): how a plan with these parameters deals its mel rows to the lanes of the gather kernel.  slot s = 64 * round +
 * lane computes mel row slot_row[s] (-1: none) from FFT bin slot_start[s] on (an even number of zero taps may precede the row's first
 * non-zero weight); round_max[r] = taps per slot of round r.  Rows are placed so that the 32 lanes of every half-wave start on 32
 * different LDS banks (spec.hip, mel_assign_slots).  Returns the number of rounds (>= 1) or a negative error code. */
int amtx_spec_mel_layout(int sample_rate, int n_fft, int n_mels, int htk, int32_t* slot_row, int32_t* slot_start, int32_t* round_max);
int amtx_spec_plan_destroy(amtx_spec_plan* plan);
int amtx_spec_num_bins(const amtx_spec_plan* plan);                    /* rows of the feature map        */
int64_t amtx_spec_num_frames(const amtx_spec_plan* plan, int64_t num_samples);   /* features/common.py:41-66 */
/* host copy of the dense filterbank (n_mels x (n_fft/2+1), float32), for inspection / tests */
int amtx_spec_filterbank(const amtx_spec_plan* plan, float* host_out);

/* K1: audio (B clips, `num_samples` each, clip b at audio + b*audio_stride, fp32) ->
 *     power  [B][T][n_bins] fp32 (mel power, or |X|^2 for an STFT plan) and clip_max[B] = max over the
 *     clip's whole map (the reduction behind power_to_db(ref=np.max), features/mel.py:94).        */
int amtx_spec_power(const amtx_spec_plan* plan, const float* audio, int64_t num_samples, int64_t audio_stride,
                    int batch, float* power, float* clip_max, void* stream);

/* K2 output layouts */
#define AMTX_LAYOUT_BFT_F32 0 /* [B][n_bins][T] fp32: the reference's (C=1,F,T) feature layout   */
#define AMTX_LAYOUT_BTF_F32 1 /* [B][T][n_bins] fp32: the model's transposed layout (onsetsframes.py:90) */
/* K2 value transforms */
#define AMTX_SCALE_DB 0       /* 10log10(max(amin,S)) - 10log10(max(amin,ref)), clamp to max-80, /80 + 1 */
#define AMTX_SCALE_POWER 1    /* S as is (decibels=False, MelSpec)                                 */
#define AMTX_SCALE_MAGNITUDE 2 /* sqrt(S) (decibels=False, STFT)                                    */
/* ref: per-clip reference power for the dB conversion (normally = clip_max from K1; pass a track-level
 * maximum to reproduce features sliced out of a longer track, SURVEY finding F7).                 */
int amtx_spec_scale(const amtx_spec_plan* plan, const float* power, const float* clip_max, const float* ref,
                    int batch, int64_t num_frames, int transform, int layout, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Onsets & Frames inference engine (one call = TranscriptionModel.run_on_batch without the loss):
 * amt_tools/models/common.py:151-184 run_on_batch, amt_tools/models/onsetsframes.py:94-136 forward,
 * :138-196 post_proc (finalize_output, threshold 0.5), AcousticModel :330-463, LanguageModel :466-575,
 * LogisticBank amt_tools/models/common.py:486-620.
 * ------------------------------------------------------------------------------------------------ */
typedef struct amtx_of_model amtx_of_model;

/* model_complexity 2, 3, 4 and 5 are built (conv channels 16 mc / 16 mc / 32 mc, fc 256 mc, LSTM hidden 128 (mc - 1): onsetsframes.py:37-41,
 * 358-364); anything else answers AMTX_ERR_UNSUPPORTED */
int amtx_of_model_create(amtx_of_model** model, int dim_in, int in_channels, int model_complexity, int n_out,
                         int has_offsets, int precision /* AMTX_PREC_* */);
/* 1 when the library was built with the half-operand kernel twins (AMTX_BUILD_F16=1): only then does amtx_of_model_create accept
 * AMTX_PREC_F16 (AMTX_ERR_UNSUPPORTED otherwise).  The default build has bf16 and x3 only -- no BASELINE config names f16. */
int amtx_has_f16(void);
int amtx_of_model_destroy(amtx_of_model* model);
/* hand over one tensor of the reference's state_dict under its own key (HOST pointer, fp32, row-major) */
int amtx_of_model_set_tensor(amtx_of_model* model, const char* name, const float* host_data, int64_t numel);
/* fold BatchNorm, permute fc1, merge LSTM biases, pack everything into MFMA fragment order, upload */
int amtx_of_model_finalize(amtx_of_model* model);
size_t amtx_of_workspace_bytes(const amtx_of_model* model, int batch, int num_frames);
/* feats: fp32 features addressed as feats[b*stride_b + c*stride_c + t*stride_t + f*stride_f] (elements), so both
 * the reference's (B,C,F,T) layout and the transposed (B,C,T,F) layout are accepted without a copy.
 * out_onsets / out_multi_pitch: (B, n_out, T) fp32 in {0,1}.  logits_*: optional (B, T, n_out) fp32 raw logits. */
int amtx_of_forward(const amtx_of_model* model, const float* feats, int64_t stride_b, int64_t stride_c, int64_t stride_t,
                    int64_t stride_f, int batch, int num_frames, void* workspace, size_t workspace_bytes,
                    float* out_onsets, float* out_multi_pitch, float* logits_onsets, float* logits_multi_pitch,
                    float* logits_pitch_head, void* stream);

/* The same forward pass fed with the front-end's RAW POWER mel spectrogram (amtx_spec_power's output, power[b*stride_b + t*stride_t +
 * f*stride_f]) instead of finished features: the dB scaling of amtx_spec_scale(AMTX_SCALE_DB) (amt_tools/features/common.py:199,
 * 218-228: power_to_db relative to `ref`, top_db = 80 below the clip's own maximum, / 80 + 1) is applied by the first conv kernel while
 * it stages the values, bit-identical to amtx_spec_scale, so the feature tensor never exists in HBM.  clip_max: [batch] own maxima
 * (amtx_spec_power), ref: [batch] reference powers or NULL (= own maximum).  Only for models where amtx_of_fuses_db_scale() is 1
 * (one input channel, first conv fused into conv2); AMTX_ERR_ARG otherwise -- callers then run amtx_spec_scale + amtx_of_forward. */
int amtx_of_fuses_db_scale(const amtx_of_model* model);
int amtx_of_forward_power(const amtx_of_model* model, const float* power, int64_t stride_b, int64_t stride_t, int64_t stride_f,
                          const float* clip_max, const float* ref, int batch, int num_frames, void* workspace, size_t workspace_bytes,
                          float* out_onsets, float* out_multi_pitch, float* logits_onsets, float* logits_multi_pitch,
                          float* logits_pitch_head, void* stream);

/* The same forward pass fed with features the front-end has laid out the way the first conv kernel stages them (amtx_cqt_forward16):
 * feats16 = [batch][num_frames][dim_in][8] bf16, the input channels (harmonics of an HCQT, amt_tools/features/hvqt.py:107-133) of a
 * position in one 16-byte slot, slots in_channels .. 7 zero.  The kernel then fetches a position with one load instead of in_channels
 * strided fp32 loads + conversions; the values are the bf16 roundings it would make of the fp32 features itself, so the results are the
 * bits of amtx_of_forward.  Only for models where amtx_of_takes_feats16() is non-zero (2 .. 8 input channels, model_complexity 2: BASELINE
 * config 3): 1 = bf16 precision, feats16 = [B][T][F][8]; 2 (round 6) = x3 precision, feats16 = the TWO planes of the split, [2][B][T][F][8] with
 * the lo plane exactly B T F 8 elements behind the hi plane (amtx_cqt_forward16_split).  AMTX_ERR_ARG otherwise -- callers then run
 * amtx_cqt_forward + amtx_of_forward. */
int amtx_of_takes_feats16(const amtx_of_model* model);
int amtx_of_forward_feats16(const amtx_of_model* model, const void* feats16, int batch, int num_frames, void* workspace, size_t workspace_bytes,
                            float* out_onsets, float* out_multi_pitch, float* logits_onsets, float* logits_multi_pitch,
                            float* logits_pitch_head, void* stream);

/* Weight RE-SYNC without leaving the GPU (validate() inside train(), amt_tools/train.py:183-189): after one amtx_of_model_finalize, later
 * weight versions can be handed over as DEVICE pointers (fp32, contiguous, same state_dict names, borrowed until finalize_device returns)
 * and are folded / packed by kernels into the model's existing buffers -- the same bits the host path produces.  Every built
 * configuration and precision; AMTX_ERR_UNSUPPORTED only for a multi-channel first conv forced onto conv.hip's kernel
 * (AMTX_NO_CONVG_MC2): callers then use set_tensor + finalize.
 * Both re-sync entry points (finalize after a first finalize, finalize_device) overwrite the packed buffers IN PLACE: they first wait for
 * every operation the device has been given (hipDeviceSynchronize), so a forward pass of the previous weight version still running on
 * another stream never reads half-updated weights; forwards enqueued AFTER the call on a stream other than `stream` must be ordered
 * behind it by the caller.  finalize_device validates every tensor (name, size) before it touches a buffer, drops the borrowed pointers
 * on every exit, and leaves the model un-finalized (forwards refuse to run) if a pack launch fails half-way. */
int amtx_of_model_set_tensor_device(amtx_of_model* model, const char* name, const float* device_data, int64_t numel);
int amtx_of_model_finalize_device(amtx_of_model* model, void* stream);

/* 1 when amtx_of_forward / amtx_of_forward_power at this batch and frame count run the acoustic heads' three convolution layers
 * (amt_tools/models/onsetsframes.py:375-412) as ONE kernel whose intermediate maps stay in LDS (convf.hip: one-channel bf16 models at
 * model_complexity 2, batches of at least 256 head x clip x 62-frame strips; AMTX_NO_CONV_FUSE=1 at model creation turns it off), 0 when
 * they run as the two kernels of conv.hip.  The results are the same bits either way; amtx_of_workspace_bytes accounts for it. */
int amtx_of_conv_stack_fused(const amtx_of_model* model, int batch, int num_frames);

/* OnsetsFrames2 (has_offsets = 1; amt_tools/models/onsetsframes.py:199-327): the offset head's output of the last amtx_of_forward
 * that used this workspace: out_offsets (B, n_out, T) sigmoid probabilities (finalize_output without threshold, :323-325),
 * logits_offsets optional (B, T, n_out). */
int amtx_of_offsets(const amtx_of_model* model, void* workspace, size_t workspace_bytes, int batch, int num_frames, float* out_offsets,
                    float* logits_offsets, void* stream);

/* Per-stage timing of amtx_of_forward with HIP events recorded on the launch stream (used by bench.py for the
 * live roofline figure; no reference counterpart -- the reference has no profiling). */
int amtx_of_num_stages(void);
const char* amtx_of_stage_name(int stage);
int amtx_of_profile_enable(amtx_of_model* model, int enable);
int amtx_of_profile_read(amtx_of_model* model, double* stage_ms /*[amtx_of_num_stages()]*/, int* num_forwards);

/* ------------------------------------------------------------------------------------------------
 * Op-level entry points (the kernels the engine is built from; used by the parity tests).
 * Element types: 0 = bf16, 1 = fp32 (2 = split planes where stated).  `planes` = 1 (bf16) or 2 (x3 split).  *_pack_* run on the host.
 * ------------------------------------------------------------------------------------------------ */
/* nn.Linear: C[M,N] = A[M,K] W[N,K]^T + bias  (models/onsetsframes.py:422-427, models/common.py:539) */
int64_t amtx_linear_packed_elems(int n, int k, int planes);
int amtx_linear_pack(const float* host_w, int n, int k, int planes, uint16_t* host_out);
int amtx_linear_fwd(const void* a, int64_t lda, int a_type, const uint16_t* w_packed, int planes, const float* bias, void* c,
                    int64_t ldc, int c_type, int64_t m, int n, int k, void* stream);
/* Element type 2 = "split" (the x3 precision's activation format since round 5): TWO 16-bit planes, hi = bf16(x) at the base pointer and
 * lo = bf16(x - hi) `split` elements behind it; four bytes per element like fp32.  amtx_split_planes writes them from fp32 rows (columns
 * n .. ld_dst zero); amtx_linear_fwd_split is amtx_linear_fwd with a split A (planes = 2 weights) and an fp32 (1) or split (2) C:
 * whole 256-column tiles with k == its packed width run on the direct-to-LDS two-plane kernel, everything else on the generic one.
 * Alignment: ldc % 4 == 0 always; the direct-to-LDS kernel stores 16 bytes at a time, so with a split C it is only chosen when
 * ldc % 8 == 0 and c_split % 8 == 0 (otherwise the generic kernel, 8-byte stores, runs -- same bits). */
int amtx_split_planes(const float* src, int64_t ld_src, int n, uint16_t* dst, int ld_dst, int64_t split, int64_t rows, void* stream);
int amtx_linear_fwd_split(const void* a, int64_t lda, int64_t a_split, const uint16_t* w_packed, const float* bias, void* c, int64_t ldc,
                          int c_type, int64_t c_split, int64_t m, int n, int k, void* stream);
/* Conv2d(32 -> c_out, 3x3, pad 1) + folded BatchNorm + ReLU + MaxPool(1,2), channels-last (models/onsetsframes.py:387-416) */
int64_t amtx_conv3x3_packed_elems(int c_out, int planes);
int amtx_conv3x3_pack(const float* host_w /*(c_out,32,3,3)*/, const float* host_scale /*[c_out] or null*/, int c_out, int planes,
                      uint16_t* host_out);
int amtx_conv3x3_fwd(const void* in, int elem_type, const uint16_t* w_packed, int planes, const float* shift, void* out, int batch,
                     int num_frames, int num_bins, int c_out, void* stream);
/* The same layer for the channel counts of the other model complexities (models/onsetsframes.py:362-364 nf = 16 mc / 16 mc / 32 mc;
 * built: 48 -> 48 and 48 -> 96 = model_complexity 3, OnsetsFrames2's default; 64 -> 64 and 64 -> 128 = model_complexity 4; 80 -> 80 and 80 -> 160 = 5).
 * `in` is [B][T][F][c_in] channels-last.
 * amtx_conv3x3g_packed_elems returns 0 for a pair of channel counts that is not built. */
int64_t amtx_conv3x3g_packed_elems(int c_in, int c_out, int planes);
int amtx_conv3x3g_pack(const float* host_w /*(c_out,c_in,3,3)*/, const float* host_scale /*[c_out] or null*/, int c_in, int c_out, int planes,
                       uint16_t* host_out);
int amtx_conv3x3g_fwd(const void* in, int elem_type, const uint16_t* w_packed, int planes, const float* shift, void* out, int batch,
                      int num_frames, int num_bins, int c_in, int c_out, void* stream);
/* Conv2d(c_in -> c_out, 3x3, pad 1) + folded BatchNorm + ReLU for the first layer (models/onsetsframes.py:375-384) */
int amtx_conv1_fwd(const float* feats, int64_t stride_b, int64_t stride_c, int64_t stride_t, int64_t stride_f, const float* w,
                   const float* shift, void* out, int out_type, int batch, int num_frames, int num_bins, int c_in, int c_out,
                   void* stream);
/* nn.LSTM(bidirectional, hidden 128) recurrence given xproj = W_ih x + b_ih + b_hh  (models/onsetsframes.py:498-529) */
int64_t amtx_bilstm_packed_elems(int planes);
int amtx_bilstm_pack(const float* host_whh_fwd, const float* host_whh_bwd /* (512,128) each */, int planes, uint16_t* host_out);
int amtx_bilstm_fwd(const void* xproj /*(B,T,2,512)*/, const uint16_t* whh_packed, int planes, int elem_type, void* out /*(B,T,256)*/,
                    int batch, int num_frames, void* stream);
/* Any built hidden size (128, 256, 384, 512 = dim_lm / 2 at model_complexity 2 .. 5; models/onsetsframes.py:57-58, 498-507):
 * W_hh (4 hidden, hidden) per direction, xproj (B,T,2,4 hidden), out (B,T,2 hidden). */
int64_t amtx_bilstm_h_packed_elems(int hidden, int planes);
int amtx_bilstm_h_pack(const float* host_whh_fwd, const float* host_whh_bwd, int hidden, int planes, uint16_t* host_out);
int amtx_bilstm_h_fwd(const void* xproj, const uint16_t* whh_packed, int hidden, int planes, int elem_type, void* out, int batch,
                      int num_frames, void* stream);
/* Training (amt_tools/train.py:126-141 drives nn.LSTM's forward + backward through autograd): the same recurrence with the
 * post-activation gates and cell states saved ([B][T][2][5][128] fp32: i, f, g, o, c), and its backward recurrence
 *   dout [B][T][256] -> dxproj [B][T][2][512] = dL/d(W_ih x + b).
 * Parameter gradients are GEMMs over B*T on the caller's side.  W_hh (fp32, DEVICE pointers, (512,128) per direction) is
 * repacked on the device into forward fragments (amtx_bilstm_packed_elems(planes) elements) and transposed fragments (same
 * size) after every optimizer step.  Two-plane (fp32-class) precision only. */
int amtx_bilstm_pack_device(const float* whh_fwd, const float* whh_bwd, int planes, uint16_t* frag_fwd, uint16_t* frag_bwd, void* stream);
int amtx_bilstm_train_fwd(const float* xproj, const uint16_t* whh_packed, int planes, float* out, float* save, int batch, int num_frames,
                          void* stream);
int amtx_bilstm_train_bwd(const float* dout, const float* save, const uint16_t* whh_t_packed, int planes, float* dxproj, int batch,
                          int num_frames, void* stream);
/* The same three calls for any built hidden size (128, 256, 384, 512): save is [B][T][2][5][hidden], dxproj [B][T][2][4 hidden], fragments
 * amtx_bilstm_h_packed_elems(hidden, planes) elements each.  Two-plane precision only.  `groups` independent LSTMs of the same
 * (batch, num_frames, hidden) run in one launch (the onset and offset recurrences of OnsetsFrames2): every array gets a leading
 * [groups] axis. */
int amtx_bilstm_h_pack_device(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, uint16_t* frag_fwd, uint16_t* frag_bwd,
                              void* stream);
int amtx_bilstm_h_train_fwd(const float* xproj, const uint16_t* whh_packed, int hidden, int planes, float* out, float* save, int batch,
                            int num_frames, int groups, void* stream);
int amtx_bilstm_h_train_bwd(const float* dout, const float* save, const uint16_t* whh_t_packed, int hidden, int planes, float* dxproj,
                            int batch, int num_frames, int groups, void* stream);
/* Training-mode BatchNorm2d (batch statistics, running statistics updated as nn.BatchNorm2d does) + ReLU (+ MaxPool2d((1,2)) when
 * pool = 1) of the acoustic model's conv stages (amt_tools/models/onsetsframes.py:375-416 under amt_tools/train.py:126-141), and its
 * backward.  Channels-last fp32: x (rows = B*T, num_bins, channels), y / dy (rows, num_bins / 2 or num_bins, channels).
 * stats [4][channels] (mean, invstd, gamma, beta) is written by the forward and read by the backward.
 * gamma / beta / running_* may be null; dgamma / dbeta may be null. */
size_t amtx_bn_train_workspace_bytes(int channels);
int amtx_bn_relu_pool_train_fwd(const float* x, int64_t rows, int num_bins, int channels, int pool, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* y, float* stats,
                                void* workspace, size_t workspace_bytes, void* stream);
int amtx_bn_relu_pool_train_bwd(const float* x, int64_t rows, int num_bins, int channels, int pool, const float* stats, const float* dy,
                                float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* LogisticBank.get_loss (amt_tools/models/common.py:541-584) forward and backward in one pass: logits (B, T, keys) fp32 with row
 * stride ld, labels (B, keys, T) fp32, weight = optional per-key weights (`OutputLayer.weights`);
 * *loss = mean_b sum_k mean_t w_k BCEWithLogits, grad (optional, (B, T, keys) contiguous) = d loss / d logits.
 * Deterministic (fixed summation order).  workspace: amtx_bce_logits_loss_workspace_bytes(batch, num_frames, keys).        */
size_t amtx_bce_logits_loss_workspace_bytes(int batch, int num_frames, int keys);
int amtx_bce_logits_loss(const float* logits, int64_t ld, const float* labels, const float* weight, int batch, int num_frames, int keys,
                         float* loss, float* grad, void* workspace, size_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Dense layers of the TRAINING step (amt_tools/train.py:126-141 drives them through autograd): fp32 in / fp32 out, every product as
 * split-bf16 ("x3", fp32-class accuracy) on the matrix cores.  No vendor BLAS / MIOpen behind any of them.
 * ------------------------------------------------------------------------------------------------ */
/* C[m][n] = sum_k A(m,k) B(n,k) (+ bias[n]);  A(m,k) = a[m*lda + k] (a_trans = 0) or a[k*lda + m] (a_trans = 1), B likewise.
 * Operands 16-byte aligned, leading dimensions and the contiguous extents multiples of 4.  workspace (optional): split-contraction
 * partials, amtx_matmul_workspace_bytes(m, n, k); used only when ldc == n. */
size_t amtx_matmul_workspace_bytes(int64_t m, int64_t n, int64_t k);
int amtx_matmul_f32(const float* a, int64_t lda, int a_trans, const float* b, int64_t ldb, int b_trans, const float* bias, float* c, int64_t ldc,
                    int64_t m, int64_t n, int64_t k, void* workspace, size_t workspace_bytes, void* stream);
/* nn.Linear in training mode: fc1 (models/onsetsframes.py:422-427), LogisticBank.output_layer (models/common.py:539), the nn.LSTM
 * input projections (onsetsframes.py:498-501).  y[m][n] = x[m][:] . w[n][:] + bias[n]. */
int amtx_linear_train_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy, int64_t m, int n,
                          int k, void* stream);
/* its backward: dx[m][k] = dy w, dw[n][k] = dy^T x (contiguous), db[n] = column sums of dy; any of dx / dw / db may be null. */
size_t amtx_linear_bwd_workspace_bytes(int64_t m, int n, int k);
int amtx_linear_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, int64_t ldw, float* dx, int64_t lddx, float* dw,
                    float* db, int64_t m, int n, int k, void* workspace, size_t workspace_bytes, void* stream);
/* nn.Conv2d(c_in, c_out, 3, padding 1) of the acoustic model's three stages in training mode (models/onsetsframes.py:375-416; the
 * BatchNorm behind it is amtx_bn_relu_pool_train_*).  Channels-last fp32: x (rows = clips x frames_per_clip, num_bins, c_in),
 * y / dy (rows, num_bins, c_out); w / dw in the reference's (c_out, c_in, 3, 3) layout; zero padding at every clip's first / last frame
 * and bin.  Implicit GEMMs (im2col on the operand fetch).  c_in not a multiple of 4 (the first layer): direct fp32 forward, weight
 * gradient through the same GEMM, no dx.  workspace: amtx_conv3x3_train_workspace_bytes(rows, num_bins, c_in, c_out). */
size_t amtx_conv3x3_train_workspace_bytes(int64_t rows, int num_bins, int c_in, int c_out);
int amtx_conv3x3_train_fwd(const float* x, const float* w, const float* bias, float* y, int64_t rows, int frames_per_clip, int num_bins, int c_in,
                           int c_out, void* workspace, size_t workspace_bytes, void* stream);
int amtx_conv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int64_t rows, int frames_per_clip,
                     int num_bins, int c_in, int c_out, void* workspace, size_t workspace_bytes, void* stream);
/* LogisticBank.finalize_output: sigmoid -> (B,keys,T) -> threshold (< 0: keep probabilities)  (models/common.py:586-620) */
int amtx_pianoroll_fwd(const float* logits, int64_t ld, int col0, int batch, int num_frames, int keys, float threshold, float* out,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * CQT / VQT / HCQT front-end.  Replaces librosa.vqt as called from amt_tools/features/vqt.py:183-193 (CQT:
 * features/cqt.py:22; HVQT / HCQT: one transform per harmonic h*fmin, features/hvqt.py:45-58,107-133) plus
 * abs -> amplitude_to_db(ref=max) per harmonic -> /80 + 1 (features/common.py:199,218-228).
 * harmonics: n_harmonics multipliers of fmin (pass {1.0} for a plain VQT/CQT).  truncate_to_expected: cut the
 * output to the reference's own frame estimate (HVQT.get_expected_frames, features/hvqt.py:60-83).
 * librosa09: 1 = librosa 0.9 conventions (alpha = 2^(1/bpo) - 1, reflect centre padding), 0 = librosa >= 0.10.
 * The 2:1 decimation between octaves is this project's documented 301-tap Kaiser low-pass (librosa's soxr
 * resampler is not bit-reproducible).  Output: [B][n_harmonics][n_bins][T] fp32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct amtx_cqt_plan amtx_cqt_plan;
int amtx_cqt_plan_create(amtx_cqt_plan** plan, int sample_rate, int hop_length, double fmin, int n_bins, int bins_per_octave,
                         double gamma, const double* harmonics, int n_harmonics, int truncate_to_expected, int librosa09);
int amtx_cqt_plan_destroy(amtx_cqt_plan* plan);
int amtx_cqt_num_harmonics(const amtx_cqt_plan* plan);
int64_t amtx_cqt_num_frames(const amtx_cqt_plan* plan, int64_t num_samples);
size_t amtx_cqt_workspace_bytes(const amtx_cqt_plan* plan, int batch, int64_t num_samples);
int amtx_cqt_forward(const amtx_cqt_plan* plan, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                     void* workspace, size_t workspace_bytes, float* out, void* stream);
/* The same transform with the map laid out for the Onsets & Frames engine's fused first conv (amtx_of_forward_feats16): out16 =
 * [B][T][n_bins][8] bf16 -- the harmonics of a (frame, bin) position in one 16-byte slot, slots n_harmonics .. 7 zero (n_harmonics <= 8).
 * The values are amtx_cqt_forward's rounded to bf16 (round to nearest even), i.e. what the conv kernel would make of them itself. */
int amtx_cqt_forward16(const amtx_cqt_plan* plan, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                       void* workspace, size_t workspace_bytes, void* out16, void* stream);
/* ... and as the TWO 16-bit planes the two-plane (x3) engine multiplies with: hi = bf16(v) at out16 (amtx_cqt_forward16's map), lo = bf16(v - hi)
 * `plane_elems` 16-bit elements behind it (>= B T n_bins 8, a multiple of 8): what amtx_of_forward_feats16 takes when amtx_of_takes_feats16() is 2. */
int amtx_cqt_forward16_split(const amtx_cqt_plan* plan, const float* audio, int64_t num_samples, int64_t audio_stride, int batch, int decibels,
                             void* workspace, size_t workspace_bytes, void* out16, int64_t plane_elems, void* stream);

/* RMS normalisation of a batch of clips: tools.rms_norm (tools/utils.py:2789-2814) as applied by
 * tools.load_normalize_audio (tools/io.py:80-82): clip / sqrt(mean(clip^2)), all-zero clips untouched. */
size_t amtx_rms_norm_workspace_bytes(int batch, int64_t num_samples);
int amtx_rms_norm(const float* audio, int64_t num_samples, int64_t audio_stride, int batch, float* out, int64_t out_stride,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Note decoding: binary piano rolls (B, keys, T) fp32 -> per (clip, key) row the list of (onset frame, offset frame)
 * events, `capacity` int32 pairs per row in DESCENDING frame order, and counts[B*keys].  onsets may be null (onsets are
 * then the positive first difference of multi_pitch).  Replaces the event walk of tools.multi_pitch_to_notes
 * (tools/utils.py:369-471) used by transcribe.NoteTranscriber (transcribe.py:420-481,722-763). */
int amtx_notes_decode(const float* onsets, const float* multi_pitch, int batch, int keys, int num_frames, int capacity,
                      int32_t* pairs, int32_t* counts, void* stream);

/* amtx_notes_decode's event lists -> ONE dense array of note rows [onset_s, offset_s, midi_pitch] (float64; the reference's batched
 * notes, tools/utils.py:135-165) in np.nonzero order per clip (key ascending, onset frame ascending = the order in which
 * tools.multi_pitch_to_notes appends, utils.py:445-463), the frames converted to seconds through the clip's time grid:
 *   times_ext   [num_frames + 1] float64 (times_stride 0: one grid for the batch) or [batch][times_stride] -- the frame times plus one
 *               more frame (utils.py:441-442); the caller builds it (estimate_hop_length is host arithmetic on a small array)
 *   rows        [rows_capacity][3], onset_col [rows_capacity] (a contiguous copy of column 0: the key of sort_notes' argsort)
 *   clip_offsets[batch + 1] int32: clip b's rows are [clip_offsets[b], clip_offsets[b + 1]); clip_offsets[batch] is the total, which may
 *               EXCEED rows_capacity -- rows past the capacity are not written, the caller retries with a larger buffer.
 * The reference's final row order (three unstable argsorts by onset, utils.py:2713-2746) is applied by the caller: the order among
 * equal onsets is whatever NumPy's sort makes of it and is not reproduced on the device. */
int amtx_notes_rows(const int32_t* pairs, const int32_t* counts, int batch, int keys, int capacity, const double* times_ext,
                    int64_t times_stride, int low_pitch, double* rows, double* onset_col, int64_t rows_capacity, int32_t* clip_offsets,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMTX_H */
