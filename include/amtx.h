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

#ifdef __cplusplus
}
#endif
#endif /* AMTX_H */
