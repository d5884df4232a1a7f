// logits -> piano roll for gfx950.
// Replaces LogisticBank.finalize_output (amt_tools/models/common.py:586-620): sigmoid -> transpose to
// (B, keys, T) -> threshold_activations (amt_tools/tools/utils.py:2896-2919: x < thr -> 0, else 1).
// HBM-bound byte work: reads keys*4 B and writes keys*4 B per frame; 32x32 LDS tile transpose so both
// the [B][T][ld] reads and the [B][keys][T] writes are coalesced.

#include "amtx_kernels.h"

#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void pianoroll_kernel(const float* __restrict__ logits, int64_t ld, int col0, int T, int keys,
                                                        float threshold, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (t < T && k < keys) {
            const float x = logits[((int64_t)b * T + t) * ld + col0 + k];
            const float s = 1.0f / (1.0f + expf(-x));            // torch.sigmoid in fp32
            v = threshold < 0.f ? s : (s < threshold ? 0.f : 1.f);
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, t = t0 + tx;
        if (t < T && k < keys) out[((int64_t)b * keys + k) * T + t] = tile[tx][ty + 8 * i];
    }
}

// fp32 rows [M][ld_src] (n_src valid columns) -> bf16 rows [M][ld_dst], zero padded: lets the adjoin input projection
// (A = joint logits) run on the direct-to-LDS bf16 GEMM path.
typedef __attribute__((ext_vector_type(2))) _Float16 head_f16x2;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {       // the AMTX_PREC_F16 engine: IEEE half instead of bf16
    const amtx_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, head_f16x2));
}
template <bool F16>
__global__ __launch_bounds__(256) void cvt_pad_bf16_kernel(const float* __restrict__ src, int64_t ld_src, int n_src, bf16_t* __restrict__ dst,
                                                           int ld_dst, int64_t rows) {
    const int groups = ld_dst >> 2;                                   // 4 columns per thread
    const int64_t total = rows * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / groups;
        const int c = (int)(i - r * groups) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < n_src) v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
        *reinterpret_cast<uint2*>(dst + r * ld_dst + c) = F16 ? make_uint2(pack_f16x2(v.x, v.y), pack_f16x2(v.z, v.w))
                                                              : make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

// fp32 rows -> the two 16-bit planes of AMTX_T_SPLIT (hi = 16-bit(x), lo = 16-bit(x - hi): split_bf16x2, the conversion every two-plane
// kernel applies to its fp32 operands), columns n_src .. ld_dst zero: the refinement stage's input projection reads the joint logits
// through the direct-to-LDS two-plane GEMM, whose stages are 32 columns deep
__global__ __launch_bounds__(256) void cvt_split_kernel(const float* __restrict__ src, int64_t ld_src, int n_src, bf16_t* __restrict__ dst,
                                                        int ld_dst, int64_t split, int64_t rows) {
    const int groups = ld_dst >> 2;                                   // 4 columns per thread
    const int64_t total = rows * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / groups;
        const int c = (int)(i - r * groups) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < n_src) v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
        uint32_t h0, h1, l0, l1;
        split_bf16x2(v.x, v.y, h0, l0);
        split_bf16x2(v.z, v.w, h1, l1);
        *reinterpret_cast<uint2*>(dst + r * ld_dst + c) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + split + r * ld_dst + c) = make_uint2(l0, l1);
    }
}

}  // namespace

// zero `width` bytes (a multiple of 16) at the start of each of `rows` rows that are `pitch` bytes apart
__global__ __launch_bounds__(256) void zero_cols_kernel(char* __restrict__ base, int64_t pitch, int w16, int64_t rows) {
    const int64_t n = rows * w16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        *reinterpret_cast<uint4*>(base + (i / w16) * pitch + (i % w16) * 16) = make_uint4(0, 0, 0, 0);
}

int amtx_launch_zero_cols(void* base, int64_t pitch_bytes, int width_bytes, int64_t rows, hipStream_t stream) {
    AMTX_REQUIRE(base && width_bytes > 0 && width_bytes % 16 == 0 && pitch_bytes % 16 == 0 && ((uintptr_t)base % 16) == 0 && rows > 0,
                 "zero_cols: bad argument");
    const int64_t n = rows * (width_bytes / 16);
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(zero_cols_kernel, dim3(blocks), dim3(256), 0, stream, (char*)base, pitch_bytes, width_bytes / 16, rows);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_launch_cvt_pad_bf16(const float* src, int64_t ld_src, int n_src, bf16_t* dst, int ld_dst, int64_t rows, hipStream_t stream, bool f16) {
    AMTX_REQUIRE(src && dst && rows > 0 && n_src % 4 == 0 && ld_dst % 4 == 0 && ld_src % 4 == 0 && ld_dst >= n_src, "cvt_pad_bf16: bad argument");
    int64_t nb = (rows * (ld_dst >> 2) + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (f16) hipLaunchKernelGGL(cvt_pad_bf16_kernel<true>, dim3((unsigned)nb), dim3(256), 0, stream, src, ld_src, n_src, dst, ld_dst, rows);
    else hipLaunchKernelGGL(cvt_pad_bf16_kernel<false>, dim3((unsigned)nb), dim3(256), 0, stream, src, ld_src, n_src, dst, ld_dst, rows);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_launch_cvt_split(const float* src, int64_t ld_src, int n_src, bf16_t* dst, int ld_dst, int64_t split, int64_t rows, hipStream_t stream) {
    AMTX_REQUIRE(src && dst && rows > 0 && n_src % 4 == 0 && ld_dst % 4 == 0 && ld_src % 4 == 0 && ld_dst >= n_src && split % 4 == 0 &&
                     split >= rows * ld_dst, "cvt_split: bad argument");
    int64_t nb = (rows * (ld_dst >> 2) + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cvt_split_kernel, dim3((unsigned)nb), dim3(256), 0, stream, src, ld_src, n_src, dst, ld_dst, split, rows);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_launch_pianoroll(const float* logits, int64_t ld, int col0, int B, int T, int keys, float threshold, float* out,
                          hipStream_t stream) {
    AMTX_REQUIRE(logits && out, "pianoroll: null pointer");
    AMTX_REQUIRE(B > 0 && B < 65536 && T > 0 && keys > 0, "pianoroll: bad sizes");
    dim3 grid((unsigned)((T + 31) / 32), (unsigned)((keys + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(pianoroll_kernel, grid, dim3(256), 0, stream, logits, ld, col0, T, keys, threshold, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// LogisticBank.get_loss (amt_tools/models/common.py:541-584) and its gradient in one pass:
//   loss = mean_b sum_k mean_t  w_k * BCEWithLogits(x[b,t,k], y[b,k,t])          (x: (B,T,K) logits, y: (B,K,T) labels)
//   grad[b,t,k] = w_k * (sigmoid(x) - y) / (B*T)                                 (d loss / d x)
// with torch's stable form max(x,0) - x*y + log1p(exp(-|x|)).  The labels arrive key-major, the logits frame-major: a 32x32
// LDS transpose keeps both streams coalesced.  Deterministic: one partial per block, summed in block order by one wave.
namespace {

__global__ __launch_bounds__(256) void bce_loss_kernel(const float* __restrict__ logits, int64_t ld, const float* __restrict__ labels,
                                                       const float* __restrict__ weight, int T, int keys, float inv_bt,
                                                       float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, t = t0 + tx;
        tile[ty + 8 * i][tx] = (t < T && k < keys) ? labels[((int64_t)b * keys + k) * T + t] : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, k = k0 + tx;
        if (t < T && k < keys) {
            const float x = logits[((int64_t)b * T + t) * ld + k];
            const float y = tile[tx][ty + 8 * i];
            const float w = weight ? weight[k] : 1.0f;
            const float e = expf(-fabsf(x));
            acc += w * (fmaxf(x, 0.f) - x * y + log1pf(e));
            if (grad) {
                const float sg = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
                grad[((int64_t)b * T + t) * keys + k] = w * (sg - y) * inv_bt;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void bce_loss_reduce_kernel(const float* __restrict__ partial, int64_t n, float inv_bt, float* __restrict__ loss) {
    // fixed order: lane l sums partial[l], partial[l+64], ... in double, then a fixed butterfly
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 64) a += (double)partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (threadIdx.x == 0) *loss = (float)(a * (double)inv_bt);
}

}  // namespace

size_t amtx_bce_loss_partials(int B, int T, int keys) { return (size_t)B * ((T + 31) / 32) * ((keys + 31) / 32); }

int amtx_launch_bce_loss(const float* logits, int64_t ld, const float* labels, const float* weight, int B, int T, int keys, float* loss,
                         float* grad, float* partial, hipStream_t stream) {
    AMTX_REQUIRE(logits && labels && loss && partial, "bce_loss: null pointer");
    AMTX_REQUIRE(B > 0 && B < 65536 && T > 0 && keys > 0 && ld >= keys, "bce_loss: bad sizes");
    const float inv_bt = 1.0f / ((float)B * (float)T);
    dim3 grid((unsigned)((T + 31) / 32), (unsigned)((keys + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(bce_loss_kernel, grid, dim3(256), 0, stream, logits, ld, labels, weight, T, keys, inv_bt, grad, partial);
    AMTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_loss_reduce_kernel, dim3(1), dim3(64), 0, stream, partial, (int64_t)amtx_bce_loss_partials(B, T, keys), inv_bt, loss);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Root-mean-square normalisation of a batch of clips on the device: tools.rms_norm (amt_tools/tools/utils.py:2789-2814,
// applied by tools.load_normalize_audio, tools/io.py:80-82): audio / sqrt(mean(audio^2)), untouched when the clip is
// all zeros.  Deterministic two-level reduction (fixed summation order), then a division like the reference's.
namespace {

constexpr int RMS_CHUNK = 8192;

__global__ __launch_bounds__(256) void rms_partial_kernel(const float* __restrict__ audio, int64_t n, int64_t stride, int nchunks,
                                                          float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y, ch = blockIdx.x;
    const float* src = audio + (int64_t)b * stride;
    const int64_t i0 = (int64_t)ch * RMS_CHUNK;
    float s = 0.f;
    for (int i = threadIdx.x; i < RMS_CHUNK; i += 256) {
        const int64_t j = i0 + i;
        const float v = j < n ? src[j] : 0.f;
        s = fmaf(v, v, s);
    }
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * nchunks + ch] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void rms_final_kernel(const float* __restrict__ partial, int nchunks, int64_t n, float* __restrict__ rms) {
    const int b = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += 64) s += partial[(int64_t)b * nchunks + i];
    s = wave_sum_f32(s);
    if (threadIdx.x == 0) rms[b] = sqrtf(s / (float)n);
}

__global__ __launch_bounds__(256) void rms_scale_kernel(const float* __restrict__ audio, int64_t n, int64_t stride, const float* __restrict__ rms,
                                                        float* __restrict__ out, int64_t out_stride) {
    const int b = blockIdx.y;
    const float r = rms[b];
    const float* src = audio + (int64_t)b * stride;
    float* dst = out + (int64_t)b * out_stride;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = r > 0.f ? src[i] / r : src[i];
}

}  // namespace

extern "C" size_t amtx_rms_norm_workspace_bytes(int batch, int64_t num_samples) {
    if (batch <= 0 || num_samples <= 0) return 0;
    const int64_t nchunks = (num_samples + RMS_CHUNK - 1) / RMS_CHUNK;
    return (size_t)batch * (nchunks + 1) * sizeof(float);
}

extern "C" int amtx_rms_norm(const float* audio, int64_t num_samples, int64_t audio_stride, int batch, float* out, int64_t out_stride,
                             void* workspace, size_t workspace_bytes, void* stream_) {
    AMTX_REQUIRE(audio && out && workspace, "amtx_rms_norm: null pointer");
    AMTX_REQUIRE(batch > 0 && batch < 65536 && num_samples > 0 && audio_stride >= num_samples && out_stride >= num_samples, "amtx_rms_norm: bad sizes");
    AMTX_REQUIRE(workspace_bytes >= amtx_rms_norm_workspace_bytes(batch, num_samples), "amtx_rms_norm: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    const int nchunks = (int)((num_samples + RMS_CHUNK - 1) / RMS_CHUNK);
    float* partial = (float*)workspace;
    float* rms = partial + (size_t)batch * nchunks;
    hipLaunchKernelGGL(rms_partial_kernel, dim3(nchunks, batch), dim3(256), 0, s, audio, num_samples, audio_stride, nchunks, partial);
    AMTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(rms_final_kernel, dim3(batch), dim3(64), 0, s, (const float*)partial, nchunks, num_samples, rms);
    AMTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(rms_scale_kernel, dim3(64, batch), dim3(256), 0, s, audio, num_samples, audio_stride, (const float*)rms, out, out_stride);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}
