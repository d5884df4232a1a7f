// Device-side weight packing for a weight RE-SYNC of the Onsets & Frames engine (amtx_of_model_finalize_device): the same folding and
// fragment layouts as the host packers (conv.hip amtx_conv1_pack_host / amtx_conv3x3_pack_host, gemm.hip amtx_gemm_pack_host, ofmodel.hip
// fold_bn and the fp64 fold of the pitch head's two Linear layers), element for element and in the same arithmetic (double where the host
// uses double, the same summation order), so that a model synced on the device holds the SAME BITS as one synced through the host
// (tests/test_gpu_model.py::test_device_side_weight_sync_equals_the_host_path).  Why: validate() inside train()
// (amt_tools/train.py:183-189) re-syncs the engine at every checkpoint; through the host that is a device-to-host copy of every tensor,
// ~164 M double multiply-adds and the packing loops on one core, and the upload -- 30 - 60 ms; here it is a handful of small kernels.
// Compiled twice like the kernels that read the packed weights (amtx_f16_names.h): the 16-bit format is the build's.

#include "amtx_f16_names.h"
#include "amtx_kernels.h"

#include <algorithm>

namespace {

__device__ __forceinline__ void put16(bf16_t* out, size_t hi_index, size_t plane_stride, int planes, float v) {
    // v is often a product (weight x BatchNorm scale).  In the half-precision build hipcc would fuse multiply and conversion into one
    // v_fma_mixlo_f16, i.e. round the exact product ONCE to half, where the host packers round it to fp32 first: rare last-bit
    // differences between a device-synced and a host-synced model (seen as 3e-5 ... 4e-4 on f16 logits).  Keep the fp32 product.
    asm volatile("" : "+v"(v));
    const bf16_t hi = f32_to_bf16_rn(v);
    out[hi_index] = hi;
    if (planes == 2) out[hi_index + plane_stride] = f32_to_bf16_rn(v - bf16_to_f32(hi));
}

// ofmodel.hip fold_bn: scale = gamma / sqrt(var + 1e-5), shift = beta + (conv_bias - mean) * scale, in double, rounded once
__global__ void bn_fold_kernel(const float* cb, const float* g, const float* be, const float* mu, const float* var, int c_out, float* scale, float* shift) {
#pragma clang fp contract(off)      // the host compiler does not fuse these either: same bits
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= c_out) return;
    const double s = (double)g[c] / sqrt((double)var[c] + 1e-5);
    scale[c] = (float)s;
    shift[c] = (float)((double)be[c] + ((double)cb[c] - (double)mu[c]) * s);
}

// conv.hip amtx_conv3x3_pack_host: [tap][nt][plane][lane][8]
__global__ void conv3x3_pack_kernel(const float* w, const float* scale, int c_out, int planes, bf16_t* out) {
    const int NT = c_out / 16;
    const int total = 9 * NT * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int nt = r % NT;
        const int tap = r / NT;
        const int row = l & 15;
        const int co = (row >> 2) * (4 * NT) + 4 * nt + (row & 3);
        const int ci = (l >> 4) * 8 + j;
        const float v = w[((size_t)co * 32 + ci) * 9 + tap] * (scale ? scale[co] : 1.0f);
        put16(out, ((size_t)(tap * NT + nt) * planes) * 512 + (size_t)l * 8 + j, 512, planes, v);
    }
}

// conv.hip amtx_conv1_pack_host, c_in = 1 (Toeplitz fragments): [q][nt][plane][lane][8]
__global__ void conv1_pack_kernel(const float* w, const float* scale, int planes, bf16_t* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * 2 * 64 * 8) return;
    int r = idx;
    const int e = r & 7; r >>= 3;
    const int l = r & 63; r >>= 6;
    const int nt = r & 1;
    const int q = r >> 1;
    const int row = l & 15, g = l >> 4;
    const int co = (row >> 2) * 8 + 4 * nt + (row & 3);
    const int kw = e - q;
    const float v = (g < 3 && kw >= 0 && kw <= 2) ? w[(size_t)co * 9 + g * 3 + kw] * (scale ? scale[co] : 1.0f) : 0.0f;
    put16(out, ((size_t)(q * 2 + nt) * planes) * 512 + (size_t)l * 8 + e, 512, planes, v);
}

// gemm.hip amtx_gemm_pack_host into rows [row0, row0 + N) of a [planes][n_pad][k_pad] matrix, zero padding included for the rows it owns
// (rows_owned >= N: the caller that packs the last row block passes the pad rows too).  perm_c > 0: column k of the packed matrix is
// column (k % perm_c) * perm_f + k / perm_c of W -- fc1's (channel, freq) -> (freq, channel) permutation of ofmodel.hip.
__global__ void linear_pack_kernel(const float* W, int64_t ldw, int N, int K, int planes, int n_pad, int k_pad, int row0, int rows_owned, int perm_c,
                                   int perm_f, bf16_t* out) {
    const int64_t total = (int64_t)rows_owned * k_pad;
    const size_t plane = (size_t)n_pad * k_pad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / k_pad), k = (int)(idx - (int64_t)n * k_pad);
        float v = 0.0f;
        if (n < N && k < K) {
            const int ks = perm_c > 0 ? (k % perm_c) * perm_f + k / perm_c : k;
            v = W[(int64_t)n * ldw + ks];
        }
        put16(out, (size_t)(row0 + n) * k_pad + k, plane, planes, v);
    }
}

// ofmodel.hip: (W_out . W_fc1) and W_out . b_fc1 + b_out in double, j ascending -- the host's summation order.  Output column k is in the
// engine's (freq, channel) order, W_fc1's columns in the reference's (channel, freq) order; columns kfc .. kfc_pad are zero.
__global__ void head_fold_kernel(const float* w_out, const float* w_fc1, const float* b_fc1, const float* b_out, int n_out, int dim_am, int kfc,
                                 int kfc_pad, int nf3, int fq, float* wfold, float* bfold) {
#pragma clang fp contract(off)
    const int64_t total = (int64_t)n_out * kfc_pad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx / kfc_pad), k = (int)(idx - (int64_t)o * kfc_pad);
        double acc = 0.0;
        if (k < kfc) {
            const int ks = (k % nf3) * fq + k / nf3;
            for (int j = 0; j < dim_am; ++j) acc += (double)w_out[(size_t)o * dim_am + j] * (double)w_fc1[(size_t)j * kfc + ks];
        }
        wfold[idx] = (float)acc;
    }
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n_out) {
        double acc = b_out[o];
        for (int j = 0; j < dim_am; ++j) acc += (double)w_out[(size_t)o * dim_am + j] * (double)b_fc1[j];
        bfold[o] = (float)acc;
    }
}

// convg.hip amtx_conv3x3_gen_pack_host: [chunk][fragment][plane][lane][8]; fragments of a chunk: the full 32-deep steps in (tap, tile,
// step) order, then (C_in with a 16-channel tail) the paired tails A (kw, tile): taps (0,kw) | (1,kw); B (tile): (2,0) | (2,1); C (tile): (2,2) | 0.
// One thread per (chunk, tile, lane, j).
__global__ void conv_gen_pack_kernel(const float* w, const float* scale, int c_in, int c_out, int ntc, int planes, bf16_t* out) {
    const int ci16 = c_in / 16, n32 = ci16 / 2, n16 = ci16 % 2;
    const int nmain = 9 * ntc * n32, nfrag = nmain + 5 * ntc * n16;
    const int nchunks = c_out / (16 * ntc);
    const int total = nchunks * ntc * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int nt = r % ntc;
        const int ch = r / ntc;
        const int row = l & 15, gq = l >> 4;
        const int co = ch * 16 * ntc + (row >> 2) * (4 * ntc) + 4 * nt + (row & 3);
        const float sc = scale ? scale[co] : 1.0f;
        bf16_t* cbase = out + (size_t)ch * nfrag * planes * 512;
        auto wv = [&](int ci, int tap) { return w[((size_t)co * c_in + ci) * 9 + tap] * sc; };
        auto put = [&](int frag, float v) { put16(cbase, (size_t)frag * planes * 512 + (size_t)l * 8 + j, 512, planes, v); };
        for (int tap = 0; tap < 9; ++tap)
            for (int ks = 0; ks < n32; ++ks) put((tap * ntc + nt) * n32 + ks, wv(32 * ks + 8 * gq + j, tap));
        if (!n16) continue;
        const int ct = 32 * n32 + 8 * (gq & 1);
        for (int kw = 0; kw < 3; ++kw) put(nmain + kw * ntc + nt, wv(ct + j, (gq < 2 ? 0 : 3) + kw));
        put(nmain + 3 * ntc + nt, wv(ct + j, gq < 2 ? 6 : 7));
        put(nmain + 4 * ntc + nt, gq < 2 ? wv(ct + j, 8) : 0.0f);
    }
}

// convg.hip amtx_conv1g_pack_host: [tile of 16 channels][k-step][plane][lane][8], k = 32 ks + 8 (lane >> 4) + j over (ci, kh, kw), zero past 9 c_in
__global__ void conv1g_pack_kernel(const float* w, const float* scale, int c_in, int c_mid, int planes, bf16_t* out) {
    if (amtx_conv1g_tapk(c_in, planes)) {          // tap-major, 8 channel slots per tap, three steps (amtx_conv1g_pack_host's other branch)
        const int total = (c_mid / 16) * 3 * 64 * 8;
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
            int r = idx;
            const int j = r & 7; r >>= 3;
            const int l = r & 63; r >>= 6;
            const int ks = r % 3, nt = r / 3;
            const int co = 16 * nt + (l & 15), tap = 4 * ks + (l >> 4);
            const float v = (tap < 9 && j < c_in) ? w[((size_t)co * c_in + j) * 9 + tap] * (scale ? scale[co] : 1.0f) : 0.0f;
            put16(out, ((size_t)(nt * 3 + ks) * planes) * 512 + (size_t)l * 8 + j, 512, planes, v);
        }
        return;
    }
    const int kvalid = 9 * c_in, ks1 = (kvalid + 31) / 32;
    const int total = (c_mid / 16) * ks1 * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int ks = r % ks1;
        const int nt = r / ks1;
        const int co = 16 * nt + (l & 15), k = 32 * ks + 8 * (l >> 4) + j;
        const float v = k < kvalid ? w[(size_t)co * kvalid + k] * (scale ? scale[co] : 1.0f) : 0.0f;
        put16(out, ((size_t)(nt * ks1 + ks) * planes) * 512 + (size_t)l * 8 + j, 512, planes, v);
    }
}

// ofmodel.hip: the fp32 weights of the unfused first convolution, out[row][i] = w[row][i] * scale[row]
__global__ void scale_rows_kernel(const float* w, const float* scale, int rows, int cols, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < rows * cols) out[idx] = w[idx] * scale[idx / cols];
}

__global__ void vec_add_kernel(const float* a, const float* b, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

}  // namespace

int amtx_pack_bn_fold_dev(const float* conv_bias, const float* gamma, const float* beta, const float* mean, const float* var, int c_out, float* scale,
                          float* shift, hipStream_t s) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((c_out + 63) / 64), dim3(64), 0, s, conv_bias, gamma, beta, mean, var, c_out, scale, shift);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_conv3x3_dev(const float* w, const float* scale, int c_out, int planes, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(64), dim3(256), 0, s, w, scale, c_out, planes, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_conv_gen_dev(const float* w, const float* scale, int c_in, int c_out, int ntc, int planes, bf16_t* out, hipStream_t s) {
    AMTX_REQUIRE(ntc > 0 && c_in % 16 == 0 && c_out % (16 * ntc) == 0, "conv pack (general): bad channel counts %d -> %d", c_in, c_out);
    hipLaunchKernelGGL(conv_gen_pack_kernel, dim3(32), dim3(256), 0, s, w, scale, c_in, c_out, ntc, planes, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_conv1g_dev(const float* w, const float* scale, int c_in, int c_mid, int planes, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(conv1g_pack_kernel, dim3(16), dim3(256), 0, s, w, scale, c_in, c_mid, planes, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_scale_rows_dev(const float* w, const float* scale, int rows, int cols, float* out, hipStream_t s) {
    hipLaunchKernelGGL(scale_rows_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, s, w, scale, rows, cols, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_conv1_dev(const float* w, const float* scale, int planes, bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(conv1_pack_kernel, dim3(16), dim3(256), 0, s, w, scale, planes, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_linear_dev(const float* W, int64_t ldw, int N, int K, int planes, int n_pad, int k_pad, int row0, int rows_owned, int perm_c, int perm_f,
                         bf16_t* out, hipStream_t s) {
    const int64_t total = (int64_t)rows_owned * k_pad;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(linear_pack_kernel, dim3(blocks), dim3(256), 0, s, W, ldw, N, K, planes, n_pad, k_pad, row0, rows_owned, perm_c, perm_f, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_head_fold_dev(const float* w_out, const float* w_fc1, const float* b_fc1, const float* b_out, int n_out, int dim_am, int kfc, int kfc_pad,
                            int nf3, int fq, float* wfold, float* bfold, hipStream_t s) {
    const int64_t total = (int64_t)n_out * kfc_pad;
    hipLaunchKernelGGL(head_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_out, w_fc1, b_fc1, b_out, n_out, dim_am, kfc, kfc_pad,
                       nf3, fq, wfold, bfold);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

int amtx_pack_vec_add_dev(const float* a, const float* b, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(vec_add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, out);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}
