// Training-mode BatchNorm2d (batch statistics) + ReLU (+ MaxPool2d((1,2))) forward and backward for the convolution stages of
// the acoustic model (amt_tools/models/onsetsframes.py:375-416, driven by amt_tools/train.py:126-141 through autograd).
// On ROCm the stock path is MIOpen BatchNorm + separate ReLU / max-pool elementwise kernels and their backward kernels over fp32
// maps (5 of the 14 ms of a training step at 8 clips x 625 frames); here each direction is two passes over the conv output:
//   forward : (1) per-channel sum / sum of squares (fp32 per thread, fp64 across partials: deterministic, no atomics)
//             (2) y = max over the pooling pair of relu(x * scale + shift), running statistics updated like nn.BatchNorm2d
//   backward: (1) dz = dy routed to the pair's winner where it is positive; per-channel sum(dz), sum(dz * xhat)
//             (2) dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat))
// Layout: channels-last fp32, x [P][C] with P = rows * F positions (rows = B * T), y [rows][F / 2 or F][C].  HBM-bound:
// forward reads x twice and writes y, backward reads x and dy twice and writes dx.

#include "amtx_kernels.h"

#include <algorithm>

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_PARTIALS = 1024;

struct BnArgs {
    const float* x; int64_t rows; int F, C;       // x [rows][F][C]
    int pool;                                     // 1: MaxPool(1,2) over F (floor), 0: none
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- forward pass 1: partial[blk][2][C] = sum, sum of squares over the block's positions
__global__ __launch_bounds__(BN_THREADS) void bn_stats_partial_kernel(const float* __restrict__ x, int64_t P, int C, float* __restrict__ partial) {
    extern __shared__ float red[];                // [2][BN_THREADS][4]
    const int c4n = C >> 2;                       // float4 groups per position
    const int tid = threadIdx.x;
    const int cg = tid % c4n, pr = tid / c4n, prn = BN_THREADS / c4n;
    float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
    if (pr < prn)
        for (int64_t p = (int64_t)blockIdx.x * prn + pr; p < P; p += (int64_t)gridDim.x * prn) {
            const float4 v = *reinterpret_cast<const float4*>(x + p * C + 4 * cg);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
        }
    float4* rs = reinterpret_cast<float4*>(red);
    rs[tid] = s; rs[BN_THREADS + tid] = q;
    __syncthreads();
    if (tid < c4n) {                              // fixed order: deterministic
        float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
        for (int r = 0; r < prn; ++r) {
            const float4 u = rs[r * c4n + tid], w = rs[BN_THREADS + r * c4n + tid];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        float* out = partial + (int64_t)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(out + 4 * tid) = a;
        *reinterpret_cast<float4*>(out + C + 4 * tid) = b;
    }
}

// ---- forward pass 1b: mean / invstd, fused affine (scale, shift), running statistics (momentum < 0: cumulative average is the caller's job)
__global__ void bn_stats_final_kernel(const float* __restrict__ partial, int nblk, int C, double n, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                      float* __restrict__ running_var, float* __restrict__ stats /*[4][C]: mean, invstd, gamma, beta*/) {
    // one wave per channel: lanes stride over the partials, fixed-order butterfly in fp64 (a single thread walking 1024 partials
    // was 0.3 ms of pure load latency per layer)
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { s += partial[(int64_t)b * 2 * C + c]; q += partial[(int64_t)b * 2 * C + C + c]; }
    s = wave_sum_f64(s); q = wave_sum_f64(q);
    if (threadIdx.x != 0) return;
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)invstd;
    stats[2 * C + c] = (float)g;
    stats[3 * C + c] = (float)bt;
    if (running_mean && running_var) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// The normalised pre-activation in nn.BatchNorm2d's own operation order ((x - mean) * invstd) * gamma + beta, not as one fused
// multiply-add with a folded scale / shift: ReLU's gradient is decided by the sign of this value, and an ulp of difference flips it
// for the rare element that sits at zero (seen as a 2 % change of one channel's beta gradient against the stock path).
__device__ __forceinline__ float bnz(float x, float mean, float inv, float gamma, float beta) { return ((x - mean) * inv) * gamma + beta; }

// ---- forward pass 2
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(BnArgs a, const float* __restrict__ stats, float* __restrict__ y) {
    const int c4n = a.C >> 2, Fo = a.pool ? a.F >> 1 : a.F;
    const int64_t total = a.rows * Fo * c4n;
    for (int64_t i = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * BN_THREADS) {
        const int cg = (int)(i % c4n);
        const int64_t po = i / c4n;
        const int fo = (int)(po % Fo);
        const int64_t row = po / Fo;
        const float4 mu = *reinterpret_cast<const float4*>(stats + 4 * cg), iv = *reinterpret_cast<const float4*>(stats + a.C + 4 * cg);
        const float4 sc = *reinterpret_cast<const float4*>(stats + 2 * a.C + 4 * cg), sh = *reinterpret_cast<const float4*>(stats + 3 * a.C + 4 * cg);
        const float* xp = a.x + ((row * a.F + (a.pool ? 2 * fo : fo)) * a.C + 4 * cg);
        float4 v = *reinterpret_cast<const float4*>(xp);
        float4 z = make_float4(bnz(v.x, mu.x, iv.x, sc.x, sh.x), bnz(v.y, mu.y, iv.y, sc.y, sh.y), bnz(v.z, mu.z, iv.z, sc.z, sh.z), bnz(v.w, mu.w, iv.w, sc.w, sh.w));
        if (a.pool) {
            v = *reinterpret_cast<const float4*>(xp + a.C);
            z.x = fmaxf(z.x, bnz(v.x, mu.x, iv.x, sc.x, sh.x)); z.y = fmaxf(z.y, bnz(v.y, mu.y, iv.y, sc.y, sh.y));
            z.z = fmaxf(z.z, bnz(v.z, mu.z, iv.z, sc.z, sh.z)); z.w = fmaxf(z.w, bnz(v.w, mu.w, iv.w, sc.w, sh.w));
        }
        *reinterpret_cast<float4*>(y + po * a.C + 4 * cg) = make_float4(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f));
    }
}

// dz of the two members of a pooling pair (or of one element without pooling): the gradient goes to the larger pre-activation
// (the first on a tie, like nn.MaxPool2d's argmax) where it is positive
__device__ __forceinline__ void route(float z0, float z1, float dy, bool pool, float& d0, float& d1) {
    if (!pool) { d0 = z0 > 0.f ? dy : 0.f; d1 = 0.f; return; }
    const bool first = z0 >= z1;
    const float zm = first ? z0 : z1;
    const float d = zm > 0.f ? dy : 0.f;
    d0 = first ? d : 0.f;
    d1 = first ? 0.f : d;
}

// ---- backward pass 1: partial[blk][2][C] = sum(dz), sum(dz * xhat)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(BnArgs a, const float* __restrict__ stats, const float* __restrict__ dy,
                                                                   float* __restrict__ partial) {
    extern __shared__ float red[];
    const int C = a.C, c4n = C >> 2, Fo = a.pool ? a.F >> 1 : a.F;
    const int tid = threadIdx.x;
    const int cg = tid % c4n, pr = tid / c4n, prn = BN_THREADS / c4n;
    const int64_t Po = a.rows * Fo;
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (pr < prn) {
        const float4 mean = *reinterpret_cast<const float4*>(stats + 4 * cg), inv = *reinterpret_cast<const float4*>(stats + C + 4 * cg);
        const float4 sc = *reinterpret_cast<const float4*>(stats + 2 * C + 4 * cg), sh = *reinterpret_cast<const float4*>(stats + 3 * C + 4 * cg);
        const float m_[4] = {mean.x, mean.y, mean.z, mean.w}, i_[4] = {inv.x, inv.y, inv.z, inv.w};
        const float sc_[4] = {sc.x, sc.y, sc.z, sc.w}, sh_[4] = {sh.x, sh.y, sh.z, sh.w};
        for (int64_t po = (int64_t)blockIdx.x * prn + pr; po < Po; po += (int64_t)gridDim.x * prn) {
            const int fo = (int)(po % Fo);
            const int64_t row = po / Fo;
            const float* xp = a.x + ((row * a.F + (a.pool ? 2 * fo : fo)) * C + 4 * cg);
            const float4 x0 = *reinterpret_cast<const float4*>(xp);
            const float4 x1 = a.pool ? *reinterpret_cast<const float4*>(xp + C) : x0;
            const float4 g = *reinterpret_cast<const float4*>(dy + po * C + 4 * cg);
            const float x0_[4] = {x0.x, x0.y, x0.z, x0.w}, x1_[4] = {x1.x, x1.y, x1.z, x1.w}, g_[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d0, d1;
                route(bnz(x0_[k], m_[k], i_[k], sc_[k], sh_[k]), bnz(x1_[k], m_[k], i_[k], sc_[k], sh_[k]), g_[k], a.pool != 0, d0, d1);
                s[k] += d0 + d1;
                q[k] = fmaf(d0, (x0_[k] - m_[k]) * i_[k], q[k]);
                q[k] = fmaf(d1, (x1_[k] - m_[k]) * i_[k], q[k]);
            }
        }
    }
    float4* rs = reinterpret_cast<float4*>(red);
    rs[tid] = make_float4(s[0], s[1], s[2], s[3]); rs[BN_THREADS + tid] = make_float4(q[0], q[1], q[2], q[3]);
    __syncthreads();
    if (tid < c4n) {
        float4 u = make_float4(0, 0, 0, 0), w = make_float4(0, 0, 0, 0);
        for (int r = 0; r < prn; ++r) {
            const float4 a_ = rs[r * c4n + tid], b_ = rs[BN_THREADS + r * c4n + tid];
            u.x += a_.x; u.y += a_.y; u.z += a_.z; u.w += a_.w;
            w.x += b_.x; w.y += b_.y; w.z += b_.z; w.w += b_.w;
        }
        float* out = partial + (int64_t)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(out + 4 * tid) = u;
        *reinterpret_cast<float4*>(out + C + 4 * tid) = w;
    }
}

// ---- backward pass 1b: dbeta, dgamma and the two per-channel constants of the dx pass
__global__ void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, int C, double n, const float* __restrict__ stats,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef /*[2][C]*/) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { s += partial[(int64_t)b * 2 * C + c]; q += partial[(int64_t)b * 2 * C + C + c]; }
    s = wave_sum_f64(s); q = wave_sum_f64(q);
    if (threadIdx.x != 0) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    coef[c] = (float)(s / n);
    coef[C + c] = (float)(q / n);
}

// ---- backward pass 2: dx over every input position (the odd last column of a pooled map gets the mean terms only)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_dx_kernel(BnArgs a, const float* __restrict__ stats, const float* __restrict__ coef,
                                                              const float* __restrict__ dy, float* __restrict__ dx) {
    const int C = a.C, c4n = C >> 2, Fo = a.pool ? a.F >> 1 : a.F;
    const int Fp = a.pool ? (a.F + 1) >> 1 : a.F;              // pairs per row incl. the unpaired last column
    const int64_t total = a.rows * Fp * c4n;
    for (int64_t i = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * BN_THREADS) {
        const int cg = (int)(i % c4n);
        const int64_t pp = i / c4n;
        const int fp = (int)(pp % Fp);
        const int64_t row = pp / Fp;
        const float4 mean = *reinterpret_cast<const float4*>(stats + 4 * cg), inv = *reinterpret_cast<const float4*>(stats + C + 4 * cg);
        const float4 sc = *reinterpret_cast<const float4*>(stats + 2 * C + 4 * cg), sh = *reinterpret_cast<const float4*>(stats + 3 * C + 4 * cg);
        const float4 c0 = *reinterpret_cast<const float4*>(coef + 4 * cg), c1 = *reinterpret_cast<const float4*>(coef + C + 4 * cg);
        const float m_[4] = {mean.x, mean.y, mean.z, mean.w}, i_[4] = {inv.x, inv.y, inv.z, inv.w};
        const float sc_[4] = {sc.x, sc.y, sc.z, sc.w}, sh_[4] = {sh.x, sh.y, sh.z, sh.w};
        const float c0_[4] = {c0.x, c0.y, c0.z, c0.w}, c1_[4] = {c1.x, c1.y, c1.z, c1.w};
        const int f0 = a.pool ? 2 * fp : fp;
        const bool paired = a.pool && fp < Fo;                 // both members exist and were pooled
        const int64_t xo = (row * a.F + f0) * C + 4 * cg;
        const float4 x0 = *reinterpret_cast<const float4*>(a.x + xo);
        const float4 x1 = paired ? *reinterpret_cast<const float4*>(a.x + xo + C) : x0;
        float4 g = make_float4(0, 0, 0, 0);
        if (!a.pool || fp < Fo) g = *reinterpret_cast<const float4*>(dy + (row * Fo + fp) * C + 4 * cg);
        const float x0_[4] = {x0.x, x0.y, x0.z, x0.w}, x1_[4] = {x1.x, x1.y, x1.z, x1.w}, g_[4] = {g.x, g.y, g.z, g.w};
        float o0[4], o1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d0 = 0.f, d1 = 0.f;
            if (!a.pool || fp < Fo) route(bnz(x0_[k], m_[k], i_[k], sc_[k], sh_[k]), bnz(x1_[k], m_[k], i_[k], sc_[k], sh_[k]), g_[k], paired, d0, d1);
            const float gs = sc_[k] * i_[k];                                     // gamma * invstd
            o0[k] = gs * (d0 - c0_[k] - (x0_[k] - m_[k]) * i_[k] * c1_[k]);
            o1[k] = gs * (d1 - c0_[k] - (x1_[k] - m_[k]) * i_[k] * c1_[k]);
        }
        *reinterpret_cast<float4*>(dx + xo) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        if (paired) *reinterpret_cast<float4*>(dx + xo + C) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    }
}

int check_args(const char* what, const float* x, int64_t rows, int F, int C, int pool) {
    AMTX_REQUIRE(x && rows > 0 && F > 0, "%s: bad argument", what);
    AMTX_REQUIRE(C >= 4 && C % 4 == 0 && C <= 4 * BN_THREADS, "%s: channel count %d not supported (a multiple of 4, at most 1024)", what, C);
    AMTX_REQUIRE(pool == 0 || (pool == 1 && F >= 2), "%s: pool must be 0 or 1 (MaxPool(1,2))", what);
    return AMTX_OK;
}

int nblocks_for(int64_t items, int per_block) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(BN_MAX_PARTIALS, (items + per_block - 1) / per_block));
}

}  // namespace

// workspace (floats): partial sums [BN_MAX_PARTIALS][2][C] + per-channel constants [2][C]
extern "C" size_t amtx_bn_train_workspace_bytes(int channels) { return ((size_t)BN_MAX_PARTIALS * 2 + 2) * channels * sizeof(float); }

extern "C" int amtx_bn_relu_pool_train_fwd(const float* x, int64_t rows, int num_bins, int channels, int pool, const float* gamma,
                                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                           float* y, float* stats, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = check_args("amtx_bn_relu_pool_train_fwd", x, rows, num_bins, channels, pool);
    if (rc != AMTX_OK) return rc;
    AMTX_REQUIRE(y && stats && workspace && workspace_bytes >= amtx_bn_train_workspace_bytes(channels), "amtx_bn_relu_pool_train_fwd: null pointer / workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    const int C = channels, prn = BN_THREADS / (C / 4);
    const int64_t P = rows * num_bins;
    float* partial = (float*)workspace;
    const int nblk = nblocks_for(P, prn * 16);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nblk), dim3(BN_THREADS), 2 * BN_THREADS * sizeof(float4), s, x, P, C, partial);
    AMTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(64), 0, s, partial, nblk, C, (double)P, gamma, beta, eps, momentum,
                       running_mean, running_var, stats);
    AMTX_CHECK_LAUNCH();
    BnArgs a{x, rows, num_bins, C, pool};
    const int64_t total = rows * (pool ? num_bins / 2 : num_bins) * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)std::min<int64_t>((total + BN_THREADS - 1) / BN_THREADS, 256 * 32)), dim3(BN_THREADS), 0, s, a, stats, y);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

extern "C" int amtx_bn_relu_pool_train_bwd(const float* x, int64_t rows, int num_bins, int channels, int pool, const float* stats,
                                           const float* dy, float* dx, float* dgamma, float* dbeta, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    int rc = check_args("amtx_bn_relu_pool_train_bwd", x, rows, num_bins, channels, pool);
    if (rc != AMTX_OK) return rc;
    AMTX_REQUIRE(stats && dy && dx && workspace && workspace_bytes >= amtx_bn_train_workspace_bytes(channels), "amtx_bn_relu_pool_train_bwd: null pointer / workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    const int C = channels, prn = BN_THREADS / (C / 4);
    const int Fo = pool ? num_bins / 2 : num_bins;
    float* partial = (float*)workspace;
    float* coef = partial + (size_t)BN_MAX_PARTIALS * 2 * C;
    BnArgs a{x, rows, num_bins, C, pool};
    const int nblk = nblocks_for(rows * Fo, prn * 16);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk), dim3(BN_THREADS), 2 * BN_THREADS * sizeof(float4), s, a, stats, dy, partial);
    AMTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, s, partial, nblk, C, (double)rows * num_bins, stats, dgamma, dbeta, coef);
    AMTX_CHECK_LAUNCH();
    const int64_t total = rows * (pool ? (num_bins + 1) / 2 : num_bins) * (C / 4);
    hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3((unsigned)std::min<int64_t>((total + BN_THREADS - 1) / BN_THREADS, 256 * 32)), dim3(BN_THREADS), 0, s, a, stats, coef, dy, dx);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}
