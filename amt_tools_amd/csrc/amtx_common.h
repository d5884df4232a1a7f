// Shared host/device helpers for the amtx HIP kernels (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/amtx.h"

// ------------------------------------------------------------------ error plumbing
void amtx_set_error(const char* fmt, ...);
// raise a kernel's dynamic-LDS limit to `bytes` on the CURRENT device if it is not there yet (thread-safe, per device)
int amtx_grant_lds(const void* kernel, size_t bytes);
#define AMTX_GRANT_LDS(kern, bytes)                                                      \
    do {                                                                                 \
        int _rc = amtx_grant_lds(reinterpret_cast<const void*>(kern), (size_t)(bytes));  \
        if (_rc != AMTX_OK) return _rc;                                                  \
    } while (0)

#define AMTX_CHECK_HIP(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            amtx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return AMTX_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define AMTX_REQUIRE(cond, ...)                                                          \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            amtx_set_error(__VA_ARGS__);                                                 \
            return AMTX_ERR_ARG;                                                         \
        }                                                                                \
    } while (0)

#define AMTX_CHECK_LAUNCH()                                                              \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            amtx_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return AMTX_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

// ------------------------------------------------------------------ 16-bit operand format of this translation unit
// conv.hip, convf.hip, gemm.hip and lstm.hip are compiled TWICE (amt_tools_amd/build.py): as they are, with bf16 operands, and with
// -DAMTX_F16 into a second object whose public functions carry the suffix _f16 (amtx_f16_names.h) and whose 16-bit values are IEEE
// half precision: the same matrix rate on gfx950 (v_mfma_f32_16x16x32_f16), three more mantissa bits -- the engine's precision 'f16'.
// In that build every name below that says "bf16" means "the 16-bit operand format of this build": bf16_t is a raw 16-bit pattern
// either way, and no kernel touches the bits except through these helpers.
typedef uint16_t bf16_t;   // raw 16-bit operand bits (bf16, or half in an AMTX_F16 build)

#ifdef AMTX_F16
static inline __host__ __device__ bf16_t f32_to_bf16_rn(float f) {      // round to nearest even (hardware or compiler conversion)
    // saturating: a folded weight beyond half's range (a BatchNorm scale over a tiny running variance) becomes +-65504, not +-inf -> NaN
    // downstream.  This is the packers' conversion (host and pack.hip); the kernels' own pack_bf16x2 of activations stays
    // a plain conversion: log-mel / HCQT features are in [0, 1] and the BatchNorm'd maps of these models far below 65504.
    if (f == f) f = f > 65504.0f ? 65504.0f : (f < -65504.0f ? -65504.0f : f);
    const _Float16 h = (_Float16)f;
    bf16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
static inline __host__ __device__ float bf16_to_f32(bf16_t u) {
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
#else
static inline __host__ __device__ bf16_t f32_to_bf16_rn(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                           // round to nearest even
    return (bf16_t)(u >> 16);
}

static inline __host__ __device__ float bf16_to_f32(bf16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
#endif

#if defined(__HIPCC__)
// vector types matching MFMA operand register counts
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (A/B operand of 16x16x32)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;   // 4 bf16 = 2 VGPRs
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // C/D of 16x16
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // C/D of 32x32

// pack two floats into one dword of two 16-bit operands (round to nearest even); lo -> bits 0..15.  gfx950 has both conversions in
// hardware (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32), reached through the native __bf16 / _Float16 types.
typedef __attribute__((ext_vector_type(2))) float amtx_f32x2;
#ifdef AMTX_F16
typedef __attribute__((ext_vector_type(2))) _Float16 amtx_bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 amtx_mfma_x8;
typedef __attribute__((ext_vector_type(4))) _Float16 amtx_mfma_x4;
#else
typedef __attribute__((ext_vector_type(2))) __bf16 amtx_bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 amtx_mfma_x8;
typedef __attribute__((ext_vector_type(4))) short amtx_mfma_x4;
#endif
static __device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const amtx_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, amtx_bf16x2));
}
// the two halves of a packed dword back as floats
static __device__ __forceinline__ float unpack16_lo(uint32_t v) {
#ifdef AMTX_F16
    return (float)__builtin_bit_cast(amtx_bf16x2, v)[0];
#else
    return __uint_as_float(v << 16);
#endif
}
static __device__ __forceinline__ float unpack16_hi(uint32_t v) {
#ifdef AMTX_F16
    return (float)__builtin_bit_cast(amtx_bf16x2, v)[1];
#else
    return __uint_as_float(v & 0xffff0000u);
#endif
}
// D = A (16 x 32) . B (32 x 16) + C on the matrix cores, operands in the build's 16-bit format, fp32 accumulate
typedef __attribute__((ext_vector_type(4))) float amtx_f32x4;
static __device__ __forceinline__ amtx_f32x4 amtx_mfma_16x16x32(uint4 a, uint4 b, amtx_f32x4 c) {
#ifdef AMTX_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(amtx_mfma_x8, a), __builtin_bit_cast(amtx_mfma_x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(amtx_mfma_x8, a), __builtin_bit_cast(amtx_mfma_x8, b), c, 0, 0, 0);
#endif
}
// the legacy 16-deep form (conv.hip's multi-channel fused first conv)
static __device__ __forceinline__ amtx_f32x4 amtx_mfma_16x16x16(uint2 a, uint2 b, amtx_f32x4 c) {
#ifdef AMTX_F16
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(amtx_mfma_x4, a), __builtin_bit_cast(amtx_mfma_x4, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(amtx_mfma_x4, a), __builtin_bit_cast(amtx_mfma_x4, b), c, 0, 0, 0);
#endif
}

// split x = hi + lo with hi = 16-bit(x), lo = 16-bit(x - hi): the two planes of the "x3" (split-bf16, fp32-class accuracy) MFMA path.
// Pairs at a time: one packed dword per plane.
static __device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - unpack16_lo(hi), b - unpack16_hi(hi));
}

static __device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// dB scaling of one power value (librosa.power_to_db with top_db = 80 relative to the clip's own maximum, then / 80 + 1; reference:
// amt_tools/features/common.py:199,218-228).  ONE definition, contraction pinned, for spec_scale_kernel and for the conv kernel that
// applies it to raw power values while staging them: both paths produce the same bits.  10 log10(x) is 10 log10(2) * v_log_f32(x)
// (the argument is clamped to >= 1e-10, so the hardware log2 sees no denormals; its ~1 ulp error is ~1e-7 on the scaled feature,
// three orders below the 1e-4 the features are held to) and / 80 is a multiplication: the conv kernel runs this per staged value.
struct DbScale { float offs, floor_db; };
static __device__ __forceinline__ float db_of_power(float s) {
#pragma clang fp contract(off)
    return 3.01029995663981195f * __log2f(fmaxf(1e-10f, s));
}
static __device__ __forceinline__ DbScale db_scale_make(float own_max, float ref) {
#pragma clang fp contract(off)
    DbScale d;
    d.offs = db_of_power(ref);
    d.floor_db = (db_of_power(own_max) - d.offs) - 80.0f;   // log_spec.max() - top_db
    return d;
}
static __device__ __forceinline__ float db_scale_apply(float s, DbScale d) {
#pragma clang fp contract(off)
    const float db = fmaxf(db_of_power(s) - d.offs, d.floor_db);
    return db * 0.0125f + 1.0f;
}
static __device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- LDS-DMA helpers ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;

// 16 bytes per lane HBM/L2 -> LDS without a VGPR round trip: LDS address = M0 (wave-uniform) + lane*16.
// Issued from inline asm on purpose: with the builtin, hipcc treats every later ds_read as a possible reader of the
// in-flight LDS write and puts `s_waitcnt vmcnt(0)` in front of it.  Ordering is the caller's job: a counted
// s_waitcnt vmcnt by the issuing wave, then a workgroup barrier, before anybody reads the data.
static __device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // provably wave-uniform for the "s" constraint
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

// Four pieces with ONE m0 set-up: piece n reads gsrc[n] and lands at lds_addr + n * 1024.  The instruction offset is added to both
// addresses, so the global pointers are passed n KiB low.
static __device__ __forceinline__ void glds16x4(const void* g0, const void* g1, const void* g2, const void* g3, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    const char* p1 = static_cast<const char*>(g1) - 1024;
    const char* p2 = static_cast<const char*>(g2) - 2048;
    const char* p3 = static_cast<const char*>(g3) - 3072;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(p1), "v"(p2), "v"(p3), "s"(lds_addr)
                 : "memory");
}

template <int N>
static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// workgroup barrier that waits for this wave's LDS traffic only (not for global loads/stores/DMA in flight)
static __device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// observed (speed only, never correctness): block b runs on XCD b % 8.  Remap so consecutive logical
// work items share an XCD (and its L2).  Bijective for any grid size.
static __device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned nx = 8;
    unsigned q = nblocks / nx, r = nblocks % nx;
    unsigned xcd = bid % nx, idx = bid / nx;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
#endif
