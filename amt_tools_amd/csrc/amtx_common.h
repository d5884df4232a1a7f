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

// ------------------------------------------------------------------ bf16 helpers
typedef uint16_t bf16_t;   // raw bf16 bits

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

#if defined(__HIPCC__)
// vector types matching MFMA operand register counts
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (A/B operand of 16x16x32)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;   // 4 bf16 = 2 VGPRs
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // C/D of 16x16
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // C/D of 32x32

// pack two floats into one dword of 2 x bf16 (round to nearest even); lo -> bits 0..15.
// gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32), reached through the native __bf16 type.
typedef __attribute__((ext_vector_type(2))) float amtx_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 amtx_bf16x2;
static __device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const amtx_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, amtx_bf16x2));
}

// split x = hi + lo with hi = bf16(x), lo = bf16(x - hi): the two planes of the "x3" (split-bf16,
// fp32-class accuracy) MFMA path.  Pairs at a time: one packed dword per plane.
static __device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
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
