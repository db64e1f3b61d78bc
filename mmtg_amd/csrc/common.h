// Shared device/host helpers for the MMTG gfx950 kernels.
// CDNA4 only: 64-lane wavefronts, MFMA 16x16 tiles, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mmtg_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---------------------------------------------------------------- errors
void mmtg_set_error(const char* fmt, ...);
#define MMTG_FAIL(code, ...)            \
    do {                                \
        mmtg_set_error(__VA_ARGS__);    \
        return (code);                  \
    } while (0)
#define MMTG_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) MMTG_FAIL(MMTG_ERR_BAD_ARG, __VA_ARGS__);    \
    } while (0)
#define MMTG_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)
int mmtg_check_launch(const char* what);
#define MMTG_LAUNCH_CHECK(what)                  \
    do {                                         \
        int _rc = mmtg_check_launch(what);       \
        if (_rc) return _rc;                     \
    } while (0)

// profiling hooks (abi.hip): bracket a launch with hipEvents when enabled
void mmtg_prof_begin(int cat, hipStream_t s);
void mmtg_prof_end(int cat, hipStream_t s, double flops, double bytes);
struct ProfScope {
    int cat; hipStream_t s; double flops, bytes;
    ProfScope(int c, hipStream_t st, double f, double b) : cat(c), s(st), flops(f), bytes(b) { mmtg_prof_begin(cat, s); }
    ~ProfScope() { mmtg_prof_end(cat, s, flops, bytes); }
};

// ---------------------------------------------------------------- scalar helpers
template <typename T> __device__ __forceinline__ float to_f(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

__device__ __forceinline__ float gelu_new_f(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))   (GPT-2 "gelu_new")
    const float k = 0.7978845608028654f;
    float u = k * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_new_grad_f(float x) {
    const float k = 0.7978845608028654f;
    float x2 = x * x;
    float u = k * (x + 0.044715f * x * x2);
    float t = tanhf(u);
    float du = k * (1.0f + 3.0f * 0.044715f * x2);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Storage-type-aware transcendental helpers: exact libm tanhf in the fp32 parity mode, the
// v_exp_f32-based form (4 instructions instead of ~25, error << bf16 rounding) for bf16 storage.
// bf16 storage: v_exp_f32 + v_rcp_f32 (1 ulp each, far below the bf16 rounding of the result) and no
// IEEE division -- `a / b` alone expands to ~10 instructions (v_div_scale/fmas/fixup), which made
// the GELU epilogues VALU-bound (6.8 us vs 2.5 us per 128x128 tile, profiles/r01_v4_gemm_timeline.log).
// gelu_new(x) = x * s,  s = sigmoid(2u) = 1 / (1 + 2^(-2u log2 e)),  u = k (x + a x^3);  saturates
// cleanly (2^+inf = inf -> rcp = 0; 2^-inf = 0 -> rcp(1) = 1).
__device__ __forceinline__ float gelu_sigmoid_fast(float x, float x2) {
    const float c0 = 2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * (c0 + c1 * x2)));
}
template <typename T> __device__ __forceinline__ float tanh_t(float x) {
    if constexpr (sizeof(T) == 2)
        return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (2.0f * 1.4426950408889634f)));
    else return tanhf(x);
}
template <typename T> __device__ __forceinline__ float gelu_new_t(float x) {
    if constexpr (sizeof(T) == 2) return x * gelu_sigmoid_fast(x, x * x);
    else {
        const float k = 0.7978845608028654f;
        return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
    }
}
template <typename T> __device__ __forceinline__ float gelu_new_grad_t(float x) {
    const float k = 0.7978845608028654f;
    const float x2 = x * x;
    if constexpr (sizeof(T) == 2) {
        // d/dx [x s] = s + x s (1 - s) 2k (1 + 3a x^2)
        const float s = gelu_sigmoid_fast(x, x2);
        return s * (1.0f + x * (1.0f - s) * (2.0f * k + (6.0f * k * 0.044715f) * x2));
    } else {
        const float t = tanhf(k * (x + 0.044715f * x * x2));
        return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k * (1.0f + 3.0f * 0.044715f * x2);
    }
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte vector load/store of N elements of T (N*sizeof(T) == 16)
template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Vec16<bf16> { typedef bf16x8 type; static constexpr int N = 8; };

// counter-based RNG for dropout: one 32-bit hash per element index.
// (murmur3 finaliser over (seed, idx); cheap, stateless, reproducible in bwd)
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint32_t idx) {
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep) {
    // keep iff hash >= thresh, thresh = p * 2^32
    return hash_u32(seed, idx) >= thresh ? inv_keep : 0.0f;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
