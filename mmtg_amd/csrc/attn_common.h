// Shared pieces of the attention kernels (attention.hip, attention_x3.hip): LDS image layouts of [rows, 64] head tiles and their
// fragment loaders.  Head dim 64; storage bf16 (v_mfma_f32_16x16x32_bf16) or f32 (v_mfma_f32_16x16x4_f32).
#pragma once
#include "mma.h"

namespace {

constexpr int DH = 64;

template <typename T> struct AT {
    static constexpr int EPC = 16 / sizeof(T);          // elements per 16-byte chunk
    static constexpr int ROWB = DH * sizeof(T);         // bytes per [*, 64] row: 128 / 256
    static constexpr int CPR = ROWB / 16;               // chunks per row: 8 / 16
    static constexpr int KSTEPS = ROWB / 64;            // 64-byte k-blocks over dh: 2 / 4
    static constexpr int KBE = 64 / sizeof(T);          // elements per k-block: 32 / 16
    static constexpr int KPW = sizeof(T) == 2 ? 64 : 32;  // backward: keys per wave
};

__device__ __forceinline__ int vswz(int row) { return ((row >> 1) & 3) << 1; }
// row-read ("KC") image: chunk ^ (row & 7);  transposed-read ("KS") image: chunk ^ vswz(row)
template <typename T> __device__ __forceinline__ int off_kc(int row, int chunk) {
    return row * AT<T>::ROWB + ((chunk ^ (row & 7)) << 4);
}
template <typename T> __device__ __forceinline__ int off_ks(int row, int chunk) {
    return row * AT<T>::ROWB + ((chunk ^ vswz(row)) << 4);
}

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type ld_kc(const char* img, int row, int ks, int g) {
    return *reinterpret_cast<const typename Vec16<T>::type*>(img + off_kc<T>(row, ks * 4 + g));
}

// A/B fragment whose k index runs over the ROWS of a KS image, column block [col0, col0+16).
// bf16: rows {r_lo .. r_lo+3} and {r_hi .. r_hi+3} (per 16-lane group), via two transposed reads.
__device__ __forceinline__ bf16x8 ld_ks(const char* img, int r_lo, int r_hi, int col0, int lane, bf16) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = (col0 >> 3) + (p >> 1), sub = 8 * (p & 1);
    return tr_read_pair(img, off_ks<bf16>(r_lo + q, chunk) + sub, off_ks<bf16>(r_hi + q, chunk) + sub);
}
// f32: rows r_lo + s, s = 0..3
__device__ __forceinline__ f32x4 ld_ks(const char* img, int r_lo, int, int col0, int lane, float) {
    const int c = col0 + (lane & 15);
    f32x4 o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = *reinterpret_cast<const float*>(img + off_ks<float>(r_lo + s, c >> 2) + (c & 3) * 4);
    return o;
}

// exp: accurate expf in the fp32 parity mode, hardware v_exp_f32 path for bf16 storage
template <typename T> __device__ __forceinline__ float fexp(float x) {
    if constexpr (sizeof(T) == 2) return __expf(x);
    else return expf(x);
}

template <typename T> __device__ __forceinline__ typename Vec16<T>::type zero16() {
    typename Vec16<T>::type v;
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) v[e] = (T)0.f;
    return v;
}

}  // namespace
