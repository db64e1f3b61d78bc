// MFMA 16x16 tile helpers shared by the GEMM and attention kernels (gfx950).
//
// One "k-block" is 64 BYTES of the reduction dimension: 32 bf16 (one
// v_mfma_f32_16x16x32_bf16) or 16 f32 (four v_mfma_f32_16x16x4_f32).
// Operand convention of mma16(a, b, c):  D[i][j] += sum_k A[i][k] B[k][j]
//   lane l = 16*g + i15 holds A[row i15][k in lane-group g] and B[k in g][col i15]
//   bf16: element e (0..7) <-> k = 8g + e          f32: element s (0..3) <-> k = 4g + s
//   D: lane holds D[row 4g + r][col i15] in c[r]   (same map for both dtypes)
#pragma once
#include "common.h"

__device__ __forceinline__ void mma16(const bf16x8& a, const bf16x8& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const f32x4& a, const f32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
}

// Two ds_read_b64_tr_b16: lane (g, i15) receives column i15 of the 4-row blocks whose row
// addresses the 16 lanes of its group supply (lane 4q+p -> row q, columns 4p..4p+3).
// `o1`/`o2` are this lane's byte offsets for the two blocks.
__device__ __forceinline__ bf16x8 tr_read_pair(const char* tile, int o1, int o2) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + o1));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + o2));
    bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
    return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Accumulator tile(s) as the next product's B operand (k = the tile's ROW index):
// bf16 packs two 16-row tiles into one 32-deep k-block: e<4 -> row 4g+e of `lo`,
// e>=4 -> row 4g+e-4 of `hi` (the other operand must use the same k order);
// f32 uses one 16-row tile per k-block: s -> row 4g+s.
__device__ __forceinline__ bf16x8 acc_as_operand(const f32x4& lo, const f32x4& hi, bf16) {
    bf16x8 o = {(bf16)lo[0], (bf16)lo[1], (bf16)lo[2], (bf16)lo[3],
                (bf16)hi[0], (bf16)hi[1], (bf16)hi[2], (bf16)hi[3]};
    return o;
}
