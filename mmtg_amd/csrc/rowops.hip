// Row-wise HBM-bound kernels: LayerNorm forward/backward and column sums.
// One 64-lane wave per row, 4 elements per lane per step (8 B bf16 / 16 B f32
// coalesced vectors), statistics in fp32 via wave shuffles.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int LN_MAXIT = 4;  // cols <= 4 * 256 = 1024

template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
    f32x4 o = *reinterpret_cast<const f32x4*>(p);
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; v[3] = o[3];
}
template <> __device__ __forceinline__ void ld4<bf16>(const bf16* p, float (&v)[4]) {
    bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)o[0]; v[1] = (float)o[1]; v[2] = (float)o[2]; v[3] = (float)o[3];
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = o;
}
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, const float (&v)[4]) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (long)row * cols;
    float v[LN_MAXIT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
            ld4<T>(xr + c, v[it]);
            s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
        }
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[it][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    T* yr = y + (long)row * cols;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
            float gm[4], bt[4], o[4];
            ld4<float>(gamma + c, gm);
            ld4<float>(beta + c, bt);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mu) * rs * gm[e] + bt[e];
            st4<T>(yr + c, o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}


// bf16 rows of 512*A + 256*B columns, the column ownership of ln_bwd3_kernel (16-byte + 8-byte vectors per lane):
// gamma / beta live in registers, a wave walks rows with the next row's vector(s) already requested.
// Same arithmetic order per row as ln_fwd_kernel is NOT required (statistics are reductions over the row; the
// two kernels agree to fp32 rounding), the stored mean / rstd feed the backward either way.
template <int A, int B>
__global__ __launch_bounds__(256) void ln_fwd3_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps) {
    typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
    constexpr int NE = 8 * A + 4 * B, cols = 512 * A + 256 * B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#define LN3_COL(j) ((j) < 8 * A ? ((j) >> 3) * 512 + lane * 8 + ((j) & 7) : 512 * A + lane * 4 + ((j) - 8 * A))
    float gm[NE], bt[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) { gm[j] = gamma[LN3_COL(j)]; bt[j] = beta[LN3_COL(j)]; }
#undef LN3_COL
    const int row0 = blockIdx.x * 4 + wave, stride = gridDim.x * 4;
    bf16x8 px[A];
    bf16x4 qx;
    if (row0 < rows) {
#pragma unroll
        for (int a = 0; a < A; ++a) px[a] = *reinterpret_cast<const bf16x8*>(x + (long)row0 * cols + a * 512 + lane * 8);
        if constexpr (B) qx = *reinterpret_cast<const bf16x4*>(x + (long)row0 * cols + 512 * A + lane * 4);
    }
    for (int row = row0; row < rows; row += stride) {
        float v[NE];
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[8 * a + e] = (float)px[a][e];
        if constexpr (B) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[8 * A + e] = (float)qx[e];
        }
        const int nrow = row + stride;
        if (nrow < rows) {
#pragma unroll
            for (int a = 0; a < A; ++a) px[a] = *reinterpret_cast<const bf16x8*>(x + (long)nrow * cols + a * 512 + lane * 8);
            if constexpr (B) qx = *reinterpret_cast<const bf16x4*>(x + (long)nrow * cols + 512 * A + lane * 4);
        }
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < NE; ++j) sm += v[j];
        const float mu = wave_sum(sm) / cols;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NE; ++j) { const float d = v[j] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / cols + eps);
        bf16* yr = y + (long)row * cols;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (bf16)((v[8 * a + e] - mu) * rs * gm[8 * a + e] + bt[8 * a + e]);
            *reinterpret_cast<bf16x8*>(yr + a * 512 + lane * 8) = t;
        }
        if constexpr (B) {
            bf16x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (bf16)((v[8 * A + e] - mu) * rs * gm[8 * A + e] + bt[8 * A + e]);
            *reinterpret_cast<bf16x4*>(yr + 512 * A + lane * 4) = t;
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// One block = 4 waves; each wave strides over rows and keeps its dgamma / dbeta (/ column-sum)
// partials in registers; the block reduces them through LDS and writes ONE partial row per block to
// a workspace, a second tiny kernel sums the <= 512 partial rows per column.  (Atomics from ~1000
// blocks onto the same `cols` addresses ran 3x slower than the streaming part of the kernel.)
// Optionally also emits dx * dropout-mask (the gradient entering the previous residual branch's
// dropout) and its column sum (= that branch's bias gradient), saving two more passes over dx.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, T* __restrict__ dxm, float* __restrict__ ws,
                                                     int rows, int cols, int want_colsum,
                                                     uint32_t thresh, uint32_t seed, float inv_keep) {
    __shared__ float sred[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[LN_MAXIT][4], ab[LN_MAXIT][4], ac[LN_MAXIT][4], gm[LN_MAXIT][4];
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[it][e] = 0.f; ab[it][e] = 0.f; ac[it][e] = 0.f; gm[it][e] = 0.f; }
        if (c < cols) ld4<float>(gamma + c, gm[it]);
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[LN_MAXIT][4], dg[LN_MAXIT][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < LN_MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < cols) {
                float xv[4], dv[4];
                ld4<T>(x + (long)row * cols + c, xv);
                ld4<T>(dy + (long)row * cols + c, dv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[it][e] = (xv[e] - mu) * rs;
                    dg[it][e] = dv[e] * gm[it][e];
                    s1 += dg[it][e];
                    s2 += dg[it][e] * xh[it][e];
                    ag[it][e] += dv[e] * xh[it][e];
                    ab[it][e] += dv[e];
                }
            }
        }
        const float c1 = wave_sum(s1) / cols, c2 = wave_sum(s2) / cols;
#pragma unroll
        for (int it = 0; it < LN_MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < cols) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (dg[it][e] - c1 - xh[it][e] * c2);
                if (dres) {
                    float r4[4];
                    ld4<T>(dres + (long)row * cols + c, r4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += r4[e];
                }
                st4<T>(dx + (long)row * cols + c, o);
                if (want_colsum || dxm) {
                    // the consumer sees the rounded dx: mask / sum exactly what it will read
                    float m4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m4[e] = (float)(T)o[e];
                        if (thresh) m4[e] *= dropout_scale(seed, (uint32_t)((long)row * cols + c + e), thresh, inv_keep);
                        ac[it][e] += (float)(T)m4[e];
                    }
                    if (dxm) st4<T>(dxm + (long)row * cols + c, m4);
                }
            }
        }
    }
    // cross-wave reduction, one quantity at a time through a single 16 KB buffer (keeps LDS small
    // enough for 8+ resident blocks per CU: this kernel lives on thread-level parallelism)
    float* out = ws + (long)blockIdx.x * 3 * cols;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k == 2 && !want_colsum && !dxm) {
            for (int c = threadIdx.x; c < cols; c += 256) out[2 * cols + c] = 0.f;
            break;
        }
#pragma unroll
        for (int it = 0; it < LN_MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < cols) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sred[wave][c + e] = k == 0 ? ag[it][e] : k == 1 ? ab[it][e] : ac[it][e];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256)
            out[k * cols + c] = sred[0][c] + sred[1][c] + sred[2][c] + sred[3][c];
        __syncthreads();
    }
}

// v2 of the streaming part: a HALF-wave per row with 16-byte vectors (lane hl of the half owns the
// chunks hl, hl+32, ... of the row: 512 contiguous bytes per load instruction and half), so a wave
// has two rows in flight and every access is a full dwordx4.  Same partial-row workspace and
// finalize kernel as v1.  NC = chunks per lane (cols <= 32 * NC * 16/sizeof(T)).
#ifndef LN_BWD_WAVES
#define LN_BWD_WAVES 2      // waves per SIMD the kernel is compiled for (3 = register cap 168 spills 35 VGPRs at NC = 3)
#endif
template <typename T, int NC>
__global__ __launch_bounds__(256, LN_BWD_WAVES) void ln_bwd2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const T* __restrict__ dres,
                                                      T* __restrict__ dx, T* __restrict__ dxm, float* __restrict__ ws,
                                                      int rows, int cols, int want_colsum,
                                                      uint32_t thresh, uint32_t seed, float inv_keep,
                                                      bf16* __restrict__ dxp = nullptr, long plane = 0) {
    // dxp (x3 mode, T = float): the masked gradient also / instead as a (hi | lo) bf16 plane pair [rows, cols] -- the operand of the
    // split-precision products that consume it (mmtg_layernorm_bwd_x3)
    typedef typename Vec16<T>::type V;
    constexpr int EPC = Vec16<T>::N;
    __shared__ float sred[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
    const int nch = cols / EPC;
    float gm[NC][EPC], ag[NC][EPC], ab[NC][EPC], ac[NC][EPC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = hl + 32 * i;
#pragma unroll
        for (int e = 0; e < EPC; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; ac[i][e] = 0.f; gm[i][e] = 0.f; }
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + ch * EPC + e);
                gm[i][e] = g4[0]; gm[i][e + 1] = g4[1]; gm[i][e + 2] = g4[2]; gm[i][e + 3] = g4[3];
            }
        }
    }
    const float inv_cols = 1.0f / cols;
    // Software pipeline over rows: the x / dy vectors of the NEXT row pair (and the residual gradient of
    // this one) are requested before this pair is reduced, so a wave keeps two row pairs of loads in flight.
    // (only where the extra 8 * NC registers fit the 256-VGPR budget of two waves per SIMD)
    constexpr bool PF = sizeof(T) == 2 ? NC <= 3 : NC <= 4;
    V px[NC], pd[NC];
    if constexpr (PF) {
        const int row = (blockIdx.x * 4 + wave) * 2 + half;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int ch = hl + 32 * i;
            if (row < rows && ch < nch) {
                px[i] = *reinterpret_cast<const V*>(x + (long)row * cols + ch * EPC);
                pd[i] = *reinterpret_cast<const V*>(dy + (long)row * cols + ch * EPC);
            }
        }
    }
    for (int r2 = (blockIdx.x * 4 + wave) * 2; r2 < rows; r2 += gridDim.x * 8) {
        const int row = r2 + half;
        const bool ok = row < rows;
        const long base = (long)row * cols;
        const float mu = ok ? mean[row] : 0.f, rs = ok ? rstd[row] : 0.f;
        V cx[NC], cd[NC], rv[NC];
        const int nrow = row + gridDim.x * 8;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int ch = hl + 32 * i;
            if constexpr (PF) {
                cx[i] = px[i];
                cd[i] = pd[i];
                if (nrow < rows && ch < nch) {
                    px[i] = *reinterpret_cast<const V*>(x + (long)nrow * cols + ch * EPC);
                    pd[i] = *reinterpret_cast<const V*>(dy + (long)nrow * cols + ch * EPC);
                }
            }
            if constexpr (PF)
                if (dres && ok && ch < nch) rv[i] = *reinterpret_cast<const V*>(dres + base + ch * EPC);
        }
        float xh[NC][EPC], dg[NC][EPC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int ch = hl + 32 * i;
            if (ok && ch < nch) {
                V xv, dv;
                if constexpr (PF) { xv = cx[i]; dv = cd[i]; }
                else {
                    xv = *reinterpret_cast<const V*>(x + base + ch * EPC);
                    dv = *reinterpret_cast<const V*>(dy + base + ch * EPC);
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d = (float)dv[e];
                    xh[i][e] = ((float)xv[e] - mu) * rs;
                    dg[i][e] = d * gm[i][e];
                    s1 += dg[i][e];
                    s2 += dg[i][e] * xh[i][e];
                    ag[i][e] += d * xh[i][e];
                    ab[i][e] += d;
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { xh[i][e] = 0.f; dg[i][e] = 0.f; }
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const float c1 = s1 * inv_cols, c2 = s2 * inv_cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int ch = hl + 32 * i;
            if (ok && ch < nch) {
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = rs * (dg[i][e] - c1 - xh[i][e] * c2);
                if (dres) {
                    if constexpr (!PF) rv[i] = *reinterpret_cast<const V*>(dres + base + ch * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] += (float)rv[i][e];
                }
                V ov;
#pragma unroll
                for (int e = 0; e < EPC; ++e) ov[e] = (T)o[e];
                *reinterpret_cast<V*>(dx + base + ch * EPC) = ov;
                if (want_colsum || dxm || dxp) {
                    // the consumer sees the rounded dx: mask / sum exactly what it will read
                    V mv;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float m = (float)ov[e];
                        if (thresh) m *= dropout_scale(seed, (uint32_t)(base + ch * EPC + e), thresh, inv_keep);
                        mv[e] = (T)m;
                        ac[i][e] += (float)mv[e];
                    }
                    if (dxm) *reinterpret_cast<V*>(dxm + base + ch * EPC) = mv;
                    if constexpr (EPC == 4) {
                        if (dxp) {
                            bf16x4 hi, lo;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { hi[e] = (bf16)(float)mv[e]; lo[e] = (bf16)((float)mv[e] - (float)hi[e]); }
                            *reinterpret_cast<bf16x4*>(dxp + base + ch * 4) = hi;
                            *reinterpret_cast<bf16x4*>(dxp + plane + base + ch * 4) = lo;
                        }
                    }
                }
            }
        }
    }
    // fold the two halves of the wave, then the four waves through LDS, one quantity at a time
    float* out = ws + (long)blockIdx.x * 3 * cols;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k == 2 && !want_colsum && !dxm && !dxp) {
            for (int c = threadIdx.x; c < cols; c += 256) out[2 * cols + c] = 0.f;
            break;
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int ch = hl + 32 * i;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float v = k == 0 ? ag[i][e] : k == 1 ? ab[i][e] : ac[i][e];
                v += __shfl_xor(v, 32, 64);
                if (half == 0 && ch < nch) sred[wave][ch * EPC + e] = v;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256)
            out[k * cols + c] = sred[0][c] + sred[1][c] + sred[2][c] + sred[3][c];
        __syncthreads();
    }
}


// v3 (bf16, cols = 512*A + 256*B): a FULL wave per row -- lane l owns the 8 columns [512a + 8l, +8) of every
// 512-column group (one 16-byte vector) and, when B, the 4 columns [512A + 4l, +4) of the 256-column rest
// (one 8-byte vector): 12 columns per lane at 768 instead of the 24 of the half-wave layout, so the 3 x 12 column
// accumulators + operands fit 128 VGPRs and FOUR waves per SIMD keep twice the bytes in flight.  Same
// next-row prefetch, partial-row workspace and finalize kernel as v2.
template <int A, int B>
__global__ __launch_bounds__(256, (8 * A + 4 * B <= 8 ? 4 : 8 * A + 4 * B <= 12 ? 3 : 2)) void ln_bwd3_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const bf16* __restrict__ dres,
                                                         bf16* __restrict__ dx, bf16* __restrict__ dxm, float* __restrict__ ws,
                                                         int rows, int want_colsum,
                                                         uint32_t thresh, uint32_t seed, float inv_keep) {
    typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
    constexpr int NE = 8 * A + 4 * B, cols = 512 * A + 256 * B;
    __shared__ float sred[4][cols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // column of element j of this lane
#define LN3_COL(j) ((j) < 8 * A ? ((j) >> 3) * 512 + lane * 8 + ((j) & 7) : 512 * A + lane * 4 + ((j) - 8 * A))
    float gm[NE], ag[NE], ab[NE], ac[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) { gm[j] = gamma[LN3_COL(j)]; ag[j] = 0.f; ab[j] = 0.f; ac[j] = 0.f; }
    const float inv_cols = 1.0f / cols;
    bf16x8 px[A], pd[A];
    bf16x4 qx, qd;
    const int row0 = blockIdx.x * 4 + wave, stride = gridDim.x * 4;
    if (row0 < rows) {
        const long base = (long)row0 * cols;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            px[a] = *reinterpret_cast<const bf16x8*>(x + base + a * 512 + lane * 8);
            pd[a] = *reinterpret_cast<const bf16x8*>(dy + base + a * 512 + lane * 8);
        }
        if constexpr (B) {
            qx = *reinterpret_cast<const bf16x4*>(x + base + 512 * A + lane * 4);
            qd = *reinterpret_cast<const bf16x4*>(dy + base + 512 * A + lane * 4);
        }
    }
    for (int row = row0; row < rows; row += stride) {
        const long base = (long)row * cols;
        const float mu = mean[row], rs = rstd[row];
        float xv[NE], dv[NE], rv[NE];
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[8 * a + e] = (float)px[a][e]; dv[8 * a + e] = (float)pd[a][e]; }
        if constexpr (B) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[8 * A + e] = (float)qx[e]; dv[8 * A + e] = (float)qd[e]; }
        }
        const int nrow = row + stride;
        if (nrow < rows) {
            const long nb = (long)nrow * cols;
#pragma unroll
            for (int a = 0; a < A; ++a) {
                px[a] = *reinterpret_cast<const bf16x8*>(x + nb + a * 512 + lane * 8);
                pd[a] = *reinterpret_cast<const bf16x8*>(dy + nb + a * 512 + lane * 8);
            }
            if constexpr (B) {
                qx = *reinterpret_cast<const bf16x4*>(x + nb + 512 * A + lane * 4);
                qd = *reinterpret_cast<const bf16x4*>(dy + nb + 512 * A + lane * 4);
            }
        }
        if (dres) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(dres + base + a * 512 + lane * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[8 * a + e] = (float)t[e];
            }
            if constexpr (B) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(dres + base + 512 * A + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[8 * A + e] = (float)t[e];
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const float d = dv[j];
            xv[j] = (xv[j] - mu) * rs;          // xhat
            dv[j] = d * gm[j];                  // dy * gamma
            s1 += dv[j];
            s2 += dv[j] * xv[j];
            ag[j] += d * xv[j];
            ab[j] += d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const float c1 = s1 * inv_cols, c2 = s2 * inv_cols;
        bf16 ov[NE], mv[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            float o = rs * (dv[j] - c1 - xv[j] * c2);
            if (dres) o += rv[j];
            ov[j] = (bf16)o;
            if (want_colsum || dxm) {
                float m = (float)ov[j];          // the consumer sees the rounded dx: mask / sum exactly what it will read
                if (thresh) m *= dropout_scale(seed, (uint32_t)(base + LN3_COL(j)), thresh, inv_keep);
                mv[j] = (bf16)m;
                ac[j] += (float)mv[j];
            }
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = ov[8 * a + e];
            *reinterpret_cast<bf16x8*>(dx + base + a * 512 + lane * 8) = t;
            if (dxm) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = mv[8 * a + e];
                *reinterpret_cast<bf16x8*>(dxm + base + a * 512 + lane * 8) = t;
            }
        }
        if constexpr (B) {
            bf16x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = ov[8 * A + e];
            *reinterpret_cast<bf16x4*>(dx + base + 512 * A + lane * 4) = t;
            if (dxm) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = mv[8 * A + e];
                *reinterpret_cast<bf16x4*>(dxm + base + 512 * A + lane * 4) = t;
            }
        }
    }
    // the four waves' column sums through LDS, one quantity at a time
    float* out = ws + (long)blockIdx.x * 3 * cols;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k == 2 && !want_colsum && !dxm) {
            for (int c = threadIdx.x; c < cols; c += 256) out[2 * cols + c] = 0.f;
            break;
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) sred[wave][LN3_COL(j)] = k == 0 ? ag[j] : k == 1 ? ab[j] : ac[j];
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256)
            out[k * cols + c] = sred[0][c] + sred[1][c] + sred[2][c] + sred[3][c];
        __syncthreads();
    }
#undef LN3_COL
}

// Deterministic column sums (round 4).  Every reduction over rows that ends in a gradient tensor -- LayerNorm gains / biases,
// the Conv1D / Linear bias gradients, the token-type embedding rows -- used to finish with fp32 atomics from many workgroups
// (order = arrival order: the 123 tensors that differed in their last bits run to run).  Now ONE workgroup owns 64 columns and
// sums ALL the rows in a fixed order: 16 row lanes stride the rows (eight independent loads in flight each), their partial sums
// are added in lane order through LDS, and the single writer adds the result to (or stores it into) the destination.
//   rows r = 0 .. M - 1 at X + r * ldx (+ col0); dst[c] (+)= sum_r X[r][c]
template <typename T>
__device__ __forceinline__ void colsum_rows_block(const T* __restrict__ X, long ldx, int M, int N, float* __restrict__ dst, int accumulate,
                                                  int cb, float (*red)[64]) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;          // 64 columns x 16 row lanes
    const int c = cb * 64 + cl;
    float a = 0.f;
    if (c < N) {
        int r = rl;
        for (; r + 7 * 16 < M; r += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (float)X[(long)(r + 16 * u) * ldx + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; r < M; r += 16) a += (float)X[(long)r * ldx + c];
    }
    red[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        dst[c] = accumulate ? dst[c] + t : t;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void colsum_rows_kernel(const T* __restrict__ X, long ldx, int M, int N, float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];
    colsum_rows_block<T>(X, ldx, M, N, out, accumulate, blockIdx.x, red);
}

// stage 1 of a tall column sum (M in the thousands: too long a serial walk for one workgroup per 64 columns): slice s of the rows
// -> row s of the workspace (plain stores); stage 2 is colsum_rows_kernel over the slices
template <typename T>
__global__ __launch_bounds__(256) void colsum_slices_kernel(const T* __restrict__ X, long ldx, int M, int N, float* __restrict__ ws, int rows_per_block) {
    __shared__ float red[4][256];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cq) * 4;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        for (int r = r0 + rl; r < r1; r += 4) {
            float v[4];
            ld4<T>(X + (long)r * ldx + c, v);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = a[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) ws[(long)blockIdx.y * N + cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// LayerNorm backward, second stage: the per-workgroup partial rows [nblocks][3][cols] (dgamma | dbeta | column sum) -> the three
// gradients, one workgroup per (64 columns, quantity), fixed order
__global__ __launch_bounds__(1024) void ln_bwd_finalize_det_kernel(const float* __restrict__ ws, int nblocks, int cols,
        float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dcol) {
    __shared__ float red[16][64];
    const int q = blockIdx.y;
    float* dst = q == 0 ? dgamma : q == 1 ? dbeta : dcol;
    if (!dst) return;
    colsum_rows_block<float>(ws + (long)q * cols, 3L * cols, nblocks, cols, dst, 1, blockIdx.x, red);
}

// grid (cols/64, 3): block = 64 columns x 4 row groups of one quantity (dgamma / dbeta / colsum)
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ ws, int nblocks, int cols,
        float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dcol) {
    __shared__ float red[4][64];
    const int q = blockIdx.y;
    float* dst = q == 0 ? dgamma : q == 1 ? dbeta : dcol;
    if (!dst) return;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    // gridDim.z slices of the partial rows (16-way atomics per output element at the end: cheap,
    // and 16x the memory-level parallelism of one block per column group)
    const int per = (nblocks + gridDim.z - 1) / gridDim.z;
    const int k0 = blockIdx.z * per, k1 = min(nblocks, k0 + per);
    float a = 0.f;
    if (c < cols)
        for (int k = k0 + rg; k < k1; k += 4) a += ws[((long)k * 3 + q) * cols + c];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < cols) atomicAdd(dst + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// out[n] += sum_m X[m,n]: block = 256 threads -> 64 column-quads x 4 row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ldx, int M, int N,
                                                     float* __restrict__ out, int rows_per_block) {
    __shared__ float red[4][256];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cq) * 4;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        for (int r = r0 + rl; r < r1; r += 4) {
            float v[4];
            ld4<T>(X + (long)r * ldx + c, v);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cq * 4 + e] = a[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) atomicAdd(out + cc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Second half of a deterministic split-K product (MMTG_EPI_SPLIT): one block per output row sums the
// `splits` fp32 slabs in index order, adds the bias, applies the activation / residual, stores the row
// in the storage type -- and, when asked, LayerNorms the row it just produced (two-pass statistics on
// the values as stored, like ln_fwd_kernel) so the decode step needs no separate LayerNorm launches.
// NV = 4-column vectors per thread (N <= 1024 * NV).
template <typename T, int NV>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int splits, long slab, int N, long ldp,
        const float* __restrict__ bias, int epi, const T* __restrict__ aux, long ldaux, T* __restrict__ out, long ldo,
        const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ ln_out, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (tid + 256 * i) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
        if (c < N) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < splits; ++k) a += *reinterpret_cast<const f32x4*>(part + k * slab + (long)row * ldp + c);
            if (bias) a += *reinterpret_cast<const f32x4*>(bias + c);
            float o[4] = {a[0], a[1], a[2], a[3]};
            if (epi == MMTG_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = gelu_new_t<T>(o[e]);
            } else if (epi == MMTG_EPI_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = tanh_t<T>(o[e]);
            } else if (epi == MMTG_EPI_RESID) {
                float r4[4];
                ld4<T>(aux + (long)row * ldaux + c, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += r4[e];
            }
            st4<T>(out + (long)row * ldo + c, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = (float)(T)o[e]; s += v[i][e]; }
        }
    }
    if (!ln_out) return;
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / N;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((tid + 256 * i) * 4 < N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / N + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (tid + 256 * i) * 4;
        if (c < N) {
            float gm[4], bt[4], o[4];
            ld4<float>(gamma + c, gm);
            ld4<float>(beta + c, bt);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * gm[e] + bt[e];
            st4<T>(ln_out + (long)row * N + c, o);
        }
    }
}

}  // namespace

extern "C" int mmtg_splitk_finish(int dtype, const float* part, int splits, int M, int N, long ldp, const float* bias,
                                  int epi, const void* aux, long ldaux, void* out, long ldo,
                                  const float* ln_gamma, const float* ln_beta, void* ln_out, float eps, void* stream) {
    MMTG_REQUIRE(dtype == MMTG_F32 || dtype == MMTG_BF16, "splitk_finish: bad dtype");
    MMTG_REQUIRE(part && out && splits > 0 && M > 0 && N > 0 && N % 4 == 0 && N <= 4096 && ldp % 4 == 0 && ldo % 4 == 0,
                 "splitk_finish: N=%d must be a multiple of 4 and <= 4096", N);
    MMTG_REQUIRE(epi == MMTG_EPI_NONE || epi == MMTG_EPI_GELU || epi == MMTG_EPI_TANH || epi == MMTG_EPI_RESID, "splitk_finish: epilogue %d unsupported", epi);
    MMTG_REQUIRE(epi != MMTG_EPI_RESID || (aux && ldaux % 4 == 0), "splitk_finish: residual epilogue needs aux");
    MMTG_REQUIRE(!ln_out || (ln_gamma && ln_beta), "splitk_finish: LayerNorm needs gamma and beta");
    MMTG_REQUIRE(MMTG_ALIGNED16(part) && (!bias || MMTG_ALIGNED16(bias)), "splitk_finish: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, (double)splits * M * N, 4.0 * splits * M * N);
    const long slab = (long)M * ldp;
    const int nv = cdiv(N, 1024);
#define FIN(T, NV) hipLaunchKernelGGL((splitk_finish_kernel<T, NV>), dim3(M), dim3(256), 0, s, part, splits, slab, N, ldp, bias, epi, \
                                      (const T*)aux, ldaux, (T*)out, ldo, ln_gamma, ln_beta, (T*)ln_out, eps)
    if (dtype == MMTG_BF16) { if (nv == 1) FIN(bf16, 1); else if (nv == 2) FIN(bf16, 2); else if (nv == 3) FIN(bf16, 3); else FIN(bf16, 4); }
    else { if (nv == 1) FIN(float, 1); else if (nv == 2) FIN(float, 2); else if (nv == 3) FIN(float, 3); else FIN(float, 4); }
#undef FIN
    MMTG_LAUNCH_CHECK("splitk_finish");
    return MMTG_OK;
}

extern "C" int mmtg_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                                  float* mean, float* rstd, int rows, int cols, float eps, void* stream) {
    MMTG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 1024, "layernorm_fwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    MMTG_REQUIRE(x && y && gamma && beta && mean && rstd, "layernorm_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_LAYERNORM, s, 8.0 * rows * cols, 2.0 * esz * rows * cols);
    dim3 grid(cdiv(rows, 4)), block(256);
    static const bool fwd1 = getenv("MMTG_LN_V1") != nullptr || getenv("MMTG_LN_V2") != nullptr;   // A/B switch
    if (dtype == MMTG_BF16 && !fwd1 && (cols == 512 || cols == 768 || cols == 1024) && rows >= 2048 && MMTG_ALIGNED16(x) && MMTG_ALIGNED16(y)) {
        static int cusN = 0;
        if (!cusN) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cusN, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cusN <= 0) cusN = 256;
        }
        const int capf = getenv("MMTG_LN_FWD_CAP") ? atoi(getenv("MMTG_LN_FWD_CAP")) : 6 * cusN;
        int nbf = min(cdiv(rows, 4), capf > 0 ? capf : 1);
        nbf = cdiv(cdiv(rows, 4), cdiv(cdiv(rows, 4), nbf));          // equal sweeps per wave
        dim3 g3(nbf);
        if (cols == 512) hipLaunchKernelGGL((ln_fwd3_kernel<1, 0>), g3, block, 0, s, (const bf16*)x, (bf16*)y, gamma, beta, mean, rstd, rows, eps);
        else if (cols == 768) hipLaunchKernelGGL((ln_fwd3_kernel<1, 1>), g3, block, 0, s, (const bf16*)x, (bf16*)y, gamma, beta, mean, rstd, rows, eps);
        else hipLaunchKernelGGL((ln_fwd3_kernel<2, 0>), g3, block, 0, s, (const bf16*)x, (bf16*)y, gamma, beta, mean, rstd, rows, eps);
    } else if (dtype == MMTG_F32)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, gamma, beta, mean, rstd, rows, cols, eps);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, s, (const bf16*)x, (bf16*)y, gamma, beta, mean, rstd, rows, cols, eps);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "layernorm_fwd: bad dtype");
    MMTG_LAUNCH_CHECK("layernorm_fwd");
    return MMTG_OK;
}

static inline int ln_bwd_blocks(int rows) { return min(cdiv(rows, 4), 1024); }

extern "C" long mmtg_layernorm_bwd_ws(int rows, int cols) { return (long)ln_bwd_blocks(rows) * 3 * cols; }

// finalize: the second stage in the same call (dgamma / dbeta / dcolsum += the ordered sums of the partial rows); otherwise the partial
// rows stay in ws ([*partial_rows][3][cols]) for the caller's mmtg_colsum_batch and `want` alone says whether quantity 2 is produced
static int ln_bwd_impl(int dtype, const void* dy, const void* x, const float* gamma,
                       const float* mean, const float* rstd, const void* dres, void* dx,
                       float* dgamma, float* dbeta, int rows, int cols,
                       void* dx_masked, unsigned drop_thresh, unsigned drop_seed, float* dcolsum, int want_colsum,
                       float* ws, long ws_floats, bool finalize, int* partial_rows, void* stream) {
    MMTG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 1024, "layernorm_bwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    MMTG_REQUIRE(dy && x && gamma && mean && rstd && dx && (!finalize || (dgamma && dbeta)), "layernorm_bwd: null pointer");
    MMTG_REQUIRE(ws && ws_floats >= mmtg_layernorm_bwd_ws(rows, cols), "layernorm_bwd: workspace of %ld floats required",
                 mmtg_layernorm_bwd_ws(rows, cols));
    MMTG_REQUIRE(!drop_thresh || dx_masked, "layernorm_bwd: dropout mask requested without dx_masked");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_LAYERNORM, s, 16.0 * rows * cols, ((dres ? 4.0 : 3.0) + (dx_masked ? 1.0 : 0.0)) * esz * rows * cols);
    int nb = ln_bwd_blocks(rows);
    const float ik = drop_thresh ? (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh)) : 1.0f;
    const int want = finalize ? dcolsum != nullptr : want_colsum != 0;
    MMTG_REQUIRE(dtype == MMTG_F32 || dtype == MMTG_BF16, "layernorm_bwd: bad dtype");
    static const bool v1 = getenv("MMTG_LN_V1") != nullptr;      // A/B switch for measurements
    const int epc = dtype == MMTG_F32 ? 4 : 8;
    const bool vec = !v1 && cols % epc == 0 && MMTG_ALIGNED16(dy) && MMTG_ALIGNED16(x) && MMTG_ALIGNED16(dx) &&
                     (!dres || MMTG_ALIGNED16(dres)) && (!dx_masked || MMTG_ALIGNED16(dx_masked)) && MMTG_ALIGNED16(gamma);
    static const bool v2only = getenv("MMTG_LN_V2") != nullptr;  // A/B switch: half-wave-per-row kernel for bf16 too
    static const bool no1024 = getenv("MMTG_LN_NO1024") != nullptr;   // A/B switch: 1024 columns on the half-wave kernel (round 2)
    if (vec && !v2only && dtype == MMTG_BF16 && (cols == 512 || cols == 768 || (cols == 1024 && !no1024))) {
        // full wave per row; four / three resident 4-wave blocks per CU at 512 / 768 columns; at 1024 (16 columns per lane: the 3 x 16
        // column accumulators need ~200 VGPRs) two blocks per CU without spills instead of three with them (round 3)
        static int cap3 = 0, cus_ = 0;
        if (!cap3) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            cus_ = cus;
            cap3 = getenv("MMTG_LN_CAP") ? atoi(getenv("MMTG_LN_CAP")) : 3 * cus;
        }
        const int capx = cols == 1024 && !getenv("MMTG_LN_CAP") ? 2 * cus_ : cap3;
        if (nb > capx) nb = capx;
        const int sweeps = cdiv(cdiv(rows, 4), nb);
        nb = cdiv(cdiv(rows, 4), sweeps);
        dim3 grid(nb), block(256);
#define LN3(A_, B_) hipLaunchKernelGGL((ln_bwd3_kernel<A_, B_>), grid, block, 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, \
                                       (const bf16*)dres, (bf16*)dx, (bf16*)dx_masked, ws, rows, want, drop_thresh, drop_seed, ik)
        if (cols == 512) LN3(1, 0); else if (cols == 768) LN3(1, 1); else LN3(2, 0);
#undef LN3
    } else if (vec) {
        // half-wave per row: a block covers 8 rows per sweep; equal sweeps per wave, <= the v1 block count
        // one resident round: the v2 kernel holds ~200 VGPRs, i.e. two 4-wave blocks per CU; more blocks
        // than that run as a second round (measured 42 us at 944 blocks, 34 us at 472, 44 us at 236)
        static int cap = 0;
        if (!cap) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            cap = getenv("MMTG_LN_CAP") ? atoi(getenv("MMTG_LN_CAP")) : 2 * cus;
        }
        if (nb > cap) nb = cap;
        const int sweeps = cdiv(cdiv(rows, 8), nb);
        nb = cdiv(cdiv(rows, 8), sweeps);
        const int nc = cdiv(cols / epc, 32);
        dim3 grid(nb), block(256);
#define LN2(T, NC) hipLaunchKernelGGL((ln_bwd2_kernel<T, NC>), grid, block, 0, s, (const T*)dy, (const T*)x, gamma, mean, rstd, \
                                      (const T*)dres, (T*)dx, (T*)dx_masked, ws, rows, cols, want, drop_thresh, drop_seed, ik)
        if (dtype == MMTG_BF16) { if (nc <= 2) LN2(bf16, 2); else if (nc == 3) LN2(bf16, 3); else LN2(bf16, 4); }
        else { if (nc <= 4) LN2(float, 4); else if (nc <= 6) LN2(float, 6); else LN2(float, 8); }
#undef LN2
    } else {
        dim3 grid(nb), block(256);
        if (dtype == MMTG_F32)
            hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, block, 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (const float*)dres, (float*)dx, (float*)dx_masked, ws, rows, cols, want, drop_thresh, drop_seed, ik);
        else
            hipLaunchKernelGGL(ln_bwd_kernel<bf16>, grid, block, 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx, (bf16*)dx_masked, ws, rows, cols, want, drop_thresh, drop_seed, ik);
    }
    if (partial_rows) *partial_rows = nb;
    if (finalize) {
        static const bool fin_atomic = getenv("MMTG_LN_FINALIZE_ATOMIC") != nullptr;      // A/B: the round-1 z-sliced finalize (fp32 atomics)
        if (fin_atomic) hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(cdiv(cols, 64), 3, 16), dim3(256), 0, s, ws, nb, cols, dgamma, dbeta, dcolsum);
        else hipLaunchKernelGGL(ln_bwd_finalize_det_kernel, dim3(cdiv(cols, 64), 3), dim3(1024), 0, s, ws, nb, cols, dgamma, dbeta, dcolsum);
    }
    MMTG_LAUNCH_CHECK("layernorm_bwd");
    return MMTG_OK;
}

extern "C" int mmtg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                                  const float* mean, const float* rstd, const void* dres, void* dx,
                                  float* dgamma, float* dbeta, int rows, int cols,
                                  void* dx_masked, unsigned drop_thresh, unsigned drop_seed, float* dcolsum,
                                  float* ws, long ws_floats, void* stream) {
    return ln_bwd_impl(dtype, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, cols, dx_masked, drop_thresh, drop_seed, dcolsum, 0,
                       ws, ws_floats, true, nullptr, stream);
}

/* First stage only (round 6): d(x) (+ the masked copy) and the partial rows ws[k][q][cols] (q = 0: d gamma, 1: d beta, 2: the column sums
 * of the masked d(x) when want_colsum) of k < *partial_rows workgroups; the caller sums them later, many LayerNorms' in one
 * mmtg_colsum_batch launch, in the order the one-call form's second stage uses (the same bits). */
extern "C" int mmtg_layernorm_bwd_partial(int dtype, const void* dy, const void* x, const float* gamma,
                                          const float* mean, const float* rstd, const void* dres, void* dx, int rows, int cols,
                                          void* dx_masked, unsigned drop_thresh, unsigned drop_seed, int want_colsum,
                                          float* ws, long ws_floats, int* partial_rows, void* stream) {
    MMTG_REQUIRE(partial_rows, "layernorm_bwd_partial: null pointer");
    return ln_bwd_impl(dtype, dy, x, gamma, mean, rstd, dres, dx, nullptr, nullptr, rows, cols, dx_masked, drop_thresh, drop_seed, nullptr,
                       want_colsum, ws, ws_floats, false, partial_rows, stream);
}

/* x3 mode (fp32 rows): mmtg_layernorm_bwd whose (dropout-masked) input gradient ALSO / INSTEAD goes to a (hi | lo) bf16 plane pair
 * [rows, cols] (lo plane `plane` elements behind) -- the operand of the block's split-precision dgrad / weight-gradient products. */
static int ln_bwd_x3_impl(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                          void* dx_planes, long plane, unsigned drop_thresh, unsigned drop_seed, float* dcolsum, int want_colsum,
                          float* ws, long ws_floats, bool finalize, int* partial_rows, void* stream) {
    MMTG_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 1024, "layernorm_bwd_x3: cols=%d must be a multiple of 8 and <= 1024", cols);
    MMTG_REQUIRE(dy && x && gamma && mean && rstd && dx && (!finalize || (dgamma && dbeta)) && dx_planes, "layernorm_bwd_x3: null pointer");
    MMTG_REQUIRE(ws && ws_floats >= mmtg_layernorm_bwd_ws(rows, cols), "layernorm_bwd_x3: workspace of %ld floats required", mmtg_layernorm_bwd_ws(rows, cols));
    MMTG_REQUIRE(MMTG_ALIGNED16(dy) && MMTG_ALIGNED16(x) && MMTG_ALIGNED16(dx) && (!dres || MMTG_ALIGNED16(dres)) && MMTG_ALIGNED16(gamma) &&
                 MMTG_ALIGNED16(dx_planes) && plane % 8 == 0 && plane >= (long)rows * cols, "layernorm_bwd_x3: alignment / plane layout");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_LAYERNORM, s, 16.0 * rows * cols, ((dres ? 4.0 : 3.0) + 1.0) * 4.0 * rows * cols);
    int nb = ln_bwd_blocks(rows);
    const float ik = drop_thresh ? (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh)) : 1.0f;
    const int want = finalize ? dcolsum != nullptr : want_colsum != 0;
    static int cap = 0;
    if (!cap) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cap = 2 * cus;
    }
    if (nb > cap) nb = cap;
    const int sweeps = cdiv(cdiv(rows, 8), nb);
    nb = cdiv(cdiv(rows, 8), sweeps);
    const int nc = cdiv(cols / 4, 32);
    dim3 grid(nb), block(256);
#define LN2X(NC) hipLaunchKernelGGL((ln_bwd2_kernel<float, NC>), grid, block, 0, s, dy, x, gamma, mean, rstd, dres, dx, (float*)nullptr, ws, rows, cols, \
                                    want, drop_thresh, drop_seed, ik, (bf16*)dx_planes, plane)
    if (nc <= 4) LN2X(4); else if (nc <= 6) LN2X(6); else LN2X(8);
#undef LN2X
    if (partial_rows) *partial_rows = nb;
    if (finalize) hipLaunchKernelGGL(ln_bwd_finalize_det_kernel, dim3(cdiv(cols, 64), 3), dim3(1024), 0, s, ws, nb, cols, dgamma, dbeta, dcolsum);
    MMTG_LAUNCH_CHECK("layernorm_bwd_x3");
    return MMTG_OK;
}

extern "C" int mmtg_layernorm_bwd_x3(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                                     void* dx_planes, long plane, unsigned drop_thresh, unsigned drop_seed, float* dcolsum,
                                     float* ws, long ws_floats, void* stream) {
    return ln_bwd_x3_impl(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, cols, dx_planes, plane, drop_thresh, drop_seed, dcolsum, 0,
                          ws, ws_floats, true, nullptr, stream);
}

/* mmtg_layernorm_bwd_partial for the x3 form: the first stage only, the partial rows ws[k][q][cols] left for mmtg_colsum_batch */
extern "C" int mmtg_layernorm_bwd_x3_partial(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                             const float* dres, float* dx, int rows, int cols, void* dx_planes, long plane,
                                             unsigned drop_thresh, unsigned drop_seed, int want_colsum,
                                             float* ws, long ws_floats, int* partial_rows, void* stream) {
    MMTG_REQUIRE(partial_rows, "layernorm_bwd_x3_partial: null pointer");
    return ln_bwd_x3_impl(dy, x, gamma, mean, rstd, dres, dx, nullptr, nullptr, rows, cols, dx_planes, plane, drop_thresh, drop_seed, nullptr,
                          want_colsum, ws, ws_floats, false, partial_rows, stream);
}

// rows above which the sum runs in two stages (slices -> workspace -> ordered sum of the slices)
constexpr int COLSUM_TALL = 2048, COLSUM_SLICES = 128;

// Many small ordered column sums in ONE launch (round 6): the backward of a GPT-2 block ends four reductions of a few hundred fp32
// rows each (two LayerNorm second stages, the dGELU bands, the attention kernels' bias rows) -- 57 launches of 4-6 us per step that
// nothing depends on before the optimizer.  Item i is workgroup row i of the grid; the items travel in the kernel arguments.
namespace {
constexpr int COLSUM_BATCH = 64;
struct ColsumBatch { mmtg_colsum_item it[COLSUM_BATCH]; };
__global__ __launch_bounds__(1024) void colsum_batch_kernel(const ColsumBatch b) {
    __shared__ float red[16][64];
    const mmtg_colsum_item& e = b.it[blockIdx.y];
    if ((int)blockIdx.x * 64 >= e.N) return;          // (uniform per workgroup)
    colsum_rows_block<float>(e.X, e.ldx, e.M, e.N, e.out, 1, blockIdx.x, red);
}
}  // namespace

extern "C" int mmtg_colsum_batch(const mmtg_colsum_item* items, int n, void* stream) {
    MMTG_REQUIRE(items && n >= 0, "colsum_batch: null pointer");
    hipStream_t s = (hipStream_t)stream;
    double elems = 0;
    for (int i = 0; i < n; ++i) {
        const mmtg_colsum_item& e = items[i];
        MMTG_REQUIRE(e.X && e.out && e.M > 0 && e.M <= COLSUM_TALL && e.N > 0 && e.ldx >= e.N,
                     "colsum_batch: item %d: fp32 rows [M <= %d, N], ldx >= N", i, COLSUM_TALL);
        elems += (double)e.M * e.N;
    }
    if (!n) return MMTG_OK;
    ProfScope prof(MMTG_PROF_MISC, s, elems, 4.0 * elems);
    for (int i0 = 0; i0 < n; i0 += COLSUM_BATCH) {
        ColsumBatch b;
        const int m = min(COLSUM_BATCH, n - i0);
        int maxn = 0;
        for (int i = 0; i < COLSUM_BATCH; ++i) {
            b.it[i] = items[i0 + (i < m ? i : 0)];
            if (i < m) maxn = max(maxn, b.it[i].N);
        }
        hipLaunchKernelGGL(colsum_batch_kernel, dim3(cdiv(maxn, 64), m), dim3(1024), 0, s, b);
    }
    MMTG_LAUNCH_CHECK("colsum_batch");
    return MMTG_OK;
}
extern "C" long mmtg_colsum_ws(int M, int N) { return M > COLSUM_TALL ? (long)COLSUM_SLICES * N : 0; }

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, float* ws, long ws_floats, void* stream) {
    MMTG_REQUIRE(M > 0 && N > 0 && (M <= COLSUM_TALL || (N % 4 == 0 && ldx % 4 == 0)), "colsum: N=%d, ldx=%ld must be multiples of 4 above %d rows", N, ldx, COLSUM_TALL);
    MMTG_REQUIRE(X && out, "colsum: null pointer");
    MMTG_REQUIRE(dtype == MMTG_F32 || dtype == MMTG_BF16, "colsum: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_MISC, s, (double)M * N, esz * M * N);
    if (M <= COLSUM_TALL) {
        if (dtype == MMTG_F32) hipLaunchKernelGGL(colsum_rows_kernel<float>, dim3(cdiv(N, 64)), dim3(1024), 0, s, (const float*)X, ldx, M, N, out, 1);
        else hipLaunchKernelGGL(colsum_rows_kernel<bf16>, dim3(cdiv(N, 64)), dim3(1024), 0, s, (const bf16*)X, ldx, M, N, out, 1);
    } else if (!ws) {
        // (no workspace: the round-1 kernel -- row slices end in fp32 atomics; correct, not bit-reproducible)
        const int row_blocks = max(1, 1024 / cdiv(N, 256));
        const int rpb = max(16, (cdiv(M, row_blocks) + 3) & ~3);
        dim3 grid(cdiv(N, 256), cdiv(M, rpb)), block(256);
        if (dtype == MMTG_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)X, ldx, M, N, out, rpb);
        else hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, s, (const bf16*)X, ldx, M, N, out, rpb);
    } else {
        MMTG_REQUIRE(ws_floats >= mmtg_colsum_ws(M, N), "colsum: %d rows need a workspace of %ld floats (mmtg_colsum_ws)", M, mmtg_colsum_ws(M, N));
        const int rpb = (cdiv(M, COLSUM_SLICES) + 3) & ~3, nsl = cdiv(M, rpb);
        dim3 grid(cdiv(N, 256), nsl), block(256);
        if (dtype == MMTG_F32) hipLaunchKernelGGL(colsum_slices_kernel<float>, grid, block, 0, s, (const float*)X, ldx, M, N, ws, rpb);
        else hipLaunchKernelGGL(colsum_slices_kernel<bf16>, grid, block, 0, s, (const bf16*)X, ldx, M, N, ws, rpb);
        hipLaunchKernelGGL(colsum_rows_kernel<float>, dim3(cdiv(N, 64)), dim3(1024), 0, s, (const float*)ws, (long)N, nsl, N, out, 1);
    }
    MMTG_LAUNCH_CHECK("colsum");
    return MMTG_OK;
}
