// Row-wise HBM-bound kernels: LayerNorm forward/backward and column sums.
// One 64-lane wave per row, 4 elements per lane per step (8 B bf16 / 16 B f32
// coalesced vectors), statistics in fp32 via wave shuffles.
#include "common.h"

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

// One block = 4 waves; each wave strides over rows, keeps dgamma/dbeta partials in
// registers, block-reduces them through LDS and issues one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows, int cols) {
    __shared__ float sg[4][1024];
    __shared__ float sb[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[LN_MAXIT][4], ab[LN_MAXIT][4], gm[LN_MAXIT][4];
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[it][e] = 0.f; ab[it][e] = 0.f; gm[it][e] = 0.f; }
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
            }
        }
    }
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sg[wave][c + e] = ag[it][e]; sb[wave][c + e] = ab[it][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
        atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
    }
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

}  // namespace

extern "C" int mmtg_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                                  float* mean, float* rstd, int rows, int cols, float eps, void* stream) {
    MMTG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 1024, "layernorm_fwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    MMTG_REQUIRE(x && y && gamma && beta && mean && rstd, "layernorm_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_LAYERNORM, s, 8.0 * rows * cols, 2.0 * esz * rows * cols);
    dim3 grid(cdiv(rows, 4)), block(256);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, gamma, beta, mean, rstd, rows, cols, eps);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, s, (const bf16*)x, (bf16*)y, gamma, beta, mean, rstd, rows, cols, eps);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "layernorm_fwd: bad dtype");
    MMTG_LAUNCH_CHECK("layernorm_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                                  const float* mean, const float* rstd, const void* dres, void* dx,
                                  float* dgamma, float* dbeta, int rows, int cols, void* stream) {
    MMTG_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 1024, "layernorm_bwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    MMTG_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_LAYERNORM, s, 16.0 * rows * cols, (dres ? 4.0 : 3.0) * esz * rows * cols);
    dim3 grid(min(cdiv(rows, 4), 1024)), block(256);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, block, 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (const float*)dres, (float*)dx, dgamma, dbeta, rows, cols);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<bf16>, grid, block, 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, rows, cols);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "layernorm_bwd: bad dtype");
    MMTG_LAUNCH_CHECK("layernorm_bwd");
    return MMTG_OK;
}

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, void* stream) {
    MMTG_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0, "colsum: N=%d, ldx=%ld must be multiples of 4", N, ldx);
    MMTG_REQUIRE(X && out, "colsum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_MISC, s, (double)M * N, esz * M * N);
    const int rpb = 256;
    dim3 grid(cdiv(N, 256), cdiv(M, rpb)), block(256);
    if (dtype == MMTG_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)X, ldx, M, N, out, rpb);
    else if (dtype == MMTG_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, s, (const bf16*)X, ldx, M, N, out, rpb);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "colsum: bad dtype");
    MMTG_LAUNCH_CHECK("colsum");
    return MMTG_OK;
}
