// Split-precision ("bf16x3") mode, round 5: the row-wise producers of (hi | lo) bf16 plane pairs.
//
// The reference's arithmetic is fp32 (/root/reference/src/model.py:279-288 and the GPT-2 blocks behind it).  The x3 mode keeps
// every tensor in fp32 and runs the big products on the bf16 matrix cores as  X W ~ X_hi W_hi + X_lo W_hi + X_hi W_lo  with
//   hi = bf16(x),  lo = bf16(x - hi)      (x - hi is exact in fp32; |x - hi - lo| <= 2^-18 |x|)
// and fp32 accumulation (mmtg_gemm_x3, mmtg_wgrad_group config 2, mmtg_decode_gemm_x3).  A product operand therefore travels
// as TWO bf16 planes of the operand's shape -- the same 4 bytes per element as the fp32 tensor -- written here:
//   mmtg_split_planes       any fp32 matrix -> its plane pair (weights once per optimizer step, activations whose producer is
//                           an fp32 kernel that also needs the fp32 value: attention context, LayerNorm-backward outputs, ...)
//   mmtg_layernorm_fwd_x3   LayerNorm whose ONLY consumer is a product: writes the plane pair instead of the fp32 rows
#include "common.h"

namespace {

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (bf16)v[e];
        lo[e] = (bf16)(v[e] - (float)hi[e]);
    }
}

// rows x cols fp32 (ld = lds) -> planes (ld = ldp, lo plane `plane` elements behind the hi plane); cols % 8 == 0
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long lds_, int rows, int cols,
                                                           bf16* __restrict__ dst, long ldp, long plane) {
    const int cpr = cols >> 3;                       // 8-element chunks per row
    const long n = (long)rows * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) * 8;
        const float* s = src + r * lds_ + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        bf16x8 hi, lo;
        split8(v, hi, lo);
        bf16* d = dst + r * ldp + c;
        *reinterpret_cast<bf16x8*>(d) = hi;
        *reinterpret_cast<bf16x8*>(d + plane) = lo;
    }
}

// LayerNorm forward, fp32 rows in, plane pair out: one wave per row, 8 elements per lane and pass (cols % 8 == 0, <= 1024)
__global__ __launch_bounds__(256) void ln_fwd_planes_kernel(const float* __restrict__ x, bf16* __restrict__ y, long ldp, long plane,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int cols, float eps, bf16* __restrict__ xb) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * cols;
    float v[2][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 512 + lane * 8;
        if (c < cols) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + c), b = *reinterpret_cast<const f32x4*>(xr + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] = a[e]; v[it][4 + e] = b[e]; }
            if (xb) {       // (bf16x3f: the rows as the bf16 backward's LayerNorm kernels read them)
                bf16x8 xo;
#pragma unroll
                for (int e = 0; e < 8; ++e) xo[e] = (bf16)v[it][e];
                *reinterpret_cast<bf16x8*>(xb + (long)row * cols + c) = xo;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[it][e];
        }
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 512 + lane * 8;
        if (c < cols) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[it][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    bf16* yr = y + (long)row * ldp;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 512 + lane * 8;
        if (c < cols) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[it][e] - mu) * rs * g0[e] + b0[e];
                o[4 + e] = (v[it][4 + e] - mu) * rs * g1[e] + b1[e];
            }
            bf16x8 hi, lo;
            split8(o, hi, lo);
            *reinterpret_cast<bf16x8*>(yr + c) = hi;
            *reinterpret_cast<bf16x8*>(yr + c + plane) = lo;
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

int x3_grid(long n) {
    const long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

}  // namespace

extern "C" int mmtg_split_planes(const float* src, long lds_, int rows, int cols, void* planes, long ldp, long plane, void* stream) {
    MMTG_REQUIRE(src && planes && rows > 0 && cols > 0, "split_planes: null pointer or empty matrix");
    MMTG_REQUIRE(cols % 8 == 0 && lds_ % 4 == 0 && ldp % 8 == 0 && plane % 8 == 0 && MMTG_ALIGNED16(src) && MMTG_ALIGNED16(planes),
                 "split_planes: cols, ldp, plane distance %% 8 == 0, lds %% 4 == 0, 16-byte aligned pointers");
    MMTG_REQUIRE(lds_ >= cols && ldp >= cols && plane >= (long)(rows - 1) * ldp + cols, "split_planes: leading dimensions / plane distance too small");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 2.0 * rows * cols, 8.0 * rows * cols);
    hipLaunchKernelGGL(split_planes_kernel, dim3(x3_grid((long)rows * (cols / 8))), dim3(256), 0, s, src, lds_, rows, cols, (bf16*)planes, ldp, plane);
    MMTG_LAUNCH_CHECK("split_planes");
    return MMTG_OK;
}

extern "C" int mmtg_layernorm_fwd_x3(const float* x, void* planes, long ldp, long plane, const float* gamma, const float* beta,
                                     float* mean, float* rstd, int rows, int cols, float eps, void* x_bf16, void* stream) {
    MMTG_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 1024, "layernorm_fwd_x3: cols=%d must be a multiple of 8 and <= 1024", cols);
    MMTG_REQUIRE(x && planes && gamma && beta && mean && rstd, "layernorm_fwd_x3: null pointer");
    MMTG_REQUIRE(ldp % 8 == 0 && plane % 8 == 0 && ldp >= cols && plane >= (long)(rows - 1) * ldp + cols && MMTG_ALIGNED16(x) && MMTG_ALIGNED16(planes) &&
                 MMTG_ALIGNED16(gamma) && MMTG_ALIGNED16(beta) && MMTG_ALIGNED16(x_bf16), "layernorm_fwd_x3: alignment / plane layout");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_LAYERNORM, s, 8.0 * rows * cols, 8.0 * rows * cols);
    hipLaunchKernelGGL(ln_fwd_planes_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, (bf16*)planes, ldp, plane, gamma, beta, mean, rstd, rows, cols, eps, (bf16*)x_bf16);
    MMTG_LAUNCH_CHECK("layernorm_fwd_x3");
    return MMTG_OK;
}
