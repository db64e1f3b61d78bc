// Step tail of the training loop (train.py:194-197): global grad-norm,
// clip + transformers.AdamW update fused with the bf16 weight-copy refresh,
// and the dtype casts the engine needs.  Pure HBM streaming kernels:
// 16-byte vectors, grid-stride, <= 2048 blocks.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float sh[4];
    float a = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) a += x[i] * x[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    // round 4: one partial per workgroup (plain store) + an ordered final sum instead of ~2000 fp32 atomics on one word -- the
    // norm, and through the clip coefficient every parameter, is reproducible bit for bit
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// out[0] = sum of the n partials in a fixed order (thread t: partials t, t + 256, ...; then lanes, then waves)
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
        float* __restrict__ m, float* __restrict__ v, bf16* __restrict__ pc, bf16* __restrict__ pl, long n, float lr, float b1, float b2,
        float eps, float wd, float step_size, const float* __restrict__ normsq, float max_norm, float gscale,
        const float* __restrict__ count) {
    // data-parallel steps hand over SUMS of per-row gradients and the global row count as a device scalar
    // (the all-reduced count never visits the host); an empty global batch leaves every buffer untouched,
    // as the reference skips the step (train.py:184-185)
    if (count) {
        const float c = *count;
        if (!(c > 0.f)) return;
        gscale *= 1.0f / c;
    }
    float coef = gscale;
    if (normsq) {
        // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
        const float nrm = sqrtf(*normsq) * gscale;
        coef *= fminf(1.0f, max_norm / (nrm + 1e-6f));
    }
    // 16-byte vectors (four parameters per lane and pass); the flat buffers are 16-byte aligned and n is a multiple of 4
    // for every layout the engine builds (a scalar tail covers anything else)
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                          reinterpret_cast<uintptr_t>(v)) & 15) == 0 && ((reinterpret_cast<uintptr_t>(pc) | reinterpret_cast<uintptr_t>(pl)) & 7) == 0;
    const long n4 = vec_ok ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 g4 = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i], p4 = reinterpret_cast<f32x4*>(p)[i];
        bf16x4 c4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = g4[e] * coef;
            const float mi = b1 * m4[e] + (1.f - b1) * gi;
            const float vi = b2 * v4[e] + (1.f - b2) * gi * gi;
            float pi = p4[e] - step_size * (mi / (sqrtf(vi) + eps));
            if (wd > 0.f) pi -= lr * wd * pi;
            m4[e] = mi; v4[e] = vi; p4[e] = pi;
            c4[e] = (bf16)pi;
            l4[e] = (bf16)(pi - (float)c4[e]);
        }
        reinterpret_cast<f32x4*>(m)[i] = m4;
        reinterpret_cast<f32x4*>(v)[i] = v4;
        reinterpret_cast<f32x4*>(p)[i] = p4;
        if (pc) reinterpret_cast<bf16x4*>(pc)[i] = c4;
        if (pl) reinterpret_cast<bf16x4*>(pl)[i] = l4;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        float pi = p[i] - step_size * (mi / (sqrtf(vi) + eps));
        if (wd > 0.f) pi -= lr * wd * pi;
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (pc) pc[i] = (bf16)pi;
        if (pl) pl[i] = (bf16)(pi - (float)(bf16)pi);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (T)src[i];
}
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)src[i];
}
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_rows_kernel(const float* __restrict__ src, long lds_, T* __restrict__ dst,
                                                            long ldd, int cols) {
    const long r = blockIdx.x;
    for (int c = threadIdx.x; c < ldd; c += 256) dst[r * ldd + c] = c < cols ? (T)src[r * lds_ + c] : (T)0.f;
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}

// Second half of a weight-gradient product stored as K-split slabs (mmtg_gemm, transA, MMTG_EPI_SPLIT):
// dst[i] (+)= sum_s part[s * stride + i], slabs added in order (deterministic).  One float4 per lane;
// the slabs were written just before and are still in the Infinity Cache.
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part, int splits, long stride,
                                                       float* __restrict__ dst, int accumulate, long n4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f4* src = reinterpret_cast<const f4*>(part) + i;
        f4 acc = accumulate ? reinterpret_cast<const f4*>(dst)[i] : f4{0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            const f4 a = src[(s + 0) * (stride / 4)], b = src[(s + 1) * (stride / 4)];
            const f4 c = src[(s + 2) * (stride / 4)], d = src[(s + 3) * (stride / 4)];
            acc += a; acc += b; acc += c; acc += d;
        }
        for (; s < splits; ++s) acc += src[s * (stride / 4)];
        reinterpret_cast<f4*>(dst)[i] = acc;
    }
}

// Batched matrix transpose (weight-copy maintenance): matrix i is [rows, cols] row-major at
// src + desc[4i], written as [cols, rows] at dst + desc[4i+3].  One 64x64 tile per workgroup through
// a padded LDS image; 16-byte global vectors on both sides (rows, cols multiples of 16/sizeof(T)).
template <typename T>
__global__ __launch_bounds__(256) void transpose_batch_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                              const long* __restrict__ desc) {
    typedef typename Vec16<T>::type V;
    constexpr int EPC = Vec16<T>::N, CPR = 64 / EPC, RPP = 256 / CPR;   // chunks per tile row, rows per pass
    constexpr int LDT = 64 + EPC;                                         // padded LDS row (16-byte multiple)
    __shared__ __attribute__((aligned(16))) T tile[64 * LDT];
    const long* d = desc + 4 * blockIdx.y;
    const long so = d[0], dof = d[3];
    const int rows = (int)d[1], cols = (int)d[2];
    const int tc = (cols + 63) >> 6, tr = (rows + 63) >> 6;
    if ((int)blockIdx.x >= tr * tc) return;
    const int r0 = ((int)blockIdx.x / tc) * 64, c0 = ((int)blockIdx.x % tc) * 64;
    const int tid = threadIdx.x, ch = tid % CPR, rr = tid / CPR;
#pragma unroll
    for (int ps = 0; ps < 64 / RPP; ++ps) {
        const int r = rr + ps * RPP;
        V v;
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = (T)0.0f;
        if (r0 + r < rows && c0 + ch * EPC < cols) v = *reinterpret_cast<const V*>(src + so + (long)(r0 + r) * cols + c0 + ch * EPC);
        *reinterpret_cast<V*>(tile + r * LDT + ch * EPC) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 64 / RPP; ++ps) {
        const int j = rr + ps * RPP;          // source column = destination row
        V v;
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = tile[(ch * EPC + e) * LDT + j];
        if (c0 + j < cols && r0 + ch * EPC < rows) *reinterpret_cast<V*>(dst + dof + (long)(c0 + j) * rows + r0 + ch * EPC) = v;
    }
}

// Pull a buffer through the memory hierarchy (into the 256 MB Infinity Cache) ahead of the kernel that will read it:
// 16-byte loads, the xor of everything goes to a sink only if it equals a value it cannot take.
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ src, long n16, unsigned* __restrict__ sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const u32x4 v = src[i];
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u && sink) *sink = 1u;      // (keeps the loads alive)
}

// Zero a list of ranges of one fp32 buffer in ONE launch: desc[2 r] = first element (a multiple of 4), desc[2 r + 1] = count (a
// multiple of 4).  The fused trainer zeroes only the gradients that are ACCUMULATED into (LayerNorm / bias columns, embeddings,
// LM head, encoder): the block matrices, 340 of the 497 MB, are overwritten by their slab sums.
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long* __restrict__ desc, int n) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < n; ++r) {
        f32x4* p = reinterpret_cast<f32x4*>(base + desc[2 * r]);
        const long cnt = desc[2 * r + 1] >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (long)gridDim.x * 256) p[i] = z;
    }
}

inline unsigned grid_for(long n) { return (unsigned)min((long)2048, (n + 255) / 256); }

}  // namespace

extern "C" int mmtg_prefetch(const void* src, long bytes, int workgroups, void* sink, void* stream) {
    MMTG_REQUIRE(src && bytes >= 16 && MMTG_ALIGNED16(src) && workgroups > 0, "prefetch: bad args");
    hipLaunchKernelGGL(prefetch_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, bytes / 16, (unsigned*)sink);
    MMTG_LAUNCH_CHECK("prefetch");
    return MMTG_OK;
}

extern "C" int mmtg_zero_ranges(float* base, const long* desc, int n, void* stream) {
    MMTG_REQUIRE(base && desc && n > 0 && MMTG_ALIGNED16(base), "zero_ranges: bad args");
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, base, desc, n);
    MMTG_LAUNCH_CHECK("zero_ranges");
    return MMTG_OK;
}

extern "C" long mmtg_sumsq_ws(long n) { return (long)grid_for(n / 4 + 1); }

extern "C" int mmtg_sumsq(const float* x, long n, float* out, float* ws, long ws_floats, void* stream) {
    MMTG_REQUIRE(x && out && n > 0 && MMTG_ALIGNED16(x), "sumsq: bad args");
    const int g = grid_for(n / 4 + 1);
    MMTG_REQUIRE(ws && ws_floats >= g, "sumsq: workspace of %d floats required (mmtg_sumsq_ws)", g);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_OPTIM, s, 2.0 * n, 4.0 * n);
    hipLaunchKernelGGL(sumsq_kernel, dim3(g), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, g, out);
    MMTG_LAUNCH_CHECK("sumsq");
    return MMTG_OK;
}

extern "C" int mmtg_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, void* p_lo, long n,
                          float lr, float beta1, float beta2, float eps, float wd, int step,
                          const float* normsq, float max_norm, float grad_scale, const float* count, void* stream) {
    MMTG_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
    MMTG_REQUIRE(!p_lo || p_bf16, "adamw: the lo plane comes with the hi plane (p_bf16)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_OPTIM, s, 12.0 * n, (28.0 + (p_bf16 ? 2 : 0) + (p_lo ? 2 : 0)) * n);
    // transformers.AdamW(correct_bias=True): step_size = lr * sqrt(1-b2^t) / (1-b1^t)
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, p, g, m, v, (bf16*)p_bf16, (bf16*)p_lo, n, lr, beta1, beta2,
                       eps, wd, step_size, normsq, max_norm, grad_scale, count);
    MMTG_LAUNCH_CHECK("adamw");
    return MMTG_OK;
}

extern "C" int mmtg_cast_f32_to(int dtype, const float* src, void* dst, long n, void* stream) {
    MMTG_REQUIRE(src && dst && n > 0, "cast_f32_to: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 0, 6.0 * n);
    if (dtype == MMTG_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16>, dim3(grid_for(n)), dim3(256), 0, s, src, (bf16*)dst, n);
    else if (dtype == MMTG_F32) hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, src, (float*)dst, n);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "cast_f32_to: bad dtype");
    MMTG_LAUNCH_CHECK("cast_f32_to");
    return MMTG_OK;
}

extern "C" int mmtg_cast_to_f32(int dtype, const void* src, float* dst, long n, void* stream) {
    MMTG_REQUIRE(src && dst && n > 0, "cast_to_f32: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 0, 6.0 * n);
    if (dtype == MMTG_BF16) hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, dim3(grid_for(n)), dim3(256), 0, s, (const bf16*)src, dst, n);
    else if (dtype == MMTG_F32) hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)src, dst, n);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "cast_to_f32: bad dtype");
    MMTG_LAUNCH_CHECK("cast_to_f32");
    return MMTG_OK;
}

extern "C" int mmtg_cast_pad_rows(int dtype, const float* src, long lds_, void* dst, long ldd, int rows, int cols, void* stream) {
    MMTG_REQUIRE(src && dst && rows > 0 && cols > 0 && ldd >= cols && lds_ >= cols, "cast_pad_rows: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 0, 6.0 * rows * cols);
    if (dtype == MMTG_BF16) hipLaunchKernelGGL(cast_pad_rows_kernel<bf16>, dim3(rows), dim3(256), 0, s, src, lds_, (bf16*)dst, ldd, cols);
    else if (dtype == MMTG_F32) hipLaunchKernelGGL(cast_pad_rows_kernel<float>, dim3(rows), dim3(256), 0, s, src, lds_, (float*)dst, ldd, cols);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "cast_pad_rows: bad dtype");
    MMTG_LAUNCH_CHECK("cast_pad_rows");
    return MMTG_OK;
}

extern "C" int mmtg_axpy_f32(float* y, const float* x, float a, long n, void* stream) {
    MMTG_REQUIRE(x && y && n > 0, "axpy: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 2.0 * n, 12.0 * n);
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, a, n);
    MMTG_LAUNCH_CHECK("axpy");
    return MMTG_OK;
}

extern "C" int mmtg_slab_sum(const float* part, int splits, long stride, float* dst, int accumulate, long n, void* stream) {
    MMTG_REQUIRE(part && dst && splits > 0 && n > 0, "slab_sum: bad args");
    MMTG_REQUIRE(n % 4 == 0 && stride % 4 == 0 && stride >= n && MMTG_ALIGNED16(part) && MMTG_ALIGNED16(dst),
                 "slab_sum: n and stride must be multiples of 4 floats, stride >= n, buffers 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, (double)splits * n, 4.0 * n * (splits + 1 + (accumulate ? 1 : 0)));
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, part, splits, stride, dst, accumulate, n4);
    MMTG_LAUNCH_CHECK("slab_sum");
    return MMTG_OK;
}

extern "C" int mmtg_transpose_batch(int dtype, const void* src, void* dst, const long* desc, int n, int max_rows, int max_cols, void* stream) {
    MMTG_REQUIRE(src && dst && desc && n > 0 && max_rows > 0 && max_cols > 0, "transpose_batch: bad args");
    MMTG_REQUIRE(MMTG_ALIGNED16(src) && MMTG_ALIGNED16(dst), "transpose_batch: 16-byte aligned buffers required");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 0, 0);
    dim3 grid(cdiv(max_rows, 64) * cdiv(max_cols, 64), n);
    if (dtype == MMTG_BF16) hipLaunchKernelGGL(transpose_batch_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src, (bf16*)dst, desc);
    else if (dtype == MMTG_F32) hipLaunchKernelGGL(transpose_batch_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, desc);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "transpose_batch: bad dtype");
    MMTG_LAUNCH_CHECK("transpose_batch");
    return MMTG_OK;
}
