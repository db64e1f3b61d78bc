// LM-head loss: row log-sum-exp + label gather, the rating-conditioned
// sequence loss of the reference (src/loss.py:45-74) with GPT-2's internal
// shifted CE (labels= at src/model.py:286), and the analytic d(loss)/d(logits).
// HBM-bound over the [M, V] fp32 logits (one read forward, one read backward).
#include "common.h"

namespace {

__device__ __forceinline__ long long label_of(const long long* topic_ids, const long long* targets,
                                              int b, int t1, int P, int L, int label_zero) {
    // label of row t is cat(topic_ids, targets)[b, t+1]; t1 = t + 1
    if (label_zero) return 0;
    return t1 < P ? topic_ids[(long)b * P + t1] : targets[(long)b * L + (t1 - P)];
}

__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    v = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return v;
}

// eight consecutive logits of a row as floats (fp32 rows: two 16-byte loads; bf16 rows: one)
__device__ __forceinline__ void load8(const float* p, float (&x)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16* p, float (&x)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)a[e];
}

// exp for the log-sum-exp: libm expf in the fp32 parity mode; for bf16-stored logits the hardware exponential (v_exp_f32,
// 1 ulp -- far below the bf16 rounding of the inputs): libm's ~20 instructions per element made the pass VALU-bound
// (150 us for 406 MB; round 2)
template <typename LT> __device__ __forceinline__ float lse_exp(float x) {
    if constexpr (sizeof(LT) == 2) return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    else return expf(x);
}

// one block per row: lse[m], nll[m] = lse - logit[label].  LT = storage type of the logits (fp32, or
// bf16 in the bf16 training path where the LM-head product stores them like every other activation).
constexpr int NSL = 8;       // 16-byte slices per thread the all-loads-first form of row_lse_kernel holds (rows <= 16384 logits)
template <typename LT>
__global__ __launch_bounds__(256) void row_lse_kernel(const LT* __restrict__ logits, long ldl, int V,
        const long long* __restrict__ topic_ids, const long long* __restrict__ targets, int label_zero,
        int P, int L, float* __restrict__ nll, float* __restrict__ lse) {
    __shared__ float sh[4];
    const int Tt = P + L;
    const long m = blockIdx.x;
    const int b = (int)(m / Tt), t = (int)(m % Tt);
    const LT* row = logits + m * ldl;
    // one pass: every thread keeps a running (max, sum of exp relative to it) over its 8-wide slices and
    // rescales when the max moves; the block then combines the 256 pairs (exact up to fp32 rounding)
    float mx = -INFINITY, sm = 0.f;
    const int V8 = V & ~7;
    if (V8 <= NSL * 2048) {
        // (round 3) rows of up to 16384 logits: every slice of the thread is REQUESTED before the first is used (the loop below
        // waits for each 16-byte load behind a data-dependent rescale: 3.1 TB/s on 406 MB), then one max and one exp pass
        float x[NSL][8];
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int v = threadIdx.x * 8 + i * 2048;
            load8(row + (v < V8 ? v : 0), x[i]);
        }
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            if (threadIdx.x * 8 + i * 2048 < V8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, x[i][e]);
            }
        }
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            if (threadIdx.x * 8 + i * 2048 < V8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += lse_exp<LT>(x[i][e] - mx);
            }
        }
    } else
    for (int v = threadIdx.x * 8; v < V8; v += 2048) {
        float x[8];
        load8(row + v, x);
        const float m8 = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
        if (m8 > mx) { sm *= lse_exp<LT>(mx - m8); mx = m8; }
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += lse_exp<LT>(x[e] - mx);
    }
    for (int v = V8 + threadIdx.x; v < V; v += 256) {
        const float x = (float)row[v];
        if (x > mx) { sm *= lse_exp<LT>(mx - x); mx = x; }
        sm += lse_exp<LT>(x - mx);
    }
    const float gmx = block_max(mx, sh);
    sm = block_sum(mx == -INFINITY ? 0.f : sm * lse_exp<LT>(mx - gmx), sh);
    mx = gmx;
    if (threadIdx.x == 0) {
        const float l = mx + logf(sm);
        lse[m] = l;
        float n = 0.f;
        if (t + 1 < Tt) {
            long long lab = label_of(topic_ids, targets, b, t + 1, P, L, label_zero);
            if (lab < 0) lab = 0;
            if (lab >= V) lab = V - 1;
            n = l - (float)row[lab];
        }
        nll[m] = n;
    }
}

// single block: per-sample CE, rating-conditioned loss, coefficients, both scalars
__global__ __launch_bounds__(1024) void sample_loss_kernel(const float* __restrict__ nll,
        const long long* __restrict__ ratings, int stage, int B, int P, int L, float batch_den,
        float* __restrict__ sample_ce, float* __restrict__ coef, float* __restrict__ scalars) {
    __shared__ float sh[32];
    const int Tt = P + L;
    const int ntok = Tt - 1 - P;  // rows P .. T-2  (loss.py:62-63)
    float my = 0.f, lm = 0.f;
    // (round 3: one wave per sample, the token sum spread over its lanes -- the one-thread-per-sample loop walked 235 dependent
    //  global loads per sample: 37 us of a 131 us loss forward; sixteen waves: every wave's samples are one load latency apart)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < B; b += 16) {
        float ce = 0.f, all = 0.f;
        for (int t = lane; t < Tt - 1; t += 64) {
            const float n = nll[(long)b * Tt + t];
            all += n;
            if (t >= P) ce += n;
        }
        ce = wave_sum(ce);
        all = wave_sum(all);
        if (lane != 0) continue;
        ce /= ntok;
        lm += all;
        float lb = 0.f, cf = 0.f;
        if (ratings) {
            const float y = ratings[b] > (stage == 1 ? 4 : 3) ? 1.f : 0.f;
            const float p = 1.0f / expf(ce);
            const float near0 = 1e-10f;
            lb = -y * logf(p + near0) - (1.f - y) * logf(1.f - p + near0);
            // d l / d ce = -p * (-y/(p+e) + (1-y)/(1-p+e))
            cf = -p * (-y / (p + near0) + (1.f - y) / (1.f - p + near0));
        }
        sample_ce[b] = ce;
        coef[b] = cf / (ntok * batch_den);
        my += lb;
    }
    if (lane == 0) { sh[wave] = my; sh[16 + wave] = lm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        my = lm = 0.f;
        for (int w = 0; w < 16; ++w) { my += sh[w]; lm += sh[16 + w]; }       // fixed order
        scalars[0] = my / batch_den;
        scalars[1] = lm / ((float)B * (Tt - 1));
    }
}

// (dlogits may alias logits when both are stored in the same type: a thread rewrites only what it read)
template <typename T, typename LT>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const LT* logits, long ldl, int V,
        const long long* __restrict__ topic_ids, const long long* __restrict__ targets,
        const float* __restrict__ lse, const float* __restrict__ coef, float gscale, float lm_coef, int P, int L,
        T* dlogits, long ldd, int Vpad, long plane = 0) {
    // plane > 0 (x3 mode, T = bf16, fp32 logits): dlogits is a (hi | lo) plane pair -- the LM head's dgrad and weight gradient are
    // split-precision products and nothing else reads d(logits), so the fp32 rows are never stored
    const int Tt = P + L;
    const long m = blockIdx.x;
    const int b = (int)(m / Tt), t = (int)(m % Tt);
    T* drow = dlogits + m * ldd;
    const bool active = t + 1 < Tt && (t >= P || lm_coef != 0.f);
    if (!active) {
        if ((Vpad & 7) == 0 && (ldd & 7) == 0) {
            typedef typename Vec16<T>::type V16;
            V16 z;
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) z[e] = (T)0.f;
            for (int v = threadIdx.x * Vec16<T>::N; v < Vpad; v += 256 * Vec16<T>::N) {
                *reinterpret_cast<V16*>(drow + v) = z;
                if (plane) *reinterpret_cast<V16*>(drow + plane + v) = z;
            }
        } else {
            for (int v = threadIdx.x; v < Vpad; v += 256) drow[v] = (T)0.f;
        }
        return;
    }
    const LT* row = logits + m * ldl;
    const float l = lse[m], cf = (t >= P ? coef[b] * gscale : 0.f) + lm_coef;
    long long lab = label_of(topic_ids, targets, b, t + 1, P, L, 0);
    if (lab < 0) lab = 0;
    if (lab >= V) lab = V - 1;
    const bool vec = (Vpad & 7) == 0 && (ldl & (sizeof(LT) == 4 ? 3 : 7)) == 0 && ldl >= Vpad && (ldd & 7) == 0;
    if (vec) {
        for (int v = threadIdx.x * 8; v < Vpad; v += 2048) {
            float d[8], x[8];
            load8(row + v, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int vv = v + e;
                d[e] = vv < V ? cf * (expf(x[e] - l) - (vv == (int)lab ? 1.f : 0.f)) : 0.f;
            }
            if constexpr (sizeof(T) == 2) {
                bf16x8 o = {(bf16)d[0], (bf16)d[1], (bf16)d[2], (bf16)d[3], (bf16)d[4], (bf16)d[5], (bf16)d[6], (bf16)d[7]};
                *reinterpret_cast<bf16x8*>(drow + v) = o;
                if (plane) {
                    bf16x8 lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) lo[e] = (bf16)(d[e] - (float)o[e]);
                    *reinterpret_cast<bf16x8*>(drow + plane + v) = lo;
                }
            } else {
                *reinterpret_cast<f32x4*>(drow + v) = f32x4{d[0], d[1], d[2], d[3]};
                *reinterpret_cast<f32x4*>(drow + v + 4) = f32x4{d[4], d[5], d[6], d[7]};
            }
        }
        return;
    }
    for (int v = threadIdx.x; v < Vpad; v += 256) {
        float d = 0.f;
        if (v < V) d = cf * (expf((float)row[v] - l) - (v == (int)lab ? 1.f : 0.f));
        drow[v] = (T)d;
    }
}

}  // namespace

extern "C" int mmtg_loss_fwd(int logits_dtype, const void* logits, long ldl, int V, const long long* topic_ids,
                             const long long* targets, const long long* ratings, int stage, int label_zero,
                             int B, int P, int L, float batch_den, float* nll, float* lse, float* sample_ce,
                             float* coef, float* scalars, void* stream) {
    MMTG_REQUIRE(logits_dtype == MMTG_F32 || logits_dtype == MMTG_BF16, "loss_fwd: bad logits dtype %d", logits_dtype);
    MMTG_REQUIRE(B > 0 && L >= 1 && P >= 0 && P + L >= 2 && V > 0 && ldl >= V && ldl % (logits_dtype == MMTG_F32 ? 4 : 8) == 0, "loss_fwd: bad sizes (V=%d ldl=%ld L=%d)", V, ldl, L);
    MMTG_REQUIRE(logits && targets && nll && lse && sample_ce && coef && scalars, "loss_fwd: null pointer");
    MMTG_REQUIRE(label_zero || P == 0 || topic_ids, "loss_fwd: topic_ids required");
    MMTG_REQUIRE(MMTG_ALIGNED16(logits), "loss_fwd: logits must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)B * (P + L);
    const double lsz = logits_dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_LOSS, s, 4.0 * M * V, lsz * M * V);
    if (logits_dtype == MMTG_F32)
        hipLaunchKernelGGL(row_lse_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (const float*)logits, ldl, V, topic_ids, targets, label_zero, P, L, nll, lse);
    else
        hipLaunchKernelGGL(row_lse_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, s, (const bf16*)logits, ldl, V, topic_ids, targets, label_zero, P, L, nll, lse);
    hipLaunchKernelGGL(sample_loss_kernel, dim3(1), dim3(1024), 0, s, nll, ratings, stage, B, P, L, batch_den, sample_ce, coef, scalars);
    MMTG_LAUNCH_CHECK("loss_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_loss_bwd(int dtype, int logits_dtype, const void* logits, long ldl, int V, const long long* topic_ids,
                             const long long* targets, const float* lse, const float* coef, float gscale, float lm_coef,
                             int B, int P, int L, void* dlogits, long ldd, int Vpad, void* stream) {
    MMTG_REQUIRE(B > 0 && L >= 1 && P + L >= 2 && V > 0 && Vpad >= V && ldd >= Vpad, "loss_bwd: bad sizes");
    MMTG_REQUIRE(logits && targets && lse && coef && dlogits, "loss_bwd: null pointer");
    MMTG_REQUIRE(dtype == MMTG_F32 || dtype == MMTG_BF16, "loss_bwd: bad dtype");
    MMTG_REQUIRE(logits_dtype == MMTG_F32 || (logits_dtype == MMTG_BF16 && dtype == MMTG_BF16), "loss_bwd: bf16 logits go with bf16 dlogits");
    MMTG_REQUIRE(logits != dlogits || (logits_dtype == dtype && ldl == ldd), "loss_bwd: in-place needs one storage type and one row stride");
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)B * (P + L);
    ProfScope prof(MMTG_PROF_LOSS, s, 3.0 * M * V, ((logits_dtype == MMTG_F32 ? 4.0 : 2.0) + (dtype == MMTG_F32 ? 4 : 2)) * M * V);
#define LB(T, LT) hipLaunchKernelGGL((loss_bwd_kernel<T, LT>), dim3((unsigned)M), dim3(256), 0, s, (const LT*)logits, ldl, V, topic_ids, targets, \
                                     lse, coef, gscale, lm_coef, P, L, (T*)dlogits, ldd, Vpad)
    if (dtype == MMTG_F32) LB(float, float);
    else if (logits_dtype == MMTG_F32) LB(bf16, float);
    else LB(bf16, bf16);
#undef LB
    MMTG_LAUNCH_CHECK("loss_bwd");
    return MMTG_OK;
}

/* x3 mode: d(logits) of fp32 logits straight into a (hi | lo) bf16 plane pair (ld = ldd, lo plane `plane` elements behind). */
extern "C" int mmtg_loss_bwd_x3(const float* logits, long ldl, int V, const long long* topic_ids, const long long* targets, const float* lse,
                                const float* coef, float gscale, float lm_coef, int B, int P, int L, void* planes, long ldd, long plane,
                                int Vpad, void* stream) {
    MMTG_REQUIRE(B > 0 && L >= 1 && P + L >= 2 && V > 0 && Vpad >= V && ldd >= Vpad && Vpad % 8 == 0 && ldd % 8 == 0 && ldl % 4 == 0 && ldl >= Vpad,
                 "loss_bwd_x3: bad sizes (Vpad, ldd multiples of 8, ldl >= Vpad)");
    MMTG_REQUIRE(logits && targets && lse && coef && planes && MMTG_ALIGNED16(planes) && MMTG_ALIGNED16(logits), "loss_bwd_x3: null / misaligned pointer");
    const long M = (long)B * (P + L);
    MMTG_REQUIRE(plane % 8 == 0 && plane >= (M - 1) * ldd + Vpad, "loss_bwd_x3: the lo plane must lie behind the hi plane");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_LOSS, s, 3.0 * M * V, 8.0 * M * V);
    hipLaunchKernelGGL((loss_bwd_kernel<bf16, float>), dim3((unsigned)M), dim3(256), 0, s, logits, ldl, V, topic_ids, targets, lse, coef, gscale, lm_coef,
                       P, L, (bf16*)planes, ldd, Vpad, plane);
    MMTG_LAUNCH_CHECK("loss_bwd_x3");
    return MMTG_OK;
}
