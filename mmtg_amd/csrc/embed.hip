// Conditioning front end: WenLan row gather + experience-vector add, its
// backward segment sum, and the GPT-2 input-embedding add (wpe + wte[type]).
// All HBM-bound: one workgroup per token row, 16-byte coalesced vectors.
#include "common.h"

namespace {

template <typename T> struct V16 { typedef typename Vec16<T>::type type; };

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type vadd(typename Vec16<T>::type a, typename Vec16<T>::type b) {
    typename Vec16<T>::type o;
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) o[e] = (T)((float)a[e] + (float)b[e]);
    return o;
}

// x[b,t,:] = E[id] (+ c[b,seg,:])
template <typename T>
__global__ __launch_bounds__(256) void embed_condition_kernel(const T* __restrict__ table,
        const long long* __restrict__ topic_ids, const long long* __restrict__ targets,
        const T* __restrict__ c, T* __restrict__ x, int P, int L, int S, int E, int two_sents, int V) {
    typedef typename Vec16<T>::type V16t;
    constexpr int N = Vec16<T>::N;
    const int Tt = P + L;
    const int b = blockIdx.x / Tt, t = blockIdx.x % Tt;
    long long id;
    int seg = -1;
    if (t < P) id = topic_ids[(long)b * P + t];
    else {
        const int p = t - P;
        id = targets[(long)b * L + p];
        const int k = p / two_sents;
        if (k < S) seg = k;
    }
    if (id < 0) id = 0;
    if (id >= V) id = V - 1;
    const T* src = table + (long)id * E;
    const T* cs = seg >= 0 ? c + ((long)b * S + seg) * E : nullptr;
    T* dst = x + (long)blockIdx.x * E;
    for (int e = threadIdx.x * N; e < E; e += 256 * N) {
        V16t v = *reinterpret_cast<const V16t*>(src + e);
        if (cs) v = vadd<T>(v, *reinterpret_cast<const V16t*>(cs + e));
        *reinterpret_cast<V16t*>(dst + e) = v;
    }
}

// out[b,k,:] = sum_{p in [k*ts, (k+1)*ts) , p < L} g[b,P+p,:]
template <typename T>
__global__ __launch_bounds__(256) void segment_sum_kernel(const T* __restrict__ g, T* __restrict__ out,
                                                          int P, int L, int S, int H, int two_sents) {
    const int b = blockIdx.x / S, k = blockIdx.x % S;
    const int Tt = P + L;
    const int p0 = k * two_sents, p1 = min(L, p0 + two_sents);
    for (int h = threadIdx.x; h < H; h += 256) {
        float a = 0.f;
        for (int p = p0; p < p1; ++p) a += (float)g[((long)b * Tt + P + p) * H + h];
        out[((long)b * S + k) * H + h] = (T)a;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_add_kernel(const T* __restrict__ g, const T* __restrict__ wpe,
        const T* __restrict__ wte, const long long* __restrict__ type_ids, T* __restrict__ h,
        int Tt, int D, uint32_t thresh, uint32_t seed, float inv_keep) {
    const long m = blockIdx.x;
    const int t = (int)(m % Tt);
    const long long ty = type_ids[m];
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (float)g[m * D + d + e] + (float)wpe[(long)t * D + d + e] + (float)wte[ty * D + d + e];
            if (thresh) v *= dropout_scale(seed, (uint32_t)(m * D + d + e), thresh, inv_keep);
            h[m * D + d + e] = (T)v;
        }
    }
}

// dwpe[t,d] += sum_b dh[b,t,d]   grid (T, D/256)
template <typename T>
__global__ __launch_bounds__(256) void embed_dwpe_kernel(const T* __restrict__ dh, float* __restrict__ dwpe,
                                                         int Bn, int Tt, int D) {
    const int t = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
    if (d >= D) return;
    float a = 0.f;
    for (int b = 0; b < Bn; ++b) a += (float)dh[((long)b * Tt + t) * D + d];
    dwpe[(long)t * D + d] += a;
}

// dwte[type,d] += sum_m [type_ids[m]==type] dh[m,d]; per-block LDS bins then atomics
template <typename T>
__global__ __launch_bounds__(256) void embed_dwte_kernel(const T* __restrict__ dh, const long long* __restrict__ type_ids,
                                                         float* __restrict__ dwte, int M, int D, int ntypes, int rows_per_block,
                                                         float* __restrict__ ws) {
    extern __shared__ float bins[];  // [ntypes][256]
    const int d = blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < ntypes; ++k) bins[k * 256 + threadIdx.x] = 0.f;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    if (d < D) {
        for (int m = r0; m < r1; ++m) {
            const int ty = (int)type_ids[m];
            if (ty >= 0 && ty < ntypes) bins[ty * 256 + threadIdx.x] += (float)dh[(long)m * D + d];
        }
        for (int k = 0; k < ntypes; ++k) {
            const float v = bins[k * 256 + threadIdx.x];
            if (ws) ws[((long)blockIdx.y * ntypes + k) * D + d] = v;        // row block's bins: summed in order by the second stage
            else if (v != 0.f) atomicAdd(dwte + (long)k * D + d, v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ y, long n,
                                                            uint32_t thresh, uint32_t seed, float inv_keep) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = (T)((float)x[i] * dropout_scale(seed, (uint32_t)i, thresh, inv_keep));
}

inline float inv_keep_of(unsigned thresh) {
    return thresh ? (float)(4294967296.0 / (4294967296.0 - (double)thresh)) : 1.0f;
}

}  // namespace

#define DISPATCH(dtype, KERN, ...)                                                     \
    if ((dtype) == MMTG_F32) { KERN(float, __VA_ARGS__); }                              \
    else if ((dtype) == MMTG_BF16) { KERN(bf16, __VA_ARGS__); }                         \
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "bad dtype %d", (dtype));

extern "C" int mmtg_embed_condition(int dtype, const void* table, const long long* topic_ids,
                                    const long long* targets, const void* c, void* x,
                                    int B, int P, int L, int S, int E, int two_sents, int V, void* stream) {
    MMTG_REQUIRE(B > 0 && P >= 0 && L > 0 && S > 0 && two_sents > 0 && V > 0, "embed_condition: bad sizes");
    MMTG_REQUIRE(E % 8 == 0, "embed_condition: E=%d must be a multiple of 8", E);
    MMTG_REQUIRE(table && targets && c && x && (P == 0 || topic_ids), "embed_condition: null pointer");
    MMTG_REQUIRE(MMTG_ALIGNED16(table) && MMTG_ALIGNED16(c) && MMTG_ALIGNED16(x), "embed_condition: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    const double rows = (double)B * (P + L);
    ProfScope prof(MMTG_PROF_EMBED, s, rows * E, esz * (2.0 * rows * E + (double)B * S * E));
    dim3 grid(B * (P + L)), block(256);
#define K_(T, ...) hipLaunchKernelGGL(embed_condition_kernel<T>, grid, block, 0, s, (const T*)table, topic_ids, targets, (const T*)c, (T*)x, P, L, S, E, two_sents, V)
    DISPATCH(dtype, K_, 0)
#undef K_
    MMTG_LAUNCH_CHECK("embed_condition");
    return MMTG_OK;
}

extern "C" int mmtg_segment_sum(int dtype, const void* g, void* out, int B, int P, int L, int S, int H,
                                int two_sents, void* stream) {
    MMTG_REQUIRE(B > 0 && L > 0 && S > 0 && H > 0 && two_sents > 0 && g && out, "segment_sum: bad args");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_EMBED, s, (double)B * L * H, esz * ((double)B * L * H + (double)B * S * H));
    dim3 grid(B * S), block(256);
#define K_(T, ...) hipLaunchKernelGGL(segment_sum_kernel<T>, grid, block, 0, s, (const T*)g, (T*)out, P, L, S, H, two_sents)
    DISPATCH(dtype, K_, 0)
#undef K_
    MMTG_LAUNCH_CHECK("segment_sum");
    return MMTG_OK;
}

extern "C" int mmtg_embed_add(int dtype, const void* g, const void* wpe, const void* wte, const long long* type_ids,
                              void* h, int M, int T, int D, unsigned drop_thresh, unsigned drop_seed, void* stream) {
    MMTG_REQUIRE(M > 0 && T > 0 && D % 4 == 0 && g && wpe && wte && type_ids && h, "embed_add: bad args");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_EMBED, s, 2.0 * M * D, esz * 4.0 * M * D);
    dim3 grid(M), block(256);
    const float ik = inv_keep_of(drop_thresh);
#define K_(T_, ...) hipLaunchKernelGGL(embed_add_kernel<T_>, grid, block, 0, s, (const T_*)g, (const T_*)wpe, (const T_*)wte, type_ids, (T_*)h, T, D, drop_thresh, drop_seed, ik)
    DISPATCH(dtype, K_, 0)
#undef K_
    MMTG_LAUNCH_CHECK("embed_add");
    return MMTG_OK;
}

extern "C" int mmtg_dropout_apply(int dtype, const void* x, void* y, long n, int N, unsigned drop_thresh,
                                  unsigned drop_seed, void* stream) {
    (void)N;
    MMTG_REQUIRE(n > 0 && x && y, "dropout_apply: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, (double)n, (dtype == MMTG_F32 ? 8.0 : 4.0) * n);
    dim3 grid((unsigned)min((long)4096, (n + 255) / 256)), block(256);
    const float ik = inv_keep_of(drop_thresh);
#define K_(T_, ...) hipLaunchKernelGGL(dropout_apply_kernel<T_>, grid, block, 0, s, (const T_*)x, (T_*)y, n, drop_thresh, drop_seed, ik)
    DISPATCH(dtype, K_, 0)
#undef K_
    MMTG_LAUNCH_CHECK("dropout_apply");
    return MMTG_OK;
}

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, float* ws, long ws_floats, void* stream);
extern "C" long mmtg_embed_add_bwd_ws(int M, int D, int ntypes) { return (long)cdiv(M, 128) * ntypes * D; }

extern "C" int mmtg_embed_add_bwd(int dtype, void* dh, const long long* type_ids, float* dwpe, float* dwte,
                                  int M, int T, int D, int ntypes, unsigned drop_thresh, unsigned drop_seed, float* ws, long ws_floats,
                                  void* stream) {
    MMTG_REQUIRE(M > 0 && T > 0 && M % T == 0 && D > 0 && ntypes > 0 && ntypes <= 32 && dh && type_ids && dwpe && dwte,
                 "embed_add_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (drop_thresh) {
        int rc = mmtg_dropout_apply(dtype, dh, dh, (long)M * D, D, drop_thresh, drop_seed, stream);
        if (rc) return rc;
    }
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_EMBED, s, 2.0 * M * D, esz * 2.0 * M * D);
    dim3 block(256);
    dim3 g1(T, cdiv(D, 256));
    const int rpb = 128;
    dim3 g2(cdiv(D, 256), cdiv(M, rpb));
    const size_t shm = (size_t)ntypes * 256 * sizeof(float);
    // token-type rows of wte: per-row-block bins -> workspace -> ordered sum over the row blocks (round 4: no fp32 atomics, the
    // gradient is reproducible bit for bit); without a workspace the round-1 atomics remain
    MMTG_REQUIRE(!ws || (ws_floats >= mmtg_embed_add_bwd_ws(M, D, ntypes) && D % 4 == 0), "embed_add_bwd: workspace of %ld floats required",
                 mmtg_embed_add_bwd_ws(M, D, ntypes));
#define K_(T_, ...)                                                                                                   \
    hipLaunchKernelGGL(embed_dwpe_kernel<T_>, g1, block, 0, s, (const T_*)dh, dwpe, M / T, T, D);                      \
    hipLaunchKernelGGL(embed_dwte_kernel<T_>, g2, block, shm, s, (const T_*)dh, type_ids, dwte, M, D, ntypes, rpb, ws)
    DISPATCH(dtype, K_, 0)
#undef K_
    MMTG_LAUNCH_CHECK("embed_add_bwd");
    if (ws) return mmtg_colsum(MMTG_F32, ws, (long)ntypes * D, cdiv(M, rpb), ntypes * D, dwte, nullptr, 0, stream);
    return MMTG_OK;
}
