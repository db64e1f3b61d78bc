// Causal multi-head self-attention for the GPT-2 decoder (head dim 64),
// flash-style: no T x T score matrix ever reaches HBM.
//
// Forward: one workgroup = (batch, head, 64 queries), 4 waves x 16 queries.
//   K/V tiles of 64 keys staged in LDS; S^T = K Q^T is computed with the KEY on
//   the accumulator rows so a query's softmax statistics need only two wave
//   shuffles, and the probability tile is fed back as the B operand of
//   O^T += V^T P^T straight from registers (no LDS round trip); V^T fragments
//   come from ds_read_b64_tr_b16 on the row-major V tile.
// Backward: one workgroup = (batch, head, key block), 4 waves x KPW keys; dK^T and
//   dV^T live in accumulators across the whole query sweep; S and dP are computed
//   with the key on the lane so P and dS feed dV^T / dK^T as B operands from
//   registers; only dS crosses LDS (once) for dQ.
// Both kernels are templated on the storage type: bf16 (v_mfma_f32_16x16x32_bf16)
// and f32 (v_mfma_f32_16x16x4_f32, the exact parity-gate mode).
#include <stdlib.h>

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

// ======================================================================== forward
// (waves per SIMD the forward kernel is compiled for; 4 = at most 128 VGPRs, four workgroups per CU)
#ifndef ATTN_FWD_WAVES
#define ATTN_FWD_WAVES 4
#endif
template <typename T>
__global__ __launch_bounds__(256, ATTN_FWD_WAVES) void attn_fwd_kernel(const T* __restrict__ qkv, const int* __restrict__ keep,
        T* __restrict__ out, float* __restrict__ lse, int Tn, int nH,
        uint32_t drop_thresh, uint32_t drop_seed, float inv_keep) {
    typedef typename Vec16<T>::type V;
    typedef AT<T> A;
    __shared__ __attribute__((aligned(16))) char sK[64 * A::ROWB];
    __shared__ __attribute__((aligned(16))) char sV[64 * A::ROWB];
    __shared__ int sKeep[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int qb = gridDim.x - 1 - blockIdx.x, h = blockIdx.y, b = blockIdx.z;   // longest (most key blocks) first
    const int D = nH * DH;
    const long ld = 3L * D;
    const T* base = qkv + (long)b * Tn * ld + h * DH;
    const int qi = qb * 64 + wave * 16 + l15;

    V qf[A::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < A::KSTEPS; ++ks) {
        V v = zero16<T>();
        if (qi < Tn) v = *reinterpret_cast<const V*>(base + (long)qi * ld + ks * A::KBE + g * A::EPC);
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) v[e] = (T)((float)v[e] * 0.125f);  // 1/sqrt(64), exact
        qf[ks] = v;
    }

    f32x4 o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    // dropout counter of (b, h, q, key) = ((b nH + h) T + q) T + key, low 32 bits: this lane's row part once
    const uint32_t drow = ((uint32_t)(b * nH + h) * (uint32_t)Tn + (uint32_t)qi) * (uint32_t)Tn;

    // K / V tiles travel global -> registers -> LDS; the NEXT tile's vectors are requested right after the
    // current ones are stored, so their latency hides behind this tile's MFMAs and softmax
    // (bf16 only: the fp32 parity kernel has no registers to spare and loads in place)
    constexpr int NIT = 64 * A::CPR / 256;
    constexpr bool PF = sizeof(T) == 2;
    V kreg[NIT], vreg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int id = tid + 256 * it, key = id / A::CPR, c = id % A::CPR;
        kreg[it] = zero16<T>();
        vreg[it] = zero16<T>();
        if (PF && key < Tn) {
            const T* src = base + (long)key * ld + c * A::EPC;
            kreg[it] = *reinterpret_cast<const V*>(src + D);
            vreg[it] = *reinterpret_cast<const V*>(src + 2 * D);
        }
    }
    for (int jb = 0; jb <= qb; ++jb) {
        const int j0 = jb * 64;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = tid + 256 * it, key = id / A::CPR, c = id % A::CPR;
            if constexpr (!PF) {
                kreg[it] = zero16<T>();
                vreg[it] = zero16<T>();
                if (j0 + key < Tn) {
                    const T* src = base + (long)(j0 + key) * ld + c * A::EPC;
                    kreg[it] = *reinterpret_cast<const V*>(src + D);
                    vreg[it] = *reinterpret_cast<const V*>(src + 2 * D);
                }
            }
            *reinterpret_cast<V*>(sK + off_kc<T>(key, c)) = kreg[it];
            *reinterpret_cast<V*>(sV + off_ks<T>(key, c)) = vreg[it];
        }
        if (PF && jb < qb) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int id = tid + 256 * it, key = id / A::CPR, c = id % A::CPR;
                kreg[it] = zero16<T>();
                vreg[it] = zero16<T>();
                if (j0 + 64 + key < Tn) {
                    const T* src = base + (long)(j0 + 64 + key) * ld + c * A::EPC;
                    kreg[it] = *reinterpret_cast<const V*>(src + D);
                    vreg[it] = *reinterpret_cast<const V*>(src + 2 * D);
                }
            }
        }
        if (tid < 64) sKeep[tid] = (j0 + tid < Tn) ? keep[(long)b * Tn + j0 + tid] : 0;
        __syncthreads();

        // S^T[key][q] for this wave's 16 queries x 64 keys
        f32x4 s_acc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s_acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < A::KSTEPS; ++ks) mma16(ld_kc<T>(sK, kt * 16 + l15, ks, g), qf[ks], s_acc[kt]);
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = kt * 16 + 4 * g + r, kj = j0 + kl;
                const bool valid = kj <= qi && sKeep[kl] != 0;
                const float s = valid ? s_acc[kt][r] : -INFINITY;
                s_acc[kt][r] = s;
                mloc = fmaxf(mloc, s);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : fexp<T>(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = s_acc[kt][r];
                float p = (s == -INFINITY) ? 0.f : fexp<T>(s - m_use);
                rs += p;
                if (drop_thresh) {
                    const int kj = j0 + kt * 16 + 4 * g + r;
                    p *= dropout_scale(drop_seed, drow + (uint32_t)kj, drop_thresh, inv_keep);
                }
                s_acc[kt][r] = p;
            }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o_acc[dt][0] *= alpha; o_acc[dt][1] *= alpha; o_acc[dt][2] *= alpha; o_acc[dt][3] *= alpha;
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q]
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = acc_as_operand(s_acc[2 * s2], s_acc[2 * s2 + 1], bf16());
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    mma16(ld_ks(sV, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, dt * 16, lane, bf16()), pb, o_acc[dt]);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    mma16(ld_ks(sV, kt * 16 + 4 * g, 0, dt * 16, lane, float()), s_acc[kt], o_acc[dt]);
        }
    }

    if (qi < Tn) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        T* dst = out + ((long)b * Tn + qi) * D + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            typedef T T4 __attribute__((ext_vector_type(4)));
            T4 o = {(T)(o_acc[dt][0] * inv), (T)(o_acc[dt][1] * inv), (T)(o_acc[dt][2] * inv), (T)(o_acc[dt][3] * inv)};
            *reinterpret_cast<T4*>(dst + dt * 16 + 4 * g) = o;
        }
        if (g == 0) lse[((long)b * nH + h) * Tn + qi] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
    }
}

// ======================================================================== backward
// delta[m, h] = sum_d dO[m,h,d] * O[m,h,d]   (m = b*T + q; layout [B*T, nH])
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o,
                                                         float* __restrict__ delta, int Tn, int nH, long rows) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows * nH) return;
    const long row = w / nH;
    const int h = (int)(w % nH), lane = threadIdx.x & 63;
    const long idx = row * (long)(nH * DH) + h * DH + lane;
    const float v = wave_sum((float)o[idx] * (float)d_o[idx]);
    if (lane == 0) delta[row * nH + h] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_dq_finish_kernel(const float* __restrict__ dq32, T* __restrict__ dqkv, long rows, int D) {
    const long n = rows * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const int c = (int)(i % D);
        dqkv[r * 3 * D + c] = (T)dq32[i];
    }
}

// NW waves share one key block of KB = 4*AT<T>::KPW keys (256 bf16 / 128 f32), KB/NW keys each.
// bf16 runs NW = 8 (two waves per SIMD: one wave's LDS / softmax latency hides behind the other's
// MFMAs; 64 accumulator registers per wave instead of 128), f32 NW = 4.
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(const T* __restrict__ qkv, const int* __restrict__ keep,
        const T* __restrict__ d_out, const float* __restrict__ lse, const float* __restrict__ delta,
        float* __restrict__ dq32, T* __restrict__ dqkv, float* __restrict__ dbias, int bias_rows, int Tn, int nH, int direct_dq,
        uint32_t drop_thresh, uint32_t drop_seed, float inv_keep, int ablate) {
    typedef typename Vec16<T>::type V;
    typedef AT<T> A;
    constexpr int KB = 4 * A::KPW, KPW = KB / NW, NKT = KPW / 16, NTHR = 64 * NW;
    constexpr int RBS = KB * sizeof(T);   // dS image row bytes (512)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sKr = smem;
    char* sKt = sKr + KB * A::ROWB;
    char* sVr = sKt + KB * A::ROWB;
    char* sQr = sVr + KB * A::ROWB;
    char* sQt = sQr + 32 * A::ROWB;
    char* sOr = sQt + 32 * A::ROWB;
    char* sOt = sOr + 32 * A::ROWB;
    char* sDS = sOt + 32 * A::ROWB;
    float* sLse = reinterpret_cast<float*>(sDS + 32 * RBS);
    float* sDel = sLse + 32;
    int* sKeep = reinterpret_cast<int*>(sDel + 32);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int kb0 = blockIdx.x * KB, h = blockIdx.y, b = blockIdx.z;
    const int D = nH * DH;
    const long ld = 3L * D;
    const T* base = qkv + (long)b * Tn * ld + h * DH;
    const T* dob = d_out + (long)b * Tn * D + h * DH;
    const float scale = 0.125f;

    // stage this block's K (two images) and V (row image) once
    for (int id = tid; id < KB * A::CPR; id += NTHR) {
        const int key = id / A::CPR, c = id % A::CPR;
        V kv = zero16<T>(), vv = zero16<T>();
        if (kb0 + key < Tn) {
            const T* src = base + (long)(kb0 + key) * ld + c * A::EPC;
            kv = *reinterpret_cast<const V*>(src + D);
            vv = *reinterpret_cast<const V*>(src + 2 * D);
        }
        *reinterpret_cast<V*>(sKr + off_kc<T>(key, c)) = kv;
        *reinterpret_cast<V*>(sKt + off_ks<T>(key, c)) = kv;
        *reinterpret_cast<V*>(sVr + off_kc<T>(key, c)) = vv;
    }
    for (int i = tid; i < KB; i += NTHR) sKeep[i] = (kb0 + i < Tn) ? keep[(long)b * Tn + kb0 + i] : 0;

    f32x4 dk_acc[4][NKT], dv_acc[4][NKT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j) { dk_acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv_acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int kw0 = KPW * wave;  // this wave's first key (block-local)
    const uint32_t dbase = (uint32_t)(b * nH + h) * (uint32_t)Tn;   // dropout counter ((b nH + h) T + q) T + key, low 32 bits
    const int nqt = (Tn + 31) / 32;
    float dq_cs = 0.f;           // column sum (over queries) of this lane's dQ column, as stored
    for (int qt = kb0 / 32; qt < nqt; ++qt) {
        const int q0 = qt * 32;
        __syncthreads();
        for (int id = tid; id < 32 * A::CPR; id += NTHR) {
            const int r = id / A::CPR, c = id % A::CPR;
            V qv = zero16<T>(), ov = zero16<T>();
            if (q0 + r < Tn && !((ablate & 8) && qt > kb0 / 32)) {
                qv = *reinterpret_cast<const V*>(base + (long)(q0 + r) * ld + c * A::EPC);
                ov = *reinterpret_cast<const V*>(dob + (long)(q0 + r) * D + c * A::EPC);
            }
            *reinterpret_cast<V*>(sQr + off_kc<T>(r, c)) = qv;
            *reinterpret_cast<V*>(sQt + off_ks<T>(r, c)) = qv;
            *reinterpret_cast<V*>(sOr + off_kc<T>(r, c)) = ov;
            *reinterpret_cast<V*>(sOt + off_ks<T>(r, c)) = ov;
        }
        if (tid < 32) {
            const bool ok = q0 + tid < Tn;
            sLse[tid] = ok ? lse[((long)b * nH + h) * Tn + q0 + tid] : 0.f;
            sDel[tid] = ok ? delta[((long)b * Tn + q0 + tid) * nH + h] : 0.f;
        }
        __syncthreads();

        const bool active = (kb0 + kw0 <= q0 + 31) && (kb0 + kw0 < Tn);
        if (active) {
            // per-row quantities of this lane's 8 query rows (qs, r), hoisted out of the key-tile loop: the
            // element loop below is VALU-bound (exp, dropout hash, masks, dS address), every op counts
            float lse8[2][4], del8[2][4];
            uint32_t drow8[2][4];
            int dsrow8[2][4];
            bool qok8[2][4];
#pragma unroll
            for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = qs * 16 + 4 * g + r;
                    lse8[qs][r] = sLse[ql];
                    del8[qs][r] = sDel[ql];
                    drow8[qs][r] = (dbase + (uint32_t)(q0 + ql)) * (uint32_t)Tn;
                    dsrow8[qs][r] = ql * RBS;
                    qok8[qs][r] = q0 + ql < Tn;
                }
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 pT[2], dsT[2];
                const int kl = kw0 + kt * 16 + l15, key = kb0 + kl;
                const bool kpok = sKeep[kl] != 0;
                const int byte = kl * (int)sizeof(T), bch = byte >> 4, blo = byte & 15;
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, dp_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < A::KSTEPS; ++ks) {
                        mma16(ld_kc<T>(sQr, qs * 16 + l15, ks, g), ld_kc<T>(sKr, kl, ks, g), s_acc);
                        mma16(ld_kc<T>(sOr, qs * 16 + l15, ks, g), ld_kc<T>(sVr, kl, ks, g), dp_acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + qs * 16 + 4 * g + r;
                        const bool valid = key <= q && qok8[qs][r] && kpok;
                        float p = valid ? ((ablate & 1) ? s_acc[r] : fexp<T>(s_acc[r] * scale - lse8[qs][r])) : 0.f;
                        float dp = dp_acc[r];
                        if (drop_thresh) {
                            const float ms = dropout_scale(drop_seed, drow8[qs][r] + (uint32_t)key, drop_thresh, inv_keep);
                            dp *= ms;
                            pT[qs][r] = p * ms;
                        } else {
                            pT[qs][r] = p;
                        }
                        const float ds = p * (dp - del8[qs][r]) * scale;
                        dsT[qs][r] = ds;
                        // dS image [q][key] for the dQ product (chunk swizzle by the row's low 3 bits = (4g + r) & 7)
                        if (!(ablate & 2)) *reinterpret_cast<T*>(sDS + dsrow8[qs][r] + ((bch ^ ((4 * g + r) & 7)) << 4) + blo) = (T)ds;
                    }
                }
                // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
                if constexpr (sizeof(T) == 2) {
                    const bf16x8 pb = acc_as_operand(pT[0], pT[1], bf16());
                    const bf16x8 db = acc_as_operand(dsT[0], dsT[1], bf16());
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        mma16(ld_ks(sOt, 4 * g, 16 + 4 * g, dt * 16, lane, bf16()), pb, dv_acc[dt][kt]);
                        mma16(ld_ks(sQt, 4 * g, 16 + 4 * g, dt * 16, lane, bf16()), db, dk_acc[dt][kt]);
                    }
                } else {
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            mma16(ld_ks(sOt, qs * 16 + 4 * g, 0, dt * 16, lane, float()), pT[qs], dv_acc[dt][kt]);
                            mma16(ld_ks(sQt, qs * 16 + 4 * g, 0, dt * 16, lane, float()), dsT[qs], dk_acc[dt][kt]);
                        }
                }
            }
        }
        __syncthreads();

        // dQ[q][d] = sum_key dS[q][key] K[key][d];  wave w owns d-tile w
        {
            int nact = (ablate & 4) ? 0 : min(KB, min(q0 + 32, Tn) - kb0);   // keys that can be <= some q of this tile
            const int nblk = (nact + A::KBE - 1) / A::KBE;        // k-blocks of 64 bytes of keys
            // 4 waves: wave w -> d-tile w, both 16-row query sub-tiles; 8 waves: one sub-tile each
            const int dt = wave & 3;
            constexpr int QS_PER_WAVE = NW == 8 ? 1 : 2;
#pragma unroll
            for (int qq = 0; qq < QS_PER_WAVE; ++qq) {
                const int qs = NW == 8 ? (wave >> 2) : qq;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int kb = 0; kb < nblk; ++kb) {
                    const int row = qs * 16 + l15;
                    const V af = *reinterpret_cast<const V*>(sDS + row * RBS + (((kb * 4 + g) ^ (row & 7)) << 4));
                    V bf;
                    if constexpr (sizeof(T) == 2) bf = ld_ks(sKt, kb * 32 + 8 * g, kb * 32 + 8 * g + 4, dt * 16, lane, bf16());
                    else bf = ld_ks(sKt, kb * 16 + 4 * g, 0, dt * 16, lane, float());
                    mma16(af, bf, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + qs * 16 + 4 * g + r;
                    if (q < Tn) {
                        if (direct_dq) {
                            dqkv[((long)b * Tn + q) * ld + h * DH + dt * 16 + l15] = (T)acc[r];
                            dq_cs += (float)(T)acc[r];
                        } else atomicAdd(dq32 + ((long)b * Tn + q) * D + h * DH + dt * 16 + l15, acc[r]);
                    }
                }
            }
        }
    }

    // dK, dV of this wave's keys
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int key = kb0 + kw0 + kt * 16 + l15;
        if (key < Tn) {
            T* dst = dqkv + ((long)b * Tn + key) * ld + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                typedef T T4 __attribute__((ext_vector_type(4)));
                T4 kk = {(T)dk_acc[dt][kt][0], (T)dk_acc[dt][kt][1], (T)dk_acc[dt][kt][2], (T)dk_acc[dt][kt][3]};
                T4 vv = {(T)dv_acc[dt][kt][0], (T)dv_acc[dt][kt][1], (T)dv_acc[dt][kt][2], (T)dv_acc[dt][kt][3]};
                *reinterpret_cast<T4*>(dst + D + dt * 16 + 4 * g) = kk;
                *reinterpret_cast<T4*>(dst + 2 * D + dt * 16 + 4 * g) = vv;
            }
        }
    }

    // c_attn bias gradient = column sums of d(qkv) over all tokens, of the values as stored: this
    // workgroup's share is head h's 64 columns of the q, k and v parts (saves a pass over [B*T, 3D]).
    // The waves' partial sums are combined in LDS first: one global atomic per column per workgroup
    // (per-wave global atomics -- 512 adds per address, all at the end of the launch -- cost +36 us).
    if (dbias) {
        float* sB = reinterpret_cast<float*>(smem);      // [3][64], overlays the K image
        __syncthreads();                                  // every wave is done with the staged tiles
        if (tid < 3 * DH) sB[tid] = 0.f;
        __syncthreads();
        if (direct_dq) {         // lane: column (wave & 3) * 16 + l15, rows 4g + r of its query sub-tiles
            dq_cs += __shfl_xor(dq_cs, 16, 64);
            dq_cs += __shfl_xor(dq_cs, 32, 64);
            if (g == 0) atomicAdd(sB + (wave & 3) * 16 + l15, dq_cs);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float sk[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kb0 + kw0 + kt * 16 + l15 < Tn) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sk[r] += (float)(T)dk_acc[dt][kt][r]; sv[r] += (float)(T)dv_acc[dt][kt][r]; }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { sk[r] += __shfl_xor(sk[r], o, 64); sv[r] += __shfl_xor(sv[r], o, 64); }
                if (l15 == 0) {
                    atomicAdd(sB + DH + dt * 16 + 4 * g + r, sk[r]);
                    atomicAdd(sB + 2 * DH + dt * 16 + 4 * g + r, sv[r]);
                }
            }
        }
        __syncthreads();
        if (tid < 3 * DH) {
            const bool mine = direct_dq || tid >= DH;
            const int col = (tid / DH) * D + h * DH + tid % DH;
            // bias_rows: one partial row per workgroup (plain stores; the host sums the rows) -- the
            // atomics onto [3D] from all batch rows at once cost +10 us per launch
            if (bias_rows) dbias[((long)b * gridDim.x + blockIdx.x) * 3 * D + col] = mine ? sB[tid] : 0.f;
            else if (mine) atomicAdd(dbias + col, sB[tid]);
        }
    }
}

template <typename T> size_t bwd_smem_bytes() {
    typedef AT<T> A;
    const int KB = 4 * A::KPW;
    return (size_t)3 * KB * A::ROWB + 4 * 32 * A::ROWB + 32 * KB * sizeof(T) + 64 * sizeof(float) + KB * sizeof(int);
}

inline float inv_keep_of(unsigned thresh) {
    return thresh ? (float)(4294967296.0 / (4294967296.0 - (double)thresh)) : 1.0f;
}

}  // namespace

extern "C" int mmtg_attn_fwd(int dtype, const void* qkv, const int* keep, void* out, float* lse,
                             int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream) {
    MMTG_REQUIRE(dh == DH, "attn_fwd: head dim %d unsupported (built for 64)", dh);
    MMTG_REQUIRE(B > 0 && T > 0 && nH > 0, "attn_fwd: bad sizes");
    MMTG_REQUIRE(qkv && keep && out && lse, "attn_fwd: null pointer");
    MMTG_REQUIRE(MMTG_ALIGNED16(qkv) && MMTG_ALIGNED16(out), "attn_fwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    // algorithmic (causal-half) flops: 2 products x 2*T*T/2*dh per head
    ProfScope prof(MMTG_PROF_ATTN_FWD, s, 2.0 * B * nH * (double)T * T * dh, esz * 4.0 * B * T * nH * dh);
    dim3 grid(cdiv(T, 64), nH, B), block(256);
    const float ik = inv_keep_of(drop_thresh);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, 0, s, (const float*)qkv, keep, (float*)out, lse, T, nH, drop_thresh, drop_seed, ik);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, block, 0, s, (const bf16*)qkv, keep, (bf16*)out, lse, T, nH, drop_thresh, drop_seed, ik);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "attn_fwd: bad dtype");
    MMTG_LAUNCH_CHECK("attn_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, void* stream);

extern "C" int mmtg_attn_bwd(int dtype, const void* qkv, const int* keep, const void* out, const void* dout,
                             const float* lse, float* delta, int delta_ready, float* dq32, void* dqkv, float* dbias, float* dbias_ws,
                             int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream) {
    MMTG_REQUIRE(dh == DH, "attn_bwd: head dim %d unsupported (built for 64)", dh);
    MMTG_REQUIRE(B > 0 && T > 0 && nH > 0, "attn_bwd: bad sizes");
    MMTG_REQUIRE(qkv && keep && out && dout && lse && delta && dq32 && dqkv, "attn_bwd: null pointer");
    MMTG_REQUIRE(MMTG_ALIGNED16(qkv) && MMTG_ALIGNED16(dout) && MMTG_ALIGNED16(dqkv), "attn_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_ATTN_BWD, s, 5.0 * B * nH * (double)T * T * dh, esz * 8.0 * B * T * nH * dh);
    const long rows = (long)B * T;
    const int D = nH * dh;
    const float ik = inv_keep_of(drop_thresh);
    float* const bias_dst = dbias && dbias_ws ? dbias_ws : dbias;
    const int bias_rows = dbias && dbias_ws ? 1 : 0;
    static bool attr_set[2] = {false, false};
    static const int ablate = getenv("MMTG_ATTN_ABLATE") ? atoi(getenv("MMTG_ATTN_ABLATE")) : 0;   // timing experiments only
    if (dtype == MMTG_F32) {
        const int KB = 4 * AT<float>::KPW;
        const int nkb = cdiv(T, KB);
        const size_t shm = bwd_smem_bytes<float>();
        if (!attr_set[0]) {
            if (hipFuncSetAttribute((const void*)attn_bwd_kernel<float, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: cannot raise dynamic LDS to %zu", shm);
            attr_set[0] = true;
        }
        if (!delta_ready) hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(cdiv(rows * nH, 4)), dim3(256), 0, s, (const float*)out, (const float*)dout, delta, T, nH, rows);
        if (nkb > 1) { if (hipMemsetAsync(dq32, 0, rows * D * sizeof(float), s) != hipSuccess) MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: memset failed"); }
        hipLaunchKernelGGL((attn_bwd_kernel<float, 4>), dim3(nkb, nH, B), dim3(256), shm, s, (const float*)qkv, keep, (const float*)dout, lse, delta, dq32, (float*)dqkv, bias_dst, bias_rows, T, nH, nkb == 1, drop_thresh, drop_seed, ik, ablate);
        if (nkb > 1) hipLaunchKernelGGL(attn_dq_finish_kernel<float>, dim3(2048), dim3(256), 0, s, dq32, (float*)dqkv, rows, D);
    } else if (dtype == MMTG_BF16) {
        const int KB = 4 * AT<bf16>::KPW;
        const int nkb = cdiv(T, KB);
        const size_t shm = bwd_smem_bytes<bf16>();
        if (!attr_set[1]) {
            if (hipFuncSetAttribute((const void*)attn_bwd_kernel<bf16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: cannot raise dynamic LDS to %zu", shm);
            attr_set[1] = true;
        }
        if (!delta_ready) hipLaunchKernelGGL(attn_delta_kernel<bf16>, dim3(cdiv(rows * nH, 4)), dim3(256), 0, s, (const bf16*)out, (const bf16*)dout, delta, T, nH, rows);
        if (nkb > 1) { if (hipMemsetAsync(dq32, 0, rows * D * sizeof(float), s) != hipSuccess) MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: memset failed"); }
        hipLaunchKernelGGL((attn_bwd_kernel<bf16, 8>), dim3(nkb, nH, B), dim3(512), shm, s, (const bf16*)qkv, keep, (const bf16*)dout, lse, delta, dq32, (bf16*)dqkv, bias_dst, bias_rows, T, nH, nkb == 1, drop_thresh, drop_seed, ik, ablate);
        if (nkb > 1) hipLaunchKernelGGL(attn_dq_finish_kernel<bf16>, dim3(2048), dim3(256), 0, s, dq32, (bf16*)dqkv, rows, D);
    } else MMTG_FAIL(MMTG_ERR_BAD_ARG, "attn_bwd: bad dtype");
    MMTG_LAUNCH_CHECK("attn_bwd");
    // several key blocks per head: dQ went through the fp32 atomics + finish pass; sum its columns here
    const int nkb_ = cdiv(T, dtype == MMTG_F32 ? 4 * AT<float>::KPW : 4 * AT<bf16>::KPW);
    if (bias_rows) {
        int rc = mmtg_colsum(MMTG_F32, dbias_ws, 3L * D, B * nkb_, 3 * D, dbias, stream);
        if (rc) return rc;
    }
    if (dbias && nkb_ > 1) return mmtg_colsum(dtype, dqkv, 3L * D, (int)rows, D, dbias, stream);
    return MMTG_OK;
}
