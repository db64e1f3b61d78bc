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

#include "attn_common.h"

namespace {

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


// ======================================================================== whole-head kernels (bf16, T <= 256)
// One workgroup holds ALL keys of one (batch, head) in LDS -- 256 rows x 128 B per operand image, two images =
// 64 KB, two workgroups per CU -- so K / V (forward, dQ) or Q / dO (dK, dV) are staged ONCE per head instead of
// once per 64-query block (the tiled kernel above re-stages a head's K/V tiles 1+2+3+4 times at T = 236), and no
// wave ever waits for another's arithmetic: every wave owns whole 16-row tiles of the output and walks the causal
// range of the other operand on its own.  Work is balanced by giving wave w the tiles w and ntile-1-w (their
// causal ranges add up to the same length for every w).
//
// Staging is streamed: the images arrive by LDS-DMA (buffer_load ... lds: global -> LDS, no register round trip,
// 1 KB = 8 rows per wave-instruction) in four chunks of 64 rows, all requested up front; chunk c is readable after
// a counted s_waitcnt + one workgroup barrier and the later chunks keep flying under the arithmetic on the earlier
// ones.  The LDS side of an LDS-DMA is lane-linear, so the image's swizzle is applied to the lane's SOURCE chunk
// (same 128-byte row segment, coalescing unchanged); rows beyond the tensor fall outside the buffer descriptor
// and read as zero.  The per-row scalars (key-padding flags, LSE, delta) come the same way, 4 bytes per lane.
// The number of DMA requests per wave is a compile-time constant, so the waits the compiler adds for the few
// plain loads (this wave's register-resident fragments) are counted ones, not vmcnt(0).
//
// All images use ONE swizzle, chunk ^ vswz(row), which is bank-conflict free both for the ds_read_b128 row reads
// of a 16x16x32 operand and for the ds_read_b64_tr_b16 column reads (16-lane groups / 32-lane halves checked
// exhaustively for 128-byte rows), so a tile that is consumed both ways (K in dQ; Q and dO in dK / dV) is stored
// once.
//
// Backward = two kernels, both without any cross-wave traffic:
//   dK, dV:  Q and dO of the head in LDS; a wave keeps K / V fragments of its 16 keys and the dK^T / dV^T
//            accumulators in registers and sweeps the 32-query blocks at or below the diagonal;
//   dQ:      K and V in LDS; a wave keeps Q / dO fragments of its 16 queries and dQ^T in registers and sweeps the
//            32-key blocks up to the diagonal.
// S and dP are computed by both (7 products instead of 5): the matrix pipe has the time -- the element-wise part
// (exp, dropout hash, masks) bounds these kernels -- and in exchange dS never crosses LDS and the three barriers per
// query tile of the tiled kernel are gone.
//
// Dropout: every workgroup first builds the head's T x T keep-bit matrix in LDS (8 KB: one 32-bit word per query row
// and 32 keys) and the element-wise code only tests bits (a sign-extending bit-field extract and an AND on the float's
// bits; the 1 / (1 - p) scale is folded into the tile's final normalisation or into a constant of the dS formula).
// A word is made by the bitwise Bernoulli construction: 12 xorshift32 steps from one counter hash of (seed, b, h, row,
// word), combined LSB-first by OR / AND according to the 12 binary digits of the keep probability, so each bit is 1 with
// probability round((1 - p) 2^12) / 2^12 -- about 2 vector instructions per decision instead of a 32-bit hash each,
// and the three kernels (forward, dK/dV, dQ) rebuild identical matrices.  (The tiled kernels hash per element; a model
// uses one family throughout.)
constexpr int SM_MAXT = 512;      // whole-head kernels: MT = 256 (8 waves, two workgroups per CU) or 512 (16 waves, one per CU; round 3)
constexpr float LOG2E = 1.4426950408889634f;

// Keep-bit matrix addressing.  MT = 256: dense, 8 words per query row.  MT = 512: only the words at or below the diagonal are
// stored, row-major (band k = rows 32 k .. 32 k + 31 has k + 1 words per row: 17 KB instead of 32 KB -- K / V images of 2 x 64 KB
// leave no room for the dense matrix).  Rows of a band are (k + 1) words apart; a read one word past a row's last word lands in
// the next row (or the two pad words at the end) and is only ever combined with probabilities that are already zero.
template <int MT> __device__ __forceinline__ int mask_row_base(int row) {
    if constexpr (MT == 256) return row * 8;
    else { const int k = row >> 5; return 16 * k * (k + 1) + (row - 32 * k) * (k + 1); }
}
template <int MT> constexpr int mask_words() { return MT == 256 ? 256 * 8 : 16 * 16 * 17 + 2; }

__device__ __forceinline__ bf16x8 ld_row(const char* img, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(img + off_ks<bf16>(row, ks * 4 + g));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const bf16* p, long ld_elems, int nrows) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p), 0, (int)(((long)(nrows - 1) * ld_elems + DH) * 2), 0x00020000);
}
// One chunk (64 rows) of both images = two requests per wave.  Chunk indices beyond the tensor's last chunk re-target
// an existing chunk (same source, same destination -- harmless), so the number of requests is a compile-time constant
// and the compiler's own waits for the few plain loads stay counted ones.
// (a chunk = MT / 4 rows = MT / 32 one-KB blocks per image: 64 rows for MT = 256, 128 for MT = 512; NW waves share them)
template <int NW = 8, int MT = 256>
__device__ __forceinline__ void dma_issue_chunk(char* imgA, __amdgpu_buffer_rsrc_t ra, long lda, char* imgB, __amdgpu_buffer_rsrc_t rb,
                                                long ldb, int c4, int nchunk, int wave, int lane) {
    const int c = c4 < nchunk ? c4 : c4 % nchunk;
#pragma unroll
    for (int i = 0; i < (MT / 32) / NW; ++i) {
        const int blk = (MT / 32) * c + wave + NW * i, row = blk * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ vswz(row);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, imgA + blk * 1024), 16, (int)(((long)row * lda + ch * 8) * 2), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, imgB + blk * 1024), 16, (int)(((long)row * ldb + ch * 8) * 2), 0, 0, 0);
    }
}
// 256 dwords (one per row), element i at src[i * stride]; wave w brings rows 64 (w & 3) .. (waves 4-7 repeat 0-3)
template <int MT = 256>
__device__ __forceinline__ void dma_issue_scalars(void* dst, const void* src, int stride, int nvalid, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)(((long)(nvalid - 1) * stride + 1) * 4), 0x00020000);
    const int ws = wave % (MT / 64), i = 64 * ws + lane;        // MT dwords, one per row; the upper waves repeat the lower ones
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, (char*)dst + 256 * ws), 4, i * stride * 4, 0, 0, 0);
}
// Streaming schedule shared by the three kernels: chunks 0 and 1 are requested in the prologue, chunk c + 2 right
// after the barrier that publishes chunk c.  Before that barrier a wave waits until only its two youngest requests
// (chunk c + 1) are outstanding -- or none, for the last of the four.
template <int NW = 8, int MT = 256>
__device__ __forceinline__ void dma_wait_chunk(int c) {
    if (c >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if ((MT / 32) / NW == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // one block per image and chunk per wave
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                              // two (the half-wave dK / dV builds)
}
// workgroup barrier WITHOUT the release fence of __syncthreads(): the fence would drain every outstanding LDS-DMA
// request (vmcnt(0)) and serialise the streamed chunks; LDS stores of this wave are drained explicitly
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// keep-bit matrix of one head: sMask[row * 8 + w] bit j <-> (query row, key 32 w + j); bit = 1 keeps the element.
// Only words at or below the diagonal of rows < Tn are ever read (a word is one work item: compacted index i ->
// row-major enumeration of the pairs (row, w <= row / 32), 4 x 32 rows + 8 ... per 32-row band).
constexpr int MASK_BITS = 12;          // keep probability in units of 2^-12
template <int NTHR = 512, int MT = 256>
__device__ __forceinline__ void gen_keep_mask(uint32_t* sMask, uint32_t bh, int Tn, uint32_t seed, uint32_t keepq, int tid) {
    // band k (rows 32 k .. 32 k + 31) has k + 1 words per row: items before band k = 32 * k (k + 1) / 2
    const int nband = (Tn + 31) >> 5, total = 16 * nband * (nband + 1);
#pragma unroll 1
    for (int i = tid; i < total; i += NTHR) {
        int k = 0;
        while (16 * (k + 1) * (k + 2) <= i) ++k;             // at most MT / 32 steps
        const int j = i - 16 * k * (k + 1), row = 32 * k + j / (k + 1), w = j % (k + 1);
        if (keepq > 4096u) {
            // ELEMENT mask (MMTG_ATTN_ELEM_MASK, round 6): keepq is the 32-bit drop threshold and bit j of the word is the keep
            // decision of the tiled / split-precision kernels for (query row, key 32 w + j) -- hash(seed, (bh T + row) T + key) >=
            // threshold (dropout_scale) -- so that this backward differentiates the mask the bf16x3f mode's forward applied
            const uint32_t base = (bh * (uint32_t)Tn + (uint32_t)row) * (uint32_t)Tn + 32u * (uint32_t)w;
            uint32_t bits = 0;
#pragma unroll 8
            for (int bit = 0; bit < 32; ++bit) bits |= (hash_u32(seed, base + (uint32_t)bit) >= keepq ? 1u : 0u) << bit;
            sMask[mask_row_base<MT>(row) + w] = bits;
            continue;
        }
        uint32_t x = hash_u32(seed, (bh * (uint32_t)Tn + (uint32_t)row) * (uint32_t)(MT / 32) + (uint32_t)w) | 1u;
        uint32_t acc = 0;
#pragma unroll
        for (int bit = 0; bit < MASK_BITS; ++bit) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            acc = ((keepq >> bit) & 1u) ? (acc | x) : (acc & x);
        }
        sMask[mask_row_base<MT>(row) + w] = acc;
    }
}
// float p kept (bit 1) or zeroed (bit 0); `w` holds the bit at position `pos`
__device__ __forceinline__ float mask_keep(float p, uint32_t w, int pos) {
    const int m = __builtin_amdgcn_sbfe(w, pos, 1);        // 0 or -1
    return __builtin_bit_cast(float, __builtin_bit_cast(int, p) & m);
}

// the same with a wave-uniform bit position: hipcc rewrites the builtin form into v_and + v_cmp + v_cndmask (three VALU per
// element); the extract is pinned here so it stays v_bfe_i32 + v_and
__device__ __forceinline__ float mask_keep_u(float p, uint32_t w, int pos) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "s"(pos));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, p) & m);
}

// ---- forward: running state of one 16-query tile, advanced by one 64-key chunk at a time
struct FwdTile {
    f32x4 o[4];
    float m, l;       // running maximum (in log2 units: s * log2 e) and sum
};

// v_max3_f32 without the NaN-quieting self-maxima hipcc puts in front of every fmaxf operand (scores are never signalling NaNs)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// DIAG: the chunk touches the causal diagonal of the tile (its last chunk): per-key-tile skips and the causal comparison.
// Every earlier chunk runs the straight-line form (all four key tiles, no comparison, no conditional register copies).
template <bool DROP, int MT = 256, bool DIAG = true>
__device__ __forceinline__ void fwd_small_chunk(FwdTile& st, const char* sK, const char* sV, const float* sBias, const uint32_t* sMask,
                                                const bf16x8 (&qf)[2], int t, int j0, int lane) {
    const int g = lane >> 4, l15 = lane & 15;
    const int qi = 16 * t + l15, qlast = 16 * t + 15;
    f32x4 s_acc[4];
    float mloc = -INFINITY;
    // per 16-key tile: S^T = K Q^T, then the masks -- key padding as an additive 0 / -inf per key (one 16-byte LDS
    // read per four keys), the causal comparison only where the tile touches the diagonal; tiles wholly above the
    // diagonal (wave-uniform) are skipped altogether
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        s_acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!DIAG || j0 + 16 * kt <= qlast) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) mma16(ld_row(sK, j0 + kt * 16 + l15, ks, g), qf[ks], s_acc[kt]);
            const int k0 = j0 + kt * 16 + 4 * g;
            const f32x4 kb = *reinterpret_cast<const f32x4*>(sBias + k0);
            if (DIAG && j0 + 16 * kt + 15 > 16 * t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[kt][r] = (k0 + r <= qi) ? s_acc[kt][r] * LOG2E + kb[r] : -INFINITY;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[kt][r] = s_acc[kt][r] * LOG2E + kb[r];
            }
            mloc = max3f(max3f(mloc, s_acc[kt][0], s_acc[kt][1]), s_acc[kt][2], s_acc[kt][3]);
        }
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(st.m, mloc);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(st.m - m_use);          // st.m = -inf -> 0
    float rs = 0.f;
    uint32_t w0 = 0, w1 = 0;
    if constexpr (DROP) {       // this query row's keep bits of the chunk's 64 keys, pre-shifted to the lane's 4 g
        if constexpr (MT == 256) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(sMask + qi * 8 + (j0 >> 5));
            w0 = w[0] >> (4 * g);
            w1 = w[1] >> (4 * g);
        } else {
            const uint32_t* wp = sMask + mask_row_base<MT>(qi) + (j0 >> 5);
            w0 = wp[0] >> (4 * g);
            w1 = wp[1] >> (4 * g);
        }
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (!DIAG || j0 + 16 * kt <= qlast) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s_acc[kt][r] - m_use);   // masked: exp2(-inf) = 0
                rs += p;
                if constexpr (DROP) p = mask_keep_u(p, kt < 2 ? w0 : w1, 16 * (kt & 1) + r);
                s_acc[kt][r] = p;
            }
        }
    }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    st.l = st.l * alpha + rs;
    st.m = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        st.o[dt][0] *= alpha; st.o[dt][1] *= alpha; st.o[dt][2] *= alpha; st.o[dt][3] *= alpha;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (!DIAG || j0 + 32 * s2 <= qlast) {
            const bf16x8 pb = acc_as_operand(s_acc[2 * s2], s_acc[2 * s2 + 1], bf16());
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                mma16(ld_ks(sV, j0 + 32 * s2 + 4 * g, j0 + 32 * s2 + 16 + 4 * g, dt * 16, lane, bf16()), pb, st.o[dt]);
        }
    }
}
// one 64-key chunk of tile t: the straight-line form unless the chunk reaches the tile's diagonal
template <bool DROP, int MT = 256>
__device__ __forceinline__ void fwd_small_chunk_any(FwdTile& st, const char* sK, const char* sV, const float* sBias, const uint32_t* sMask,
                                                    const bf16x8 (&qf)[2], int t, int j0, int lane) {
    if (j0 + 63 <= 16 * t) fwd_small_chunk<DROP, MT, false>(st, sK, sV, sBias, sMask, qf, t, j0, lane);
    else fwd_small_chunk<DROP, MT, true>(st, sK, sV, sBias, sMask, qf, t, j0, lane);
}

__device__ __forceinline__ void fwd_small_store(const FwdTile& st, int t, int lane, int b, int h, int Tn, int nH, float inv_keep,
                                                bf16* __restrict__ out, float* __restrict__ lse) {
    const int g = lane >> 4, l15 = lane & 15, qi = 16 * t + l15, D = nH * DH;
    if (qi < Tn) {
        const float inv = st.l > 0.f ? inv_keep / st.l : 0.f;          // dropout's 1 / (1 - p) rides on the normalisation
        bf16* dst = out + ((long)b * Tn + qi) * D + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 o = {(bf16)(st.o[dt][0] * inv), (bf16)(st.o[dt][1] * inv), (bf16)(st.o[dt][2] * inv), (bf16)(st.o[dt][3] * inv)};
            *reinterpret_cast<bf16x4*>(dst + dt * 16 + 4 * g) = o;
        }
        if (g == 0) lse[((long)b * nH + h) * Tn + qi] = st.l > 0.f ? st.m * (1.0f / LOG2E) + logf(st.l) : -INFINITY;
    }
}

template <bool DROP, int MT = 256>
__global__ __launch_bounds__(2 * MT, 4) void attn_fwd_small_kernel(const bf16* __restrict__ qkv, const int* __restrict__ keep,
        bf16* __restrict__ out, float* __restrict__ lse, int Tn, int nH,
        uint32_t keep16, uint32_t drop_seed, float inv_keep, unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    ts[0] = __builtin_amdgcn_s_memrealtime();        // (unconditional: testing `trace` first would put a kernel-argument round trip in front of everything)
    constexpr int NW = MT / 32, CROWS = MT / 4;
    const int nchunk = (Tn + CROWS - 1) / CROWS, rows_pad = nchunk * CROWS;
    char* sK = smem;
    char* sV = sK + rows_pad * 128;
    int* sKeep = reinterpret_cast<int*>(sV + rows_pad * 128);            // [MT] flags
    float* sBias = reinterpret_cast<float*>(sKeep + MT);                // [MT] 0 / -inf
    uint32_t* sMask = reinterpret_cast<uint32_t*>(sBias + MT + MT);     // keep bits (after one spare [MT])
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x % nH, b = blockIdx.x / nH;
    const int D = nH * DH;
    const long ld = 3L * D;
    const bf16* base = qkv + (long)b * Tn * ld + h * DH;
    const int ntile = (Tn + 15) >> 4, npair = (ntile + 1) >> 1;
    const int tA = wave, tB = ntile - 1 - wave;          // tB >= tA for wave < npair
    const bool work = wave < npair;
    // this wave's query fragments (B operand of S^T = K Q^T): plain loads, issued first
    bf16x8 qfA[2], qfB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qfA[ks] = zero16<bf16>();
        qfB[ks] = zero16<bf16>();
        if (work) {
            const int qa = 16 * tA + l15, qb = 16 * tB + l15;
            if (qa < Tn) qfA[ks] = *reinterpret_cast<const bf16x8*>(base + (long)qa * ld + ks * 32 + g * 8);
            if (qb < Tn) qfB[ks] = *reinterpret_cast<const bf16x8*>(base + (long)qb * ld + ks * 32 + g * 8);
        }
    }
    const __amdgpu_buffer_rsrc_t rk = rows_rsrc(base + D, ld, Tn), rv = rows_rsrc(base + 2 * D, ld, Tn);
    dma_issue_scalars<MT>(sKeep, keep + (long)b * Tn, 1, Tn, wave, lane);
    dma_issue_chunk<NW, MT>(sK, rk, ld, sV, rv, ld, 0, nchunk, wave, lane);
    dma_issue_chunk<NW, MT>(sK, rk, ld, sV, rv, ld, 1, nchunk, wave, lane);
    if (trace) ts[1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (DROP) gen_keep_mask<2 * MT, MT>(sMask, (uint32_t)(b * nH + h), Tn, drop_seed, keep16, tid);
    // 1/sqrt(64) pre-scale of the queries (exact in bf16); also the first use of the plain loads (a counted wait here)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qfA[ks][e] = (bf16)((float)qfA[ks][e] * 0.125f);
            qfB[ks][e] = (bf16)((float)qfB[ks][e] * 0.125f);
        }
    FwdTile st;
#pragma unroll
    for (int i = 0; i < 4; ++i) st.o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.m = -INFINITY;
    st.l = 0.f;
    // the longer tile (B) consumes the key chunks as they land; the shorter one (A) runs afterwards on resident data
#pragma unroll 1
    for (int jb = 0; jb < nchunk; ++jb) {
        dma_wait_chunk<NW, MT>(jb);
        raw_barrier();
        if (jb < 2) dma_issue_chunk<NW, MT>(sK, rk, ld, sV, rv, ld, jb + 2, nchunk, wave, lane);
        if (jb == 0) {
            if (tid < MT) sBias[tid] = sKeep[tid] != 0 ? 0.f : -INFINITY;        // key-padding flags -> additive bias
            raw_barrier();
            if (trace) ts[2] = __builtin_amdgcn_s_memrealtime();
        }
#pragma unroll 1
        for (int j0 = jb * CROWS; j0 < (jb + 1) * CROWS; j0 += 64)
            if (work && j0 <= 16 * tB + 15) fwd_small_chunk_any<DROP, MT>(st, sK, sV, sBias, sMask, qfB, tB, j0, lane);
    }
    if (trace) ts[3] = __builtin_amdgcn_s_memrealtime();
    if (work) fwd_small_store(st, tB, lane, b, h, Tn, nH, inv_keep, out, lse);
    if (trace) ts[4] = __builtin_amdgcn_s_memrealtime();
    if (work && tA != tB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st.o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.m = -INFINITY;
        st.l = 0.f;
#pragma unroll 1
        for (int j0 = 0; j0 <= 16 * tA + 15; j0 += 64) fwd_small_chunk_any<DROP, MT>(st, sK, sV, sBias, sMask, qfA, tA, j0, lane);
        fwd_small_store(st, tA, lane, b, h, Tn, nH, inv_keep, out, lse);
    }
    if (trace && lane == 0) {
        ts[5] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* r = trace + ((long)blockIdx.x * NW + wave) * 8;
        r[0] = ts[0]; r[1] = ts[1]; r[2] = ts[2]; r[3] = ts[3]; r[4] = ts[4]; r[5] = ts[5];
        r[6] = __builtin_amdgcn_s_getreg((4 << 11) | 20);    // XCC_ID
        r[7] = 1;
    }
}

// ---- backward, dK / dV kernel: the 16 keys of tile t (K, V fragments in registers), one 32-query block
struct BwdKeys {
    f32x4 dk[4], dv[4];
};
// DIAG: the block holds the tile's diagonal (the first block of a tile's sweep); every later block is wholly below it.
// NQS 16-query sub-blocks per call: 2 (one 32-query block) or 4 (two blocks back to back below the diagonal -- twice the
// independent work between the dependent steps LDS read -> product -> element-wise -> product of a two-waves-per-SIMD kernel)
template <bool DROP, int MT = 256, bool DIAG = true, int NQS = 2>
__device__ __forceinline__ void bwd_small_keys_block(BwdKeys& st, const char* sQ, const char* sO, const float* sLse, const float* sDel,
                                                     const uint32_t* sMask, const bf16x8 (&kf)[2], const bf16x8 (&vf)[2],
                                                     int t, int q0, int lane, float ik_scale) {
    const int g = lane >> 4, l15 = lane & 15;
    const int key = 16 * t + l15;
    const float c1 = 0.125f * LOG2E;
    f32x4 pT[NQS], dsT[NQS];
#pragma unroll
    for (int qs = 0; qs < NQS; ++qs) {
        pT[qs] = f32x4{0.f, 0.f, 0.f, 0.f};
        dsT[qs] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!DIAG || q0 + 16 * qs + 15 >= 16 * t) { // wave-uniform: a query sub-block wholly above the diagonal contributes nothing
            f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, dp_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                mma16(ld_row(sQ, q0 + qs * 16 + l15, ks, g), kf[ks], s_acc);
                mma16(ld_row(sO, q0 + qs * 16 + l15, ks, g), vf[ks], dp_acc);
            }
            const int qr = q0 + qs * 16 + 4 * g;
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + qr);       // LSE * log2 e per query row
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + qr);       // delta / sqrt(64)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = s_acc[r] * c1 - l4[r];            // (a padded key's column is zeroed at the store, not here)
                if constexpr (DIAG) x = key <= qr + r ? x : -INFINITY;
                const float p = __builtin_amdgcn_exp2f(x);
                float dp = dp_acc[r];
                if constexpr (DROP) {
                    const uint32_t w = sMask[mask_row_base<MT>(qr + r) + (key >> 5)];
                    dp = mask_keep(dp, w, key & 31);
                    pT[qs][r] = mask_keep(p, w, key & 31);                     // (x 1/(1-p) at the end, on dV)
                } else {
                    pT[qs][r] = p;
                }
                dsT[qs][r] = p * (dp * ik_scale - d4[r]);                      // ik_scale = (1/(1-p)) / sqrt(64)
            }
        }
    }
    // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
    for (int h2 = 0; h2 < NQS / 2; ++h2) {
        const bf16x8 pb = acc_as_operand(pT[2 * h2], pT[2 * h2 + 1], bf16());
        const bf16x8 db = acc_as_operand(dsT[2 * h2], dsT[2 * h2 + 1], bf16());
        const int qa = q0 + 32 * h2 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            mma16(ld_ks(sO, qa, qa + 16, dt * 16, lane, bf16()), pb, st.dv[dt]);
            mma16(ld_ks(sQ, qa, qa + 16, dt * 16, lane, bf16()), db, st.dk[dt]);
        }
    }
}
// the sweep of key tile t over the resident query blocks from block `qb` (below the diagonal) on: pairs, then a last single
template <bool DROP, int MT, bool PAIRS>
__device__ __forceinline__ void bwd_small_keys_rest(BwdKeys& st, const char* sQ, const char* sO, const float* sLse, const float* sDel,
                                                    const uint32_t* sMask, const bf16x8 (&kf)[2], const bf16x8 (&vf)[2],
                                                    int t, int qb, int nqb, int lane, float ik_scale) {
    if constexpr (PAIRS) {
#pragma unroll 1
        for (; qb + 1 < nqb; qb += 2) bwd_small_keys_block<DROP, MT, false, 4>(st, sQ, sO, sLse, sDel, sMask, kf, vf, t, 32 * qb, lane, ik_scale);
        if (qb < nqb) bwd_small_keys_block<DROP, MT, false, 2>(st, sQ, sO, sLse, sDel, sMask, kf, vf, t, 32 * qb, lane, ik_scale);
    } else {
#pragma unroll 1
        for (; qb < nqb; ++qb) bwd_small_keys_block<DROP, MT, false, 2>(st, sQ, sO, sLse, sDel, sMask, kf, vf, t, 32 * qb, lane, ik_scale);
    }
}
// sum over the 16 lanes of a row (the 16 keys / queries a tile's lanes own) with DPP rotations: four full-rate VALU adds, no
// LDS traffic (the __shfl_xor form is a ds_bpermute per step); every lane of the row ends up with the row's sum
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_mov<0x128>(x);        // row_ror:8
    x += dpp_mov<0x124>(x);        // row_ror:4
    x += dpp_mov<0x122>(x);        // row_ror:2
    x += dpp_mov<0x121>(x);        // row_ror:1
    return x;
}
// c_attn bias gradient, per wave: the column sums of a tile's values AS STORED (v[dt][r] = channel 16 dt + 4 g + r of the
// lane's key / query, 0 beyond the tensor) folded over the tile's 16 rows; lane l15 = 4 dt + r keeps the sum of channel
// (l15 >> 2) * 16 + 4 g + (l15 & 3) -- one register per lane carries the wave's running sum over all its tiles (round 4: the
// 8-wave builds reduced per tile with 128 ds_bpermutes, the 4-wave build carried 32 accumulators and flushed them through
// eight barrier-separated phases; now every wave writes ONE row of partial sums at the end and the rows are added in wave
// order -- bit-identical run to run, one barrier).
__device__ __forceinline__ float bias_fold16(const f32x4 (&v)[4], int l15) {
    float c = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sm = row_sum16(v[dt][r]);
            if (l15 == dt * 4 + r) c = sm;
        }
    return c;
}
__device__ __forceinline__ void bwd_small_keys_store(const BwdKeys& st, int t, int lane, int b, int h, int Tn, int nH, float inv_keep,
                                                     bool kpok, bf16* __restrict__ dqkv, bool bias, float& ck, float& cv) {
    const int g = lane >> 4, l15 = lane & 15, key = 16 * t + l15, D = nH * DH;
    const long ld = 3L * D;
    const bool kin = key < Tn;
    // a padded key never receives attention: its probabilities are zero for every query, hence dK = dV = 0.  Each lane
    // owns one key's column of both products, so the sweep runs unmasked and the column is zeroed here.
    const float kz = kpok ? 1.f : 0.f, vz = kpok ? inv_keep : 0.f;
    bf16x4 kk[4], vv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        kk[dt] = bf16x4{(bf16)(st.dk[dt][0] * kz), (bf16)(st.dk[dt][1] * kz), (bf16)(st.dk[dt][2] * kz), (bf16)(st.dk[dt][3] * kz)};
        vv[dt] = bf16x4{(bf16)(st.dv[dt][0] * vz), (bf16)(st.dv[dt][1] * vz), (bf16)(st.dv[dt][2] * vz), (bf16)(st.dv[dt][3] * vz)};
    }
    if (kin) {
        bf16* dst = dqkv + ((long)b * Tn + key) * ld + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *reinterpret_cast<bf16x4*>(dst + D + dt * 16 + 4 * g) = kk[dt];
            *reinterpret_cast<bf16x4*>(dst + 2 * D + dt * 16 + 4 * g) = vv[dt];
        }
    }
    if (bias) {        // c_attn bias gradient: column sums over keys of the values as stored
        f32x4 fk[4], fv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fk[dt][r] = kin ? (float)kk[dt][r] : 0.f;
                fv[dt][r] = kin ? (float)vv[dt][r] : 0.f;
            }
        ck += bias_fold16(fk, l15);
        cv += bias_fold16(fv, l15);
    }
}

// NW waves per workgroup: 8 (two workgroups = 16 waves per CU, <= 128 VGPRs) or 4 (two workgroups = 8 waves per CU, up to
// 256 VGPRs, every wave owns two tile pairs)
template <bool DROP, int NW, int MT = 256>
__global__ __launch_bounds__(64 * NW, (NW / (MT == 256 ? 2 : 4))) void attn_bwd_small_kv_kernel(const bf16* __restrict__ qkv, const int* __restrict__ keep,
        const bf16* __restrict__ d_out, const float* __restrict__ lse, const float* __restrict__ delta,
        bf16* __restrict__ dqkv, float* __restrict__ dbias, int bias_rows, int Tn, int nH,
        uint32_t keep16, uint32_t drop_seed, float inv_keep, unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    ts[0] = __builtin_amdgcn_s_memrealtime();        // (unconditional: testing `trace` first would put a kernel-argument round trip in front of everything)
    constexpr int CROWS = MT / 4, PPW = (MT / 32) / NW;            // rows per DMA chunk; tile pairs per wave
    const int nchunk = (Tn + CROWS - 1) / CROWS, rows_pad = nchunk * CROWS;
    char* sQ = smem;
    char* sO = sQ + rows_pad * 128;
    float* sLse = reinterpret_cast<float*>(sO + rows_pad * 128);    // [MT]
    float* sDel = sLse + MT;                                        // [MT]
    int* sKeep = reinterpret_cast<int*>(sDel + MT);                 // [MT]
    uint32_t* sMask = reinterpret_cast<uint32_t*>(sKeep + 2 * MT);  // keep bits (behind a spare [MT] slot)
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x % nH, b = blockIdx.x / nH;
    const int D = nH * DH;
    const long ld = 3L * D;
    const bf16* base = qkv + (long)b * Tn * ld + h * DH;
    const bf16* dob = d_out + (long)b * Tn * D + h * DH;
    const int ntile = (Tn + 15) >> 4, npair = (ntile + 1) >> 1;
    // K / V fragments of a key tile (B operands: lane holds row key = 16 t + l15, columns 32 ks + 8 g ..)
    bf16x8 kf[2], vf[2];
    auto load_frags = [&](int t) {
        const int key = 16 * t + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = vf[ks] = zero16<bf16>();
            if (key < Tn) {
                kf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)key * ld + D + ks * 32 + g * 8);
                vf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)key * ld + 2 * D + ks * 32 + g * 8);
            }
        }
    };
    if (wave < npair) load_frags(wave);
    else { kf[0] = kf[1] = vf[0] = vf[1] = zero16<bf16>(); }
    const __amdgpu_buffer_rsrc_t rq = rows_rsrc(base, ld, Tn), ro = rows_rsrc(dob, D, Tn);
    dma_issue_scalars<MT>(sKeep, keep + (long)b * Tn, 1, Tn, wave, lane);
    dma_issue_scalars<MT>(sLse, lse + ((long)b * nH + h) * Tn, 1, Tn, wave, lane);
    dma_issue_scalars<MT>(sDel, delta + (long)b * Tn * nH + h, nH, Tn, wave, lane);
    dma_issue_chunk<NW, MT>(sQ, rq, ld, sO, ro, D, 0, nchunk, wave, lane);
    dma_issue_chunk<NW, MT>(sQ, rq, ld, sO, ro, D, 1, nchunk, wave, lane);
    if (trace) ts[1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (DROP) gen_keep_mask<64 * NW, MT>(sMask, (uint32_t)(b * nH + h), Tn, drop_seed, keep16, tid);
    // first use of the plain loads: a counted wait here instead of a full drain inside the loop
    asm volatile("" :: "v"(kf[0]), "v"(kf[1]), "v"(vf[0]), "v"(vf[1]));
    BwdKeys st;
#pragma unroll
    for (int i = 0; i < 4; ++i) st.dk[i] = st.dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool PAIRS = PPW == 2;          // the half-wave build has the registers for two query blocks per call
    float ck = 0.f, cv = 0.f;                 // the wave's running bias-gradient sums (bias_fold16)
    const int nqb = (Tn + 31) >> 5;
    const float ik_scale = inv_keep * 0.125f;
    // The wave's first tile (low keys: the longest sweep) takes the query blocks from its diagonal on as they land;
    // every later tile runs on resident data.
    {
        const int tA = wave;
        const bool work = wave < npair;
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            dma_wait_chunk<NW, MT>(c);
            raw_barrier();
            if (c < 2) dma_issue_chunk<NW, MT>(sQ, rq, ld, sO, ro, D, c + 2, nchunk, wave, lane);
            if (c == 0) {
                for (int i = tid; i < MT; i += 64 * NW) { sLse[i] *= LOG2E; sDel[i] *= 0.125f; }
                raw_barrier();
                if (trace) ts[2] = __builtin_amdgcn_s_memrealtime();
            }
#pragma unroll 1
            for (int qb = c * (CROWS / 32); qb < (c + 1) * (CROWS / 32) && qb < nqb; ++qb)
                if (work && qb >= ((16 * tA) >> 5)) {
                    if (qb == ((16 * tA) >> 5)) bwd_small_keys_block<DROP, MT, true>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tA, 32 * qb, lane, ik_scale);
                    else if (PAIRS && !(qb & 1) && qb + 1 < nqb) {      // both blocks of this chunk lie below the diagonal
                        bwd_small_keys_block<DROP, MT, false, 4>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tA, 32 * qb, lane, ik_scale);
                        ++qb;
                    } else bwd_small_keys_block<DROP, MT, false>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tA, 32 * qb, lane, ik_scale);
                }
        }
    }
    if (trace) ts[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int pi = 0; pi < PPW; ++pi) {
        const int p = wave + NW * pi;
        if (p >= npair) break;
        const int tA = p, tB = ntile - 1 - p;
        if (pi > 0) {
            load_frags(tA);
#pragma unroll
            for (int i = 0; i < 4; ++i) st.dk[i] = st.dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            bwd_small_keys_block<DROP, MT, true>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tA, 32 * ((16 * tA) >> 5), lane, ik_scale);
            bwd_small_keys_rest<DROP, MT, PAIRS>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tA, ((16 * tA) >> 5) + 1, nqb, lane, ik_scale);
        }
        if (tA != tB) load_frags(tB);
        bwd_small_keys_store(st, tA, lane, b, h, Tn, nH, inv_keep, sKeep[(16 * tA + l15) & (MT - 1)] != 0, dqkv, dbias != nullptr, ck, cv);
        if (tA != tB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st.dk[i] = st.dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            bwd_small_keys_block<DROP, MT, true>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tB, 32 * ((16 * tB) >> 5), lane, ik_scale);
            bwd_small_keys_rest<DROP, MT, PAIRS>(st, sQ, sO, sLse, sDel, sMask, kf, vf, tB, ((16 * tB) >> 5) + 1, nqb, lane, ik_scale);
            bwd_small_keys_store(st, tB, lane, b, h, Tn, nH, inv_keep, sKeep[(16 * tB + l15) & (MT - 1)] != 0, dqkv, dbias != nullptr, ck, cv);
        }
    }
    if (trace) ts[4] = __builtin_amdgcn_s_memrealtime();
    if (dbias) {
        // every wave's row of partial sums -> sBw[wave][128] (K channels, then V; over the first bytes of the Q image once
        // every wave is done with it), then the rows added in wave order
        __syncthreads();
        float* sBw = reinterpret_cast<float*>(smem);
        const int ch = (l15 >> 2) * 16 + 4 * g + (l15 & 3);
        sBw[wave * 128 + ch] = ck;
        sBw[wave * 128 + DH + ch] = cv;
        __syncthreads();
        // this kernel owns the K and V parts of head h's columns: one partial row per batch row
        // (bias_rows: plain stores into row b of the scratch; otherwise atomics onto the gradient)
        if (tid < 2 * DH) {
            float sm = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sm += sBw[w * 128 + tid];
            const int col = (1 + tid / DH) * D + h * DH + tid % DH;
            if (bias_rows) dbias[(long)b * 3 * D + col] = sm;
            else atomicAdd(dbias + col, sm);
        }
    }
    if (trace && lane == 0) {
        ts[5] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* r = trace + ((long)blockIdx.x * 16 + wave) * 8;
        r[0] = ts[0]; r[1] = ts[1]; r[2] = ts[2]; r[3] = ts[3]; r[4] = ts[4]; r[5] = ts[5];
        r[6] = __builtin_amdgcn_s_getreg((4 << 11) | 20);    // XCC_ID
        r[7] = 1;
    }
}

// ---- backward, dK / dV kernel, 32 keys per wave (round 4): a wave owns a UNIT of two adjacent key tiles (keys 32 u .. 32 u + 31)
// and sweeps the 32-query blocks u .. from their common diagonal block on.  The Q / dO row fragments of a query sub-block and
// the transposed Q / dO fragments of a block are read from LDS ONCE for both tiles -- half the LDS traffic per product of the
// 16-key form (whose two-waves-per-SIMD sweep ran the LDS pipe at about half its bandwidth besides the vector work) and half
// the per-block overhead.  Units pair up like tiles did (u, nunit - 1 - u: 9 blocks of 32 x 32 per wave at T = 236), NW = MT / 64
// waves per workgroup, up to 256 VGPRs.
template <bool DROP, int MT, bool DIAG>
__device__ __forceinline__ void bwd_small_keys2_block(BwdKeys (&st)[2], const char* sQ, const char* sO, const float* sLse, const float* sDel,
                                                      const uint32_t* sMask, const bf16x8 (&kf)[2][2], const bf16x8 (&vf)[2][2],
                                                      int u, int q0, int lane, float ik_scale) {
    const int g = lane >> 4, l15 = lane & 15;
    const float c1 = 0.125f * LOG2E;
    f32x4 pT[2][2], dsT[2][2];          // [key tile][query sub-block]
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        bf16x8 qrow[2], orow[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qrow[ks] = ld_row(sQ, q0 + qs * 16 + l15, ks, g);
            orow[ks] = ld_row(sO, q0 + qs * 16 + l15, ks, g);
        }
        const int qr = q0 + qs * 16 + 4 * g;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + qr);       // LSE * log2 e per query row
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + qr);       // delta / sqrt(64)
        uint32_t w[4] = {0, 0, 0, 0};
        if constexpr (DROP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = sMask[mask_row_base<MT>(qr + r) + u];       // both tiles' keys live in word u
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pT[j][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
            dsT[j][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DIAG && j == 1 && qs == 0) continue;        // the diagonal block's upper-right quarter: wholly above the diagonal
            f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, dp_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                mma16(qrow[ks], kf[j][ks], s_acc);
                mma16(orow[ks], vf[j][ks], dp_acc);
            }
            const int key = 32 * u + 16 * j + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = s_acc[r] * c1 - l4[r];            // (a padded key's column is zeroed at the store, not here)
                if (DIAG && qs == j) x = key <= qr + r ? x : -INFINITY;
                const float p = __builtin_amdgcn_exp2f(x);
                float dp = dp_acc[r];
                if constexpr (DROP) {
                    dp = mask_keep(dp, w[r], 16 * j + l15);
                    pT[j][qs][r] = mask_keep(p, w[r], 16 * j + l15);           // (x 1/(1-p) at the end, on dV)
                } else {
                    pT[j][qs][r] = p;
                }
                dsT[j][qs][r] = p * (dp * ik_scale - d4[r]);                   // ik_scale = (1/(1-p)) / sqrt(64)
            }
        }
    }
    // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
    bf16x8 pb[2], db[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        pb[j] = acc_as_operand(pT[j][0], pT[j][1], bf16());
        db[j] = acc_as_operand(dsT[j][0], dsT[j][1], bf16());
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 ao = ld_ks(sO, q0 + 4 * g, q0 + 16 + 4 * g, dt * 16, lane, bf16());
        const bf16x8 aq = ld_ks(sQ, q0 + 4 * g, q0 + 16 + 4 * g, dt * 16, lane, bf16());
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            mma16(ao, pb[j], st[j].dv[dt]);
            mma16(aq, db[j], st[j].dk[dt]);
        }
    }
}
// both tiles of a unit: rounded, stored, and their column sums (as stored) folded into the wave's running bias sums
__device__ __forceinline__ void bwd_small_keys2_store(const BwdKeys (&st)[2], int u, int lane, int b, int h, int Tn, int nH, float inv_keep,
                                                      const int* sKeep, int mt_mask, bf16* __restrict__ dqkv, bool bias, float& ck, float& cv) {
    const int g = lane >> 4, l15 = lane & 15, D = nH * DH;
    const long ld = 3L * D;
    f32x4 fk[4], fv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) fk[dt] = fv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = 32 * u + 16 * j + l15;
        const bool kin = key < Tn, kpok = sKeep[key & mt_mask] != 0;
        // a padded key never receives attention: its probabilities are zero for every query, hence dK = dV = 0.  Each lane
        // owns one key's column of both products, so the sweep runs unmasked and the column is zeroed here.
        const float kz = kpok ? 1.f : 0.f, vz = kpok ? inv_keep : 0.f;
        bf16x4 kk[4], vv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            kk[dt] = bf16x4{(bf16)(st[j].dk[dt][0] * kz), (bf16)(st[j].dk[dt][1] * kz), (bf16)(st[j].dk[dt][2] * kz), (bf16)(st[j].dk[dt][3] * kz)};
            vv[dt] = bf16x4{(bf16)(st[j].dv[dt][0] * vz), (bf16)(st[j].dv[dt][1] * vz), (bf16)(st[j].dv[dt][2] * vz), (bf16)(st[j].dv[dt][3] * vz)};
        }
        if (kin) {
            bf16* dst = dqkv + ((long)b * Tn + key) * ld + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<bf16x4*>(dst + D + dt * 16 + 4 * g) = kk[dt];
                *reinterpret_cast<bf16x4*>(dst + 2 * D + dt * 16 + 4 * g) = vv[dt];
            }
        }
        if (bias) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    fk[dt][r] += kin ? (float)kk[dt][r] : 0.f;
                    fv[dt][r] += kin ? (float)vv[dt][r] : 0.f;
                }
        }
    }
    if (bias) {
        ck += bias_fold16(fk, l15);
        cv += bias_fold16(fv, l15);
    }
}

template <bool DROP, int MT = 256>
__global__ __launch_bounds__(MT, (MT == 256 ? 2 : 1)) void attn_bwd_small_kv2_kernel(const bf16* __restrict__ qkv, const int* __restrict__ keep,
        const bf16* __restrict__ d_out, const float* __restrict__ lse, const float* __restrict__ delta,
        bf16* __restrict__ dqkv, float* __restrict__ dbias, int bias_rows, int Tn, int nH,
        uint32_t keep16, uint32_t drop_seed, float inv_keep, unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    ts[0] = __builtin_amdgcn_s_memrealtime();
    constexpr int NW = MT / 64, CROWS = MT / 4;
    const int nchunk = (Tn + CROWS - 1) / CROWS, rows_pad = nchunk * CROWS;
    char* sQ = smem;
    char* sO = sQ + rows_pad * 128;
    float* sLse = reinterpret_cast<float*>(sO + rows_pad * 128);    // [MT]
    float* sDel = sLse + MT;                                        // [MT]
    int* sKeep = reinterpret_cast<int*>(sDel + MT);                 // [MT]
    uint32_t* sMask = reinterpret_cast<uint32_t*>(sKeep + 2 * MT);  // keep bits (behind a spare [MT] slot)
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x % nH, b = blockIdx.x / nH;
    const int D = nH * DH;
    const long ld = 3L * D;
    const bf16* base = qkv + (long)b * Tn * ld + h * DH;
    const bf16* dob = d_out + (long)b * Tn * D + h * DH;
    const int nunit = (Tn + 31) >> 5, nupair = (nunit + 1) >> 1;      // units of 32 keys = the 32-query blocks
    const int uA = wave, uB = nunit - 1 - wave;
    const bool work = wave < nupair;
    // K / V fragments of a unit's two key tiles (B operands: lane holds row key = 32 u + 16 j + l15, columns 32 ks + 8 g ..)
    bf16x8 kf[2][2], vf[2][2];
    auto load_frags = [&](int u) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int key = 32 * u + 16 * j + l15;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[j][ks] = vf[j][ks] = zero16<bf16>();
                if (key < Tn) {
                    kf[j][ks] = *reinterpret_cast<const bf16x8*>(base + (long)key * ld + D + ks * 32 + g * 8);
                    vf[j][ks] = *reinterpret_cast<const bf16x8*>(base + (long)key * ld + 2 * D + ks * 32 + g * 8);
                }
            }
        }
    };
    if (work) load_frags(uA);
    else {
#pragma unroll
        for (int j = 0; j < 2; ++j) kf[j][0] = kf[j][1] = vf[j][0] = vf[j][1] = zero16<bf16>();
    }
    const __amdgpu_buffer_rsrc_t rq = rows_rsrc(base, ld, Tn), ro = rows_rsrc(dob, D, Tn);
    dma_issue_scalars<MT>(sKeep, keep + (long)b * Tn, 1, Tn, wave, lane);
    dma_issue_scalars<MT>(sLse, lse + ((long)b * nH + h) * Tn, 1, Tn, wave, lane);
    dma_issue_scalars<MT>(sDel, delta + (long)b * Tn * nH + h, nH, Tn, wave, lane);
    dma_issue_chunk<NW, MT>(sQ, rq, ld, sO, ro, D, 0, nchunk, wave, lane);
    dma_issue_chunk<NW, MT>(sQ, rq, ld, sO, ro, D, 1, nchunk, wave, lane);
    if (trace) ts[1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (DROP) gen_keep_mask<64 * NW, MT>(sMask, (uint32_t)(b * nH + h), Tn, drop_seed, keep16, tid);
    // first use of the plain loads: a counted wait here instead of a full drain inside the loop
    asm volatile("" :: "v"(kf[0][0]), "v"(kf[0][1]), "v"(vf[0][0]), "v"(vf[0][1]), "v"(kf[1][0]), "v"(kf[1][1]), "v"(vf[1][0]), "v"(vf[1][1]));
    BwdKeys st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) st[j].dk[i] = st[j].dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ck = 0.f, cv = 0.f;                 // the wave's running bias-gradient sums (bias_fold16)
    const float ik_scale = inv_keep * 0.125f;
    // The wave's first unit (low keys: the longest sweep) takes the query blocks from its diagonal on as they land;
    // the second one runs on resident data.
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        dma_wait_chunk<NW, MT>(c);
        raw_barrier();
        if (c < 2) dma_issue_chunk<NW, MT>(sQ, rq, ld, sO, ro, D, c + 2, nchunk, wave, lane);
        if (c == 0) {
            for (int i = tid; i < MT; i += 64 * NW) { sLse[i] *= LOG2E; sDel[i] *= 0.125f; }
            raw_barrier();
            if (trace) ts[2] = __builtin_amdgcn_s_memrealtime();
        }
#pragma unroll 1
        for (int qb = c * (CROWS / 32); qb < (c + 1) * (CROWS / 32) && qb < nunit; ++qb)
            if (work && qb >= uA) {
                if (qb == uA) bwd_small_keys2_block<DROP, MT, true>(st, sQ, sO, sLse, sDel, sMask, kf, vf, uA, 32 * qb, lane, ik_scale);
                else bwd_small_keys2_block<DROP, MT, false>(st, sQ, sO, sLse, sDel, sMask, kf, vf, uA, 32 * qb, lane, ik_scale);
            }
    }
    if (trace) ts[3] = __builtin_amdgcn_s_memrealtime();
    if (work) {
        if (uB != uA) load_frags(uB);            // (in flight under the first unit's store)
        bwd_small_keys2_store(st, uA, lane, b, h, Tn, nH, inv_keep, sKeep, MT - 1, dqkv, dbias != nullptr, ck, cv);
        if (uB != uA) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) st[j].dk[i] = st[j].dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            bwd_small_keys2_block<DROP, MT, true>(st, sQ, sO, sLse, sDel, sMask, kf, vf, uB, 32 * uB, lane, ik_scale);
#pragma unroll 1
            for (int qb = uB + 1; qb < nunit; ++qb)
                bwd_small_keys2_block<DROP, MT, false>(st, sQ, sO, sLse, sDel, sMask, kf, vf, uB, 32 * qb, lane, ik_scale);
            bwd_small_keys2_store(st, uB, lane, b, h, Tn, nH, inv_keep, sKeep, MT - 1, dqkv, dbias != nullptr, ck, cv);
        }
    }
    if (trace) ts[4] = __builtin_amdgcn_s_memrealtime();
    if (dbias) {
        // every wave's row of partial sums -> sBw[wave][128] (K channels, then V; over the first bytes of the Q image once
        // every wave is done with it), then the rows added in wave order
        __syncthreads();
        float* sBw = reinterpret_cast<float*>(smem);
        const int ch = (l15 >> 2) * 16 + 4 * g + (l15 & 3);
        sBw[wave * 128 + ch] = ck;
        sBw[wave * 128 + DH + ch] = cv;
        __syncthreads();
        if (tid < 2 * DH) {
            float sm = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sm += sBw[w * 128 + tid];
            const int col = (1 + tid / DH) * D + h * DH + tid % DH;
            if (bias_rows) dbias[(long)b * 3 * D + col] = sm;
            else atomicAdd(dbias + col, sm);
        }
    }
    if (trace && lane == 0) {
        ts[5] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* r = trace + ((long)blockIdx.x * 16 + wave) * 8;
        r[0] = ts[0]; r[1] = ts[1]; r[2] = ts[2]; r[3] = ts[3]; r[4] = ts[4]; r[5] = ts[5];
        r[6] = __builtin_amdgcn_s_getreg((4 << 11) | 20);    // XCC_ID
        r[7] = 1;
    }
}

// ---- backward, dQ kernel: the 16 queries of tile t (Q, dO fragments in registers), one 32-key block
// DIAG: the block holds the tile's diagonal (the last block of a tile's sweep); every earlier block is wholly below it
template <bool DROP, int MT = 256, bool DIAG = true>
__device__ __forceinline__ void bwd_small_queries_block(f32x4 (&dq)[4], const char* sK, const char* sV, const float* sBias,
                                                        const uint32_t* sMask, const bf16x8 (&qf)[2], const bf16x8 (&of)[2],
                                                        float lse2_q, float dels_q, int t, int j0, int lane, float ik_scale) {
    const int g = lane >> 4, l15 = lane & 15;
    const int qi = 16 * t + l15, qlast = 16 * t + 15;
    const float c1 = 0.125f * LOG2E;
    f32x4 dsT[2];
    uint32_t wq = 0;
    if constexpr (DROP) wq = sMask[mask_row_base<MT>(qi) + (j0 >> 5)] >> (4 * g);      // this query row's keep bits of the block's 32 keys
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        dsT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!DIAG || j0 + 16 * kt <= qlast) {
            f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, dp_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                mma16(ld_row(sK, j0 + kt * 16 + l15, ks, g), qf[ks], s_acc);       // S^T[key][q]
                mma16(ld_row(sV, j0 + kt * 16 + l15, ks, g), of[ks], dp_acc);      // dP^T[key][q]
            }
            const int k0 = j0 + kt * 16 + 4 * g;
            const f32x4 kb = *reinterpret_cast<const f32x4*>(sBias + k0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = s_acc[r] * c1 - lse2_q + kb[r];
                if constexpr (DIAG) x = k0 + r <= qi ? x : -INFINITY;
                const float p = __builtin_amdgcn_exp2f(x);
                float dp = dp_acc[r];
                if constexpr (DROP) dp = mask_keep_u(dp, wq, 16 * kt + r);
                dsT[kt][r] = p * (dp * ik_scale - dels_q);
            }
        }
    }
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
    const bf16x8 db = acc_as_operand(dsT[0], dsT[1], bf16());
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        mma16(ld_ks(sK, j0 + 4 * g, j0 + 16 + 4 * g, dt * 16, lane, bf16()), db, dq[dt]);
}
__device__ __forceinline__ void bwd_small_queries_store(const f32x4 (&dq)[4], int t, int lane, int b, int h, int Tn, int nH,
                                                        bf16* __restrict__ dqkv, bool bias, float& cq) {
    const int g = lane >> 4, l15 = lane & 15, qi = 16 * t + l15, D = nH * DH;
    const long ld = 3L * D;
    const bool qok = qi < Tn;
    if (qok) {
        bf16* dst = dqkv + ((long)b * Tn + qi) * ld + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 qq = {(bf16)dq[dt][0], (bf16)dq[dt][1], (bf16)dq[dt][2], (bf16)dq[dt][3]};
            *reinterpret_cast<bf16x4*>(dst + dt * 16 + 4 * g) = qq;
        }
    }
    if (bias) {      // bias-gradient sums of the values as stored: folded over the tile's 16 queries into the wave's running sum
        f32x4 fq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) fq[dt][r] = qok ? (float)(bf16)dq[dt][r] : 0.f;
        cq += bias_fold16(fq, l15);
    }
}

template <bool DROP, int MT = 256>
__global__ __launch_bounds__(2 * MT, 4) void attn_bwd_small_q_kernel(const bf16* __restrict__ qkv, const int* __restrict__ keep,
        const bf16* __restrict__ d_out, const float* __restrict__ lse, const float* __restrict__ delta,
        bf16* __restrict__ dqkv, float* __restrict__ dbias, int bias_rows, int Tn, int nH,
        uint32_t keep16, uint32_t drop_seed, float inv_keep, unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    ts[0] = __builtin_amdgcn_s_memrealtime();        // (unconditional: testing `trace` first would put a kernel-argument round trip in front of everything)
    constexpr int NW = MT / 32, CROWS = MT / 4;
    const int nchunk = (Tn + CROWS - 1) / CROWS, rows_pad = nchunk * CROWS;
    char* sK = smem;
    char* sV = sK + rows_pad * 128;
    int* sKeep = reinterpret_cast<int*>(sV + rows_pad * 128);      // [MT] key-padding flags, turned IN PLACE into ...
    float* sBias = reinterpret_cast<float*>(sKeep);                // ... the additive 0 / -inf per key
    float* sLse = sBias + MT;                                       // [MT] LSE per query row (x log2 e on use)
    float* sDel = sLse + MT;                                        // [MT] delta per query row (/ sqrt(64) on use)
    uint32_t* sMask = reinterpret_cast<uint32_t*>(sDel + 2 * MT);   // keep bits (behind a spare [MT] slot)
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x % nH, b = blockIdx.x / nH;
    const int D = nH * DH;
    const long ld = 3L * D;
    const bf16* base = qkv + (long)b * Tn * ld + h * DH;
    const bf16* dob = d_out + (long)b * Tn * D + h * DH;
    const int ntile = (Tn + 15) >> 4, npair = (ntile + 1) >> 1;
    const int tA = wave, tB = ntile - 1 - wave;
    const bool work = wave < npair;
    // Requests, in this order: flags and the per-row scalars (LDS-DMA), chunk 0 of K / V, the Q / dO fragments of the wave's
    // LONG tile (plain loads, branch-free: a row beyond the tensor re-reads the last row and is zeroed below -- a conditional
    // load ends in a full `s_waitcnt vmcnt(0)` where its value joins the zero of the other path; with the row scalars as
    // conditional plain loads the prologue waited out two memory round trips before it requested anything: 5.3 us from
    // entry to "requests issued" in the timeline of round 4), chunk 1.  The short tile's fragments are requested after the
    // long sweep, under its store (16 registers that are not live during the sweep of a build capped at 128).
    const __amdgpu_buffer_rsrc_t rk = rows_rsrc(base + D, ld, Tn), rv = rows_rsrc(base + 2 * D, ld, Tn);
    dma_issue_scalars<MT>(sKeep, keep + (long)b * Tn, 1, Tn, wave, lane);
    dma_issue_scalars<MT>(sLse, lse + ((long)b * nH + h) * Tn, 1, Tn, wave, lane);
    dma_issue_scalars<MT>(sDel, delta + (long)b * Tn * nH + h, nH, Tn, wave, lane);
    dma_issue_chunk<NW, MT>(sK, rk, ld, sV, rv, ld, 0, nchunk, wave, lane);
    const int qa = 16 * tA + l15, qb_ = 16 * tB + l15;
    bf16x8 qfB[2], ofB[2];
    {
        const int qc = work ? (qb_ < Tn ? qb_ : Tn - 1) : 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qfB[ks] = *reinterpret_cast<const bf16x8*>(base + (long)qc * ld + ks * 32 + g * 8);
            ofB[ks] = *reinterpret_cast<const bf16x8*>(dob + (long)qc * D + ks * 32 + g * 8);
        }
    }
    dma_issue_chunk<NW, MT>(sK, rk, ld, sV, rv, ld, 1, nchunk, wave, lane);
    if (trace) ts[1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (DROP) gen_keep_mask<2 * MT, MT>(sMask, (uint32_t)(b * nH + h), Tn, drop_seed, keep16, tid);
    // first use of the plain loads: a counted wait here instead of a full drain inside the loop
    asm volatile("" :: "v"(qfB[0]), "v"(qfB[1]), "v"(ofB[0]), "v"(ofB[1]));
    if (!(work && qb_ < Tn)) { qfB[0] = qfB[1] = ofB[0] = ofB[1] = zero16<bf16>(); }
    float lseB = 0.f, delB = 0.f;
    f32x4 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float ik_scale = inv_keep * 0.125f;
    bf16x8 qfA[2], ofA[2];
    // tile B (high queries) consumes the key chunks as they land; tile A (a short range) runs afterwards
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        dma_wait_chunk<NW, MT>(c);
        raw_barrier();
        if (c < 2) dma_issue_chunk<NW, MT>(sK, rk, ld, sV, rv, ld, c + 2, nchunk, wave, lane);
        if (c == 0) {
            if (tid < MT) sBias[tid] = sKeep[tid] != 0 ? 0.f : -INFINITY;      // (in place: thread tid owns word tid)
            lseB = sLse[qb_ & (MT - 1)] * LOG2E;                                // (rows beyond the tensor: the DMA's bounds check left 0)
            delB = sDel[qb_ & (MT - 1)] * 0.125f;
            raw_barrier();
            if (trace) ts[2] = __builtin_amdgcn_s_memrealtime();
        }
#pragma unroll 1
        for (int kb = c * (CROWS / 32); kb < (c + 1) * (CROWS / 32); ++kb)
            if (work && 32 * kb <= 16 * tB + 15) {
                if (32 * kb + 31 <= 16 * tB) bwd_small_queries_block<DROP, MT, false>(dq, sK, sV, sBias, sMask, qfB, ofB, lseB, delB, tB, 32 * kb, lane, ik_scale);
                else bwd_small_queries_block<DROP, MT, true>(dq, sK, sV, sBias, sMask, qfB, ofB, lseB, delB, tB, 32 * kb, lane, ik_scale);
            }
    }
    if (trace) ts[3] = __builtin_amdgcn_s_memrealtime();
    float cq = 0.f;                 // the wave's running bias-gradient sums (bias_fold16)
    if (work) {
        if (tA != tB) {            // the short tile's fragments: requested here, in flight under the long tile's store
            const int qc = qa < Tn ? qa : Tn - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qfA[ks] = *reinterpret_cast<const bf16x8*>(base + (long)qc * ld + ks * 32 + g * 8);
                ofA[ks] = *reinterpret_cast<const bf16x8*>(dob + (long)qc * D + ks * 32 + g * 8);
            }
        }
        bwd_small_queries_store(dq, tB, lane, b, h, Tn, nH, dqkv, dbias != nullptr, cq);
        if (tA != tB) {
            if (!(qa < Tn)) { qfA[0] = qfA[1] = ofA[0] = ofA[1] = zero16<bf16>(); }
            const float lseA = sLse[qa & (MT - 1)] * LOG2E, delA = sDel[qa & (MT - 1)] * 0.125f;
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            int j0 = 0;
#pragma unroll 1
            for (; j0 + 31 <= 16 * tA; j0 += 32)
                bwd_small_queries_block<DROP, MT, false>(dq, sK, sV, sBias, sMask, qfA, ofA, lseA, delA, tA, j0, lane, ik_scale);
            bwd_small_queries_block<DROP, MT, true>(dq, sK, sV, sBias, sMask, qfA, ofA, lseA, delA, tA, j0, lane, ik_scale);
            bwd_small_queries_store(dq, tA, lane, b, h, Tn, nH, dqkv, dbias != nullptr, cq);
        }
    }
    if (trace) ts[4] = __builtin_amdgcn_s_memrealtime();
    if (dbias) {
        // every wave's row of partial sums -> sBw[wave][64] (over the first bytes of the K image once every wave is done with
        // it), then the rows added in wave order: the Q part of head h's columns
        __syncthreads();
        float* sBw = reinterpret_cast<float*>(smem);
        sBw[wave * DH + (l15 >> 2) * 16 + 4 * g + (l15 & 3)] = cq;
        __syncthreads();
        if (tid < DH) {
            float sm = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sm += sBw[w * DH + tid];
            const int col = h * DH + tid;
            if (bias_rows) dbias[(long)b * 3 * D + col] = sm;
            else atomicAdd(dbias + col, sm);
        }
    }
    if (trace && lane == 0) {
        ts[5] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* r = trace + ((long)(gridDim.x + blockIdx.x) * 16 + wave) * 8;      // (behind the dK / dV kernel's rows)
        r[0] = ts[0]; r[1] = ts[1]; r[2] = ts[2]; r[3] = ts[3]; r[4] = ts[4]; r[5] = ts[5];
        r[6] = __builtin_amdgcn_s_getreg((4 << 11) | 20);
        r[7] = 1;
    }
}

// two images + four [MT]-dword scalar slots + the keep-bit matrix
template <int MT = 256> inline size_t small_smem(int T) {
    const int cr = MT / 4, rp = ((T + cr - 1) / cr) * cr;
    return (size_t)rp * 256 + 4 * MT * 4 + (size_t)mask_words<MT>() * 4;
}
// keep probability of the whole-head kernels' dropout in units of 2^-12 and the matching scale
inline unsigned small_keep16(unsigned thresh32) { return 4096u - ((thresh32 + 0x80000u) >> 20); }
inline float small_inv_keep(unsigned keepq) { return (float)(4096.0 / (double)keepq); }


template <typename T> size_t bwd_smem_bytes() {
    typedef AT<T> A;
    const int KB = 4 * A::KPW;
    return (size_t)3 * KB * A::ROWB + 4 * 32 * A::ROWB + 32 * KB * sizeof(T) + 64 * sizeof(float) + KB * sizeof(int);
}

inline float inv_keep_of(unsigned thresh) {
    return thresh ? (float)(4294967296.0 / (4294967296.0 - (double)thresh)) : 1.0f;
}

}  // namespace

// diagnostic timeline of the whole-head kernels: per wave 8 x u64 (s_memrealtime stamps, XCC id, valid flag)
static unsigned long long* g_attn_trace = nullptr;
extern "C" int mmtg_attn_trace(void* buf) {
    g_attn_trace = reinterpret_cast<unsigned long long*>(buf);
    return MMTG_OK;
}

// A/B route switches, read ONCE per process and shared by the forward and the backward: the two passes must pick the same
// kernel family (whole-head and tiled kernels draw different dropout-mask families), so a switch flipped mid-process cannot
// pair a forward of one family with a backward of the other.
static bool attn_route_tiled() { static const bool v = getenv("MMTG_ATTN_TILED") != nullptr; return v; }     // tiled kernels for every T
static bool attn_route_no512() { static const bool v = getenv("MMTG_ATTN_NO512") != nullptr; return v; }     // tiled kernels for 256 < T <= 512 (round 2)

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
    const bool legacy = attn_route_tiled(), no512 = attn_route_no512();
    if (dtype == MMTG_BF16 && T <= (no512 ? 256 : SM_MAXT) && !legacy) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)attn_fwd_small_kernel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_smem<256>(256)) != hipSuccess ||
                hipFuncSetAttribute((const void*)attn_fwd_small_kernel<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_smem<256>(256)) != hipSuccess ||
                hipFuncSetAttribute((const void*)attn_fwd_small_kernel<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_smem<512>(512)) != hipSuccess ||
                hipFuncSetAttribute((const void*)attn_fwd_small_kernel<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_smem<512>(512)) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "attn_fwd: cannot raise dynamic LDS");
            attr_set = true;
        }
        const unsigned k16 = small_keep16(drop_thresh);
        const bool drop = k16 < 4096u;
        const unsigned kq = drop ? k16 : 0u;
        const float ikq = drop ? small_inv_keep(k16) : 1.0f;
        // one workgroup per (batch row, head): 8 waves and two workgroups per CU up to T = 256, 16 waves and one per CU up to 512
#define FWD_SMALL(DROP_, MT_) hipLaunchKernelGGL((attn_fwd_small_kernel<DROP_, MT_>), dim3(B * nH), dim3(2 * MT_), small_smem<MT_>(T), s, \
                                                 (const bf16*)qkv, keep, (bf16*)out, lse, T, nH, kq, drop_seed, ikq, g_attn_trace)
        if (T <= 256) { if (drop) FWD_SMALL(true, 256); else FWD_SMALL(false, 256); }
        else { if (drop) FWD_SMALL(true, 512); else FWD_SMALL(false, 512); }
#undef FWD_SMALL
        MMTG_LAUNCH_CHECK("attn_fwd");
        return MMTG_OK;
    }
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, 0, s, (const float*)qkv, keep, (float*)out, lse, T, nH, drop_thresh, drop_seed, ik);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, block, 0, s, (const bf16*)qkv, keep, (bf16*)out, lse, T, nH, drop_thresh, drop_seed, ik);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "attn_fwd: bad dtype");
    MMTG_LAUNCH_CHECK("attn_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, float* ws, long ws_floats, void* stream);

/* rows of dbias_ws ([rows, 3D] fp32 partial sums of the c_attn bias gradient) that mmtg_attn_bwd leaves UNSUMMED under MMTG_ATTN_DBIAS_ROWS
 * for this shape; 0: the shape runs on kernels that reduce inside the call whatever the flag says */
extern "C" int mmtg_attn_bwd_dbias_rows(int dtype, int B, int T) {
    return (dtype == MMTG_BF16 && T <= (attn_route_no512() ? 256 : SM_MAXT) && !attn_route_tiled()) ? B : 0;
}

extern "C" int mmtg_attn_bwd(int dtype, const void* qkv, const int* keep, const void* out, const void* dout,
                             const float* lse, float* delta, int delta_ready, float* dq32, void* dqkv, float* dbias, float* dbias_ws,
                             int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, int flags, void* stream) {
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
    bool small_path = false;
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
    } else if (dtype == MMTG_BF16 && T <= (attn_route_no512() ? 256 : SM_MAXT) && !attn_route_tiled()) {
        small_path = true;
        static bool attr_small = false;
        if (!attr_small) {
            const int shm2 = (int)small_smem<256>(256), shm5 = (int)small_smem<512>(512);
            bool ok = true;
#define SET_LDS(K_, B_) ok = ok && hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, B_) == hipSuccess
            SET_LDS((attn_bwd_small_kv_kernel<false, 8, 256>), shm2); SET_LDS((attn_bwd_small_kv_kernel<true, 8, 256>), shm2);
            SET_LDS((attn_bwd_small_kv_kernel<false, 4, 256>), shm2); SET_LDS((attn_bwd_small_kv_kernel<true, 4, 256>), shm2);
            SET_LDS((attn_bwd_small_q_kernel<false, 256>), shm2); SET_LDS((attn_bwd_small_q_kernel<true, 256>), shm2);
            SET_LDS((attn_bwd_small_kv_kernel<false, 16, 512>), shm5); SET_LDS((attn_bwd_small_kv_kernel<true, 16, 512>), shm5);
            SET_LDS((attn_bwd_small_kv_kernel<false, 8, 512>), shm5); SET_LDS((attn_bwd_small_kv_kernel<true, 8, 512>), shm5);
            SET_LDS((attn_bwd_small_q_kernel<false, 512>), shm5); SET_LDS((attn_bwd_small_q_kernel<true, 512>), shm5);
            SET_LDS((attn_bwd_small_kv2_kernel<false, 256>), shm2); SET_LDS((attn_bwd_small_kv2_kernel<true, 256>), shm2);
            SET_LDS((attn_bwd_small_kv2_kernel<false, 512>), shm5); SET_LDS((attn_bwd_small_kv2_kernel<true, 512>), shm5);
#undef SET_LDS
            if (!ok) MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: cannot raise dynamic LDS");
            attr_small = true;
        }
        if (!delta_ready) hipLaunchKernelGGL(attn_delta_kernel<bf16>, dim3(cdiv(rows * nH, 4)), dim3(256), 0, s, (const bf16*)out, (const bf16*)dout, delta, T, nH, rows);
        const unsigned th16 = small_keep16(drop_thresh);
        // MMTG_ATTN_ELEM_MASK: the per-element mask of the tiled / split-precision kernels (the 32-bit threshold travels in the
        // keep argument, the exact 1 / keep scale beside it) instead of the whole-head kernels' own 12-bit word masks
        const bool elem = (flags & MMTG_ATTN_ELEM_MASK) && drop_thresh > 4096u;
        const bool drop = elem || th16 < 4096u;
        const unsigned kq = elem ? drop_thresh : drop ? th16 : 0u;
        const float ik16 = elem ? ik : drop ? small_inv_keep(th16) : 1.0f;
        // dK/dV kernel: with dropout the half-wave build (every wave two tile pairs, up to 256 VGPRs, no spills) wins at T <= 256
        // (117 vs 135 us for backward + delta in isolation); without, the full-wave build does (106 vs 111).
        // MMTG_ATTN_KV_NW=4|8 (T <= 256) / 8|16 (T <= 512) forces one.
        static const int kv_nw = getenv("MMTG_ATTN_KV_NW") ? atoi(getenv("MMTG_ATTN_KV_NW")) : 0;
#define KV_SMALL(DROP_, NW_, MT_) hipLaunchKernelGGL((attn_bwd_small_kv_kernel<DROP_, NW_, MT_>), dim3(B * nH), dim3(64 * NW_), small_smem<MT_>(T), s, \
                                   (const bf16*)qkv, keep, (const bf16*)dout, lse, delta, (bf16*)dqkv, bias_dst, bias_rows, T, nH, kq, drop_seed, ik16, g_attn_trace)
#define Q_SMALL(DROP_, MT_) hipLaunchKernelGGL((attn_bwd_small_q_kernel<DROP_, MT_>), dim3(B * nH), dim3(2 * MT_), small_smem<MT_>(T), sq, \
                                   (const bf16*)qkv, keep, (const bf16*)dout, lse, delta, (bf16*)dqkv, bias_dst, bias_rows, T, nH, kq, drop_seed, ik16, g_attn_trace)
        static const bool kv2 = !(getenv("MMTG_ATTN_KV2") && atoi(getenv("MMTG_ATTN_KV2")) == 0);     // 32 keys per wave (round 4); 0 = the 16-key builds
#define KV2_SMALL(DROP_, MT_) hipLaunchKernelGGL((attn_bwd_small_kv2_kernel<DROP_, MT_>), dim3(B * nH), dim3(MT_), small_smem<MT_>(T), s, \
                                   (const bf16*)qkv, keep, (const bf16*)dout, lse, delta, (bf16*)dqkv, bias_dst, bias_rows, T, nH, kq, drop_seed, ik16, g_attn_trace)
        // MMTG_ATTN_BWD_FORK=1 (A/B, round 4): the dQ kernel on a second stream beside the dK / dV kernel (they share inputs only;
        // the 768 + 768 workgroups then fill the half-empty second round of either kernel with the other's)
        static const bool fork = getenv("MMTG_ATTN_BWD_FORK") && atoi(getenv("MMTG_ATTN_BWD_FORK")) == 1;
        static hipStream_t s2 = nullptr;
        static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        if (fork && !s2) {
            if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: cannot create the side stream");
        }
        hipStream_t sq = s;
        if (fork && kv2 && !kv_nw) {
            sq = s2;
            if (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent(s2, ev_fork, 0) != hipSuccess) MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: fork failed");
        }
        if (kv2 && !kv_nw) {
            if (T <= 256) { if (drop) KV2_SMALL(true, 256); else KV2_SMALL(false, 256); }
            else { if (drop) KV2_SMALL(true, 512); else KV2_SMALL(false, 512); }
            if (T <= 256) { if (drop) Q_SMALL(true, 256); else Q_SMALL(false, 256); }
            else { if (drop) Q_SMALL(true, 512); else Q_SMALL(false, 512); }
            if (sq != s) {
                if (hipEventRecord(ev_join, s2) != hipSuccess || hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd: join failed");
            }
        } else if (T <= 256) {
            const bool half = kv_nw ? kv_nw == 4 : drop;
            if (drop) { if (half) KV_SMALL(true, 4, 256); else KV_SMALL(true, 8, 256); Q_SMALL(true, 256); }
            else { if (half) KV_SMALL(false, 4, 256); else KV_SMALL(false, 8, 256); Q_SMALL(false, 256); }
        } else {
            const bool half = kv_nw ? kv_nw == 8 : drop;
            if (drop) { if (half) KV_SMALL(true, 8, 512); else KV_SMALL(true, 16, 512); Q_SMALL(true, 512); }
            else { if (half) KV_SMALL(false, 8, 512); else KV_SMALL(false, 16, 512); Q_SMALL(false, 512); }
        }
#undef KV_SMALL
#undef KV2_SMALL
#undef Q_SMALL
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
    // (the whole-head kernels write ONE partial bias row per batch row and leave nothing to the dQ column pass)
    const int nkb_ = small_path ? 1 : cdiv(T, dtype == MMTG_F32 ? 4 * AT<float>::KPW : 4 * AT<bf16>::KPW);
    // MMTG_ATTN_DBIAS_ROWS (whole-head kernels only, mmtg_attn_bwd_dbias_rows() > 0): the B partial rows stay in dbias_ws for the caller's
    // batched sum (mmtg_colsum_batch) instead of a launch of their own here
    if (bias_rows && !((flags & MMTG_ATTN_DBIAS_ROWS) && small_path)) {
        int rc = mmtg_colsum(MMTG_F32, dbias_ws, 3L * D, B * nkb_, 3 * D, dbias, nullptr, 0, stream);      // (few rows: one ordered pass)
        if (rc) return rc;
    }
    // (tall: its slices go into the 44 spare rows behind the partial bias rows of dbias_ws -- mmtg_attn_bwd's contract)
    if (dbias && nkb_ > 1)
        return mmtg_colsum(dtype, dqkv, 3L * D, (int)rows, D, dbias, dbias_ws ? dbias_ws + (long)B * nkb_ * 3 * D : nullptr, dbias_ws ? 44L * 3 * D : 0, stream);
    return MMTG_OK;
}
