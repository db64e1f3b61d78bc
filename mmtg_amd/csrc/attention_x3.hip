// Causal multi-head self-attention of the split-precision ("bf16x3") mode, round 5.
//
// The fp32-storage modes ran attention on v_mfma_f32_16x16x4_f32 (attention.hip, attn_fwd_kernel<float> / attn_bwd_kernel<float, 4>):
// 2.4 + 8.1 ms of a 44 ms bf16x3 train step, a sixteenth of the bf16 matrix rate.  Here the SAME tiled algorithm runs on the bf16
// matrix cores with every product as three passes over (hi | lo) splits, like the mode's GEMMs (gemm.hip, gemm_p8_kernel<X3>):
//   * Q, K, V, dO arrive as (hi | lo) bf16 plane pairs -- what the c_attn product and the c_proj dgrad write for every other
//     split-precision consumer too (their fp32 outputs are gone: same bytes, lane-contiguous stores) -- and go global ->
//     registers -> hi and lo images in LDS (the bf16 kernels' image layouts, attn_common.h) without arithmetic;
//   * the probabilities P and the score gradients dS are split in registers where they become MFMA operands;
//   * S = Qh Kh + Ql Kh + Qh Kl, O = Ph Vh + Pl Vh + Ph Vl, and likewise dP, dV, dK, dQ; softmax, masks, dropout in fp32
//     as the fp32 kernels do them (same counter-hash dropout stream: forward and backward of either family pair up), with the
//     hardware exponential (v_exp_f32, ~1e-6 relative: below the products' own 4e-6) instead of libm's expf.
// Outputs fp32 (+ the plane pairs the next split-precision products read).  Reference arithmetic: the fp32 attention of
// transformers' GPT-2 behind /root/reference/src/model.py:282-288.
#include <stdlib.h>

#include "attn_common.h"

namespace {

typedef AT<bf16> A2;

// three-pass product: acc += (ah + al) (bh + bl) without the lo x lo term
__device__ __forceinline__ void mma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4& acc) {
    mma16(ah, bh, acc);
    mma16(al, bh, acc);
    mma16(ah, bl, acc);
}
// accumulator tiles (fp32) -> (hi | lo) B operands of the next product (k = the tiles' row index, mma.h acc_as_operand)
__device__ __forceinline__ void acc_split(const f32x4& lo_t, const f32x4& hi_t, bf16x8& oh, bf16x8& ol) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        oh[e] = (bf16)lo_t[e]; ol[e] = (bf16)(lo_t[e] - (float)oh[e]);
        oh[4 + e] = (bf16)hi_t[e]; ol[4 + e] = (bf16)(hi_t[e] - (float)oh[4 + e]);
    }
}

// ======================================================================== forward
// one workgroup = (batch, head, 64 queries), 4 waves x 16 queries; K / V tiles of 64 keys as hi / lo images (32 KB of LDS)
__global__ __launch_bounds__(256, 2) void attn_fwd_x3_kernel(const bf16* __restrict__ qkv, long qplane, const int* __restrict__ keep,
        float* __restrict__ out, bf16* __restrict__ oplanes, long oplane, float* __restrict__ lse, int Tn, int nH,
        uint32_t drop_thresh, uint32_t drop_seed, float inv_keep) {
    __shared__ __attribute__((aligned(16))) char sKh[64 * 128];
    __shared__ __attribute__((aligned(16))) char sKl[64 * 128];
    __shared__ __attribute__((aligned(16))) char sVh[64 * 128];
    __shared__ __attribute__((aligned(16))) char sVl[64 * 128];
    __shared__ int sKeep[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int qb = gridDim.x - 1 - blockIdx.x, h = blockIdx.y, b = blockIdx.z;   // longest (most key blocks) first
    const int D = nH * DH;
    const long ld = 3L * D;
    const bf16* base = qkv + (long)b * Tn * ld + h * DH;
    const int qi = qb * 64 + wave * 16 + l15;

    // (the 1 / sqrt(64) of the scores is applied to the fp32 accumulators: a power of two, exact either way)
    bf16x8 qh[2], ql[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qh[ks] = ql[ks] = zero16<bf16>();
        if (qi < Tn) {
            const bf16* src = base + (long)qi * ld + ks * 32 + g * 8;
            qh[ks] = *reinterpret_cast<const bf16x8*>(src);
            ql[ks] = *reinterpret_cast<const bf16x8*>(src + qplane);
        }
    }

    f32x4 o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t drow = ((uint32_t)(b * nH + h) * (uint32_t)Tn + (uint32_t)qi) * (uint32_t)Tn;

    // a tile = 64 keys x 8 chunks of 8 elements: two chunks per thread; the next tile's rows are requested right after this
    // one's images are written, so their latency hides behind this tile's MFMAs and softmax
    bf16x8 kreg[2][2], vreg[2][2];      // [it][hi | lo]
#define X3_FETCH(J0)                                                                                          \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                        \
        const int id = tid + 256 * it, key = id >> 3, c = id & 7;                                             \
        kreg[it][0] = kreg[it][1] = vreg[it][0] = vreg[it][1] = zero16<bf16>();                                      \
        if ((J0) + key < Tn) {                                                                                \
            const bf16* src = base + (long)((J0) + key) * ld + c * 8;                                         \
            kreg[it][0] = *reinterpret_cast<const bf16x8*>(src + D);                                          \
            kreg[it][1] = *reinterpret_cast<const bf16x8*>(src + D + qplane);                                 \
            vreg[it][0] = *reinterpret_cast<const bf16x8*>(src + 2 * D);                                      \
            vreg[it][1] = *reinterpret_cast<const bf16x8*>(src + 2 * D + qplane);                             \
        }                                                                                                     \
    }
    X3_FETCH(0)
    for (int jb = 0; jb <= qb; ++jb) {
        const int j0 = jb * 64;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int id = tid + 256 * it, key = id >> 3, c = id & 7;
            *reinterpret_cast<bf16x8*>(sKh + off_kc<bf16>(key, c)) = kreg[it][0];
            *reinterpret_cast<bf16x8*>(sKl + off_kc<bf16>(key, c)) = kreg[it][1];
            *reinterpret_cast<bf16x8*>(sVh + off_ks<bf16>(key, c)) = vreg[it][0];
            *reinterpret_cast<bf16x8*>(sVl + off_ks<bf16>(key, c)) = vreg[it][1];
        }
        if (jb < qb) { X3_FETCH(j0 + 64) }
        if (tid < 64) sKeep[tid] = (j0 + tid < Tn) ? keep[(long)b * Tn + j0 + tid] : 0;
        __syncthreads();

        // S^T[key][q] for this wave's 16 queries x 64 keys
        f32x4 s_acc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s_acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                mma3(ld_kc<bf16>(sKh, kt * 16 + l15, ks, g), ld_kc<bf16>(sKl, kt * 16 + l15, ks, g), qh[ks], ql[ks], s_acc[kt]);
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = kt * 16 + 4 * g + r, kj = j0 + kl;
                const bool valid = kj <= qi && sKeep[kl] != 0;
                const float s = valid ? s_acc[kt][r] * 0.125f : -INFINITY;
                s_acc[kt][r] = s;
                mloc = fmaxf(mloc, s);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = s_acc[kt][r];
                float p = (s == -INFINITY) ? 0.f : __expf(s - m_use);
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
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl;
            acc_split(s_acc[2 * s2], s_acc[2 * s2 + 1], ph, pl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                mma3(ld_ks(sVh, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, dt * 16, lane, bf16()),
                     ld_ks(sVl, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, dt * 16, lane, bf16()), ph, pl, o_acc[dt]);
        }
    }

#undef X3_FETCH
    if (qi < Tn) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        float* dst = out + ((long)b * Tn + qi) * D + h * DH;
        bf16* pd = oplanes ? oplanes + ((long)b * Tn + qi) * D + h * DH : nullptr;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 o = {o_acc[dt][0] * inv, o_acc[dt][1] * inv, o_acc[dt][2] * inv, o_acc[dt][3] * inv};
            *reinterpret_cast<f32x4*>(dst + dt * 16 + 4 * g) = o;
            if (pd) {
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hi[e] = (bf16)o[e]; lo[e] = (bf16)(o[e] - (float)hi[e]); }
                *reinterpret_cast<bf16x4*>(pd + dt * 16 + 4 * g) = hi;
                *reinterpret_cast<bf16x4*>(pd + oplane + dt * 16 + 4 * g) = lo;
            }
        }
        if (g == 0) lse[((long)b * nH + h) * Tn + qi] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
    }
}

// ======================================================================== backward
// one workgroup = (batch, head, block of KB = 128 keys), 8 waves x 16 keys; dK^T and dV^T live in accumulators across the query
// sweep (32-query tiles); S and dP are computed with the key on the lane so P and dS feed dV^T / dK^T as B operands from registers;
// dS crosses LDS once (hi and lo images) for dQ = dS K.  Every image exists twice (hi | lo): 144 KB of LDS, one workgroup per CU.
constexpr int XKB = 128, XNW = 8, XKPW = XKB / XNW;          // 16 keys per wave: one key tile
constexpr int XRBS = XKB * 2;                                // dS image row bytes

struct BwdLds {      // byte offsets inside the dynamic LDS block
    static constexpr int KR = 0, KT = KR + 2 * XKB * 128, VR = KT + 2 * XKB * 128, QR = VR + 2 * XKB * 128,
                         QT = QR + 2 * 32 * 128, OR = QT + 2 * 32 * 128, OT = OR + 2 * 32 * 128, DS = OT + 2 * 32 * 128,
                         LSE = DS + 2 * 32 * XRBS, DEL = LSE + 32 * 4, KEEP = DEL + 32 * 4, END = KEEP + XKB * 4;
};
constexpr int XIMG_K = XKB * 128, XIMG_Q = 32 * 128, XIMG_DS = 32 * XRBS;      // bytes from a hi image to its lo image

__global__ __launch_bounds__(64 * XNW, 1) void attn_bwd_x3_kernel(const bf16* __restrict__ qkv, long qplane, const int* __restrict__ keep,
        const bf16* __restrict__ d_out, long doplane, const float* __restrict__ lse, const float* __restrict__ delta,
        float* __restrict__ dq32, long dq_stride, bf16* __restrict__ dqkv_p, long dplane, float* __restrict__ dbias, int Tn, int nH,
        uint32_t drop_thresh, uint32_t drop_seed, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sKr = smem + BwdLds::KR;
    char* const sKt = smem + BwdLds::KT;
    char* const sVr = smem + BwdLds::VR;
    char* const sQr = smem + BwdLds::QR;
    char* const sQt = smem + BwdLds::QT;
    char* const sOr = smem + BwdLds::OR;
    char* const sOt = smem + BwdLds::OT;
    char* const sDS = smem + BwdLds::DS;
    float* const sLse = reinterpret_cast<float*>(smem + BwdLds::LSE);
    float* const sDel = reinterpret_cast<float*>(smem + BwdLds::DEL);
    int* const sKeep = reinterpret_cast<int*>(smem + BwdLds::KEEP);
    constexpr int NTHR = 64 * XNW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int kb0 = blockIdx.x * XKB, h = blockIdx.y, b = blockIdx.z;
    const int D = nH * DH;
    const long ld = 3L * D;
    const bf16* base = qkv + (long)b * Tn * ld + h * DH;
    const bf16* dob = d_out + (long)b * Tn * D + h * DH;
    const float scale = 0.125f;

    // stage this block's K (row and transposed-read images) and V (row image) once
    for (int id = tid; id < XKB * 8; id += NTHR) {
        const int key = id >> 3, c = id & 7;
        bf16x8 kh = zero16<bf16>(), kl = kh, vh = kh, vl = kh;
        if (kb0 + key < Tn) {
            const bf16* src = base + (long)(kb0 + key) * ld + c * 8;
            kh = *reinterpret_cast<const bf16x8*>(src + D); kl = *reinterpret_cast<const bf16x8*>(src + D + qplane);
            vh = *reinterpret_cast<const bf16x8*>(src + 2 * D); vl = *reinterpret_cast<const bf16x8*>(src + 2 * D + qplane);
        }
        *reinterpret_cast<bf16x8*>(sKr + off_kc<bf16>(key, c)) = kh;
        *reinterpret_cast<bf16x8*>(sKr + XIMG_K + off_kc<bf16>(key, c)) = kl;
        *reinterpret_cast<bf16x8*>(sKt + off_ks<bf16>(key, c)) = kh;
        *reinterpret_cast<bf16x8*>(sKt + XIMG_K + off_ks<bf16>(key, c)) = kl;
        *reinterpret_cast<bf16x8*>(sVr + off_kc<bf16>(key, c)) = vh;
        *reinterpret_cast<bf16x8*>(sVr + XIMG_K + off_kc<bf16>(key, c)) = vl;
    }
    for (int i = tid; i < XKB; i += NTHR) sKeep[i] = (kb0 + i < Tn) ? keep[(long)b * Tn + kb0 + i] : 0;

    f32x4 dk_acc[4], dv_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dk_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int kw0 = XKPW * wave;  // this wave's first key (block-local)
    const uint32_t dbase = (uint32_t)(b * nH + h) * (uint32_t)Tn;
    const int nqt = (Tn + 31) / 32;
    const int kl = kw0 + l15, key = kb0 + kl;
    // this wave's K / V fragments (its 16 keys, both k-blocks, hi | lo) do not change over the query sweep: registers, not 16 LDS
    // reads per query tile
    __syncthreads();
    bf16x8 kfh[2], kfl[2], vfh[2], vfl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kfh[ks] = ld_kc<bf16>(sKr, kl, ks, g); kfl[ks] = ld_kc<bf16>(sKr + XIMG_K, kl, ks, g);
        vfh[ks] = ld_kc<bf16>(sVr, kl, ks, g); vfl[ks] = ld_kc<bf16>(sVr + XIMG_K, kl, ks, g);
    }
    // the Q / dO rows of a query tile travel global -> registers -> hi / lo images; the NEXT tile's rows are requested right after this
    // tile's images are written, so their HBM latency (one workgroup per CU: nothing else would hide it) overlaps this tile's arithmetic
    bf16x8 qhv = zero16<bf16>(), qlv = qhv, ohv = qhv, olv = qhv;
    float lse_n = 0.f, del_n = 0.f;
#define X3_FETCH_Q(Q0)                                                                                        \
    do {                                                                                                      \
        if (tid < 32 * 8) {                                                                                   \
            const int r = tid >> 3, c = tid & 7;                                                              \
            qhv = qlv = ohv = olv = zero16<bf16>();                                                                  \
            if ((Q0) + r < Tn) {                                                                              \
                const bf16* qs_ = base + (long)((Q0) + r) * ld + c * 8;                                       \
                const bf16* os_ = dob + (long)((Q0) + r) * D + c * 8;                                         \
                qhv = *reinterpret_cast<const bf16x8*>(qs_); qlv = *reinterpret_cast<const bf16x8*>(qs_ + qplane);   \
                ohv = *reinterpret_cast<const bf16x8*>(os_); olv = *reinterpret_cast<const bf16x8*>(os_ + doplane);  \
            }                                                                                                 \
        } else if (tid < 32 * 8 + 32) {                                                                       \
            const int r = tid - 256;                                                                          \
            const bool ok = (Q0) + r < Tn;                                                                    \
            lse_n = ok ? lse[((long)b * nH + h) * Tn + (Q0) + r] : 0.f;                                       \
            del_n = ok ? delta[((long)b * Tn + (Q0) + r) * nH + h] : 0.f;                                     \
        }                                                                                                     \
    } while (0)
    X3_FETCH_Q((kb0 / 32) * 32);
    for (int qt = kb0 / 32; qt < nqt; ++qt) {
        const int q0 = qt * 32;
        __syncthreads();
        if (tid < 32 * 8) {
            const int r = tid >> 3, c = tid & 7;
            *reinterpret_cast<bf16x8*>(sQr + off_kc<bf16>(r, c)) = qhv;
            *reinterpret_cast<bf16x8*>(sQr + XIMG_Q + off_kc<bf16>(r, c)) = qlv;
            *reinterpret_cast<bf16x8*>(sQt + off_ks<bf16>(r, c)) = qhv;
            *reinterpret_cast<bf16x8*>(sQt + XIMG_Q + off_ks<bf16>(r, c)) = qlv;
            *reinterpret_cast<bf16x8*>(sOr + off_kc<bf16>(r, c)) = ohv;
            *reinterpret_cast<bf16x8*>(sOr + XIMG_Q + off_kc<bf16>(r, c)) = olv;
            *reinterpret_cast<bf16x8*>(sOt + off_ks<bf16>(r, c)) = ohv;
            *reinterpret_cast<bf16x8*>(sOt + XIMG_Q + off_ks<bf16>(r, c)) = olv;
        } else if (tid < 32 * 8 + 32) {
            sLse[tid - 256] = lse_n;
            sDel[tid - 256] = del_n;
        }
        if (qt + 1 < nqt) X3_FETCH_Q(q0 + 32);
        __syncthreads();

        const bool active = (kb0 + kw0 <= q0 + 31) && (kb0 + kw0 < Tn);
        if (active) {
            f32x4 pT[2], dsT[2];
            const bool kpok = sKeep[kl] != 0;
            const int byte = kl * 2, bch = byte >> 4, blo = byte & 15;
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, dp_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    mma3(ld_kc<bf16>(sQr, qs * 16 + l15, ks, g), ld_kc<bf16>(sQr + XIMG_Q, qs * 16 + l15, ks, g), kfh[ks], kfl[ks], s_acc);
                    mma3(ld_kc<bf16>(sOr, qs * 16 + l15, ks, g), ld_kc<bf16>(sOr + XIMG_Q, qs * 16 + l15, ks, g), vfh[ks], vfl[ks], dp_acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql_ = qs * 16 + 4 * g + r, q = q0 + ql_;
                    const bool valid = key <= q && q < Tn && kpok;
                    const float p = valid ? __expf(s_acc[r] * scale - sLse[ql_]) : 0.f;
                    float dp = dp_acc[r];
                    if (drop_thresh) {
                        const float ms = dropout_scale(drop_seed, (dbase + (uint32_t)q) * (uint32_t)Tn + (uint32_t)key, drop_thresh, inv_keep);
                        dp *= ms;
                        pT[qs][r] = p * ms;
                    } else {
                        pT[qs][r] = p;
                    }
                    const float ds = p * (dp - sDel[ql_]) * scale;
                    dsT[qs][r] = ds;
                    // dS images [q][key] (hi | lo) for the dQ product (chunk swizzle by the row's low 3 bits)
                    const bf16 dh = (bf16)ds;
                    char* dst = sDS + ql_ * XRBS + ((bch ^ ((4 * g + r) & 7)) << 4) + blo;
                    *reinterpret_cast<bf16*>(dst) = dh;
                    *reinterpret_cast<bf16*>(dst + XIMG_DS) = (bf16)(ds - (float)dh);
                }
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
            bf16x8 pbh, pbl, dbh, dbl;
            acc_split(pT[0], pT[1], pbh, pbl);
            acc_split(dsT[0], dsT[1], dbh, dbl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                mma3(ld_ks(sOt, 4 * g, 16 + 4 * g, dt * 16, lane, bf16()), ld_ks(sOt + XIMG_Q, 4 * g, 16 + 4 * g, dt * 16, lane, bf16()), pbh, pbl, dv_acc[dt]);
                mma3(ld_ks(sQt, 4 * g, 16 + 4 * g, dt * 16, lane, bf16()), ld_ks(sQt + XIMG_Q, 4 * g, 16 + 4 * g, dt * 16, lane, bf16()), dbh, dbl, dk_acc[dt]);
            }
        } else {
            // (an inactive wave still owns its 16 columns of the dS images: the dQ product reads every column below nact)
            const int byte = kl * 2, bch = byte >> 4, blo = byte & 15;
#pragma unroll
            for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql_ = qs * 16 + 4 * g + r;
                    char* dst = sDS + ql_ * XRBS + ((bch ^ ((4 * g + r) & 7)) << 4) + blo;
                    *reinterpret_cast<bf16*>(dst) = (bf16)0.f;
                    *reinterpret_cast<bf16*>(dst + XIMG_DS) = (bf16)0.f;
                }
        }
        __syncthreads();

        // dQ[q][d] = sum_key dS[q][key] K[key][d]; wave w: d-tile w & 3, query sub-tile w >> 2
        {
            const int nact = min(XKB, min(q0 + 32, Tn) - kb0);   // keys that can be <= some q of this tile
            const int nblk = (nact + 31) / 32;                    // k-blocks of 32 keys
            const int dt = wave & 3, qs = wave >> 2;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int row = qs * 16 + l15;
            for (int kb = 0; kb < nblk; ++kb) {
                const int o = row * XRBS + (((kb * 4 + g) ^ (row & 7)) << 4);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sDS + o), al = *reinterpret_cast<const bf16x8*>(sDS + XIMG_DS + o);
                mma3(ah, al, ld_ks(sKt, kb * 32 + 8 * g, kb * 32 + 8 * g + 4, dt * 16, lane, bf16()),
                     ld_ks(sKt + XIMG_K, kb * 32 + 8 * g, kb * 32 + 8 * g + 4, dt * 16, lane, bf16()), acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + qs * 16 + 4 * g + r;
                // (this key block's OWN dQ buffer, plain stores: every (query tile, key block) pair belongs to exactly one workgroup;
                //  the finish kernel adds the blocks in order -- no atomics, no zero fill, the same bits every run)
                if (q < Tn) dq32[(long)blockIdx.x * dq_stride + ((long)b * Tn + q) * D + h * DH + dt * 16 + l15] = acc[r];
            }
        }
    }

#undef X3_FETCH_Q
    // dK, dV of this wave's keys: straight into the (hi | lo) plane pair of d(qkv) -- only split-precision products read it
    if (key < Tn) {
        bf16* dst = dqkv_p + ((long)b * Tn + key) * ld + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 kh, klo, vh, vlo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kh[e] = (bf16)dk_acc[dt][e]; klo[e] = (bf16)(dk_acc[dt][e] - (float)kh[e]);
                vh[e] = (bf16)dv_acc[dt][e]; vlo[e] = (bf16)(dv_acc[dt][e] - (float)vh[e]);
            }
            *reinterpret_cast<bf16x4*>(dst + D + dt * 16 + 4 * g) = kh;
            *reinterpret_cast<bf16x4*>(dst + dplane + D + dt * 16 + 4 * g) = klo;
            *reinterpret_cast<bf16x4*>(dst + 2 * D + dt * 16 + 4 * g) = vh;
            *reinterpret_cast<bf16x4*>(dst + dplane + 2 * D + dt * 16 + 4 * g) = vlo;
        }
    }

    // c_attn bias gradient, k and v parts: column sums over this block's keys, one partial row per workgroup (the host sums the
    // rows in a fixed order); the q part comes from the dQ finish kernel
    if (dbias) {
        float* sB = reinterpret_cast<float*>(smem);      // [waves][2][64], overlays the K image: a slot per wave, summed in wave order
        __syncthreads();                                  // every wave is done with the staged tiles
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float sk[4], sv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { sk[r] = key < Tn ? dk_acc[dt][r] : 0.f; sv[r] = key < Tn ? dv_acc[dt][r] : 0.f; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { sk[r] += __shfl_xor(sk[r], o, 64); sv[r] += __shfl_xor(sv[r], o, 64); }
                if (l15 == 0) {
                    sB[wave * 2 * DH + dt * 16 + 4 * g + r] = sk[r];
                    sB[wave * 2 * DH + DH + dt * 16 + 4 * g + r] = sv[r];
                }
            }
        }
        __syncthreads();
        if (tid < 2 * DH) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < XNW; ++w) t += sB[w * 2 * DH + tid];
            const int col = (1 + tid / DH) * D + h * DH + tid % DH;
            dbias[((long)b * gridDim.x + blockIdx.x) * 3 * D + col] = t;
        }
        if (tid < DH) dbias[((long)b * gridDim.x + blockIdx.x) * 3 * D + h * DH + tid] = 0.f;      // (q part: the finish kernel's)
    }
}

// dq32 [key blocks][rows, D] (fp32: block kb holds the rows of the queries q >= 128 kb) -> their sum in block order = the q columns of
// d(qkv)'s plane pair + their column sums
// (the q part of the c_attn bias gradient): one workgroup per 16-row band (a wave per 4 rows x 256 columns at a time), partial
// sums to qsum[band][D] (the host sums the bands in a fixed order)
constexpr int XFB = 16;      // rows per band
__global__ __launch_bounds__(256) void attn_dq_finish_x3_kernel(const float* __restrict__ dq32, long dq_stride, int Tn, bf16* __restrict__ dqkv_p, long dplane,
                                                               float* __restrict__ qsum, long rows, int D) {
    __shared__ float sred[4][1024];
    const long r0 = (long)blockIdx.x * XFB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c0 = 0; c0 < D; c0 += 256) {
        const int c = c0 + lane * 4;
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
        if (c < D) {
#pragma unroll
            for (int i = 0; i < XFB / 4; ++i) {
                const long r = r0 + wave * (XFB / 4) + i;
                if (r < rows) {
                    const int nb = (int)(r % Tn) / XKB + 1;                  // key blocks that hold keys <= this query
                    f32x4 v = *reinterpret_cast<const f32x4*>(dq32 + r * D + c);
                    for (int kb = 1; kb < nb; ++kb) v += *reinterpret_cast<const f32x4*>(dq32 + kb * dq_stride + r * D + c);
                    bf16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); cs[e] += v[e]; }
                    bf16* dst = dqkv_p + r * 3 * D + c;
                    *reinterpret_cast<bf16x4*>(dst) = hi;
                    *reinterpret_cast<bf16x4*>(dst + dplane) = lo;
                }
            }
        }
        if (qsum) {
            __syncthreads();
            if (c < D) *reinterpret_cast<f32x4*>(&sred[wave][lane * 4]) = cs;
            __syncthreads();
            if (wave == 0 && c < D) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&sred[0][lane * 4]), b = *reinterpret_cast<const f32x4*>(&sred[1][lane * 4]);
                const f32x4 d = *reinterpret_cast<const f32x4*>(&sred[2][lane * 4]), e = *reinterpret_cast<const f32x4*>(&sred[3][lane * 4]);
                *reinterpret_cast<f32x4*>(qsum + (long)blockIdx.x * D + c) = (a + b) + (d + e);
            }
        }
    }
}

__global__ __launch_bounds__(256) void attn_delta_x3_kernel(const float* __restrict__ o, const bf16* __restrict__ d_o, long doplane,
                                                            float* __restrict__ delta, int nH, long rows) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows * nH) return;
    const long row = w / nH;
    const int h = (int)(w % nH), lane = threadIdx.x & 63;
    const long idx = row * (long)(nH * DH) + h * DH + lane;
    const float v = wave_sum(o[idx] * ((float)d_o[idx] + (float)d_o[idx + doplane]));
    if (lane == 0) delta[row * nH + h] = v;
}

inline float inv_keep_x3(unsigned thresh) { return thresh ? (float)(4294967296.0 / (4294967296.0 - (double)thresh)) : 1.0f; }

}  // namespace

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, float* ws, long ws_floats, void* stream);
extern "C" long mmtg_colsum_ws(int M, int N);

extern "C" int mmtg_attn_fwd_x3(const void* qkv_planes, long qplane, const int* keep, float* out, void* out_planes, long plane, float* lse,
                                int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream) {
    MMTG_REQUIRE(dh == DH, "attn_fwd_x3: head dim %d unsupported (built for 64)", dh);
    MMTG_REQUIRE(B > 0 && T > 0 && nH > 0 && qkv_planes && keep && out && lse, "attn_fwd_x3: bad sizes / null pointer");
    MMTG_REQUIRE(MMTG_ALIGNED16(qkv_planes) && qplane % 8 == 0 && qplane >= 3L * B * T * nH * DH && MMTG_ALIGNED16(out) &&
                 (((uintptr_t)out_planes) & 7) == 0, "attn_fwd_x3: alignment / qkv plane layout");
    MMTG_REQUIRE(!out_planes || (plane % 4 == 0 && plane >= (long)B * T * nH * DH), "attn_fwd_x3: the lo plane must lie behind the hi plane");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ATTN_FWD, s, 2.0 * B * nH * (double)T * T * dh, 4.0 * 5.0 * B * T * nH * dh);
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3(cdiv(T, 64), nH, B), dim3(256), 0, s, (const bf16*)qkv_planes, qplane, keep, out, (bf16*)out_planes, plane, lse, T, nH,
                       drop_thresh, drop_seed, inv_keep_x3(drop_thresh));
    MMTG_LAUNCH_CHECK("attn_fwd_x3");
    return MMTG_OK;
}

/* Backward of mmtg_attn_fwd_x3.  qkv [B*T, 3D] and dout [B*T, D] as (hi | lo) bf16 plane pairs (lo planes `qplane` / `doplane` elements
 * behind), out fp32; d(qkv) is written as a plane pair [B*T, 3D] (lo plane `dplane` elements behind) -- the c_attn dgrad and weight
 * gradient are split-precision products and nothing else reads it.  dq32: fp32 scratch of ceil(T / 128) x [B*T, D] floats (every block
 * of 128 keys stores its dQ contribution into its own buffer, the finish kernel adds them in block order: no atomics, bit-reproducible);
 * delta: [B*T, nH] scratch; dbias (nullable):
 * [3D] += column sums of d(qkv); dbias_ws: >= (B * ceil(T / 128) + ceil(B*T / 16)) * 3D floats + mmtg_colsum_ws of the two partial-row
 * reductions (more than 2048 partial rows: batch 256 at T = 236 -- the ordered two-stage sum, never the fp32-atomic fallback);
 * delta_ready != 0: delta was filled by the caller (the c_proj dgrad's MMTG_EPI_ROWDOT epilogue). */
extern "C" int mmtg_attn_bwd_x3(const void* qkv_planes, long qplane, const int* keep, const float* out, const void* dout_planes, long doplane,
                                const float* lse, float* delta,
                                int delta_ready, float* dq32, long dq32_floats, void* dqkv_planes, long dplane, float* dbias, float* dbias_ws, long dbias_ws_floats,
                                int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream) {
    MMTG_REQUIRE(dh == DH, "attn_bwd_x3: head dim %d unsupported (built for 64)", dh);
    MMTG_REQUIRE(B > 0 && T > 0 && nH > 0 && qkv_planes && keep && out && dout_planes && lse && delta && dq32 && dqkv_planes,
                 "attn_bwd_x3: bad sizes / null pointer");
    const long rows = (long)B * T;
    const int D = nH * dh;
    MMTG_REQUIRE(MMTG_ALIGNED16(qkv_planes) && qplane % 8 == 0 && qplane >= rows * 3 * D && MMTG_ALIGNED16(dout_planes) && doplane % 8 == 0 &&
                 doplane >= rows * D && MMTG_ALIGNED16(dq32) && (((uintptr_t)dqkv_planes) & 7) == 0 && dplane % 4 == 0 && dplane >= rows * 3 * D,
                 "attn_bwd_x3: alignment / plane layout");
    const int nkb = cdiv(T, XKB), nband = cdiv(rows, XFB);
    MMTG_REQUIRE(dq32_floats >= (long)nkb * rows * D, "attn_bwd_x3: dq32 needs %ld floats (one [B*T, D] buffer per block of 128 keys)", (long)nkb * rows * D);
    const long ws_rows = ((long)B * nkb + nband) * 3 * D, ws_kv = mmtg_colsum_ws(B * nkb, 3 * D), ws_q = mmtg_colsum_ws(nband, D);
    MMTG_REQUIRE(!dbias || (dbias_ws && dbias_ws_floats >= ws_rows + ws_kv + ws_q), "attn_bwd_x3: the bias gradient needs %ld workspace floats",
                 ws_rows + ws_kv + ws_q);
    // dbias == NULL with a workspace (round 6): the partial bias rows only -- kv rows [B * nkb][3D] then q rows [nband][D] at the head of
    // dbias_ws -- for the caller's mmtg_colsum_batch (both <= 2048 rows there) instead of the two ordered sums of this call
    const bool rows_only = !dbias && dbias_ws;
    MMTG_REQUIRE(!rows_only || dbias_ws_floats >= ws_rows, "attn_bwd_x3: the partial bias rows need %ld workspace floats", ws_rows);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ATTN_BWD, s, 5.0 * B * nH * (double)T * T * dh, 4.0 * 8.0 * B * T * nH * dh);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_bwd_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BwdLds::END) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "attn_bwd_x3: cannot raise dynamic LDS to %d", BwdLds::END);
        attr_set = true;
    }
    if (!delta_ready)
        hipLaunchKernelGGL(attn_delta_x3_kernel, dim3(cdiv(rows * nH, 4)), dim3(256), 0, s, out, (const bf16*)dout_planes, doplane, delta, nH, rows);
    float* const kv_rows = (dbias || rows_only) ? dbias_ws : nullptr;                       // [B * nkb][3D]: k and v parts (q part zero)
    float* const q_rows = (dbias || rows_only) ? dbias_ws + (long)B * nkb * 3 * D : nullptr; // [nband][D]
    hipLaunchKernelGGL(attn_bwd_x3_kernel, dim3(nkb, nH, B), dim3(64 * XNW), BwdLds::END, s, (const bf16*)qkv_planes, qplane, keep, (const bf16*)dout_planes, doplane, lse, delta, dq32, rows * D, (bf16*)dqkv_planes, dplane,
                       kv_rows, T, nH, drop_thresh, drop_seed, inv_keep_x3(drop_thresh));
    hipLaunchKernelGGL(attn_dq_finish_x3_kernel, dim3(nband), dim3(256), 0, s, dq32, rows * D, T, (bf16*)dqkv_planes, dplane, q_rows, rows, D);
    MMTG_LAUNCH_CHECK("attn_bwd_x3");
    if (dbias) {
        // (tall partial-row matrices take mmtg_colsum's ordered two-stage sum through the tail of the workspace: bit-reproducible at any batch)
        int rc = mmtg_colsum(MMTG_F32, kv_rows, 3 * D, B * nkb, 3 * D, dbias, ws_kv ? dbias_ws + ws_rows : nullptr, ws_kv, stream);
        if (rc) return rc;
        rc = mmtg_colsum(MMTG_F32, q_rows, D, nband, D, dbias, ws_q ? dbias_ws + ws_rows + ws_kv : nullptr, ws_q, stream);
        if (rc) return rc;
    }
    return MMTG_OK;
}
