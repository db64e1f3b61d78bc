// Multi-channel experience encoder pieces that are not plain GEMMs:
// the GRU cell (gates), the inner-modal "alpha" attention with its Gaussian
// prior KL term, and the multi-modal "beta" fuser (3-way softmax blend).
// Shapes are tiny (B x 5 steps x 512): these kernels are launch/latency bound;
// they keep everything in registers / wave shuffles and never touch LDS tiles.
#include "common.h"

namespace {

// ------------------------------------------------------------------ GRU cell
// nn.GRU math (reference src/model.py:78-79):
//   r = s(gi_r + gh_r)  z = s(gi_z + gh_z)  n = tanh(gi_n + r * gh_n)  h = (1-z) n + z h_prev
template <typename T>
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const T* __restrict__ gi, long ld_gi, const T* __restrict__ gh, long ld_gh,
        const T* __restrict__ h_prev, long ld_hp, T* __restrict__ h, long ld_h, float* __restrict__ save, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), j = (int)(i % H);
    const long o = (long)b * ld_gh + j, oi = (long)b * ld_gi + j;     // ld_gh = 0: one row for every b (step 0: gh = b_hh)
    const float r = sigmoid_f((float)gi[oi] + (float)gh[o]);
    const float z = sigmoid_f((float)gi[oi + H] + (float)gh[o + H]);
    const float ghn = (float)gh[o + 2 * H];
    const float n = tanhf((float)gi[oi + 2 * H] + r * ghn);
    const float hp = h_prev ? (float)h_prev[(long)b * ld_hp + j] : 0.f;
    h[(long)b * ld_h + j] = (T)((1.f - z) * n + z * hp);
    const long BH = (long)B * H;
    save[i] = r; save[BH + i] = z; save[2 * BH + i] = n; save[3 * BH + i] = ghn;
}

template <typename T>
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ save,
        const T* __restrict__ h_prev, long ld_hp, T* __restrict__ dgi, long ld_dgi, T* __restrict__ dgh,
        float* __restrict__ dh_prev, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), j = (int)(i % H);
    const long BH = (long)B * H;
    const float r = save[i], z = save[BH + i], n = save[2 * BH + i], ghn = save[3 * BH + i];
    const float hp = h_prev ? (float)h_prev[(long)b * ld_hp + j] : 0.f;
    const float d = dh[i];
    const float dz = d * (hp - n);
    const float dn = d * (1.f - z);
    const float da = dn * (1.f - n * n);
    const float dr = da * ghn;
    const float dr_pre = dr * r * (1.f - r);
    const float dz_pre = dz * z * (1.f - z);
    const long o = (long)b * 3 * H + j, oi = (long)b * ld_dgi + j;
    dgi[oi] = (T)dr_pre; dgi[oi + H] = (T)dz_pre; dgi[oi + 2 * H] = (T)da;
    dgh[o] = (T)dr_pre; dgh[o + H] = (T)dz_pre; dgh[o + 2 * H] = (T)(da * r);
    dh_prev[i] = d * z;
}

// Same cell backward with the gradient assembly fused in (one launch per BPTT step instead of a strided copy,
// two adds and the cell): dh_t = rows[b, :] (the LayerNorm path's gradient of step t, storage type, row stride
// ld_rows) + carry (dh_{t+1} * z_{t+1}, f32, nullable) + the ordered sum of the `splits` fp32 slabs of the carry
// product d(gh_{t+1}) W_hh (nullable).
template <typename T>
__global__ __launch_bounds__(256) void gru_cell_bwd_fused_kernel(const T* __restrict__ rows, long ld_rows,
        const float* __restrict__ carry, const float* __restrict__ part, int splits, const float* __restrict__ save,
        const T* __restrict__ h_prev, long ld_hp, T* __restrict__ dgi, long ld_dgi, T* __restrict__ dgh,
        float* __restrict__ dh_prev, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), j = (int)(i % H);
    const long BH = (long)B * H;
    float d = (float)rows[(long)b * ld_rows + j];
    if (carry) d += carry[i];
    for (int k = 0; k < splits; ++k) d += part[k * BH + i];
    const float r = save[i], z = save[BH + i], n = save[2 * BH + i], ghn = save[3 * BH + i];
    const float hp = h_prev ? (float)h_prev[(long)b * ld_hp + j] : 0.f;
    const float dz = d * (hp - n);
    const float dn = d * (1.f - z);
    const float da = dn * (1.f - n * n);
    const float dr = da * ghn;
    const float dr_pre = dr * r * (1.f - r);
    const float dz_pre = dz * z * (1.f - z);
    const long o = (long)b * 3 * H + j, oi = (long)b * ld_dgi + j;
    dgi[oi] = (T)dr_pre; dgi[oi + H] = (T)dz_pre; dgi[oi + 2 * H] = (T)da;
    dgh[o] = (T)dr_pre; dgh[o + H] = (T)dz_pre; dgh[o + 2 * H] = (T)(da * r);
    dh_prev[i] = d * z;
}

// ------------------------------------------------------------------ LSTM / ReLU-RNN cells (the other encoder types of model.py:39-58)
// Pre-activations a = gi + gh (gi = W_ih x_t + b_ih for all steps in one product, gh = W_hh h_{t-1} + b_hh per step).
//   kind 1, nn.LSTM (gate order i, f, g, o):  i = s(a_i) f = s(a_f) g = tanh(a_g) o = s(a_o);  c = f c_prev + i g;  h = o tanh(c)
//   kind 0, nn.RNN(nonlinearity="relu"):     h = max(a, 0)
// The cell state stays fp32 ([B, H] per step); save = [5, B, H] fp32 per step: i, f, g, o, tanh(c).
template <typename T, int KIND>
__global__ __launch_bounds__(256) void rnn_cell_fwd_kernel(const T* __restrict__ gi, long ld_gi, const T* __restrict__ gh, long ld_gh,
        const float* __restrict__ c_prev, T* __restrict__ h, long ld_h, float* __restrict__ c, float* __restrict__ save, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), j = (int)(i % H);
    const long o = (long)b * ld_gh + j, oi = (long)b * ld_gi + j;     // ld_gh = 0: one row for every b (step 0: gh = b_hh)
    if constexpr (KIND == 0) {
        h[(long)b * ld_h + j] = (T)fmaxf((float)gi[oi] + (float)gh[o], 0.f);
    } else {
        const float gi_ = sigmoid_f((float)gi[oi] + (float)gh[o]);
        const float gf = sigmoid_f((float)gi[oi + H] + (float)gh[o + H]);
        const float gg = tanhf((float)gi[oi + 2 * H] + (float)gh[o + 2 * H]);
        const float go = sigmoid_f((float)gi[oi + 3 * H] + (float)gh[o + 3 * H]);
        const float cn = gf * (c_prev ? c_prev[i] : 0.f) + gi_ * gg;
        const float tc = tanhf(cn);
        c[i] = cn;
        h[(long)b * ld_h + j] = (T)(go * tc);
        const long BH = (long)B * H;
        save[i] = gi_; save[BH + i] = gf; save[2 * BH + i] = gg; save[3 * BH + i] = go; save[4 * BH + i] = tc;
    }
}

// Backward of one step with the gradient assembly fused in (as gru_cell_bwd_fused): dh_t = rows[b, :] (gradient of the layer's
// output row of step t: the LayerNorm path, or the layer above) + part (d(a_{t+1}) W_hh, fp32 [B, H], nullable).
// LSTM: dc (fp32 [B, H]) carries dc_{t+1} f_{t+1} in (when dc_in != 0) and dc_t f_t out.  da: [B, G H] rows of stride ld_da.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void rnn_cell_bwd_kernel(const T* __restrict__ rows, long ld_rows, const float* __restrict__ part,
        const float* __restrict__ save, const float* __restrict__ c_prev, const T* __restrict__ h, long ld_h, float* __restrict__ dc,
        int dc_in, T* __restrict__ da, long ld_da, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), j = (int)(i % H);
    float d = (float)rows[(long)b * ld_rows + j];
    if (part) d += part[i];
    const long oa = (long)b * ld_da + j;
    if constexpr (KIND == 0) {
        da[oa] = (T)((float)h[(long)b * ld_h + j] > 0.f ? d : 0.f);
    } else {
        const long BH = (long)B * H;
        const float gi_ = save[i], gf = save[BH + i], gg = save[2 * BH + i], go = save[3 * BH + i], tc = save[4 * BH + i];
        const float dct = (dc_in ? dc[i] : 0.f) + d * go * (1.f - tc * tc);
        da[oa] = (T)(dct * gg * gi_ * (1.f - gi_));
        da[oa + H] = (T)(dct * (c_prev ? c_prev[i] : 0.f) * gf * (1.f - gf));
        da[oa + 2 * H] = (T)(dct * gi_ * (1.f - gg * gg));
        da[oa + 3 * H] = (T)(d * tc * go * (1.f - go));
        dc[i] = dct * gf;
    }
}

// ------------------------------------------------------------------ alpha attention
// One wave per (b, head): S <= 8 steps, dh = H/heads <= 256.  Lane owns dh/64
// consecutive channels; QK^T dot products by wave reduction.
constexpr int AS_MAX = 8;

// (S is a template parameter: with a run-time S the score arrays are indexed dynamically and live in scratch
//  memory -- 272 / 528 bytes per lane, 36 us per launch; unrolled they are registers)
template <typename T, int S>
__global__ __launch_bounds__(64) void alpha_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ prior,
        T* __restrict__ ctx, float* __restrict__ probs, float* __restrict__ kl, int B, int H, int heads) {
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int dh = H / heads, lane = threadIdx.x;
    const float scale = rsqrtf((float)dh);
    float sc[S][S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float a = 0.f;
            for (int c = lane; c < dh; c += 64) {
                const float q = (float)qkv[((long)b * S + i) * 3 * H + hd * dh + c];
                const float k = (float)qkv[((long)b * S + j) * 3 * H + H + hd * dh + c];
                a += q * k;
            }
            sc[i][j] = wave_sum(a) * scale;
        }
    float klacc = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        float mx = sc[i][0];
#pragma unroll
        for (int j = 1; j < S; ++j) mx = fmaxf(mx, sc[i][j]);
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < S; ++j) { sc[i][j] = expf(sc[i][j] - mx); sm += sc[i][j]; }
        const float inv = 1.f / sm;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            sc[i][j] *= inv;
            const float q = prior[i * S + j];
            klacc += q * (logf(q) - logf(sc[i][j]));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) probs[(((long)b * heads + hd) * S + i) * S + j] = sc[i][j];
        atomicAdd(kl, klacc / ((float)B * S));
    }
    for (int c = lane; c < dh; c += 64) {
        float vv[S];
#pragma unroll
        for (int j = 0; j < S; ++j) vv[j] = (float)qkv[((long)b * S + j) * 3 * H + 2 * H + hd * dh + c];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) a += sc[i][j] * vv[j];
            ctx[((long)b * S + i) * H + hd * dh + c] = (T)a;
        }
    }
}

template <typename T, int S>
__global__ __launch_bounds__(64) void alpha_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ prior,
        const float* __restrict__ probs, const T* __restrict__ dctx, float dkl, T* __restrict__ dqkv,
        int B, int H, int heads) {
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int dh = H / heads, lane = threadIdx.x;
    const float scale = rsqrtf((float)dh);
    float P[S][S], dS[S][S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) P[i][j] = probs[(((long)b * heads + hd) * S + i) * S + j];
    // dP[i][j] = dctx[i] . V[j]  - dkl * prior[i][j] / (P[i][j] * B * S)
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float a = 0.f;
            for (int c = lane; c < dh; c += 64)
                a += (float)dctx[((long)b * S + i) * H + hd * dh + c] * (float)qkv[((long)b * S + j) * 3 * H + 2 * H + hd * dh + c];
            dS[i][j] = wave_sum(a) - dkl * prior[i * S + j] / (P[i][j] * (float)B * S);
        }
#pragma unroll
    for (int i = 0; i < S; ++i) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < S; ++j) dot += P[i][j] * dS[i][j];
#pragma unroll
        for (int j = 0; j < S; ++j) dS[i][j] = P[i][j] * (dS[i][j] - dot) * scale;
    }
    for (int c = lane; c < dh; c += 64) {
        float q[S], k[S], dc[S];
#pragma unroll
        for (int j = 0; j < S; ++j) {
            q[j] = (float)qkv[((long)b * S + j) * 3 * H + hd * dh + c];
            k[j] = (float)qkv[((long)b * S + j) * 3 * H + H + hd * dh + c];
            dc[j] = (float)dctx[((long)b * S + j) * H + hd * dh + c];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                dq += dS[i][j] * k[j];
                dk += dS[j][i] * q[j];
                dv += P[j][i] * dc[j];
            }
            const long o = ((long)b * S + i) * 3 * H + hd * dh + c;
            dqkv[o] = (T)dq; dqkv[o + H] = (T)dk; dqkv[o + 2 * H] = (T)dv;
        }
    }
}

// ------------------------------------------------------------------ beta fuser
// One wave per (b, i): scores of {topic, img_i, txt_i} with step i's own weight row,
// softmax over the 3, blend.  H <= 1024 (lane keeps H/64 channels).
constexpr int BF_MAXC = 16;

template <typename T>
__global__ __launch_bounds__(64) void beta_fwd_kernel(const T* __restrict__ topic, const T* __restrict__ img,
        const T* __restrict__ txt, const float* __restrict__ att_w, const float* __restrict__ att_b,
        T* __restrict__ o, float* __restrict__ a_out, int B, int S, int H) {
    const int b = blockIdx.x / S, i = blockIdx.x % S, lane = threadIdx.x;
    const long r = (long)b * S + i;
    float tv[BF_MAXC], iv[BF_MAXC], xv[BF_MAXC];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    int n = 0;
    for (int c = lane; c < H; c += 64, ++n) {
        const float w = att_w[i * H + c];
        tv[n] = (float)topic[(long)b * H + c];
        iv[n] = (float)img[r * H + c];
        xv[n] = (float)txt[r * H + c];
        s0 += w * tv[n]; s1 += w * iv[n]; s2 += w * xv[n];
    }
    const float bb = att_b[i];
    s0 = wave_sum(s0) + bb; s1 = wave_sum(s1) + bb; s2 = wave_sum(s2) + bb;
    const float mx = fmaxf(s0, fmaxf(s1, s2));
    float e0 = expf(s0 - mx), e1 = expf(s1 - mx), e2 = expf(s2 - mx);
    const float inv = 1.f / (e0 + e1 + e2);
    e0 *= inv; e1 *= inv; e2 *= inv;
    n = 0;
    for (int c = lane; c < H; c += 64, ++n) o[r * H + c] = (T)(e0 * tv[n] + e1 * iv[n] + e2 * xv[n]);
    if (lane == 0) { a_out[r * 3] = e0; a_out[r * 3 + 1] = e1; a_out[r * 3 + 2] = e2; }
}

// Backward of the fuser.  ws != null (round 4, the engine's form): one workgroup of S waves per batch row, wave i = step i; d topic
// -- the sum over the steps -- is folded through LDS in step order and written by its single owner; the per-(b, step)
// contributions to the step's weight row / bias go to a workspace [B][S*H] | [B][S] that the host sums over b in a fixed order:
// no fp32 atomics, so the att_matrices, topic_fc and ln_layer1 gradients are reproducible bit for bit.  ws == null keeps the
// round-1 form (grid = B x S single-wave workgroups, atomics).
template <typename T>
__global__ __launch_bounds__(1024) void beta_bwd_kernel(const T* __restrict__ topic, const T* __restrict__ img,
        const T* __restrict__ txt, const float* __restrict__ att_w, const float* __restrict__ a_in,
        const T* __restrict__ d_o, float* __restrict__ dtopic, T* __restrict__ dimg, T* __restrict__ dtxt,
        float* __restrict__ datt_w, float* __restrict__ datt_b, int B, int S, int H, float* __restrict__ ws) {
    extern __shared__ float sdt[];           // [S][H] (ws form)
    const int lane = threadIdx.x & 63;
    const int b = ws ? blockIdx.x : blockIdx.x / S;
    const int i = ws ? (int)(threadIdx.x >> 6) : blockIdx.x % S;
    const long r = (long)b * S + i;
    const float a0 = a_in[r * 3], a1 = a_in[r * 3 + 1], a2 = a_in[r * 3 + 2];
    float tv[BF_MAXC], iv[BF_MAXC], xv[BF_MAXC], dv[BF_MAXC];
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    int n = 0;
    for (int c = lane; c < H; c += 64, ++n) {
        tv[n] = (float)topic[(long)b * H + c];
        iv[n] = (float)img[r * H + c];
        xv[n] = (float)txt[r * H + c];
        dv[n] = (float)d_o[r * H + c];
        d0 += dv[n] * tv[n]; d1 += dv[n] * iv[n]; d2 += dv[n] * xv[n];
    }
    d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
    const float dot = a0 * d0 + a1 * d1 + a2 * d2;
    const float ds0 = a0 * (d0 - dot), ds1 = a1 * (d1 - dot), ds2 = a2 * (d2 - dot);
    n = 0;
    for (int c = lane; c < H; c += 64, ++n) {
        const float w = att_w[i * H + c];
        const float gt = a0 * dv[n] + ds0 * w, gw = ds0 * tv[n] + ds1 * iv[n] + ds2 * xv[n];
        dimg[r * H + c] = (T)(a1 * dv[n] + ds1 * w);
        dtxt[r * H + c] = (T)(a2 * dv[n] + ds2 * w);
        if (ws) { sdt[i * H + c] = gt; ws[((long)b * S + i) * H + c] = gw; }
        else { atomicAdd(dtopic + (long)b * H + c, gt); atomicAdd(datt_w + i * H + c, gw); }
    }
    if (lane == 0) {
        if (ws) ws[(long)B * S * H + (long)b * S + i] = ds0 + ds1 + ds2;
        else atomicAdd(datt_b + i, ds0 + ds1 + ds2);
    }
    if (ws) {
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < S; ++k) t += sdt[k * H + c];
            dtopic[(long)b * H + c] += t;
        }
    }
}

}  // namespace

#define DISPATCH(dtype, KERN)                                              \
    if ((dtype) == MMTG_F32) { KERN(float); }                               \
    else if ((dtype) == MMTG_BF16) { KERN(bf16); }                          \
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "bad dtype %d", (dtype));

extern "C" int mmtg_gru_cell_fwd(int dtype, const void* gi, long ld_gi, const void* gh, long ld_gh, const void* h_prev, long ld_hp,
                                 void* h, long ld_h, float* save, int B, int H, void* stream) {
    MMTG_REQUIRE(gi && gh && h && save && B > 0 && H > 0 && (ld_gh == 0 || ld_gh >= 3L * H), "gru_cell_fwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 20.0 * B * H, 30.0 * B * H);
    dim3 grid(cdiv((long)B * H, 256)), block(256);
#define K_(T) hipLaunchKernelGGL(gru_cell_fwd_kernel<T>, grid, block, 0, s, (const T*)gi, ld_gi, (const T*)gh, ld_gh, (const T*)h_prev, ld_hp, (T*)h, ld_h, save, B, H)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("gru_cell_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_gru_cell_bwd(int dtype, const float* dh, const float* save, const void* h_prev, long ld_hp,
                                 void* dgi, long ld_dgi, void* dgh, float* dh_prev, int B, int H, void* stream) {
    MMTG_REQUIRE(dh && save && dgi && dgh && dh_prev && B > 0 && H > 0, "gru_cell_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 20.0 * B * H, 40.0 * B * H);
    dim3 grid(cdiv((long)B * H, 256)), block(256);
#define K_(T) hipLaunchKernelGGL(gru_cell_bwd_kernel<T>, grid, block, 0, s, dh, save, (const T*)h_prev, ld_hp, (T*)dgi, ld_dgi, (T*)dgh, dh_prev, B, H)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("gru_cell_bwd");
    return MMTG_OK;
}

extern "C" int mmtg_gru_cell_bwd_fused(int dtype, const void* rows, long ld_rows, const float* carry, const float* part, int splits,
                                       const float* save, const void* h_prev, long ld_hp, void* dgi, long ld_dgi, void* dgh,
                                       float* dh_prev, int B, int H, void* stream) {
    MMTG_REQUIRE(rows && save && dgi && dgh && dh_prev && B > 0 && H > 0 && splits >= 0 && (splits == 0 || part), "gru_cell_bwd_fused: bad args");
    // (carry may alias dh_prev: a thread reads its element before it writes it)
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 24.0 * B * H, (44.0 + 4.0 * splits) * B * H);
    dim3 grid(cdiv((long)B * H, 256)), block(256);
#define K_(T) hipLaunchKernelGGL(gru_cell_bwd_fused_kernel<T>, grid, block, 0, s, (const T*)rows, ld_rows, carry, part, splits, save, \
                                 (const T*)h_prev, ld_hp, (T*)dgi, ld_dgi, (T*)dgh, dh_prev, B, H)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("gru_cell_bwd_fused");
    return MMTG_OK;
}

extern "C" int mmtg_rnn_cell_fwd(int dtype, int kind, const void* gi, long ld_gi, const void* gh, long ld_gh, const float* c_prev,
                                 void* h, long ld_h, float* c, float* save, int B, int H, void* stream) {
    MMTG_REQUIRE(kind == MMTG_RNN_RELU || kind == MMTG_RNN_LSTM, "rnn_cell_fwd: kind %d", kind);
    const int G = kind == MMTG_RNN_LSTM ? 4 : 1;
    MMTG_REQUIRE(gi && gh && h && B > 0 && H > 0 && (ld_gh == 0 || ld_gh >= (long)G * H) && ld_gi >= (long)G * H && ld_h >= H, "rnn_cell_fwd: bad args");
    MMTG_REQUIRE(kind != MMTG_RNN_LSTM || (c && save), "rnn_cell_fwd: the LSTM cell needs its state and save buffers");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 8.0 * G * B * H, 12.0 * G * B * H);
    dim3 grid(cdiv((long)B * H, 256)), block(256);
#define K_(T) do { if (kind == MMTG_RNN_LSTM) hipLaunchKernelGGL((rnn_cell_fwd_kernel<T, 1>), grid, block, 0, s, (const T*)gi, ld_gi, (const T*)gh, ld_gh, c_prev, (T*)h, ld_h, c, save, B, H); \
                   else hipLaunchKernelGGL((rnn_cell_fwd_kernel<T, 0>), grid, block, 0, s, (const T*)gi, ld_gi, (const T*)gh, ld_gh, c_prev, (T*)h, ld_h, c, save, B, H); } while (0)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("rnn_cell_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_rnn_cell_bwd(int dtype, int kind, const void* rows, long ld_rows, const float* part, const float* save,
                                 const float* c_prev, const void* h, long ld_h, float* dc, int dc_in, void* da, long ld_da,
                                 int B, int H, void* stream) {
    MMTG_REQUIRE(kind == MMTG_RNN_RELU || kind == MMTG_RNN_LSTM, "rnn_cell_bwd: kind %d", kind);
    const int G = kind == MMTG_RNN_LSTM ? 4 : 1;
    MMTG_REQUIRE(rows && da && B > 0 && H > 0 && ld_rows >= H && ld_da >= (long)G * H, "rnn_cell_bwd: bad args");
    MMTG_REQUIRE(kind == MMTG_RNN_LSTM ? (save && dc) : (h && ld_h >= H), "rnn_cell_bwd: LSTM needs save + dc, the ReLU cell its output h");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 10.0 * G * B * H, 14.0 * G * B * H);
    dim3 grid(cdiv((long)B * H, 256)), block(256);
#define K_(T) do { if (kind == MMTG_RNN_LSTM) hipLaunchKernelGGL((rnn_cell_bwd_kernel<T, 1>), grid, block, 0, s, (const T*)rows, ld_rows, part, save, c_prev, (const T*)h, ld_h, dc, dc_in, (T*)da, ld_da, B, H); \
                   else hipLaunchKernelGGL((rnn_cell_bwd_kernel<T, 0>), grid, block, 0, s, (const T*)rows, ld_rows, part, save, c_prev, (const T*)h, ld_h, dc, dc_in, (T*)da, ld_da, B, H); } while (0)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("rnn_cell_bwd");
    return MMTG_OK;
}

extern "C" int mmtg_alpha_attn_fwd(int dtype, const void* qkv, const float* prior, void* ctx, float* probs,
                                   float* kl, int B, int S, int H, int heads, void* stream) {
    MMTG_REQUIRE(qkv && prior && ctx && probs && kl, "alpha_attn_fwd: null pointer");
    MMTG_REQUIRE(B > 0 && S > 0 && S <= AS_MAX && heads > 0 && H % heads == 0, "alpha_attn_fwd: S=%d must be <= %d, H %% heads == 0", S, AS_MAX);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 4.0 * B * S * S * H, 8.0 * B * S * H);
    dim3 grid(B * heads), block(64);
#define KS_(T, S_) hipLaunchKernelGGL((alpha_fwd_kernel<T, S_>), grid, block, 0, s, (const T*)qkv, prior, (T*)ctx, probs, kl, B, H, heads)
#define K_(T) switch (S) { case 1: KS_(T, 1); break; case 2: KS_(T, 2); break; case 3: KS_(T, 3); break; case 4: KS_(T, 4); break; \
                           case 5: KS_(T, 5); break; case 6: KS_(T, 6); break; case 7: KS_(T, 7); break; default: KS_(T, 8); break; }
    DISPATCH(dtype, K_)
#undef K_
#undef KS_
    MMTG_LAUNCH_CHECK("alpha_attn_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_alpha_attn_bwd(int dtype, const void* qkv, const float* prior, const float* probs,
                                   const void* dctx, float dkl, void* dqkv, int B, int S, int H, int heads, void* stream) {
    MMTG_REQUIRE(qkv && prior && probs && dctx && dqkv, "alpha_attn_bwd: null pointer");
    MMTG_REQUIRE(B > 0 && S > 0 && S <= AS_MAX && heads > 0 && H % heads == 0, "alpha_attn_bwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_ENCODER, s, 8.0 * B * S * S * H, 16.0 * B * S * H);
    dim3 grid(B * heads), block(64);
#define KS_(T, S_) hipLaunchKernelGGL((alpha_bwd_kernel<T, S_>), grid, block, 0, s, (const T*)qkv, prior, probs, (const T*)dctx, dkl, (T*)dqkv, B, H, heads)
#define K_(T) switch (S) { case 1: KS_(T, 1); break; case 2: KS_(T, 2); break; case 3: KS_(T, 3); break; case 4: KS_(T, 4); break; \
                           case 5: KS_(T, 5); break; case 6: KS_(T, 6); break; case 7: KS_(T, 7); break; default: KS_(T, 8); break; }
    DISPATCH(dtype, K_)
#undef K_
#undef KS_
    MMTG_LAUNCH_CHECK("alpha_attn_bwd");
    return MMTG_OK;
}

extern "C" int mmtg_beta_fuse_fwd(int dtype, const void* topic, const void* img, const void* txt,
                                  const float* att_w, const float* att_b, void* o, float* a,
                                  int B, int S, int H, void* stream) {
    MMTG_REQUIRE(topic && img && txt && att_w && att_b && o && a, "beta_fuse_fwd: null pointer");
    MMTG_REQUIRE(B > 0 && S > 0 && H > 0 && H <= 64 * BF_MAXC, "beta_fuse_fwd: H=%d too large", H);
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_ENCODER, s, 12.0 * B * S * H, esz * (3.0 * B * S * H + B * H));
    dim3 grid(B * S), block(64);
#define K_(T) hipLaunchKernelGGL(beta_fwd_kernel<T>, grid, block, 0, s, (const T*)topic, (const T*)img, (const T*)txt, att_w, att_b, (T*)o, a, B, S, H)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("beta_fuse_fwd");
    return MMTG_OK;
}

extern "C" int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, float* ws, long ws_floats, void* stream);
extern "C" long mmtg_beta_fuse_bwd_ws(int B, int S, int H) { return (long)B * S * (H + 1); }

extern "C" int mmtg_beta_fuse_bwd(int dtype, const void* topic, const void* img, const void* txt,
                                  const float* att_w, const float* a, const void* d_o,
                                  float* dtopic, void* dimg, void* dtxt, float* datt_w, float* datt_b,
                                  int B, int S, int H, float* ws, long ws_floats, void* stream) {
    MMTG_REQUIRE(topic && img && txt && att_w && a && d_o && dtopic && dimg && dtxt && datt_w && datt_b, "beta_fuse_bwd: null pointer");
    MMTG_REQUIRE(B > 0 && S > 0 && H > 0 && H <= 64 * BF_MAXC, "beta_fuse_bwd: H=%d too large", H);
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_ENCODER, s, 24.0 * B * S * H, esz * 6.0 * B * S * H);
    MMTG_REQUIRE(!ws || ws_floats >= mmtg_beta_fuse_bwd_ws(B, S, H), "beta_fuse_bwd: workspace of %ld floats required", mmtg_beta_fuse_bwd_ws(B, S, H));
    MMTG_REQUIRE(!ws || S <= 16, "beta_fuse_bwd: at most 16 experience steps in the workspace form");
    dim3 grid(ws ? B : B * S), block(ws ? 64 * S : 64);
    const size_t shm = ws ? (size_t)S * H * sizeof(float) : 0;
#define K_(T) hipLaunchKernelGGL(beta_bwd_kernel<T>, grid, block, shm, s, (const T*)topic, (const T*)img, (const T*)txt, att_w, a, (const T*)d_o, dtopic, (T*)dimg, (T*)dtxt, datt_w, datt_b, B, S, H, ws)
    DISPATCH(dtype, K_)
#undef K_
    MMTG_LAUNCH_CHECK("beta_fuse_bwd");
    if (ws) {       // ordered sums over the batch rows: the step weights [S*H] and biases [S]
        int rc = mmtg_colsum(MMTG_F32, ws, (long)S * H, B, S * H, datt_w, nullptr, 0, stream);
        if (rc) return rc;
        return mmtg_colsum(MMTG_F32, ws + (long)B * S * H, (long)S, B, S, datt_b, nullptr, 0, stream);
    }
    return MMTG_OK;
}
