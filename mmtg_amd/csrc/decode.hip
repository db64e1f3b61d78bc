// Generation-side kernels (reference src/generate.py:127-141): the fused logits
// processor + greedy arg-max.  One workgroup per batch row.  The reference divides a logit by
// the repetition penalty once PER OCCURRENCE of its id among the generated tokens; the workgroup
// first counts the occurrences into a 16-bit-per-vocabulary-slot LDS table (LDS atomics), then
// every lane applies that many sequential fp32 divisions to its own slots -- bit-for-bit the
// reference's arithmetic, in O(V + n) instead of the O(V n) scan of the first version (which made
// this kernel 9 % of a decode step: 122 us at 220 generated tokens).
#include "gemm_common.h"

namespace {

constexpr int MAXGEN = 2048;      // < 65536: the counts are 16-bit

// cnt: (V + 1) / 2 words of dynamic LDS.  Counts the ids gen[0..n) (all threads of the block call).
__device__ __forceinline__ void count_ids(unsigned* cnt, int V, const long long* gen, int n, int tid) {
    for (int i = tid; i < (V + 1) / 2; i += 256) cnt[i] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int v = (int)gen[i];
        if (v >= 0 && v < V) atomicAdd(cnt + (v >> 1), 1u << (16 * (v & 1)));
    }
    __syncthreads();
}

// one logit through the reference's processor (generate.py:127-136), folded into the running arg-max (lowest index wins ties)
__device__ __forceinline__ void scan_one(float x, int v, const unsigned* cnt, float temperature, float rep_penalty, float& best, int& besti) {
    if (v != 0 && v != 102) {
        const int c = (cnt[v >> 1] >> (16 * (v & 1))) & 0xFFFF;
        for (int i = 0; i < c; ++i) x = x / rep_penalty;
    }
    x = x / temperature;
    if (v == 1 || v == 2 || v == 100 || v == 102) x = -INFINITY;
    if (x > best || (x == best && v < besti)) { best = x; besti = v; }
}
// this thread's share of the processed-logit arg-max.  16-byte aligned rows are read as float4 with all of a pass's loads
// (up to 16 per thread = 16 K logits per block) requested before the first is used: one memory round trip per row instead of a
// dependent 4-byte load per 256 logits (round 3: 21.5 -> see profiles/ us for 256 rows of V = 13317).
__device__ __forceinline__ void scan_row(const float* row, int V, const unsigned* cnt, float temperature, float rep_penalty,
                                         int tid, float& best, int& besti) {
    best = -INFINITY;
    besti = 0x7fffffff;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        constexpr int UN = 16;
        const int nvec = V >> 2;
        const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
        for (int base = 0; base < nvec; base += 256 * UN) {
            f32x4 r[UN];
#pragma unroll
            for (int i = 0; i < UN; ++i) {
                const int idx = base + i * 256 + tid;
                r[i] = idx < nvec ? row4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < UN; ++i) {
                const int idx = base + i * 256 + tid;
                if (idx < nvec) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) scan_one(r[i][e], 4 * idx + e, cnt, temperature, rep_penalty, best, besti);
                }
            }
        }
        done = nvec << 2;
    }
    for (int v = done + tid; v < V; v += 256) scan_one(row[v], v, cnt, temperature, rep_penalty, best, besti);
}

__global__ __launch_bounds__(256) void logits_argmax_kernel(const float* __restrict__ logits, long ldl, int V,
        const long long* __restrict__ generated, long ldg, const int* __restrict__ gen_len,
        float temperature, float rep_penalty, long long* __restrict__ next) {
    extern __shared__ unsigned cnt[];
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(gen_len[b], MAXGEN);
    const long long* gen = generated + (long)b * ldg;
    if (n > 0 && gen[n - 1] == 0) {  // sticky PAD (generate.py:137-138)
        if (tid == 0) next[b] = 0;
        return;
    }
    count_ids(cnt, V, gen, n, tid);
    float best;
    int besti;
    scan_row(logits + (long)b * ldl, V, cnt, temperature, rep_penalty, tid, best, besti);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((tid & 63) == 0) { sval[tid >> 6] = best; sidx[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < besti)) { best = sval[w]; besti = sidx[w]; }
        next[b] = besti == 0x7fffffff ? 0 : besti;
    }
}

// ------------------------------------------------------------------ stochastic selection (generate.py:64-94,127-141)
// One workgroup per row: processed logits (penalty, temperature, bans, as above) -> top-k filter
// (`logits < kth largest` dropped, ties at the k-th value kept) -> nucleus filter over the survivors (sorted
// descending, an element stays while the probability mass of the elements before it is <= top_p) -> softmax ->
// one draw by inverse CDF over the kept ids in index order with the caller's uniform u in [0,1).
// No sort: both thresholds come from a 4-pass radix select over the order-preserving integer image of the
// floats -- by COUNT for top-k, by accumulated probability MASS for top-p.  px: V floats of LDS.
__device__ __forceinline__ unsigned fkey(float x) {
    const unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ int sample_row(const float* row, int V, const unsigned* cnt, float* px, float temperature, float rep_penalty,
                          int top_k, float top_p, float u, int tid, float* filtered) {
    __shared__ unsigned hcnt[256];
    __shared__ float hmass[256];
    __shared__ float sred[4];
    __shared__ unsigned s_prefix, s_rem;
    __shared__ float s_acc;
    __shared__ int s_pick;
    for (int v = tid; v < V; v += 256) {
        float x = row[v];
        if (v != 0 && v != 102) {
            const int c = (cnt[v >> 1] >> (16 * (v & 1))) & 0xFFFF;
            for (int i = 0; i < c; ++i) x = x / rep_penalty;
        }
        x = x / temperature;
        if (v == 1 || v == 2 || v == 100 || v == 102) x = -INFINITY;
        px[v] = x;
    }
    __syncthreads();
    // ---- top-k: key of the k-th largest value
    unsigned keyk = 0u;
    if (top_k > 0 && top_k < V) {
        if (tid == 0) { s_prefix = 0u; s_rem = (unsigned)top_k; }
        unsigned mask = 0u;
        for (int shift = 24; shift >= 0; shift -= 8) {
            hcnt[tid] = 0u;
            __syncthreads();
            const unsigned prefix = s_prefix;
            for (int v = tid; v < V; v += 256) {
                const unsigned k = fkey(px[v]);
                if ((k & mask) == prefix) atomicAdd(hcnt + ((k >> shift) & 255u), 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned rem = s_rem;
                int bin = 255;
                for (; bin > 0; --bin) {
                    if (hcnt[bin] >= rem) break;
                    rem -= hcnt[bin];
                }
                s_rem = rem;
                s_prefix = prefix | ((unsigned)bin << shift);
            }
            mask |= 255u << shift;
            __syncthreads();
        }
        keyk = s_prefix;
    }
    // ---- max and total mass of the survivors
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256)
        if (fkey(px[v]) >= keyk) mx = fmaxf(mx, px[v]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) sred[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float tot = 0.f;
    for (int v = tid; v < V; v += 256)
        if (fkey(px[v]) >= keyk) tot += expf(px[v] - mx);
    tot = wave_sum(tot);
    if ((tid & 63) == 0) sred[tid >> 6] = tot;
    __syncthreads();
    tot = sred[0] + sred[1] + sred[2] + sred[3];
    __syncthreads();
    // ---- top-p: key of the first element (descending) whose inclusive mass exceeds top_p
    unsigned keyp = 0u;
    if (top_p > 0.f) {
        const float P = top_p * tot;
        if (tid == 0) { s_prefix = 0u; s_acc = 0.f; s_pick = 1; }
        unsigned mask = 0u;
        for (int shift = 24; shift >= 0; shift -= 8) {
            hmass[tid] = 0.f;
            __syncthreads();
            const unsigned prefix = s_prefix;
            if (s_pick)
                for (int v = tid; v < V; v += 256) {
                    const unsigned k = fkey(px[v]);
                    if (k >= keyk && (k & mask) == prefix) atomicAdd(hmass + ((k >> shift) & 255u), expf(px[v] - mx));
                }
            __syncthreads();
            if (tid == 0 && s_pick) {
                float acc = s_acc;
                int bin = 255;
                for (; bin >= 0; --bin) {
                    if (acc + hmass[bin] > P) break;
                    acc += hmass[bin];
                }
                if (bin < 0) s_pick = 0;                 // the whole mass stays within top_p: keep every survivor
                else { s_acc = acc; s_prefix = prefix | ((unsigned)bin << shift); }
            }
            mask |= 255u << shift;
            __syncthreads();
        }
        keyp = s_pick ? s_prefix : 0u;
        __syncthreads();
    }
    const unsigned keyt = keyk > keyp ? keyk : keyp;
    if (filtered)
        for (int v = tid; v < V; v += 256) filtered[v] = fkey(px[v]) >= keyt ? px[v] : -INFINITY;
    // ---- inverse CDF over the kept ids in index order
    const int chunk = (V + 255) / 256, v0 = tid * chunk, v1 = min(V, v0 + chunk);
    float mine = 0.f;
    for (int v = v0; v < v1; ++v)
        if (fkey(px[v]) >= keyt && px[v] > -INFINITY) mine += expf(px[v] - mx);
    // inclusive scan of the 256 chunk sums (in hmass)
    hmass[tid] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const float add = tid >= o ? hmass[tid - o] : 0.f;
        __syncthreads();
        hmass[tid] += add;
        __syncthreads();
    }
    const float Z = hmass[255], target = u * Z;
    if (tid == 0) s_pick = 0x7fffffff;
    __syncthreads();
    const float before = tid ? hmass[tid - 1] : 0.f;
    if (mine > 0.f && target < hmass[tid] && target >= before) {
        float acc = before;
        int pick = -1;
        for (int v = v0; v < v1; ++v)
            if (fkey(px[v]) >= keyt && px[v] > -INFINITY) {
                acc += expf(px[v] - mx);
                pick = v;
                if (target < acc) break;
            }
        if (pick >= 0) atomicMin(&s_pick, pick);
    }
    __syncthreads();
    if (s_pick == 0x7fffffff) {          // rounding at the upper end: the last kept id
        int last = -1;
        for (int v = v0; v < v1; ++v)
            if (fkey(px[v]) >= keyt && px[v] > -INFINITY) last = v;
        __syncthreads();
        if (tid == 0) s_pick = -1;
        __syncthreads();
        if (last >= 0) atomicMax(&s_pick, last);
        __syncthreads();
    }
    return s_pick < 0 ? 0 : s_pick;
}

__global__ __launch_bounds__(256) void logits_sample_kernel(const float* __restrict__ logits, long ldl, int V,
        const long long* __restrict__ generated, long ldg, const int* __restrict__ gen_len,
        float temperature, float rep_penalty, int top_k, float top_p, const float* __restrict__ uniforms,
        long long* __restrict__ next, float* __restrict__ filtered) {
    extern __shared__ unsigned cnt[];
    float* px = reinterpret_cast<float*>(cnt + (V + 1) / 2);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(gen_len[b], MAXGEN);
    const long long* gen = generated + (long)b * ldg;
    if (n > 0 && gen[n - 1] == 0) {  // sticky PAD (generate.py:137-138)
        if (tid == 0) next[b] = 0;
        return;
    }
    count_ids(cnt, V, gen, n, tid);
    const int pick = sample_row(logits + (long)b * ldl, V, cnt, px, temperature, rep_penalty, top_k, top_p, uniforms[b], tid,
                                filtered ? filtered + (long)b * ldl : nullptr);
    if (tid == 0) next[b] = pick;
}

}  // namespace

extern "C" int mmtg_logits_process_argmax(const float* logits, long ldl, int V, const long long* generated,
                                          long ldg, const int* gen_len, float temperature, float rep_penalty,
                                          long long* next, int B, void* stream) {
    MMTG_REQUIRE(logits && generated && gen_len && next && B > 0 && V > 0 && ldl >= V, "logits_process_argmax: bad args");
    MMTG_REQUIRE(temperature > 0.f && rep_penalty > 0.f, "logits_process_argmax: temperature / penalty must be > 0");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 4.0 * B * V, 4.0 * B * V);
    MMTG_REQUIRE(V <= 30000, "logits_process_argmax: vocabulary of %d exceeds the 60 KB LDS occurrence table", V);
    hipLaunchKernelGGL(logits_argmax_kernel, dim3(B), dim3(256), (size_t)((V + 1) / 2) * 4, s, logits, ldl, V, generated, ldg, gen_len,
                       temperature, rep_penalty, next);
    MMTG_LAUNCH_CHECK("logits_process_argmax");
    return MMTG_OK;
}

extern "C" int mmtg_logits_process_sample(const float* logits, long ldl, int V, const long long* generated, long ldg,
                                          const int* gen_len, float temperature, float rep_penalty, int top_k, float top_p,
                                          const float* uniforms, long long* next, float* filtered, int B, void* stream) {
    MMTG_REQUIRE(logits && generated && gen_len && uniforms && next && B > 0 && V > 0 && ldl >= V, "logits_process_sample: bad args");
    MMTG_REQUIRE(temperature > 0.f && rep_penalty > 0.f && top_k >= 0 && top_p >= 0.f, "logits_process_sample: bad sampling parameters");
    MMTG_REQUIRE(V <= 24000, "logits_process_sample: vocabulary of %d exceeds the LDS row image (24000)", V);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 12.0 * B * V, 4.0 * B * V);
    const size_t shm = (size_t)((V + 1) / 2) * 4 + (size_t)V * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)logits_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "logits_process_sample: cannot raise dynamic LDS");
        attr_done = true;
    }
    hipLaunchKernelGGL(logits_sample_kernel, dim3(B), dim3(256), shm, s, logits, ldl, V, generated, ldg, gen_len, temperature,
                       rep_penalty, top_k, top_p, uniforms, next, filtered);
    MMTG_LAUNCH_CHECK("logits_process_sample");
    return MMTG_OK;
}

// =====================================================================================
// KV-cached single-token decode step (batched, lock-step positions).
//
// The whole step is device-driven: the current position lives in device memory (`pos_ptr`),
// so one captured hipGraph of the step can be replayed for every position without any
// host-side argument changes.  Sequence buffer seq[B, ldseq] holds prompt ids (positions
// 0..P-1) followed by lyric ids (lyric index j = pos - P, j = 0 is the initial [#START#]).
// Type ids / key mask follow the inference branch of GPT2_Decoder.forward (model.py:290-312).
namespace {

template <typename T>
__global__ __launch_bounds__(256) void decode_embed_kernel(const T* __restrict__ table, const long long* __restrict__ seq,
        long ldseq, const T* __restrict__ c, T* __restrict__ x, const int* __restrict__ pos_ptr,
        const long long* __restrict__ tpw_type, const long long* __restrict__ tpw_mask,
        long long* __restrict__ type_out, int* __restrict__ keep, long ldkeep,
        int P, int S, int E, int two_sents, int V, int sent, int max_sent_num, bf16* __restrict__ xp = nullptr, long planeX = 0) {
    typedef typename Vec16<T>::type V16t;
    constexpr int N = Vec16<T>::N;
    const int b = blockIdx.x, pos = *pos_ptr;
    long long tok = seq[(long)b * ldseq + pos];
    if (threadIdx.x == 0) {
        long long ty;
        int kp;
        if (pos < P) {
            ty = tpw_type[(long)b * P + pos];
            kp = tpw_mask[(long)b * P + pos] != 0;
        } else {
            const int i = pos - P;
            const bool pad = tok == 0;
            const bool edge = ((i + 1) % sent == 0) || ((i + 1) % sent == 1);
            const int sidx = i / sent;
            const int slot = sidx < max_sent_num - 1 ? sidx + 1 : 1;     // [1..max_sent_num-1, 1]
            ty = (edge || pad) ? 0 : slot;
            kp = !pad;
        }
        type_out[b] = ty;
        keep[(long)b * ldkeep + pos] = kp;
    }
    if (tok < 0) tok = 0;
    if (tok >= V) tok = V - 1;
    const int seg = pos < P ? -1 : (pos - P) / two_sents;
    const T* src = table + tok * E;
    const T* cs = (seg >= 0 && seg < S) ? c + ((long)b * S + seg) * E : nullptr;
    T* dst = x + (long)b * E;
    for (int e = threadIdx.x * N; e < E; e += 256 * N) {
        V16t v = *reinterpret_cast<const V16t*>(src + e);
        if (cs) {
            V16t w = *reinterpret_cast<const V16t*>(cs + e);
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = (T)((float)v[k] + (float)w[k]);
        }
        if constexpr (N == 4) {
            if (xp) {        // x3: the conditioned embedding as a (hi | lo) plane pair (the projector product is its only reader)
                bf16x4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) { hi[k] = (bf16)(float)v[k]; lo[k] = (bf16)((float)v[k] - (float)hi[k]); }
                *reinterpret_cast<bf16x4*>(xp + (long)b * E + e) = hi;
                *reinterpret_cast<bf16x4*>(xp + planeX + (long)b * E + e) = lo;
                continue;
            }
        }
        *reinterpret_cast<V16t*>(dst + e) = v;
    }
}

// =====================================================================================
// Round 3: the decode step's batch-sized products with the row-wise glue fused in (5 graph nodes per GPT-2 block instead of 7).
//
// Until now every split-K product of the step was followed by a "finish" launch (sum the slabs, bias, residual, and the LayerNorm
// of the NEXT product's input).  A graph node costs ~4 us whatever it does, so the finish kernels were 24 x 5.2 us
// of a 758 us step.  They are gone:
//   * mode 2 ("reduce"): a split-K product whose K slices are reduced IN the kernel -- every split stores its raw fp32 partial
//     wave tile into slot (tile, split) of a workspace with agent-scope stores, bumps the wave tile's arrival counter, and the
//     wave that arrives last adds the partials in split order (its own from registers), adds bias and residual, rounds and stores
//     the bf16 row segment AND the (sum, sum of squares) of what it stored for each of its rows -- the LayerNorm statistics of
//     the new residual stream, as 32-column partials (the mechanism of mmtg_wgrad_group, csrc/wgrad.hip);
//   * modes 0 / 1 ("LN-fold"): the product that consumes a LayerNorm takes the UN-normalised rows and applies the LayerNorm
//     algebraically: LN(x) W = rstd (x W' - mu c) + b' with W' = gamma (.) W (a bf16 copy, mmtg_ln_fold_weights), c_n = sum_k
//     W'_kn, b' = b + beta W; mu / rstd come from the statistics partials.  Decode has no backward: nothing else needs LN(x).
// Tile = 64 x 64 per 256-thread workgroup (2 x 2 waves of 32 x 32), K-contiguous operands, 4-deep LDS-DMA ring, counted vmcnt
// (the main loop of gemm_dma_kernel<false, false, 64, 64, 2, 2, 4>, gemm.hip).
// (DG_NP = 32 statistics partials per row: gemm_common.h)

// Partial tiles cross XCDs: their stores and loads carry the agent-scope bit (sc1: write through / always miss; cache-policy
// operand 16 of the raw buffer builtins) instead of an L2-wide write-back + invalidate per wave (csrc/wgrad.hip).  Compiler-visible
// loads: every other split's pieces are requested back to back and waited for once.
__device__ __forceinline__ void dg_st4_agent(__amdgpu_buffer_rsrc_t r, int byte_off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ f32x4 dg_ld4_agent(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

struct DgArgs {
    const bf16* A; const bf16* W; void* C;
    long lda, ldw, ldc, ldr;
    int M, N, K, kper, splits, ntiles, tiles_n;
    int bytesA, bytesW;
    const float* bias; const float* colsum; const float* stats_in; int np_in; float eps; float inv_k;
    int act, out_f32;
    const bf16* resid; float* stats_out; int np_out;
    float* ws; unsigned* cnt; int ws_bytes;
    // mode 2, residual = wpe[*pos_ptr] + wte[type_ids[m]] instead of a tensor (the GPT-2 input embedding added in the projector's epilogue)
    const bf16* emb_pos; const bf16* emb_type; const long long* type_ids; const int* pos_ptr;
    // x3 (round 5, mmtg_decode_gemm_x3): A and W are (hi | lo) bf16 plane pairs of fp32 tensors, walked as three passes; C / resid /
    // the embedding tables are fp32; Cp (nullable) receives the result as a plane pair for the next product
    int planeA, planeW;        // bytes from the hi plane to the lo plane
    bf16* Cp; long ldcp, planeC;       // planeC: elements
};

// One 64 x 64 output tile (x one K slice) = work item `bid` of a stand-alone launch (one item per workgroup).
// (The persistent one-launch token step and the chained launches of round 4 -- bit-equal, measured slower -- live in
//  tools/experiments/decode_persistent_and_chained.patch.)
// X3 (round 5): the split-precision form -- operands as (hi | lo) plane pairs, the K slice walked three times (A hi x W hi, A lo x W hi,
// A hi x W lo: only the scalar offsets of a tile change), fp32 residual stream / embeddings, results as fp32 and / or plane pairs.
// X3C: the x3 form with ONE K loop over combined stages -- a stage holds the K tile of all four planes (A hi | A lo | W hi | W lo,
// 32 KB) and every fragment pair feeds three MFMAs -- instead of three passes that re-stage A hi and W hi: two thirds of the bytes
// through the CU's L2 -> LDS port, a third of the barriers; NBUF = 2 (64 KB: still two workgroups per CU).
template <int MODE, int NBUF, int MAXS = 8, bool X3 = false, bool X3C = false>
__device__ __forceinline__ void dg_tile(const DgArgs& p, const int bid, char* smem) {
    static_assert(!X3C || X3, "combined stages belong to the x3 form");
    constexpr int TB = 64, NW = 4, BK = 64, NB = 2;
    constexpr int TA = TB * 128, STAGE = (X3C ? 4 : 2) * TA;
    float* const smu = reinterpret_cast<float*>(smem + NBUF * STAGE);
    float* const srs = smu + TB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    // plain item order: consecutive workgroups = consecutive column tiles of one row tile, dealt round-robin to the 8 XCDs.
    // (An XCD-contiguous order with the row tiles of a weight panel side by side on one XCD -- the training kernels' order --
    //  measured 789 vs 728 us per token step here: the step is latency-bound, and spreading a product's first requests over all
    //  eight L2s matters more than fetching a 96 KB panel once.)
    const int split = bid / p.ntiles, t = bid - split * p.ntiles;
    const int m0 = (t / p.tiles_n) * TB, n0 = (t % p.tiles_n) * TB;
    const int kbeg = split * p.kper;
    const int klen = max(0, min(p.K, kbeg + p.kper) - kbeg);
    const int nk1 = (klen + BK - 1) / BK;                  // (x3: the host keeps K slices whole 64-deep tiles)
    const int nk = (X3 && !X3C) ? 3 * nk1 : nk1, nk_full = X3 ? nk : klen / BK;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, p.bytesW, 0x00020000);
    int sa = (int)(((long)m0 * p.lda + kbeg) * 2), sb = (int)(((long)n0 * p.ldw + kbeg) * 2);
    int va[NB], vb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        va[i] = dma_voff<false, TB>(p.lda, m0, p.M, BK, wave + NW * i, lane);
        vb[i] = dma_voff<false, TB>(p.ldw, n0, p.N, BK, wave + NW * i, lane);
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define DG_ISSUE(tt)                                                                                                   \
    do {                                                                                                               \
        char* st_ = smem + ((tt) % NBUF) * STAGE;                                                                      \
        const bool full_ = (tt) < nk_full, live_ = (tt) < nk;                                                          \
        const int krem_ = klen - (tt) * BK;                                                                            \
        /* x3: pass = tt / nk1 (1 reads A's lo plane, 2 reads W's), K tile tt % nk1 of the slice */                     \
        const int pass_ = (X3 && !X3C) ? ((tt) >= nk1 ? 1 : 0) + ((tt) >= 2 * nk1 ? 1 : 0) : 0;                        \
        const int sa_ = X3 ? sa + ((tt) - pass_ * nk1) * (BK * 2) + (pass_ == 1 ? p.planeA : 0) : sa;                  \
        const int sb_ = X3 ? sb + ((tt) - pass_ * nk1) * (BK * 2) + (pass_ == 2 ? p.planeW : 0) : sb;                  \
        if constexpr (X3C) {        /* all four planes of K tile tt: A hi | A lo | W hi | W lo */                      \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                           \
                const int oa_ = live_ ? va[i] : OOB, ob_ = live_ ? vb[i] : OOB;                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, st_ + (wave + NW * i) * 1024), 16, oa_, live_ ? sa_ : 0, 0, 0); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, st_ + TA + (wave + NW * i) * 1024), 16, oa_, live_ ? sa_ + p.planeA : 0, 0, 0); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, st_ + 2 * TA + (wave + NW * i) * 1024), 16, ob_, live_ ? sb_ : 0, 0, 0); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, st_ + 3 * TA + (wave + NW * i) * 1024), 16, ob_, live_ ? sb_ + p.planeW : 0, 0, 0); \
            }                                                                                                          \
        } else                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                               \
            const int oa_ = !live_ ? OOB : full_ ? va[i] : dma_voff<false, TB>(p.lda, m0, p.M, krem_, wave + NW * i, lane); \
            const int ob_ = !live_ ? OOB : full_ ? vb[i] : dma_voff<false, TB>(p.ldw, n0, p.N, krem_, wave + NW * i, lane); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, st_ + (wave + NW * i) * 1024), 16, oa_, live_ ? sa_ : 0, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, st_ + TA + (wave + NW * i) * 1024), 16, ob_, live_ ? sb_ : 0, 0, 0); \
        }                                                                                                              \
        if constexpr (!X3) {                                                                                           \
            sa += BK * 2;                                                                                              \
            sb += BK * 2;                                                                                              \
        }                                                                                                              \
    } while (0)
#pragma unroll
    for (int t0 = 0; t0 < NBUF - 1; ++t0) DG_ISSUE(t0);
    const int mw0 = m0 + wm * 32, nw0 = n0 + wn * 32;
    // LN-fold: mu / rstd of the tile's 64 rows from the statistics partials, and this lane's column sums / folded biases, while
    // the first K tiles are in flight (published to the other waves by the K loop's barriers; consumed in the epilogue)
    f32x4 c4[2], b4[2];
    if constexpr (MODE != 2) {
        if (tid < TB) {
            const int m = min(m0 + tid, p.M - 1);
            const f32x4* src = reinterpret_cast<const f32x4*>(p.stats_in + (long)m * DG_NP * 2);
            f32x4 sp[DG_NP / 2];
#pragma unroll
            for (int i = 0; i < DG_NP / 2; ++i) sp[i] = i * 2 < p.np_in ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < DG_NP / 2; ++i) { s1 += sp[i][0] + sp[i][2]; s2 += sp[i][1] + sp[i][3]; }
            const float mu = s1 * p.inv_k;
            const float var = fmaxf(s2 * p.inv_k - mu * mu, 0.f);
            smu[tid] = mu;
            srs[tid] = rsqrtf(var + p.eps);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // stored before this wave passes the first K-loop barrier
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = min(nw0 + j * 16 + 4 * g, p.N - 4);
            c4[j] = *reinterpret_cast<const f32x4*>(p.colsum + n);
            b4[j] = MODE == 0 ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(NBUF - 2) * (X3C ? 4 : 2) * NB>();     // my part of tile kt has landed (the younger tiles may be in flight)
        __builtin_amdgcn_s_barrier();              // ... and everyone's; every wave is done reading tile kt - 1
        DG_ISSUE(kt + NBUF - 1);
        const char* tA = smem + (kt % NBUF) * STAGE;
        if constexpr (X3C) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fah[i] = ld_frag_kc<bf16>(tA, wm * 32 + i * 16 + l15, kk, g);
                    fal[i] = ld_frag_kc<bf16>(tA + TA, wm * 32 + i * 16 + l15, kk, g);
                    fbh[i] = ld_frag_kc<bf16>(tA + 2 * TA, wn * 32 + i * 16 + l15, kk, g);
                    fbl[i] = ld_frag_kc<bf16>(tA + 3 * TA, wn * 32 + i * 16 + l15, kk, g);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        mma16(fbh[j], fah[i], acc[i][j]);
                        mma16(fbh[j], fal[i], acc[i][j]);
                        mma16(fbl[j], fah[i], acc[i][j]);
                    }
            }
            continue;
        }
        const char* tB = tA + TA;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = ld_frag_kc<bf16>(tA, wm * 32 + i * 16 + l15, kk, g);
                fb[i] = ld_frag_kc<bf16>(tB, wn * 32 + i * 16 + l15, kk, g);
            }
            // (swapped operands: a lane holds 4 consecutive columns of row l15)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16(fb[j], fa[i], acc[i][j]);
        }
    }
#undef DG_ISSUE
    wait_vmcnt<0>();                               // the zero-fill tail requests
    if constexpr (MODE != 2) {
        // ---- LN-fold epilogue
        if (nk == 0) __syncthreads();              // (no K loop barrier has published smu / srs)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = wm * 32 + i * 16 + l15, m = m0 + rl;
            const float mu = smu[rl], rs = srs[rl];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nw0 + j * 16 + 4 * g;
                if (m >= p.M || n >= p.N) continue;
                f32x4 v;
                if constexpr (MODE == 1) {
                    // slab `split` of the consumer's fp32 input: the mean term rides on slab 0, the bias is the consumer's
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rs * acc[i][j][r] - (split == 0 ? rs * mu * c4[j][r] : 0.f);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + ((long)split * p.M + m) * p.ldc + n) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = rs * (acc[i][j][r] - mu * c4[j][r]) + b4[j][r];
                        if (p.act == MMTG_EPI_GELU) v[r] = gelu_new_t<bf16>(v[r]);
                    }
                    if (p.out_f32) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = v;
                    } else if constexpr (X3) {          // the activation as a plane pair: only the next product reads it
                        bf16x4 hi, lo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { hi[r] = (bf16)v[r]; lo[r] = (bf16)(v[r] - (float)hi[r]); }
                        bf16* dst = p.Cp + (long)m * p.ldcp + n;
                        *reinterpret_cast<bf16x4*>(dst) = hi;
                        *reinterpret_cast<bf16x4*>(dst + p.planeC) = lo;
                    } else {
                        const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n) = o;
                    }
                }
            }
        }

    } else {
        // ---- split-K reduced by the last-arriving wave (csrc/wgrad.hip), + bias + residual + row statistics
        const int S = p.splits;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, p.ws_bytes, 0x00020000);
        const int slot0 = (t * S * (TB * TB) + wave * 1024 + lane * 4) * 4;        // byte offset of my lane in slot (t, 0)
        if (S > 1) {
            const int mine = slot0 + split * (TB * TB * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) dg_st4_agent(rw, mine + (i * 2 + j) * 1024, acc[i][j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every write-through store has been acknowledged
            unsigned* const cnt = p.cnt + (long)t * NW + wave;
            unsigned old = 0;
            if (lane == 0) old = atomicAdd(cnt, 1u);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != (unsigned)(S - 1)) return;
            // (re-armed for the next product: an agent-scope store in the persistent form, where no kernel boundary flushes it)
            if (lane == 0) *cnt = 0u;
        }
        // every split's partial (my own slot included: statically indexed registers, no scratch), all requested back to back:
        // one memory round trip for the whole reduction
        f32x4 prt[MAXS][4];        // (MAXS = most K splits the instantiation serves: 8 stand-alone, 4 in the register-capped persistent kernel)
#pragma unroll
        for (int s_ = 0; s_ < MAXS; ++s_) {
            const int se = s_ < S ? s_ : split;
#pragma unroll
            for (int q = 0; q < 4; ++q) prt[s_][q] = S > 1 ? dg_ld4_agent(rw, slot0 + se * (TB * TB * 4) + q * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mw0 + i * 16 + l15;
            float r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nw0 + j * 16 + 4 * g;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s_ = 0; s_ < MAXS; ++s_) {               // split order; mine from registers
                    const f32x4 v = s_ == split ? acc[i][j] : prt[s_][i * 2 + j];
                    if (s_ < S) sum += v;
                }
                if constexpr (X3) {
                    if (m < p.M && n < p.N) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                        f32x4 xr = {0.f, 0.f, 0.f, 0.f};
                        if (p.type_ids) {
                            const f32x4 e0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.emb_pos) + (long)*p.pos_ptr * p.ldr + n);
                            const f32x4 e1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.emb_type) + p.type_ids[m] * p.ldr + n);
                            xr = e0 + e1;
                        } else if (p.resid) xr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.resid) + (long)m * p.ldr + n);
                        f32x4 f;
                        bf16x4 hi, lo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = sum[r] + b4[r];
                            if (p.act == MMTG_EPI_TANH) f[r] = tanh_t<bf16>(f[r]);
                            f[r] += xr[r];
                            r1 += f[r];
                            r2 += f[r] * f[r];
                            hi[r] = (bf16)f[r];
                            lo[r] = (bf16)(f[r] - (float)hi[r]);
                        }
                        if (p.C) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = f;
                        if (p.Cp) {
                            bf16* dst = p.Cp + (long)m * p.ldcp + n;
                            *reinterpret_cast<bf16x4*>(dst) = hi;
                            *reinterpret_cast<bf16x4*>(dst + p.planeC) = lo;
                        }
                    }
                    continue;
                }
                if (m < p.M && n < p.N) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                    float xr[4];
                    if (p.type_ids) {
                        const bf16x4 e0 = *reinterpret_cast<const bf16x4*>(p.emb_pos + (long)*p.pos_ptr * p.ldr + n);
                        const bf16x4 e1 = *reinterpret_cast<const bf16x4*>(p.emb_type + p.type_ids[m] * p.ldr + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) xr[r] = (float)e0[r] + (float)e1[r];
                    } else {
                        bf16x4 x4;
                        x4 = *reinterpret_cast<const bf16x4*>(p.resid + (long)m * p.ldr + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) xr[r] = (float)x4[r];
                    }
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (bf16)(sum[r] + b4[r] + xr[r]);
                        const float f = (float)o[r];
                        r1 += f;
                        r2 += f * f;
                    }
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n) = o;
                }
            }
            // the row's 32 columns of this wave tile live in the four lane groups g: fold them
            r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
            r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
            if (g == 0 && m < p.M && nw0 < p.N && (!X3 || p.stats_out)) {
                float* dst = p.stats_out + ((long)m * DG_NP + (nw0 >> 5)) * 2;
                dst[0] = r1;
                dst[1] = r2;
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void decode_gemm_kernel(DgArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 4 stages | row statistics
    dg_tile<MODE, 4>(p, blockIdx.x, smem);
}
template <int MODE, bool COMB = false>
__global__ __launch_bounds__(256, 2) void decode_gemm_x3_kernel(DgArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 4 stages of 16 KB (three passes) / 2 of 32 KB (combined) | row statistics
    dg_tile<MODE, (COMB ? 2 : 4), 8, true, COMB>(p, blockIdx.x, smem);
}

// x3 weight preparation, one wave per output row n of a K-contiguous weight given as a plane pair W = hi + lo ([N, K]):
//   W'[n, k] = gamma[k] W[n, k] re-split into a plane pair, c[n] = sum_k (W'_hi + W'_lo)[n, k] (of what the product multiplies),
//   bf[n] = bias[n] + sum_k beta[k] W[n, k]
__global__ __launch_bounds__(256) void ln_fold_x3_kernel(const bf16* __restrict__ W, long ldw, long planeW, const float* __restrict__ gamma,
        const float* __restrict__ beta, const float* __restrict__ bias, bf16* __restrict__ Wf, long ldf, long planeF, float* __restrict__ c,
        float* __restrict__ bf, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float sc = 0.f, sb = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(W + (long)n * ldw + k);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(W + planeW + (long)n * ldw + k);
        bf16x8 oh, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = (float)wh[e] + (float)wl[e];
            const float f = gamma[k + e] * w;
            oh[e] = (bf16)f;
            ol[e] = (bf16)(f - (float)oh[e]);
            sc += (float)oh[e] + (float)ol[e];
            sb += beta[k + e] * w;
        }
        *reinterpret_cast<bf16x8*>(Wf + (long)n * ldf + k) = oh;
        *reinterpret_cast<bf16x8*>(Wf + planeF + (long)n * ldf + k) = ol;
    }
    sc = wave_sum(sc);
    sb = wave_sum(sb);
    if (lane == 0) { c[n] = sc; bf[n] = (bias ? bias[n] : 0.f) + sb; }
}

// W'[n, k] = gamma[k] W[n, k] (bf16), c[n] = sum_k W'[n, k] (of the ROUNDED values: what the product will multiply),
// bf[n] = bias[n] + sum_k beta[k] W[n, k]: one wave per output row
__global__ __launch_bounds__(256) void ln_fold_kernel(const bf16* __restrict__ W, long ldw, const float* __restrict__ gamma,
        const float* __restrict__ beta, const float* __restrict__ bias, bf16* __restrict__ Wf, float* __restrict__ c,
        float* __restrict__ bf, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float sc = 0.f, sb = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(W + (long)n * ldw + k);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (bf16)(gamma[k + e] * (float)w[e]);
            sc += (float)o[e];
            sb += beta[k + e] * (float)w[e];
        }
        *reinterpret_cast<bf16x8*>(Wf + (long)n * ldw + k) = o;
    }
    sc = wave_sum(sc);
    sb = wave_sum(sb);
    if (lane == 0) { c[n] = sc; bf[n] = (bias ? bias[n] : 0.f) + sb; }
}

template <typename T>
__global__ __launch_bounds__(256) void decode_embed_add_kernel(const T* __restrict__ g, const T* __restrict__ wpe,
        const T* __restrict__ wte, const long long* __restrict__ type_ids, const int* __restrict__ pos_ptr,
        T* __restrict__ h, int D, float* __restrict__ stats) {
    __shared__ float red[8];
    const int b = blockIdx.x, pos = *pos_ptr;
    const long long ty = type_ids[b];
    float s1 = 0.f, s2 = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const T o = (T)((float)g[(long)b * D + d] + (float)wpe[(long)pos * D + d] + (float)wte[ty * D + d]);
        h[(long)b * D + d] = o;
        s1 += (float)o;
        s2 += (float)o * (float)o;
    }
    if (stats) {        // LayerNorm statistics of the row as stored (fused decode path): partial 0 carries all of it
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s1; red[4 + (threadIdx.x >> 6)] = s2; }
        __syncthreads();
        float* dst = stats + (long)b * DG_NP * 2;
        if (threadIdx.x == 0) { dst[0] = red[0] + red[1] + red[2] + red[3]; dst[1] = red[4] + red[5] + red[6] + red[7]; }
        else if (threadIdx.x < DG_NP * 2 && threadIdx.x >= 2) dst[threadIdx.x] = 0.f;
    }
}

// One wave per (b, head): append this token's K/V to the cache, then attend over keys 0..pos.
// Cache layout [B, nH, Tmax, 64].  Both passes over the cache use 16-byte vectors: a key's 64 channels are spread over
// OCT = 8 (bf16) / 16 (f32) adjacent lanes, so one wave-instruction reads 64 / OCT whole cache rows -- 1 KB, fully
// coalesced.  Scores: partial dot products folded across the OCT lanes; values: each lane accumulates its channel octet
// over its keys, the key groups are folded at the end.  Fixed summation order (ascending key per lane).
// Round 3: the K AND V rows of the first UN0 * KPI keys are requested at the very top, before the token's own q / k / v are
// even assembled from the c_attn slabs -- they depend on nothing this kernel computes -- so a step with a short prefix costs one
// memory round trip instead of three dependent ones (slabs -> scores -> values).  Branch-free: rows at or past `pos` are
// requested from the last valid row and replaced afterwards (key == pos: the token's own row, held in LDS exactly as stored).
// The KV cache rows are read once per token step and never again by this step: with `nt` (non-temporal) they need not displace the
// step's 193 MB of weights from the 256 MB Infinity Cache, which every step re-reads (A/B: MMTG_DECODE_KV_NT, profiles/r04_*).
// cross-lane folds of the score loop by DPP (full-rate VALU, no LDS crossbar: the __shfl_xor forms are a ds_bpermute + address
// arithmetic per step and sat on the per-chunk dependent chain).  fold_sum_lanes<OCT>: the sum over the OCT adjacent lanes of a key
// (quad swaps, then the mirror inside 8 / 16 lanes: every lane ends with the group's sum).  wave_max_of_groups<OCT>: the maximum
// over the whole wave of a value that is already uniform inside each group of OCT lanes; returned wave-uniform (readlane 63).
template <int CTRL, int ROWS = 0xF> __device__ __forceinline__ float dpp_keep(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), CTRL, ROWS, 0xF, false));
}
template <int OCT> __device__ __forceinline__ float fold_sum_lanes(float a) {
    a += dpp_keep<0xB1>(a);                    // quad_perm [1,0,3,2]
    a += dpp_keep<0x4E>(a);                    // quad_perm [2,3,0,1]
    a += dpp_keep<0x141>(a);                   // row_half_mirror
    if constexpr (OCT == 16) a += dpp_keep<0x140>(a);      // row_mirror
    return a;
}
template <int OCT> __device__ __forceinline__ float wave_max_of_groups(float x) {
    if constexpr (OCT == 8) x = fmaxf(x, dpp_keep<0x140>(x));          // the row's two groups
    x = fmaxf(x, dpp_keep<0x142, 0xA>(x));     // row_bcast15 into rows 1, 3
    x = fmaxf(x, dpp_keep<0x143, 0xC>(x));     // row_bcast31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

template <typename V, bool NT> __device__ __forceinline__ V ld_kv(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// bytes of LDS one (b, head) item needs: scores, key flags, q (f32), this token's k and v
template <typename T> constexpr int da_lds_bytes() { return 1024 * 4 + 1024 * 4 + 64 * 4 + 2 * 64 * (int)sizeof(T); }

// one 64-thread workgroup per (b, head)
template <typename T, bool NT = false>
__device__ __forceinline__ void da_body(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
        const int* __restrict__ keep, long ldkeep, const int pos, T* __restrict__ out, const int B,
        int nH, int Tmax, const float* __restrict__ part, int splits, long slab, const float* __restrict__ bias,
        const int h, const int b, const int lane, char* lds, bf16* __restrict__ oplanes = nullptr, long oplane = 0) {
    typedef typename Vec16<T>::type V;
    constexpr int EPL = Vec16<T>::N, OCT = 64 / EPL, KPI = 64 / OCT;      // elements per lane, lanes per key, keys per instruction
    // (the SAME chunking in both forms: the first chunk's scores and the later chunks' come out of differently contracted
    //  loops, so moving the chunk boundary moves last bits -- the persistent step stays bit-equal to the per-launch one)
    constexpr int UN = 4, CH = UN * KPI;         // a chunk: UN keys per lane group = 32 keys (bf16) / 16 (f32)
    int* const skeep = reinterpret_cast<int*>(lds + 4096);
    float* const sq = reinterpret_cast<float*>(lds + 8192);
    T* const sk = reinterpret_cast<T*>(lds + 8192 + 256);
    T* const sv = sk + 64;
#define DA_SYNC() __syncthreads()
    const int D = nH * 64;
    const T* row = qkv + (long)b * 3 * D + h * 64;
    T* kbase = kc + (((long)b * nH + h) * Tmax) * 64;
    T* vbase = vc + (((long)b * nH + h) * Tmax) * 64;
    const int nkeys = pos + 1;
    const int oc = lane % OCT, kg = lane / OCT;
    const int last = pos > 0 ? pos - 1 : 0;
    // chunks 0 and 1 (K and V rows) are requested at the very top: they depend on nothing this kernel computes.  Branch-free:
    // rows at or past `pos` are requested from the last valid row and replaced (key == pos: the token's own row, from LDS) or masked
    V kA[UN], vA[UN], kB[UN], vB[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int key = u * KPI + kg, kr = key < pos ? key : last;
        kA[u] = ld_kv<V, NT>(reinterpret_cast<const V*>(kbase + (long)kr * 64 + oc * EPL));
        vA[u] = ld_kv<V, NT>(reinterpret_cast<const V*>(vbase + (long)kr * 64 + oc * EPL));
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int key = CH + u * KPI + kg, kr = key < pos ? key : last;
        kB[u] = ld_kv<V, NT>(reinterpret_cast<const V*>(kbase + (long)kr * 64 + oc * EPL));
        vB[u] = ld_kv<V, NT>(reinterpret_cast<const V*>(vbase + (long)kr * 64 + oc * EPL));
    }
    // the key-padding flags of the whole prefix in one coalesced pass, into LDS (round 3: the score loop used to read keep[]
    // from global memory key group by key group -- up to 16 DEPENDENT 4-byte loads, each a full memory round trip)
    // (the first 256 flags as four independent loads, stored after the slab sums; longer prefixes: a plain loop)
    int kp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kp[i] = keep[(long)b * ldkeep + min(lane + 64 * i, pos)];
    asm volatile("" ::: "memory");      // compiler barrier: the requests above are ISSUED here (nothing waits for them yet)
    if (part) {
        // the c_attn product arrives as split-K slabs (MMTG_EPI_SPLIT): sum them in order, add the bias
        // and round to the storage type exactly as mmtg_splitk_finish would (saves that launch); up to four slabs are
        // requested together (one round trip), more in a plain loop
        const float* pr = part + (long)b * 3 * D + h * 64 + lane;
        float q = 0.f, k = 0.f, v = 0.f;
        const float bq = bias[h * 64 + lane], bk = bias[D + h * 64 + lane], bv = bias[2 * D + h * 64 + lane];
        float pq[4], pk[4], pv[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const long o = (long)(s_ < splits ? s_ : 0) * slab;
            pq[s_] = pr[o]; pk[s_] = pr[o + D]; pv[s_] = pr[o + 2 * D];
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
            if (s_ < splits) { q += pq[s_]; k += pk[s_]; v += pv[s_]; }
        for (int s_ = 4; s_ < splits; ++s_) { q += pr[s_ * slab]; k += pr[s_ * slab + D]; v += pr[s_ * slab + 2 * D]; }
        q += bq; k += bk; v += bv;
        sq[lane] = (float)(T)q * 0.125f;
        sk[lane] = (T)k;
        sv[lane] = (T)v;
    } else {
        sq[lane] = (float)row[lane] * 0.125f;
        sk[lane] = row[D + lane];
        sv[lane] = row[2 * D + lane];
    }
    kbase[(long)pos * 64 + lane] = sk[lane];
    vbase[(long)pos * 64 + lane] = sv[lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) skeep[lane + 64 * i] = kp[i];
    for (int key = lane + 256; key < nkeys; key += 64) skeep[key] = keep[(long)b * ldkeep + key];
    DA_SYNC();
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[e] = sq[oc * EPL + e];
    const V knew = *reinterpret_cast<const V*>(sk + oc * EPL), vnew = *reinterpret_cast<const V*>(sv + oc * EPL);
    // ---- ONE pass over the cache (round 4): scores, running maximum / sum and the value accumulation per chunk of CH keys, all in
    // registers -- a lane group's eight (bf16) lanes all hold the score of the group's key after the fold, and the same lanes own the
    // channel octets of that key's value row, so no score ever goes through LDS.  Two chunks are always in flight (buffer c & 1 is
    // refilled with chunk c + 2 as soon as chunk c is consumed): the two-pass form walked the cache as K chunks, then softmax, then V
    // chunks -- up to five DEPENDENT memory round trips per layer at a full prefix against two here.
    float m_run = -INFINITY, sum = 0.f;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#define DA_STEP(KB_, VB_, K0_)                                                                                       \
    do {                                                                                                             \
        float a_[UN];                                                                                                \
        float cm_ = -INFINITY;                                                                                       \
        _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                                             \
            const int key = (K0_) + u * KPI + kg;                                                                    \
            const bool isnew = key == pos;                                                                           \
            float a = 0.f;                                                                                           \
            _Pragma("unroll") for (int e = 0; e < EPL; ++e) {                                                        \
                a += qv[e] * (float)(isnew ? knew[e] : KB_[u][e]);                                                   \
                VB_[u][e] = isnew ? vnew[e] : VB_[u][e];                                                             \
            }                                                                                                        \
            a = fold_sum_lanes<OCT>(a);                                                                              \
            a_[u] = (key < nkeys && skeep[key < nkeys ? key : pos]) ? a : -INFINITY;                                 \
            cm_ = fmaxf(cm_, a_[u]);                                                                                 \
        }                                                                                                            \
        cm_ = wave_max_of_groups<OCT>(cm_);                                                                          \
        const float m_new = fmaxf(m_run, cm_);                                                                       \
        const float muse = m_new == -INFINITY ? 0.f : m_new;                                                         \
        const float alpha = expf(m_run - muse);                                                                      \
        sum *= alpha;                                                                                                \
        _Pragma("unroll") for (int e = 0; e < EPL; ++e) acc[e] *= alpha;                                             \
        _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                                             \
            const float p = a_[u] == -INFINITY ? 0.f : expf(a_[u] - muse);                                           \
            sum += p;                                                                                                \
            _Pragma("unroll") for (int e = 0; e < EPL; ++e) acc[e] += p * (float)VB_[u][e];                          \
        }                                                                                                            \
        m_run = m_new;                                                                                               \
    } while (0)
#define DA_ISSUE(KB_, VB_, K0_)                                                                                      \
    do {                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                                             \
            const int key = (K0_) + u * KPI + kg, kr = key < pos ? key : last;                                       \
            KB_[u] = ld_kv<V, NT>(reinterpret_cast<const V*>(kbase + (long)kr * 64 + oc * EPL));                     \
            VB_[u] = ld_kv<V, NT>(reinterpret_cast<const V*>(vbase + (long)kr * 64 + oc * EPL));                     \
        }                                                                                                            \
    } while (0)
#pragma unroll 1
    for (int k0 = 0; k0 < nkeys; k0 += 2 * CH) {
        DA_STEP(kA, vA, k0);
        if (k0 + 2 * CH < nkeys) DA_ISSUE(kA, vA, k0 + 2 * CH);
        if (k0 + CH < nkeys) {
            DA_STEP(kB, vB, k0 + CH);
            if (k0 + 3 * CH < nkeys) DA_ISSUE(kB, vB, k0 + 3 * CH);
        }
    }
#undef DA_STEP
#undef DA_ISSUE
    // fold over the key groups: every lane group holds the partial sums of ITS keys (the same value in its OCT lanes)
#pragma unroll
    for (int o = OCT; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
        for (int o = OCT; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (kg == 0) {
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        V o;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = (T)(acc[e] * inv);
        if constexpr (EPL == 4) {
            if (oplanes) {        // x3: the context row as a (hi | lo) plane pair (attn.c_proj is its only reader)
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hi[e] = (bf16)(float)o[e]; lo[e] = (bf16)((float)o[e] - (float)hi[e]); }
                bf16* dst = oplanes + (long)b * D + h * 64 + oc * 4;
                *reinterpret_cast<bf16x4*>(dst) = hi;
                *reinterpret_cast<bf16x4*>(dst + oplane) = lo;
                return;
            }
        }
        *reinterpret_cast<V*>(out + (long)b * D + h * 64 + oc * EPL) = o;
    }
#undef DA_SYNC
}

template <typename T, bool NT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void decode_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
        const int* __restrict__ keep, long ldkeep, const int* __restrict__ pos_ptr, T* __restrict__ out,
        int nH, int Tmax, const float* __restrict__ part, int splits, long slab, const float* __restrict__ bias,
        bf16* __restrict__ oplanes = nullptr, long oplane = 0) {
    __shared__ __attribute__((aligned(16))) char lds[da_lds_bytes<T>()];
    da_body<T, NT>(qkv, kc, vc, keep, ldkeep, *pos_ptr, out, (int)gridDim.y, nH, Tmax, part, splits, slab, bias,
                      (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, lds, oplanes, oplane);
}

// logits processor + arg-max + forced-token cadence + append (generate.py:117-142), device-driven.
__global__ __launch_bounds__(256) void decode_select_kernel(const float* __restrict__ logits, long ldl, int V,
        long long* __restrict__ seq, long ldseq, const int* __restrict__ pos_ptr, int P, int sent,
        float temperature, float rep_penalty, int have_logits, int top_k, float top_p,
        const float* __restrict__ uniforms, long ldu, int* __restrict__ pos_next) {
    extern __shared__ unsigned cnt[];
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int b = blockIdx.x, tid = threadIdx.x, pos = *pos_ptr;
    // the step's last kernel also publishes the next position -- into the OTHER slot of the decoder's position pair (every
    // kernel of this step reads *pos_ptr; nobody reads *pos_next before the next step), which replaces a one-thread launch
    if (pos_next && b == 0 && tid == 0) *pos_next = pos + 1;
    const int j = pos + 1 - P;                 // lyric index of the token to append
    if (j < 1) return;                         // still inside the prompt / the initial [#START#]
    long long* gen = seq + (long)b * ldseq + P;
    if (j > 1 && (j + 1) % sent == 0) { if (tid == 0) gen[j] = 2; return; }     // [#EOS#]
    if (j > 1 && (j + 1) % sent == 1) { if (tid == 0) gen[j] = 1; return; }     // [#START#]
    if (!have_logits) return;
    const int n = min(j, MAXGEN);
    if (gen[n - 1] == 0) { if (tid == 0) gen[j] = 0; return; }                   // sticky PAD
    count_ids(cnt, V, gen, n, tid);
    if (uniforms) {            // stochastic: uniforms[pos, b] drives the draw (top_k == 1 and top_p == 0 is the greedy path)
        float* px = reinterpret_cast<float*>(cnt + (V + 1) / 2);
        const int pick = sample_row(logits + (long)b * ldl, V, cnt, px, temperature, rep_penalty, top_k, top_p,
                                    uniforms[(long)pos * ldu + b], tid, nullptr);
        if (tid == 0) gen[j] = pick;
        return;
    }
    float best;
    int besti;
    scan_row(logits + (long)b * ldl, V, cnt, temperature, rep_penalty, tid, best, besti);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((tid & 63) == 0) { sval[tid >> 6] = best; sidx[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < besti)) { best = sval[w]; besti = sidx[w]; }
        gen[j] = besti == 0x7fffffff ? 0 : besti;
    }
}

__global__ void decode_advance_kernel(int* pos_ptr) { *pos_ptr += 1; }

}  // namespace

extern "C" int mmtg_decode_embed(int dtype, const void* table, const long long* seq, long ldseq, const void* c, void* x,
                                 const int* pos_ptr, const long long* tpw_type, const long long* tpw_mask,
                                 long long* type_out, int* keep, long ldkeep, int B, int P, int S, int E, int two_sents,
                                 int V, int sent, int max_sent_num, void* stream) {
    MMTG_REQUIRE(table && seq && c && x && pos_ptr && tpw_type && tpw_mask && type_out && keep, "decode_embed: null pointer");
    MMTG_REQUIRE(B > 0 && E % 8 == 0 && sent > 1 && max_sent_num > 1, "decode_embed: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, (double)B * E, (dtype == MMTG_F32 ? 12.0 : 6.0) * B * E);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(decode_embed_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)table, seq, ldseq, (const float*)c, (float*)x, pos_ptr, tpw_type, tpw_mask, type_out, keep, ldkeep, P, S, E, two_sents, V, sent, max_sent_num);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(decode_embed_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)table, seq, ldseq, (const bf16*)c, (bf16*)x, pos_ptr, tpw_type, tpw_mask, type_out, keep, ldkeep, P, S, E, two_sents, V, sent, max_sent_num);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "decode_embed: bad dtype");
    MMTG_LAUNCH_CHECK("decode_embed");
    return MMTG_OK;
}

extern "C" int mmtg_decode_embed_add(int dtype, const void* g, const void* wpe, const void* wte, const long long* type_ids,
                                     const int* pos_ptr, void* h, int B, int D, float* stats, void* stream) {
    MMTG_REQUIRE(g && wpe && wte && type_ids && pos_ptr && h && B > 0 && D > 0, "decode_embed_add: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 2.0 * B * D, (dtype == MMTG_F32 ? 16.0 : 8.0) * B * D);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(decode_embed_add_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)g, (const float*)wpe, (const float*)wte, type_ids, pos_ptr, (float*)h, D, stats);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(decode_embed_add_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)g, (const bf16*)wpe, (const bf16*)wte, type_ids, pos_ptr, (bf16*)h, D, stats);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "decode_embed_add: bad dtype");
    MMTG_LAUNCH_CHECK("decode_embed_add");
    return MMTG_OK;
}

extern "C" int mmtg_decode_attn(int dtype, const void* qkv, void* kcache, void* vcache, const int* keep, long ldkeep,
                                const int* pos_ptr, void* out, int B, int nH, int dh, int Tmax, void* stream) {
    MMTG_REQUIRE(qkv && kcache && vcache && keep && pos_ptr && out, "decode_attn: null pointer");
    MMTG_REQUIRE(dh == 64 && B > 0 && nH > 0 && Tmax > 0 && Tmax <= 1024, "decode_attn: head dim 64, Tmax <= 1024");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_DECODE, s, 4.0 * B * nH * (double)Tmax * dh, esz * 2.0 * B * nH * (double)Tmax * dh);
    dim3 grid(nH, B), block(64);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(decode_attn_kernel<float>, grid, block, 0, s, (const float*)qkv, (float*)kcache, (float*)vcache, keep, ldkeep, pos_ptr, (float*)out, nH, Tmax, nullptr, 0, 0, nullptr);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(decode_attn_kernel<bf16>, grid, block, 0, s, (const bf16*)qkv, (bf16*)kcache, (bf16*)vcache, keep, ldkeep, pos_ptr, (bf16*)out, nH, Tmax, nullptr, 0, 0, nullptr);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "decode_attn: bad dtype");
    MMTG_LAUNCH_CHECK("decode_attn");
    return MMTG_OK;
}

extern "C" int mmtg_decode_attn_split(int dtype, const float* part, int splits, const float* bias, void* kcache, void* vcache,
                                      const int* keep, long ldkeep, const int* pos_ptr, void* out, int B, int nH, int dh, int Tmax,
                                      void* stream) {
    MMTG_REQUIRE(part && bias && splits > 0 && kcache && vcache && keep && pos_ptr && out, "decode_attn_split: null pointer");
    MMTG_REQUIRE(dh == 64 && B > 0 && nH > 0 && Tmax > 0 && Tmax <= 1024, "decode_attn_split: head dim 64, Tmax <= 1024");
    hipStream_t s = (hipStream_t)stream;
    const double esz = dtype == MMTG_F32 ? 4 : 2;
    ProfScope prof(MMTG_PROF_DECODE, s, 4.0 * B * nH * (double)Tmax * dh, esz * 2.0 * B * nH * (double)Tmax * dh);
    dim3 grid(nH, B), block(64);
    const long slab = (long)B * 3 * nH * 64;
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(decode_attn_kernel<float>, grid, block, 0, s, (const float*)nullptr, (float*)kcache, (float*)vcache, keep, ldkeep, pos_ptr, (float*)out, nH, Tmax, part, splits, slab, bias);
    else if (dtype == MMTG_BF16) {
        static const bool kv_nt = !getenv("MMTG_DECODE_KV_NT") || atoi(getenv("MMTG_DECODE_KV_NT")) != 0;      // default on (700 vs 721 us per token step; =0 for the A/B)
        if (kv_nt) hipLaunchKernelGGL((decode_attn_kernel<bf16, true>), grid, block, 0, s, (const bf16*)nullptr, (bf16*)kcache, (bf16*)vcache, keep, ldkeep, pos_ptr, (bf16*)out, nH, Tmax, part, splits, slab, bias);
        else hipLaunchKernelGGL(decode_attn_kernel<bf16>, grid, block, 0, s, (const bf16*)nullptr, (bf16*)kcache, (bf16*)vcache, keep, ldkeep, pos_ptr, (bf16*)out, nH, Tmax, part, splits, slab, bias);
    } else MMTG_FAIL(MMTG_ERR_BAD_ARG, "decode_attn_split: bad dtype");
    MMTG_LAUNCH_CHECK("decode_attn_split");
    return MMTG_OK;
}

static int decode_select_launch(const float* logits, long ldl, int V, long long* seq, long ldseq, const int* pos_ptr,
                                int P, int sent, float temperature, float rep_penalty, int top_k, float top_p,
                                const float* uniforms, long ldu, int B, int* pos_next, void* stream);

extern "C" int mmtg_decode_select(const float* logits, long ldl, int V, long long* seq, long ldseq, const int* pos_ptr,
                                  int P, int sent, float temperature, float rep_penalty, int B, int* pos_next, void* stream) {
    return decode_select_launch(logits, ldl, V, seq, ldseq, pos_ptr, P, sent, temperature, rep_penalty, 1, 0.f, nullptr, 0, B, pos_next, stream);
}

extern "C" int mmtg_decode_sample(const float* logits, long ldl, int V, long long* seq, long ldseq, const int* pos_ptr,
                                  int P, int sent, float temperature, float rep_penalty, int top_k, float top_p,
                                  const float* uniforms, long ldu, int B, int* pos_next, void* stream) {
    MMTG_REQUIRE(uniforms && ldu >= B && top_k >= 0 && top_p >= 0.f, "decode_sample: uniforms [positions, ldu >= B], top_k >= 0, top_p >= 0");
    return decode_select_launch(logits, ldl, V, seq, ldseq, pos_ptr, P, sent, temperature, rep_penalty, top_k, top_p, uniforms, ldu, B, pos_next, stream);
}

static int decode_select_launch(const float* logits, long ldl, int V, long long* seq, long ldseq, const int* pos_ptr,
                                int P, int sent, float temperature, float rep_penalty, int top_k, float top_p,
                                const float* uniforms, long ldu, int B, int* pos_next, void* stream) {
    MMTG_REQUIRE(seq && pos_ptr && B > 0 && sent > 1, "decode_select: bad args");
    MMTG_REQUIRE(pos_next != pos_ptr, "decode_select: pos_next must not alias pos_ptr (blocks read the position while block 0 writes the next)");
    MMTG_REQUIRE(!logits || (V > 0 && ldl >= V && temperature > 0.f && rep_penalty > 0.f), "decode_select: bad logits args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 4.0 * B * V, 4.0 * B * V);
    MMTG_REQUIRE(!logits || V <= (uniforms ? 24000 : 30000), "decode_select: vocabulary of %d exceeds the LDS tables", V);
    size_t shm = logits ? (size_t)((V + 1) / 2) * 4 : 0;
    if (logits && uniforms) shm += (size_t)V * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)decode_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "decode_select: cannot raise dynamic LDS");
        attr_done = true;
    }
    hipLaunchKernelGGL(decode_select_kernel, dim3(B), dim3(256), shm, s, logits, ldl, V, seq, ldseq,
                       pos_ptr, P, sent, temperature, rep_penalty, logits != nullptr, top_k, top_p, logits ? uniforms : nullptr, ldu, pos_next);
    MMTG_LAUNCH_CHECK("decode_select");
    return MMTG_OK;
}

extern "C" int mmtg_decode_advance(int* pos_ptr, void* stream) {
    MMTG_REQUIRE(pos_ptr, "decode_advance: null pointer");
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos_ptr);
    MMTG_LAUNCH_CHECK("decode_advance");
    return MMTG_OK;
}

extern "C" int mmtg_ln_fold_weights(const void* W, long ldw, const float* gamma, const float* beta, const float* bias, void* Wf,
                                    float* colsum, float* bias_f, int N, int K, void* stream) {
    MMTG_REQUIRE(W && gamma && beta && Wf && colsum && bias_f && N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K,
                 "ln_fold_weights: bad arguments (K, ldw multiples of 8)");
    MMTG_REQUIRE(MMTG_ALIGNED16(W) && MMTG_ALIGNED16(Wf), "ln_fold_weights: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 3.0 * N * (double)K, 4.0 * N * (double)K);
    hipLaunchKernelGGL(ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, (const bf16*)W, ldw, gamma, beta, bias, (bf16*)Wf, colsum, bias_f, N, K);
    MMTG_LAUNCH_CHECK("ln_fold_weights");
    return MMTG_OK;
}

static int dg_fill(DgArgs& a, int mode, int M, int N, int K, const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                   const float* bias, const float* colsum, const float* stats_in, int np_in, float eps, int act,
                   int out_f32, const void* resid, long ldr, float* stats_out, int splits, float* ws, long ws_floats,
                   unsigned* counters, long n_counters, const void* emb_pos, const void* emb_type,
                   const long long* type_ids, const int* pos_ptr) {
    MMTG_REQUIRE(mode >= 0 && mode <= 2, "decode_gemm: mode 0 (LN-fold), 1 (LN-fold slabs) or 2 (in-kernel split-K reduce)");
    MMTG_REQUIRE(M > 0 && N > 0 && K > 0 && A && W && C, "decode_gemm: bad arguments");
    MMTG_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && N % 4 == 0 && ldc % 4 == 0, "decode_gemm: K, lda, ldw multiples of 8; N, ldc of 4");
    MMTG_REQUIRE(MMTG_ALIGNED16(A) && MMTG_ALIGNED16(W) && MMTG_ALIGNED16(C), "decode_gemm: 16-byte alignment");
    if (splits < 1) splits = 1;
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)A; a.W = (const bf16*)W; a.C = C; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.M = M; a.N = N; a.K = K;
    a.kper = cdiv(cdiv(K, splits), 64) * 64;
    a.splits = cdiv(K, a.kper);
    a.tiles_n = cdiv(N, 64);
    a.ntiles = cdiv(M, 64) * a.tiles_n;
    const long bytesA = ((long)(M - 1) * lda + K) * 2, bytesW = ((long)(N - 1) * ldw + K) * 2;
    MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesW < 0x7FFFFF00L, "decode_gemm: operands must stay below 2 GiB");
    a.bytesA = (int)bytesA; a.bytesW = (int)bytesW;
    a.bias = bias; a.colsum = colsum; a.stats_in = stats_in; a.np_in = np_in; a.eps = eps; a.inv_k = 1.0f / (float)K;
    a.act = act; a.out_f32 = out_f32; a.resid = (const bf16*)resid; a.stats_out = stats_out; a.ws = ws; a.cnt = counters;
    a.ws_bytes = (int)(ws_floats < (1L << 29) ? ws_floats * 4 : 0x7FFFFFF0L);
    if (mode != 2) {
        MMTG_REQUIRE(colsum && stats_in && np_in >= 1 && np_in <= DG_NP && MMTG_ALIGNED16(stats_in) && MMTG_ALIGNED16(colsum),
                     "decode_gemm: LN-fold needs the weight column sums and the row statistics ([M][%d][2] floats, np_in partials)", DG_NP);
        MMTG_REQUIRE(mode == 1 || (bias && MMTG_ALIGNED16(bias) && a.splits == 1), "decode_gemm: mode 0 is a single K slice with the folded bias");
        MMTG_REQUIRE(mode == 0 || out_f32 == 1, "decode_gemm: mode 1 writes fp32 slabs [splits][M][ldc]");
        MMTG_REQUIRE(act == MMTG_EPI_NONE || act == MMTG_EPI_GELU, "decode_gemm: activation NONE or GELU");
    } else {
        MMTG_REQUIRE(!type_ids || (emb_pos && emb_type && pos_ptr && !resid), "decode_gemm: the embedding residual takes wpe, wte, type ids and the position (and no resid)");
        a.emb_pos = (const bf16*)emb_pos; a.emb_type = (const bf16*)emb_type; a.type_ids = type_ids; a.pos_ptr = pos_ptr;
        MMTG_REQUIRE(bias && (resid || type_ids) && stats_out && N % 32 == 0 && N / 32 <= DG_NP && ldr % 4 == 0 && MMTG_ALIGNED16(bias),
                     "decode_gemm: the reduce mode needs bias, residual and the statistics output; N a multiple of 32, at most %d", 32 * DG_NP);
        MMTG_REQUIRE(a.splits <= 8, "decode_gemm: at most 8 K splits in the reduce mode");
        MMTG_REQUIRE(a.splits == 1 || (ws && counters && ws_floats >= (long)a.ntiles * a.splits * 4096 && n_counters >= (long)a.ntiles * 4),
                     "decode_gemm: split products need %ld workspace floats and %ld zeroed counters", (long)a.ntiles * a.splits * 4096, (long)a.ntiles * 4);
    }
    return MMTG_OK;
}

extern "C" int mmtg_decode_gemm(int mode, int M, int N, int K, const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                                const float* bias, const float* colsum, const float* stats_in, int np_in, float eps, int act,
                                int out_f32, const void* resid, long ldr, float* stats_out, int splits, float* ws, long ws_floats,
                                unsigned* counters, long n_counters, const void* emb_pos, const void* emb_type,
                                const long long* type_ids, const int* pos_ptr, void* stream) {
    DgArgs a;
    int rc_ = dg_fill(a, mode, M, N, K, A, lda, W, ldw, C, ldc, bias, colsum, stats_in, np_in, eps, act, out_f32, resid, ldr, stats_out, splits,
                      ws, ws_floats, counters, n_counters, emb_pos, emb_type, type_ids, pos_ptr);
    if (rc_) return rc_;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K, 2.0 * ((double)M * K + (double)N * K) + (out_f32 ? 4.0 : 2.0) * (double)M * N);
    const size_t shm = 4 * 2 * 64 * 128 + 2 * 64 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)decode_gemm_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "decode_gemm: cannot raise dynamic LDS");
        attr_done = true;
    }
    const dim3 grid(a.ntiles * a.splits), block(256);
    if (mode == 0) hipLaunchKernelGGL(decode_gemm_kernel<0>, grid, block, shm, s, a);
    else if (mode == 1) hipLaunchKernelGGL(decode_gemm_kernel<1>, grid, block, shm, s, a);
    else hipLaunchKernelGGL(decode_gemm_kernel<2>, grid, block, shm, s, a);
    MMTG_LAUNCH_CHECK("decode_gemm");
    return MMTG_OK;
}

// ------------------------------------------------------------------ split-precision token step (round 5)
// The same fused products on (hi | lo) plane pairs: fp32 residual stream, fp32 KV cache, fp32 logits -- the decode step whose
// greedy ids equal the reference's (generate.py:117-142 in fp32 arithmetic) at a multiple of the exact-fp32 kernels' speed.
extern "C" int mmtg_decode_gemm_x3(int mode, int M, int N, int K, const void* A, long lda, long planeA, const void* W, long ldw, long planeW,
                                   float* C, long ldc, void* Cp, long ldcp, long planeC, const float* bias, const float* colsum,
                                   const float* stats_in, int np_in, float eps, int act, const float* resid, long ldr, float* stats_out,
                                   int splits, float* ws, long ws_floats, unsigned* counters, long n_counters, const float* emb_pos,
                                   const float* emb_type, const long long* type_ids, const int* pos_ptr, void* stream) {
    MMTG_REQUIRE(mode >= 0 && mode <= 2, "decode_gemm_x3: mode 0 (LN-fold), 1 (LN-fold slabs) or 2 (in-kernel split-K reduce)");
    MMTG_REQUIRE(M > 0 && N > 0 && K > 0 && A && W && (C || Cp), "decode_gemm_x3: bad arguments");
    MMTG_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0 && N % 4 == 0 && planeA % 8 == 0 && planeW % 8 == 0,
                 "decode_gemm_x3: K a multiple of 64; lda, ldw, plane distances of 8; N of 4");
    MMTG_REQUIRE(MMTG_ALIGNED16(A) && MMTG_ALIGNED16(W) && MMTG_ALIGNED16(C) && MMTG_ALIGNED16(Cp) && (!C || ldc % 4 == 0) && (!Cp || (ldcp % 4 == 0 && planeC % 4 == 0)),
                 "decode_gemm_x3: 16-byte alignment");
    MMTG_REQUIRE(planeA >= (long)(M - 1) * lda + K && planeW >= (long)(N - 1) * ldw + K && (!Cp || planeC >= (long)(M - 1) * ldcp + N),
                 "decode_gemm_x3: a lo plane must lie behind its hi plane");
    if (splits < 1) splits = 1;
    DgArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)A; a.W = (const bf16*)W; a.C = C; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.Cp = (bf16*)Cp; a.ldcp = ldcp; a.planeC = planeC;
    a.M = M; a.N = N; a.K = K;
    a.kper = cdiv(cdiv(K, splits), 64) * 64;
    a.splits = cdiv(K, a.kper);
    a.tiles_n = cdiv(N, 64);
    a.ntiles = cdiv(M, 64) * a.tiles_n;
    const long bytesA = (planeA + (long)(M - 1) * lda + K) * 2, bytesW = (planeW + (long)(N - 1) * ldw + K) * 2;
    MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesW < 0x7FFFFF00L, "decode_gemm_x3: plane pairs must stay below 2 GiB");
    a.bytesA = (int)bytesA; a.bytesW = (int)bytesW; a.planeA = (int)(planeA * 2); a.planeW = (int)(planeW * 2);
    a.bias = bias; a.colsum = colsum; a.stats_in = stats_in; a.np_in = np_in; a.eps = eps; a.inv_k = 1.0f / (float)K;
    a.act = act; a.out_f32 = C != nullptr; a.resid = (const bf16*)resid; a.stats_out = stats_out; a.ws = ws; a.cnt = counters;
    a.ws_bytes = (int)(ws_floats < (1L << 29) ? ws_floats * 4 : 0x7FFFFFF0L);
    if (mode != 2) {
        MMTG_REQUIRE(colsum && stats_in && np_in >= 1 && np_in <= DG_NP && MMTG_ALIGNED16(stats_in) && MMTG_ALIGNED16(colsum),
                     "decode_gemm_x3: LN-fold needs the weight column sums and the row statistics ([M][%d][2] floats, np_in partials)", DG_NP);
        MMTG_REQUIRE(mode == 1 || (bias && MMTG_ALIGNED16(bias) && a.splits == 1), "decode_gemm_x3: mode 0 is a single K slice with the folded bias");
        MMTG_REQUIRE(mode == 0 || (C && !Cp), "decode_gemm_x3: mode 1 writes fp32 slabs [splits][M][ldc]");
        MMTG_REQUIRE(mode == 1 || ((C != nullptr) != (Cp != nullptr)), "decode_gemm_x3: mode 0 writes fp32 rows OR a plane pair");
        MMTG_REQUIRE(act == MMTG_EPI_NONE || act == MMTG_EPI_GELU, "decode_gemm_x3: LN-fold activation NONE or GELU");
    } else {
        MMTG_REQUIRE(!type_ids || (emb_pos && emb_type && pos_ptr && !resid), "decode_gemm_x3: the embedding residual takes wpe, wte, type ids and the position (and no resid)");
        a.emb_pos = (const bf16*)emb_pos; a.emb_type = (const bf16*)emb_type; a.type_ids = type_ids; a.pos_ptr = pos_ptr;
        MMTG_REQUIRE(bias && MMTG_ALIGNED16(bias) && N % 32 == 0 && ldr % 4 == 0 && (!stats_out || N / 32 <= DG_NP),
                     "decode_gemm_x3: the reduce mode needs a bias; N a multiple of 32 (at most %d with statistics)", 32 * DG_NP);
        MMTG_REQUIRE(act == MMTG_EPI_NONE || act == MMTG_EPI_TANH, "decode_gemm_x3: reduce-mode activation NONE or TANH (before the residual)");
        MMTG_REQUIRE(a.splits <= 8, "decode_gemm_x3: at most 8 K splits in the reduce mode");
        MMTG_REQUIRE(a.splits == 1 || (ws && counters && ws_floats >= (long)a.ntiles * a.splits * 4096 && n_counters >= (long)a.ntiles * 4),
                     "decode_gemm_x3: split products need %ld workspace floats and %ld zeroed counters", (long)a.ntiles * a.splits * 4096, (long)a.ntiles * 4);
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const size_t shm = 4 * 2 * 64 * 128 + 2 * 64 * 4;        // (both forms: 4 x 16 KB or 2 x 32 KB)
    // MMTG_DECODE_X3_COMBINED (A/B switch): 1 = one K loop over combined stages, 0 = three passes over 16 KB stages
    static const int comb = getenv("MMTG_DECODE_X3_COMBINED") ? atoi(getenv("MMTG_DECODE_X3_COMBINED")) : 1;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)decode_gemm_x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_x3_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_x3_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_gemm_x3_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "decode_gemm_x3: cannot raise dynamic LDS");
        attr_done = true;
    }
    const dim3 grid(a.ntiles * a.splits), block(256);
    if (comb) {
        if (mode == 0) hipLaunchKernelGGL((decode_gemm_x3_kernel<0, true>), grid, block, shm, s, a);
        else if (mode == 1) hipLaunchKernelGGL((decode_gemm_x3_kernel<1, true>), grid, block, shm, s, a);
        else hipLaunchKernelGGL((decode_gemm_x3_kernel<2, true>), grid, block, shm, s, a);
    } else if (mode == 0) hipLaunchKernelGGL(decode_gemm_x3_kernel<0>, grid, block, shm, s, a);
    else if (mode == 1) hipLaunchKernelGGL(decode_gemm_x3_kernel<1>, grid, block, shm, s, a);
    else hipLaunchKernelGGL(decode_gemm_x3_kernel<2>, grid, block, shm, s, a);
    MMTG_LAUNCH_CHECK("decode_gemm_x3");
    return MMTG_OK;
}

extern "C" int mmtg_ln_fold_weights_x3(const void* W, long ldw, long planeW, const float* gamma, const float* beta, const float* bias,
                                       void* Wf, long ldf, long planeF, float* colsum, float* bias_f, int N, int K, void* stream) {
    MMTG_REQUIRE(W && gamma && beta && Wf && colsum && bias_f && N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K && ldf % 8 == 0 && ldf >= K &&
                 planeW % 8 == 0 && planeF % 8 == 0, "ln_fold_weights_x3: bad arguments (K, leading dimensions, plane distances multiples of 8)");
    MMTG_REQUIRE(MMTG_ALIGNED16(W) && MMTG_ALIGNED16(Wf), "ln_fold_weights_x3: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_MISC, s, 5.0 * N * (double)K, 8.0 * N * (double)K);
    hipLaunchKernelGGL(ln_fold_x3_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, (const bf16*)W, ldw, planeW, gamma, beta, bias, (bf16*)Wf, ldf, planeF,
                       colsum, bias_f, N, K);
    MMTG_LAUNCH_CHECK("ln_fold_weights_x3");
    return MMTG_OK;
}

extern "C" int mmtg_decode_attn_split_x3(const float* part, int splits, const float* bias, float* kcache, float* vcache, const int* keep,
                                         long ldkeep, const int* pos_ptr, void* out_planes, long plane, int B, int nH, int dh, int Tmax,
                                         void* stream) {
    MMTG_REQUIRE(part && bias && splits > 0 && kcache && vcache && keep && pos_ptr && out_planes, "decode_attn_split_x3: null pointer");
    MMTG_REQUIRE(dh == 64 && B > 0 && nH > 0 && Tmax > 0 && Tmax <= 1024 && plane % 4 == 0 && plane >= (long)B * nH * 64 && (((uintptr_t)out_planes) & 7) == 0,
                 "decode_attn_split_x3: head dim 64, Tmax <= 1024, the lo plane behind the hi plane");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 4.0 * B * nH * (double)Tmax * dh, 8.0 * B * nH * (double)Tmax * dh);
    const long slab = (long)B * 3 * nH * 64;
    static const bool kv_nt = !getenv("MMTG_DECODE_KV_NT") || atoi(getenv("MMTG_DECODE_KV_NT")) != 0;
    if (kv_nt) hipLaunchKernelGGL((decode_attn_kernel<float, true>), dim3(nH, B), dim3(64), 0, s, (const float*)nullptr, kcache, vcache, keep, ldkeep, pos_ptr,
                                  (float*)nullptr, nH, Tmax, part, splits, slab, bias, (bf16*)out_planes, plane);
    else hipLaunchKernelGGL((decode_attn_kernel<float, false>), dim3(nH, B), dim3(64), 0, s, (const float*)nullptr, kcache, vcache, keep, ldkeep, pos_ptr,
                            (float*)nullptr, nH, Tmax, part, splits, slab, bias, (bf16*)out_planes, plane);
    MMTG_LAUNCH_CHECK("decode_attn_split_x3");
    return MMTG_OK;
}

extern "C" int mmtg_decode_embed_x3(const float* table, const long long* seq, long ldseq, const float* c, void* x_planes, long plane,
                                    const int* pos_ptr, const long long* tpw_type, const long long* tpw_mask, long long* type_out, int* keep,
                                    long ldkeep, int B, int P, int S, int E, int two_sents, int V, int sent, int max_sent_num, void* stream) {
    MMTG_REQUIRE(table && seq && c && x_planes && pos_ptr && tpw_type && tpw_mask && type_out && keep, "decode_embed_x3: null pointer");
    MMTG_REQUIRE(B > 0 && E % 8 == 0 && sent > 1 && max_sent_num > 1 && plane % 4 == 0 && plane >= (long)B * E && (((uintptr_t)x_planes) & 7) == 0,
                 "decode_embed_x3: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, (double)B * E, 12.0 * B * E);
    hipLaunchKernelGGL(decode_embed_kernel<float>, dim3(B), dim3(256), 0, s, table, seq, ldseq, c, (float*)nullptr, pos_ptr, tpw_type, tpw_mask, type_out, keep,
                       ldkeep, P, S, E, two_sents, V, sent, max_sent_num, (bf16*)x_planes, plane);
    MMTG_LAUNCH_CHECK("decode_embed_x3");
    return MMTG_OK;
}
