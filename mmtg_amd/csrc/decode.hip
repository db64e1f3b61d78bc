// Generation-side kernels (reference src/generate.py:127-141): the fused logits
// processor + greedy arg-max.  One workgroup per batch row.  The reference divides a logit by
// the repetition penalty once PER OCCURRENCE of its id among the generated tokens; the workgroup
// first counts the occurrences into a 16-bit-per-vocabulary-slot LDS table (LDS atomics), then
// every lane applies that many sequential fp32 divisions to its own slots -- bit-for-bit the
// reference's arithmetic, in O(V + n) instead of the O(V n) scan of the first version (which made
// this kernel 9 % of a decode step: 122 us at 220 generated tokens).
#include "common.h"

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
        int P, int S, int E, int two_sents, int V, int sent, int max_sent_num) {
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
        *reinterpret_cast<V16t*>(dst + e) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void decode_embed_add_kernel(const T* __restrict__ g, const T* __restrict__ wpe,
        const T* __restrict__ wte, const long long* __restrict__ type_ids, const int* __restrict__ pos_ptr,
        T* __restrict__ h, int D) {
    const int b = blockIdx.x, pos = *pos_ptr;
    const long long ty = type_ids[b];
    for (int d = threadIdx.x; d < D; d += 256)
        h[(long)b * D + d] = (T)((float)g[(long)b * D + d] + (float)wpe[(long)pos * D + d] + (float)wte[ty * D + d]);
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
template <typename T>
__global__ __launch_bounds__(64) void decode_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
        const int* __restrict__ keep, long ldkeep, const int* __restrict__ pos_ptr, T* __restrict__ out,
        int nH, int Tmax, const float* __restrict__ part, int splits, long slab, const float* __restrict__ bias) {
    typedef typename Vec16<T>::type V;
    constexpr int EPL = Vec16<T>::N, OCT = 64 / EPL, KPI = 64 / OCT;      // elements per lane, lanes per key, keys per instruction
    constexpr int UN = 16, UN0 = 12;
    __shared__ float sp[1024];
    __shared__ int skeep[1024];
    __shared__ float sq[64];
    __shared__ __attribute__((aligned(16))) T sk[64];
    __shared__ __attribute__((aligned(16))) T sv[64];
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, pos = *pos_ptr;
    const int D = nH * 64;
    const T* row = qkv + (long)b * 3 * D + h * 64;
    T* kbase = kc + (((long)b * nH + h) * Tmax) * 64;
    T* vbase = vc + (((long)b * nH + h) * Tmax) * 64;
    const int nkeys = pos + 1;
    const int oc = lane % OCT, kg = lane / OCT;
    const int last = pos > 0 ? pos - 1 : 0;
    V kv0[UN0], vv0[UN0];
#pragma unroll
    for (int u = 0; u < UN0; ++u) {
        const int key = u * KPI + kg, kr = key < pos ? key : last;
        kv0[u] = *reinterpret_cast<const V*>(kbase + (long)kr * 64 + oc * EPL);
        vv0[u] = *reinterpret_cast<const V*>(vbase + (long)kr * 64 + oc * EPL);
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
    __syncthreads();
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[e] = sq[oc * EPL + e];
    const V knew = *reinterpret_cast<const V*>(sk + oc * EPL), vnew = *reinterpret_cast<const V*>(sv + oc * EPL);
#pragma unroll
    for (int u = 0; u < UN0; ++u) {
        const int key = u * KPI + kg;
        const bool isnew = key == pos;
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            a += qv[e] * (float)(isnew ? knew[e] : kv0[u][e]);
            vv0[u][e] = isnew ? vnew[e] : vv0[u][e];
        }
#pragma unroll
        for (int o = 1; o < OCT; o <<= 1) a += __shfl_xor(a, o, 64);
        if (oc == 0 && key < nkeys) sp[key] = skeep[key] ? a : -INFINITY;
    }
    // ---- scores of the later chunks: UN independent 16-byte loads per lane are requested before the first is used
    //      (the token's own row was stored above and is published by the barrier: re-read from the cache like any other)
#pragma unroll 1
    for (int k0 = UN0 * KPI; k0 < nkeys; k0 += UN * KPI) {
        V kv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int key = k0 + u * KPI + kg;
            kv[u] = *reinterpret_cast<const V*>(kbase + (long)(key < nkeys ? key : pos) * 64 + oc * EPL);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int key = k0 + u * KPI + kg;
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) a += qv[e] * (float)kv[u][e];
#pragma unroll
            for (int o = 1; o < OCT; o <<= 1) a += __shfl_xor(a, o, 64);
            if (oc == 0 && key < nkeys) sp[key] = skeep[key] ? a : -INFINITY;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < nkeys; key += 64) mx = fmaxf(mx, sp[key]);
    mx = wave_max(mx);
    const float muse = mx == -INFINITY ? 0.f : mx;
    float sum = 0.f;
    for (int key = lane; key < nkeys; key += 64) {
        const float s = sp[key];
        const float p = s == -INFINITY ? 0.f : expf(s - muse);
        sp[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    // ---- values: the first chunk from the rows already in registers, later chunks from the cache
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
    for (int u = 0; u < UN0; ++u) {
        const int key = u * KPI + kg;
        const float pp = key < nkeys ? sp[key] : 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += pp * (float)vv0[u][e];
    }
#pragma unroll 1
    for (int k0 = UN0 * KPI; k0 < nkeys; k0 += UN * KPI) {
        V vv[UN];
        float pp[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int key = k0 + u * KPI + kg;
            vv[u] = *reinterpret_cast<const V*>(vbase + (long)(key < nkeys ? key : pos) * 64 + oc * EPL);
            pp[u] = key < nkeys ? sp[key] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] += pp[u] * (float)vv[u][e];
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
        for (int o = OCT; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (kg == 0) {
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        V o;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = (T)(acc[e] * inv);
        *reinterpret_cast<V*>(out + (long)b * D + h * 64 + oc * EPL) = o;
    }
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
                                     const int* pos_ptr, void* h, int B, int D, void* stream) {
    MMTG_REQUIRE(g && wpe && wte && type_ids && pos_ptr && h && B > 0 && D > 0, "decode_embed_add: bad args");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 2.0 * B * D, (dtype == MMTG_F32 ? 16.0 : 8.0) * B * D);
    if (dtype == MMTG_F32)
        hipLaunchKernelGGL(decode_embed_add_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)g, (const float*)wpe, (const float*)wte, type_ids, pos_ptr, (float*)h, D);
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(decode_embed_add_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)g, (const bf16*)wpe, (const bf16*)wte, type_ids, pos_ptr, (bf16*)h, D);
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
    else if (dtype == MMTG_BF16)
        hipLaunchKernelGGL(decode_attn_kernel<bf16>, grid, block, 0, s, (const bf16*)nullptr, (bf16*)kcache, (bf16*)vcache, keep, ldkeep, pos_ptr, (bf16*)out, nH, Tmax, part, splits, slab, bias);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "decode_attn_split: bad dtype");
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
