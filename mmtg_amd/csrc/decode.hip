// Generation-side kernels (reference src/generate.py:127-141): the fused logits
// processor + greedy arg-max.  One workgroup per batch row; the generated ids
// are staged in LDS and every lane scans them for its own vocabulary slots so
// the reference's "divide once PER OCCURRENCE" repetition penalty is reproduced
// bit-for-bit (sequential fp32 divisions).
#include "common.h"

namespace {

constexpr int MAXGEN = 2048;

__global__ __launch_bounds__(256) void logits_argmax_kernel(const float* __restrict__ logits, long ldl, int V,
        const long long* __restrict__ generated, long ldg, const int* __restrict__ gen_len,
        float temperature, float rep_penalty, long long* __restrict__ next) {
    __shared__ int sgen[MAXGEN];
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(gen_len[b], MAXGEN);
    for (int i = tid; i < n; i += 256) sgen[i] = (int)generated[(long)b * ldg + i];
    __syncthreads();
    if (n > 0 && sgen[n - 1] == 0) {  // sticky PAD (generate.py:137-138)
        if (tid == 0) next[b] = 0;
        return;
    }
    const float* row = logits + (long)b * ldl;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        float x = row[v];
        if (v != 0 && v != 102) {
            for (int i = 0; i < n; ++i)
                if (sgen[i] == v) x = x / rep_penalty;
        }
        x = x / temperature;
        if (v == 1 || v == 2 || v == 100 || v == 102) x = -INFINITY;
        if (x > best || (x == best && v < besti)) { best = x; besti = v; }
    }
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

}  // namespace

extern "C" int mmtg_logits_process_argmax(const float* logits, long ldl, int V, const long long* generated,
                                          long ldg, const int* gen_len, float temperature, float rep_penalty,
                                          long long* next, int B, void* stream) {
    MMTG_REQUIRE(logits && generated && gen_len && next && B > 0 && V > 0 && ldl >= V, "logits_process_argmax: bad args");
    MMTG_REQUIRE(temperature > 0.f && rep_penalty > 0.f, "logits_process_argmax: temperature / penalty must be > 0");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_DECODE, s, 4.0 * B * V, 4.0 * B * V);
    hipLaunchKernelGGL(logits_argmax_kernel, dim3(B), dim3(256), 0, s, logits, ldl, V, generated, ldg, gen_len,
                       temperature, rep_penalty, next);
    MMTG_LAUNCH_CHECK("logits_process_argmax");
    return MMTG_OK;
}
