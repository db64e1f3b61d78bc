// Bare MFMA issue-rate probe: operands in registers, no memory traffic in the loop.
// waves_per_simd = blocks/CU * 256 / 64 / 4.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed, unsigned long long* clk) {
    // shader clock actually sustained under this load: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + 0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.5f - 0.002f * (threadIdx.x % 7 + e)); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// the 32x32x16 form: twice the work per instruction (16 accumulator registers per tile)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma32_loop(float* out, int iters, float seed, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + 0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.5f - 0.002f * (threadIdx.x % 7 + e)); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

extern "C" float mfma_run(int blocks, int iters, int nacc, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (nacc == 32) hipLaunchKernelGGL(mfma32_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25f, clk);
        else if (nacc == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25f, clk);
        else hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25f, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
