// What a dependent kernel boundary costs inside a captured hipGraph, by launch shape (round 6).
// The decode token step is ~60 dependent launches; its kernels' own timelines add up to ~4-5 us LESS per launch than the replay
// period (tools/decode_mlp_timeline.py: 15.8 us first start -> last end inside a 20.5 us period).  This probe replays chains of N
// dependent kernels of a given shape and prints the period per node, to see which launch attribute the gap follows.
//   hipcc --offload-arch=gfx950 -O3 -o node_floor tools/micro/node_floor.hip && ./node_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct BigArgs { long a[40]; };        // 320 bytes of kernel arguments (the decode products pass ~250)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFF) *p = 1; }
__global__ void k_lds(int* p) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    if (p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFF) *p = smem[0];
}
__global__ void k_args(BigArgs a, int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFF) *p = (int)a.a[3]; }
// every thread stores 16 bytes: blocks * threads * 16 bytes left dirty at the boundary
__global__ void k_dirty(float4* p) { p[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); }
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ void k_dirty_sc1(float4* p) {
    __builtin_nontemporal_store(f32x4_t{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f32x4_t*>(p) + (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// holds every CU for ~`ticks` of the 100 MHz counter: a kernel with a body, so that the boundary is not hidden by an empty queue
__global__ void k_busy(int ticks, int* p) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    }
    if (p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFF) *p = smem[0];
}

template <typename F>
static int run(const char* name, int nodes, double body_us, F launch) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 4; ++i) launch(s);
    CK(hipStreamSynchronize(s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per = 1e3 * ms / (reps * nodes);
    // eager (host-paced) for comparison
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < nodes * 4; ++i) launch(s);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-66s graph %6.2f us/node (gap %5.2f)   eager %6.2f us/launch\n", name, per, per - body_us, 1e3 * ms / (nodes * 4));
    fflush(stdout);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
    return 0;
}

int main() {
    int* d;
    float4* buf;
    CK(hipMalloc(&d, 4096));
    CK(hipMalloc(&buf, (size_t)64 << 20));
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 64;
    BigArgs big;
    memset(&big, 0, sizeof(big));
    run("empty <<<1, 64>>>", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d); });
    run("empty <<<256, 256>>>", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, d); });
    run("empty <<<3072, 64>>> (decode_attn's grid)", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(3072), dim3(64), 0, s, d); });
    run("<<<256, 256>>> + 157.5 KB dynamic LDS (decode_mlp)", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 161280, s, d); });
    run("<<<288, 256>>> + 64.5 KB dynamic LDS (decode_gemm)", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(288), dim3(256), 66048, s, d); });
    run("<<<256, 256>>> + 320 B of kernel arguments", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_args, dim3(256), dim3(256), 0, s, big, d); });
    run("<<<256, 256>>> storing 1 MB (plain stores)", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_dirty, dim3(256), dim3(256), 0, s, buf); });
    run("<<<2048, 256>>> storing 8 MB (plain stores)", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_dirty, dim3(2048), dim3(256), 0, s, buf); });
    run("<<<2048, 256>>> storing 8 MB (non-temporal stores)", N, 0, [&](hipStream_t s) { hipLaunchKernelGGL(k_dirty_sc1, dim3(2048), dim3(256), 0, s, buf); });
    for (int us : {5, 10, 20}) {
        char name[128];
        snprintf(name, sizeof name, "<<<256, 256>>> + 157.5 KB LDS, every CU held %d us", us);
        run(name, N, us, [&](hipStream_t s) { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 161280, s, us * 100, d); });
        snprintf(name, sizeof name, "<<<256, 256>>> no LDS, every CU held %d us", us);
        run(name, N, us, [&](hipStream_t s) { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 16, s, us * 100, d); });
        snprintf(name, sizeof name, "<<<3072, 64>>> no LDS, every workgroup holds %d us (12 per CU)", us);
        run(name, N, us, [&](hipStream_t s) { hipLaunchKernelGGL(k_busy, dim3(3072), dim3(64), 16, s, us * 100, d); });
    }
    return 0;
}
