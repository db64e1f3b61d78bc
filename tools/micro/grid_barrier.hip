// Microbenchmark: what does a device-wide barrier between the stages of a persistent kernel cost on MI355X (8 XCDs, non-coherent L2s)?
// Decode's token step is 66 dependent graph nodes at a ~4 us floor; a persistent per-token kernel would replace
// node boundaries by such barriers.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gb tools/micro/grid_barrier.hip && /tmp/gb
//   mode 0: flat      -- every workgroup bumps ONE agent-scope counter (relaxed) and polls it
//   mode 1: two-level -- a counter per XCD (workgroup id % 8), the XCD's last arriver bumps the global one; everybody polls a generation word
//   mode 2: mode 1 + a 1 KB hand-over per workgroup through FENCES: plain stores, release on the arrival atomic, acquire after the poll
//           (buffer_wbl2 / buffer_inv of the XCD's whole L2), neighbour on another XCD reads and checks
//   mode 3: mode 1 + the same hand-over through agent-scope (sc1: write-through / always-miss) stores and loads, relaxed counters,
//           s_waitcnt vmcnt(0) before the arrival (the scheme of csrc/wgrad.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* flat, unsigned* xcd, unsigned* gen, float* data, int iters, unsigned* bad) {
    const int nb = gridDim.x, b = blockIdx.x, x = b & 7;
    const int per_xcd = (nb + 7 - x) / 8;          // workgroups with id % 8 == x
    unsigned errors = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2) data[(size_t)b * 256 + threadIdx.x] = (float)(it * 1024 + b);
        if constexpr (MODE == 3) {
            __hip_atomic_store(data + (size_t)b * 256 + threadIdx.x, (float)(it * 1024 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            constexpr int REL = MODE == 2 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED;
            if constexpr (MODE == 0) {
                __hip_atomic_fetch_add(flat, 1u, REL, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)(it + 1) * nb;
                while (ld_agent(flat) < want) __builtin_amdgcn_s_sleep(1);
            } else {
                const unsigned old = __hip_atomic_fetch_add(xcd + x * 32, 1u, REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)(it + 1) * per_xcd - 1) {
                    const unsigned g = __hip_atomic_fetch_add(flat, 1u, REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (g == (unsigned)(it + 1) * 8 - 1) __hip_atomic_store(gen, (unsigned)(it + 1), REL, __HIP_MEMORY_SCOPE_AGENT);
                }
                while (ld_agent(gen) < (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (MODE == 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        if constexpr (MODE >= 2) {
            const int nbr = (b + 37) % nb;           // a workgroup on another XCD
            float got;
            if constexpr (MODE == 2) got = data[(size_t)nbr * 256 + threadIdx.x];
            else got = __hip_atomic_load(data + (size_t)nbr * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (float)(it * 1024 + nbr)) ++errors;
            data += (it & 1) ? -(ptrdiff_t)nb * 256 : (ptrdiff_t)nb * 256;       // two alternating buffers
        }
    }
    if (errors) atomicAdd(bad, errors);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *flat, *xcd, *gen, *bad;
    float* data;
    CHECK(hipMalloc(&flat, 4096)); CHECK(hipMalloc(&xcd, 4096)); CHECK(hipMalloc(&gen, 4096)); CHECK(hipMalloc(&bad, 4096));
    CHECK(hipMalloc(&data, sizeof(float) * 2 * 1024 * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs; us per barrier over %d iterations\n", prop.name, cus, iters);
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int mode = 0; mode < 4; ++mode) {
            const int nb = cus * per_cu;
            float best = 1e30f;
            unsigned hbad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(flat, 0, 4096)); CHECK(hipMemset(xcd, 0, 4096)); CHECK(hipMemset(gen, 0, 4096)); CHECK(hipMemset(bad, 0, 4096));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(nb), dim3(256), 0, 0, flat, xcd, gen, data, iters, bad);
                if (mode == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(nb), dim3(256), 0, 0, flat, xcd, gen, data, iters, bad);
                if (mode == 2) hipLaunchKernelGGL(barrier_kernel<2>, dim3(nb), dim3(256), 0, 0, flat, xcd, gen, data, iters, bad);
                if (mode == 3) hipLaunchKernelGGL(barrier_kernel<3>, dim3(nb), dim3(256), 0, 0, flat, xcd, gen, data, iters, bad);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            }
            printf("workgroups %4d (%d per CU)  mode %d  %8.3f us per barrier%s\n", nb, per_cu, mode, best * 1e3f / iters,
                   mode >= 2 ? (hbad ? "  HAND-OVER ERRORS" : "  hand-over ok") : "");
        }
    return 0;
}
