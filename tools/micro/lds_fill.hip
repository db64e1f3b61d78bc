// L2 -> LDS fill-rate probe (buffer_load ... lds, 16 B per lane = 1 KB per wave-instruction): how many
// bytes per clock a CU can pull from its XCD's L2 as a function of the access shape, the number of
// workgroups per CU and the number of 32 KB tiles each keeps in flight.  No MFMAs, no LDS reads.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// shape 0: 8 rows x 128 B per instruction (K-contiguous operand rows, row stride ld)
// shape 1: 4 rows x 256 B per instruction (K-strided operand rows)
// shape 2: 1 KB contiguous per instruction (pre-tiled operand)
template <int SHAPE, int DEPTH>
__global__ __launch_bounds__(256) void fill_loop(const char* src, int bytes, int ld, int rows, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, bytes, 0x00020000);
    int voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int blk = wave + 4 * i;
        if (SHAPE == 0) voff[i] = (blk * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
        else if (SHAPE == 1) voff[i] = (blk * 4 + (lane >> 4)) * ld + (lane & 15) * 16;
        else voff[i] = blk * 1024 + lane * 16;
    }
    const int w = blockIdx.x;
    for (int it = 0; it < iters + DEPTH - 1; ++it) {
        if (it < iters) {
            int soff;
            if (SHAPE == 0) soff = ((w * 37 + it * 11) % (rows - 256)) * ld + ((it * 5 + w) % (ld / 128)) * 128;
            else if (SHAPE == 1) soff = ((w * 37 + it * 11) % (rows - 128)) * ld + ((it * 5 + w) % (ld / 256)) * 256;
            else soff = ((w * 37 + it * 11) % (bytes / 32768 - 1)) * 32768;
            char* st = smem + (it % DEPTH) * 32768;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, st + (wave + 4 * i) * 1024), 16, voff[i], soff, 0, 0);
        }
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the older tile has landed
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[(blockIdx.x * 16) & 32767];
}

template <int SHAPE, int DEPTH>
static float run(const char* src, int bytes, int ld, int rows, int iters, int blocks, int lds, int* sink) {
    hipFuncSetAttribute((const void*)fill_loop<SHAPE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fill_loop<SHAPE, DEPTH>), dim3(blocks), dim3(256), lds, 0, src, bytes, ld, rows, iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

extern "C" float fill_run(int shape, int depth, const char* src, int bytes, int ld, int rows, int iters, int blocks, int lds, int* sink) {
    if (shape == 0) return depth == 1 ? run<0, 1>(src, bytes, ld, rows, iters, blocks, lds, sink) : run<0, 2>(src, bytes, ld, rows, iters, blocks, lds, sink);
    if (shape == 1) return depth == 1 ? run<1, 1>(src, bytes, ld, rows, iters, blocks, lds, sink) : run<1, 2>(src, bytes, ld, rows, iters, blocks, lds, sink);
    return depth == 1 ? run<2, 1>(src, bytes, ld, rows, iters, blocks, lds, sink) : run<2, 2>(src, bytes, ld, rows, iters, blocks, lds, sink);
}
