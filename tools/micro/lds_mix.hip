// Do L2 -> LDS fills (buffer_load ... lds) and LDS fragment reads (ds_read_b128) of OTHER waves on the same
// CU overlap?  512-thread workgroups: waves 0-3 fill 32 KB tiles (as the GEMM K loop does), waves 4-7 stream
// conflict-free ds_read_b128 from a separate 32 KB region.  mode 1 = fills only, 2 = reads only, 3 = both.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool TO_REGS>
__global__ __launch_bounds__(512) void mix_loop(const char* src, int bytes, int fill_iters, int read_iters, int mode, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (!(mode & 1)) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, bytes, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 junk = {0, 0, 0, 0};
        for (int it = 0; it < fill_iters; ++it) {
            const int soff = ((blockIdx.x * 37 + it * 11) % (bytes / 32768 - 1)) * 32768;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int blk = wave + 4 * i;
                if (TO_REGS) junk ^= __builtin_amdgcn_raw_buffer_load_b128(rs, blk * 1024 + lane * 16, soff, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, smem + blk * 1024), 16, blk * 1024 + lane * 16, soff, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (TO_REGS && junk[0] == 0x12345678u) sink[threadIdx.x] = 1.f;
    } else {
        if (!(mode & 2)) return;
        const char* base = smem + 32768 + (wave - 4) * 8192 + lane * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < read_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += *reinterpret_cast<const volatile f32x4*>(base + i * 1024);
        }
        if (acc[0] == 1.2345f) sink[threadIdx.x] = acc[1];
    }
}

extern "C" float mix_run(int to_regs, const char* src, int bytes, int fill_iters, int read_iters, int mode, int blocks, int lds, float* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (to_regs) {
            (void)hipFuncSetAttribute((const void*)mix_loop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((mix_loop<true>), dim3(blocks), dim3(512), lds, 0, src, bytes, fill_iters, read_iters, mode, sink);
        } else {
            (void)hipFuncSetAttribute((const void*)mix_loop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((mix_loop<false>), dim3(blocks), dim3(512), lds, 0, src, bytes, fill_iters, read_iters, mode, sink);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
