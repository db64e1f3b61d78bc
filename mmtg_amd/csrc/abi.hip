// Error reporting, ABI version and the live per-kernel event profiler.
#include <stdarg.h>

#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void mmtg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mmtg_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mmtg_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MMTG_ERR_HIP;
    }
    return MMTG_OK;
}

extern "C" int mmtg_abi_version(void) { return MMTG_ABI_VERSION; }
#ifndef MMTG_BUILD_FLAGS
#define MMTG_BUILD_FLAGS ""
#endif
extern "C" const char* mmtg_build_flags(void) { return MMTG_BUILD_FLAGS; }
extern "C" const char* mmtg_last_error(void) { return g_err; }

// ---------------------------------------------------------------- profiler
// When enabled every launcher brackets its kernel(s) with two hipEvents on the
// launch stream.  Events are pooled and only synchronised in mmtg_prof_read(),
// so the timed region itself stays free of host syncs.
namespace {
struct Rec { int cat; hipEvent_t a, b; double flops, bytes; };
bool g_prof_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[MMTG_PROF_NCAT];

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void mmtg_prof_begin(int cat, hipStream_t s) {
    if (!g_prof_on) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, s);
    g_open[cat] = e;
}
void mmtg_prof_end(int cat, hipStream_t s, double flops, double bytes) {
    if (!g_prof_on) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, s);
    g_recs.push_back(Rec{cat, g_open[cat], e, flops, bytes});
}

extern "C" int mmtg_prof_enable(int on) {
    g_prof_on = on != 0;
    return MMTG_OK;
}

extern "C" int mmtg_prof_read(int* launches, double* ms, double* flops, double* bytes) {
    for (int i = 0; i < MMTG_PROF_NCAT; ++i) {
        launches[i] = 0; ms[i] = 0; flops[i] = 0; bytes[i] = 0;
    }
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        launches[r.cat] += 1;
        ms[r.cat] += t;
        flops[r.cat] += r.flops;
        bytes[r.cat] += r.bytes;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return MMTG_OK;
}

// ---------------------------------------------------------------- measurement hook: hold CUs
// `workgroups` 256-thread workgroups that each declare `lds_bytes` of LDS (163840 = a whole CU: nothing else fits beside it) and
// sleep until `usec` microseconds have passed on the 100 MHz wall counter -- the shape of a collective's ring kernels, which hold
// their CUs for the whole exchange.  tools/ddp_contention.py runs the training step beside it to price the CU reservation of
// mmtg_amd.ddp on ONE GPU.  Never launched by the product path.
namespace {
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, int touch) {
    extern __shared__ char occ_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (touch) occ_lds[threadIdx.x] = 0;          // keeps the LDS allocation live
}
}  // namespace

extern "C" int mmtg_debug_occupy(int workgroups, int lds_bytes, double usec, void* stream) {
    MMTG_REQUIRE(workgroups >= 1 && workgroups <= 4096 && lds_bytes >= 0 && lds_bytes <= 163840 && usec >= 0 && usec <= 5e6,
                 "debug_occupy: 1..4096 workgroups, <= 163840 B of LDS, <= 5 s");
    static int attr_bytes = -1;
    if (lds_bytes > attr_bytes) {
        if (hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "debug_occupy: cannot raise the LDS limit to %d", lds_bytes);
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), lds_bytes, (hipStream_t)stream, (unsigned long long)(usec * 100.0), 0);
    return mmtg_check_launch("debug_occupy");
}
