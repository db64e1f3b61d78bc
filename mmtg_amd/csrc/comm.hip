// The data-parallel exchange as a second small ABI: one RCCL communicator per process (= per GPU), gradient
// buckets all-reduced (SUM, in place) over xGMI.  Replaces what the reference gets from torch.nn.DataParallel (src/train.py:112-114:
// reduce_add_coalesced of every gradient to GPU 0 + a parameter broadcast per step) with the one exchange the arithmetic needs.
//
// RCCL is bound at first use with dlopen / dlsym -- the copy the host process already holds (PyTorch ships one) when there is one --
// so the kernel library itself carries no link-time dependency on it and loads on a box without RCCL; the comm entry points then
// fail with a message, like everything else here: nothing falls back to a host path.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

static_assert(NCCL_UNIQUE_ID_BYTES == MMTG_COMM_ID_BYTES, "include/mmtg_hip.h: MMTG_COMM_ID_BYTES is RCCL's ncclUniqueId");

namespace {
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    char where[256] = "";
};
Rccl g_rccl;
std::mutex g_mu;               // init / destroy / the fork-join bookkeeping; collectives of one process are issued from one thread

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0, dev = -1;
    hipStream_t side = nullptr;         // the exchange's own stream (mmtg_allreduce_bucket_async)
    hipEvent_t ready = nullptr, done = nullptr;
    bool pending = false;               // something was enqueued on `side` since the last join
};
Comm g_comm;

// the host's copy first (RTLD_NOLOAD: PyTorch's librccl.so has no SONAME and is known to the loader under the name it was asked
// for), then MMTG_RCCL_LIB, then the system one
int rccl_bind() {
    if (g_rccl.h) return MMTG_OK;
    const char* env = getenv("MMTG_RCCL_LIB");
    struct { const char* name; int flags; } tries[] = {
        {env, RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
    };
    void* h = nullptr;
    for (auto& t : tries) {
        if (!t.name || !*t.name) continue;
        h = dlopen(t.name, t.flags);
        if (h) { snprintf(g_rccl.where, sizeof(g_rccl.where), "%s%s", t.name, (t.flags & RTLD_NOLOAD) ? " (already loaded)" : ""); break; }
    }
    if (!h) MMTG_FAIL(MMTG_ERR_UNSUPPORTED, "comm: RCCL not found (librccl.so / librccl.so.1 / MMTG_RCCL_LIB): %s", dlerror());
#define BIND(sym)                                                                                    \
    do {                                                                                             \
        g_rccl.sym = reinterpret_cast<decltype(g_rccl.sym)>(dlsym(h, "nccl" #sym));                  \
        if (!g_rccl.sym) { dlclose(h); MMTG_FAIL(MMTG_ERR_UNSUPPORTED, "comm: %s has no nccl" #sym, g_rccl.where); } \
    } while (0)
    BIND(GetUniqueId); BIND(CommInitRank); BIND(AllReduce); BIND(CommDestroy); BIND(GetErrorString); BIND(GetVersion);
#undef BIND
    g_rccl.h = h;
    return MMTG_OK;
}

#define RCCL_CHECK(call, what)                                                                       \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) MMTG_FAIL(MMTG_ERR_HIP, "comm: %s: %s", what, g_rccl.GetErrorString(r_)); \
    } while (0)
#define HIP_CHECK(call, what)                                                                        \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) MMTG_FAIL(MMTG_ERR_HIP, "comm: %s: %s", what, hipGetErrorString(e_));  \
    } while (0)

int reduce_on(void* ptr, long count, int dtype, hipStream_t s) {
    MMTG_REQUIRE(g_comm.comm, "allreduce_bucket: no communicator (mmtg_comm_init first)");
    MMTG_REQUIRE(ptr && count > 0, "allreduce_bucket: null pointer or count %ld", count);
    MMTG_REQUIRE(dtype == MMTG_F32 || dtype == MMTG_BF16, "allreduce_bucket: dtype %d (MMTG_F32 or MMTG_BF16)", dtype);
    MMTG_REQUIRE(((uintptr_t)ptr & (dtype == MMTG_F32 ? 3 : 1)) == 0, "allreduce_bucket: misaligned bucket");
    RCCL_CHECK(g_rccl.AllReduce(ptr, ptr, (size_t)count, dtype == MMTG_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, g_comm.comm, s),
               "ncclAllReduce");
    return MMTG_OK;
}
}  // namespace

extern "C" int mmtg_comm_unique_id(void* id) {
    MMTG_REQUIRE(id, "comm_unique_id: null pointer");
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = rccl_bind()) return rc;
    ncclUniqueId u;
    RCCL_CHECK(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return MMTG_OK;
}

extern "C" int mmtg_comm_init(int rank, int world, const void* id) {
    MMTG_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    std::lock_guard<std::mutex> lk(g_mu);
    MMTG_REQUIRE(!g_comm.comm, "comm_init: this process already holds a communicator (rank %d of %d): one process per GPU", g_comm.rank, g_comm.world);
    if (int rc = rccl_bind()) return rc;
    int dev = -1;
    HIP_CHECK(hipGetDevice(&dev), "hipGetDevice");
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    RCCL_CHECK(g_rccl.CommInitRank(&c, world, u, rank), "ncclCommInitRank");
    Comm n;
    n.comm = c; n.rank = rank; n.world = world; n.dev = dev;
    int lo = 0, hi = 0;                 // the exchange ahead of the backward's kernels at every dispatch decision
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
    if (hipStreamCreateWithPriority(&n.side, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&n.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n.done, hipEventDisableTiming) != hipSuccess) {
        g_rccl.CommDestroy(c);
        MMTG_FAIL(MMTG_ERR_HIP, "comm_init: cannot create the exchange stream / events");
    }
    g_comm = n;
    return MMTG_OK;
}

extern "C" int mmtg_comm_info(int* rank, int* world, int* device, int* rccl_version) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (rank) *rank = g_comm.comm ? g_comm.rank : -1;
    if (world) *world = g_comm.comm ? g_comm.world : 0;
    if (device) *device = g_comm.comm ? g_comm.dev : -1;
    if (rccl_version) {
        *rccl_version = 0;
        if (g_rccl.h) (void)g_rccl.GetVersion(rccl_version);
    }
    return MMTG_OK;
}

extern "C" int mmtg_allreduce_bucket(void* ptr, long count, int dtype, void* stream) {
    return reduce_on(ptr, count, dtype, (hipStream_t)stream);
}

extern "C" int mmtg_allreduce_bucket_async(void* ptr, long count, int dtype, void* after_stream) {
    MMTG_REQUIRE(g_comm.comm, "allreduce_bucket_async: no communicator (mmtg_comm_init first)");
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_CHECK(hipEventRecord(g_comm.ready, (hipStream_t)after_stream), "hipEventRecord");
    HIP_CHECK(hipStreamWaitEvent(g_comm.side, g_comm.ready, 0), "hipStreamWaitEvent");
    if (int rc = reduce_on(ptr, count, dtype, g_comm.side)) return rc;
    g_comm.pending = true;
    return MMTG_OK;
}

extern "C" int mmtg_comm_join(void* stream) {
    MMTG_REQUIRE(g_comm.comm, "comm_join: no communicator (mmtg_comm_init first)");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm.pending) return MMTG_OK;
    HIP_CHECK(hipEventRecord(g_comm.done, g_comm.side), "hipEventRecord");
    HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g_comm.done, 0), "hipStreamWaitEvent");
    g_comm.pending = false;
    return MMTG_OK;
}

extern "C" int mmtg_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm.comm) return MMTG_OK;
    (void)hipStreamSynchronize(g_comm.side);
    ncclResult_t r = g_rccl.CommDestroy(g_comm.comm);
    (void)hipEventDestroy(g_comm.ready);
    (void)hipEventDestroy(g_comm.done);
    (void)hipStreamDestroy(g_comm.side);
    g_comm = Comm();
    if (r != ncclSuccess) MMTG_FAIL(MMTG_ERR_HIP, "comm_destroy: %s", g_rccl.GetErrorString(r));
    return MMTG_OK;
}
