// LDS-tiled MFMA GEMM for gfx950 with fused epilogues.
//
//   C[m,n] = epi( sum_k opA(m,k) * opB(k,n) )
//
// One 256-thread workgroup (4 waves) owns a 128x128 output tile; each wave a
// 64x64 quadrant as 4x4 MFMA 16x16 tiles (bf16: v_mfma_f32_16x16x32_bf16,
// f32: v_mfma_f32_16x16x4_f32 -- exact fp32, used by the parity-gate mode).
// K advances 128 BYTES per tile (64 bf16 / 32 f32).
//
// Operand layouts ("KC" = K-contiguous rows, "KS" = K-strided):
//   A KC: A[m*lda + k]      A KS: A[k*lda + m]
//   B KC: B[n*ldb + k]      B KS: B[k*ldb + n]
// so forward / dgrad of nn.Linear ([out,in]) and Conv1D ([in,out]) weights and
// the weight-gradient product X^T dY all run without materialised transposes.
//
// Two staging pipelines over the same LDS images (gemm_common.h):
//   gemm_dma_kernel  global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no
//                    ds_write): NBUF-deep ring, loads of tile t+NBUF-1 issued right after the
//                    barrier of tile t, counted s_waitcnt vmcnt, ONE raw s_barrier per K tile.
//                    The LDS destination of a wave-instruction is lane-linear (1 KB), so the
//                    XOR swizzle is applied to each lane's SOURCE address; out-of-range chunks
//                    read a global zero page.
//   gemm_kernel      register-staged double buffer (v1); kept for the f32 mode and as the A/B
//                    reference (flags & MMTG_GEMM_REGSTAGE).
#include "gemm_common.h"

namespace {

__device__ __attribute__((aligned(16))) char g_zero_page[16] = {0};

// ------------------------------------------------------------------ register-staged pipeline
template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(NTHR, 2) void gemm_kernel(GemmArgs p) {
    typedef typename Vec16<T>::type V;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    tile_origin(p, m0, n0);
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const int nk = (kend - kbeg + GT<T>::BK - 1) / GT<T>::BK;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    V ra[4], rb[4];
    stage_load<T, AKS>(A, p.lda, m0, p.M, kbeg, kend, tid, ra);
    stage_load<T, BKS>(B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
    stage_store<T, AKS>(smem, tid, ra);
    stage_store<T, BKS>(smem + TILE_BYTES, tid, rb);
    __syncthreads();
    constexpr bool std_orient = AKS && BKS;
    int oa[4], ob[4];
    ks_lane_offsets(wm, lane, oa);
    ks_lane_offsets(wn, lane, ob);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = kbeg + (kt + 1) * GT<T>::BK;
            stage_load<T, AKS>(A, p.lda, m0, p.M, k0, kend, tid, ra);
            stage_load<T, BKS>(B, p.ldb, n0, p.N, k0, kend, tid, rb);
        }
        const char* tA = smem + cur * 2 * TILE_BYTES;
        compute_tile<T, AKS, BKS, std_orient>(tA, tA + TILE_BYTES, acc, wm, wn, lane, p.use_tr, oa, ob);
        if (kt + 1 < nk) {
            stage_store<T, AKS>(smem + (cur ^ 1) * 2 * TILE_BYTES, tid, ra);
            stage_store<T, BKS>(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, tid, rb);
        }
        __syncthreads();
    }
    gemm_epilogue<T, std_orient>(p, acc, m0, n0, wm, wn, g, l15);
}

// ------------------------------------------------------------------ LDS-DMA pipeline
// One wave-instruction moves 64 lanes x 16 B = 1 KB into LDS at (wave-uniform base + lane*16).
// Tile = 16 such 1-KB blocks; wave w issues blocks 4w..4w+3 of each operand tile.
template <typename T, bool KS>
__device__ __forceinline__ const T* dma_src(const T* __restrict__ base, long ld, int row0, int nrows,
                                            int k0, int kend, int blk, int lane) {
    constexpr int EPC = GT<T>::EPC;
    if (!KS) {
        const int r = blk * 8 + (lane >> 3), pc = lane & 7;
        const int c = pc ^ (r & 7);
        const int gr = row0 + r, gk = k0 + c * EPC;
        return (gr < nrows && gk < kend) ? base + (long)gr * ld + gk : reinterpret_cast<const T*>(g_zero_page);
    } else {
        constexpr int CPR = GT<T>::CPR;              // 16 (bf16) / 32 (f32) chunks per k-row
        constexpr int RPB = 64 / CPR;                // k-rows per 1-KB block: 4 / 2
        const int k = blk * RPB + lane / CPR, pc = lane % CPR;
        const int c = pc ^ ks_swz(k);
        const int gk = k0 + k, gc = row0 + c * EPC;
        return (gk < kend && gc < nrows) ? base + (long)gk * ld + gc : reinterpret_cast<const T*>(g_zero_page);
    }
}

template <typename T, bool AKS, bool BKS>
__device__ __forceinline__ void dma_issue(const GemmArgs& p, const T* A, const T* B, int m0, int n0, int k0, int kend,
                                          char* stage, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int blk = wave * 4 + i;
        const T* sa = dma_src<T, AKS>(A, p.lda, m0, p.M, k0, kend, blk, lane);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                         LDS_PTR(void, stage + blk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int blk = wave * 4 + i;
        const T* sb = dma_src<T, BKS>(B, p.ldb, n0, p.N, k0, kend, blk, lane);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                         LDS_PTR(void, stage + TILE_BYTES + blk * 1024), 16, 0, 0);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
}

template <typename T, bool AKS, bool BKS, int NBUF>
__global__ __launch_bounds__(NTHR, (NBUF == 2 ? 2 : 1)) void gemm_dma_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF stages of [A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    tile_origin(p, m0, n0);
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const int nk = (kend - kbeg + GT<T>::BK - 1) / GT<T>::BK;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    constexpr bool std_orient = AKS && BKS;
    constexpr int STAGE = 2 * TILE_BYTES;
    int oa[4], ob[4];
    ks_lane_offsets(wm, lane, oa);
    ks_lane_offsets(wn, lane, ob);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: tiles 0 .. NBUF-2 (a tile index past the end still issues: every chunk is then out
    // of range and reads the zero page, which keeps the vmcnt arithmetic uniform)
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        dma_issue<T, AKS, BKS>(p, A, B, m0, n0, kbeg + s * GT<T>::BK, kend, smem + s * STAGE, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt's loads were issued NBUF-1 tiles ago; (NBUF-2) younger tiles (8 loads each) may stay in flight
        wait_vmcnt<8 * (NBUF - 2)>();
        __builtin_amdgcn_s_barrier();      // tile kt landed for every wave; everyone is done with tile kt-1
        dma_issue<T, AKS, BKS>(p, A, B, m0, n0, kbeg + (kt + NBUF - 1) * GT<T>::BK, kend,
                               smem + ((kt + NBUF - 1) % NBUF) * STAGE, wave, lane);
        const char* tA = smem + (kt % NBUF) * STAGE;
        compute_tile<T, AKS, BKS, std_orient>(tA, tA + TILE_BYTES, acc, wm, wn, lane, p.use_tr, oa, ob);
    }
    wait_vmcnt<0>();   // drain the dummy tail loads before the LDS is released
    gemm_epilogue<T, std_orient>(p, acc, m0, n0, wm, wn, g, l15);
}

// ------------------------------------------------------------------ launch
template <typename K> int set_lds(K kern, size_t bytes) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        MMTG_FAIL(MMTG_ERR_HIP, "gemm: cannot raise dynamic LDS to %zu bytes", bytes);
    return MMTG_OK;
}

template <typename T>
int launch_regstage(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
    dim3 block(NTHR);
    if (!transA && transB) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, stream, a);
    else if (!transA && !transB) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, block, 0, stream, a);
    return MMTG_OK;
}

template <typename T, int NBUF>
int launch_dma(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
    dim3 block(NTHR);
    const size_t shm = (size_t)NBUF * 2 * TILE_BYTES;
    static bool attr_done[3] = {false, false, false};
    const int li = (!transA && transB) ? 0 : (!transA && !transB) ? 1 : 2;
    if (!attr_done[li]) {
        int rc = li == 0 ? set_lds(gemm_dma_kernel<T, false, false, NBUF>, shm)
               : li == 1 ? set_lds(gemm_dma_kernel<T, false, true, NBUF>, shm)
                         : set_lds(gemm_dma_kernel<T, true, true, NBUF>, shm);
        if (rc) return rc;
        attr_done[li] = true;
    }
    if (li == 0) hipLaunchKernelGGL((gemm_dma_kernel<T, false, false, NBUF>), grid, block, shm, stream, a);
    else if (li == 1) hipLaunchKernelGGL((gemm_dma_kernel<T, false, true, NBUF>), grid, block, shm, stream, a);
    else hipLaunchKernelGGL((gemm_dma_kernel<T, true, true, NBUF>), grid, block, shm, stream, a);
    return MMTG_OK;
}

}  // namespace

// C ABI ---------------------------------------------------------------------
extern "C" int mmtg_gemm(int dtype, int transA, int transB, int M, int N, int K,
                         const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                         const float* bias, int epi, const void* aux, long ldaux, void* aux2,
                         int out_f32, float alpha, int splits, unsigned drop_thresh, unsigned drop_seed,
                         int flags, void* stream) {
    MMTG_REQUIRE(dtype == MMTG_F32 || dtype == MMTG_BF16, "gemm: bad dtype %d", dtype);
    MMTG_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem %d %d %d", M, N, K);
    MMTG_REQUIRE(A && B && C, "gemm: null operand");
    MMTG_REQUIRE(!(transA && transB), "gemm: layout transA=1,transB=1 is not built");
    const int epc = dtype == MMTG_F32 ? 4 : 8;
    MMTG_REQUIRE(MMTG_ALIGNED16(A) && MMTG_ALIGNED16(B) && MMTG_ALIGNED16(C), "gemm: operands must be 16-byte aligned");
    MMTG_REQUIRE(lda % epc == 0 && ldb % epc == 0, "gemm: lda/ldb must be multiples of %d elements", epc);
    // the contiguous extent of every operand tile is read in whole 16-byte chunks
    MMTG_REQUIRE((transA ? M : K) % epc == 0, "gemm: contiguous extent of A (%d) must be a multiple of %d", transA ? M : K, epc);
    MMTG_REQUIRE((transB ? K : N) % epc == 0, "gemm: contiguous extent of B (%d) must be a multiple of %d", transB ? K : N, epc);
    MMTG_REQUIRE((epi == MMTG_EPI_ATOMIC) == (transA && !transB),
                 "gemm: the atomic epilogue and the transA=1,transB=0 (weight-gradient) layout go together");
    if (epi != MMTG_EPI_ATOMIC) {
        MMTG_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "gemm: N and ldc must be multiples of 4 (N=%d ldc=%ld)", N, ldc);
        MMTG_REQUIRE(!bias || MMTG_ALIGNED16(bias), "gemm: bias must be 16-byte aligned");
        MMTG_REQUIRE(splits <= 1, "gemm: split-K needs the atomic epilogue");
    } else {
        MMTG_REQUIRE(!bias, "gemm: atomic epilogue takes no bias");
    }
    if (epi == MMTG_EPI_RESID || epi == MMTG_EPI_DGELU || epi == MMTG_EPI_DTANH)
        MMTG_REQUIRE(aux && ldaux % 4 == 0 && (((uintptr_t)aux) & 7) == 0, "gemm: epilogue %d needs an aligned aux operand", epi);
    if (epi == MMTG_EPI_GELU) MMTG_REQUIRE(aux2, "gemm: GELU epilogue needs aux2 for the pre-activation");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.aux2 = aux2;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
    a.epi = epi; a.out_f32 = out_f32; a.use_tr = !(flags & MMTG_GEMM_NO_TR);
    a.tiles_n = cdiv(N, BN); a.alpha = alpha;
    const int bk = dtype == MMTG_F32 ? 32 : 64;
    if (splits < 1) splits = 1;
    int kper = cdiv(cdiv(K, splits), bk) * bk;
    splits = cdiv(K, kper);
    a.kper = kper;
    a.drop_thresh = drop_thresh; a.drop_seed = drop_seed;
    a.drop_inv_keep = drop_thresh ? (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh)) : 1.0f;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(dtype == MMTG_F32 ? MMTG_PROF_GEMM_F32 : MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K,
                   (double)(dtype == MMTG_F32 ? 4 : 2) * ((double)M * K + (double)N * K) + (double)M * N * (out_f32 ? 4 : (dtype == MMTG_F32 ? 4 : 2)));
    dim3 grid(cdiv(M, BM) * cdiv(N, BN), 1, splits);
    int rc;
    if (dtype == MMTG_F32) rc = launch_regstage<float>(a, transA, transB, grid, s);
    else if (flags & MMTG_GEMM_REGSTAGE) rc = launch_regstage<bf16>(a, transA, transB, grid, s);
    else if (flags & MMTG_GEMM_3STAGE) rc = launch_dma<bf16, 3>(a, transA, transB, grid, s);
    else rc = launch_dma<bf16, 2>(a, transA, transB, grid, s);
    if (rc) return rc;
    MMTG_LAUNCH_CHECK("gemm");
    return MMTG_OK;
}
