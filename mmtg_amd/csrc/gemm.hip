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
//                    XOR swizzle is applied to each lane's SOURCE offset; out-of-range chunks
//                    are zero-filled by the buffer descriptor's bounds check.
//   gemm_kernel      register-staged double buffer (v1); kept for the f32 mode and as the A/B
//                    reference (flags & MMTG_GEMM_REGSTAGE).
#include "gemm_common.h"

namespace {

// ------------------------------------------------------------------ register-staged pipeline
template <typename T, bool AKS, bool BKS, bool USE_TR>
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
        compute_tile<T, AKS, BKS, std_orient, USE_TR>(tA, tA + TILE_BYTES, acc, wm, wn, lane, oa, ob);
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
// Addressing is buffer-style: a wave-uniform descriptor over the whole operand, a per-lane byte
// offset that is computed ONCE (the lane's swizzled source chunk relative to the tile origin) and a
// scalar offset that advances by one K tile per iteration -> no vector address arithmetic in the
// loop.  Rows/columns outside the matrix get an out-of-range offset (the descriptor's bounds check
// returns zeros); only a ragged last K tile recomputes its offsets.
constexpr int OOB = 0x7FFFFFF0;

template <typename T, bool KS>
__device__ __forceinline__ int dma_voff(long ld, int row0, int nrows, int krem, int blk, int lane) {
    constexpr int EPC = GT<T>::EPC;
    if (!KS) {   // tile rows = operand rows (m or n), chunks along K
        const int r = blk * 8 + (lane >> 3), pc = lane & 7;
        const int c = pc ^ (r & 7);
        const bool ok = (row0 + r < nrows) && (c * EPC < krem);
        return ok ? (int)(((long)r * ld + c * EPC) * sizeof(T)) : OOB;
    } else {     // tile rows = K, chunks along the operand's contiguous (m or n) extent
        constexpr int CPR = GT<T>::CPR, RPB = 64 / CPR;
        const int k = blk * RPB + lane / CPR, pc = lane % CPR;
        const int c = pc ^ ks_swz(k);
        const bool ok = (k < krem) && (row0 + c * EPC < nrows);
        return ok ? (int)(((long)k * ld + c * EPC) * sizeof(T)) : OOB;
    }
}

template <bool AKS, bool BKS, typename T>
__device__ __forceinline__ void dma_offsets(const GemmArgs& p, int m0, int n0, int krem, int wave, int lane,
                                            int (&va)[4], int (&vb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        va[i] = dma_voff<T, AKS>(p.lda, m0, p.M, krem, wave * 4 + i, lane);
        vb[i] = dma_voff<T, BKS>(p.ldb, n0, p.N, krem, wave * 4 + i, lane);
    }
}

__device__ __forceinline__ void dma_issue(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, const int (&va)[4],
                                          const int (&vb)[4], int sa, int sb, char* stage, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave * 4 + i) * 1024), 16, va[i], sa, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TILE_BYTES + (wave * 4 + i) * 1024), 16, vb[i], sb, 0, 0);
}

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(NTHR, 2) void gemm_dma_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages of [A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    tile_origin(p, m0, n0);
    constexpr int BK = GT<T>::BK;
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nk_full = (kend - kbeg) / BK;
    constexpr bool std_orient = AKS && BKS;
    constexpr int STAGE = 2 * TILE_BYTES;
    int oa[4], ob[4];
    ks_lane_offsets(wm, lane, oa);
    ks_lane_offsets(wn, lane, ob);

    // descriptors over the whole operands (bounds check = zero fill), scalar tile-origin offsets
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.bytesB, 0x00020000);
    int sa = (int)((AKS ? ((long)kbeg * p.lda + m0) : ((long)m0 * p.lda + kbeg)) * sizeof(T));
    int sb = (int)((BKS ? ((long)kbeg * p.ldb + n0) : ((long)n0 * p.ldb + kbeg)) * sizeof(T));
    const int stepa = (int)((AKS ? (long)BK * p.lda : (long)BK) * sizeof(T));
    const int stepb = (int)((BKS ? (long)BK * p.ldb : (long)BK) * sizeof(T));
    int va[4], vb[4];
    dma_offsets<AKS, BKS, T>(p, m0, n0, BK, wave, lane, va, vb);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // issue tile `t` into stage t&1 (full tiles: loop-invariant lane offsets; ragged last tile: recomputed)
    auto issue = [&](int t) {
        if (t < nk_full) {
            dma_issue(ra, rb, va, vb, sa, sb, smem + (t & 1) * STAGE, wave);
        } else if (t < nk) {
            int ta[4], tb[4];
            dma_offsets<AKS, BKS, T>(p, m0, n0, kend - kbeg - t * BK, wave, lane, ta, tb);
            dma_issue(ra, rb, ta, tb, sa, sb, smem + (t & 1) * STAGE, wave);
        }
        sa += stepa;
        sb += stepb;
    };
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my part of tile kt has landed
        __builtin_amdgcn_s_barrier();      // ... and everyone's; every wave is done reading tile kt-1
        issue(kt + 1);                     // overwrites the stage tile kt-1 lived in
        const char* tA = smem + (kt & 1) * STAGE;
        compute_tile<T, AKS, BKS, std_orient, true>(tA, tA + TILE_BYTES, acc, wm, wn, lane, oa, ob);
    }
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
    const bool tr = a.use_tr && sizeof(T) == 2;
    if (!transA && transB) hipLaunchKernelGGL((gemm_kernel<T, false, false, false>), grid, block, 0, stream, a);
    else if (!transA && !transB) {
        if (tr) hipLaunchKernelGGL((gemm_kernel<T, false, true, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((gemm_kernel<T, false, true, false>), grid, block, 0, stream, a);
    } else {
        if (tr) hipLaunchKernelGGL((gemm_kernel<T, true, true, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((gemm_kernel<T, true, true, false>), grid, block, 0, stream, a);
    }
    return MMTG_OK;
}

template <typename T>
int launch_dma(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
    dim3 block(NTHR);
    const size_t shm = (size_t)4 * TILE_BYTES;
    static bool attr_done[3] = {false, false, false};
    const int li = (!transA && transB) ? 0 : (!transA && !transB) ? 1 : 2;
    if (!attr_done[li]) {
        int rc = li == 0 ? set_lds(gemm_dma_kernel<T, false, false>, shm)
               : li == 1 ? set_lds(gemm_dma_kernel<T, false, true>, shm)
                         : set_lds(gemm_dma_kernel<T, true, true>, shm);
        if (rc) return rc;
        attr_done[li] = true;
    }
    if (li == 0) hipLaunchKernelGGL((gemm_dma_kernel<T, false, false>), grid, block, shm, stream, a);
    else if (li == 1) hipLaunchKernelGGL((gemm_dma_kernel<T, false, true>), grid, block, shm, stream, a);
    else hipLaunchKernelGGL((gemm_dma_kernel<T, true, true>), grid, block, shm, stream, a);
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
    // byte extents for the buffer descriptors of the LDS-DMA pipeline (offsets are 32-bit)
    const long esz = dtype == MMTG_F32 ? 4 : 2;
    const long bytesA = ((long)((transA ? K : M) - 1) * lda + (transA ? M : K)) * esz;
    const long bytesB = ((long)((transB ? N : K) - 1) * ldb + (transB ? K : N)) * esz;
    const bool small = bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L;
    a.bytesA = (int)(small ? bytesA : 0); a.bytesB = (int)(small ? bytesB : 0);
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
    else if ((flags & (MMTG_GEMM_REGSTAGE | MMTG_GEMM_NO_TR)) || !small) rc = launch_regstage<bf16>(a, transA, transB, grid, s);
    else rc = launch_dma<bf16>(a, transA, transB, grid, s);
    if (rc) return rc;
    MMTG_LAUNCH_CHECK("gemm");
    return MMTG_OK;
}
