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
    int split;
    tile_origin(p, m0, n0, split);
    const int kbeg = split * p.kper;
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
    // (the loop's final __syncthreads() already separates the last tile reads from this overlay)
    gemm_epilogue<T, std_orient, 4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, g, l15,
                                       smem + wave * epi_scratch_bytes<4, 4>(), lane);
}

// ------------------------------------------------------------------ LDS-DMA pipeline (bf16)
// Tile configuration: a BM x BN output tile per workgroup of WM x WN waves, each wave a
// (BM/WM) x (BN/WN) sub-tile of TM x TN MFMA 16x16 tiles.  Built configurations:
//   128x128, 2x2 waves (64x64 per wave)      training GEMMs, all three layouts, 2 workgroups/CU
//   256x32,  4x1 waves (64x32 per wave)      decode (M = batch <= 256): N/32 workgroups stream the
//                                            weights once each; NT layout only; 4-deep ring (144 KB)
//                                            because a lone workgroup per CU must hide the load
//                                            latency itself
// One wave-instruction moves 64 lanes x 16 B = 1 KB into LDS at (wave-uniform base + lane*16).
// Addressing is buffer-style: a wave-uniform descriptor over the whole operand, a per-lane byte
// offset that is computed ONCE (the lane's swizzled source chunk relative to the tile origin) and a
// scalar offset that advances by one K tile per iteration -> no vector address arithmetic in the
// loop.  Rows/columns outside the matrix get an out-of-range offset (the descriptor's bounds check
// returns zeros); only a ragged last K tile recomputes its offsets.
constexpr int OOB = 0x7FFFFFF0;

// lane offset of 1-KB block `blk` of an operand tile.  KC: R rows x 128 B.  KS: 64 k-rows x (W*2) B.
template <bool KS, int EXT>
__device__ __forceinline__ int dma_voff(long ld, int row0, int nrows, int krem, int blk, int lane) {
    constexpr int EPC = 8;
    if constexpr (!KS) {
        const int r = blk * 8 + (lane >> 3), pc = lane & 7;
        const int c = pc ^ (r & 7);
        const bool ok = (row0 + r < nrows) && (c * EPC < krem);
        return ok ? (int)(((long)r * ld + c * EPC) * 2) : OOB;
    } else {
        constexpr int CPR = EXT / 8, RPB = 64 / CPR;      // chunks per k-row, k-rows per 1-KB block
        static_assert(CPR >= 16, "K-strided tiles need >= 128 columns for the transposed-read swizzle");
        const int k = blk * RPB + lane / CPR, pc = lane % CPR;
        const int c = pc ^ ks_swz(k);
        const bool ok = (k < krem) && (row0 + c * EPC < nrows);
        return ok ? (int)(((long)k * ld + c * EPC) * 2) : OOB;
    }
}

// per-lane offsets of the transposed-read fragments of a K-strided tile of EXT columns (row bytes 2*EXT)
template <int EXT, int NT>
__device__ __forceinline__ void ks_offsets(int col0, int lane, int (&o)[NT]) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int k = 8 * g + q;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int chunk = ((col0 + i * 16) >> 3) + (pp >> 1);
        o[i] = k * (2 * EXT) + ((chunk ^ ks_swz(k)) << 4) + 8 * (pp & 1);
    }
}

template <bool AKS, bool BKS, int TBM, int TBN, int NBA, int NBB, int NW>
__device__ __forceinline__ void dma_issue_tile(const GemmArgs& p, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                               const int (&va)[NBA], const int (&vb)[NBB], int sa, int sb, char* stage, int TA,
                                               int t, int nk_full, int nk, int klen, int m0, int n0, int wave, int lane,
                                               bool dummy_tail) {
    if (t < nk_full) {          // full tile: loop-invariant lane offsets
#pragma unroll
        for (int i = 0; i < NBA; ++i)
            if (wave + NW * i < TBM / 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave + NW * i) * 1024), 16, va[i], sa, 0, 0);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
            if (wave + NW * i < TBN / 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TA + (wave + NW * i) * 1024), 16, vb[i], sb, 0, 0);
    } else if (t < nk) {        // ragged last K tile: recompute the offsets with the remaining K
        const int krem = klen - t * 64;
#pragma unroll
        for (int i = 0; i < NBA; ++i)
            if (wave + NW * i < TBM / 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave + NW * i) * 1024), 16,
                                                         dma_voff<AKS, TBM>(p.lda, m0, p.M, krem, wave + NW * i, lane), sa, 0, 0);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
            if (wave + NW * i < TBN / 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TA + (wave + NW * i) * 1024), 16,
                                                         dma_voff<BKS, TBN>(p.ldb, n0, p.N, krem, wave + NW * i, lane), sb, 0, 0);
    } else if (dummy_tail) {    // deep rings: keep the per-tile load count uniform for the counted vmcnt
#pragma unroll
        for (int i = 0; i < NBA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave + NW * i) * 1024), 16, OOB, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TA + (wave + NW * i) * 1024), 16, OOB, 0, 0, 0);
    }
}

template <int TBM, int TBN, int WM, int WN, int NBUF> struct DmaCfg {
    // two workgroups per CU when their LDS rings fit in 80 KB each
    static constexpr int MINW = ((TBM + TBN) * 128 * NBUF <= 80 * 1024 ? 2 : 1) * WM * WN / 4;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NBUF-deep LDS ring: tile t+NBUF-1 is issued right after the barrier of tile t; the wait before that
// barrier leaves the (NBUF-2) younger tiles in flight (counted vmcnt, raw s_barrier).
template <bool AKS, bool BKS, int TBM, int TBN, int WM, int WN, int NBUF>
__global__ __launch_bounds__((64 * WM * WN), (DmaCfg<TBM, TBN, WM, WN, NBUF>::MINW)) void gemm_dma_kernel(GemmArgs p) {
    typedef bf16 T;
    constexpr int NW = WM * WN, WTM = TBM / WM, WTN = TBN / WN, TM = WTM / 16, TN = WTN / 16;
    constexpr int TA = TBM * 128, TB = TBN * 128, STAGE = TA + TB;     // bytes
    // 1-KB DMA blocks per wave: block b of a tile is issued by wave b % NW (uneven splits allowed for
    // 2-deep rings, whose wait is vmcnt(0); deeper rings count loads per tile and need an even split)
    constexpr int NBA = (TBM / 8 + NW - 1) / NW, NBB = (TBN / 8 + NW - 1) / NW;
    static_assert(NBUF == 2 || (TBM % (8 * NW) == 0 && TBN % (8 * NW) == 0), "deep rings need an even block split");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // NBUF stages of [A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0;
    int split;
    tile_origin(p, m0, n0, split, TBM, TBN);
    constexpr int BK = 64;
    const int kbeg = split * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nk_full = (kend - kbeg) / BK;
    constexpr bool std_orient = AKS && BKS;

    // fragment addressing: KC rows via ld_frag_kc; KS via hoisted transposed-read offsets
    int oa[TM], ob[TN];
    if constexpr (AKS) ks_offsets<TBM, TM>(wm * WTM, lane, oa);
    if constexpr (BKS) ks_offsets<TBN, TN>(wn * WTN, lane, ob);

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.bytesB, 0x00020000);
    int sa = (int)((AKS ? ((long)kbeg * p.lda + m0) : ((long)m0 * p.lda + kbeg)) * 2);
    int sb = (int)((BKS ? ((long)kbeg * p.ldb + n0) : ((long)n0 * p.ldb + kbeg)) * 2);
    const int stepa = (int)((AKS ? (long)BK * p.lda : (long)BK) * 2);
    const int stepb = (int)((BKS ? (long)BK * p.ldb : (long)BK) * 2);
    int va[NBA], vb[NBB];
#pragma unroll
    for (int i = 0; i < NBA; ++i) va[i] = dma_voff<AKS, TBM>(p.lda, m0, p.M, BK, wave + NW * i, lane);
#pragma unroll
    for (int i = 0; i < NBB; ++i) vb[i] = dma_voff<BKS, TBN>(p.ldb, n0, p.N, BK, wave + NW * i, lane);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (a __device__ helper, not a lambda: a lambda in a __global__ body is host+device to clang and
    //  the amdgcn LDS-DMA builtin inside it silently drops the kernel's host stub)
#define ISSUE_TILE(t)                                                                                          \
    do {                                                                                                       \
        dma_issue_tile<AKS, BKS, TBM, TBN, NBA, NBB, NW>(p, ra, rb, va, vb, sa, sb, smem + ((t) % NBUF) * STAGE, TA, \
                                                     (t), nk_full, nk, kend - kbeg, m0, n0, wave, lane, NBUF > 2); \
        sa += stepa;                                                                                           \
        sb += stepb;                                                                                           \
    } while (0)
#pragma unroll
    for (int t0 = 0; t0 < NBUF - 1; ++t0) ISSUE_TILE(t0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(NBUF - 2) * (NBA + NBB)>();            // my part of tile kt has landed
        __builtin_amdgcn_s_barrier();      // ... and everyone's; every wave is done reading tile kt-1
        ISSUE_TILE(kt + NBUF - 1);         // overwrites the stage tile kt-1 lived in
        const char* tA = smem + (kt % NBUF) * STAGE;
        const char* tB = tA + TA;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (!AKS) fa[i] = ld_frag_kc<T>(tA, wm * WTM + i * 16 + l15, kk, g);
                else fa[i] = tr_read_pair(tA, oa[i] + kk * 32 * 2 * TBM, oa[i] + kk * 32 * 2 * TBM + 4 * 2 * TBM);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (!BKS) fb[j] = ld_frag_kc<T>(tB, wn * WTN + j * 16 + l15, kk, g);
                else fb[j] = tr_read_pair(tB, ob[j] + kk * 32 * 2 * TBN, ob[j] + kk * 32 * 2 * TBN + 4 * 2 * TBN);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (std_orient) mma16(fa[i], fb[j], acc[i][j]);
                    else mma16(fb[j], fa[i], acc[i][j]);
                }
        }
    }
#undef ISSUE_TILE
    if constexpr (NBUF > 2) wait_vmcnt<0>();   // drain the dummy tail loads before the LDS is released
    static_assert(NW * epi_scratch_bytes<TM, TN>() <= NBUF * STAGE, "epilogue scratch must fit the ring");
    if constexpr (!std_orient) __builtin_amdgcn_s_barrier();   // every wave is done reading the last tile
    gemm_epilogue<T, std_orient, TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, g, l15,
                                         smem + wave * epi_scratch_bytes<TM, TN>(), lane);
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

template <bool AKS, bool BKS, int BM_, int BN_, int WM, int WN, int NBUF>
int launch_dma_cfg(const GemmArgs& a, int splits, hipStream_t stream) {
    static bool attr_done = false;
    const size_t shm = (size_t)NBUF * (BM_ + BN_) * 128;
    if (!attr_done) {
        int rc = set_lds(gemm_dma_kernel<AKS, BKS, BM_, BN_, WM, WN, NBUF>, shm);
        if (rc) return rc;
        attr_done = true;
    }
    GemmArgs b = a;
    b.tiles_n = cdiv(a.N, BN_);
    b.ntiles = cdiv(a.M, BM_) * b.tiles_n;
    dim3 grid(b.ntiles * splits), block(64 * WM * WN);
    hipLaunchKernelGGL((gemm_dma_kernel<AKS, BKS, BM_, BN_, WM, WN, NBUF>), grid, block, shm, stream, b);
    return MMTG_OK;
}

int launch_dma(const GemmArgs& a, int transA, int transB, int splits, int skinny, int wide, hipStream_t stream) {
    if (wide) {   // 192x128 tiles, 3x2 waves: N = 768 products of M = 15104 fit one round of 2 workgroups/CU
        if (!transA && transB) return launch_dma_cfg<false, false, 192, 128, 3, 2, 2>(a, splits, stream);
        if (!transA && !transB) return launch_dma_cfg<false, true, 192, 128, 3, 2, 2>(a, splits, stream);
    }
    if (!transA && transB) {
        if (skinny) return launch_dma_cfg<false, false, 256, 32, 4, 1, 4>(a, splits, stream);
        return launch_dma_cfg<false, false, 128, 128, 2, 2, 2>(a, splits, stream);
    }
    if (!transA && !transB) return launch_dma_cfg<false, true, 128, 128, 2, 2, 2>(a, splits, stream);
    return launch_dma_cfg<true, true, 128, 128, 2, 2, 2>(a, splits, stream);
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
        MMTG_REQUIRE(N % 8 == 0 && ldc % 8 == 0, "gemm: N and ldc must be multiples of 8 (N=%d ldc=%ld)", N, ldc);
        MMTG_REQUIRE(!bias || MMTG_ALIGNED16(bias), "gemm: bias must be 16-byte aligned");
        MMTG_REQUIRE(splits <= 1, "gemm: split-K needs the atomic epilogue");
    } else {
        MMTG_REQUIRE(!bias, "gemm: atomic epilogue takes no bias");
    }
    if (epi == MMTG_EPI_ROWDOT)
        MMTG_REQUIRE(aux2 && !bias && N % 64 == 0 && dtype == MMTG_BF16 && !transA && M > 256 && !(flags & (MMTG_GEMM_REGSTAGE | MMTG_GEMM_NO_TR | MMTG_GEMM_SKINNY)),
                     "gemm: ROWDOT epilogue needs aux2 (f32 [M, N/64]), no bias, N %% 64 == 0, the bf16 128x128 LDS-DMA configuration");
    if (epi == MMTG_EPI_RESID || epi == MMTG_EPI_DGELU || epi == MMTG_EPI_DTANH || epi == MMTG_EPI_ROWDOT)
        MMTG_REQUIRE(aux && ldaux % 8 == 0 && MMTG_ALIGNED16(aux), "gemm: epilogue %d needs a 16-byte aligned aux operand with ldaux %% 8 == 0", epi);
    if (epi == MMTG_EPI_GELU) MMTG_REQUIRE(aux2 && MMTG_ALIGNED16(aux2), "gemm: GELU epilogue needs a 16-byte aligned aux2 for the pre-activation");
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
    a.ntiles = cdiv(M, BM) * cdiv(N, BN);
    dim3 grid(a.ntiles * splits);
    // small-M products with K-contiguous weights (decode): 256x32 tiles -> N/32 workgroups
    const int skinny = (flags & MMTG_GEMM_SKINNY) || (!transA && transB && M <= 256 && !(flags & MMTG_GEMM_NO_SKINNY));
    int rc;
    if (dtype == MMTG_F32) rc = launch_regstage<float>(a, transA, transB, grid, s);
    else if ((flags & (MMTG_GEMM_REGSTAGE | MMTG_GEMM_NO_TR)) || !small) rc = launch_regstage<bf16>(a, transA, transB, grid, s);
    else {
        // Tile choice (measured, profiles/r01_gemm_tile_configs.log): 192x128 when it turns a
        // 1.x-round grid of 128x128 tiles into one full round of the 512 workgroup slots (N = 768 at
        // M = 15104: 708 -> 474 tiles, -27 %), and for very wide outputs (LM head, -12 %).
        bool wide = (flags & MMTG_GEMM_WIDE) != 0;
        if (!wide && !(flags & MMTG_GEMM_NO_WIDE) && !transA && !skinny && M >= 1024) {
            const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t192 = (long)cdiv(M, 192) * cdiv(N, 128);
            wide = (t128 > 512 && t192 <= 512) || N >= 4096;
        }
        rc = launch_dma(a, transA, transB, splits, skinny && !transA && transB, wide, s);
    }
    if (rc) return rc;
    MMTG_LAUNCH_CHECK("gemm");
    return MMTG_OK;
}
