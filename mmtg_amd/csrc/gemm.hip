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
#include <stdlib.h>

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
    tile_origin(p, blockIdx.x, gridDim.x, m0, n0, split);
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
    p.C = reinterpret_cast<char*>(p.C) + (long)split * p.split_stride;     // MMTG_EPI_SPLIT slab (0 otherwise), every layout
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
// (TAG only makes the instance unique per calling kernel: a second kernel that instantiates this helper with the same
//  arguments fails hipcc's host pass with "no matching function" -- see DESIGN.md 4b)
template <bool AKS, bool BKS, int TBM, int TBN, int NBA, int NBB, int NW, int TAG>
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

// full K tile with precomputed lane offsets (the gathered-row forward product)
template <int TBM, int TBN, int NBA, int NBB, int NW>
__device__ __forceinline__ void dma_issue_full(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, const int (&va)[NBA], const int (&vb)[NBB],
                                               int sa, int sb, char* stage, int TA, bool live, int wave) {
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NBA; ++i)
        if (wave + NW * i < TBM / 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave + NW * i) * 1024), 16, va[i], sa, 0, 0);
#pragma unroll
    for (int i = 0; i < NBB; ++i)
        if (wave + NW * i < TBN / 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TA + (wave + NW * i) * 1024), 16, vb[i], sb, 0, 0);
}

template <int TBM, int TBN, int WM, int WN, int NBUF> struct DmaCfg {
    // two workgroups per CU when their LDS rings fit in 80 KB each.  Waves per SIMD the register
    // budget must allow: the 6 waves of a 3x2 workgroup land 2,2,1,1 on the four SIMDs and the second
    // workgroup is not placed complementarily, so two of them need FOUR waves on a SIMD (<= 128 VGPRs);
    // at 129+ VGPRs the 192x128 kernel silently runs one workgroup per CU (measured: 85 -> 140 us).
    static constexpr int WGS = (TBM + TBN) * 128 * NBUF <= 80 * 1024 ? 2 : 1;
    static constexpr int MINW = WGS * ((WM * WN + 3) / 4);
};


// NBUF-deep LDS ring: tile t+NBUF-1 is issued right after the barrier of tile t; the wait before that
// barrier leaves the (NBUF-2) younger tiles in flight (counted vmcnt, raw s_barrier).
template <bool AKS, bool BKS, int TBM, int TBN, int WM, int WN, int NBUF, bool GATHER = false>
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
    // diagnostic timeline (mmtg_gemm_trace); not in the 6-wave configuration, which has no register to spare
    const bool tr = WM * WN == 4 && NBUF == 2 && p.trace != nullptr;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (tr) ts0 = __builtin_amdgcn_s_memrealtime();
    int m0, n0;
    int split;
    tile_origin(p, blockIdx.x, gridDim.x, m0, n0, split, TBM, TBN);
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
    if constexpr (GATHER) {
        // mmtg_gemm_gather mode 0: output row m reads table row gather[m] -- the row index enters the lane's SOURCE offset
        // (K-contiguous A image: 1-KB block = 8 rows x 128 B, chunk swizzle as in dma_voff); K % 64 == 0, so the offsets
        // are loop-invariant and only the scalar K offset advances
        static_assert(!AKS, "gathered A rows are K-contiguous");
        sa = kbeg * 2;
#pragma unroll
        for (int i = 0; i < NBA; ++i) {
            const int r = (wave + NW * i) * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
            va[i] = (wave + NW * i < TBM / 8 && m0 + r < p.M) ? (int)(((long)p.gather[m0 + r] * p.lda + c * 8) * 2) : OOB;
        }
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (a __device__ helper, not a lambda: a lambda in a __global__ body is host+device to clang and
    //  the amdgcn LDS-DMA builtin inside it silently drops the kernel's host stub)
    // (the gather instantiation has its own issue helper: a second kernel instantiating dma_issue_tile with the same
    //  arguments trips hipcc's host pass; its K is a multiple of 64, so every tile is a full one)
#define ISSUE_TILE(t)                                                                                          \
    do {                                                                                                       \
        if constexpr (GATHER)                                                                                  \
            dma_issue_full<TBM, TBN, NBA, NBB, NW>(ra, rb, va, vb, sa, sb, smem + ((t) % NBUF) * STAGE, TA, (t) < nk, wave); \
        else                                                                                                   \
            dma_issue_tile<AKS, BKS, TBM, TBN, NBA, NBB, NW, NBUF>(p, ra, rb, va, vb, sa, sb, smem + ((t) % NBUF) * STAGE, TA, \
                                                         (t), nk_full, nk, kend - kbeg, m0, n0, wave, lane, NBUF > 2); \
        sa += stepa;                                                                                           \
        sb += stepb;                                                                                           \
    } while (0)
#pragma unroll
    for (int t0 = 0; t0 < NBUF - 1; ++t0) ISSUE_TILE(t0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(NBUF - 2) * (NBA + NBB)>();            // my part of tile kt has landed
        __builtin_amdgcn_s_barrier();      // ... and everyone's; every wave is done reading tile kt-1
        if (tr && kt == 0) ts1 = __builtin_amdgcn_s_memrealtime();
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
    if (tr) ts2 = __builtin_amdgcn_s_memrealtime();
    if constexpr (NBUF > 2) wait_vmcnt<0>();   // drain the dummy tail loads before the LDS is released
    static_assert(NW * epi_scratch_bytes<TM, TN>() <= NBUF * STAGE, "epilogue scratch must fit the ring");
    if constexpr (!std_orient) __builtin_amdgcn_s_barrier();   // every wave is done reading the last tile
    // MMTG_EPI_SPLIT: K split s stores its raw partial product in slab s of the fp32 output
    if constexpr (!std_orient) p.C = reinterpret_cast<char*>(p.C) + (long)split * p.split_stride;
    // (the 6-wave configuration has to stay within 128 VGPRs: aux vectors one band ahead, not all four)
    gemm_epilogue<T, std_orient, TM, TN, (WM * WN > 4 ? -1 : (TM == 4 ? TM : -TM))>(p, acc, m0 + wm * WTM, n0 + wn * WTN, g, l15,
                                         smem + wave * epi_scratch_bytes<TM, TN>(), lane);
    if (tr && wave == 0 && (int)blockIdx.x < p.trace_n) {
        wait_vmcnt<0>();                       // the output stores are part of the epilogue's time
        const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* r = p.trace + 6 * (size_t)blockIdx.x;
            r[0] = ts0; r[1] = ts1; r[2] = ts2; r[3] = ts3; r[4] = (unsigned long long)nk;
            r[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
}

// ------------------------------------------------------------------ persistent, pipelined LDS-DMA kernel
// One resident workgroup per CU slot (2 per CU) walks work items (output tile x K split) and never
// drains its DMA ring between them: the last K tile of item i issues K tile 0 of item i+1, and item
// i's epilogue (LDS-staged vector stores / atomics) runs from a register copy of its accumulators
// during K tile 0 of item i+1 -- after that step's DMA issue, before its MFMAs -- so the stores have
// a whole K tile of MFMA work to drain before the next vmcnt(0).  What this hides: the per-item
// first-load latency and store drain, ~6 us of an 18 us 128x128x768 item (K sweep in profiles/).
// The epilogue scratch (4 KB per wave) sits behind the ring: 2 x 32 KB + 16 KB = 80 KB, two per CU.
template <int NBA, int NBB> struct DmaItem {
    int m0, n0, klen, nk, nk_full;
    int sa, sb;                 // scalar byte offsets of the next K tile to issue
    int va[NBA], vb[NBB];       // lane byte offsets (swizzled source chunk, or OOB)
};

template <bool AKS, bool BKS, int TBM, int TBN, int NBA, int NBB, int NW>
__device__ __forceinline__ void item_setup(const GemmArgs& p, int item, int wave, int lane, DmaItem<NBA, NBB>& s) {
    int split;
    tile_origin(p, item, p.nitems, s.m0, s.n0, split, TBM, TBN);
    const int kbeg = split * p.kper;
    s.klen = min(p.K, kbeg + p.kper) - kbeg;
    s.nk = (s.klen + 63) >> 6;
    s.nk_full = s.klen >> 6;
    s.sa = (int)((AKS ? ((long)kbeg * p.lda + s.m0) : ((long)s.m0 * p.lda + kbeg)) * 2);
    s.sb = (int)((BKS ? ((long)kbeg * p.ldb + s.n0) : ((long)s.n0 * p.ldb + kbeg)) * 2);
#pragma unroll
    for (int i = 0; i < NBA; ++i) s.va[i] = dma_voff<AKS, TBM>(p.lda, s.m0, p.M, 64, wave + NW * i, lane);
#pragma unroll
    for (int i = 0; i < NBB; ++i) s.vb[i] = dma_voff<BKS, TBN>(p.ldb, s.n0, p.N, 64, wave + NW * i, lane);
}

// K tile `t` of item `s` -> LDS stage (same block/wave split as dma_issue_tile)
template <bool AKS, bool BKS, int TBM, int TBN, int NBA, int NBB, int NW>
__device__ __forceinline__ void dma_issue_item(const GemmArgs& p, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                               const DmaItem<NBA, NBB>& s, char* stage, int t, int wave, int lane) {
    constexpr int TA = TBM * 128;
    if (t < s.nk_full) {
#pragma unroll
        for (int i = 0; i < NBA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave + NW * i) * 1024), 16, s.va[i], s.sa, 0, 0);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TA + (wave + NW * i) * 1024), 16, s.vb[i], s.sb, 0, 0);
    } else {                    // ragged last K tile: lane offsets with the remaining K
        const int krem = s.klen - t * 64;
#pragma unroll
        for (int i = 0; i < NBA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (wave + NW * i) * 1024), 16,
                                                     dma_voff<AKS, TBM>(p.lda, s.m0, p.M, krem, wave + NW * i, lane), s.sa, 0, 0);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + TA + (wave + NW * i) * 1024), 16,
                                                     dma_voff<BKS, TBN>(p.ldb, s.n0, p.N, krem, wave + NW * i, lane), s.sb, 0, 0);
    }
}

// the MFMAs of one staged K tile (two 32-wide k blocks) of a wave's TM x TN sub-tile
template <bool AKS, bool BKS, int TBM, int TBN, int TM, int TN>
__device__ __forceinline__ void mma_stage(const char* tA, const char* tB, const int (&oa)[TM], const int (&ob)[TN],
                                          int arow0, int brow0, int g, int l15, f32x4 (&acc)[TM][TN]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (!AKS) fa[i] = ld_frag_kc<bf16>(tA, arow0 + i * 16 + l15, kk, g);
            else fa[i] = tr_read_pair(tA, oa[i] + kk * 32 * 2 * TBM, oa[i] + kk * 32 * 2 * TBM + 4 * 2 * TBM);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (!BKS) fb[j] = ld_frag_kc<bf16>(tB, brow0 + j * 16 + l15, kk, g);
            else fb[j] = tr_read_pair(tB, ob[j] + kk * 32 * 2 * TBN, ob[j] + kk * 32 * 2 * TBN + 4 * 2 * TBN);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (AKS && BKS) mma16(fa[i], fb[j], acc[i][j]);
                else mma16(fb[j], fa[i], acc[i][j]);
            }
    }
}

// TBM = 128: 4 waves, 80 KB, two workgroups per CU.  TBM = 256 (round 2): 8 waves (4 x 2 of 64x64), two 48 KB stages + 32 KB of
// epilogue scratch = 128 KB, ONE workgroup per CU whose eight waves are in the K loop together (two MFMA-issuing waves per
// SIMD), a quarter fewer fill bytes per FLOP than 128x128 tiles.
template <bool AKS, bool BKS, int TBM = 128>
__global__ __launch_bounds__(TBM * 2, 2) void gemm_pp_kernel(GemmArgs p) {
    typedef bf16 T;
    constexpr int TBN = 128, NW = TBM / 32, TM = 4, TN = 4, NBA = TBM / 8 / NW, NBB = TBN / 8 / NW;
    constexpr int TA = TBM * 128, STAGE = (TBM + TBN) * 128;
    constexpr bool std_orient = AKS && BKS;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 stages [A | B], then NW epilogue images
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    char* const scratch = smem + 2 * STAGE + wave * epi_scratch_bytes<TM, TN>();
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.bytesB, 0x00020000);
    const int stepa = (int)((AKS ? (long)64 * p.lda : (long)64) * 2);
    const int stepb = (int)((BKS ? (long)64 * p.ldb : (long)64) * 2);
    int oa[TM], ob[TN];
    if constexpr (AKS) ks_offsets<TBM, TM>(wm * 64, lane, oa);
    if constexpr (BKS) ks_offsets<TBN, TN>(wn * 64, lane, ob);

#define PP_ISSUE(s, t, ring_)                                                                                      \
    do {                                                                                                           \
        dma_issue_item<AKS, BKS, TBM, TBN, NBA, NBB, NW>(p, ra, rb, (s), smem + (ring_) * STAGE, (t), wave, lane); \
        (s).sa += stepa;                                                                                           \
        (s).sb += stepb;                                                                                           \
    } while (0)

    // `it` describes the item whose K tiles are being issued; it is re-initialised for the next item
    // when the last K tile of the current one has been issued (its origin is kept in xm0/xn0).
#define PP_NEXT(tn)                                                              \
    do {                                                                         \
        if ((tn) < nk) PP_ISSUE(it, (tn), ring ^ 1);                             \
        else if (has_next) {                                                     \
            item += (int)gridDim.x;                                              \
            item_setup<AKS, BKS, TBM, TBN, NBA, NBB, NW>(p, item, wave, lane, it); \
            PP_ISSUE(it, 0, ring ^ 1);                                           \
        }                                                                        \
    } while (0)
#define PP_MMA() \
    mma_stage<AKS, BKS, TBM, TBN, TM, TN>(smem + ring * STAGE, smem + ring * STAGE + TA, oa, ob, wm * 64, wn * 64, g, l15, acc)

    DmaItem<NBA, NBB> it;
    int item = blockIdx.x;
    item_setup<AKS, BKS, TBM, TBN, NBA, NBB, NW>(p, item, wave, lane, it);
    int ring = 0;
    PP_ISSUE(it, 0, 0);
    f32x4 acc[TM][TN];
    int pm0 = 0, pn0 = 0;
    bool pending = false, have = true;
    for (;;) {
        int nk = 0, xm0 = 0, xn0 = 0;
        bool has_next = false;
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
        const int rec = item;
        if (p.trace) ts0 = __builtin_amdgcn_s_memrealtime();
        if (have) {
            nk = it.nk;
            xm0 = it.m0;
            xn0 = it.n0;
            has_next = item + (int)gridDim.x < p.nitems;
            wait_vmcnt<0>();                   // my share of K tile 0 has landed (and my last stores are out)
            __builtin_amdgcn_s_barrier();      // ... everyone's; every wave is done with the other stage
            PP_NEXT(1);
        }
        if (p.trace) ts1 = __builtin_amdgcn_s_memrealtime();
        // the finished item's epilogue: its stores drain under the MFMAs of this item's K tile 0
        if (pending) gemm_epilogue<T, std_orient, TM, TN>(p, acc, pm0, pn0, g, l15, scratch, lane);
        if (!have) break;
        if (p.trace) ts2 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        PP_MMA();
        ring ^= 1;
        for (int kt = 1; kt + 1 < nk; ++kt) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            PP_ISSUE(it, kt + 1, ring ^ 1);
            PP_MMA();
            ring ^= 1;
        }
        if (nk > 1) {                          // last K tile: the next item's K tile 0 is issued under it
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            PP_NEXT(nk);
            PP_MMA();
            ring ^= 1;
        }
        pm0 = xm0 + wm * 64;
        pn0 = xn0 + wn * 64;
        pending = true;
        have = has_next;
        if (p.trace && wave == 0 && lane == 0 && rec < p.trace_n) {
            // per ITEM: top of the step, K tile 0 landed + next issue done, deferred epilogue issued, K loop done
            unsigned long long* r = p.trace + 6 * (size_t)rec;
            r[0] = ts0; r[1] = ts1; r[2] = ts2; r[3] = __builtin_amdgcn_s_memrealtime(); r[4] = (unsigned long long)nk;
            r[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
#undef PP_NEXT
#undef PP_MMA
#undef PP_ISSUE
}

// ------------------------------------------------------------------ single-stage, 4-workgroups-per-CU kernel
// One 32 KB LDS stage per workgroup and <= 128 VGPRs: four 128x128 workgroups share a CU (four waves
// per SIMD).  A workgroup does not overlap its own DMA with its MFMAs at all -- the other three do.
// (Bare-MFMA probe: the matrix pipe needs several waves per SIMD with MFMAs ready; two per SIMD that
// each alternate 16 fragment reads / 32 MFMAs leave it at ~32 cycles per MFMA.)
// (waves per SIMD the forward / dgrad instantiations are compiled for: their staged epilogue with
//  prefetched aux vectors and the dGELU column sums does not fit 128 VGPRs without spilling)
#ifndef OCC_FWD
#define OCC_FWD 3
#endif
// SLAB (weight gradients only): instead of fp32 atomics, K split s stores its partial tile with plain
// 16-byte stores into slab s of an fp32 workspace (MMTG_EPI_SPLIT); mmtg_slab_sum adds the slabs up.
template <bool AKS, bool BKS, bool SLAB = false, bool GATHER = false>
__global__ __launch_bounds__(256, (AKS && BKS ? 4 : OCC_FWD)) void gemm_occ4_kernel(GemmArgs p) {
    typedef bf16 T;
    constexpr int TBM = 128, TBN = 128, NW = 4, TM = 4, TN = 4, BK = 64;
    constexpr int NB = TBM / 8 / NW, NBB = TBN / 8 / NW;         // 1-KB DMA blocks per wave: A, B
    constexpr int TA = TBM * 128;
    constexpr bool std_orient = AKS && BKS && !SLAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    // (the prologue's kernel arguments in one batch of scalar loads -- see gemm_p8_kernel)
    asm volatile("" :: "s"(p.A), "s"(p.B), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.lda), "s"(p.ldb), "s"(p.ntiles), "s"(p.tiles_n),
                 "s"(p.tiles_m_fast), "s"(p.cbw), "s"(p.bytesA), "s"(p.bytesB), "s"(p.kper));
    int m0, n0, split;
    tile_origin(p, blockIdx.x, gridDim.x, m0, n0, split, TBM, TBN);
    const int kbeg = split * p.kper;
    const int klen = min(p.K, kbeg + p.kper) - kbeg;
    const int nk = (klen + BK - 1) / BK, nk_full = klen / BK;
    int oa[TM], ob[TN];
    if constexpr (AKS) ks_offsets<TBM, TM>(wm * 64, lane, oa);
    if constexpr (BKS) ks_offsets<TBN, TN>(wn * 64, lane, ob);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.bytesB, 0x00020000);
    int sa = (int)((AKS ? ((long)kbeg * p.lda + m0) : ((long)m0 * p.lda + kbeg)) * 2);
    int sb = (int)((BKS ? ((long)kbeg * p.ldb + n0) : ((long)n0 * p.ldb + kbeg)) * 2);
    const int stepa = (int)((AKS ? (long)BK * p.lda : (long)BK) * 2);
    const int stepb = (int)((BKS ? (long)BK * p.ldb : (long)BK) * 2);
    int va[NB], vb[NBB];
#pragma unroll
    for (int i = 0; i < NB; ++i) va[i] = dma_voff<AKS, TBM>(p.lda, m0, p.M, BK, wave + NW * i, lane);
#pragma unroll
    for (int i = 0; i < NBB; ++i) vb[i] = dma_voff<BKS, TBN>(p.ldb, n0, p.N, BK, wave + NW * i, lane);
    // mmtg_gemm_gather mode 1: reduction index k reads table row gather[k] (K-strided B image: 1-KB block = 4 k-rows x 256 B);
    // the four k-rows this lane fetches per K tile are looked up one tile ahead
    int gid[GATHER ? NBB : 1];
    if constexpr (GATHER) {
        static_assert(BKS, "gathered B rows are the reduction index");
        sb = 0;
#pragma unroll
        for (int i = 0; i < NBB; ++i) {
            const int k = (wave + NW * i) * 4 + (lane >> 4);
            gid[i] = (k < klen) ? p.gather[kbeg + k] : 0;
        }
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        if (kt) __builtin_amdgcn_s_barrier();      // every wave is done reading the previous tile
        {   // (inline, not dma_issue_tile: a second kernel instantiating that helper with the same
            //  arguments trips hipcc's host pass)
            const bool full = kt < nk_full;
            const int krem = klen - kt * BK;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int v = full ? va[i] : dma_voff<AKS, TBM>(p.lda, m0, p.M, krem, wave + NW * i, lane);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, smem + (wave + NW * i) * 1024), 16, v, sa, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NBB; ++i) {
                int v;
                if constexpr (GATHER) {
                    const int k = (wave + NW * i) * 4 + (lane >> 4), pc = lane & 15, c = pc ^ ks_swz(k);
                    v = (k < krem && n0 + c * 8 < p.N) ? (int)(((long)gid[i] * p.ldb + n0 + c * 8) * 2) : OOB;
                } else {
                    v = full ? vb[i] : dma_voff<BKS, TBN>(p.ldb, n0, p.N, krem, wave + NW * i, lane);
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, smem + TA + (wave + NW * i) * 1024), 16, v, sb, 0, 0);
            }
            if constexpr (GATHER) {
#pragma unroll
                for (int i = 0; i < NBB; ++i) {
                    const int k = (kt + 1) * BK + (wave + NW * i) * 4 + (lane >> 4);
                    gid[i] = (k < klen) ? p.gather[kbeg + k] : 0;
                }
            }
        }
        sa += stepa;
        if constexpr (!GATHER) sb += stepb;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // the tile is complete
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (!AKS) fa[i] = ld_frag_kc<T>(smem, wm * 64 + i * 16 + l15, kk, g);
                else fa[i] = tr_read_pair(smem, oa[i] + kk * 32 * 2 * TBM, oa[i] + kk * 32 * 2 * TBM + 4 * 2 * TBM);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (!BKS) fb[j] = ld_frag_kc<T>(smem + TA, wn * 64 + j * 16 + l15, kk, g);
                else fb[j] = tr_read_pair(smem + TA, ob[j] + kk * 32 * 2 * TBN, ob[j] + kk * 32 * 2 * TBN + 4 * 2 * TBN);
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
    if constexpr (!std_orient) {
        __builtin_amdgcn_s_barrier();
        p.C = reinterpret_cast<char*>(p.C) + (long)split * p.split_stride;     // MMTG_EPI_SPLIT slab
    }
    gemm_epilogue<T, std_orient, TM, TN, 1>(p, acc, m0 + wm * 64, n0 + wn * 64, g, l15, smem + wave * epi_scratch_bytes<TM, TN>(), lane);
}

// ------------------------------------------------------------------ launch
int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// workgroups of `kern` (with `bytes` of dynamic LDS) that fit one CU, as the runtime sees it
template <typename K> int resident_wgs(K kern, int threads, size_t bytes) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, threads, bytes) != hipSuccess) n = 0;
    return n;
}

// Column-block width of the item order of a forward / dgrad product (0 = plain tile_n-fastest order).
// The B operand (weights, N x K) is re-read from the Infinity Cache by every XCD once per sweep over its
// tile columns when it does not stay in the 4 MB L2 (measured: LM head 1.6 GB of fabric reads for 43 MB of
// operands; L2-miss fills run at 13 B/clk/CU against 52 for hits, tools/micro/lds_fill.py).  Blocks of
// `w` tile columns keep w B panels resident while the A rows stream by; A is then read once per block.
int column_block(const GemmArgs& a, int bm, int bn, int slots) {
    static const long budget = getenv("MMTG_GEMM_CB_KB") ? atol(getenv("MMTG_GEMM_CB_KB")) * 1024 : 2048L * 1024;
    const long tiles_m = cdiv(a.M, bm), tiles_n = cdiv(a.N, bn);
    if (a.dbg_flags & 16) return tiles_n > 2 ? 2 : 0;      // MMTG_GEMM_COL_BLOCK: test hook
    if (budget <= 0) return 0;
    const long panel = (long)bn * a.K * 2, bytesB = (long)a.N * a.K * 2, bytesA = (long)a.M * a.K * 2;
    if (bytesB <= budget || tiles_n < 2) return 0;
    const double passes = (double)(tiles_m * tiles_n) / slots;            // sweeps of an XCD over all columns
    if (passes <= 1.0) return 0;
    const long wmax = budget / panel;
    if (wmax < 1) return 0;
    const long nblocks = cdiv(tiles_n, wmax), w = cdiv(tiles_n, nblocks);
    const double plain = (double)bytesA + 8.0 * bytesB * passes;
    const double xcds = 8.0 * w / tiles_n;
    const double blocked = (double)bytesA * nblocks + (double)bytesB * (xcds < 1.0 ? 1.0 : xcds);
    return blocked < 0.8 * plain ? (int)w : 0;
}

template <typename K> int set_lds(K kern, size_t bytes, int threads = 0, const char* what = "") {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        MMTG_FAIL(MMTG_ERR_HIP, "gemm: cannot raise dynamic LDS to %zu bytes", bytes);
    if (threads && getenv("MMTG_GEMM_DEBUG"))
        fprintf(stderr, "[mmtg gemm] %s: %d threads, %zu B LDS -> %d workgroups/CU (%d CUs)\n", what, threads, bytes,
                resident_wgs(kern, threads, bytes), num_cus());
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
        char what[64];
        snprintf(what, sizeof what, "dma %dx%d ring %d %s%s", BM_, BN_, NBUF, AKS ? "T" : "N", BKS ? "N" : "T");
        int rc = set_lds(gemm_dma_kernel<AKS, BKS, BM_, BN_, WM, WN, NBUF>, shm, 64 * WM * WN, what);
        if (rc) return rc;
        attr_done = true;
    }
    GemmArgs b = a;
    b.tiles_n = cdiv(a.N, BN_);
    b.ntiles = cdiv(a.M, BM_) * b.tiles_n;
    b.nitems = b.ntiles * splits;
    // Weight gradients (both operands are K slices of activations, fetched from HBM): order the items
    // so that the run of consecutive items an XCD gets is the squarer block of the tile grid -- with
    // 6 x 24 tiles and 54 items per XCD, 6 rows x 9 columns (15 operand panels) instead of 2.25 rows x
    // 24 columns (27 panels).
    b.tiles_m_fast = AKS && BKS && b.tiles_n > cdiv(a.M, BM_) && !(a.dbg_flags & 1);
    if (!(AKS && BKS) && splits == 1 && !(a.dbg_flags & 1)) b.cbw = column_block(a, BM_, BN_, 2 * num_cus());
    dim3 grid(b.nitems), block(64 * WM * WN);
    hipLaunchKernelGGL((gemm_dma_kernel<AKS, BKS, BM_, BN_, WM, WN, NBUF>), grid, block, shm, stream, b);
    return MMTG_OK;
}

// persistent pipelined 128x128 kernel: one workgroup per CU slot (a multiple of 8, so that the
// item stride keeps every item on the XCD the tile order assumes)
template <bool AKS, bool BKS, int TBM = 128>
int launch_pp(const GemmArgs& a, int splits, hipStream_t stream) {
    static bool attr_done = false;
    constexpr int NW = TBM / 32;
    const size_t shm = 2 * (TBM + 128) * 128 + NW * epi_scratch_bytes<4, 4>();
    static_assert(TBM == 256 || 2 * (128 + 128) * 128 + 4 * epi_scratch_bytes<4, 4>() <= 80 * 1024, "two workgroups per CU");
    if (!attr_done) {
        int rc = set_lds(gemm_pp_kernel<AKS, BKS, TBM>, shm, 64 * NW, TBM == 256 ? "persistent 256x128" : "persistent 128x128");
        if (rc) return rc;
        attr_done = true;
    }
    GemmArgs b = a;
    b.tiles_n = cdiv(a.N, 128);
    b.ntiles = cdiv(a.M, TBM) * b.tiles_n;
    b.nitems = b.ntiles * splits;
    if (TBM == 256 && !(AKS && BKS) && splits == 1 && !(a.dbg_flags & 1)) b.cbw = column_block(a, TBM, 128, num_cus());
    const int slots = ((TBM == 256 ? 1 : 2) * num_cus()) & ~7;
    dim3 grid(b.nitems < slots ? b.nitems : slots), block(64 * NW);
    hipLaunchKernelGGL((gemm_pp_kernel<AKS, BKS, TBM>), grid, block, shm, stream, b);
    return MMTG_OK;
}

// ------------------------------------------------------------------ 256x256 eight-phase kernel (NT, bf16; round 2)
// One 512-thread workgroup per CU computes a 256x256 tile; wave (wr, wc) of the 2x4 grid owns 128x64 of it as four
// 64x32 quadrants (a0|a1) x (b0|b1).  A K tile (64 deep) is consumed in FOUR phases -- (a0,b0) (a0,b1) (a1,b1) (a1,b0),
// 16 MFMAs each, fragments read at the start of the phase that first needs them (12 / 4 / 8 / 0 ds_read_b128) -- and the two
// wave groups wr = 0 / 1 run exactly one barrier apart, so between two barriers one wave of every SIMD issues its 16 MFMAs
// while the other reads its fragments and issues its share of the LDS-DMA prefetch: fills, fragment reads and MFMAs
// overlap by construction instead of by luck of the wave scheduler.  Staging: two 64 KB stages (tiles kt, kt+1), each
// K tile as four 16 KB "half tiles" named by the phase that retires them (B-b0, A-a0, B-b1, A-a1); every phase restages
// ONE half tile (two 1-KB DMA blocks per wave) as soon as its last reader has passed a barrier, three half tiles ahead
// of the tile being consumed; one counted vmcnt(6) per K tile (phase 4) retires the next tile, never vmcnt(0) in the loop.
// (structure after the guide's 256^2 8-phase template; LDS images, swizzle, descriptors and epilogue are this file's own)
// (the scalar offset is wave-uniform by construction; said explicitly -- readfirstlane -- because hipcc otherwise wraps each request of
//  the K-strided form, whose offset goes through a runtime stride, in a waterfall loop: ~10 extra instructions + a branch per
//  request inside the load wave's phase, found in round 4 with per-phase stamps, profiles/r04_v7_p8_phase_stamps_nt_vs_k_strided.txt)
__device__ __forceinline__ void p8_issue(__amdgpu_buffer_rsrc_t r, char* d0, char* d1, int v0, int v1, int soff) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, d0), 16, v0, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, d1), 16, v1, soff, 0, 0);
}
__device__ __forceinline__ void p8_issue1(__amdgpu_buffer_rsrc_t r, char* d0, int v0, int soff) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, d0), 16, v0, soff, 0, 0);
}

template <int AH, int BH, int NA, int TMW, int FAN>
__device__ __forceinline__ void p8_mfma(const bf16x8 (&fa)[FAN][2], const bf16x8 (&fb)[2][2], f32x4 (&acc)[TMW][4], const bool prio = true) {
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma16(fb[j][kk], fa[i][kk], acc[4 * AH + i][2 * BH + j]);
    if (prio) __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ bf16x8 p8_ld(const char* q) { return *reinterpret_cast<const bf16x8*>(q); }

// phase boundaries (asm with a memory clobber: the compiler may move neither LDS reads nor DMA issues across a barrier -- the
// RAW / WAR argument counts barriers in program order; sched_barrier keeps the MFMA clusters inside their phase)
#define P8_SYNC_IN()  do { asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P8_SYNC_OUT() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); } while (0)

// TBM = 256: wave tile 128x64, a0 = a1 = 64 rows.  TBM = 192: wave tile 96x64, a0 = 64 rows, a1 = 32 rows (phases of
// 16 / 16 / 8 / 8 MFMAs): 79 x 3 = 237 tiles for the N = 768 products of M = 15104 -- one 93 %-full round of the 256 CUs
// instead of 177 tiles on 256 -- and 2.78 instead of 2.07 rounds at N = 2304.
// KS = true (weight gradients, both operands K-strided: A [K, M], B [K, N]; TBM = 256): the same protocol on transposed
// images.  A half tile is a 64 k x 128 column image in the K-strided format of the older kernels (256-byte k-rows, chunk ^
// ks_swz(k), fragments by ds_read_b64_tr_b16) whose 128 columns are GATHERED by the DMA lanes' source offsets: image a_h holds,
// for each wave-row group, the 64 columns of its quadrant h (source columns 128 wr + 64 h + ...), image b_h the 32 columns of
// quadrant h of each of the four wave columns -- so the quadrant / phase / restaging structure above carries over unchanged.
// K split s covers k in [s kper, (s + 1) kper) (kper a multiple of 128) and stores its fp32 partial tile into slab s.
template <int GRPCH>
__device__ __forceinline__ int p8_ks_voff(long ld, int col0, int ncols, int h, int blk, int lane) {
    const int k = blk * 4 + (lane >> 4), pc = lane & 15, c = pc ^ ks_swz(k);
    const int col = (c / GRPCH) * (2 * GRPCH * 8) + h * (GRPCH * 8) + (c % GRPCH) * 8;
    return (col0 + col < ncols) ? (int)(((long)k * ld + col) * 2) : OOB;
}

// X3 = true (round 5, the split-precision mode): both operands are (hi | lo) bf16 plane pairs of fp32 tensors and the K loop
// walks the SAME K range three times -- (A hi, B hi), (A lo, B hi), (A hi, B lo) -- as one 3 nk-tile loop: only the scalar
// offset of a K tile changes (t -> pass t / nk, tile t % nk, + the plane distance where the pass reads a lo plane), so the
// phase protocol, the counted waits and the LDS images are untouched.  The epilogue is the fp32 one (aux / outputs fp32,
// optionally the result's own plane pair for the next product).
template <int TBM, bool KS = false, bool X3 = false>
__global__ __launch_bounds__(512, 1) void gemm_p8_kernel(GemmArgs p) {
    typedef typename std::conditional<X3, float, bf16>::type TE;      // epilogue storage type
    static_assert(!(X3 && (KS || TBM == 288)), "x3: K-contiguous operands, 256- or 192-row tiles");
    static_assert(TBM == 256 || ((TBM == 192 || TBM == 288) && !KS), "row tiles of 256 or (K-contiguous operands) 192 / 288");
    constexpr int WTM = TBM / 2, TMW = WTM / 16, NA1 = TMW - 4;       // wave rows, 16-row tiles per wave, tiles in a1
    constexpr int NBG = TBM / 16;                                     // DMA blocks (8 rows) per wave-row group
    constexpr int TA = TBM * 128, STAGE = TA + 256 * 128;             // bytes: A tile | B tile, 128-byte rows (64 k)
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = wave >> 2, wc = wave & 3;
    // Every kernel argument the prologue needs, requested in ONE batch: left to itself hipcc loads them where they are first
    // used -- behind the branches of the trace test and of tile_origin's order modes -- as a chain of five dependent
    // scalar-cache round trips in front of the first LDS-DMA request.
    asm volatile("" :: "s"(p.A), "s"(p.B), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.lda), "s"(p.ldb), "s"(p.ntiles), "s"(p.tiles_n),
                 "s"(p.tiles_m_fast), "s"(p.cbw), "s"(p.trace), "s"(p.dbg_flags), "s"(p.bytesA), "s"(p.bytesB), "s"(p.kper));
    const bool tr = p.trace != nullptr;                     // diagnostic timeline (mmtg_gemm_trace)
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tc1 = 0, tc2 = 0;      // wall stamps (100 MHz) + shader-clock stamps around the K loop
    ts0 = __builtin_amdgcn_s_memrealtime();
    int m0, n0, split;
    tile_origin(p, blockIdx.x, gridDim.x, m0, n0, split, TBM, 256);
    const int kbeg = KS ? split * p.kper : 0;               // host: K % 128 == 0; K splits (kper % 128 == 0) for KS only
    const int nk1 = max(0, (KS ? min(p.K, kbeg + p.kper) : p.K) - kbeg) >> 6;
    const int nk = X3 ? 3 * nk1 : nk1;                      // x3: three passes over the K range

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.bytesB, 0x00020000);
    const int sa0 = (int)((KS ? (long)kbeg * p.lda + m0 : (long)m0 * p.lda) * 2), sb0 = (int)((KS ? (long)kbeg * p.ldb + n0 : (long)n0 * p.ldb) * 2);
    const int stepa = KS ? (int)(64 * p.lda * 2) : 128, stepb = KS ? (int)(64 * p.ldb * 2) : 128;      // bytes per K tile
    // DMA blocks of this wave (two per half tile; one for a1 at 192 rows) as byte offsets inside a stage, and their lane offsets.
    // K-contiguous operands: a block is 8 rows x 128 B.  Half tile a0 = the first 64 rows of both wave-row groups -> blocks w and
    // NBG + w; a1 = the rest of both groups -> 2 blocks per wave at 256 rows (8 + w, NBG + 8 + w), 1 at 192 rows (group w / 4,
    // block 8 + w % 4).  Half tile b0 = columns 64 wc' + [0, 32) -> blocks 8 (i >> 2) + (i & 3), i = w, w + 8; b1 = those + 4.
    // K-strided operands: half tile h is the 16 KB image at h * 16 KB, a block is 4 k-rows x 256 B -> blocks w and w + 8.
    int lA00, lA01, lA10, lA11, lB00, lB01, lB10, lB11, vA00, vA01, vA10, vA11, vB00, vB01, vB10, vB11;
    int lA12 = 0, vA12 = OOB;
    if constexpr (!KS) {
        const int bA0 = wave, bA1 = bA0 + NBG;
        const int bA2 = TBM != 192 ? 8 + wave : (wave >> 2) * NBG + 8 + (wave & 3), bA3 = NBG + 8 + wave;      // bA3: 256 / 288 rows only
        // 288 rows (round 4): a1 = 80 rows of each wave-row group = 10 blocks per group; the last two of each group go to waves 0-3
        // as a THIRD request (the count of A1 requests enters no counted wait: only B0 / A0 / B1 of the tile after next are in
        // flight behind vmcnt(6), and the prologue's wait covers A1(0) however many requests it was)
        const int bA4 = wave < 2 ? 16 + wave : NBG + 16 + ((wave - 2) & 1);
        lA12 = bA4 * 1024;
        vA12 = (TBM == 288 && wave < 4) ? dma_voff<false, TBM>(p.lda, m0, p.M, 64, bA4, lane) : OOB;
        const int bB = 8 * (wave >> 2) + (wave & 3);
        lA00 = bA0 * 1024; lA01 = bA1 * 1024; lA10 = bA2 * 1024; lA11 = bA3 * 1024;
        lB00 = TA + bB * 1024; lB01 = TA + (bB + 16) * 1024; lB10 = TA + (bB + 4) * 1024; lB11 = TA + (bB + 20) * 1024;
        vA00 = dma_voff<false, TBM>(p.lda, m0, p.M, 64, bA0, lane); vA01 = dma_voff<false, TBM>(p.lda, m0, p.M, 64, bA1, lane);
        vA10 = dma_voff<false, TBM>(p.lda, m0, p.M, 64, bA2, lane);
        vA11 = TBM != 192 ? dma_voff<false, TBM>(p.lda, m0, p.M, 64, bA3, lane) : OOB;
        vB00 = dma_voff<false, 256>(p.ldb, n0, p.N, 64, bB, lane); vB01 = dma_voff<false, 256>(p.ldb, n0, p.N, 64, bB + 16, lane);
        vB10 = dma_voff<false, 256>(p.ldb, n0, p.N, 64, bB + 4, lane); vB11 = dma_voff<false, 256>(p.ldb, n0, p.N, 64, bB + 20, lane);
    } else {
        lA00 = wave * 1024; lA01 = (wave + 8) * 1024; lA10 = 16384 + lA00; lA11 = 16384 + lA01;
        lB00 = TA + lA00; lB01 = TA + lA01; lB10 = TA + lA10; lB11 = TA + lA11;
        vA00 = p8_ks_voff<8>(p.lda, m0, p.M, 0, wave, lane); vA01 = p8_ks_voff<8>(p.lda, m0, p.M, 0, wave + 8, lane);
        vA10 = p8_ks_voff<8>(p.lda, m0, p.M, 1, wave, lane); vA11 = p8_ks_voff<8>(p.lda, m0, p.M, 1, wave + 8, lane);
        vB00 = p8_ks_voff<4>(p.ldb, n0, p.N, 0, wave, lane); vB01 = p8_ks_voff<4>(p.ldb, n0, p.N, 0, wave + 8, lane);
        vB10 = p8_ks_voff<4>(p.ldb, n0, p.N, 1, wave, lane); vB11 = p8_ks_voff<4>(p.ldb, n0, p.N, 1, wave + 8, lane);
    }
    // scalar byte offset of K tile t of an operand (x3: pass = t / nk1 -- 1 reads A's lo plane, 2 reads B's -- tile t % nk1)
#define P8_PASS(t) (((t) >= nk1 ? 1 : 0) + ((t) >= 2 * nk1 ? 1 : 0))
#define P8_SA(t) (X3 ? sa0 + ((t) - P8_PASS(t) * nk1) * stepa + (P8_PASS(t) == 1 ? p.planeA : 0) : sa0 + (t) * stepa)
#define P8_SB(t) (X3 ? sb0 + ((t) - P8_PASS(t) * nk1) * stepb + (P8_PASS(t) == 2 ? p.planeB : 0) : sb0 + (t) * stepb)
    // half tile H of K tile t into stage t & 1 (a tile past the end issues out-of-range, zero-filling loads: uniform counts)
#define P8_A0(t) p8_issue(ra, smem + ((t) & 1) * STAGE + lA00, smem + ((t) & 1) * STAGE + lA01, (t) < nk ? vA00 : OOB, (t) < nk ? vA01 : OOB, P8_SA(t))
#define P8_A1(t)                                                                                                                       \
    do {                                                                                                                               \
        if constexpr (TBM != 192)                                                                                                      \
            p8_issue(ra, smem + ((t) & 1) * STAGE + lA10, smem + ((t) & 1) * STAGE + lA11, (t) < nk ? vA10 : OOB, (t) < nk ? vA11 : OOB, P8_SA(t)); \
        else                                                                                                                           \
            p8_issue1(ra, smem + ((t) & 1) * STAGE + lA10, (t) < nk ? vA10 : OOB, P8_SA(t));                                  \
        if constexpr (TBM == 288) {                                                                                                    \
            if (wave < 4 && (t) < nk) p8_issue1(ra, smem + ((t) & 1) * STAGE + lA12, vA12, P8_SA(t));                         \
        }                                                                                                                              \
    } while (0)
#define P8_B0(t) p8_issue(rb, smem + ((t) & 1) * STAGE + lB00, smem + ((t) & 1) * STAGE + lB01, (t) < nk ? vB00 : OOB, (t) < nk ? vB01 : OOB, P8_SB(t))
#define P8_B1(t) p8_issue(rb, smem + ((t) & 1) * STAGE + lB10, smem + ((t) & 1) * STAGE + lB11, (t) < nk ? vB10 : OOB, (t) < nk ? vB11 : OOB, P8_SB(t))

    // lane offsets of the fragments inside a stage.  K-contiguous: A rows WTM wr + 64 ah + 16 i + l15, B rows (= columns) 64 wc +
    // 32 bh + 16 j + l15; the chunk swizzle (row & 7) does not depend on wr / ah / bh / i / j (all multiples of 8), so one offset
    // per kk serves all.  K-strided: transposed-read offsets of columns 64 wr + 16 i (A images) / 32 wc + 16 j (B images).
    const int rowa = WTM * wr + l15, rowb = 64 * wc + l15;
    int oa[KS ? 4 : 2], ob[2];
    if constexpr (!KS) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            oa[kk] = rowa * 128 + ((((kk << 2) + g) ^ (rowa & 7)) << 4);
            ob[kk] = TA + rowb * 128 + ((((kk << 2) + g) ^ (rowb & 7)) << 4);
        }
    } else {
        ks_offsets<128, 4>(wr * 64, lane, oa);
        ks_offsets<128, 2>(wc * 32, lane, ob);
        ob[0] += TA; ob[1] += TA;
    }

    f32x4 acc[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // in flight behind the counted wait: B0, A0, B1 of the tile after next = 6 DMA requests per wave in both configurations
    P8_B0(0); P8_A0(0); P8_B1(0); P8_A1(0);
    P8_B0(1); P8_A0(1); P8_B1(1);
    wait_vmcnt<6>();                       // K tile 0 has landed (mine; the barrier makes it everyone's)
    asm volatile("s_barrier" ::: "memory");
    if (tr) { ts1 = __builtin_amdgcn_s_memrealtime(); tc1 = __builtin_amdgcn_s_memtime(); }
    if (wr == 1) asm volatile("s_barrier" ::: "memory");          // the second wave group runs one barrier behind the first

    // (A/B: MMTG_P8_NOPRIO=1 -> dbg_flags bit 128: no priority raise around the MFMA clusters -- the K-strided form's load wave
    //  needs twice the issue slots of the K-contiguous one: 24 transposed reads in phase 1 against 12 ds_read_b128)
    const bool mprio = !(p.dbg_flags & 128);
    bf16x8 fa[NA1 > 4 ? NA1 : 4][2], fb0[2][2], fb1[2][2];
#define P8_RD_A(ST, AH, NA)                                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < (NA); ++i) {     \
        if constexpr (!KS) fa[i][kk] = p8_ld(smem + (ST) * STAGE + oa[kk] + ((AH) * 64 + i * 16) * 128);    \
        else fa[i][kk] = tr_read_pair(smem + (ST) * STAGE + (AH) * 16384, oa[KS ? i : 0] + kk * 8192, oa[KS ? i : 0] + kk * 8192 + 1024); \
    }
#define P8_RD_B(ST, BH, F)                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int j = 0; j < 2; ++j) {        \
        if constexpr (!KS) F[j][kk] = p8_ld(smem + (ST) * STAGE + ob[kk] + ((BH) * 32 + j * 16) * 128);     \
        else F[j][kk] = tr_read_pair(smem + (ST) * STAGE + (BH) * 16384, ob[j] + kk * 8192, ob[j] + kk * 8192 + 1024); \
    }
    // one K tile held in stage ST (a compile-time 0 / 1: every fragment address is lane offset + immediate)
    // MMTG_P8_PHASE_TRACE (diagnostic build only, tools/p8_phase_trace.py): shader-clock stamps of K tile 8 in workgroup 0 -- per
    // phase: start, loads issued, barrier + LDS wait passed (MFMAs start), MFMAs issued, second barrier passed
#ifdef MMTG_P8_PHASE_TRACE
    unsigned pst[20];
#pragma unroll
    for (int i_ = 0; i_ < 20; ++i_) pst[i_] = 0;
#define P8_STAMP(kt_, i_) do { if (tr && (kt_) == 8 && blockIdx.x == 0) pst[i_] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define P8_STAMP(kt_, i_) do { } while (0)
#endif
#define P8_TILE(ST, kt)                                                                                    \
    do {                                                                                                   \
        P8_STAMP(kt, 0);                                                                                   \
        P8_RD_B(ST, 0, fb0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        P8_RD_A(ST, 0, 4);                                                                                 \
        P8_A1((kt) + 1);                                                                                   \
        /* the b0 reads are done: b0 may be restaged next phase (K-strided: 8 + 16 reads, the counter saturates at 15) */ \
        if constexpr (!KS) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                               \
        else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");                                           \
        P8_STAMP(kt, 1);                                                                                   \
        P8_SYNC_IN(); P8_STAMP(kt, 2); p8_mfma<0, 0, 4, TMW>(fa, fb0, acc, mprio); P8_STAMP(kt, 3); P8_SYNC_OUT(); P8_STAMP(kt, 4);  \
        P8_RD_B(ST, 1, fb1);                                                                               \
        P8_B0((kt) + 2);                                                                                   \
        P8_STAMP(kt, 5);                                                                                   \
        P8_SYNC_IN(); P8_STAMP(kt, 6); p8_mfma<0, 1, 4, TMW>(fa, fb1, acc, mprio); P8_STAMP(kt, 7); P8_SYNC_OUT(); P8_STAMP(kt, 8);  \
        P8_RD_A(ST, 1, NA1);                                                                               \
        P8_A0((kt) + 2);                                                                                   \
        P8_STAMP(kt, 9);                                                                                   \
        P8_SYNC_IN(); P8_STAMP(kt, 10); p8_mfma<1, 1, NA1, TMW>(fa, fb1, acc, mprio); P8_STAMP(kt, 11); P8_SYNC_OUT(); P8_STAMP(kt, 12); \
        P8_B1((kt) + 2);                                                                                   \
        wait_vmcnt<6>();                                       /* K tile kt + 1 has landed */               \
        P8_STAMP(kt, 13);                                                                                  \
        P8_SYNC_IN(); P8_STAMP(kt, 14); p8_mfma<1, 0, NA1, TMW>(fa, fb0, acc, mprio); P8_STAMP(kt, 15); P8_SYNC_OUT(); P8_STAMP(kt, 16); \
    } while (0)
    for (int kt = 0; kt < nk; kt += 2) {
        P8_TILE(0, kt);
        P8_TILE(1, kt + 1);
    }
#undef P8_TILE
#undef P8_STAMP
#undef P8_RD_A
#undef P8_RD_B
#undef P8_A0
#undef P8_A1
#undef P8_B0
#undef P8_B1
#undef P8_SA
#undef P8_SB
#undef P8_PASS
    wait_vmcnt<0>();                                     // the zero-fill tail loads
    if (tr) { ts2 = __builtin_amdgcn_s_memrealtime(); tc2 = __builtin_amdgcn_s_memtime(); }
    if (wr == 0) asm volatile("s_barrier" ::: "memory");           // re-join the two groups
    asm volatile("s_barrier" ::: "memory");                        // every wave is done with the stages: they become epilogue scratch
    if constexpr (KS) p.C = reinterpret_cast<char*>(p.C) + (long)split * p.split_stride;     // MMTG_EPI_SPLIT slab
    // (aux vectors two bands ahead; the 256-row configuration -- 128-row wave tiles -- also carries the dGELU column sums)
    gemm_epilogue<TE, false, TMW, 4, (TBM == 256 ? 2 : TBM == 288 ? -1 : -2), X3>(p, acc, m0 + wr * WTM, n0 + wc * 64, g, l15, smem + wave * epi_scratch_bytes<TMW, 4>(), lane);
#ifdef MMTG_P8_PHASE_TRACE
    if (tr && blockIdx.x == 0 && lane == 0 && (int)(gridDim.x + 4 * wave + 4) <= p.trace_n) {
        unsigned long long* r = p.trace + 6 * (size_t)(gridDim.x + 4 * wave);       // 24 words behind the workgroups' rows
#pragma unroll
        for (int i_ = 0; i_ < 20; ++i_) r[i_] = pst[i_];
    }
#endif
    if (tr && wave == 0 && (int)blockIdx.x < p.trace_n) {
        wait_vmcnt<0>();                       // the output stores are part of the epilogue's time
        const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* r = p.trace + 6 * (size_t)blockIdx.x;
            // r[4]: K tiles in the low 20 bits, shader cycles of the K loop above them (in-loop clock = cycles / wall: the chip
            // lowers its clock under MFMA load, MI355X_MICROARCH.md "DVFS give-back" item 6)
            r[0] = ts0; r[1] = ts1; r[2] = ts2; r[3] = ts3; r[4] = (unsigned long long)nk | ((tc2 - tc1) << 20);
            r[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
}

// Persistent form of the eight-phase kernel (K-contiguous operands): one workgroup per CU walks the output tiles
// item, item + grid, ... and the staging protocol never stops -- the issue slots of an item's last two K tiles, which would
// stage tiles nk and nk + 1, stage K tiles 0 and 1 of the NEXT item instead, so there is no prologue after the first item, no
// launch gap between rounds, and the counted wait of the last tile's fourth phase is the next item's "tile 0 has landed".  The
// epilogue runs between two items from a scratch area of its own behind the stages (2 x 64 KB + 8 x 4 KB = the whole 160 KB
// LDS at 256 rows), without a barrier: its stores sit in the vmcnt queue in front of the next DMA requests and are retired
// by the first counted wait of the next item (vmcnt counts loads, stores and LDS-DMA together, in issue order).  The DMA
// lane offsets do not depend on the item (the tile origin travels in the scalar offset); only the row-limit tests do.
template <int TBM>
__global__ __launch_bounds__(512, 1) void gemm_p8p_kernel(GemmArgs p) {
    typedef bf16 T;
    static_assert(TBM == 256 || TBM == 192, "row tiles of 256 or 192");
    constexpr int WTM = TBM / 2, TMW = WTM / 16, NA1 = TMW - 4;
    constexpr int NBG = TBM / 16;
    constexpr int TA = TBM * 128, STAGE = TA + 256 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two stages | epilogue scratch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15, l8 = lane >> 3;
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = p.K >> 6;                                // host: K % 128 == 0, no K splits
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.bytesB, 0x00020000);
    // this wave's DMA blocks (see gemm_p8_kernel): first row of each block, its LDS offset inside a stage, its lane offset
    const int bA0 = wave, bA1 = bA0 + NBG;
    const int bA2 = TBM == 256 ? 8 + wave : (wave >> 2) * NBG + 8 + (wave & 3), bA3 = NBG + 8 + wave;
    const int bB0 = 8 * (wave >> 2) + (wave & 3), bB1 = bB0 + 16, bB2 = bB0 + 4, bB3 = bB0 + 20;
    const int c8 = ((lane & 7) ^ l8) * 16;                   // chunk swizzle of dma_voff (block rows are multiples of 8)
    const int vrA = (int)((long)l8 * p.lda * 2) + c8, vrB = (int)((long)l8 * p.ldb * 2) + c8;
    const int vA00 = vrA + (int)((long)bA0 * 8 * p.lda * 2), vA01 = vrA + (int)((long)bA1 * 8 * p.lda * 2);
    const int vA10 = vrA + (int)((long)bA2 * 8 * p.lda * 2), vA11 = vrA + (int)((long)bA3 * 8 * p.lda * 2);
    const int vB00 = vrB + (int)((long)bB0 * 8 * p.ldb * 2), vB01 = vrB + (int)((long)bB1 * 8 * p.ldb * 2);
    const int vB10 = vrB + (int)((long)bB2 * 8 * p.ldb * 2), vB11 = vrB + (int)((long)bB3 * 8 * p.ldb * 2);

    int item = blockIdx.x, m0, n0, split, m0n = 0, n0n = 0;
    tile_origin(p, item, p.nitems, m0, n0, split, TBM, 256);
    bool has_next = false;
    // K tile t of the current item (t < nk) or K tile t - nk of the next one; rows past M / N and tiles past the last item
    // are requested out of range (zero fill, uniform request counts)
#define P8P_SEL(t)                                                                                          \
    const bool cur_ = (t) < nk;                                                                             \
    const bool live_ = cur_ || has_next;                                                                    \
    const int tt_ = cur_ ? (t) : (t) - nk;                                                                  \
    const int mm_ = cur_ ? m0 : m0n, nn_ = cur_ ? n0 : n0n;                                                 \
    const int sa_ = (int)(((long)mm_ * p.lda + (long)tt_ * 64) * 2), sb_ = (int)(((long)nn_ * p.ldb + (long)tt_ * 64) * 2); \
    const int la_ = live_ ? p.M - mm_ : 0, lb_ = live_ ? p.N - nn_ : 0;                                     \
    char* st_ = smem + ((t) & 1) * STAGE
#define P8P_A0(t) do { P8P_SEL(t); (void)sb_; (void)lb_;                                                     \
        p8_issue(ra, st_ + bA0 * 1024, st_ + bA1 * 1024, bA0 * 8 + l8 < la_ ? vA00 : OOB, bA1 * 8 + l8 < la_ ? vA01 : OOB, sa_); } while (0)
#define P8P_A1(t) do { P8P_SEL(t); (void)sb_; (void)lb_;                                                     \
        if constexpr (TBM == 256)                                                                           \
            p8_issue(ra, st_ + bA2 * 1024, st_ + bA3 * 1024, bA2 * 8 + l8 < la_ ? vA10 : OOB, bA3 * 8 + l8 < la_ ? vA11 : OOB, sa_); \
        else p8_issue1(ra, st_ + bA2 * 1024, bA2 * 8 + l8 < la_ ? vA10 : OOB, sa_); } while (0)
#define P8P_B0(t) do { P8P_SEL(t); (void)sa_; (void)la_;                                                     \
        p8_issue(rb, st_ + TA + bB0 * 1024, st_ + TA + bB1 * 1024, bB0 * 8 + l8 < lb_ ? vB00 : OOB, bB1 * 8 + l8 < lb_ ? vB01 : OOB, sb_); } while (0)
#define P8P_B1(t) do { P8P_SEL(t); (void)sa_; (void)la_;                                                     \
        p8_issue(rb, st_ + TA + bB2 * 1024, st_ + TA + bB3 * 1024, bB2 * 8 + l8 < lb_ ? vB10 : OOB, bB3 * 8 + l8 < lb_ ? vB11 : OOB, sb_); } while (0)

    const int rowa = WTM * wr + l15, rowb = 64 * wc + l15;
    int oa[2], ob[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        oa[kk] = rowa * 128 + ((((kk << 2) + g) ^ (rowa & 7)) << 4);
        ob[kk] = TA + rowb * 128 + ((((kk << 2) + g) ^ (rowb & 7)) << 4);
    }
    f32x4 acc[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    P8P_B0(0); P8P_A0(0); P8P_B1(0); P8P_A1(0);
    P8P_B0(1); P8P_A0(1); P8P_B1(1);
    wait_vmcnt<6>();
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");          // the second wave group runs one barrier behind the first

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define P8_RD_A(ST, AH, NA)                                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < (NA); ++i)       \
        fa[i][kk] = p8_ld(smem + (ST) * STAGE + oa[kk] + ((AH) * 64 + i * 16) * 128)
#define P8_RD_B(ST, BH, F)                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
        F[j][kk] = p8_ld(smem + (ST) * STAGE + ob[kk] + ((BH) * 32 + j * 16) * 128)
#define P8P_TILE(ST, kt)                                                                                   \
    do {                                                                                                   \
        P8_RD_B(ST, 0, fb0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        P8_RD_A(ST, 0, 4);                                                                                 \
        P8P_A1((kt) + 1);                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                 \
        P8_SYNC_IN(); p8_mfma<0, 0, 4, TMW>(fa, fb0, acc); P8_SYNC_OUT();                                   \
        P8_RD_B(ST, 1, fb1);                                                                               \
        P8P_B0((kt) + 2);                                                                                  \
        P8_SYNC_IN(); p8_mfma<0, 1, 4, TMW>(fa, fb1, acc); P8_SYNC_OUT();                                   \
        P8_RD_A(ST, 1, NA1);                                                                               \
        P8P_A0((kt) + 2);                                                                                  \
        P8_SYNC_IN(); p8_mfma<1, 1, NA1, TMW>(fa, fb1, acc); P8_SYNC_OUT();                                 \
        P8P_B1((kt) + 2);                                                                                  \
        wait_vmcnt<6>();                                                                                   \
        P8_SYNC_IN(); p8_mfma<1, 0, NA1, TMW>(fa, fb0, acc); P8_SYNC_OUT();                                 \
    } while (0)
    for (;;) {
        const int nxt = item + (int)gridDim.x;
        has_next = nxt < p.nitems;
        if (has_next) tile_origin(p, nxt, p.nitems, m0n, n0n, split, TBM, 256);
        for (int kt = 0; kt < nk; kt += 2) {
            P8P_TILE(0, kt);
            P8P_TILE(1, kt + 1);
        }
        gemm_epilogue<T, false, TMW, 4, (TBM == 256 ? 2 : -2)>(p, acc, m0 + wr * WTM, n0 + wc * 64, g, l15,
                                                               smem + 2 * STAGE + wave * epi_scratch_bytes<TMW, 4>(), lane);
        if (!has_next) break;
        // (no unroll pragma: constant trip counts, fully unrolled anyway -- with the pragma hipcc reports the already-unrolled
        //  loop as "not unrolled" after the epilogue's inlining)
        for (int i = 0; i < TMW; ++i)
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        item = nxt; m0 = m0n; n0 = n0n;
    }
#undef P8P_TILE
#undef P8_RD_A
#undef P8_RD_B
#undef P8P_A0
#undef P8P_A1
#undef P8P_B0
#undef P8P_B1
#undef P8P_SEL
    if (wr == 0) asm volatile("s_barrier" ::: "memory");           // balance the second group's extra barrier
}

template <int TBM>
int launch_p8p(const GemmArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    const size_t shm = 2 * (TBM + 256) * 128 + 8 * epi_scratch_bytes<TBM / 32, 4>();
    if (!attr_done) {
        int rc = set_lds(gemm_p8p_kernel<TBM>, shm, 512, TBM == 256 ? "persistent eight-phase 256x256" : "persistent eight-phase 192x256");
        if (rc) return rc;
        attr_done = true;
    }
    GemmArgs b = a;
    b.tiles_n = cdiv(a.N, 256);
    b.ntiles = cdiv(a.M, TBM) * b.tiles_n;
    b.nitems = b.ntiles;
    b.tiles_m_fast = 0;
    b.cbw = 0;
    if (!(a.dbg_flags & 1)) b.cbw = column_block(a, TBM, 256, num_cus());
    const int grid = b.nitems < num_cus() ? b.nitems : num_cus();      // (a multiple of 8 when it is not nitems: XCD-contiguous runs)
    hipLaunchKernelGGL(gemm_p8p_kernel<TBM>, dim3(grid), dim3(512), shm, stream, b);
    return MMTG_OK;
}

template <int TBM, bool KS = false, bool X3 = false>
int launch_p8(const GemmArgs& a, int splits, hipStream_t stream) {
    static bool attr_done = false;
    const size_t shm = 2 * (TBM + 256) * 128;
    if (!attr_done) {
        int rc = set_lds(gemm_p8_kernel<TBM, KS, X3>, shm, 512, KS ? "eight-phase 256x256 K-strided" : TBM == 256 ? "eight-phase 256x256" : TBM == 288 ? "eight-phase 288x256" : "eight-phase 192x256");
        if (rc) return rc;
        attr_done = true;
    }
    GemmArgs b = a;
    b.tiles_n = cdiv(a.N, 256);
    b.ntiles = cdiv(a.M, TBM) * b.tiles_n;
    b.nitems = b.ntiles * splits;
    b.tiles_m_fast = KS && b.tiles_n > cdiv(a.M, TBM) && !(a.dbg_flags & 1);
    b.cbw = 0;
    if (!KS && !(a.dbg_flags & 1)) b.cbw = column_block(a, TBM, 256, num_cus());
    hipLaunchKernelGGL((gemm_p8_kernel<TBM, KS, X3>), dim3(b.nitems), dim3(512), shm, stream, b);
    return MMTG_OK;
}

template <bool AKS, bool BKS, bool SLAB = false>
int launch_occ4(const GemmArgs& a, int splits, hipStream_t stream) {
    static bool attr_done = false;
    constexpr int TBM = 128;
    const size_t shm = (TBM + 128) * 128;
    if (!attr_done) {
        int rc = set_lds(gemm_occ4_kernel<AKS, BKS, SLAB>, shm, 256, "single-stage 128x128, 4 per CU");
        if (rc) return rc;
        attr_done = true;
    }
    GemmArgs b = a;
    b.tiles_n = cdiv(a.N, 128);
    b.ntiles = cdiv(a.M, TBM) * b.tiles_n;
    b.nitems = b.ntiles * splits;
    b.tiles_m_fast = AKS && BKS && b.tiles_n > cdiv(a.M, TBM);
    if (!(AKS && BKS) && splits == 1 && !(a.dbg_flags & 1)) b.cbw = column_block(a, TBM, 128, 3 * num_cus());
    hipLaunchKernelGGL((gemm_occ4_kernel<AKS, BKS, SLAB>), dim3(b.nitems), dim3(256), shm, stream, b);
    return MMTG_OK;
}

int launch_dma(const GemmArgs& a, int transA, int transB, int splits, int skinny, int wide, int persist, hipStream_t stream) {
    if (a.dbg_flags & 2) {
        if (!transA && transB) return launch_occ4<false, false>(a, splits, stream);
        if (!transA && !transB) return launch_occ4<false, true>(a, splits, stream);
        if (a.epi == MMTG_EPI_SPLIT) return launch_occ4<true, true, true>(a, splits, stream);
        return launch_occ4<true, true>(a, splits, stream);
    }
    if (persist && !wide && !skinny && (a.dbg_flags & 32)) {      // 256x128 persistent (NT / NN)
        if (!transA && transB) return launch_pp<false, false, 256>(a, splits, stream);
        if (!transA && !transB) return launch_pp<false, true, 256>(a, splits, stream);
    }
    if (persist && !wide && !skinny) {
        if (!transA && transB) return launch_pp<false, false>(a, splits, stream);
        if (!transA && !transB) return launch_pp<false, true>(a, splits, stream);
        return launch_pp<true, true>(a, splits, stream);
    }
    if (wide) {   // 192x128 tiles, 3x2 waves: N = 768 products of M = 15104 fit one round of 2 workgroups/CU
        if (!transA && transB) return launch_dma_cfg<false, false, 192, 128, 3, 2, 2>(a, splits, stream);
        if (!transA && !transB) return launch_dma_cfg<false, true, 192, 128, 3, 2, 2>(a, splits, stream);
    }
    if (!transA && transB && !skinny && (a.dbg_flags & 64)) {
        // experiment (MMTG_GEMM_BIG=1): the vendor library's shape compiled from the generic kernel -- 256x256 tiles, FOUR waves of
        // 128x128 (one per SIMD, 256 accumulator registers each), two 64 KB stages, one workgroup per CU
        return launch_dma_cfg<false, false, 256, 256, 2, 2, 2>(a, splits, stream);
    }
    if (!transA && transB) {
        if (skinny) {
            // batch-sized M (decode): tile shape of the weight-streaming products.  A workgroup pulls (BM + BN) x K
            // bytes through its CU's L2 port, so with 256x32 tiles every workgroup re-reads ALL activation rows and
            // the product is bound by that panel, not by the weights; 64x64 tiles pull less than half as much per
            // workgroup and give 4x the workgroups (graph-replayed times at M = 256: c_attn 7.3 -> 5.8 us, attn c_proj
            // 5.9 -> 4.1, c_fc 10.0 -> 6.7, mlp c_proj 7.7 -> 6.2, LM head 20.2 -> 16.0; profiles/r02_decode_gemm_tiles.log)
            const char* e = getenv("MMTG_SKINNY_CFG");
            const int cfg = e ? atoi(e) : 1;
            switch (cfg) {
                case 1: return launch_dma_cfg<false, false, 64, 64, 2, 2, 4>(a, splits, stream);
                case 2: return launch_dma_cfg<false, false, 128, 64, 2, 2, 4>(a, splits, stream);
                case 3: return launch_dma_cfg<false, false, 64, 128, 2, 2, 4>(a, splits, stream);
                case 4: return launch_dma_cfg<false, false, 128, 32, 4, 1, 4>(a, splits, stream);
                case 5: return launch_dma_cfg<false, false, 64, 32, 2, 1, 4>(a, splits, stream);
                case 6: return launch_dma_cfg<false, false, 64, 64, 2, 2, 5>(a, splits, stream);    // 80 KB: still two per CU
                case 7: return launch_dma_cfg<false, false, 64, 64, 2, 2, 8>(a, splits, stream);    // 128 KB: a 384-deep K slice all in flight
                default: return launch_dma_cfg<false, false, 256, 32, 4, 1, 4>(a, splits, stream);
            }
        }
        return launch_dma_cfg<false, false, 128, 128, 2, 2, 2>(a, splits, stream);
    }
    if (!transA && !transB) return launch_dma_cfg<false, true, 128, 128, 2, 2, 2>(a, splits, stream);
    return launch_dma_cfg<true, true, 128, 128, 2, 2, 2>(a, splits, stream);
}

}  // namespace

// C ABI ---------------------------------------------------------------------
static unsigned long long* g_trace = nullptr;
static int g_trace_n = 0;
// CUs the tile-shape rule of the eight-phase kernel may count on (0 = all; > 0 = that many; < 0 = all but that many).  One
// workgroup of that kernel takes a whole CU (all 512 registers per lane, 112-128 KB of LDS), so a long-running kernel on
// another stream -- the RCCL collectives of the data-parallel step -- takes CUs away from it for its whole duration: a
// 237-tile product sized for 256 CUs would need a second, almost empty round.  mmtg_amd.ddp reserves CUs when collectives run.
static int g_cu_budget = 0;
extern "C" int mmtg_gemm_cu_budget(int cus) {
    g_cu_budget = cus;
    return MMTG_OK;
}
static long p8_cus() {
    const long n = num_cus();
    long b = g_cu_budget > 0 ? g_cu_budget : n + g_cu_budget;
    if (b > n) b = n;
    if (b < 8) b = 8;
    return b;
}

extern "C" int mmtg_gemm_trace(void* buf, int max_wgs) {
    MMTG_REQUIRE(!buf || max_wgs > 0, "gemm_trace: max_wgs must be positive");
    g_trace = reinterpret_cast<unsigned long long*>(buf);
    g_trace_n = buf ? max_wgs : 0;
    return MMTG_OK;
}

extern "C" int mmtg_gemm_gather(int mode, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                const float* bias, int epi, const int* rows, int table_rows, const void* aux, long ldaux,
                                const int* aux_rows, int splits, void* stream) {
    MMTG_REQUIRE(mode == 0 || mode == 1, "gemm_gather: mode 0 (forward) or 1 (weight gradient)");
    MMTG_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && rows && table_rows > 0, "gemm_gather: bad arguments");
    MMTG_REQUIRE(MMTG_ALIGNED16(A) && MMTG_ALIGNED16(B) && MMTG_ALIGNED16(C) && lda % 8 == 0 && ldb % 8 == 0, "gemm_gather: 16-byte alignment");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.aux_rows = aux_rows; a.gather = rows;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
    a.epi = epi; a.use_tr = 1; a.alpha = 1.0f; a.drop_inv_keep = 1.0f;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) {
        // A = the table [table_rows, lda] (rows gathered), B = weights [N, K] K-contiguous
        MMTG_REQUIRE(K % 64 == 0 && N % 8 == 0 && ldc % 8 == 0, "gemm_gather: forward needs K %% 64 == 0 and N, ldc %% 8 == 0");
        MMTG_REQUIRE(epi == MMTG_EPI_NONE || epi == MMTG_EPI_TANH || (epi == MMTG_EPI_TANH_ADD && aux && ldaux % 8 == 0 && MMTG_ALIGNED16(aux)),
                     "gemm_gather: forward epilogues are NONE, TANH and TANH_ADD (with aux)");
        const long bytesA = ((long)(table_rows - 1) * lda + K) * 2, bytesB = ((long)(N - 1) * ldb + K) * 2;
        MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L, "gemm_gather: operands must stay below 2 GiB");
        a.bytesA = (int)bytesA; a.bytesB = (int)bytesB;
        a.kper = K;
        a.tiles_n = cdiv(N, 128);
        a.ntiles = cdiv(M, 128) * a.tiles_n;
        a.nitems = a.ntiles;
        ProfScope prof(MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K, 2.0 * ((double)M * K + (double)N * K + (double)M * N));
        static bool attr_done = false;
        const size_t shm = 2 * (128 + 128) * 128;
        if (!attr_done) {
            int rc = set_lds(gemm_dma_kernel<false, false, 128, 128, 2, 2, 2, true>, shm, 256, "gather forward 128x128");
            if (rc) return rc;
            attr_done = true;
        }
        hipLaunchKernelGGL((gemm_dma_kernel<false, false, 128, 128, 2, 2, 2, true>), dim3(a.nitems), dim3(256), shm, s, a);
    } else {
        // A = d(pre-activation) [K, lda >= M] (K-strided), B = the table [table_rows, ldb] (k-rows gathered), C = fp32 slabs
        MMTG_REQUIRE(epi == MMTG_EPI_SPLIT && splits >= 1 && M % 8 == 0 && N % 8 == 0 && ldc % 8 == 0, "gemm_gather: weight gradient writes MMTG_EPI_SPLIT slabs");
        const long bytesA = ((long)(K - 1) * lda + M) * 2, bytesB = ((long)(table_rows - 1) * ldb + N) * 2;
        MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L, "gemm_gather: operands must stay below 2 GiB");
        a.bytesA = (int)bytesA; a.bytesB = (int)bytesB;
        a.out_f32 = 1;
        a.kper = cdiv(cdiv(K, splits), 64) * 64;
        a.split_stride = (long)M * ldc * 4;
        a.tiles_n = cdiv(N, 128);
        a.ntiles = cdiv(M, 128) * a.tiles_n;
        a.nitems = a.ntiles * splits;
        a.tiles_m_fast = a.tiles_n > cdiv(M, 128);
        ProfScope prof(MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K, 2.0 * ((double)M * K + (double)N * K) + 4.0 * (double)M * N);
        static bool attr_done = false;
        const size_t shm = (128 + 128) * 128;
        if (!attr_done) {
            int rc = set_lds(gemm_occ4_kernel<true, true, true, true>, shm, 256, "gather weight gradient");
            if (rc) return rc;
            attr_done = true;
        }
        hipLaunchKernelGGL((gemm_occ4_kernel<true, true, true, true>), dim3(a.nitems), dim3(256), shm, s, a);
    }
    MMTG_LAUNCH_CHECK("gemm_gather");
    return MMTG_OK;
}

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
    const bool wgrad = transA && !transB;
    MMTG_REQUIRE(epi != MMTG_EPI_ATOMIC || wgrad, "gemm: the atomic epilogue belongs to the transA=1,transB=0 (weight-gradient) layout");
    MMTG_REQUIRE(!wgrad || epi == MMTG_EPI_ATOMIC || epi == MMTG_EPI_SPLIT,
                 "gemm: the weight-gradient layout has the atomic and the split (slab) epilogues only");
    if (epi != MMTG_EPI_ATOMIC) {
        MMTG_REQUIRE(N % 8 == 0 && ldc % 8 == 0, "gemm: N and ldc must be multiples of 8 (N=%d ldc=%ld)", N, ldc);
        MMTG_REQUIRE(!bias || MMTG_ALIGNED16(bias), "gemm: bias must be 16-byte aligned");
        MMTG_REQUIRE(splits <= 1 || epi == MMTG_EPI_SPLIT, "gemm: split-K needs the atomic or the split epilogue");
        if (epi == MMTG_EPI_SPLIT)
            MMTG_REQUIRE(out_f32 && !bias && (dtype == MMTG_BF16 ? !(flags & (MMTG_GEMM_REGSTAGE | MMTG_GEMM_NO_TR | MMTG_GEMM_PERSIST)) : true),
                         "gemm: the split epilogue stores raw fp32 partial products (out_f32, no bias) from the bf16 LDS-DMA kernels "
                         "or the fp32 kernel");
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
    a.trace = g_trace; a.trace_n = g_trace_n;
    a.gelu_grad = (flags & MMTG_GEMM_GELU_GRAD) ? 1 : 0;
    a.dbg_flags = ((flags & MMTG_GEMM_ROW_ORDER) ? 1 : 0) | ((flags & MMTG_GEMM_COL_BLOCK) ? 16 : 0) | ((flags & MMTG_GEMM_P256) ? 32 : 0);     // bit 1 (value 2): single-stage kernel, set below
    static const int p8_noprio = getenv("MMTG_P8_NOPRIO") ? atoi(getenv("MMTG_P8_NOPRIO")) : 0;
    if (p8_noprio) a.dbg_flags |= 128;
    a.tiles_n = cdiv(N, BN); a.alpha = alpha;
    // byte extents for the buffer descriptors of the LDS-DMA pipeline (offsets are 32-bit)
    const long esz = dtype == MMTG_F32 ? 4 : 2;
    const long bytesA = ((long)((transA ? K : M) - 1) * lda + (transA ? M : K)) * esz;
    const long bytesB = ((long)((transB ? N : K) - 1) * ldb + (transB ? K : N)) * esz;
    const bool small = bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L;
    MMTG_REQUIRE(!(wgrad && epi == MMTG_EPI_SPLIT) || small, "gemm: the weight-gradient slab epilogue needs operands below 2 GiB");
    a.bytesA = (int)(small ? bytesA : 0); a.bytesB = (int)(small ? bytesB : 0);
    const int bk = dtype == MMTG_F32 ? 32 : 64;
    if (splits < 1) splits = 1;
    int kper = cdiv(cdiv(K, splits), bk) * bk;
    // (the split epilogue keeps the caller's slab count: a K slice past the end stores zeros)
    if (epi != MMTG_EPI_SPLIT) splits = cdiv(K, kper);
    a.kper = kper;
    a.split_stride = epi == MMTG_EPI_SPLIT ? (long)M * ldc * 4 : 0;
    a.drop_thresh = drop_thresh; a.drop_seed = drop_seed;
    a.drop_inv_keep = drop_thresh ? (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh)) : 1.0f;
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes of the launch: A and B once, C once, plus what the fused epilogue must read or write by its
    // definition -- the residual / saved pre-activation / attention context it consumes (aux) and the pre-activation it
    // emits (GELU's aux2).  Extra split-K slabs are a design choice, not algorithmic bytes: C counts once.
    const double esz_ = dtype == MMTG_F32 ? 4 : 2;
    double alg_bytes = esz_ * ((double)M * K + (double)N * K) + (double)M * N * (out_f32 ? 4 : esz_);
    if (epi == MMTG_EPI_RESID || epi == MMTG_EPI_DGELU || epi == MMTG_EPI_DTANH || epi == MMTG_EPI_ROWDOT) alg_bytes += esz_ * (double)M * N;
    if (epi == MMTG_EPI_GELU) alg_bytes += esz_ * (double)M * N;
    ProfScope prof(dtype == MMTG_F32 ? MMTG_PROF_GEMM_F32 : MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K, alg_bytes);
    a.ntiles = cdiv(M, BM) * cdiv(N, BN);
    a.nitems = a.ntiles * splits;
    // (tile_origin divides by float reciprocal: exact below 2^20 work items, gemm_common.h fdiv_small)
    MMTG_REQUIRE((long)cdiv(M, BM) * cdiv(N, BN) * splits < (1L << 20), "gemm: %ld work items exceed the tile-order arithmetic's 2^20", (long)cdiv(M, BM) * cdiv(N, BN) * splits);
    dim3 grid(a.nitems);
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
        // MMTG_GEMM_WIDE_RULE (A/B): 0 = never, 1 = one-round rule only, 2 = one-round rule or N >= 4096 (round 1's)
        static const int wide_rule = getenv("MMTG_GEMM_WIDE_RULE") ? atoi(getenv("MMTG_GEMM_WIDE_RULE")) : 1;
        if (!wide && !(flags & MMTG_GEMM_NO_WIDE) && !transA && !skinny && M >= 1024 && wide_rule) {
            const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t192 = (long)cdiv(M, 192) * cdiv(N, 128);
            wide = (t128 > 512 && t192 <= 512) || (wide_rule >= 2 && N >= 4096);
        }
        if (epi == MMTG_EPI_DGELU && aux2) {   // the fused column sums are not built into the 6-wave kernel
            MMTG_REQUIRE(!(flags & MMTG_GEMM_WIDE), "gemm: DGELU with column sums (aux2) has no 192x128 configuration");
            wide = false;
        }
        // The persistent pipelined kernel is opt-in: with more 128x128 items than CU slots it measured
        // within +-3 % of the plain launch (its deferred epilogue still occupies the wave for 2.8-7 us per
        // item; timelines in profiles/r01_v4_gemm_timeline.log).
        bool persist = (flags & (MMTG_GEMM_PERSIST | MMTG_GEMM_P256)) != 0;
        // 256x128 persistent kernel (round 2, MMTG_GEMM_P256): a quarter fewer fill bytes per FLOP, but ONE workgroup per CU,
        // so every item's epilogue is exposed.  Warm and without dropout it wins on the N <= 1024, K >= 2048 products (fc2
        // forward 88 -> 79 us, dgrad fc1 79 -> 77) and loses 20-50 % at K = 768 with N >= 2304; inside the training step
        // (cold operands, dropout hashing in the exposed epilogue) the rule below costs +1.6 ms per step
        // (profiles/r02_v4_gemm_p256_ab.txt), so it is OFF unless MMTG_GEMM_P256_RULE=1|2 asks for it.
        static const int p256_rule = getenv("MMTG_GEMM_P256_RULE") ? atoi(getenv("MMTG_GEMM_P256_RULE")) : 0;
        if (p256_rule && !persist && !transA && !skinny && !(flags & (MMTG_GEMM_WIDE | MMTG_GEMM_OCC4 | MMTG_GEMM_NO_PERSIST)) && M >= 4096 && N <= 1024 &&
            K >= (p256_rule >= 2 ? 512 : 2048) && epi != MMTG_EPI_SPLIT && epi != MMTG_EPI_ROWDOT && epi != MMTG_EPI_DGELU) {
            persist = true;
            wide = false;
            a.dbg_flags |= 32;
        }
        // Single-stage kernel at four workgroups per CU (measured, profiles/r01_v6_gemm_per_shape.log):
        // weight gradients always (with the split counts of engine._wgrad_splits), forward / dgrad
        // products whose 128x128 tiles make more than one round of the 2-per-CU kernel (qkv 84 -> 68 us,
        // fc1+GELU 120 -> 103, dGELU 106 -> 94); the one-round 192x128 and the small-M cases keep theirs.
        static const bool env_no_occ4 = getenv("MMTG_GEMM_NO_OCC4") != nullptr;      // A/B switch for whole-step runs
        if (!(flags & MMTG_GEMM_NO_OCC4) && !env_no_occ4 && !persist && !skinny && !(flags & MMTG_GEMM_WIDE)) {
            const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
            if (transA || (flags & MMTG_GEMM_OCC4) || (!wide && t128 > 2L * num_cus())) a.dbg_flags |= 2;
        }
        if (wgrad && epi == MMTG_EPI_SPLIT) a.dbg_flags |= 2;      // the slab epilogue lives in the single-stage kernel only
        // Eight-phase kernel (round 2): every K-contiguous x K-contiguous product of a training step -- forward and dgrad of the
        // Conv1D / Linear layers through the [out,in] weight copies, the LM head -- with K a multiple of 128.  Row tiles of
        // 192 where they fill the 256 CUs better (a 192-row tile costs 0.75 of a 256-row one): N = 768 -> 237 tiles in one
        // round instead of 177, N = 2304 -> 2.78 rounds instead of 2.07.  MMTG_GEMM_P8=0 switches it off (A/B), MMTG_GEMM_BIG=1
        // keeps the plain 256x256 four-wave experiment.
        static const int big = getenv("MMTG_GEMM_BIG") ? atoi(getenv("MMTG_GEMM_BIG")) : 0;
        static const int p8 = getenv("MMTG_GEMM_P8") ? atoi(getenv("MMTG_GEMM_P8")) : 1;
        static const int p8_rows_env = getenv("MMTG_GEMM_P8_ROWS") ? atoi(getenv("MMTG_GEMM_P8_ROWS")) : 0;
        const bool nt_big = !transA && transB && !skinny && !wgrad && M >= 1024 && N >= 256 && splits == 1 && epi != MMTG_EPI_SPLIT &&
                            epi != MMTG_EPI_ATOMIC && !(flags & (MMTG_GEMM_WIDE | MMTG_GEMM_OCC4 | MMTG_GEMM_P256 | MMTG_GEMM_NO_P8)) &&
                            (!(flags & MMTG_GEMM_PERSIST) || (flags & MMTG_GEMM_P8));
        if (big == 1 && nt_big && epi != MMTG_EPI_ROWDOT && !(epi == MMTG_EPI_DGELU && aux2)) {
            wide = false; persist = false; a.dbg_flags &= ~(2 | 32); a.dbg_flags |= 64;
        } else if (p8 && nt_big && K % 128 == 0 && (epi != MMTG_EPI_DGELU || p8 >= 2 || (flags & MMTG_GEMM_P8))) {
            // (dGELU keeps the single-stage kernel: with the saved pre-activation read in its exposed epilogue the eight-phase
            //  kernel measured 115.7 vs 112.5 us inside the training step; MMTG_GEMM_P8=2 or the MMTG_GEMM_P8 flag routes it here too -- bit-equal)
            const long t256 = (long)cdiv(M, 256) * cdiv(N, 256), t192 = (long)cdiv(M, 192) * cdiv(N, 256), t288 = (long)cdiv(M, 288) * cdiv(N, 256);
            const long ncu = p8_cus();
            // Row tile from a measured cost model (in-kernel timelines, profiles/r04_*): a round of tiles costs a fixed ~5.6 us
            // (first K tile landing + epilogue + drain) plus K/64 K-tile steps of 1.41 us x rows / 256 with every CU in the loop
            // (the chip's power-limited MFMA rate, ~1450 TFLOP/s, whatever the tile) but never below the phase protocol's 1.15 us.
            // 288-row tiles (round 4) turn N = 2304 at M = 15104 from 3 rounds into 2 (477 tiles) and the LM head from 13 into 11.
            static const int p8_288 = getenv("MMTG_GEMM_P8_288") ? atoi(getenv("MMTG_GEMM_P8_288")) : 1;
            const double nkt = (double)K / 64.0;
            auto cost = [&](long tiles, double r) { const double tk = 1.41 * r / 256.0; return (double)cdiv(tiles, ncu) * (5.6 + nkt * (tk < 1.15 ? 1.15 : tk)); };
            const double c192 = cost(t192, 192), c256 = cost(t256, 256), c288 = cost(t288, 288);
            int rows = c192 < 0.97 * c256 ? 192 : 256;
            if (p8_288 && c288 < 0.95 * (rows == 192 ? c192 : c256)) rows = 288;
            if (flags & MMTG_GEMM_P8_288) rows = 288;
            if (p8_rows_env) rows = p8_rows_env;
            if (epi == MMTG_EPI_DGELU && aux2) rows = 256;      // the fused column sums need 64-row-aligned wave tiles
            // Persistent form (more than one round of tiles): OPT-IN, MMTG_GEMM_P8_PERSIST=1 or flags P8 | PERSIST.  Bit-equal, and
            // measured slower (profiles/r02_v5_gemm_eight_phase_persistent_ab.txt): qkv 62.7 -> 74.9 us, fc1 + GELU 95.9 -> 102.0,
            // LM head 316 -> 338, +0.2 ms per training step.  A workgroup that exits leaves its output stores to drain while the
            // next workgroup of that CU is already fetching its first tiles; a persistent one has them in its own vmcnt queue in
            // front of the next item's tiles, and every CU writes its 128-256 KB at the same moment.
            static const int p8_persist = getenv("MMTG_GEMM_P8_PERSIST") ? atoi(getenv("MMTG_GEMM_P8_PERSIST")) : 0;
            const bool want_persist = p8_persist || ((flags & MMTG_GEMM_P8) && (flags & MMTG_GEMM_PERSIST));
            if (want_persist && rows == 288) rows = 256;      // (no persistent 288-row instantiation)
            const long tiles = rows == 192 ? t192 : rows == 288 ? t288 : t256;
            if (want_persist && tiles > num_cus() && num_cus() % 8 == 0) rc = rows == 192 ? launch_p8p<192>(a, s) : launch_p8p<256>(a, s);
            else rc = rows == 192 ? launch_p8<192>(a, 1, s) : rows == 288 ? launch_p8<288>(a, 1, s) : launch_p8<256>(a, 1, s);
            if (rc) return rc;
            MMTG_LAUNCH_CHECK("gemm");
            return MMTG_OK;
        }
        // ... and its K-strided form for the weight gradients stored as K-split slabs: OPT-IN (MMTG_GEMM_P8T=1 or the MMTG_GEMM_P8
        // flag).  One round of tiles x splits <= CUs workgroups (engine._wgrad_splits_p8).  Measured (profiles/
        // r02_v5_gemm_eight_phase_tn.txt): the product alone 76 vs 82 us (fc, 7 vs 5 slabs), 64 vs 60 (qkv), with the slab sum 89
        // vs 88 / 77 vs 68, and +0.47 ms per training step -- its K tile takes 2.0 us against 1.25-1.4 us for the K-contiguous
        // form: 48 ds_read_b64_tr_b16 per wave and K tile do not fit under the other wave group's 16-MFMA phases, and with one
        // workgroup per CU the cold first tiles and the 256 KB fp32 slab store of every item are exposed.
        static const int p8t = getenv("MMTG_GEMM_P8T") ? atoi(getenv("MMTG_GEMM_P8T")) : 0;
        if ((p8t || (flags & MMTG_GEMM_P8)) && wgrad && epi == MMTG_EPI_SPLIT && K % 128 == 0 && M >= 256 && N >= 256 && !(flags & (MMTG_GEMM_NO_P8 | MMTG_GEMM_PERSIST))) {
            a.kper = cdiv(cdiv(K, splits), 128) * 128;
            rc = launch_p8<256, true>(a, splits, s);
            if (rc) return rc;
            MMTG_LAUNCH_CHECK("gemm");
            return MMTG_OK;
        }
        rc = launch_dma(a, transA, transB, splits, skinny && !transA && transB, wide, persist, s);
    }
    if (rc) return rc;
    MMTG_LAUNCH_CHECK("gemm");
    return MMTG_OK;
}

// ------------------------------------------------------------------ split-precision ("bf16x3") product, round 5
// C[M,N] (fp32) = epi( A B^T ) with A [M,K], B [N,K] both given as (hi | lo) pairs of bf16 planes of fp32 tensors:
// A B^T ~ A_hi B_hi^T + A_lo B_hi^T + A_hi B_lo^T, fp32 accumulate, on the eight-phase kernel (one 3K-deep loop).
// The reference's arithmetic is fp32 (/root/reference/src/model.py:279-288, the Conv1D / Linear products of GPT-2); this is
// the mode that keeps north_star's "logits within 1e-3, greedy ids bit-exact" at bf16 matrix-core speed.
extern "C" int mmtg_gemm_x3(int M, int N, int K, const void* A, long lda, long planeA, const void* B, long ldb, long planeB,
                            float* C, long ldc, void* planes, long ldp, long plane_out, const float* bias, int epi,
                            const float* aux, long ldaux, void* aux2, unsigned drop_thresh, unsigned drop_seed, int flags, void* stream) {
    MMTG_REQUIRE(M > 0 && N > 0 && K > 0 && A && B, "gemm_x3: empty problem or null operand");
    MMTG_REQUIRE(C || planes, "gemm_x3: needs an fp32 output, a plane-pair output, or both");
    MMTG_REQUIRE(K % 128 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_x3: K %% 128 == 0, N, lda, ldb %% 8 == 0 (K=%d N=%d)", K, N);
    MMTG_REQUIRE(MMTG_ALIGNED16(A) && MMTG_ALIGNED16(B) && MMTG_ALIGNED16(C) && MMTG_ALIGNED16(planes) && planeA % 8 == 0 && planeB % 8 == 0,
                 "gemm_x3: operands and plane distances must be 16-byte aligned");
    MMTG_REQUIRE(!C || ldc % 8 == 0, "gemm_x3: ldc %% 8 == 0");
    MMTG_REQUIRE(!planes || (ldp % 8 == 0 && plane_out % 8 == 0 && plane_out >= (long)(M - 1) * ldp + N), "gemm_x3: plane output: ldp, plane distance %% 8 == 0, planes disjoint");
    MMTG_REQUIRE(planeA >= (long)(M - 1) * lda + K && planeB >= (long)(N - 1) * ldb + K, "gemm_x3: an operand's lo plane must lie behind its hi plane");
    MMTG_REQUIRE(epi == MMTG_EPI_NONE || epi == MMTG_EPI_GELU || epi == MMTG_EPI_TANH || epi == MMTG_EPI_RESID || epi == MMTG_EPI_DGELU ||
                 epi == MMTG_EPI_DTANH || epi == MMTG_EPI_ROWDOT, "gemm_x3: epilogue %d is not built for the split-precision product", epi);
    if (epi == MMTG_EPI_RESID || epi == MMTG_EPI_DGELU || epi == MMTG_EPI_DTANH || epi == MMTG_EPI_ROWDOT)
        MMTG_REQUIRE(aux && ldaux % 8 == 0 && MMTG_ALIGNED16(aux), "gemm_x3: epilogue %d needs a 16-byte aligned fp32 aux with ldaux %% 8 == 0", epi);
    if (epi == MMTG_EPI_GELU) MMTG_REQUIRE(aux2 && MMTG_ALIGNED16(aux2) && ldc % 8 == 0 && ldc >= N, "gemm_x3: GELU needs aux2 (fp32 pre-activation, ld = ldc)");
    if (epi == MMTG_EPI_ROWDOT) MMTG_REQUIRE(aux2 && !bias && N % 64 == 0, "gemm_x3: ROWDOT needs aux2 (f32 [M, N/64]), no bias, N %% 64 == 0");
    MMTG_REQUIRE(!bias || MMTG_ALIGNED16(bias), "gemm_x3: bias must be 16-byte aligned");
    const long bytesA = (planeA + (long)(M - 1) * lda + K) * 2, bytesB = (planeB + (long)(N - 1) * ldb + K) * 2;
    MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L, "gemm_x3: an operand's plane pair must stay below 2 GiB");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.aux2 = aux2;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
    a.epi = epi; a.out_f32 = 1; a.use_tr = 1; a.alpha = 1.0f;
    a.trace = g_trace; a.trace_n = g_trace_n;
    a.gelu_grad = (flags & MMTG_GEMM_GELU_GRAD) ? 1 : 0;
    a.dbg_flags = (flags & MMTG_GEMM_ROW_ORDER) ? 1 : 0;
    a.bytesA = (int)bytesA; a.bytesB = (int)bytesB;
    a.planeA = (int)(planeA * 2); a.planeB = (int)(planeB * 2);
    a.planes = planes; a.ldp = ldp; a.plane_out = plane_out;
    a.aux2_bf16 = (flags & MMTG_GEMM_AUX2_BF16) ? 1 : 0;
    a.x3 = 1;
    a.kper = K;
    a.drop_thresh = drop_thresh; a.drop_seed = drop_seed;
    a.drop_inv_keep = drop_thresh ? (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh)) : 1.0f;
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: the fp32 tensors the product stands for (A, B, C once, + the epilogue's aux / second output)
    double alg_bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N);
    if (epi == MMTG_EPI_RESID || epi == MMTG_EPI_DGELU || epi == MMTG_EPI_DTANH || epi == MMTG_EPI_ROWDOT || epi == MMTG_EPI_GELU) alg_bytes += 4.0 * (double)M * N;
    ProfScope prof(MMTG_PROF_GEMM_BF16, s, 2.0 * M * N * (double)K, alg_bytes);
    // row tile: the cost model of mmtg_gemm with three times the K tiles
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256), t192 = (long)cdiv(M, 192) * cdiv(N, 256);
    MMTG_REQUIRE(t192 < (1L << 20), "gemm_x3: too many output tiles for the tile-order arithmetic (%ld)", t192);
    const long ncu = p8_cus();
    const double nkt = 3.0 * (double)K / 64.0;
    auto cost = [&](long tiles, double r) { const double tk = 1.41 * r / 256.0; return (double)cdiv(tiles, ncu) * (5.6 + nkt * (tk < 1.15 ? 1.15 : tk)); };
    int rows = cost(t192, 192) < 0.97 * cost(t256, 256) ? 192 : 256;
    static const int rows_env = getenv("MMTG_X3_P8_ROWS") ? atoi(getenv("MMTG_X3_P8_ROWS")) : 0;
    if (rows_env == 192 || rows_env == 256) rows = rows_env;
    if (epi == MMTG_EPI_DGELU && aux2) rows = 256;      // the fused column sums need 64-row-aligned wave tiles
    const int rc = rows == 192 ? launch_p8<192, false, true>(a, 1, s) : launch_p8<256, false, true>(a, 1, s);
    if (rc) return rc;
    MMTG_LAUNCH_CHECK("gemm_x3");
    return MMTG_OK;
}

