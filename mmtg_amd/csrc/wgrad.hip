// Grouped weight-gradient products with an in-kernel, deterministic split-K reduction (round 3).
//
//   C_p[M_p, N_p] (fp32) (+)= A_p^T B_p      A_p [K, lda_p], B_p [K, ldb_p] bf16, K = tokens, p = 0 .. n-1
//
// The weight gradients of one GPT-2 block -- c_fc, mlp.c_proj, attn.c_proj, c_attn (reference call sites: the
// autograd of the Conv1D layers behind /root/reference/src/model.py:282-288) -- used to be four launches, each
// split 5 / 5 / 12 / 7 ways over K to fill the chip, each followed by an ordered sum of its fp32 slabs
// (mmtg_slab_sum): 172 MB of slab stores + the same read back per block, 2 GB per step, 100 launches.  Here they
// are ONE launch: 432 tiles x 2 K halves = 864 workgroups in a single round of the 1024 slots (four per CU), a
// K loop of 118 tiles per workgroup instead of 19-47 (the exposed prologue / epilogue of a workgroup is paid once
// per 118 tiles), and the reduction happens in the kernel:
//
//   * K split s of tile t stores its raw fp32 partial tile into slot (t, s) of a tile-contiguous workspace
//     (every wave-instruction writes 1 KB of consecutive addresses, whatever the gradient's own leading dimension);
//   * each WAVE then bumps the arrival counter of its 64x64 quadrant (release / acquire fences at agent scope:
//     the per-XCD L2s are not coherent with each other);
//   * the wave that arrives LAST adds the S partial quadrants in split order -- its own from registers -- and writes
//     (or accumulates into) the gradient through a wave-private LDS image with 32-byte row segments.
//
// Nobody ever waits for anybody, so nothing depends on the workgroups being co-resident (an RCCL kernel may hold
// CUs beside this one); the sum order is the split order whoever arrives last, so results are bit-reproducible.
// The last arriver resets its counter: the counter buffer is zero before and after every launch.
//
// Main loop = the single-stage, four-workgroups-per-CU kernel of gemm.hip (gemm_occ4_kernel<true, true>): one
// 32 KB LDS stage per 128x128 workgroup filled by LDS-DMA, K-strided images read with ds_read_b64_tr_b16.
#include "gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int WG_MAXP = 8;

struct WgProb {
    const void* A; const void* B; float* C;
    long lda, ldb, ldc;
    int M, N, tiles_m, tiles_n, tile0, m_fast;
    int bytesA, bytesB;
    int planeA, planeB;    // x3: bytes from the hi plane to the lo plane of an operand
};
struct WgArgs {
    WgProb pr[WG_MAXP];
    int n, K, kper, splits, ntiles, accumulate;
    int touch;             // MMTG_WGRAD_TOUCH: K tiles of L2 touch-prefetch lead (0 = off)
    float* ws;
    unsigned* cnt;
};

// final tile of one wave (64x64 quadrant at rows mw0.., columns nw0..) from per-band sums: LDS-staged 32-byte row stores
__device__ __forceinline__ void wg_store_band(float* C, long ldc, int M, int N, int mw0, int nw0, int h, const f32x4 (&v)[4],
                                              char* lds, int lane, bool accumulate) {
    const int g = lane >> 4, l15 = lane & 15;
    const int col = (lane & 7) * 8, n = nw0 + col, r0 = lane >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(lds + l15 * 256 + (((j * 4 + g) ^ l15) << 4)) = v[j];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = q * 8 + r0, m = mw0 + h * 16 + row;
        const int c0 = 2 * (lane & 7);
        f32x4 lo = *reinterpret_cast<const f32x4*>(lds + row * 256 + ((c0 ^ row) << 4));
        f32x4 hi = *reinterpret_cast<const f32x4*>(lds + row * 256 + (((c0 + 1) ^ row) << 4));
        if (m < M && n < N) {
            float* dst = C + (long)m * ldc + n;
            if (accumulate) {
                lo += *reinterpret_cast<const f32x4*>(dst);
                hi += *reinterpret_cast<const f32x4*>(dst + 4);
            }
            *reinterpret_cast<f32x4*>(dst) = lo;
            *reinterpret_cast<f32x4*>(dst + 4) = hi;
        }
    }
}

// Partial tiles cross XCDs (the K halves of a tile run on different XCDs, whose L2s are not coherent with each other).
// A release / acquire fence pair at agent scope costs an L2-wide write-back + invalidate PER WAVE (buffer_wbl2 / buffer_inv:
// measured +180 us on the 864-workgroup launch, +500 us once a second round of workgroups has its operands invalidated under
// it).  Instead the partial stores and loads themselves carry the agent-scope bit (sc1: write through / miss always), which
// is what the memory model prescribes for agent-scope atomic stores and loads; the store acknowledgements (vmcnt) order them
// before the counter increment.  FENCE = true keeps the fence pair (A/B, MMTG_WGRAD_FENCE=1).
__device__ __forceinline__ void st4_agent(float* p, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld4x4_agent(const float* p, f32x4 (&x)[4]) {      // four 16-byte loads 1 KB apart, then wait
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p) : "memory");
}


// Epilogue of both kernels, per WAVE (wave tile = TM bands of 16 rows x 64 columns at rows mw0.., columns nw0..): store / reduce /
// write as described at the top of the file.  `scratch`: 4 KB of wave-private LDS.  TILE = elements of a workgroup tile.
template <bool FENCE, int TM, int TILE>
__device__ __forceinline__ void wg_reduce_store(const WgArgs& a, float* C, long ldc, int M, int N, int mw0, int nw0, int t, int split,
                                                int nwaves, int wave, int lane, f32x4 (&acc)[TM][4], char* scratch) {
    const int S = a.splits;
    if (S == 1) {
#pragma unroll
        for (int h = 0; h < TM; ++h) wg_store_band(C, ldc, M, N, mw0, nw0, h, acc[h], scratch, lane, a.accumulate != 0);
        return;
    }
    // my partial wave tile -> slot (t, split) of the workspace, accumulator order (1 KB per wave-instruction)
    float* const slot0 = a.ws + ((long)t * S) * TILE + wave * (TM * 1024) + lane * 4;
    {
        float* const mine = slot0 + (long)split * TILE;
#pragma unroll
        for (int h = 0; h < TM; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (FENCE) *reinterpret_cast<f32x4*>(mine + (h * 4 + j) * 256) = acc[h][j];
                else st4_agent(mine + (h * 4 + j) * 256, acc[h][j]);
            }
    }
    if constexpr (FENCE) __threadfence();          // release: my stores are visible at device scope before the count moves
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... every write-through store has been acknowledged
    unsigned* const cnt = a.cnt + (long)t * nwaves + wave;
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(cnt, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(S - 1)) return;          // not the last arriver of this wave tile: done
    if constexpr (FENCE) __threadfence();          // acquire: the other splits' stores
    if (lane == 0) *cnt = 0u;                      // leave the counters zeroed for the next launch
#pragma unroll
    for (int h = 0; h < TM; ++h) {
        f32x4 sum[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {              // split order, my own share from registers
            if (s == split) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[j] += acc[h][j];
            } else {
                const float* src = slot0 + (long)s * TILE + h * 1024;
                f32x4 x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if constexpr (FENCE) x[j] = *reinterpret_cast<const f32x4*>(src + j * 256);
                if constexpr (!FENCE) ld4x4_agent(src, x);
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[j] += x[j];
            }
        }
        wg_store_band(C, ldc, M, N, mw0, nw0, h, sum, scratch, lane, a.accumulate != 0);
    }
}

// item -> (problem, tile origin, split): XCD-contiguous order (gemm_common.h, tile_origin) over v = split * ntiles + tile, so an XCD
// works inside one K slice and the tiles of an XCD's run share operand panels in its L2
template <int TB>
__device__ __forceinline__ int wg_locate(const WgArgs& a, int& t, int& split, int& m0, int& n0) {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int v = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    split = v / a.ntiles;
    t = v - split * a.ntiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < WG_MAXP; ++i)
        if (i < a.n && t >= a.pr[i].tile0) pi = i;
    const WgProb& P = a.pr[pi];
    const int tl = t - P.tile0;
    if (P.m_fast) { m0 = (tl % P.tiles_m) * TB; n0 = (tl / P.tiles_m) * TB; }
    else { m0 = (tl / P.tiles_n) * TB; n0 = (tl % P.tiles_n) * TB; }
    return pi;
}

// ABLATE (timing only, wrong results; MMTG_WGRAD_ABLATE=1|2): 1 = every fragment by ONE ds_read_b128 instead of two
// ds_read_b64_tr_b16 (what an 8-row register transpose of plain reads would issue); 2 = no LDS-DMA fills after the first tile.
// X3 (round 5, the split-precision mode): A_p and B_p are (hi | lo) bf16 plane pairs of fp32 activations / gradients and the K
// slice is walked three times -- (A hi, B hi), (A lo, B hi), (A hi, B lo) -- into the same accumulators (gemm.hip, gemm_p8_kernel).
// X3C (round 5, late): the x3 form with ONE K loop over combined stages -- a stage holds the K tile of all four planes (A hi | A lo |
// B hi | B lo, 64 KB: two workgroups per CU instead of four) and every fragment pair feeds three MFMAs -- instead of three passes that
// re-stage A hi and B hi: two thirds of the bytes through the CU's L2 -> LDS port (what bounds this kernel), a third of the barriers.
template <bool FENCE, int ABLATE = 0, bool X3 = false, bool X3C = false>
__global__ __launch_bounds__(256, (X3C ? 2 : 4)) void wgrad_group_kernel(WgArgs a) {
    static_assert(!X3C || X3, "combined stages belong to the x3 form");
    constexpr int TBM = 128, TBN = 128, NW = 4, TM = 4, TN = 4, BK = 64;
    constexpr int NB = TBM / 8 / NW;
    constexpr int TA = TBM * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int t, split, m0, n0;
    const WgProb& P = a.pr[wg_locate<128>(a, t, split, m0, n0)];
    const int M = P.M, N = P.N;
    const long lda = P.lda, ldb = P.ldb;
    const int kbeg = split * a.kper;
    const int klen = max(0, min(a.K, kbeg + a.kper) - kbeg);
    const int nk = (klen + BK - 1) / BK, nk_full = klen / BK;
    int oa[TM], ob[TN];
    ks_offsets<TBM, TM>(wm * 64, lane, oa);
    ks_offsets<TBN, TN>(wn * 64, lane, ob);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.A), 0, P.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.B), 0, P.bytesB, 0x00020000);
    const int sa_base = (int)(((long)kbeg * lda + m0) * 2);
    const int sb_base = (int)(((long)kbeg * ldb + n0) * 2);
    int sa = sa_base, sb = sb_base;
    const int stepa = (int)((long)BK * lda * 2), stepb = (int)((long)BK * ldb * 2);
    int va[NB], vb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        va[i] = dma_voff<true, TBM>(lda, m0, M, BK, wave + NW * i, lane);
        vb[i] = dma_voff<true, TBN>(ldb, n0, N, BK, wave + NW * i, lane);
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // TOUCH (ABLATE == 3, MMTG_WGRAD_TOUCH=P): every thread requests ONE dword of one 128-byte line of the tile P steps ahead (waves
    // 0-1: the 64 k-rows x 2 lines of the A slice, waves 2-3: of the B slice) into a register nobody reads -- the line is in the L2
    // when the LDS-DMA of that tile asks for it.  A quarter of this launch's fill bytes are compulsory L2 misses (a K slice of 7552
    // tokens against 3-24 tiles per operand panel), and a miss costs its whole latency: the single-stage loop has nothing else in
    // flight.  The request is the youngest of its wave at the tile wait, so that wait becomes vmcnt(1); returns are in order, so
    // the hipcc-tracked load is consumed one tile later, behind that tile's DMA requests (a counted wait).
    int touch_sink = 0;
    const bool t_b = wave >= 2;
    const int t_row = (tid & 127) >> 1, t_half = tid & 1;
    const int t_ok = t_b ? (n0 + t_half * 64 < N) : (m0 + t_half * 64 < M);
    const int t_v = (int)((long)t_row * (t_b ? ldb : lda) * 2) + t_half * 128;
    if constexpr (X3C) {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt) __builtin_amdgcn_s_barrier();          // every wave is done reading the previous stage
            const bool full = kt < nk_full;
            const int krem = klen - kt * BK;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int o = full ? va[i] : dma_voff<true, TBM>(lda, m0, M, krem, wave + NW * i, lane);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, smem + (wave + NW * i) * 1024), 16, o, sa, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, smem + TA + (wave + NW * i) * 1024), 16, o, sa + P.planeA, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int o = full ? vb[i] : dma_voff<true, TBN>(ldb, n0, N, krem, wave + NW * i, lane);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, smem + 2 * TA + (wave + NW * i) * 1024), 16, o, sb, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, smem + 3 * TA + (wave + NW * i) * 1024), 16, o, sb + P.planeB, 0, 0);
            }
            sa += stepa;
            sb += stepb;
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                  // the stage is complete
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int o = oa[i] + kk * 32 * 2 * TBM;
                    fah[i] = tr_read_pair(smem, o, o + 4 * 2 * TBM);
                    fal[i] = tr_read_pair(smem + TA, o, o + 4 * 2 * TBM);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int o = ob[j] + kk * 32 * 2 * TBN;
                    fbh[j] = tr_read_pair(smem + 2 * TA, o, o + 4 * 2 * TBN);
                    fbl[j] = tr_read_pair(smem + 3 * TA, o, o + 4 * 2 * TBN);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        mma16(fbh[j], fah[i], acc[i][j]);
                        mma16(fbh[j], fal[i], acc[i][j]);
                        mma16(fbl[j], fah[i], acc[i][j]);
                    }
            }
        }
    } else
    for (int pass = 0; pass < (X3 ? 3 : 1); ++pass) {
    if constexpr (X3) {
        sa = sa_base + (pass == 1 ? P.planeA : 0);
        sb = sb_base + (pass == 2 ? P.planeB : 0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        if (kt || pass) __builtin_amdgcn_s_barrier();      // every wave is done reading the previous tile
        const bool full = kt < nk_full;
        const int krem = klen - kt * BK;
        if (ABLATE != 2 || kt == 0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int o = full ? va[i] : dma_voff<true, TBM>(lda, m0, M, krem, wave + NW * i, lane);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, smem + (wave + NW * i) * 1024), 16, o, sa, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int o = full ? vb[i] : dma_voff<true, TBN>(ldb, n0, N, krem, wave + NW * i, lane);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, smem + TA + (wave + NW * i) * 1024), 16, o, sb, 0, 0);
        }
        }
        if constexpr (ABLATE == 3) {
            const int tp = a.touch;
            const int vo = (t_ok && (kt + tp) * BK + t_row < klen) ? t_v : OOB;
            const int so = __builtin_amdgcn_readfirstlane(t_b ? sb + tp * stepb : sa + tp * stepa);
            asm volatile("" :: "v"(touch_sink));           // the previous tile's touch is consumed HERE (a counted wait behind this tile's DMA requests)
            if (t_b) touch_sink = __builtin_amdgcn_raw_buffer_load_b32(rb, vo, so, 0);
            else touch_sink = __builtin_amdgcn_raw_buffer_load_b32(ra, vo, so, 0);
        }
        sa += stepa;
        sb += stepb;
        if constexpr (ABLATE == 3) wait_vmcnt<1>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // the tile is complete
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (ABLATE == 1) fa[i] = *reinterpret_cast<const bf16x8*>(smem + ((oa[i] + kk * 8192) & ~15));
                else fa[i] = tr_read_pair(smem, oa[i] + kk * 32 * 2 * TBM, oa[i] + kk * 32 * 2 * TBM + 4 * 2 * TBM);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (ABLATE == 1) fb[j] = *reinterpret_cast<const bf16x8*>(smem + TA + ((ob[j] + kk * 8192) & ~15));
                else fb[j] = tr_read_pair(smem + TA, ob[j] + kk * 32 * 2 * TBN, ob[j] + kk * 32 * 2 * TBN + 4 * 2 * TBN);
            }
            // (swapped operands: a lane then holds 4 consecutive columns of row l15 -- row-contiguous stores)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma16(fb[j], fa[i], acc[i][j]);
        }
    }
    }
    if constexpr (ABLATE == 3) { wait_vmcnt<0>(); asm volatile("" :: "v"(touch_sink)); }
    __builtin_amdgcn_s_barrier();                  // the stage becomes the waves' private epilogue scratch
    wg_reduce_store<FENCE, TM, TBM * TBN>(a, P.C, P.ldc, M, N, m0 + wm * 64, n0 + wn * 64, t, split, NW, wave, lane, acc,
                                          smem + wave * 4096);
}


// ------------------------------------------------------------------ 256x256 eight-phase form (one workgroup per CU)
// The same grouped launch on the eight-phase K-strided main loop of gemm.hip (gemm_p8_kernel<256, true>, restated here because
// its phase / restaging protocol is a macro schedule inside that kernel): a 512-thread workgroup computes a 256x256 tile,
// wave (wr, wc) of the 2x4 grid owns 128x64 of it; a 64-deep K tile is consumed in four phases of 16 MFMAs, the two wave-row
// groups one barrier apart; two 64 KB stages, every K tile as four 16 KB half-tile images (64 k-rows x 128 gathered columns,
// fragments by ds_read_b64_tr_b16) restaged three half tiles ahead of their use, one counted vmcnt(6) per K tile.  Half the
// LDS fill bytes per FLOP of the 128x128 kernel (the fill path bounds that one: four workgroups pull 4 x 32 KB per K step
// through the CU's 52 B/clk L2 port).  As a per-product kernel it lost (gemm.hip: 7-24 slabs per product, a 256 KB slab
// store and a cold prologue exposed per 17-34 K tiles); grouped, a workgroup runs 118 K tiles and 108 tiles x 2 K halves =
// 216 workgroups fill 84 % of the CUs in one round.  Rows of K past the end of a slice are requested out of range per lane (zeros).
__device__ __forceinline__ void g8_issue(__amdgpu_buffer_rsrc_t r, char* d0, char* d1, int v0, int v1, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, d0), 16, v0, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(void, d1), 16, v1, soff, 0, 0);
}
template <int AH, int BH>
__device__ __forceinline__ void g8_mfma(const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], f32x4 (&acc)[8][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma16(fb[j][kk], fa[i][kk], acc[4 * AH + i][2 * BH + j]);
    __builtin_amdgcn_s_setprio(0);
}
// lane offset of 1-KB block `blk` (4 k-rows x 256 B) of half-tile image h: image chunk c holds source columns
// (c / GRPCH) * 2 * GRPCH * 8 + h * GRPCH * 8 + (c % GRPCH) * 8 (GRPCH = 8: A, two wave-row groups of 64; 4: B, four wave columns of 32)
template <int GRPCH>
__device__ __forceinline__ int g8_voff(long ld, int col0, int ncols, int h, int blk, int lane) {
    const int k = blk * 4 + (lane >> 4), pc = lane & 15, c = pc ^ ks_swz(k);
    const int col = (c / GRPCH) * (2 * GRPCH * 8) + h * (GRPCH * 8) + (c % GRPCH) * 8;
    return (col0 + col < ncols) ? (int)(((long)k * ld + col) * 2) : OOB;
}
#define G8_SYNC_IN()  do { asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_SYNC_OUT() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); } while (0)

__global__ __launch_bounds__(512, 1) void wgrad_group_p8_kernel(WgArgs a) {
    constexpr int TA = 256 * 128, STAGE = 2 * TA;                   // bytes: A images | B images, two 16 KB half tiles each
    extern __shared__ __attribute__((aligned(16))) char smem[];    // two stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int t, split, m0, n0;
    const WgProb& P = a.pr[wg_locate<256>(a, t, split, m0, n0)];
    const int M = P.M, N = P.N;
    const long lda = P.lda, ldb = P.ldb;
    const int kbeg = split * a.kper;                                // kper % 128 == 0
    const int klen = max(0, min(a.K, kbeg + a.kper) - kbeg);
    const int nk = 2 * ((klen + 127) >> 7);                         // K tiles of 64, an even number (rows past K read zeros)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.A), 0, P.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.B), 0, P.bytesB, 0x00020000);
    const int sa0 = (int)(((long)kbeg * lda + m0) * 2), sb0 = (int)(((long)kbeg * ldb + n0) * 2);
    const int stepa = (int)(64 * lda * 2), stepb = (int)(64 * ldb * 2);
    const int lA00 = wave * 1024, lA01 = (wave + 8) * 1024, lA10 = 16384 + lA00, lA11 = 16384 + lA01;
    const int lB00 = TA + lA00, lB01 = TA + lA01, lB10 = TA + lA10, lB11 = TA + lA11;
    const int vA00 = g8_voff<8>(lda, m0, M, 0, wave, lane), vA01 = g8_voff<8>(lda, m0, M, 0, wave + 8, lane);
    const int vA10 = g8_voff<8>(lda, m0, M, 1, wave, lane), vA11 = g8_voff<8>(lda, m0, M, 1, wave + 8, lane);
    const int vB00 = g8_voff<4>(ldb, n0, N, 0, wave, lane), vB01 = g8_voff<4>(ldb, n0, N, 0, wave + 8, lane);
    const int vB10 = g8_voff<4>(ldb, n0, N, 1, wave, lane), vB11 = g8_voff<4>(ldb, n0, N, 1, wave + 8, lane);
    // k-row (inside a K tile) of this lane in its two DMA blocks: rows at or past the end of the K slice are requested out of
    // range PER LANE (the row offset of a K tile travels in the scalar offset, which the descriptor's bounds check may not
    // cover -- a ragged last slice must not depend on it)
    const int kl0 = wave * 4 + (lane >> 4), kl1 = kl0 + 32;
#define G8_A0(t_) g8_issue(ra, smem + ((t_) & 1) * STAGE + lA00, smem + ((t_) & 1) * STAGE + lA01, (t_) * 64 + kl0 < klen ? vA00 : OOB, (t_) * 64 + kl1 < klen ? vA01 : OOB, sa0 + (t_) * stepa)
#define G8_A1(t_) g8_issue(ra, smem + ((t_) & 1) * STAGE + lA10, smem + ((t_) & 1) * STAGE + lA11, (t_) * 64 + kl0 < klen ? vA10 : OOB, (t_) * 64 + kl1 < klen ? vA11 : OOB, sa0 + (t_) * stepa)
#define G8_B0(t_) g8_issue(rb, smem + ((t_) & 1) * STAGE + lB00, smem + ((t_) & 1) * STAGE + lB01, (t_) * 64 + kl0 < klen ? vB00 : OOB, (t_) * 64 + kl1 < klen ? vB01 : OOB, sb0 + (t_) * stepb)
#define G8_B1(t_) g8_issue(rb, smem + ((t_) & 1) * STAGE + lB10, smem + ((t_) & 1) * STAGE + lB11, (t_) * 64 + kl0 < klen ? vB10 : OOB, (t_) * 64 + kl1 < klen ? vB11 : OOB, sb0 + (t_) * stepb)
    int oa[4], ob[2];
    ks_offsets<128, 4>(wr * 64, lane, oa);
    ks_offsets<128, 2>(wc * 32, lane, ob);
    ob[0] += TA; ob[1] += TA;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    G8_B0(0); G8_A0(0); G8_B1(0); G8_A1(0);
    G8_B0(1); G8_A0(1); G8_B1(1);
    wait_vmcnt<6>();                       // K tile 0 has landed (mine; the barrier makes it everyone's)
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");          // the second wave group runs one barrier behind the first

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define G8_RD_A(ST, AH)                                                                                    \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < 4; ++i)          \
        fa[i][kk] = tr_read_pair(smem + (ST) * STAGE + (AH) * 16384, oa[i] + kk * 8192, oa[i] + kk * 8192 + 1024)
#define G8_RD_B(ST, BH, F)                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
        F[j][kk] = tr_read_pair(smem + (ST) * STAGE + (BH) * 16384, ob[j] + kk * 8192, ob[j] + kk * 8192 + 1024)
#define G8_TILE(ST, kt)                                                                                    \
    do {                                                                                                   \
        G8_RD_B(ST, 0, fb0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        G8_RD_A(ST, 0);                                                                                    \
        G8_A1((kt) + 1);                                                                                   \
        /* the b0 reads are done: b0 may be restaged next phase (8 + 16 reads, the counter saturates at 15) */ \
        asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");                                                \
        G8_SYNC_IN(); g8_mfma<0, 0>(fa, fb0, acc); G8_SYNC_OUT();                                           \
        G8_RD_B(ST, 1, fb1);                                                                               \
        G8_B0((kt) + 2);                                                                                   \
        G8_SYNC_IN(); g8_mfma<0, 1>(fa, fb1, acc); G8_SYNC_OUT();                                           \
        G8_RD_A(ST, 1);                                                                                    \
        G8_A0((kt) + 2);                                                                                   \
        G8_SYNC_IN(); g8_mfma<1, 1>(fa, fb1, acc); G8_SYNC_OUT();                                           \
        G8_B1((kt) + 2);                                                                                   \
        wait_vmcnt<6>();                                       /* K tile kt + 1 has landed */               \
        G8_SYNC_IN(); g8_mfma<1, 0>(fa, fb0, acc); G8_SYNC_OUT();                                           \
    } while (0)
    for (int kt = 0; kt < nk; kt += 2) {
        G8_TILE(0, kt);
        G8_TILE(1, kt + 1);
    }
#undef G8_TILE
#undef G8_RD_A
#undef G8_RD_B
#undef G8_A0
#undef G8_A1
#undef G8_B0
#undef G8_B1
    wait_vmcnt<0>();                                     // the zero-fill tail loads
    if (wr == 0) asm volatile("s_barrier" ::: "memory");           // re-join the two groups
    asm volatile("s_barrier" ::: "memory");                        // every wave is done with the stages: they become epilogue scratch
    wg_reduce_store<false, 8, 256 * 256>(a, P.C, P.ldc, M, N, m0 + wr * 128, n0 + wc * 64, t, split, 8, wave, lane, acc, smem + wave * 4096);
}

}  // namespace

extern "C" int mmtg_wgrad_group(int config, int n, const mmtg_wgrad_problem* probs, int K, int splits, float* ws, long ws_floats,
                                unsigned* counters, long n_counters, int accumulate, void* stream) {
    const bool x3 = (config & 2) != 0;             // split-precision operands (plane pairs), 128x128 tiles only
    const bool x3c = x3 && (config & 4) != 0;      // ... on combined stages (all four planes of a K tile per stage, two workgroups per CU)
    config &= ~6;
    MMTG_REQUIRE(config == 0 || (config == 1 && !x3),
                 "wgrad_group: config 0 (128x128 tiles), 1 (256x256 eight-phase tiles), 2 (128x128 tiles, x3 operands) or 6 (x3, combined stages)");
    const int TB = config ? 256 : 128, NWV = config ? 8 : 4, KQ = config ? 128 : 64;
    MMTG_REQUIRE(n >= 1 && n <= WG_MAXP && probs, "wgrad_group: 1..%d problems", WG_MAXP);
    MMTG_REQUIRE(K > 0 && splits >= 1, "wgrad_group: K and splits must be positive");
    WgArgs a;
    memset(&a, 0, sizeof(a));
    int tiles = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        const mmtg_wgrad_problem& q = probs[i];
        MMTG_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0, "wgrad_group: problem %d has a null operand or an empty shape", i);
        MMTG_REQUIRE(MMTG_ALIGNED16(q.A) && MMTG_ALIGNED16(q.B) && MMTG_ALIGNED16(q.C), "wgrad_group: operands must be 16-byte aligned");
        MMTG_REQUIRE(q.lda % 8 == 0 && q.ldb % 8 == 0 && q.M % 8 == 0 && q.N % 8 == 0 && q.ldc % 4 == 0 && q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N,
                     "wgrad_group: problem %d: M, N, lda, ldb multiples of 8, ldc of 4, leading dimensions >= extents", i);
        if (x3) MMTG_REQUIRE(q.planeA % 8 == 0 && q.planeB % 8 == 0 && q.planeA >= (long)(K - 1) * q.lda + q.M && q.planeB >= (long)(K - 1) * q.ldb + q.N,
                             "wgrad_group: problem %d: an operand's lo plane must lie behind its hi plane (distance %% 8 == 0)", i);
        const long bytesA = ((x3 ? q.planeA : 0) + (long)(K - 1) * q.lda + q.M) * 2, bytesB = ((x3 ? q.planeB : 0) + (long)(K - 1) * q.ldb + q.N) * 2;
        MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L, "wgrad_group: operands must stay below 2 GiB");
        WgProb& p = a.pr[i];
        p.A = q.A; p.B = q.B; p.C = q.C; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.M = q.M; p.N = q.N;
        p.tiles_m = cdiv(q.M, TB); p.tiles_n = cdiv(q.N, TB);
        p.tile0 = tiles;
        // the squarer block of the tile grid per XCD run (see launch_dma_cfg in gemm.hip)
        p.m_fast = p.tiles_n > p.tiles_m;
        p.bytesA = (int)bytesA; p.bytesB = (int)bytesB;
        p.planeA = x3 ? (int)(q.planeA * 2) : 0; p.planeB = x3 ? (int)(q.planeB * 2) : 0;
        tiles += p.tiles_m * p.tiles_n;
        flops += 2.0 * q.M * q.N * (double)K;
        bytes += 2.0 * ((double)q.M * K + (double)q.N * K) + 4.0 * (double)q.M * q.N;
    }
    const int kper = cdiv(cdiv(K, splits), KQ) * KQ;
    splits = cdiv(K, kper);                        // no empty K slice: every slot a last arriver reads was written
    MMTG_REQUIRE(splits == 1 || (ws && counters && MMTG_ALIGNED16(ws)), "wgrad_group: split products need the workspace and the counters");
    MMTG_REQUIRE(splits == 1 || (ws_floats >= (long)tiles * splits * TB * TB && n_counters >= (long)tiles * NWV),
                 "wgrad_group: workspace needs %ld floats and %ld counters", (long)tiles * splits * TB * TB, (long)tiles * NWV);
    a.n = n; a.K = K; a.kper = kper; a.splits = splits; a.ntiles = tiles; a.accumulate = accumulate;
    a.ws = ws; a.cnt = counters;
    static const int touch = getenv("MMTG_WGRAD_TOUCH") ? atoi(getenv("MMTG_WGRAD_TOUCH")) : 0;
    a.touch = touch;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_GEMM_BF16, s, flops, bytes);
    static bool attr_done = false;
    static const bool fence = getenv("MMTG_WGRAD_FENCE") != nullptr;
    if (config == 1) {
        const size_t shm8 = 2 * (256 + 256) * 128;
        static bool attr8_done = false;
        if (!attr8_done) {
            if (hipFuncSetAttribute((const void*)wgrad_group_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm8) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "wgrad_group: cannot raise dynamic LDS to %zu bytes", shm8);
            attr8_done = true;
        }
        hipLaunchKernelGGL(wgrad_group_p8_kernel, dim3(tiles * splits), dim3(512), shm8, s, a);
        MMTG_LAUNCH_CHECK("wgrad_group");
        return MMTG_OK;
    }
    const size_t shm = (128 + 128) * 128;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)wgrad_group_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_group_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "wgrad_group: cannot raise dynamic LDS to %zu bytes", shm);
        attr_done = true;
    }
    static const int ablate = getenv("MMTG_WGRAD_ABLATE") ? atoi(getenv("MMTG_WGRAD_ABLATE")) : 0;
    if (x3c) {        // combined stages: A hi | A lo | B hi | B lo
        const size_t shmc = 2 * shm;
        static bool x3c_done = false;
        if (!x3c_done) {
            if (hipFuncSetAttribute((const void*)wgrad_group_kernel<false, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmc) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "wgrad_group: cannot raise dynamic LDS to %zu bytes", shmc);
            x3c_done = true;
        }
        hipLaunchKernelGGL((wgrad_group_kernel<false, 0, true, true>), dim3(tiles * splits), dim3(256), shmc, s, a);
    } else if (x3) {
        static bool x3_done = false;
        if (!x3_done) {
            if (hipFuncSetAttribute((const void*)wgrad_group_kernel<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                MMTG_FAIL(MMTG_ERR_HIP, "wgrad_group: cannot raise dynamic LDS to %zu bytes", shm);
            x3_done = true;
        }
        hipLaunchKernelGGL((wgrad_group_kernel<false, 0, true>), dim3(tiles * splits), dim3(256), shm, s, a);
    } else if (ablate) {
        static bool abl_done = false;
        if (!abl_done) {
            (void)hipFuncSetAttribute((const void*)wgrad_group_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            (void)hipFuncSetAttribute((const void*)wgrad_group_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            abl_done = true;
        }
        if (ablate == 1) hipLaunchKernelGGL((wgrad_group_kernel<false, 1>), dim3(tiles * splits), dim3(256), shm, s, a);
        else hipLaunchKernelGGL((wgrad_group_kernel<false, 2>), dim3(tiles * splits), dim3(256), shm, s, a);
    } else if (fence) hipLaunchKernelGGL(wgrad_group_kernel<true>, dim3(tiles * splits), dim3(256), shm, s, a);
    else if (touch > 0) {
        static bool t_done = false;
        if (!t_done) { (void)hipFuncSetAttribute((const void*)wgrad_group_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); t_done = true; }
        hipLaunchKernelGGL((wgrad_group_kernel<false, 3>), dim3(tiles * splits), dim3(256), shm, s, a);
    } else hipLaunchKernelGGL(wgrad_group_kernel<false>, dim3(tiles * splits), dim3(256), shm, s, a);
    MMTG_LAUNCH_CHECK("wgrad_group");
    return MMTG_OK;
}
