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
};
struct WgArgs {
    WgProb pr[WG_MAXP];
    int n, K, kper, splits, ntiles, accumulate;
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

template <bool FENCE>
__global__ __launch_bounds__(256, 4) void wgrad_group_kernel(WgArgs a) {
    constexpr int TBM = 128, TBN = 128, NW = 4, TM = 4, TN = 4, BK = 64;
    constexpr int NB = TBM / 8 / NW;
    constexpr int TA = TBM * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-contiguous item order (gemm_common.h, tile_origin): v = split * ntiles + tile, so an XCD works inside one K slice
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int v = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int split = v / a.ntiles;
    const int t = v - split * a.ntiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < WG_MAXP; ++i)
        if (i < a.n && t >= a.pr[i].tile0) pi = i;
    const WgProb& P = a.pr[pi];
    const int tl = t - P.tile0;
    const int M = P.M, N = P.N;
    const long lda = P.lda, ldb = P.ldb;
    int m0, n0;
    if (P.m_fast) { m0 = (tl % P.tiles_m) * TBM; n0 = (tl / P.tiles_m) * TBN; }
    else { m0 = (tl / P.tiles_n) * TBM; n0 = (tl % P.tiles_n) * TBN; }
    const int kbeg = split * a.kper;
    const int klen = max(0, min(a.K, kbeg + a.kper) - kbeg);
    const int nk = (klen + BK - 1) / BK, nk_full = klen / BK;
    int oa[TM], ob[TN];
    ks_offsets<TBM, TM>(wm * 64, lane, oa);
    ks_offsets<TBN, TN>(wn * 64, lane, ob);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.A), 0, P.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.B), 0, P.bytesB, 0x00020000);
    int sa = (int)(((long)kbeg * lda + m0) * 2);
    int sb = (int)(((long)kbeg * ldb + n0) * 2);
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
    for (int kt = 0; kt < nk; ++kt) {
        if (kt) __builtin_amdgcn_s_barrier();      // every wave is done reading the previous tile
        const bool full = kt < nk_full;
        const int krem = klen - kt * BK;
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
        sa += stepa;
        sb += stepb;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // the tile is complete
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = tr_read_pair(smem, oa[i] + kk * 32 * 2 * TBM, oa[i] + kk * 32 * 2 * TBM + 4 * 2 * TBM);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = tr_read_pair(smem + TA, ob[j] + kk * 32 * 2 * TBN, ob[j] + kk * 32 * 2 * TBN + 4 * 2 * TBN);
            // (swapped operands: a lane then holds 4 consecutive columns of row l15 -- row-contiguous stores)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma16(fb[j], fa[i], acc[i][j]);
        }
    }
    __builtin_amdgcn_s_barrier();                  // the stage becomes the waves' private epilogue scratch
    char* const scratch = smem + wave * 4096;
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * 64;
    const int S = a.splits;
    if (S == 1) {
#pragma unroll
        for (int h = 0; h < TM; ++h) wg_store_band(P.C, P.ldc, M, N, mw0, nw0, h, acc[h], scratch, lane, a.accumulate != 0);
        return;
    }
    // my partial quadrant -> slot (t, split) of the workspace, accumulator order (1 KB per wave-instruction)
    float* const slot0 = a.ws + ((long)t * S) * (TBM * TBN) + wave * 4096 + lane * 4;
    {
        float* const mine = slot0 + (long)split * (TBM * TBN);
#pragma unroll
        for (int h = 0; h < TM; ++h)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (FENCE) *reinterpret_cast<f32x4*>(mine + (h * 4 + j) * 256) = acc[h][j];
                else st4_agent(mine + (h * 4 + j) * 256, acc[h][j]);
            }
    }
    if constexpr (FENCE) __threadfence();          // release: my stores are visible at device scope before the count moves
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... every write-through store has been acknowledged
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(a.cnt + (long)t * 4 + wave, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(S - 1)) return;          // not the last arriver of this quadrant: done
    if constexpr (FENCE) __threadfence();          // acquire: the other splits' stores
    if (lane == 0) a.cnt[(long)t * 4 + wave] = 0u; // leave the counters zeroed for the next launch
#pragma unroll
    for (int h = 0; h < TM; ++h) {
        f32x4 sum[4];
#pragma unroll
        for (int j = 0; j < TN; ++j) sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {              // split order, my own share from registers
            if (s == split) {
#pragma unroll
                for (int j = 0; j < TN; ++j) sum[j] += acc[h][j];
            } else {
                const float* src = slot0 + (long)s * (TBM * TBN) + h * 1024;
                f32x4 x[4];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if constexpr (FENCE) x[j] = *reinterpret_cast<const f32x4*>(src + j * 256);
                if constexpr (!FENCE) ld4x4_agent(src, x);
#pragma unroll
                for (int j = 0; j < TN; ++j) sum[j] += x[j];
            }
        }
        wg_store_band(P.C, P.ldc, M, N, mw0, nw0, h, sum, scratch, lane, a.accumulate != 0);
    }
}

}  // namespace

extern "C" int mmtg_wgrad_group(int n, const mmtg_wgrad_problem* probs, int K, int splits, float* ws, long ws_floats,
                                unsigned* counters, long n_counters, int accumulate, void* stream) {
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
        const long bytesA = ((long)(K - 1) * q.lda + q.M) * 2, bytesB = ((long)(K - 1) * q.ldb + q.N) * 2;
        MMTG_REQUIRE(bytesA < 0x7FFFFF00L && bytesB < 0x7FFFFF00L, "wgrad_group: operands must stay below 2 GiB");
        WgProb& p = a.pr[i];
        p.A = q.A; p.B = q.B; p.C = q.C; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.M = q.M; p.N = q.N;
        p.tiles_m = cdiv(q.M, 128); p.tiles_n = cdiv(q.N, 128);
        p.tile0 = tiles;
        // the squarer block of the tile grid per XCD run (see launch_dma_cfg in gemm.hip)
        p.m_fast = p.tiles_n > p.tiles_m;
        p.bytesA = (int)bytesA; p.bytesB = (int)bytesB;
        tiles += p.tiles_m * p.tiles_n;
        flops += 2.0 * q.M * q.N * (double)K;
        bytes += 2.0 * ((double)q.M * K + (double)q.N * K) + 4.0 * (double)q.M * q.N;
    }
    const int kper = cdiv(cdiv(K, splits), 64) * 64;
    splits = cdiv(K, kper);                        // no empty K slice: every slot a last arriver reads was written
    MMTG_REQUIRE(splits == 1 || (ws && counters && MMTG_ALIGNED16(ws)), "wgrad_group: split products need the workspace and the counters");
    MMTG_REQUIRE(splits == 1 || (ws_floats >= (long)tiles * splits * 16384 && n_counters >= (long)tiles * 4),
                 "wgrad_group: workspace needs %ld floats and %ld counters", (long)tiles * splits * 16384, (long)tiles * 4);
    a.n = n; a.K = K; a.kper = kper; a.splits = splits; a.ntiles = tiles; a.accumulate = accumulate;
    a.ws = ws; a.cnt = counters;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_GEMM_BF16, s, flops, bytes);
    static bool attr_done = false;
    static const bool fence = getenv("MMTG_WGRAD_FENCE") != nullptr;
    const size_t shm = (128 + 128) * 128;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)wgrad_group_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_group_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "wgrad_group: cannot raise dynamic LDS to %zu bytes", shm);
        attr_done = true;
    }
    if (fence) hipLaunchKernelGGL(wgrad_group_kernel<true>, dim3(tiles * splits), dim3(256), shm, s, a);
    else hipLaunchKernelGGL(wgrad_group_kernel<false>, dim3(tiles * splits), dim3(256), shm, s, a);
    MMTG_LAUNCH_CHECK("wgrad_group");
    return MMTG_OK;
}
