// Shared pieces of the GEMM kernels: argument block, LDS tile layouts, fragment
// loaders and the fused epilogue.
//
// LDS images (16 KB per operand tile, K advances 128 BYTES per tile):
//   KC (K-contiguous operand): [128 rows][128 B], 16-byte chunk c of row r stored at c ^ (r & 7)
//       -> ds_read_b128 fragment reads are bank-conflict free.
//   KS (K-strided operand):    [k rows][128 cols], chunk c of row k stored at c ^ swz(k),
//       swz(k) = ((k&3)<<1) | (((k>>3)&1)<<3) -> ds_read_b64_tr_b16 reads are conflict free.
#pragma once
#include "mma.h"
#include <type_traits>

namespace {

enum { BM = 128, BN = 128, TILE_BYTES = 16384, NTHR = 256 };
constexpr int DG_NP = 32;     // decode step: LayerNorm statistics partials per row the buffers are sized for (decode.hip, decode_mlp.hip)

struct GemmArgs {
    const void* A; const void* B; void* C;
    const float* bias;     // [N] fp32 or null
    const void* aux;       // residual / saved pre-activation (type T), ld = ldaux
    void* aux2;            // second output (pre-activation for GELU), ld = ldc
    int M, N, K;
    long lda, ldb, ldc, ldaux;
    int epi;               // MMTG_EPI_*
    int out_f32;           // C is float regardless of T
    int use_tr;            // bf16 KS fragments via ds_read_b64_tr_b16 (1) or scalar gathers (0)
    int tiles_n;
    int tiles_m_fast;      // item order inside a K split: tile_m fastest (1) or tile_n fastest (0)
    int cbw;               // > 0: column-blocked order -- blocks of cbw tile columns, tile rows inside a block
    int ntiles;            // output tiles per K split (grid = ntiles * splits work items)
    int kper;              // K elements per split (multiple of BK)
    long split_stride;     // MMTG_EPI_SPLIT: bytes between the per-split output slabs (0 otherwise)
    int nitems;            // ntiles * splits work items (a persistent launch has fewer workgroups)
    unsigned long long* trace; int trace_n;   // diagnostic timeline (mmtg_gemm_trace) or null
    int dbg_flags;         // 1: keep the tile_n-fastest item order (MMTG_GEMM_ROW_ORDER, A/B only)
    float alpha;
    uint32_t drop_thresh; uint32_t drop_seed; float drop_inv_keep;
    int bytesA, bytesB;    // operand extents for the LDS-DMA buffer descriptors
    int gelu_grad;         // MMTG_GEMM_GELU_GRAD: GELU stores gelu'(pre) in aux2; DGELU multiplies by aux as stored
    const int* gather;     // mmtg_gemm_gather: table row of output row m (mode 0: A rows) / of reduction index k (mode 1: B rows)
    const int* aux_rows;   // row of `aux` that output row m reads (null: row m)
    // split-precision ("bf16x3") products: every fp32 operand travels as a (hi | lo) pair of bf16 planes, x ~ hi + lo with
    // hi = bf16(x), lo = bf16(x - hi); the K loop runs three passes hi*hi + lo*hi + hi*lo (fp32 accumulate) over the same tiles
    int x3;
    int planeA, planeB;    // bytes from an operand's hi plane to its lo plane (x3 kernels only)
    void* planes;          // optional second output of the x3 epilogues: the result as a (hi | lo) plane pair, ld = ldp
    long ldp, plane_out;   // plane_out: ELEMENTS from the hi plane to the lo plane of `planes`
    int aux2_bf16;         // MMTG_GEMM_AUX2_BF16: the x3 GELU epilogue stores its second output (pre-activation) as bf16 rows
};

template <typename T> struct GT {
    static constexpr int EPC = 16 / sizeof(T);    // elements per 16-byte chunk
    static constexpr int BK = 128 / sizeof(T);    // K elements per LDS tile
    static constexpr int KSTEP = 64 / sizeof(T);  // K elements per fragment (64 B of K)
    static constexpr int RB = 128 * sizeof(T);    // row bytes of a K-strided tile (128 columns)
    static constexpr int CPR = RB / 16;           // 16-byte chunks per K-strided row
};

__device__ __forceinline__ int ks_swz(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }

// ---- fragment reads --------------------------------------------------------
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type ld_frag_kc(const char* tile, int row, int kk, int g) {
    int off = row * 128 + ((((kk << 2) + g) ^ (row & 7)) << 4);
    return *reinterpret_cast<const typename Vec16<T>::type*>(tile + off);
}

__device__ __forceinline__ bf16x8 ld_frag_ks(const char* tile, int col0, int kk, int lane, int use_tr, bf16) {
    const int g = lane >> 4;
    bf16x8 out;
    if (use_tr) {
        const int q = (lane & 15) >> 2, p = lane & 3;
        const int k = kk * 32 + 8 * g + q;
        const int chunk = (col0 >> 3) + (p >> 1);
        const int o1 = k * 256 + ((chunk ^ ks_swz(k)) << 4) + 8 * (p & 1);
        const int o2 = (k + 4) * 256 + ((chunk ^ ks_swz(k + 4)) << 4) + 8 * (p & 1);
        out = tr_read_pair(tile, o1, o2);
    } else {
        const int c = col0 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kk * 32 + 8 * g + j;
            const int o = k * 256 + (((c >> 3) ^ ks_swz(k)) << 4) + (c & 7) * 2;
            out[j] = *reinterpret_cast<const bf16*>(tile + o);
        }
    }
    return out;
}
__device__ __forceinline__ f32x4 ld_frag_ks(const char* tile, int col0, int kk, int lane, int, float) {
    const int g = lane >> 4, c = col0 + (lane & 15);
    f32x4 out;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = kk * 16 + 4 * g + s;
        const int o = k * 512 + (((c >> 2) ^ ks_swz(k)) << 4) + (c & 3) * 4;
        out[s] = *reinterpret_cast<const float*>(tile + o);
    }
    return out;
}

// ---- global -> register staging of one 16 KB operand tile -------------------
template <typename T, bool KS>
__device__ __forceinline__ void stage_load(const T* __restrict__ base, long ld, int row0, int nrows,
                                           int k0, int kend, int tid, typename Vec16<T>::type (&r)[4]) {
    typedef typename Vec16<T>::type V;
    constexpr int EPC = GT<T>::EPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        V v;
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) v[e] = (T)0.0f;
        if (!KS) {
            const int c = tid & 7, row = (tid >> 3) + 32 * i;
            const int gr = row0 + row, gk = k0 + c * EPC;
            if (gr < nrows && gk < kend) v = *reinterpret_cast<const V*>(base + (long)gr * ld + gk);
        } else {
            const int id = tid + NTHR * i;
            const int k = id / GT<T>::CPR, c = id % GT<T>::CPR;
            const int gk = k0 + k, gc = row0 + c * EPC;
            if (gk < kend && gc < nrows) v = *reinterpret_cast<const V*>(base + (long)gk * ld + gc);
        }
        r[i] = v;
    }
}
template <typename T, bool KS>
__device__ __forceinline__ void stage_store(char* tile, int tid, const typename Vec16<T>::type (&r)[4]) {
    typedef typename Vec16<T>::type V;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int off;
        if (!KS) {
            const int c = tid & 7, row = (tid >> 3) + 32 * i;
            off = row * 128 + ((c ^ (row & 7)) << 4);
        } else {
            const int id = tid + NTHR * i;
            const int k = id / GT<T>::CPR, c = id % GT<T>::CPR;
            off = k * GT<T>::RB + ((c ^ ks_swz(k)) << 4);
        }
        *reinterpret_cast<V*>(tile + off) = r[i];
    }
}

template <typename TO> __device__ __forceinline__ void store4(TO* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = o;
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float (&v)[4]) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <typename TI> __device__ __forceinline__ void load4(const TI* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    f32x4 o = *reinterpret_cast<const f32x4*>(p);
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; v[3] = o[3];
}
template <> __device__ __forceinline__ void load4<bf16>(const bf16* p, float (&v)[4]) {
    bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)o[0]; v[1] = (float)o[1]; v[2] = (float)o[2]; v[3] = (float)o[3];
}


// ---- LDS-DMA addressing (shared by gemm.hip and wgrad.hip) -------------------
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

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// XCD-aware work order over a 1-D grid of (split, tile) items.  Workgroups are dealt round-robin to
// the 8 XCDs, so block b and b+8 share an L2: the bijective remap below gives every XCD one contiguous
// run of the virtual order  v = split * ntiles + tile_m * tiles_n + tile_n.  Neighbours in v share
// the A row panel, and -- for split-K weight gradients -- a whole XCD works inside ONE K slice, so
// each slice of the activations is fetched by one or two XCDs instead of all eight (measured HBM
// traffic of the wgrad GEMMs 2.3-2.8x algorithmic before, see profiles/).
// (`bid` of `nwg` work items; a persistent launch passes item numbers bid = blockIdx.x + i * gridDim.x
//  with gridDim.x a multiple of 8, which keeps every item on the XCD the map assumes.)
// floor(a / b) for 0 <= a < 2^20, 0 < b < 2^20 (work-item and tile counts): one v_rcp_f32 and a multiply instead of the ~40
// instructions of a 32-bit integer division -- a workgroup's prologue ran four of them in front of its first LDS-DMA request.
// (a + 0.5) / b is at least 0.5 / b away from an integer, i.e. a relative 0.5 / (a + 0.5) > 4.7e-7 x 2^20 / a away; rcp + mul err
// by < 3 ulp = 3.6e-7 relative, so the truncation lands on the exact quotient for every a < 2^20.)
__device__ __forceinline__ int fdiv_small(int a, int b) {
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
}
__device__ __forceinline__ void tile_origin(const GemmArgs& p, int bid, int nwg, int& m0, int& n0, int& split, int bm = BM, int bn = BN) {
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int v = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    split = fdiv_small(v, p.ntiles);
    const int t = v - split * p.ntiles;
    if (p.tiles_m_fast) {       // tile_m fastest: an XCD's run is (all tile rows) x (a few tile columns)
        const int tm = fdiv_small(p.ntiles, p.tiles_n);
        const int tn = fdiv_small(t, tm);
        m0 = (t - tn * tm) * bm;
        n0 = tn * bn;
    } else if (p.cbw > 0) {     // column blocks whose B panels stay in the XCD's L2 while the A rows stream by
        const int per = fdiv_small(p.ntiles, p.tiles_n) * p.cbw;
        const int cb = fdiv_small(t, per), u = t - cb * per;
        const int w = min(p.cbw, p.tiles_n - cb * p.cbw);
        const int um = fdiv_small(u, w);
        m0 = um * bm;
        n0 = (cb * p.cbw + u - um * w) * bn;
    } else {
        const int tmi = fdiv_small(t, p.tiles_n);
        m0 = tmi * bm;
        n0 = (t - tmi * p.tiles_n) * bn;
    }
}

// Per-lane byte offsets of the transposed-read fragments of a K-strided bf16 operand, hoisted out of
// the K loop: ks_swz(k) only depends on k&3 and (k>>3)&1, which are lane constants (k = 8g+q [+4] [+32kk]),
// so every fragment address is o[i] + kk*8192 (+1024 for the second 4-row block): one VGPR per 16-column
// block, everything else an instruction immediate.
__device__ __forceinline__ void ks_lane_offsets(int wq, int lane, int (&o)[4]) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int k = 8 * g + q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chunk = wq * 8 + 2 * i + (pp >> 1);
        o[i] = k * 256 + ((chunk ^ ks_swz(k)) << 4) + 8 * (pp & 1);
    }
}

// One K tile (two 64-byte k-blocks) of a wave's 64x64 quadrant: 16 fragment reads, 32 mma16.
template <typename T, bool AKS, bool BKS, bool std_orient, bool use_tr>
__device__ __forceinline__ void compute_tile(const char* tA, const char* tB, f32x4 (&acc)[4][4],
                                             int wm, int wn, int lane,
                                             const int (&oa)[4], const int (&ob)[4]) {
    typedef typename Vec16<T>::type V;
    const int g = lane >> 4, l15 = lane & 15;
    constexpr bool hoist = sizeof(T) == 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        V fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (!AKS) fa[i] = ld_frag_kc<T>(tA, wm * 64 + i * 16 + l15, kk, g);
            else if constexpr (hoist && use_tr) fa[i] = tr_read_pair(tA, oa[i] + kk * 8192, oa[i] + kk * 8192 + 1024);
            else fa[i] = ld_frag_ks(tA, wm * 64 + i * 16, kk, lane, 0, T());
            if constexpr (!BKS) fb[i] = ld_frag_kc<T>(tB, wn * 64 + i * 16 + l15, kk, g);
            else if constexpr (hoist && use_tr) fb[i] = tr_read_pair(tB, ob[i] + kk * 8192, ob[i] + kk * 8192 + 1024);
            else fb[i] = ld_frag_ks(tB, wn * 64 + i * 16, kk, lane, 0, T());
        }
        if (std_orient) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(fa[i], fb[j], acc[i][j]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(fb[j], fa[i], acc[i][j]);
        }
    }
}

// Fused epilogue.  STD orientation (weight gradients): lane holds D[m = 4g+r][n = l15] -> fp32 atomics,
// 16 contiguous floats per row per wave-instruction.  Swapped orientation: lane holds 4 consecutive n
// for m = l15 -> 8/16-byte vector stores.  The epilogue kind is dispatched ONCE (outside the unrolled
// tile loops) so each instantiation stays small.
// 8 consecutive elements of T <-> floats (16 B of bf16, 32 B of f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
template <typename TO> __device__ __forceinline__ void store8(TO* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
    *reinterpret_cast<bf16x8*>(p) = o;
}
// Epilogue through LDS.  The accumulator layout gives a lane 4 consecutive columns of ONE row per
// 16x16 tile, i.e. 32-byte global segments; staged through a wave-private LDS image the same data is
// re-read with 8 consecutive columns per lane and WTN/8 lanes per row, so every global access
// (output, pre-activation, residual / aux) is a 16-byte vector and a wave touches whole rows of the
// tile.  One pass per 16-row band of the wave tile: the image is [16 rows][WTN floats] with the
// 16-byte chunk c of row r stored at c ^ (r & (chunks-1)) -- conflict free for the accumulator
// writes (16 rows, same c) and for the row-wise read-back -- i.e. 4 KB per wave at WTN = 64.
// `lds` = this wave's scratch (epi_scratch_bytes); LDS operations of one wave complete in order, so
// the passes need no barrier between them.
template <typename T, int EPI, int TM, int TN, int PFD, bool X3 = false>
__device__ __forceinline__ void epi_tiles(const GemmArgs& p, f32x4 (&acc)[TM][TN], int mw0, int nw0, int g, int l15,
                                          char* lds, int lane) {
    constexpr int WTN = TN * 16;               // wave-tile columns
    constexpr int NCH = WTN / 4;               // 16-byte chunks per image row
    constexpr int LPR = WTN / 8;               // lanes per row on read-back
    constexpr int RPI = 64 / LPR;              // rows per read instruction
    constexpr int NQ = 16 / RPI;               // read instructions per 16-row band
    constexpr bool HAS_AUX = EPI == MMTG_EPI_RESID || EPI == MMTG_EPI_DGELU || EPI == MMTG_EPI_DTANH || EPI == MMTG_EPI_ROWDOT ||
                             EPI == MMTG_EPI_TANH_ADD;
    const T* aux = reinterpret_cast<const T*>(p.aux);
    // transcendental forms: the x3 mode stores fp32 but takes the v_exp_f32 / v_rcp_f32 forms (~1 ulp of fp32 each: two orders
    // below the split products' own error) -- libm tanhf in an exposed epilogue costs ~25 instructions per element
    typedef typename std::conditional<X3, bf16, T>::type TF;
    // Everything that comes from global memory is requested ahead of its use -- the lane's 8 bias
    // values once (its columns are the same in every band) and the aux vectors of its rows PFD bands
    // ahead (all of them up front where the register budget allows) -- so the epilogue waits for one
    // memory latency, under the LDS staging, instead of one per band (measured: the per-band loads
    // were ~6 us of an 18 us 128x128x768 item).
    const int col = (lane % LPR) * 8, n = nw0 + col, r0 = lane / LPR;
    const bool nok = n < p.N;
    float b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[e] = 0.f;
    if (p.bias && nok) load8<float>(p.bias + n, b8);
    typedef typename Vec16<T>::type V;
    constexpr int VPA = 8 / Vec16<T>::N;       // 16-byte vectors per 8 elements (bf16: 1, f32: 2)
    // |PFD| = bands of aux vectors requested ahead; PFD < 0 additionally drops the DGELU column sums
    // (the register-capped 6-wave configuration)
    constexpr int DEPTH = PFD < 0 ? -PFD : PFD;
    constexpr int RING = DEPTH + 1 < TM ? DEPTH + 1 : TM;     // bands of aux vectors held at once
    V ax[HAS_AUX ? RING * NQ * VPA : 1];
    // (clamped address: rows past M are loaded from row M-1 and never stored)
#define EPI_LOAD_BAND(hb)                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < NQ; ++q_) {                                                     \
        const int m_ = mw0 + (hb) * 16 + q_ * RPI + r0;                                                     \
        const int mc_ = m_ < p.M ? m_ : p.M - 1;                                                            \
        const V* src_ = reinterpret_cast<const V*>(aux + (long)(p.aux_rows ? p.aux_rows[mc_] : mc_) * p.ldaux + (nok ? n : 0)); \
        _Pragma("unroll") for (int u_ = 0; u_ < VPA; ++u_) ax[(((hb) % RING) * NQ + q_) * VPA + u_] = src_[u_]; \
    }
    if constexpr (HAS_AUX) {
#pragma unroll
        for (int hb = 0; hb < RING - 1; ++hb) { EPI_LOAD_BAND(hb) }
    }
    // DGELU: column sums of this lane's 8 columns (not in the register-capped 6-wave configuration,
    // PFD < TM, which the host never picks for a DGELU product with aux2)
    constexpr bool COLSUM = EPI == MMTG_EPI_DGELU && PFD > 0;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < TM; ++h) {
        if constexpr (HAS_AUX) {
            if (h + RING - 1 < TM) { EPI_LOAD_BAND(h + RING - 1) }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4*>(lds + l15 * (WTN * 4) + (((j * 4 + g) ^ (l15 & (NCH - 1))) << 4)) = acc[h][j];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = q * RPI + r0;
            const int m = mw0 + h * 16 + row;
            float v[8];
            {
                const int c0 = 2 * (lane % LPR), sw = row & (NCH - 1);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(lds + row * (WTN * 4) + ((c0 ^ sw) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(lds + row * (WTN * 4) + (((c0 + 1) ^ sw) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e] + b8[e]; v[4 + e] = hi[e] + b8[4 + e]; }
            }
            if (m >= p.M || !nok) continue;
            float a8[8];
            if constexpr (HAS_AUX) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = (float)ax[((h % RING) * NQ + q) * VPA + e / Vec16<T>::N][e % Vec16<T>::N];
            }
            if constexpr (EPI == MMTG_EPI_GELU) {
                if (p.gelu_grad) {          // the backward's only use of the pre-activation is gelu': store THAT (round 3)
                    float gd[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) gd[e] = gelu_new_grad_t<TF>(v[e]);
                    if (X3 && p.aux2_bf16) store8<bf16>(reinterpret_cast<bf16*>(p.aux2) + (long)m * p.ldc + n, gd);
                    else store8<T>(reinterpret_cast<T*>(p.aux2) + (long)m * p.ldc + n, gd);
                } else if (X3 && p.aux2_bf16) {      // (bf16x3f: the bf16 backward's dGELU epilogue reads bf16 rows)
                    store8<bf16>(reinterpret_cast<bf16*>(p.aux2) + (long)m * p.ldc + n, v);
                } else {
                    store8<T>(reinterpret_cast<T*>(p.aux2) + (long)m * p.ldc + n, v);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_new_t<TF>(v[e]);
            } else if constexpr (EPI == MMTG_EPI_TANH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tanh_t<TF>(v[e]);
            } else if constexpr (EPI == MMTG_EPI_TANH_ADD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tanh_t<TF>(v[e] + a8[e]);
            } else if constexpr (EPI == MMTG_EPI_RESID) {
                if (p.drop_thresh) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] *= dropout_scale(p.drop_seed, (uint32_t)((long)m * p.N + n + e), p.drop_thresh, p.drop_inv_keep);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += a8[e];
            } else if constexpr (EPI == MMTG_EPI_DGELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] *= p.gelu_grad ? a8[e] : gelu_new_grad_t<TF>(a8[e]);
                    if constexpr (COLSUM) cs[e] += (float)(T)v[e];      // column sums of the output as stored
                }
            } else if constexpr (EPI == MMTG_EPI_DTANH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (1.0f - a8[e] * a8[e]);
            } else if constexpr (EPI == MMTG_EPI_ROWDOT) {
                // attention backward's delta[m, head] = sum over the head's 64 columns of out * aux
                // (out = d ctx as stored, aux = ctx): a wave tile row IS one head (WTN == 64), held by 8 lanes
                static_assert(WTN == 64, "ROWDOT needs 64-column wave tiles (one attention head)");
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dot += (float)(T)v[e] * a8[e];
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
                dot += __shfl_xor(dot, 4, 64);
                if ((lane & 7) == 0) reinterpret_cast<float*>(p.aux2)[(long)m * (p.N >> 6) + (n >> 6)] = dot;
            }
            if constexpr (X3) {
                // fp32 result (optional) and / or its (hi | lo) bf16 planes: what the next split-precision product reads
                if (p.C) store8<float>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v);
                if (p.planes) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
                    bf16* dst = reinterpret_cast<bf16*>(p.planes) + (long)m * p.ldp + n;
                    *reinterpret_cast<bf16x8*>(dst) = hi;
                    *reinterpret_cast<bf16x8*>(dst + p.plane_out) = lo;
                }
            } else {
                if (p.out_f32) store8<float>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v);
                else store8<T>(reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n, v);
            }
        }
        if constexpr (COLSUM) {
            // aux2 (optional): f32 [ceil(M/64)][N] <- column sums of the output per 64-row band (four 16-row passes of this
            // wave tile), plain stores; summing the bands gives the bias gradient of the layer whose pre-activation
            // gradient this product computes (a [M/64, N] reduction instead of a pass over the [M,N] result).  Not atomics
            // onto [N]: 118 row tiles adding to the same addresses serialise in the L2 (+29 us on the 15104 x 3072 product,
            // measured in the training step).  Lanes with equal lane % LPR hold the same 8 columns for different rows:
            // fold them first.
            static_assert(!COLSUM || TM % 4 == 0, "column-sum bands are 64 rows: wave tiles of 64 or 128 rows");
            if (h % 4 == 3) {
                if (p.aux2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
#pragma unroll
                        for (int o = LPR; o < 64; o <<= 1) cs[e] += __shfl_xor(cs[e], o, 64);
                    }
                    if (lane < LPR && nok && mw0 + (h - 3) * 16 < p.M)
                        store8<float>(reinterpret_cast<float*>(p.aux2) + (long)((mw0 >> 6) + (h >> 2)) * p.N + n, cs);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] = 0.f;
            }
        }
    }
}

#undef EPI_LOAD_BAND

// mw0 / nw0: global row / column of the wave's sub-tile origin
// bytes of wave-private LDS scratch the staged epilogue needs
template <int TM, int TN> constexpr int epi_scratch_bytes() { return 16 * (TN * 16) * 4; }

// `lds`: this wave's scratch of epi_scratch_bytes<TM,TN>() bytes; the caller has made sure (barrier) that
// no wave still reads the main-loop tiles it overlays.
template <typename T, bool std_orient, int TM, int TN, int PFD = TM, bool X3 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TM][TN], int mw0, int nw0, int g, int l15,
                                              char* lds, int lane) {
    if constexpr (std_orient) {
        float* C = reinterpret_cast<float*>(p.C);
        // MMTG_EPI_SPLIT (the register-staged kernel, exact-fp32 weight gradients): the caller has moved C to this K split's slab --
        // the raw partial product with plain stores, mmtg_slab_sum adds the slabs in index order (no fp32 atomics: reproducible)
        const bool slab = p.epi == MMTG_EPI_SPLIT;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nw0 + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mw0 + i * 16 + 4 * g + r;
                    if (m < p.M && n < p.N) {
                        if (slab) C[(long)m * p.ldc + n] = acc[i][j][r];
                        else atomicAdd(C + (long)m * p.ldc + n, acc[i][j][r] * p.alpha);
                    }
                }
            }
    } else {
        switch (p.epi) {
            case MMTG_EPI_GELU: epi_tiles<T, MMTG_EPI_GELU, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
            case MMTG_EPI_TANH: epi_tiles<T, MMTG_EPI_TANH, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
            case MMTG_EPI_TANH_ADD: epi_tiles<T, MMTG_EPI_TANH_ADD, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
            case MMTG_EPI_RESID: epi_tiles<T, MMTG_EPI_RESID, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
            case MMTG_EPI_DGELU: epi_tiles<T, MMTG_EPI_DGELU, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
            case MMTG_EPI_DTANH: epi_tiles<T, MMTG_EPI_DTANH, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
            case MMTG_EPI_ROWDOT:
                if constexpr (TN == 4) epi_tiles<T, MMTG_EPI_ROWDOT, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane);
                break;
            default: epi_tiles<T, MMTG_EPI_NONE, TM, TN, PFD, X3>(p, acc, mw0, nw0, g, l15, lds, lane); break;
        }
    }
}

}  // namespace
