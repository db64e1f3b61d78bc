// Round 6: the MLP of a GPT-2 block in the decode token step -- c_fc (LayerNorm applied algebraically, + GELU) AND mlp.c_proj
// (+ bias + residual + the next LayerNorm's statistics) -- as ONE launch whose GEMM -> GEMM hand-off never leaves an XCD.
// (the GPT-2 block behind model.py:320-326, called once per token by generate.py:124; replaces two mmtg_decode_gemm launches.)
//
// Why this decomposition.  The batch rows of a decode step never interact, but an 8-way ROW split (a 32-row block per XCD) makes
// every XCD fetch every weight: 8x the fabric bytes of a step that already reads its weights 3.8x (profiles/r05_v4_decode_pmc_*).
// The split here is over the HIDDEN dimension instead:
//   * XCD x owns hidden columns [x HID/8, (x+1) HID/8).  Phase 1 computes G[:, slice x] = gelu(LN(X) W1[slice x]^T + b1) for ALL rows:
//     its 32 workgroups are 4 row blocks (64 rows) x 8 column tiles (48 columns), so the XCD's L2 fetches its 590 KB of W1 ONCE.
//   * mlp.c_proj contracts over the hidden dimension, so slice x IS a K slice of it: phase 2 computes the partial product
//     G[:, slice x] W2[:, slice x]^T on the same XCD.  The workgroup (row block rb, tile ct) needs G[rb, slice x] -- written by the 8
//     workgroups (rb, *) of ITS OWN XCD: the hand-off is an 8-workgroup barrier on one L2 (one 64-bit arrival word), not a device
//     barrier and not a kernel boundary.  Its W2 panel (96 x 384, 72 KB) is requested DURING phase 1 and is resident in LDS when
//     the hand-off completes: phase 2 starts with every weight byte already on the CU.
//   * The 8 partial products of an output tile (one per XCD) are reduced by the LAST ARRIVER, without waiting: every wave publishes its
//     16 x 96 fp32 partial with write-through stores and bumps the arrival counters of its two 48-column strips; the wave whose add
//     comes last sums the strip over the 8 slices IN SLICE ORDER (bit-reproducible whoever is last), adds bias + residual, rounds,
//     stores bf16 rows and the (sum, sum of squares) of what it stored: the LayerNorm statistics of the new residual stream as
//     48-column partials.  (The first version met on a per-tile counter and reduce-scattered 3 MFMA tiles per slice: the wait cost
//     3.4-3.8 us per launch, profiles/r06_v1_decode_mlp_one_launch_timeline_reduce_scatter.txt.)
// Weights cross the fabric once (9.4 MB per block instead of ~38), no split-K slab of the hidden activations exists, and the
// launch count of a block drops from 5 to 4.
//
// Placement.  Workgroup b of the 256 is (virtual XCD v = b % 8, slot b / 8): the hardware deals workgroups round-robin over the
// XCDs, so equal v means one physical XCD.  That is an observation, not a contract, so it is CHECKED on every launch: the arrival
// word of a hand-off group carries a 4-bit census per physical XCC_ID; with hand-off mode "plain" (stores that stay in the L2) a
// group found on two XCDs raises the error word.  Mode "sc1" (write-through stores + agent-scope loads) is correct under any
// placement and only slower under a wrong one.  Every wait is bounded by a wall-clock limit (4 ms) that raises the error word
// instead of hanging.  The last workgroup to finish re-arms every counter.
#include "gemm_common.h"

namespace {

constexpr int DM_D = 768, DM_HID = 4 * DM_D, DM_NX = 8;            // GPT-2 base; 8 XCDs
constexpr int DM_RB = 64;                                          // rows per row block (4 waves x 16)
constexpr int DM_SLICE = DM_HID / DM_NX;                           // 384 hidden columns per XCD = the K slice of phase 2
constexpr int DM_NCT = 8;                                          // column tiles per row block and XCD
constexpr int DM_TN1 = DM_SLICE / DM_NCT, DM_NT1 = DM_TN1 / 16;    // 48 columns = 3 MFMA tiles (phase 1)
constexpr int DM_TN2 = DM_D / DM_NCT, DM_NT2 = DM_TN2 / 16;        // 96 columns = 6 MFMA tiles (phase 2)
constexpr int DM_NK1 = DM_D / 64, DM_NK2 = DM_SLICE / 64;          // 12 / 6 K tiles of 64
constexpr int DM_NB1 = 6;                                          // phase-1 ring depth
constexpr int DM_ST1 = DM_RB * 128 + DM_TN1 * 128;                 // 8 KB of X rows + 6 KB of W1 rows per stage
constexpr int DM_W2ST = DM_TN2 * 128;                              // 12 KB of W2 rows per K tile
constexpr int DM_W2OFF = DM_NB1 * DM_ST1;                          // 86016
constexpr int DM_PAD = DM_W2OFF + DM_NK2 * DM_W2ST;                // 159744: 1 KB sink of the padding requests / the statistics scratch
constexpr int DM_STAT = DM_PAD + 1024;                             // mu | rstd of the 64 rows
constexpr int DM_LDS = DM_STAT + 2 * DM_RB * 4;                    // 161280 bytes
constexpr int DM_GST = DM_RB * 128;                                // phase 2: 8 KB of G rows per K tile (overlays the ring)
constexpr int DM_PART = DM_RB * DM_TN2 * 4;                        // 24 KB: one workgroup's fp32 partial
static_assert(DM_LDS <= 160 * 1024 && DM_NK2 * DM_GST <= DM_W2OFF, "LDS plan");
// sync words (unsigned long long): [0, 32) hand-off groups (v * 4 + rb), [64] finished, [65] error report, then 256 32-bit arrival
// counters of the reduction ((tile * 4 + wave row) * 2 + 48-column strip), each re-armed by its last arriver
constexpr int DM_SY_GROUP = 0, DM_SY_DONE = 64, DM_SY_ERR = 65, DM_SY_WORDS = 66, DM_SY_TOTAL = DM_SY_WORDS + 128;
constexpr unsigned long long DM_TIMEOUT_TICKS = 400000ull;        // 4 ms of the 100 MHz real-time counter

struct DmArgs {
    const bf16* X; long ldx;
    const float* stats_in; int np_in; float eps;
    const bf16* W1; long ldw1; const float* c1; const float* b1;
    const bf16* W2; long ldw2; const float* b2;
    bf16* G; long ldg;
    bf16* C; long ldc; float* stats_out;
    float* ws; unsigned long long* sync;
    int M, nwg;
    int bytesX, bytesW1, bytesW2, bytesG, bytesWs;
    int plain;                       // hand-off stores stay in the L2 (1) / are written through (0)
    unsigned long long* trace;       // diagnostic: 12 real-time stamps per workgroup, or null
};

__device__ __forceinline__ unsigned long long dm_ld64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane waits for `word` to count `want` arrivals (low 32 bits); bounded.  Returns the word (0 on a timeout / raised error).
__device__ __forceinline__ unsigned long long dm_wait(const unsigned long long* word, unsigned want, unsigned long long* err) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0;; ++it) {
        const unsigned long long v = dm_ld64(word);
        if ((unsigned)v >= want) return v;
        __builtin_amdgcn_s_sleep(4);
        if ((it & 15) == 15 && __builtin_amdgcn_s_memrealtime() - t0 > DM_TIMEOUT_TICKS) {
            __hip_atomic_fetch_or(err, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0ull;
        }
    }
}

#define DM_STAMP(i)                                                                          \
    do {                                                                                     \
        if (p.trace && tid == 0) p.trace[(long)blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// vmcnt a wave may leave outstanding when it needs ring tile KT: the ring tiles behind it plus the W2 stages requested since
constexpr int dm_vm(int kt) {
    const int ring = (DM_NK1 - 1 - kt < DM_NB1 - 2 ? DM_NK1 - 1 - kt : DM_NB1 - 2);
    int w2 = 0;
    if (kt <= DM_NB1 - 2) w2 = kt < DM_NK2 ? kt : DM_NK2;                       // tile kt was requested in the prologue
    else {
        const int lo = kt - (DM_NB1 - 1), hi = kt - 1 < DM_NK2 - 1 ? kt - 1 : DM_NK2 - 1;
        w2 = hi >= lo ? hi - lo + 1 : 0;
    }
    return 4 * ring + 3 * w2;
}

template <bool PLAIN>
__global__ __launch_bounds__(256) void decode_mlp_kernel(DmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int v = blockIdx.x & 7, slot = blockIdx.x >> 3;              // virtual XCD, slot on it
    const int rb = slot >> 3, ct = slot & 7;
    const int m0 = rb * DM_RB;
    unsigned long long* const err = p.sync + DM_SY_ERR;
    DM_STAMP(0);
    // (the error word is NOT read here: a dependent load in front of the first tile request would cost every launch ~1 us; it is a
    //  report for the host -- GreedyDecoder clears it per generation and raises if it finds it set)
    bool live = m0 < p.M;
    float* const smu = reinterpret_cast<float*>(smem + DM_STAT);
    float* const srs = smu + DM_RB;
    if (live) {
        const int n1 = v * DM_SLICE + ct * DM_TN1;                     // first hidden column of this workgroup (phase 1)
        const int n2 = ct * DM_TN2;                                    // first output column (phase 2)
        const int k2 = v * DM_SLICE;                                   // K slice of phase 2
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.X), 0, p.bytesX, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W1), 0, p.bytesW1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W2), 0, p.bytesW2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(p.G, 0, p.bytesG, 0x00020000);
        // per-lane source offsets of this wave's 1-KB blocks (the XOR swizzle lives in the source offset: an LDS-DMA lands lane-linear)
        int vx[2], vw1[2], vw2[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) vx[i] = dma_voff<false, DM_RB>(p.ldx, m0, p.M, 64, wave + 4 * i, lane);
        vw1[0] = dma_voff<false, DM_TN1>(p.ldw1, 0, DM_TN1, 64, wave, lane);
        vw1[1] = wave < 2 ? dma_voff<false, DM_TN1>(p.ldw1, 0, DM_TN1, 64, 4 + wave, lane) : OOB;      // 6 blocks: waves 2, 3 pad
#pragma unroll
        for (int i = 0; i < 3; ++i) vw2[i] = dma_voff<false, DM_TN2>(p.ldw2, 0, DM_TN2, 64, wave + 4 * i, lane);
        const int sx0 = (int)((long)m0 * p.ldx * 2), sw10 = (int)((long)n1 * p.ldw1 * 2), sw20 = (int)(((long)n2 * p.ldw2 + k2) * 2);
        char* const w1pad = wave < 2 ? nullptr : smem + DM_PAD;
#define DM_ISSUE1(tt)                                                                                                      \
    do {                                                                                                                   \
        char* st_ = smem + ((tt) % DM_NB1) * DM_ST1;                                                                       \
        const int so_ = (tt) * 128;                                                                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(void, st_ + wave * 1024), 16, vx[0], sx0 + so_, 0, 0);       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(void, st_ + (wave + 4) * 1024), 16, vx[1], sx0 + so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, LDS_PTR(void, st_ + DM_RB * 128 + wave * 1024), 16, vw1[0], sw10 + so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, LDS_PTR(void, w1pad ? w1pad : st_ + DM_RB * 128 + (4 + wave) * 1024), 16, vw1[1], sw10 + so_, 0, 0); \
    } while (0)
#define DM_ISSUE_W2(tt)                                                                                                    \
    do {                                                                                                                   \
        char* st_ = smem + DM_W2OFF + (tt) * DM_W2ST;                                                                      \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, LDS_PTR(void, st_ + (wave + 4 * i) * 1024), 16, vw2[i], sw20 + (tt) * 128, 0, 0); \
    } while (0)
        // requested BEFORE the first tiles (older in the wave's vmcnt order: their use below waits for them alone, not for the
        // ring): the LayerNorm statistics partials of the 64 rows, this lane's column sums / folded biases
        f32x4 sp[DG_NP / 2];
        if (tid < DM_RB) {
            const int m = min(m0 + tid, p.M - 1);
            const f32x4* src = reinterpret_cast<const f32x4*>(p.stats_in + (long)m * DG_NP * 2);
#pragma unroll
            for (int i = 0; i < DG_NP / 2; ++i) sp[i] = src[i];          // (all 32 slots exist; the unused ones are masked below)
        }
        f32x4 c4[DM_NT1], b4[DM_NT1];
#pragma unroll
        for (int j = 0; j < DM_NT1; ++j) {
            c4[j] = *reinterpret_cast<const f32x4*>(p.c1 + n1 + j * 16 + 4 * g);
            b4[j] = *reinterpret_cast<const f32x4*>(p.b1 + n1 + j * 16 + 4 * g);
        }
#pragma unroll
        for (int t0 = 0; t0 < DM_NB1 - 1; ++t0) DM_ISSUE1(t0);
        if (tid < DM_RB) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < DG_NP / 2; ++i) {
                const bool a = 2 * i < p.np_in, b = 2 * i + 1 < p.np_in;
                s1 += (a ? sp[i][0] : 0.f) + (b ? sp[i][2] : 0.f);
                s2 += (a ? sp[i][1] : 0.f) + (b ? sp[i][3] : 0.f);
            }
            const float mu = s1 * (1.0f / DM_D);
            const float var = fmaxf(s2 * (1.0f / DM_D) - mu * mu, 0.f);
            smu[tid] = mu;
            srs[tid] = rsqrtf(var + p.eps);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (smu / srs stored before this wave passes the first K-loop barrier)
        f32x4 acc1[DM_NT1];
#pragma unroll
        for (int j = 0; j < DM_NT1; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---------------------------------------------------------------- phase 1: 64 x 48 of gelu(LN(X) W1^T + b1), K = 768
#pragma unroll
        for (int kt = 0; kt < DM_NK1; ++kt) {
            switch (kt) {          // (compile-time after unrolling: the counted wait of this K tile)
#define DM_CASE(k_) case k_: wait_vmcnt<dm_vm(k_)>(); break;
                DM_CASE(0) DM_CASE(1) DM_CASE(2) DM_CASE(3) DM_CASE(4) DM_CASE(5) DM_CASE(6) DM_CASE(7) DM_CASE(8) DM_CASE(9) DM_CASE(10) DM_CASE(11)
#undef DM_CASE
            }
            __builtin_amdgcn_s_barrier();
            if (kt == 0) DM_STAMP(1);
            if (kt + DM_NB1 - 1 < DM_NK1) DM_ISSUE1(kt + DM_NB1 - 1);
            if (kt < DM_NK2) DM_ISSUE_W2(kt);                  // the whole W2 panel of phase 2, one K tile per step
            const char* tA = smem + (kt % DM_NB1) * DM_ST1;
            const char* tB = tA + DM_RB * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 fa = ld_frag_kc<bf16>(tA, wave * 16 + l15, kk, g);
                bf16x8 fb[DM_NT1];
#pragma unroll
                for (int j = 0; j < DM_NT1; ++j) fb[j] = ld_frag_kc<bf16>(tB, j * 16 + l15, kk, g);
#pragma unroll
                for (int j = 0; j < DM_NT1; ++j) mma16(fb[j], fa, acc1[j]);      // (swapped: a lane holds 4 consecutive columns of row l15)
            }
        }
        DM_STAMP(2);
        // ---- LN-fold + GELU epilogue -> G (bf16), 8 bytes per lane and tile
        {
            const int rl = wave * 16 + l15, m = m0 + rl;
            const float mu = smu[rl], rs = srs[rl];
#pragma unroll
            for (int j = 0; j < DM_NT1; ++j) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)gelu_new_t<bf16>(rs * (acc1[j][r] - mu * c4[j][r]) + b4[j][r]);
                const int off = (int)(((long)m * p.ldg + n1 + j * 16 + 4 * g) * 2);
                const int offc = m < p.M ? off : OOB;
                if constexpr (PLAIN) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rg, offc, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rg, offc, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every store of this wave acknowledged (and the W2 panel has landed)
        __syncthreads();                                       // ... of every wave; and every wave is done reading the ring
        DM_STAMP(3);
        // ---- the hand-off: the 8 workgroups (v, rb, *) meet on one arrival word; the 4-bit census of physical XCDs rides along
        __shared__ int s_ok;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
            unsigned long long* word = p.sync + DM_SY_GROUP + v * 4 + rb;
            __hip_atomic_fetch_add(word, 1ull | (1ull << (32 + 4 * xcc)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long w = dm_wait(word, DM_NCT, err);
            int ok = w != 0ull;
            if (ok && PLAIN) {                                 // plain hand-off: the group must sit on ONE physical XCD
                const unsigned census = (unsigned)(w >> 32);
                if (((census >> (4 * xcc)) & 15u) != (unsigned)DM_NCT) {
                    __hip_atomic_fetch_or(err, 4ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                }
            }
            s_ok = ok;
        }
        __syncthreads();
        live = s_ok != 0;
        DM_STAMP(4);
        if (live) {
            // ---- G[rb, slice v] -> LDS (agent-scope loads: never this CU's L1), 6 K tiles of 64 rows x 128 B over the ring
            int vg[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) vg[i] = dma_voff<false, DM_RB>(p.ldg, m0, p.M, 64, wave + 4 * i, lane);
            const int sg0 = (int)(((long)m0 * p.ldg + k2) * 2);
#pragma unroll
            for (int t = 0; t < DM_NK2; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, LDS_PTR(void, smem + t * DM_GST + (wave + 4 * i) * 1024), 16, vg[i], sg0 + t * 128, 0, 16);
            f32x4 acc2[DM_NT2];
#pragma unroll
            for (int j = 0; j < DM_NT2; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            DM_STAMP(5);
            // ---------------------------------------------------------------- phase 2: 64 x 96 partial of G W2^T over the K slice
#pragma unroll
            for (int kt = 0; kt < DM_NK2; ++kt) {
                const char* tA = smem + kt * DM_GST;
                const char* tB = smem + DM_W2OFF + kt * DM_W2ST;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 fa = ld_frag_kc<bf16>(tA, wave * 16 + l15, kk, g);
                    bf16x8 fb[DM_NT2];
#pragma unroll
                    for (int j = 0; j < DM_NT2; ++j) fb[j] = ld_frag_kc<bf16>(tB, j * 16 + l15, kk, g);
#pragma unroll
                    for (int j = 0; j < DM_NT2; ++j) mma16(fb[j], fa, acc2[j]);
                }
            }
            DM_STAMP(6);
            // ---- publish the partial: slot (tile, slice v), [wave][MFMA tile][lane] f32x4, write-through
            const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, p.bytesWs, 0x00020000);
            const int t2 = rb * DM_NCT + ct;
            const int tbase = t2 * (DM_NX * DM_PART);
#pragma unroll
            for (int j = 0; j < DM_NT2; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc2[j]), rws, tbase + v * DM_PART + (wave * DM_NT2 + j) * 1024 + lane * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every write-through store of THIS wave acknowledged
            DM_STAMP(7);
            // ---- reduce by the last arriver, no waiting: an arrival counter per (tile, wave row, 48-column strip); the wave of the 8
            // slices whose add comes last sums the strip over the slices IN SLICE ORDER (its own from registers: bit-reproducible
            // whoever arrives last), adds bias + residual, rounds, stores the bf16 rows and their (sum, sum of squares) -- and re-arms
            // the counter.  A wave signals only for its own stores (behind its own wait above).
            unsigned* const cnt = reinterpret_cast<unsigned*>(p.sync + DM_SY_WORDS) + (t2 * 4 + wave) * 2;
            unsigned old = 0;
            if (lane < 2) old = __hip_atomic_fetch_add(cnt + lane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned old0 = __builtin_amdgcn_readlane(old, 0), old1 = __builtin_amdgcn_readlane(old, 1);
            DM_STAMP(8);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((h ? old1 : old0) != (unsigned)(DM_NX - 1)) continue;
                if (lane == 0) __hip_atomic_store(cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f32x4 prt[DM_NX][3];
#pragma unroll
                for (int s = 0; s < DM_NX; ++s)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        prt[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, tbase + s * DM_PART + (wave * DM_NT2 + 3 * h + q) * 1024 + lane * 16, 0, 16));
                const int m = m0 + wave * 16 + l15;
                const bool ok = m < p.M;
                float r1 = 0.f, r2 = 0.f;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int j = 3 * h + q, n = n2 + j * 16 + 4 * g;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + n);
                    bf16x4 x4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                    if (ok) x4 = *reinterpret_cast<const bf16x4*>(p.X + (long)m * p.ldx + n);
                    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < DM_NX; ++s) sum += s == v ? acc2[j] : prt[s][q];        // slice order; mine from registers
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (bf16)(sum[r] + bb[r] + (float)x4[r]);
                        const float f = (float)o[r];
                        r1 += f;
                        r2 += f * f;
                    }
                    if (ok) *reinterpret_cast<bf16x4*>(p.C + (long)m * p.ldc + n) = o;
                }
                r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
                r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
                if (g == 0 && ok) {
                    float* dst = p.stats_out + ((long)m * DG_NP + ct * 2 + h) * 2;
                    dst[0] = r1;
                    dst[1] = r2;
                }
            }
        }
#undef DM_ISSUE1
#undef DM_ISSUE_W2
    }
    DM_STAMP(9);
    // ---- the last workgroup to finish re-arms the counters (every other workgroup has passed its last wait)
    if (tid == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(p.sync + DM_SY_DONE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned long long)(p.nwg - 1)) {
            for (int i = 0; i < 32; ++i) __hip_atomic_store(p.sync + DM_SY_GROUP + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.sync + DM_SY_DONE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// census of the placement assumption: out[xcc] += 1 for workgroups with (blockIdx % 8 == 0) ... out[8 * v + xcc]
__global__ __launch_bounds__(256) void decode_mlp_census_kernel(unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 0;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
        atomicAdd(out + 8 * (blockIdx.x & 7) + xcc, 1u);
        // (hold the CU for a moment so that the 256 workgroups have to be co-resident, as in the real launch)
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 2000ull) __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace

extern "C" long mmtg_decode_mlp_ws_floats(int M) {
    return (long)cdiv(M, DM_RB) * DM_NCT * DM_NX * (DM_PART / 4);
}
extern "C" int mmtg_decode_mlp_sync_words(void) { return DM_SY_TOTAL; }

extern "C" int mmtg_decode_mlp_census(unsigned* out64, void* stream) {
    MMTG_REQUIRE(out64, "decode_mlp_census: null pointer");
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)decode_mlp_census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "decode_mlp_census: cannot raise dynamic LDS");
        attr_done = true;
    }
    hipLaunchKernelGGL(decode_mlp_census_kernel, dim3(256), dim3(256), DM_LDS, s, out64);
    MMTG_LAUNCH_CHECK("decode_mlp_census");
    return MMTG_OK;
}

extern "C" int mmtg_decode_mlp(int M, int D, const void* X, long ldx, const float* stats_in, int np_in, float eps,
                               const void* W1f, long ldw1, const float* colsum1, const float* bias1f,
                               const void* W2t, long ldw2, const float* bias2, void* G, long ldg, void* C, long ldc,
                               float* stats_out, float* ws, long ws_floats, unsigned long long* sync, int plain_handoff,
                               unsigned long long* trace, void* stream) {
    MMTG_REQUIRE(D == DM_D, "decode_mlp: built for n_embd = %d (got %d)", DM_D, D);
    MMTG_REQUIRE(M > 0 && M <= 4 * DM_RB, "decode_mlp: 1..%d rows (got %d)", 4 * DM_RB, M);
    MMTG_REQUIRE(X && stats_in && W1f && colsum1 && bias1f && W2t && bias2 && G && C && stats_out && ws && sync, "decode_mlp: null pointer");
    MMTG_REQUIRE(np_in >= 1 && np_in <= DG_NP, "decode_mlp: 1..%d statistics partials", DG_NP);
    MMTG_REQUIRE(ldx % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && ldg % 8 == 0 && ldc % 4 == 0 && ldx >= D && ldw1 >= D && ldw2 >= DM_HID && ldg >= DM_HID && ldc >= D,
                 "decode_mlp: leading dimensions (multiples of 8 elements, at least the row length)");
    MMTG_REQUIRE(MMTG_ALIGNED16(X) && MMTG_ALIGNED16(W1f) && MMTG_ALIGNED16(W2t) && MMTG_ALIGNED16(G) && MMTG_ALIGNED16(C) && MMTG_ALIGNED16(ws) &&
                 MMTG_ALIGNED16(stats_in) && MMTG_ALIGNED16(colsum1) && MMTG_ALIGNED16(bias1f) && MMTG_ALIGNED16(bias2) && (((uintptr_t)sync) & 7) == 0,
                 "decode_mlp: 16-byte alignment");
    MMTG_REQUIRE(ws_floats >= mmtg_decode_mlp_ws_floats(M), "decode_mlp: %ld workspace floats needed", mmtg_decode_mlp_ws_floats(M));
    DmArgs a;
    memset(&a, 0, sizeof(a));
    a.X = (const bf16*)X; a.ldx = ldx; a.stats_in = stats_in; a.np_in = np_in; a.eps = eps;
    a.W1 = (const bf16*)W1f; a.ldw1 = ldw1; a.c1 = colsum1; a.b1 = bias1f;
    a.W2 = (const bf16*)W2t; a.ldw2 = ldw2; a.b2 = bias2;
    a.G = (bf16*)G; a.ldg = ldg; a.C = (bf16*)C; a.ldc = ldc; a.stats_out = stats_out;
    a.ws = ws; a.sync = sync; a.M = M; a.nwg = 256; a.plain = plain_handoff; a.trace = trace;
    const long bx = ((long)(M - 1) * ldx + D) * 2, bw1 = ((long)(DM_HID - 1) * ldw1 + D) * 2, bw2 = ((long)(D - 1) * ldw2 + DM_HID) * 2,
               bg = ((long)(M - 1) * ldg + DM_HID) * 2, bws = mmtg_decode_mlp_ws_floats(M) * 4;
    MMTG_REQUIRE(bx < 0x7FFFFF00L && bw1 < 0x7FFFFF00L && bw2 < 0x7FFFFF00L && bg < 0x7FFFFF00L && bws < 0x7FFFFF00L, "decode_mlp: operands must stay below 2 GiB");
    a.bytesX = (int)bx; a.bytesW1 = (int)bw1; a.bytesW2 = (int)bw2; a.bytesG = (int)bg; a.bytesWs = (int)bws;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(MMTG_PROF_GEMM_BF16, s, 4.0 * M * (double)D * DM_HID, 2.0 * (2.0 * D * DM_HID + 2.0 * M * D));
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)decode_mlp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_mlp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS) != hipSuccess)
            MMTG_FAIL(MMTG_ERR_HIP, "decode_mlp: cannot raise dynamic LDS");
        attr_done = true;
    }
    if (plain_handoff) hipLaunchKernelGGL(decode_mlp_kernel<true>, dim3(256), dim3(256), DM_LDS, s, a);
    else hipLaunchKernelGGL(decode_mlp_kernel<false>, dim3(256), dim3(256), DM_LDS, s, a);
    MMTG_LAUNCH_CHECK("decode_mlp");
    return MMTG_OK;
}
