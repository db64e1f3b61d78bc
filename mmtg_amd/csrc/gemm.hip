// LDS-tiled MFMA GEMM for gfx950 with fused epilogues.
//
//   C[m,n] = epi( sum_k opA(m,k) * opB(k,n) )
//
// One 256-thread workgroup (4 waves) owns a 128x128 output tile; each wave a
// 64x64 quadrant as 4x4 MFMA 16x16 tiles (bf16: v_mfma_f32_16x16x32_bf16,
// f32: v_mfma_f32_16x16x4_f32 -- exact fp32, used by the parity-gate mode).
// K advances 128 BYTES per tile (64 bf16 / 32 f32), double-buffered in LDS,
// next tile's global loads issued before the MFMA phase and written to LDS
// after it (one barrier per K tile).
//
// Operand layouts ("KC" = K-contiguous rows, "KS" = K-strided):
//   A KC: A[m*lda + k]      A KS: A[k*lda + m]
//   B KC: B[n*ldb + k]      B KS: B[k*ldb + n]
// so forward / dgrad of nn.Linear ([out,in]) and Conv1D ([in,out]) weights and
// the weight-gradient product X^T dY all run without materialised transposes.
// KS bf16 fragments are gathered by ds_read_b64_tr_b16 from an XOR-swizzled
// [k][col] LDS image; KC fragments by ds_read_b128 from a (row&7)-swizzled
// [row][k] image (both conflict-free).
#include "mma.h"

namespace {

enum { BM = 128, BN = 128, TILE_BYTES = 16384, NTHR = 256 };

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
    int kper;              // K elements per split (multiple of BK)
    float alpha;
    uint32_t drop_thresh; uint32_t drop_seed; float drop_inv_keep;
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

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(NTHR, 2) void gemm_kernel(GemmArgs p) {
    typedef typename Vec16<T>::type V;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    // buffer b: A tile at smem + 2*b*TILE_BYTES, B tile right after it

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: consecutive tiles (sharing the A row panel) land on one XCD's L2
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int m0 = (swz / p.tiles_n) * BM, n0 = (swz % p.tiles_n) * BN;

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

    // swapped operand roles (D[n][m]) give each lane 4 consecutive n -> vector epilogue;
    // the atomic (weight-gradient) epilogue keeps D[m][n] so a wave adds 16 contiguous floats per row.
    constexpr bool std_orient = AKS && BKS;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = kbeg + (kt + 1) * GT<T>::BK;
            stage_load<T, AKS>(A, p.lda, m0, p.M, k0, kend, tid, ra);
            stage_load<T, BKS>(B, p.ldb, n0, p.N, k0, kend, tid, rb);
        }
        const char* tA = smem + cur * 2 * TILE_BYTES;
        const char* tB = tA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            V fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!AKS) fa[i] = ld_frag_kc<T>(tA, wm * 64 + i * 16 + l15, kk, g);
                else fa[i] = ld_frag_ks(tA, wm * 64 + i * 16, kk, lane, p.use_tr, T());
                if (!BKS) fb[i] = ld_frag_kc<T>(tB, wn * 64 + i * 16 + l15, kk, g);
                else fb[i] = ld_frag_ks(tB, wn * 64 + i * 16, kk, lane, p.use_tr, T());
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
        if (kt + 1 < nk) {
            stage_store<T, AKS>(smem + (cur ^ 1) * 2 * TILE_BYTES, tid, ra);
            stage_store<T, BKS>(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, tid, rb);
        }
        __syncthreads();
    }

    // ------------------------------------------------------------ epilogue
    if constexpr (std_orient) {
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + 4 * g + r;
                    if (m < p.M && n < p.N) atomicAdd(C + (long)m * p.ldc + n, acc[i][j][r] * p.alpha);
                }
            }
    } else {
    const T* aux = reinterpret_cast<const T*>(p.aux);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * g;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
                v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
            }
            float a4[4];
            switch (p.epi) {
                case MMTG_EPI_GELU:
                    store4<T>(reinterpret_cast<T*>(p.aux2) + (long)m * p.ldc + n, v);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_new_f(v[r]);
                    break;
                case MMTG_EPI_TANH:
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
                    break;
                case MMTG_EPI_RESID:
                    if (p.drop_thresh) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] *= dropout_scale(p.drop_seed, (uint32_t)((long)m * p.N + n + r), p.drop_thresh, p.drop_inv_keep);
                    }
                    load4<T>(aux + (long)m * p.ldaux + n, a4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += a4[r];
                    break;
                case MMTG_EPI_DGELU:
                    load4<T>(aux + (long)m * p.ldaux + n, a4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= gelu_new_grad_f(a4[r]);
                    break;
                case MMTG_EPI_DTANH:
                    load4<T>(aux + (long)m * p.ldaux + n, a4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (1.0f - a4[r] * a4[r]);
                    break;
                default: break;
            }
            if (p.out_f32) store4<float>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v);
            else store4<T>(reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n, v);
        }
    }
    }
}

template <typename T>
int launch_gemm(const GemmArgs& a, int transA, int transB, int splits, hipStream_t stream) {
    const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
    dim3 grid(tiles_m * tiles_n, 1, splits), block(NTHR);
    if (!transA && transB) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, stream, a);
    else if (!transA && !transB) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, 0, stream, a);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, block, 0, stream, a);
    else MMTG_FAIL(MMTG_ERR_BAD_ARG, "gemm: layout transA=1,transB=1 is not built");
    MMTG_LAUNCH_CHECK("gemm");
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
    if (dtype == MMTG_F32) return launch_gemm<float>(a, transA, transB, splits, s);
    return launch_gemm<bf16>(a, transA, transB, splits, s);
}
