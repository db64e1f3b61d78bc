/* libmmtg_hip.so -- C ABI of the MI355X (gfx950) kernels behind MMTG's
 * training + generation hot path.
 *
 * The reference (Aman-4-Real/MMTG) has no native layer: every operator on the
 * path is a stock PyTorch call inside src/model.py / src/loss.py /
 * src/generate.py.  Each entry point below therefore cites the reference
 * *operator call site(s)* it replaces (file:line into /root/reference/src).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless
 *     stated; the caller owns every buffer (nothing is allocated, freed or
 *     retained across calls);
 *   - `dtype` selects the storage type of activations / weight copies:
 *     MMTG_F32 (exact fp32 MFMA, the parity-gate mode) or MMTG_BF16 (bf16
 *     storage, fp32 accumulate).  Statistics, losses, gradients of parameters
 *     and optimizer state are always fp32;
 *   - `stream` is a hipStream_t; every call is asynchronous and stream-ordered,
 *     performs no host synchronisation and is hipGraph-capturable;
 *   - return 0 on success, a negative MMTG_ERR_* otherwise; the message is
 *     available (thread-local) from mmtg_last_error().  Nothing throws/exits.
 */
#ifndef MMTG_HIP_H
#define MMTG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MMTG_ABI_VERSION 12

/* The library is built with -fvisibility=hidden: only the entry points below are exported. */
#define MMTG_API __attribute__((visibility("default")))

enum { MMTG_OK = 0, MMTG_ERR_BAD_ARG = -1, MMTG_ERR_HIP = -2, MMTG_ERR_UNSUPPORTED = -3 };
enum { MMTG_F32 = 0, MMTG_BF16 = 1 };

/* GEMM epilogues */
enum {
    MMTG_EPI_NONE = 0,   /* C = acc + bias                                    */
    MMTG_EPI_GELU = 1,   /* aux2 = acc + bias ; C = gelu_new(aux2)            */
    MMTG_EPI_TANH = 2,   /* C = tanh(acc + bias)                              */
    MMTG_EPI_RESID = 3,  /* C = dropout(acc + bias) + aux                     */
    MMTG_EPI_DGELU = 4,  /* C = acc * gelu_new'(aux) ; optional aux2 (f32 [ceil(M/64)][N]) <- column sums of C
                            per 64-row band; mmtg_colsum over the bands = the c_fc bias gradient, without a
                            pass over the [M,N] pre-activation gradient */
    MMTG_EPI_DTANH = 5,  /* C = acc * (1 - aux^2)                             */
    MMTG_EPI_ATOMIC = 6, /* C(f32) += alpha * acc  (atomics; split-K allowed) */
    MMTG_EPI_ROWDOT = 7, /* C = acc ; aux2(f32)[m, n/64] = sum over each 64-column group of C * aux
                            (attention backward's delta = rowsum(dO * O) per head, fused into the
                            GEMM that produces dO; N % 64 == 0, no bias) */
    MMTG_EPI_SPLIT = 8,  /* deterministic split-K for small M (decode): C is fp32 [splits][M][ldc]; K split s
                            stores its raw partial product in slab s (plain stores; out_f32, no bias);
                            mmtg_splitk_finish sums the slabs in order and applies bias / activation /
                            residual (/ LayerNorm) */
    MMTG_EPI_TANH_ADD = 9 /* C = tanh(acc + bias + aux[aux_rows[m]]) -- mmtg_gemm_gather only */
};
#define MMTG_GEMM_NO_TR 1 /* flags: gather K-strided bf16 fragments without ds_read_b64_tr_b16 */
#define MMTG_GEMM_REGSTAGE 2 /* flags: register-staged v1 pipeline instead of the LDS-DMA one (bf16) */
#define MMTG_GEMM_SKINNY 4    /* flags: force the 256x32-tile small-M configuration (bf16, transA=0,transB=1) */
#define MMTG_GEMM_NO_SKINNY 8 /* flags: never pick it automatically (it is the default for M <= 256) */
#define MMTG_GEMM_WIDE 16     /* flags: force 192x128 tiles / 6 waves (bf16, transA=0); chosen automatically when it
                                 saves a partial round of workgroups (N = 768) or for N >= 4096 */
#define MMTG_GEMM_NO_WIDE 32  /* flags: never pick 192x128 automatically */
#define MMTG_GEMM_PERSIST 64  /* flags: use the persistent pipelined 128x128 kernel (one workgroup per CU slot walks the
                                 (tile, K split) items, next item prefetched, epilogue deferred into it); opt-in:
                                 measured on par with the plain launch */
#define MMTG_GEMM_NO_PERSIST 128 /* flags: reserved (the persistent kernel is never picked automatically) */
#define MMTG_GEMM_ROW_ORDER 256  /* flags: keep the plain tile_n-fastest item order: no tile_m-fastest weight gradients, no
                                    column blocks for forward / dgrad products (A/B measurements) */
#define MMTG_GEMM_OCC4 512       /* flags: force the single-stage 128x128 kernel, four workgroups per CU (bf16); automatic for
                                    weight gradients and for forward / dgrad products of more than 2 x CUs tiles */
#define MMTG_GEMM_NO_OCC4 1024   /* flags: never pick it automatically (A/B measurements) */
#define MMTG_GEMM_NO_P8 8192     /* flags: never pick the eight-phase 256x256 / 192x256 kernel (bf16, transA = 0, transB = 1, K % 128 == 0: the
                                    default for M >= 1024, N >= 256) -- A/B measurements and bit-equality tests against the older kernels */
#define MMTG_GEMM_GELU_GRAD 32768 /* flags (round 3): MMTG_EPI_GELU stores gelu_new'(pre-activation) in aux2 instead of the pre-activation, and
                                    MMTG_EPI_DGELU multiplies by aux as it is (C = acc * aux): the backward then needs no transcendental --
                                    the only consumer of the saved pre-activation was gelu' (both products of a pair must carry the flag) */
#define MMTG_GEMM_P8 16384       /* flags: eight-phase kernel also for the dGELU product (otherwise on the single-stage kernel by measurement) */
#define MMTG_GEMM_AUX2_BF16 131072 /* flags (round 6, mmtg_gemm_x3 only): MMTG_EPI_GELU stores aux2 (the pre-activation) as bf16 rows [M, ldc] instead of
                                     fp32 -- the hybrid mode (compute_dtype "bf16x3f") runs its backward on the bf16 kernels */
#define MMTG_GEMM_P8_288 65536   /* flags (round 4): force the eight-phase kernel's 288-row tiles where it applies (the tile rule picks them by cost; tests / A-B) */
#define MMTG_GEMM_P256 4096      /* flags: persistent pipelined kernel with 256x128 tiles, 8 waves, one workgroup per CU (bf16, transA = 0) */
#define MMTG_GEMM_COL_BLOCK 2048 /* flags: force the column-blocked item order with blocks of two tile columns (test hook;
                                    automatic when the weights do not stay in an XCD's L2 over several sweeps) */

/* profiling categories (mmtg_prof_*) */
enum {
    MMTG_PROF_GEMM_BF16 = 0, MMTG_PROF_GEMM_F32, MMTG_PROF_ATTN_FWD, MMTG_PROF_ATTN_BWD,
    MMTG_PROF_LAYERNORM, MMTG_PROF_EMBED, MMTG_PROF_LOSS, MMTG_PROF_OPTIM, MMTG_PROF_ENCODER,
    MMTG_PROF_DECODE, MMTG_PROF_MISC, MMTG_PROF_NCAT
};

MMTG_API int mmtg_abi_version(void);
/* extra -D flags the library was compiled with ("" for the product build; diagnostic builds name theirs) */
MMTG_API const char* mmtg_build_flags(void);
MMTG_API const char* mmtg_last_error(void);

/* Live per-kernel timing with HIP events recorded on the launch stream.
 * enable(1) starts bracketing every launch; read() synchronises the recorded
 * events and returns, per category, launches / total ms / algorithmic flops /
 * algorithmic bytes (arrays of MMTG_PROF_NCAT), then clears the log. */
MMTG_API int mmtg_prof_enable(int on);
MMTG_API int mmtg_prof_read(int* launches, double* ms, double* flops, double* bytes);

/* ---------------------------------------------------------------- GEMM
 * C[M,N] = epi(opA[M,K] * opB[K,N]).  transA=0: A[m*lda+k]; 1: A[k*lda+m].
 * transB=0: B[k*ldb+n]; 1: B[n*ldb+k] (an nn.Linear weight [out,in]).
 * Replaces every nn.Linear / Conv1D / lm_head product on the path:
 * model.py:77 (topic_fc), :78-79 (GRU projections), :134-136 (alpha QKV),
 * :199 (out_linear), :279-281 (projector), and the c_attn / c_proj / c_fc /
 * lm_head products inside GPT2LMHeadModel (call sites model.py:282-288,
 * 320-326), plus their autograd backward products (train.py:193).           */
MMTG_API int mmtg_gemm(int dtype, int transA, int transB, int M, int N, int K,
              const void* A, long lda, const void* B, long ldb, void* C, long ldc,
              const float* bias, int epi, const void* aux, long ldaux, void* aux2,
              int out_f32, float alpha, int splits, unsigned drop_thresh, unsigned drop_seed,
              int flags, void* stream);

/* Split-precision ("bf16x3") product, round 5 -- the mode in which north_star's numeric gates (logits within 1e-3, greedy ids
 * bit-exact) hold at bf16 matrix-core speed.  Replaces the SAME fp32 products as mmtg_gemm (the Conv1D / Linear layers of GPT-2
 * behind model.py:282-288,320-326 and the tied lm_head) when both operands are K-contiguous:
 *   C[M, N] (fp32) = epi( A . B^T + bias ),   A [M, K], B [N, K] fp32 tensors handed over as (hi | lo) PLANE PAIRS of bf16
 *   (hi = bf16(x), lo = bf16(x - hi); the lo plane lies planeA / planeB ELEMENTS behind the hi plane, same leading dimension),
 *   computed as A_hi B_hi^T + A_lo B_hi^T + A_hi B_lo^T with fp32 accumulation in ONE kernel (a 3K-deep loop of the eight-phase
 *   kernel).  aux / aux2 / C are fp32 with the meanings of mmtg_gemm's epilogues NONE, GELU, TANH, RESID (+ dropout), DGELU
 *   (+ column sums), DTANH, ROWDOT.  `planes` (nullable; ld = ldp, lo plane plane_out elements behind) receives the epilogue's
 *   result as a plane pair for the next split-precision product; C may be NULL when only the planes are consumed.
 *   K % 128 == 0; N, lda, ldb, ldc, ldp % 8 == 0; every plane pair below 2 GiB.  Plane pairs come from mmtg_split_planes,
 *   mmtg_layernorm_fwd_x3, this function's `planes` output, and mmtg_adamw (weights).                                      */
MMTG_API int mmtg_gemm_x3(int M, int N, int K, const void* A, long lda, long planeA, const void* B, long ldb, long planeB,
                 float* C, long ldc, void* planes, long ldp, long plane_out, const float* bias, int epi,
                 const float* aux, long ldaux, void* aux2, unsigned drop_thresh, unsigned drop_seed, int flags, void* stream);
/* fp32 [rows, cols] (ld = lds) -> its (hi | lo) bf16 plane pair (ld = ldp, lo plane `plane` elements behind the hi plane). */
MMTG_API int mmtg_split_planes(const float* src, long lds, int rows, int cols, void* planes, long ldp, long plane, void* stream);
/* LayerNorm forward (fp32 rows in, statistics out as mmtg_layernorm_fwd) whose output goes straight to a plane pair: the
 * GPT-2 LayerNorms feed products only (ln_1 -> c_attn, ln_2 -> c_fc, ln_f -> lm_head), so the fp32 rows are never stored.
 * x_bf16 (nullable, round 6): also receives bf16(x) [rows, cols] -- the input rows as the bf16 LayerNorm backward reads them
 * (compute_dtype "bf16x3f": split-precision forward, bf16 backward). */
MMTG_API int mmtg_layernorm_fwd_x3(const float* x, void* planes, long ldp, long plane, const float* gamma, const float* beta,
                          float* mean, float* rstd, int rows, int cols, float eps, void* x_bf16, void* stream);

/* Products whose operand rows are GATHERED from a table by index -- the multi-modal conditioning front end of
 * GPT2_Decoder.forward (model.py:254-281) without materialising X[m] = E[id_m] + c[b, seg_m] (62 MB at B = 64):
 *   mode 0 (forward):  C[M, N] = epi( E[rows[m], :] . B[N, K]^T + bias [+ aux[aux_rows[m], :]] ), epi in {NONE, TANH, TANH_ADD}.
 *                      By linearity (E[id] + c) W1^T = E[id] W1^T + (c W1^T)[b, seg]: the caller computes the small product
 *                      c W1^T once ([B*S + 1, N], last row zero) and passes it as aux with aux_rows[m] = b*S + seg or B*S.
 *                      The table row of output row m enters the LDS-DMA as the lane's source offset.
 *   mode 1 (weight gradient):  slabs C[s][M, N] (fp32, MMTG_EPI_SPLIT layout) = A[k, M]^T . E[rows[k], :] over K split s
 *                      (A = d(pre-activation) [K = tokens, M], K-strided; the table rows of a K tile are looked up one tile ahead);
 *                      mmtg_slab_sum finishes it; the c-part of the gradient is the small product segment_sum(dA)^T . c.
 * bf16 only; K % 64 == 0 (mode 0); rows / aux_rows: int32 device arrays; E: [*, ldb_or_lda] row-major, below 2 GiB.        */
MMTG_API int mmtg_gemm_gather(int mode, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                     const float* bias, int epi, const int* rows, int table_rows, const void* aux, long ldaux, const int* aux_rows,
                     int splits, void* stream);

/* Diagnostic timeline of the bf16 LDS-DMA GEMM kernels (tools/gemm_timeline.py): while `buf` is
 * non-null, wave 0 of workgroup w < max_wgs of every such launch writes 6 x u64 at buf + 48*w --
 * s_memrealtime (100 MHz) at kernel entry, after the first K tile has landed, at the end of the K
 * loop, at exit; the K tile count; the hardware id (XCC / SE / CU).  Null switches it off (default). */
MMTG_API int mmtg_gemm_trace(void* buf, int max_wgs);
/* CUs the tile-shape rule of the eight-phase kernel (one workgroup = one whole CU) may count on: 0 = all, > 0 = that many,
 * < 0 = all but that many.  A rule input only -- results never depend on it.  mmtg_amd.ddp reserves CUs for the RCCL kernels
 * that run beside the backward (replaces nothing in the reference: its nn.DataParallel, train.py:112-114, serialises). */
MMTG_API int mmtg_gemm_cu_budget(int cus);
/* Measurement hook (tools/ddp_contention.py; never on the product path): `workgroups` 256-thread workgroups that each hold
 * `lds_bytes` of LDS (163840 = a whole CU) and sleep for `usec` microseconds -- a stand-in for a collective's ring kernels holding
 * their CUs beside the backward, so that the CU reservation above can be priced on one GPU. */
MMTG_API int mmtg_debug_occupy(int workgroups, int lds_bytes, double usec, void* stream);

/* Second half of a MMTG_EPI_SPLIT product: out[m, :] = epi(sum_s part[s][m][:] + bias) in the storage type
 * (slabs summed in index order -> deterministic); epi in {NONE, GELU, TANH, RESID (+ aux)}.  With ln_out the
 * row just produced is also LayerNormed (gamma, beta, eps) into ln_out [M, N] -- in the KV-cached decode step
 * (generate.py:117-126 -> GPT2Block) this replaces the ln_1 / ln_2 / ln_f launches.  part: [splits][M][ldp].   */
MMTG_API int mmtg_splitk_finish(int dtype, const float* part, int splits, int M, int N, long ldp, const float* bias,
                       int epi, const void* aux, long ldaux, void* out, long ldo,
                       const float* ln_gamma, const float* ln_beta, void* ln_out, float eps, void* stream);

/* column sums: out[n] += sum_m X[m,n]  (bias gradients), X of `dtype`, out f32.  Round 4: summed in a FIXED order -- one workgroup
 * owns 64 columns and walks all the rows; inputs above 2048 rows go through row slices in `ws` first -- so the result is bit-identical
 * run to run (the round-1 kernel finished with fp32 atomics from ~1000 workgroups). */
MMTG_API long mmtg_colsum_ws(int M, int N);      /* workspace floats mmtg_colsum needs for M rows (0: none) */
MMTG_API int mmtg_colsum(int dtype, const void* X, long ldx, int M, int N, float* out, float* ws, long ws_floats, void* stream);
/* Round 6: many small ordered column sums in ONE launch -- out_i[c] += sum_{r < M_i} X_i[r * ldx_i + c], c < N_i, every item summed in
 * the order mmtg_colsum uses for the same rows (the same bits); fp32 rows, M_i <= 2048.  `items` is a HOST array (the items travel in
 * the kernel arguments, 64 per launch); the items of a call run concurrently, so no two may share output elements.  The backward of a GPT-2 block ends four such reductions (two LayerNorm second stages, the
 * dGELU bands, the attention kernels' bias rows: the bias / LayerNorm gradients autograd produces behind model.py:282-288) that nothing
 * reads before the optimizer: mmtg_layernorm_bwd_partial / MMTG_ATTN_DBIAS_ROWS leave them as partial rows, this call sums them.      */
typedef struct { const float* X; float* out; long ldx; int M; int N; } mmtg_colsum_item;
MMTG_API int mmtg_colsum_batch(const mmtg_colsum_item* items, int n, void* stream);

/* ---------------------------------------------------------------- LayerNorm
 * torch.nn.LayerNorm (model.py:380-382) and GPT-2's ln_1/ln_2/ln_f.          */
MMTG_API int mmtg_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                       float* mean, float* rstd, int rows, int cols, float eps, void* stream);
/* dx = LN'(dy) (+ dres if non-null); dgamma/dbeta accumulated (+=) in fp32.
 * Optional fused tail for the residual stream: dx_masked = dx * dropout_mask(drop_seed) (the
 * gradient entering the previous residual branch; null = not needed) and
 * dcolsum[c] += sum_rows dx_masked (that branch's bias gradient; without dx_masked: of dx).
 * ws: caller-owned scratch of >= mmtg_layernorm_bwd_ws(rows, cols) floats.     */
MMTG_API long mmtg_layernorm_bwd_ws(int rows, int cols);
MMTG_API int mmtg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                       const float* mean, const float* rstd, const void* dres, void* dx,
                       float* dgamma, float* dbeta, int rows, int cols,
                       void* dx_masked, unsigned drop_thresh, unsigned drop_seed, float* dcolsum,
                       float* ws, long ws_floats, void* stream);
/* The first stage of mmtg_layernorm_bwd alone (round 6): dx (+ dx_masked) as above, and the partial rows ws[k][q][cols], k < *partial_rows
 * (host, out), q = 0: d gamma, 1: d beta, 2: column sums of dx_masked / dx when want_colsum -- to be summed by mmtg_colsum_batch
 * (X = ws + q * cols, ldx = 3 * cols, M = *partial_rows, N = cols): the same bits as the one-call form.  ws must stay untouched until then. */
MMTG_API int mmtg_layernorm_bwd_partial(int dtype, const void* dy, const void* x, const float* gamma,
                       const float* mean, const float* rstd, const void* dres, void* dx, int rows, int cols,
                       void* dx_masked, unsigned drop_thresh, unsigned drop_seed, int want_colsum,
                       float* ws, long ws_floats, int* partial_rows, void* stream);
/* x3 mode: mmtg_layernorm_bwd on fp32 rows whose dropout-masked input gradient (dx itself without dropout) goes to a (hi | lo) bf16
 * plane pair [rows, cols] (lo plane `plane` elements behind) instead of an fp32 tensor: only split-precision products read it. */
MMTG_API int mmtg_layernorm_bwd_x3(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                          void* dx_planes, long plane, unsigned drop_thresh, unsigned drop_seed, float* dcolsum,
                          float* ws, long ws_floats, void* stream);
/* ... and its first stage alone (as mmtg_layernorm_bwd_partial): the partial rows stay in ws for mmtg_colsum_batch */
MMTG_API int mmtg_layernorm_bwd_x3_partial(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, int rows, int cols, void* dx_planes, long plane,
                          unsigned drop_thresh, unsigned drop_seed, int want_colsum,
                          float* ws, long ws_floats, int* partial_rows, void* stream);

/* ---------------------------------------------------------------- causal self-attention
 * GPT2Attention._attn: softmax(QK^T/sqrt(dh) + causal + key padding) V with
 * attention dropout, heads merged (call sites model.py:282-288, 320-326).
 * qkv: [B*T, 3*D] rows = tokens, columns q|k|v with head h at h*dh (dh = 64);
 * keep: [B,T] int32 key mask (1 = attend); out: [B*T, D]; lse: [B,nH,T] f32.  */
MMTG_API int mmtg_attn_fwd(int dtype, const void* qkv, const int* keep, void* out, float* lse,
                  int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream);
/* x3 mode (round 5): causal attention of the fp32-storage split-precision mode on the bf16 matrix cores -- the tiled algorithm of
 * mmtg_attn_fwd / mmtg_attn_bwd with every product (Q K^T, P V, dO V^T, dO^T P, Q^T dS, dS K) as three passes over (hi | lo) splits;
 * P and dS are split in registers; softmax / masks / dropout in fp32 as the fp32 kernels (same dropout counter stream).
 * qkv [B*T, 3D] and dout [B*T, D] arrive as (hi | lo) bf16 plane pairs (lo planes qplane / doplane elements behind the hi planes:
 * what mmtg_gemm_x3 writes for c_attn / the c_proj dgrad); out fp32.  Forward: context rows to `out` (fp32: the backward's delta
 * reads them) AND, when out_planes is given, to a plane pair [B*T, D] (attn.c_proj's operand).  Backward: d(qkv) is written ONLY as
 * a plane pair [B*T, 3D] (lo plane dplane elements behind): the c_attn dgrad / weight gradient read nothing else.  dq32: fp32
 * scratch of dq32_floats >= ceil(T/128) * B*T * D floats (one [B*T, D] buffer per block of 128 keys: plain stores, summed in block
 * order by the finish kernel -- no atomics, bit-reproducible); delta: [B*T, nH] scratch; dbias
 * (nullable): [3D] += column sums of d(qkv), through dbias_ws (>= (B * ceil(T/128) + ceil(B*T/16)) * 3D floats); delta_ready != 0:
 * delta[m, h] = sum_d dout * out was filled by the caller (mmtg_gemm_x3's MMTG_EPI_ROWDOT epilogue does it for free).
 * Round 6: dbias == NULL with dbias_ws non-null leaves the partial bias rows UNSUMMED at the head of dbias_ws -- [B * ceil(T/128)][3D]
 * (k and v parts; the q part zero), then [ceil(B*T/16)][D] (the q part) -- for the caller's mmtg_colsum_batch (as two items with
 * disjoint output columns: [D, 3D) from the first rows, [0, D) from the second).                                                    */
MMTG_API int mmtg_attn_fwd_x3(const void* qkv_planes, long qplane, const int* keep, float* out, void* out_planes, long plane, float* lse,
                     int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream);
MMTG_API int mmtg_attn_bwd_x3(const void* qkv_planes, long qplane, const int* keep, const float* out, const void* dout_planes, long doplane,
                     const float* lse, float* delta,
                     int delta_ready, float* dq32, long dq32_floats, void* dqkv_planes, long dplane, float* dbias, float* dbias_ws,
                     long dbias_ws_floats, int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, void* stream);
/* delta: [B*T, nH] f32, delta[m,h] = sum_d dout[m,h,d] * out[m,h,d]: computed by the call, or --
 * delta_ready != 0 -- already filled by the caller (the GEMM producing dout with
 * MMTG_EPI_ROWDOT does it for free); dq32: [B*T, D] f32 scratch (zeroed by the call);
 * dqkv: [B*T, 3*D] output; dbias (optional): f32 [3*D] += column sums of dqkv as stored (the
 * c_attn bias gradient, from the workgroups that produce each head's columns); dbias_ws (optional
 * scratch, f32 [B * ceil(T / key block) + 44][3*D], key block = 256 bf16 / 128 f32): partial rows that the
 * call sums into dbias in a fixed order (the 44 spare rows: workspace of the tiled kernels' column sum of dQ) --
 * without it the workgroups use atomics on dbias (slower: contended, and not reproducible bit for bit).       */
/* diagnostic: per-wave timeline of the whole-head forward kernel (bf16, T <= 256): buf = u64 [B*nH*8][8]
 * (s_memrealtime at entry / loads issued / first chunk landed / long tile done / stored / exit, XCC id, valid) or NULL */
MMTG_API int mmtg_attn_trace(void* buf);
/* flags (round 6): MMTG_ATTN_ELEM_MASK -- the bf16 whole-head kernels (T <= 512) regenerate the attention-dropout mask of the tiled /
 * split-precision kernels (keep (q, k) iff hash(seed, ((b nH + h) T + q) T + k) >= drop_thresh, scale 1 / (1 - p) exactly) instead of
 * their own 12-bit word masks: the bf16x3f mode's backward then differentiates the mask its split-precision forward applied.        */
#define MMTG_ATTN_ELEM_MASK 1
/* MMTG_ATTN_DBIAS_ROWS (round 6) -- where mmtg_attn_bwd_dbias_rows(dtype, B, T) > 0 (the bf16 whole-head kernels) the c_attn bias
 * gradient is left as that many partial rows [rows, 3D] at the head of dbias_ws for the caller's mmtg_colsum_batch (dbias is then only
 * the request for them); other shapes ignore the flag and sum inside the call.                                                        */
#define MMTG_ATTN_DBIAS_ROWS 2
MMTG_API int mmtg_attn_bwd_dbias_rows(int dtype, int B, int T);
MMTG_API int mmtg_attn_bwd(int dtype, const void* qkv, const int* keep, const void* out, const void* dout,
                  const float* lse, float* delta, int delta_ready, float* dq32, void* dqkv, float* dbias, float* dbias_ws,
                  int B, int T, int nH, int dh, unsigned drop_thresh, unsigned drop_seed, int flags, void* stream);

/* ---------------------------------------------------------------- conditioning front end
 * WenLan lookup + experience add (model.py:254-268):
 *   x[b,t] = E[topic_ids[b,t]]                           t <  P
 *   x[b,P+p] = E[targets[b,p]] + (p/two_sents < S ? c[b,p/two_sents] : 0)
 * table [V,E], c [B,S,E], x [B,P+L,E] of `dtype`; ids int64.                  */
MMTG_API int mmtg_embed_condition(int dtype, const void* table, const long long* topic_ids,
                         const long long* targets, const void* c, void* x,
                         int B, int P, int L, int S, int E, int two_sents, int V, void* stream);
/* out[b,k,:] = sum_{p in segment k} g[b,P+p,:]   (backward of the add), out of `dtype` */
MMTG_API int mmtg_segment_sum(int dtype, const void* g, void* out, int B, int P, int L, int S, int H,
                     int two_sents, void* stream);
/* GPT-2 input embedding: h[m,:] = dropout(g[m,:] + wpe[m % T,:] + wte[type_ids[m],:])
 * (GPT2Model.forward via model.py:282-288).  In place on g allowed.           */
MMTG_API int mmtg_embed_add(int dtype, const void* g, const void* wpe, const void* wte, const long long* type_ids,
                   void* h, int M, int T, int D, unsigned drop_thresh, unsigned drop_seed, void* stream);
/* backward: dwpe[t,:] += sum_b dh[b,t,:]; dwte[type,:] += sum dh; (dh masked in place if dropout).  ws (optional, f32,
 * mmtg_embed_add_bwd_ws(M, D, ntypes) floats): per-row-block bins of the token-type rows, summed in a fixed order (round 4:
 * bit-reproducible); without it the bins end in fp32 atomics on dwte. */
MMTG_API long mmtg_embed_add_bwd_ws(int M, int D, int ntypes);
MMTG_API int mmtg_embed_add_bwd(int dtype, void* dh, const long long* type_ids, float* dwpe, float* dwte,
                       int M, int T, int D, int ntypes, unsigned drop_thresh, unsigned drop_seed, float* ws, long ws_floats,
                       void* stream);
/* elementwise dropout mask application (backward of a fused-epilogue dropout) */
MMTG_API int mmtg_dropout_apply(int dtype, const void* x, void* y, long n, int N, unsigned drop_thresh,
                       unsigned drop_seed, void* stream);

/* ---------------------------------------------------------------- LM head loss (loss.py:45-74 + GPT-2's internal CE)
 * logits: f32 [M, ldl] (columns >= V ignored).  labels per row t are
 * cat(topic_ids, targets)[b, t+1] (label_zero != 0: all labels 0, the
 * inference branch's dummy labels, model.py:314).
 * Outputs (all f32): nll[M] (0 on each sample's last row), lse[M],
 * sample_ce[B], coef[B] = d loss / d CE_b / n_tok / batch_denominator,
 * scalars[0] = MyLoss (sum_b l_b / batch_den), scalars[1] = GPT-2 LM loss.    */
MMTG_API int mmtg_loss_fwd(int logits_dtype, const void* logits, long ldl, int V, const long long* topic_ids,
                  const long long* targets, const long long* ratings, int stage, int label_zero,
                  int B, int P, int L, float batch_den, float* nll, float* lse, float* sample_ce,
                  float* coef, float* scalars, void* stream);
/* dlogits[m,v] = rc[m] * (softmax(logits[m])_v - [v == label]),
 * rc[m] = gscale * coef[b] on MyLoss rows (P <= t <= T-2) + lm_coef on GPT-2 LM-loss rows
 * (t <= T-2; pass d lm_loss / B / (T-1), 0 when the LM loss is unused as in train.py:188);
 * 0 on each sample's last row and on pad columns; dlogits of `dtype`, ld = ldd.  `logits_dtype` is the
 * storage type of the logits (fp32, or bf16 with bf16 dlogits -- then dlogits may alias logits). */
MMTG_API int mmtg_loss_bwd(int dtype, int logits_dtype, const void* logits, long ldl, int V, const long long* topic_ids,
                  const long long* targets, const float* lse, const float* coef, float gscale, float lm_coef,
                  int B, int P, int L, void* dlogits, long ldd, int Vpad, void* stream);
/* x3 mode: the same gradient of fp32 logits written as a (hi | lo) bf16 plane pair (ld = ldd, lo plane `plane` elements behind
 * the hi plane) -- the operand of the LM head's split-precision dgrad / weight gradient; the fp32 rows are never stored. */
MMTG_API int mmtg_loss_bwd_x3(const float* logits, long ldl, int V, const long long* topic_ids, const long long* targets, const float* lse,
                     const float* coef, float gscale, float lm_coef, int B, int P, int L, void* planes, long ldd, long plane,
                     int Vpad, void* stream);

/* ---------------------------------------------------------------- encoder / fuser pieces
 * GRU cell (nn.GRU math, model.py:78-79): gi (row stride ld_gi), gh (row stride ld_gh >= 3H, or 0 = one
 * [3H] row for every b: at step 0, h_prev = 0, the recurrent product is just b_hh)
 * pre-activations (r|z|n), h_prev (row stride ld_hp; null = zeros) -> h (row
 * stride ld_h); saves r,z,n,ghn (each [B,H]) in `save` [4,B,H] f32.  The row
 * strides address one step of a batch-first [B,S,*] sequence in place.         */
MMTG_API int mmtg_gru_cell_fwd(int dtype, const void* gi, long ld_gi, const void* gh, long ld_gh, const void* h_prev, long ld_hp,
                      void* h, long ld_h, float* save, int B, int H, void* stream);
/* dh: total gradient wrt h_t (f32 [B,H]); outputs dgi (row stride ld_dgi), dgh [B,3H]
 * of `dtype`, dh_prev (f32 [B,H]) = dh * z (caller adds dgh * W_hh).           */
MMTG_API int mmtg_gru_cell_bwd(int dtype, const float* dh, const float* save, const void* h_prev, long ld_hp,
                      void* dgi, long ld_dgi, void* dgh, float* dh_prev, int B, int H, void* stream);
/* The same with the assembly of dh_t fused in: dh_t = rows[b, :] (gradient arriving through the LayerNorm of step t,
 * `dtype`, row stride ld_rows) + carry (f32 [B,H], nullable; may alias dh_prev) + the ordered sum of the `splits`
 * fp32 slabs part[splits][B][H] of the carry product d(gh_{t+1}) W_hh (MMTG_EPI_SPLIT; splits = 0: none).          */
MMTG_API int mmtg_gru_cell_bwd_fused(int dtype, const void* rows, long ld_rows, const float* carry, const float* part, int splits,
                            const float* save, const void* h_prev, long ld_hp, void* dgi, long ld_dgi, void* dgh,
                            float* dh_prev, int B, int H, void* stream);
/* The other recurrent cells MultiModalEncoder accepts (reference src/model.py:41-59: nn.RNN(nonlinearity="relu") / nn.LSTM).
 * Pre-activations a = gi + gh ([B, G*H] rows; G = 1 / 4, LSTM gate order i|f|g|o; ld_gh = 0: one bias row for every b).
 * LSTM: c_prev / c fp32 [B,H] (c_prev null = zeros), save [5,B,H] f32 = i, f, g, o, tanh(c);  ReLU cell: c_prev, c, save unused. */
#define MMTG_RNN_RELU 0
#define MMTG_RNN_LSTM 1
MMTG_API int mmtg_rnn_cell_fwd(int dtype, int kind, const void* gi, long ld_gi, const void* gh, long ld_gh, const float* c_prev,
                      void* h, long ld_h, float* c, float* save, int B, int H, void* stream);
/* One BPTT step: dh_t = rows[b, :] (`dtype`, row stride ld_rows) + part (f32 [B,H], nullable: d(a_{t+1}) W_hh) -> da (`dtype`,
 * [B, G*H] rows of stride ld_da; it is d(gi) and d(gh) at once).  LSTM: dc (f32 [B,H]) is read when dc_in != 0 (dc_{t+1} f_{t+1})
 * and rewritten with dc_t f_t; ReLU cell: h = the step's output rows (stride ld_h).                                              */
MMTG_API int mmtg_rnn_cell_bwd(int dtype, int kind, const void* rows, long ld_rows, const float* part, const float* save,
                      const float* c_prev, const void* h, long ld_h, float* dc, int dc_in, void* da, long ld_da,
                      int B, int H, void* stream);
/* alpha attention (model.py:138-161): qkv [B*S, 3H] -> ctx [B*S, H], probs [B,heads,S,S] f32,
 * kl += mean_i KLDiv_batchmean(log P[:,:,i,:], prior_i); prior [S,S] f32.       */
MMTG_API int mmtg_alpha_attn_fwd(int dtype, const void* qkv, const float* prior, void* ctx, float* probs,
                        float* kl, int B, int S, int H, int heads, void* stream);
MMTG_API int mmtg_alpha_attn_bwd(int dtype, const void* qkv, const float* prior, const float* probs,
                        const void* dctx, float dkl, void* dqkv, int B, int S, int H, int heads,
                        void* stream);
/* beta attention / multi-modal fuser (model.py:191-198): topic [B,H], img/txt [B*S,H] (row b*S+i)
 * att_w [S,H] f32, att_b [S] f32 -> o [B*S,H], a [B,S,3] f32.                   */
MMTG_API int mmtg_beta_fuse_fwd(int dtype, const void* topic, const void* img, const void* txt,
                       const float* att_w, const float* att_b, void* o, float* a,
                       int B, int S, int H, void* stream);
MMTG_API long mmtg_beta_fuse_bwd_ws(int B, int S, int H);
/* ws (optional, f32, mmtg_beta_fuse_bwd_ws floats): per-(b, step) contributions to d att_w / d att_b, summed over b in a fixed
 * order, d topic by its single owner (round 4: bit-reproducible); without it the round-1 fp32 atomics. */
MMTG_API int mmtg_beta_fuse_bwd(int dtype, const void* topic, const void* img, const void* txt,
                       const float* att_w, const float* a, const void* d_o,
                       float* dtopic, void* dimg, void* dtxt, float* datt_w, float* datt_b,
                       int B, int S, int H, float* ws, long ws_floats, void* stream);

/* Software prefetch (no counterpart in the reference): streams `bytes` of `src` through the cache hierarchy with
 * `workgroups` workgroups of 16-byte loads so that they sit in the 256 MB Infinity Cache when the next kernel of the
 * backward pass reads them (saved activations are cold by then).  Launched on a side stream, gated by events; `sink` is a
 * 4-byte device scratch that is never written in practice.                                                       */
MMTG_API int mmtg_prefetch(const void* src, long bytes, int workgroups, void* sink, void* stream);
/* zero n ranges of an fp32 buffer in one launch: desc (device, int64) holds (first element, count) pairs, both multiples of 4
 * (optimizer.zero_grad of train.py:192 for the gradients that are accumulated into; the rest is overwritten) */
MMTG_API int mmtg_zero_ranges(float* base, const long* desc, int n, void* stream);

/* ---------------------------------------------------------------- optimizer (train.py:194-197)
 * sumsq: *out = sum x^2 (global grad-norm), one partial per workgroup in `ws` (mmtg_sumsq_ws(n) floats) + an ordered final sum:
 * bit-reproducible (round 4; the round-1 kernel added ~2000 partials to *out with fp32 atomics).                          */
MMTG_API long mmtg_sumsq_ws(long n);
MMTG_API int mmtg_sumsq(const float* x, long n, float* out, float* ws, long ws_floats, void* stream);
/* clip (coef = min(1, max_norm / (sqrt(*normsq) + 1e-6))) + transformers.AdamW
 * (bias-corrected, eps outside the sqrt, decoupled wd) + optional bf16 copy.
 * count (optional, device scalar): g holds a SUM over rows and *count the global row count
 * (all-reduced on the device, never read by the host): the gradient is g * grad_scale / *count;
 * *count == 0 leaves every buffer untouched (the reference skips an empty batch, train.py:184-185). */
/* p_lo (nullable; with p_bf16): the x3 mode's lo plane, bf16(p - bf16(p)) -- p_bf16 / p_lo are then the weights' plane pair. */
MMTG_API int mmtg_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, void* p_lo, long n,
               float lr, float beta1, float beta2, float eps, float wd, int step,
               const float* normsq, float max_norm, float grad_scale, const float* count, void* stream);
MMTG_API int mmtg_cast_f32_to(int dtype, const float* src, void* dst, long n, void* stream);
/* dst[r, 0:cols] = cast(src[r, 0:cols]); dst[r, cols:ldd] = 0 */
MMTG_API int mmtg_cast_pad_rows(int dtype, const float* src, long lds_, void* dst, long ldd, int rows, int cols, void* stream);
MMTG_API int mmtg_cast_to_f32(int dtype, const void* src, float* dst, long n, void* stream);
MMTG_API int mmtg_axpy_f32(float* y, const float* x, float a, long n, void* stream);
/* Second half of a weight-gradient product computed as K-split slabs (mmtg_gemm with transA = 1 and
 * MMTG_EPI_SPLIT: slab s = fp32 [M, ldc] at C + s * M * ldc): dst[i] (+)= sum_s part[s * stride + i],
 * slabs added in order -- the deterministic replacement of the fp32-atomic epilogue for the autograd
 * weight gradients of nn.Linear / Conv1D (model.py:77-79,134-136,199,279-281 and the GPT-2 products).
 * Slab producers: the bf16 single-stage LDS-DMA kernel and (round 6) the exact-fp32 kernel.
 * n and stride in floats, multiples of 4; accumulate = 0 overwrites dst.                              */
MMTG_API int mmtg_slab_sum(const float* part, int splits, long stride, float* dst, int accumulate, long n, void* stream);
/* Grouped weight-gradient products with an in-kernel deterministic split-K reduction (csrc/wgrad.hip): ONE launch for
 * the n <= 8 products  C_p[M_p, N_p] (fp32, ld = ldc_p) (+)= A_p^T . B_p,  A_p [K, lda_p], B_p [K, ldb_p] bf16 row-major with
 * K = tokens -- the autograd weight gradients of the four Conv1D layers of a GPT-2 block (transformers 4.12.3 behind
 * model.py:282-288), which become available within one block of the backward and otherwise need 5-12 K splits EACH to fill
 * the chip.  Work items = (sum of 128x128 tiles) x splits.  K split s of a tile stores its raw partial tile in slot
 * (tile, s) of `ws` (tile-contiguous, >= tiles * splits * 16384 floats); every wave then bumps the arrival counter of
 * its 64x64 quadrant (`counters`: >= 4 * tiles unsigned, ZERO on entry, zero again on return) and the wave that arrives
 * last adds the partial quadrants in split order and writes (accumulate = 0) or adds to (accumulate = 1) C.  No workgroup
 * ever waits for another; results are bit-reproducible.  splits = 1 needs neither ws nor counters.
 * `probs` is a HOST array (read during the call).  M, N, lda, ldb multiples of 8; operands below 2 GiB.
 * config 0: 128x128 tiles, four 256-thread workgroups per CU (any shape; 16384 floats and 4 counters per tile and split);
 * config 1: 256x256 tiles on the eight-phase K-strided main loop, one 512-thread workgroup per CU (65536 floats and 8
 *           counters per tile and split; K slices are multiples of 128).
 * config 2 (round 5, the split-precision mode; 128x128 tiles): A_p and B_p are (hi | lo) bf16 plane pairs of fp32 tensors -- the
 *           lo plane planeA / planeB ELEMENTS behind the hi plane -- and every K slice is walked three times,
 *           A_hi^T B_hi + A_lo^T B_hi + A_hi^T B_lo (see mmtg_gemm_x3); planeA / planeB are ignored otherwise.
 * config 6 (= 2 | 4): the same products on combined stages -- a 64 KB stage holds the K tile of all four planes and every fragment
 *           pair feeds the three MFMA passes (two workgroups per CU; a third fewer bytes through the L2 -> LDS port).      */
typedef struct mmtg_wgrad_problem {
    const void* A; long lda;
    const void* B; long ldb;
    float* C; long ldc;
    int M, N;
    long planeA, planeB;
} mmtg_wgrad_problem;
MMTG_API int mmtg_wgrad_group(int config, int n, const mmtg_wgrad_problem* probs, int K, int splits, float* ws, long ws_floats,
                     unsigned* counters, long n_counters, int accumulate, void* stream);
/* Batched transpose (bf16 mode keeps K-contiguous [out,in] copies of GPT-2's Conv1D [in,out] weights
 * so that forward products run in the NT layout): matrix i = [rows, cols] row-major at src + desc[4i]
 * elements -> [cols, rows] at dst + desc[4i+3]; desc = n x {src_off, rows, cols, dst_off} (long, device);
 * every offset / extent a multiple of 16 bytes; max_rows / max_cols bound the grid.                    */
MMTG_API int mmtg_transpose_batch(int dtype, const void* src, void* dst, const long* desc, int n, int max_rows, int max_cols, void* stream);

/* ---------------------------------------------------------------- generation (generate.py:127-141)
 * per row: repetition penalty per occurrence (ids 0 and 102 skipped), /temperature,
 * ban {1,2,100,102}, sticky PAD, arg-max (lowest index on ties) -> next[B].     */
MMTG_API int mmtg_logits_process_argmax(const float* logits, long ldl, int V, const long long* generated,
                               long ldg, const int* gen_len, float temperature, float rep_penalty,
                               long long* next, int B, void* stream);
/* Stochastic counterpart (generate.py:64-94,127-141 with top_k / top_p as given): same processed logits, then the
 * reference's top-k filter (values below the k-th largest dropped, ties kept; 0 = off) and nucleus filter (sorted
 * descending, an id stays while the probability mass before it is <= top_p; 0 = off), softmax, and ONE draw per row
 * by inverse CDF over the kept ids in index order with the caller's uniforms[b] in [0,1) (torch.multinomial's
 * generator is not reproducible outside torch; the distribution is the reference's).  filtered (nullable, [B, ldl]):
 * the filtered processed logits (-inf where dropped), as top_k_top_p_filtering returns them.                        */
MMTG_API int mmtg_logits_process_sample(const float* logits, long ldl, int V, const long long* generated, long ldg,
                               const int* gen_len, float temperature, float rep_penalty, int top_k, float top_p,
                               const float* uniforms, long long* next, float* filtered, int B, void* stream);

/* ---------------------------------------------------------------- KV-cached decode step (generate.py:117-142)
 * Batched, lock-step: every row is at position *pos_ptr (a DEVICE int, so a captured hipGraph of
 * the step replays for every position).  seq[B, ldseq] = prompt ids (0..P-1) then lyric ids.
 * decode_embed: x[b] = E[seq[b,pos]] (+ c[b,(pos-P)/two_sents]) and the inference-branch type id /
 *   key mask of the token (model.py:296-312): type_out[b], keep[b,pos].
 * decode_embed_add: h = g + wpe[pos] + wte[type].
 * decode_attn: appends the token's K/V to the caches [B,nH,Tmax,64] and attends over keys 0..pos.
 * decode_select: next lyric token = forced [#EOS#]/[#START#] by the 22-slot cadence, sticky PAD, or
 *   arg-max of the processed logits (null logits: forced tokens only); writes seq[b, pos+1].
 * decode_advance: *pos_ptr += 1.                                                              */
MMTG_API int mmtg_decode_embed(int dtype, const void* table, const long long* seq, long ldseq, const void* c, void* x,
                      const int* pos_ptr, const long long* tpw_type, const long long* tpw_mask,
                      long long* type_out, int* keep, long ldkeep, int B, int P, int S, int E, int two_sents,
                      int V, int sent, int max_sent_num, void* stream);
/* stats (optional, f32 [B][32][2]): the LayerNorm statistics (sum, sum of squares) of every row of h as stored, in partial 0
 * (partials 1..31 zeroed) -- the input of the first LN-fold product of the fused decode step (mmtg_decode_gemm).          */
MMTG_API int mmtg_decode_embed_add(int dtype, const void* g, const void* wpe, const void* wte, const long long* type_ids,
                          const int* pos_ptr, void* h, int B, int D, float* stats, void* stream);
/* Round 3: the decode step's batch-sized products with the row-wise glue fused in (bf16; A [M, lda], W [N, ldw] K-contiguous;
 * 64 x 64 tiles; replaces the products + mmtg_splitk_finish launches behind generate.py:124's per-token model call):
 *   mode 0 "LN-fold"        C = act( rstd_m (A W^T - mu_m colsum_n) + bias_n ), act in {MMTG_EPI_NONE, MMTG_EPI_GELU}; C bf16
 *                           [M, ldc] or fp32 (out_f32).  W = the gamma-folded weight copy, colsum / bias = its column sums and
 *                           folded bias (mmtg_ln_fold_weights); mu_m / rstd_m from stats_in.
 *   mode 1 "LN-fold slabs"  split-K: C = fp32 [splits][M][ldc], slab s = rstd_m (A_s W_s^T) - [s == 0] rstd_m mu_m colsum_n
 *                           (the bias is left to the consumer, mmtg_decode_attn_split).
 *   mode 2 "reduce"         split-K reduced IN the kernel by the last-arriving wave of every 32 x 32 wave tile (partials in
 *                           `ws`: >= tiles * splits * 4096 floats; `counters`: >= 4 * tiles, zero on entry and on return):
 *                           C (bf16) = sum_s partial_s + bias + resid, and stats_out[m][n / 32] = (sum, sum of squares) of the
 *                           stored row segment -- the LayerNorm statistics of the new residual stream as 32-column partials.
 *                           With type_ids != null the residual is the GPT-2 input embedding instead of a tensor:
 *                           emb_pos[*pos_ptr][n] + emb_type[type_ids[m]][n] (rows of stride ldr; resid = null) -- projector_layer2
 *                           and model.py:282-288's position / token-type embedding add in one launch.
 * stats_in / stats_out: f32 [M][32][2] (N <= 1024 in mode 2); np_in = partials to add (1 after mmtg_decode_embed_add, N / 32 after a mode-2 product).  */
MMTG_API int mmtg_decode_gemm(int mode, int M, int N, int K, const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                     const float* bias, const float* colsum, const float* stats_in, int np_in, float eps, int act, int out_f32,
                     const void* resid, long ldr, float* stats_out, int splits, float* ws, long ws_floats, unsigned* counters,
                     long n_counters, const void* emb_pos, const void* emb_type, const long long* type_ids, const int* pos_ptr,
                     void* stream);
/* ---- split-precision ("bf16x3") token step, round 5.  The products of the decode step on (hi | lo) bf16 plane pairs of fp32 tensors
 * (three passes A_hi W_hi + A_lo W_hi + A_hi W_lo, fp32 accumulate; see mmtg_gemm_x3): fp32 residual stream, fp32 KV cache, fp32
 * logits -- greedy ids equal to the reference's fp32 loop (generate.py:117-142) at a multiple of the exact-fp32 kernels' speed.
 * mmtg_decode_gemm_x3: the three modes of mmtg_decode_gemm with A [M, K] / W [N, K] as plane pairs (lo plane planeA / planeW elements
 *   behind), K % 64 == 0.  mode 0: fp32 rows C (the logits) OR the activation as a plane pair Cp (ld = ldcp, lo plane planeC elements
 *   behind); mode 1: fp32 slabs C [splits][M][ldc]; mode 2: v = act(sum + bias) + residual with act in {NONE, TANH}, residual = resid
 *   (fp32, nullable) or wpe[*pos] + wte[type_ids[m]] (fp32 tables), written as fp32 rows C and / or a plane pair Cp, + the row
 *   statistics of v (stats_out nullable).
 * mmtg_ln_fold_weights_x3: W' = gamma (.) W re-split into a plane pair, colsum[n] = sum_k W'[n, k], bias_f = bias + beta W.
 * mmtg_decode_attn_split_x3: mmtg_decode_attn_split on an fp32 KV cache, the context rows written as a plane pair.
 * mmtg_decode_embed_x3: mmtg_decode_embed on the fp32 table / experience vectors, x written as a plane pair.                         */
MMTG_API int mmtg_decode_gemm_x3(int mode, int M, int N, int K, const void* A, long lda, long planeA, const void* W, long ldw, long planeW,
                        float* C, long ldc, void* Cp, long ldcp, long planeC, const float* bias, const float* colsum,
                        const float* stats_in, int np_in, float eps, int act, const float* resid, long ldr, float* stats_out,
                        int splits, float* ws, long ws_floats, unsigned* counters, long n_counters, const float* emb_pos,
                        const float* emb_type, const long long* type_ids, const int* pos_ptr, void* stream);
MMTG_API int mmtg_ln_fold_weights_x3(const void* W, long ldw, long planeW, const float* gamma, const float* beta, const float* bias,
                            void* Wf, long ldf, long planeF, float* colsum, float* bias_f, int N, int K, void* stream);
MMTG_API int mmtg_decode_attn_split_x3(const float* part, int splits, const float* bias, float* kcache, float* vcache, const int* keep,
                              long ldkeep, const int* pos_ptr, void* out_planes, long plane, int B, int nH, int dh, int Tmax, void* stream);
MMTG_API int mmtg_decode_embed_x3(const float* table, const long long* seq, long ldseq, const float* c, void* x_planes, long plane,
                         const int* pos_ptr, const long long* tpw_type, const long long* tpw_mask, long long* type_out, int* keep,
                         long ldkeep, int B, int P, int S, int E, int two_sents, int V, int sent, int max_sent_num, void* stream);

/* Wf[n, k] = gamma[k] W[n, k] (bf16), colsum[n] = sum_k Wf[n, k], bias_f[n] = bias[n] + sum_k beta[k] W[n, k]: the operands of the
 * LN-fold products, LN(x) W^T + b = rstd (x Wf^T - mu colsum) + bias_f.  W: [N, ldw] bf16 K-contiguous; bias may be null.   */
MMTG_API int mmtg_ln_fold_weights(const void* W, long ldw, const float* gamma, const float* beta, const float* bias, void* Wf,
                         float* colsum, float* bias_f, int N, int K, void* stream);
MMTG_API int mmtg_decode_attn(int dtype, const void* qkv, void* kcache, void* vcache, const int* keep, long ldkeep,
                     const int* pos_ptr, void* out, int B, int nH, int dh, int Tmax, void* stream);
/* the same, taking the c_attn product as MMTG_EPI_SPLIT slabs (part: f32 [splits][B][3*D]) plus its bias:
 * the slabs are summed in order and rounded to the storage type as mmtg_splitk_finish would (one launch less) */
MMTG_API int mmtg_decode_attn_split(int dtype, const float* part, int splits, const float* bias, void* kcache, void* vcache,
                           const int* keep, long ldkeep, const int* pos_ptr, void* out, int B, int nH, int dh, int Tmax,
                           void* stream);
/* pos_next (optional, must not alias pos_ptr): receives *pos_ptr + 1 -- the decoder keeps the position in a PAIR of slots
 * and alternates them step by step (every kernel of a step reads one slot, this last kernel writes the other), which removes
 * the one-thread mmtg_decode_advance launch from the captured step.                                                        */
MMTG_API int mmtg_decode_select(const float* logits, long ldl, int V, long long* seq, long ldseq, const int* pos_ptr,
                       int P, int sent, float temperature, float rep_penalty, int B, int* pos_next, void* stream);
/* decode_select with the stochastic selection above; the draw of position pos uses uniforms[pos * ldu + b]
 * (a [positions, ldu] device array filled before the graph is replayed).                                    */
MMTG_API int mmtg_decode_sample(const float* logits, long ldl, int V, long long* seq, long ldseq, const int* pos_ptr,
                       int P, int sent, float temperature, float rep_penalty, int top_k, float top_p,
                       const float* uniforms, long ldu, int B, int* pos_next, void* stream);
MMTG_API int mmtg_decode_advance(int* pos_ptr, void* stream);

/* ---- Round 6: the MLP of a GPT-2 block of the decode token step as ONE launch (bf16, n_embd = 768, M <= 256 rows) -- the
 * c_fc -> gelu_new -> mlp.c_proj -> + residual half of the transformers GPT-2 block the reference runs per token through
 * model.py:320-326 / generate.py:124; replaces the mode-0 + mode-2 mmtg_decode_gemm pair of the fused step.
 *   C = X + gelu( rstd (X W1f^T - mu colsum1) + bias1f ) W2t^T + bias2,   stats_out[m][p] = (sum, sum of squares) of C[m, 48 p .. 48 p + 48)
 * X [M, ldx] bf16: the un-normalised residual rows (also the residual added back); stats_in [M][32][2] their statistics partials
 * (np_in of them); W1f [3072, ldw1] / colsum1 / bias1f: the gamma-folded c_fc copy of mmtg_ln_fold_weights; W2t [768, ldw2]: the
 * [out,in] copy of mlp.c_proj; G [M, ldg] bf16: scratch for the hidden activations (written and read inside the launch); C [M, ldc] bf16
 * (may not alias X); stats_out: 16 partials per row.  ws: mmtg_decode_mlp_ws_floats(M) floats of fp32 partial products; sync:
 * mmtg_decode_mlp_sync_words() 64-bit words, ZERO before the first launch (the launch re-arms them; word [65] is the error report:
 * 2 = a wait ran into its 4 ms bound, 4 = a hand-off group was found on two XCDs in the plain mode -- results are then undefined,
 * the host must check and clear it).  The launch is 256 workgroups that must be co-resident, one per CU (mmtg_decode_mlp_census).
 * plain_handoff: 0 = the hidden activations cross workgroups through write-through stores + agent-scope loads (valid under any
 * workgroup placement); 1 = through the XCD's L2 (valid when workgroups b and b + 8 k share an XCD: checked per launch, error bit 4).
 * trace (nullable): 12 x 256 real-time-counter stamps (diagnostic).
 * mmtg_decode_mlp_census: out64[8 v + x] += workgroups b with b % 8 == v found on physical XCD x, for a launch of the same shape
 * (256 workgroups, one per CU): the placement the plain hand-off relies on holds when every row v has one entry of 32.          */
MMTG_API long mmtg_decode_mlp_ws_floats(int M);
MMTG_API int mmtg_decode_mlp_sync_words(void);
MMTG_API int mmtg_decode_mlp_census(unsigned* out64, void* stream);
MMTG_API int mmtg_decode_mlp(int M, int D, const void* X, long ldx, const float* stats_in, int np_in, float eps,
                    const void* W1f, long ldw1, const float* colsum1, const float* bias1f,
                    const void* W2t, long ldw2, const float* bias2, void* G, long ldg, void* C, long ldc,
                    float* stats_out, float* ws, long ws_floats, unsigned long long* sync, int plain_handoff,
                    unsigned long long* trace, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * The data-parallel exchange ("a second small ABI").  One process per GPU holds ONE RCCL communicator;
 * gradient buckets -- contiguous slices of the flat fp32 gradient, in the order the backward finishes them -- are SUM all-reduced
 * in place over xGMI.  Replaces torch.nn.DataParallel's per-step reduce_add_coalesced of every gradient to GPU 0 and its
 * parameter broadcast (reference src/train.py:112-114; generate.py:191, predict.py:195 wrap the model the same way).
 * Conventions that differ from the kernels above: the library owns the communicator, one side stream and two events (created by
 * mmtg_comm_init, released by mmtg_comm_destroy); RCCL is bound at first use (dlopen: the host process's copy when it has one,
 * else MMTG_RCCL_LIB, else librccl.so.1) -- without it these calls return MMTG_ERR_UNSUPPORTED, nothing falls back to the host.
 * Collectives must be issued in the same order on every rank, from one thread.
 *   mmtg_comm_unique_id   rank 0 draws the 128-byte id (ncclGetUniqueId); the host ships it to the other ranks (any side channel)
 *   mmtg_comm_init        collective: every rank calls it with the same id, on the device it has made current
 *   mmtg_comm_info        rank / world / device of the live communicator (world 0: none) and the bound RCCL's version code
 *   mmtg_allreduce_bucket in-place SUM of `count` elements (MMTG_F32 | MMTG_BF16) enqueued on `stream`
 *   mmtg_allreduce_bucket_async  the same on the library's side stream, ordered after everything already on `after_stream`
 *                         (the compute stream): the exchange runs beside the rest of the backward
 *   mmtg_comm_join        `stream` waits for every bucket enqueued with _async so far (no host wait)
 *   mmtg_comm_destroy     drains the side stream and releases the communicator (no-op without one)                            */
#define MMTG_COMM_ID_BYTES 128
MMTG_API int mmtg_comm_unique_id(void* id /* host, MMTG_COMM_ID_BYTES */);
MMTG_API int mmtg_comm_init(int rank, int world, const void* id /* host, MMTG_COMM_ID_BYTES */);
MMTG_API int mmtg_comm_info(int* rank, int* world, int* device, int* rccl_version);
MMTG_API int mmtg_allreduce_bucket(void* ptr, long count, int dtype, void* stream);
MMTG_API int mmtg_allreduce_bucket_async(void* ptr, long count, int dtype, void* after_stream);
MMTG_API int mmtg_comm_join(void* stream);
MMTG_API int mmtg_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* MMTG_HIP_H */
