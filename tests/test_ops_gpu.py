"""Per-kernel parity tests: every C-ABI entry point against a plain fp32
PyTorch/oracle restatement of the same op, on the same seeded inputs.

f32 mode is the exact-fp32 MFMA path (tolerances ~1e-5 relative); bf16 mode
feeds the reference the SAME bf16-rounded inputs and allows bf16 output
rounding (2^-8 relative) plus fp32-accumulation-order noise.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmtg_amd import hip  # noqa: E402
from oracle import mmtg_oracle as O  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(dtype)


def tol(dtype, k=1):
    """(atol, rtol) for an output of O(1) magnitude reduced over k terms."""
    if dtype == torch.float32:
        return 2e-5 * max(1.0, math.sqrt(k) / 8), 2e-5
    return 1e-2 * max(1.0, math.sqrt(k) / 16), 1.2e-2


def close(got, ref, dtype, k=1, name="", scale=None):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    atol, rtol = tol(dtype, k)
    s = float(ref.abs().max()) if scale is None else scale
    atol *= max(s, 1e-6)
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), "%s: %d/%d mismatches, max err %.3e (ref max %.3e, atol %.2e)" % (
        name, int(bad.sum()), bad.numel(), float(err.max()), s, atol)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("shape", [(200, 136, 96), (712, 768, 512), (1024, 256, 2048)])
@pytest.mark.parametrize("variant", ["dma2", "wide", "regstage", "no_tr", "occ4"])
def test_gemm_layouts(dtype, layout, shape, variant):
    no_tr = variant == "no_tr"
    if variant == "occ4" and dtype == torch.float32:
        pytest.skip("the single-stage 4-per-CU kernel is a bf16 configuration")
    if no_tr and (dtype == torch.float32 or layout == "NT"):
        pytest.skip("scalar-gather variant only differs for bf16 K-strided operands")
    if dtype == torch.float32 and variant != "dma2":
        pytest.skip("f32 has one pipeline (register-staged)")
    if variant == "wide" and layout == "TN":
        pytest.skip("192x128 tiles are built for the forward / dgrad layouts")
    M, N, K = shape
    a = rnd(M, K, dtype=dtype, seed=1)
    b = rnd(K, N, dtype=dtype, seed=2)
    ref = a.float() @ b.float()
    if layout == "NT":
        A, B, tA, tB = a, b.t().contiguous(), False, True
    elif layout == "NN":
        A, B, tA, tB = a, b, False, False
    else:
        A, B, tA, tB = a.t().contiguous(), b, True, False
    A, B = A.to(DEV), B.to(DEV)
    flags = {"dma2": hip.GEMM_NO_WIDE | hip.GEMM_NO_OCC4, "wide": hip.GEMM_WIDE, "regstage": hip.GEMM_REGSTAGE,
             "no_tr": hip.GEMM_NO_TR, "occ4": hip.GEMM_OCC4 | hip.GEMM_NO_WIDE}[variant]
    if layout == "TN":
        for splits in (1, 3):
            Cf = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
            hip.gemm(A, B, Cf, M, N, K, transA=tA, transB=tB, epi=hip.EPI_ATOMIC, alpha=2.0, splits=splits, flags=flags)
            close(Cf, 0.5 + 2.0 * ref, torch.float32 if dtype == torch.float32 else dtype, K, "gemm TN splits=%d" % splits)
    else:
        Cm = torch.empty(M, N, device=DEV, dtype=dtype)
        hip.gemm(A, B, Cm, M, N, K, transA=tA, transB=tB, flags=flags)
        close(Cm, ref, dtype, K, "gemm " + layout)


@pytest.mark.parametrize("layout", ["NT", "NN"])
@pytest.mark.parametrize("kernel", ["dma2", "wide", "occ4"])
@pytest.mark.parametrize("shape", [(712, 648, 200), (1300, 904, 64), (256, 2304, 128)])
def test_gemm_column_blocked_order_matches_plain(layout, kernel, shape):
    """The column-blocked item order (blocks of tile columns whose weight panels stay in L2; here forced to
    two columns, with a ragged last block) only renumbers the workgroups: results are bit-identical to the
    plain tile_n-fastest order and every output element is written."""
    M, N, K = shape
    a = rnd(M, K, dtype=torch.bfloat16, seed=5).to(DEV)
    b = rnd(K, N, dtype=torch.bfloat16, seed=6)
    B = (b.t().contiguous() if layout == "NT" else b).to(DEV)
    base = {"dma2": hip.GEMM_NO_WIDE | hip.GEMM_NO_OCC4, "wide": hip.GEMM_WIDE, "occ4": hip.GEMM_OCC4 | hip.GEMM_NO_WIDE}[kernel]
    outs = []
    for extra in (hip.GEMM_ROW_ORDER, hip.GEMM_COL_BLOCK):
        C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        hip.gemm(a, B, C, M, N, K, transB=layout == "NT", flags=base | extra)
        outs.append(C)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0], outs[1])
    close(outs[1], a.float().cpu() @ b.float(), torch.bfloat16, K, "column-blocked " + layout)


@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("K", [64, 200, 4744])
def test_gemm_persistent_pipeline_matches_plain(layout, K):
    """More work items than CU slots: the persistent kernel (next item's first K tile prefetched under
    the last one, epilogue deferred into the next item) must reproduce the one-item-per-workgroup
    kernel -- bit for bit where the K order is the same (stores), to rounding for the atomics.
    Ragged M, N and K; K = 64 is the single-K-tile item."""
    dtype = torch.bfloat16
    if layout == "TN":
        M, N, splits = 768, 776, 20
    else:
        M, N, splits = 4741, 2120, 1
    a = rnd(M, K, dtype=dtype, seed=11)
    b = rnd(K, N, dtype=dtype, seed=12)
    if layout == "NT":
        A, B, tA, tB = a, b.t().contiguous(), False, True
    elif layout == "NN":
        A, B, tA, tB = a, b, False, False
    else:
        A, B, tA, tB = a.t().contiguous(), b, True, False
    A, B = A.to(DEV), B.to(DEV)
    outs = []
    variants = [hip.GEMM_NO_PERSIST | hip.GEMM_NO_WIDE, hip.GEMM_PERSIST | hip.GEMM_NO_WIDE]
    if layout != "TN":
        variants.append(hip.GEMM_P256 | hip.GEMM_NO_WIDE)        # 256x128 tiles, 8 waves, one workgroup per CU (323 items on 256 slots)
    for flags in variants:
        if layout == "TN":
            C = torch.zeros(M, N, device=DEV, dtype=torch.float32)
            hip.gemm(A, B, C, M, N, K, transA=True, transB=False, epi=hip.EPI_ATOMIC, splits=splits, flags=flags)
            outs.append((C,))
        else:
            bias = rnd(N, seed=13).to(DEV)
            C = torch.empty(M, N, device=DEV, dtype=dtype)
            pre = torch.empty(M, N, device=DEV, dtype=dtype)
            hip.gemm(A, B, C, M, N, K, transA=tA, transB=tB, bias=bias, epi=hip.EPI_GELU, aux2=pre, flags=flags)
            Cf = torch.empty(M, N, device=DEV, dtype=torch.float32)
            hip.gemm(A, B, Cf, M, N, K, transA=tA, transB=tB, out_f32=True, flags=flags)
            outs.append((C, pre, Cf))
    ref = a.float() @ b.float()
    if layout == "TN":
        close(outs[1][0], ref, dtype, K, "persistent TN vs fp32")
        close(outs[1][0], outs[0][0], torch.float32, K, "persistent TN vs plain")
    else:
        close(outs[1][2], ref, dtype, K, "persistent %s vs fp32" % layout)
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert torch.equal(x, y)


@pytest.mark.parametrize("shape", [(2000, 768, 256), (1310, 2304, 128), (1100, 3080, 384), (4200, 1000, 640),
                                   (8200, 2312, 128), (7672, 5112, 256)])
def test_gemm_eight_phase_kernel_matches_older_kernels(shape):
    """The eight-phase kernel (256x256 / 192x256 tiles, two wave groups one barrier apart; the default for K-contiguous
    products with M >= 1024, N >= 256, K % 128 == 0) against the older kernels (GEMM_NO_P8) on the same inputs: every
    fused epilogue bit for bit -- plain, f32 output, bias + GELU with the saved pre-activation, residual + dropout,
    dGELU with its 64-row-band column sums, tanh, ROWDOT -- on ragged M / N edges; N = 768 and 2304 take 192-row tiles,
    N ~ 3072 and 1000 take 256-row tiles; the last two shapes have more tiles than CUs and also run the PERSISTENT form (430
    tiles of 192 rows, 600 tiles of 256 rows: two to three items per workgroup, the staging stream continuing across
    items).  Repeated launches must agree with themselves (race screen)."""
    M, N, K = shape
    dtype = torch.bfloat16
    a = rnd(M, K, dtype=dtype, seed=61).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=62, scale=0.3).to(DEV)
    bias = rnd(N, seed=63).to(DEV)
    aux = rnd(M, N, dtype=dtype, seed=64).to(DEV)

    def run(flags):
        outs = []
        c = torch.full((M, N), 9.0, device=DEV, dtype=dtype)
        hip.gemm(a, w, c, M, N, K, transB=True, bias=bias, flags=flags)
        outs.append(c)
        cf = torch.full((M, N), 9.0, device=DEV, dtype=torch.float32)
        hip.gemm(a, w, cf, M, N, K, transB=True, out_f32=True, flags=flags)
        outs.append(cf)
        c, pre = torch.full((M, N), 9.0, device=DEV, dtype=dtype), torch.full((M, N), 9.0, device=DEV, dtype=dtype)
        hip.gemm(a, w, c, M, N, K, transB=True, bias=bias, epi=hip.EPI_GELU, aux2=pre, flags=flags)
        outs += [c, pre]
        c = torch.full((M, N), 9.0, device=DEV, dtype=dtype)
        hip.gemm(a, w, c, M, N, K, transB=True, bias=bias, epi=hip.EPI_RESID, aux=aux, flags=flags, drop_p=0.1, drop_seed=77)
        outs.append(c)
        c = torch.full((M, N), 9.0, device=DEV, dtype=dtype)
        bands = torch.full(((M + 63) // 64, N), 9.0, device=DEV)
        hip.gemm(a, w, c, M, N, K, transB=True, epi=hip.EPI_DGELU, aux=aux, aux2=bands, flags=flags)
        outs += [c, bands]
        c = torch.full((M, N), 9.0, device=DEV, dtype=dtype)
        hip.gemm(a, w, c, M, N, K, transB=True, bias=bias, epi=hip.EPI_TANH, flags=flags)
        outs.append(c)
        if N % 64 == 0:
            c = torch.full((M, N), 9.0, device=DEV, dtype=dtype)
            delta = torch.full((M, N // 64), 9.0, device=DEV)
            hip.gemm(a, w, c, M, N, K, transB=True, epi=hip.EPI_ROWDOT, aux=aux, aux2=delta, flags=flags)
            outs += [c, delta]
        return outs

    old = run(hip.GEMM_NO_P8)
    close(old[1], a.float().cpu() @ w.float().cpu().t(), dtype, K, "older kernels vs fp32")
    for rep in range(6):
        # default routing; the dGELU product on the eight-phase kernel too (twice); the persistent form (opt-in); 288-row tiles
        # (round 4; twice -- the dGELU product with column sums keeps its 256-row tiles there)
        new = run((0, hip.GEMM_P8, hip.GEMM_P8, hip.GEMM_P8 | hip.GEMM_PERSIST, hip.GEMM_P8_288, hip.GEMM_P8_288)[rep])
        for i, (x, y) in enumerate(zip(old, new)):
            assert torch.equal(x, y), (shape, rep, i)


def test_gemm_eight_phase_tile_rule_follows_the_cu_budget():
    """mmtg_gemm_cu_budget (set by mmtg_amd.ddp when RCCL kernels run beside the backward): with all 256 CUs an
    N = 768 product of 15104 rows takes 192-row tiles (79 x 3 = 237 workgroups, one round); with 32 CUs left to the
    collectives 237 would spill into a second round, so the rule falls back to 256-row tiles (59 x 3 = 177).  The
    result does not depend on the choice."""
    M, N, K = 15104, 768, 256
    dtype = torch.bfloat16
    a = rnd(M, K, dtype=dtype, seed=81).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=82).to(DEV)
    outs, counts = [], []
    try:
        for budget in (0, -32):
            hip.gemm_cu_budget(budget)
            buf = torch.zeros(4096, 6, device=DEV, dtype=torch.int64)
            c = torch.empty(M, N, device=DEV, dtype=dtype)
            hip.gemm_trace(buf)
            hip.gemm(a, w, c, M, N, K, transB=True)
            torch.cuda.synchronize()
            hip.gemm_trace(None)
            outs.append(c)
            counts.append(int((buf[:, 3] != 0).sum()))
    finally:
        hip.gemm_cu_budget(0)
        hip.gemm_trace(None)
    assert torch.equal(outs[0], outs[1])
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert counts == [237, 177], counts


@pytest.mark.parametrize("shape", [(768, 2304, 1536, 3), (3072, 768, 2048, 4), (520, 776, 1280, 5), (256, 256, 128, 1)])
def test_gemm_eight_phase_weight_gradient(shape):
    """Slab weight gradients (transA, K-split slabs, MMTG_EPI_SPLIT) on the eight-phase K-strided kernel: with one split the
    slab is bit-equal to the older single-stage kernel's (same K order); with several the slabs partition K in 128-deep units
    -- their ordered sum matches the fp32 product to bf16-product accuracy and the older kernel's sum to fp32 rounding, the
    launch is reproducible bit for bit, ragged M / N edges included."""
    M, N, K, splits = shape
    dtype = torch.bfloat16
    a = rnd(K, M, dtype=dtype, seed=71).to(DEV)
    b = rnd(K, N, dtype=dtype, seed=72).to(DEV)
    ref = a.float().cpu().t() @ b.float().cpu()

    def run(flags, sp):
        part = torch.full((sp, M, N), 5.0, device=DEV, dtype=torch.float32)
        hip.gemm(a, b, part, M, N, K, transA=True, transB=False, epi=hip.EPI_SPLIT, out_f32=True, splits=sp, flags=flags)
        return part

    P8 = hip.GEMM_P8      # (opt-in for weight gradients: the older single-stage kernel measured faster inside the step)
    one_old, one_new = run(hip.GEMM_NO_P8, 1), run(P8, 1)
    assert torch.equal(one_old, one_new)
    close(one_new[0], ref, dtype, K, "one slab vs fp32")
    if splits > 1:
        new = run(P8, splits)
        for _ in range(3):
            assert torch.equal(run(P8, splits), new)
        close(new.sum(0), ref, dtype, K, "slab sum vs fp32")
        close(new.sum(0), run(hip.GEMM_NO_P8, splits).sum(0), torch.float32, K, "slab sum vs older kernel")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(dtype):
    M, N, K = 300, 192, 128
    a = rnd(M, K, dtype=dtype, seed=3).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=4, scale=0.2).to(DEV)
    bias = rnd(N, seed=5).to(DEV)
    aux = rnd(M, N, dtype=dtype, seed=6).to(DEV)
    acc = a.float() @ w.float().t()
    lin = acc + bias

    def run(epi, **kw):
        out = torch.empty(M, N, device=DEV, dtype=kw.pop("odt", dtype))
        hip.gemm(a, w, out, M, N, K, transB=True, epi=epi, **kw)
        return out

    if dtype == torch.bfloat16:   # the 192x128 configuration shares the epilogue code: spot-check it too
        pre_w = torch.empty(M, N, device=DEV, dtype=dtype)
        close(run(hip.EPI_GELU, bias=bias, aux2=pre_w, flags=hip.GEMM_WIDE), O.gelu_new(lin), dtype, K, "wide gelu")
        close(run(hip.EPI_RESID, bias=bias, aux=aux, flags=hip.GEMM_WIDE), lin + aux.float(), dtype, K, "wide resid")
        # ... and the single-stage 4-per-CU kernel (its epilogue requests aux one band ahead)
        f4 = hip.GEMM_OCC4
        close(run(hip.EPI_GELU, bias=bias, aux2=pre_w, flags=f4), O.gelu_new(lin), dtype, K, "occ4 gelu")
        close(run(hip.EPI_RESID, bias=bias, aux=aux, flags=f4), lin + aux.float(), dtype, K, "occ4 resid")
        b4 = torch.full(((M + 63) // 64, N), float("nan"), device=DEV, dtype=torch.float32)
        d4 = run(hip.EPI_DGELU, aux=aux, aux2=b4, flags=f4)
        close(b4.sum(0), d4.float().sum(0), torch.float32, M, "occ4 dgelu column sums")
    close(run(hip.EPI_NONE, bias=bias), lin, dtype, K, "bias")
    close(run(hip.EPI_NONE, bias=bias, out_f32=True, odt=torch.float32), lin, torch.float32 if dtype == torch.float32 else dtype, K, "out_f32")
    close(run(hip.EPI_TANH, bias=bias), torch.tanh(lin), dtype, K, "tanh")
    pre = torch.empty(M, N, device=DEV, dtype=dtype)
    close(run(hip.EPI_GELU, bias=bias, aux2=pre), O.gelu_new(lin), dtype, K, "gelu")
    close(pre, lin, dtype, K, "gelu pre-activation")
    close(run(hip.EPI_RESID, bias=bias, aux=aux), lin + aux.float(), dtype, K, "resid")
    x = aux.float().requires_grad_(True)
    O.gelu_new(x).sum().backward()
    close(run(hip.EPI_DGELU, aux=aux), acc * x.grad, dtype, K, "dgelu")
    # ... with the fused column sums of the stored output (the c_fc bias gradient), accumulated (+=)
    bands = torch.full(((M + 63) // 64, N), float("nan"), device=DEV, dtype=torch.float32)
    dg = run(hip.EPI_DGELU, aux=aux, aux2=bands)
    for i in range(bands.shape[0]):          # per 64-row band, every entry written
        close(bands[i], dg[64 * i:64 * i + 64].float().sum(0), torch.float32, 64, "dgelu column sums, band %d" % i)
    close(run(hip.EPI_DTANH, aux=aux), acc * (1 - aux.float() ** 2), dtype, K, "dtanh")
    # fused dropout: deterministic mask, ~p zeros, survivors scaled by 1/(1-p)
    zero = torch.zeros(M, N, device=DEV, dtype=dtype)
    d1 = run(hip.EPI_RESID, bias=bias, aux=zero, drop_p=0.25, drop_seed=7)
    d2 = run(hip.EPI_RESID, bias=bias, aux=zero, drop_p=0.25, drop_seed=7)
    assert torch.equal(d1, d2)
    frac = float((d1 == 0).float().mean())
    assert 0.2 < frac < 0.3, frac
    keep = d1 != 0
    close(d1[keep], (lin / 0.75)[keep], dtype, K, "dropout survivors")
    # the standalone mask kernel reproduces the epilogue's mask
    ones = torch.ones(M * N, device=DEV, dtype=dtype)
    mk = torch.empty_like(ones)
    hip.dropout_apply(ones, mk, M * N, 0.25, 7)
    assert bool((mk.view(M, N) != 0)[keep].all())
    assert int(((mk.view(M, N) != 0) != keep).sum()) <= 8   # only where the linear output itself rounds to 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_gelu_saved_derivative_pair(dtype):
    """MMTG_GEMM_GELU_GRAD (round 3): the fc1 epilogue stores gelu_new'(pre-activation) instead of the pre-activation and the dGELU
    product multiplies by it as stored -- the pair gives the same forward output and the same d(pre-activation) as the
    pre-activation-saving pair, to the rounding of the stored derivative; both routes of the bf16 kernels (default and
    GEMM_NO_P8) agree bit for bit."""
    M, N, K = 1200, 512, 256
    a = rnd(M, K, dtype=dtype, seed=71).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=72, scale=0.2).to(DEV)
    bias = rnd(N, seed=73).to(DEV)
    dy = rnd(M, K, dtype=dtype, seed=74).to(DEV)           # product 2: d(pre) = (dy @ w2^T) * gelu'(pre), w2 [N, K]
    pre_ref = a.float() @ w.float().t() + bias
    xg = pre_ref.clone().requires_grad_(True)
    O.gelu_new(xg).sum().backward()
    gprime_ref = xg.grad
    outs = []
    for flags in ((0, hip.GEMM_NO_P8) if dtype == torch.bfloat16 else (0,)):
        g_u, u = torch.empty(M, N, device=DEV, dtype=dtype), torch.empty(M, N, device=DEV, dtype=dtype)
        hip.gemm(a, w, g_u, M, N, K, transB=True, bias=bias, epi=hip.EPI_GELU, aux2=u, flags=flags)
        g_d, gd = torch.empty(M, N, device=DEV, dtype=dtype), torch.empty(M, N, device=DEV, dtype=dtype)
        hip.gemm(a, w, g_d, M, N, K, transB=True, bias=bias, epi=hip.EPI_GELU, aux2=gd, flags=flags | hip.GEMM_GELU_GRAD)
        assert torch.equal(g_u, g_d)                        # the forward output does not depend on what is saved
        close(gd, gprime_ref, dtype, 1, "saved gelu'")
        du_u, du_d = torch.empty(M, N, device=DEV, dtype=dtype), torch.empty(M, N, device=DEV, dtype=dtype)
        hip.gemm(dy, w, du_u, M, N, K, transB=True, epi=hip.EPI_DGELU, aux=u, flags=flags)
        hip.gemm(dy, w, du_d, M, N, K, transB=True, epi=hip.EPI_DGELU, aux=gd, flags=flags | hip.GEMM_GELU_GRAD)
        ref = (dy.float() @ w.float().t()) * gprime_ref.to(DEV)
        close(du_u, ref, dtype, K, "dGELU from the pre-activation")
        close(du_d, ref, dtype, K, "dGELU from the saved derivative")
        outs.append((gd, du_d))
    for x, y in zip(outs[0], outs[-1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_weight_gradient_ragged_tokens(dtype):
    """X^T dY with a token count (reduction length) that is no multiple of anything."""
    Mtok, Kin, Nout = 708, 256, 384
    x = rnd(Mtok, Kin, dtype=dtype, seed=8)
    dy = rnd(Mtok, Nout, dtype=dtype, seed=9)
    ref = x.float().t() @ dy.float()
    dW = torch.zeros(Kin, Nout, device=DEV)
    hip.gemm(x.to(DEV), dy.to(DEV), dW, Kin, Nout, Mtok, transA=True, transB=False, lda=Kin, ldb=Nout,
             epi=hip.EPI_ATOMIC, splits=4)
    close(dW, ref, torch.float32 if dtype == torch.float32 else dtype, Mtok, "wgrad")


@pytest.mark.parametrize("shape", [(256, 768, 3072), (200, 2304, 768), (37, 136, 96), (256, 13440, 768)])
def test_gemm_skinny_config(shape):
    """256x32-tile small-M configuration (decode): forced and default selection, every epilogue it serves."""
    M, N, K = shape
    dtype = torch.bfloat16
    a = rnd(M, K, dtype=dtype, seed=11).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=12, scale=0.1).to(DEV)
    bias = rnd(N, seed=13).to(DEV)
    aux = rnd(M, N, dtype=dtype, seed=14).to(DEV)
    lin = a.float() @ w.float().t() + bias
    for flags in (hip.GEMM_SKINNY, 0, hip.GEMM_NO_SKINNY):
        out = torch.empty(M, N, device=DEV, dtype=dtype)
        hip.gemm(a, w, out, M, N, K, transB=True, bias=bias, flags=flags)
        close(out, lin, dtype, K, "skinny flags=%d" % flags)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    hip.gemm(a, w, out, M, N, K, transB=True, bias=bias, epi=hip.EPI_RESID, aux=aux, flags=hip.GEMM_SKINNY)
    close(out, lin + aux.float(), dtype, K, "skinny resid")
    pre = torch.empty(M, N, device=DEV, dtype=dtype)
    hip.gemm(a, w, out, M, N, K, transB=True, bias=bias, epi=hip.EPI_GELU, aux2=pre, flags=hip.GEMM_SKINNY)
    close(out, O.gelu_new(lin), dtype, K, "skinny gelu")
    of = torch.empty(M, N, device=DEV, dtype=torch.float32)
    hip.gemm(a, w, of, M, N, K, transB=True, out_f32=True, flags=hip.GEMM_SKINNY)
    close(of, lin - bias, dtype, K, "skinny f32 out")


@pytest.mark.parametrize("shape", [(256, 768, 3072, 12), (200, 2304, 768, 4), (64, 776, 200, 3)])
def test_gemm_split_k_finish(shape):
    """Deterministic split-K for decode: EPI_SPLIT slabs + splitk_finish (bias, residual, fused LayerNorm).
    (64, 776, 200, 3): the third K slice lies past the end and must contribute zeros."""
    M, N, K, S = shape
    dtype = torch.bfloat16
    a = rnd(M, K, dtype=dtype, seed=31).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=32, scale=0.2).to(DEV)
    bias = rnd(N, seed=33).to(DEV)
    res = rnd(M, N, dtype=dtype, seed=34).to(DEV)
    gam, bet = (1 + 0.1 * rnd(N, seed=35)).to(DEV), (0.1 * rnd(N, seed=36)).to(DEV)
    part = torch.full((S, M, N), float("nan"), device=DEV, dtype=torch.float32)
    outs = []
    for _ in range(2):
        hip.gemm(a, w, part, M, N, K, transB=True, epi=hip.EPI_SPLIT, out_f32=True, splits=S)
        out = torch.empty(M, N, device=DEV, dtype=dtype)
        kw = dict(ln_gamma=gam, ln_beta=bet, ln_out=torch.empty(M, N, device=DEV, dtype=dtype)) if N <= 1024 else {}
        hip.splitk_finish(part, S, M, N, out, bias=bias, epi=hip.EPI_RESID, aux=res, **kw)
        outs.append((out, kw.get("ln_out")))
    ref = a.float() @ w.float().t() + bias + res.float()
    close(outs[0][0], ref, dtype, K, "split-K + finish")
    assert torch.equal(outs[0][0], outs[1][0])                       # fixed summation order
    if N <= 1024:
        x = outs[0][0].float()
        ln = torch.nn.functional.layer_norm(x, (N,), gam, bet, 1e-5)
        close(outs[0][1], ln, dtype, 1, "fused LayerNorm of the finished row")
    g = torch.empty(M, N, device=DEV, dtype=dtype)
    hip.splitk_finish(part, S, M, N, g, bias=bias, epi=hip.EPI_GELU)
    close(g, O.gelu_new(a.float() @ w.float().t() + bias), dtype, K, "split-K + gelu")


@pytest.mark.parametrize("shape", [(32, 768, 3072, 24), (40, 2304, 768, 8), (64, 776, 200, 3)])
def test_gemm_split_k_slabs_fp32(shape):
    """The fp32 kernel's slab epilogue (decode in the fp32 parity mode): Conv1D weights as stored ([in, out], NN layout),
    K slices of whole 32-deep tiles, slices past the end store zeros; splitk_finish adds them in order."""
    M, N, K, S = shape
    a = rnd(M, K, seed=41).to(DEV)
    w = rnd(K, N, seed=42, scale=0.2).to(DEV)
    bias = rnd(N, seed=43).to(DEV)
    res = rnd(M, N, seed=44).to(DEV)
    part = torch.full((S, M, N), float("nan"), device=DEV, dtype=torch.float32)
    outs = []
    for _ in range(2):
        hip.gemm(a, w, part, M, N, K, ldb=N, ldc=N, epi=hip.EPI_SPLIT, out_f32=True, splits=S)
        out = torch.empty(M, N, device=DEV)
        hip.splitk_finish(part, S, M, N, out, bias=bias, epi=hip.EPI_RESID, aux=res)
        outs.append(out)
    assert torch.isfinite(part).all()
    kper = -(-(-(-K // S)) // 32) * 32
    for s in range(S):      # every slab is the product over its own K slice
        lo, hi = min(K, s * kper), min(K, (s + 1) * kper)
        ref_s = a[:, lo:hi].double() @ w[lo:hi].double()
        close(part[s], ref_s.float(), torch.float32, max(hi - lo, 1), "slab %d" % s)
    ref = (a.double() @ w.double() + bias.double() + res.double()).float()
    close(outs[0], ref, torch.float32, K, "fp32 split-K + finish")
    assert torch.equal(outs[0], outs[1])


def test_gemm_rowdot_epilogue_is_attention_delta():
    M, N, K = 600, 192, 128
    dtype = torch.bfloat16
    a = rnd(M, K, dtype=dtype, seed=21).to(DEV)
    w = rnd(N, K, dtype=dtype, seed=22, scale=0.2).to(DEV)
    o = rnd(M, N, dtype=dtype, seed=23).to(DEV)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    delta = torch.full((M, N // 64), float("nan"), device=DEV)
    hip.gemm(a, w, out, M, N, K, transB=True, epi=hip.EPI_ROWDOT, aux=o, aux2=delta)
    ref = a.float() @ w.float().t()
    close(out, ref, dtype, K, "rowdot out")
    dref = (out.float() * o.float()).view(M, N // 64, 64).sum(-1)
    close(delta, dref, torch.float32, 64, "rowdot delta")


def test_zero_ranges_touches_exactly_the_listed_ranges():
    """mmtg_zero_ranges: one launch zeroes a list of (first element, count) ranges of an fp32 buffer and nothing else."""
    n = 3_000_000
    x = torch.arange(1, n + 1, device=DEV, dtype=torch.float32)
    ranges = [(0, 4), (64, 1024), (4096, 4), (100_000, 2_000_000), (2_999_996, 4)]
    desc = torch.tensor(ranges, dtype=torch.int64, device=DEV)
    hip.zero_ranges(x, desc, len(ranges))
    ref = torch.arange(1, n + 1, dtype=torch.float32)
    for o, c in ranges:
        ref[o:o + c] = 0
    assert torch.equal(x.cpu(), ref)


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(16, 16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple"):
        hip.gemm(a, a, a, 16, 16, 6, transB=True, lda=16, ldb=16)
    with pytest.raises(RuntimeError, match="atomic"):
        hip.gemm(a, a, a, 16, 16, 16, transA=True, transB=False)


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cols,rows", [(512, 333), (768, 333), (768, 5001), (1024, 2100), (512, 2048)])
def test_layernorm(dtype, cols, rows):
    # 333 rows: the small-launch kernels; >= 2048 rows: the grid-stride bf16 kernels (a full wave per row with
    # 16-byte + 8-byte vectors per lane, next row prefetched) incl. a ragged last sweep; 1024 columns: GPT-2-medium
    x = rnd(rows, cols, dtype=dtype, seed=1, scale=2.0)
    g = 1 + 0.1 * rnd(cols, seed=2)
    b = 0.1 * rnd(cols, seed=3)
    dy = rnd(rows, cols, dtype=dtype, seed=4)
    dres = rnd(rows, cols, dtype=dtype, seed=5)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-5)
    yr.backward(dy.float())
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    y = torch.empty_like(xd)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    hip.layernorm_fwd(xd, y, gd, bd, mean, rstd, rows, cols)
    close(y, yr, dtype, 1, "ln fwd")
    close(mean, x.float().mean(-1), torch.float32, 1, "ln mean")
    dx = torch.empty_like(xd)
    dg = torch.full((cols,), 1.0, device=DEV)
    db = torch.full((cols,), -1.0, device=DEV)
    hip.layernorm_bwd(dy.to(DEV), xd, gd, mean, rstd, dres.to(DEV), dx, dg, db, rows, cols)
    close(dx, xr.grad + dres.float(), dtype, 4, "ln dx")
    close(dg - 1.0, gr.grad, torch.float32, rows, "ln dgamma")
    close(db + 1.0, br.grad, torch.float32, rows, "ln dbeta")
    cs = torch.zeros(cols, device=DEV)
    hip.colsum(dy.to(DEV), rows, cols, cs)
    close(cs, dy.float().sum(0), torch.float32, rows, "colsum")
    # fused tail: masked copy (same mask as the standalone kernel) + its column sum
    dxm = torch.empty_like(xd)
    dcs = torch.full((cols,), 2.0, device=DEV)
    dg.zero_(); db.zero_()
    hip.layernorm_bwd(dy.to(DEV), xd, gd, mean, rstd, dres.to(DEV), dx, dg, db, rows, cols,
                      dx_masked=dxm, drop_p=0.2, drop_seed=99, dcolsum=dcs)
    ref_m = torch.empty_like(xd)
    hip.dropout_apply(dx, ref_m, rows * cols, 0.2, 99)
    assert torch.equal(dxm, ref_m)
    close(dcs - 2.0, ref_m.float().sum(0), torch.float32, rows, "ln fused colsum")
    close(dg, gr.grad, torch.float32, rows, "ln dgamma (fused)")
    dcs.zero_()
    hip.layernorm_bwd(dy.to(DEV), xd, gd, mean, rstd, dres.to(DEV), dx, dg, db, rows, cols, dcolsum=dcs)
    close(dcs, dx.float().sum(0), torch.float32, rows, "ln colsum without mask")


def test_batched_column_sums_equal_the_one_call_forms_bit_for_bit():
    """Round 6: mmtg_colsum_batch sums many small fp32 row sets in ONE launch, each in mmtg_colsum's order -- so (a) a batch of plain
    items equals mmtg_colsum per item, (b) mmtg_layernorm_bwd_partial + the batch equals mmtg_layernorm_bwd (d gamma, d beta, the fused
    column sum of the masked d(x); same d(x) and masked copy), (c) mmtg_attn_bwd with MMTG_ATTN_DBIAS_ROWS + the batch equals the call
    that sums its bias rows itself -- bit for bit, the accumulate (+=) semantics included; more than 64 items take several launches."""
    # (a) 70 items of mixed shapes, one of them a strided view
    items, want = [], []
    keep_alive = []
    for i in range(70):
        M, N = [(236, 3072), (64, 2304), (756, 768), (5, 8), (2048, 64), (33, 100)][i % 6]
        ldx = N + (8 if i % 3 == 0 else 0)
        X = rnd(M, ldx, seed=100 + i).to(DEV)
        out = torch.full((N,), float(i), device=DEV)
        ref = torch.full((N,), float(i), device=DEV)
        hip.colsum(X, M, N, ref, ldx=ldx)
        items.append((X.data_ptr(), out.data_ptr(), ldx, M, N))
        keep_alive.append((X, out))
        want.append(ref)
    hip.colsum_batch(items)
    for (X, out), ref in zip(keep_alive, want):
        assert torch.equal(out, ref)
    with pytest.raises(RuntimeError, match="colsum_batch"):
        hip.colsum_batch([(keep_alive[0][0].data_ptr(), keep_alive[0][1].data_ptr(), 3072, 4096, 3072)])     # > 2048 rows
    # (b) LayerNorm backward, bf16 rows at the training shape's width, dropout tail on
    rows, cols = 5001, 768
    x = rnd(rows, cols, dtype=torch.bfloat16, seed=1, scale=2.0).to(DEV)
    dy = rnd(rows, cols, dtype=torch.bfloat16, seed=4).to(DEV)
    dres = rnd(rows, cols, dtype=torch.bfloat16, seed=5).to(DEV)
    g = (1 + 0.1 * rnd(cols, seed=2)).to(DEV)
    b = (0.1 * rnd(cols, seed=3)).to(DEV)
    y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    hip.layernorm_fwd(x, y, g, b, mean, rstd, rows, cols)
    res = []
    for partial in (False, True):
        dx, dxm = torch.empty_like(x), torch.empty_like(x)
        dg, db, dcs = (torch.full((cols,), v, device=DEV) for v in (1.0, -1.0, 2.0))
        if partial:
            ws = torch.full((int(hip.lib().mmtg_layernorm_bwd_ws(rows, cols)),), float("nan"), device=DEV)
            nb = hip.layernorm_bwd_partial(dy, x, g, mean, rstd, dres, dx, rows, cols, ws, dx_masked=dxm, drop_p=0.2, drop_seed=99,
                                           want_colsum=True)
            assert 0 < nb <= 1024
            hip.colsum_batch([(ws.data_ptr() + 4 * q * cols, t.data_ptr(), 3 * cols, nb, cols) for q, t in enumerate((dg, db, dcs))])
        else:
            hip.layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dg, db, rows, cols, dx_masked=dxm, drop_p=0.2, drop_seed=99, dcolsum=dcs)
        res.append((dx, dxm, dg, db, dcs))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    assert float((res[0][2] - 1.0).abs().max()) > 0
    # (c) whole-head attention backward
    B, T, nH, dh = 4, 236, 3, 64
    D = nH * dh
    qkv = rnd(B, T, 3 * D, dtype=torch.bfloat16, seed=7).to(DEV)
    keep = torch.ones(B, T, dtype=torch.int32, device=DEV)
    dout = rnd(B, T, D, dtype=torch.bfloat16, seed=8).to(DEV)
    out, lse = torch.empty(B, T, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, nH, T, device=DEV)
    hip.attn_fwd(qkv, keep, out, lse, B, T, nH, dh)
    assert hip.attn_bwd_dbias_rows(hip.BF16, B, T) == B and hip.attn_bwd_dbias_rows(hip.F32, B, T) == 0
    assert hip.attn_bwd_dbias_rows(hip.BF16, B, 1024) == 0           # the tiled kernels reduce inside the call
    got = []
    for defer in (False, True):
        delta, dq32 = torch.empty(B * T, nH, device=DEV), torch.empty(B * T, D, device=DEV)
        dqkv = torch.empty(B, T, 3 * D, device=DEV, dtype=torch.bfloat16)
        dbias = torch.full((3 * D,), 0.5, device=DEV)
        rws = torch.full((hip.attn_bwd_bias_rows(B, T, hip.BF16), 3 * D), float("nan"), device=DEV)
        hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, dbias=dbias, dbias_ws=rws,
                     flags=hip.ATTN_DBIAS_ROWS if defer else 0)
        if defer:
            assert torch.equal(dbias, torch.full_like(dbias, 0.5))       # nothing summed yet
            hip.colsum_batch([(rws.data_ptr(), dbias.data_ptr(), 3 * D, B, 3 * D)])
        got.append((dqkv, dbias))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    assert float((got[0][1] - 0.5).abs().max()) > 0


# ------------------------------------------------------------------ attention
def ref_attention(qkv, keep, nH):
    B, T, D3 = qkv.shape
    D = D3 // 3
    dh = D // nH
    q, k, v = (t.view(B, T, nH, dh).transpose(1, 2) for t in qkv.split(D, -1))
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    allow = causal[None, None] & keep.bool()[:, None, None, :]
    sc = sc.masked_fill(~allow, float("-inf"))
    p = torch.softmax(sc, -1)
    return (p @ v).transpose(1, 2).reshape(B, T, D), torch.logsumexp(sc, -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [236, 104, 64, 300, 1, 17, 1024, 256, 240, 33, 257, 384, 512, 497])
def test_attention(dtype, T):
    B, nH, dh = 2, 3, 64
    D = nH * dh
    qkv = rnd(B, T, 3 * D, dtype=dtype, seed=T, scale=1.0)
    keep = torch.ones(B, T, dtype=torch.int32)
    keep[0, 9:15] = 0
    keep[1, 20:41] = 0
    if T > 3:               # (T = 1: the single-token sequence; 1024 = GPT-2's n_positions, four key blocks in bf16)
        keep[1, T - 3:] = 0
    dout = rnd(B, T, D, dtype=dtype, seed=T + 1)
    qr = qkv.float().requires_grad_(True)
    oref, lref = ref_attention(qr, keep, nH)
    oref.backward(dout.float())
    qd = qkv.to(DEV)
    kd = keep.to(DEV)
    out = torch.empty(B, T, D, device=DEV, dtype=dtype)
    lse = torch.empty(B, nH, T, device=DEV)
    hip.attn_fwd(qd, kd, out, lse, B, T, nH, dh)
    close(out, oref, dtype, 16, "attn out")
    close(lse, lref, torch.float32 if dtype == torch.float32 else dtype, 16, "attn lse")
    delta = torch.empty(B * T, nH, device=DEV)
    dq32 = torch.empty(B * T, D, device=DEV)
    dqkv = torch.full((B, T, 3 * D), float("nan"), device=DEV, dtype=dtype)
    dbias = torch.full((3 * D,), 0.5, device=DEV, dtype=torch.float32)
    rows = torch.full((hip.attn_bwd_bias_rows(B, T, hip.dt(qd)), 3 * D), float("nan"), device=DEV, dtype=torch.float32)
    hip.attn_bwd(qd, kd, out, dout.to(DEV), lse, delta, dq32, dqkv, B, T, nH, dh, dbias=dbias, dbias_ws=rows)
    dbias_at = torch.full((3 * D,), 0.5, device=DEV, dtype=torch.float32)     # atomics variant (no scratch)
    hip.attn_bwd(qd, kd, out, dout.to(DEV), lse, delta, dq32, torch.empty_like(dqkv), B, T, nH, dh, dbias=dbias_at)
    close(dbias_at, dbias, torch.float32, B * T, "attn fused d(bias): atomics vs partial rows")
    g = qr.grad
    for i, nm in enumerate("qkv"):
        close(dqkv[..., i * D:(i + 1) * D], g[..., i * D:(i + 1) * D], dtype, 64, "attn d" + nm,
              scale=float(g.abs().max()))
    # fused c_attn bias gradient: += column sums over all tokens of d(qkv) as stored
    close(dbias, 0.5 + dqkv.float().sum((0, 1)), torch.float32, B * T, "attn fused d(bias)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [64, 200, 320, 512])
def test_attention_dropout_consistency(dtype, T):
    """Forward/backward share one counter-based mask: check against an fp32
    reference that uses the mask recovered from probe runs (V = identity rows, 64 keys per run).  T = 64 / 200: the
    whole-head kernels of up to 256 positions (bf16), 320 / 512: their 16-wave form with the compact keep-bit matrix;
    fp32 and T > 512: the tiled kernels' per-element hash."""
    B, nH, dh = 1, 1, 64
    D = nH * dh
    p = 0.3
    qkv = rnd(B, T, 3 * D, dtype=dtype, seed=5, scale=0.5)
    keep = torch.ones(B, T, dtype=torch.int32)
    out = torch.empty(B, T, D, device=DEV, dtype=dtype)
    lse = torch.empty(B, nH, T, device=DEV)
    mask = torch.zeros(T, T)
    for c0 in range(0, T, dh):
        probe = qkv.clone()
        probe[..., :2 * D] = 0                       # uniform attention: P = 1/(t+1)
        probe[..., 2 * D:] = 0
        n = min(dh, T - c0)
        probe[0, c0:c0 + n, 2 * D:2 * D + n] = torch.eye(n).to(dtype)     # V rows c0.. = unit vectors -> out[t, j] = mask[t, c0 + j] P / (1-p)
        hip.attn_fwd(probe.to(DEV), keep.to(DEV), out, lse, B, T, nH, dh, drop_p=p, drop_seed=11)
        mask[:, c0:c0 + n] = (out[0, :, :n].float().cpu() != 0).float()
    tri = torch.tril(torch.ones(T, T))
    frac = float((mask * tri).sum() / tri.sum())
    assert 0.6 < frac < 0.8, frac
    qr = qkv.float().requires_grad_(True)
    q, k, v = qr[0].split(D, -1)
    sc = (q @ k.t()) / 8.0
    sc = sc.masked_fill(tri == 0, float("-inf"))
    pr = torch.softmax(sc, -1) * mask / (1 - p)
    oref = pr @ v
    dout = rnd(B, T, D, dtype=dtype, seed=6)
    oref.backward(dout[0].float())
    hip.attn_fwd(qkv.to(DEV), keep.to(DEV), out, lse, B, T, nH, dh, drop_p=p, drop_seed=11)
    close(out[0], oref, dtype, 16, "attn dropout out")
    delta = torch.empty(B, nH, T, device=DEV)
    dq32 = torch.empty(B * T, D, device=DEV)
    dqkv = torch.empty(B, T, 3 * D, device=DEV, dtype=dtype)
    hip.attn_bwd(qkv.to(DEV), keep.to(DEV), out, dout.to(DEV), lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=p, drop_seed=11)
    close(dqkv, qr.grad, dtype, 64, "attn dropout grads")


# ------------------------------------------------------------------ conditioning front end
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [5, 2])
def test_embed_condition_and_segment_sum(dtype, S):
    B, P, E, V, ts = 3, 15, 2048, 160, 44
    L = ts * S + 1
    table = rnd(V, E, dtype=dtype, seed=1)
    c = rnd(B, S, E, dtype=dtype, seed=2)
    g = torch.Generator().manual_seed(3)
    topic = torch.randint(0, V, (B, P), generator=g)
    targ = torch.randint(0, V, (B, L), generator=g)
    ref = O.condition_embeddings(table.float(), topic, targ, c.float(), ts)
    x = torch.empty(B, P + L, E, device=DEV, dtype=dtype)
    hip.embed_condition(table.to(DEV), topic.to(DEV), targ.to(DEV), c.to(DEV), x, B, P, L, S, E, ts, V)
    close(x, ref, dtype, 1, "embed_condition")
    H = 512
    gr = rnd(B, P + L, H, dtype=dtype, seed=4)
    out = torch.empty(B, S, H, device=DEV, dtype=dtype)
    hip.segment_sum(gr.to(DEV), out, B, P, L, S, H, ts)
    refs = torch.stack([gr.float()[:, P + k * ts:P + min((k + 1) * ts, L)].sum(1) for k in range(S)], 1)
    close(out, refs, dtype, ts, "segment_sum")


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_add_fwd_bwd(dtype):
    B, T, D, NT = 3, 104, 768, 11
    M = B * T
    g = rnd(M, D, dtype=dtype, seed=1)
    wpe = rnd(256, D, dtype=dtype, seed=2)
    wte = rnd(40, D, dtype=dtype, seed=3)
    ty = torch.randint(0, NT, (M,), generator=torch.Generator().manual_seed(4))
    pos = torch.arange(M) % T
    ref = g.float() + wpe.float()[pos] + wte.float()[ty]
    h = torch.empty(M, D, device=DEV, dtype=dtype)
    hip.embed_add(g.to(DEV), wpe.to(DEV), wte.to(DEV), ty.to(DEV), h, M, T, D)
    close(h, ref, dtype, 1, "embed_add")
    dh = rnd(M, D, dtype=dtype, seed=5)
    dwpe = torch.zeros(256, D, device=DEV)
    dwte = torch.zeros(40, D, device=DEV)
    hip.embed_add_bwd(dh.to(DEV), ty.to(DEV), dwpe, dwte, M, T, D, NT)
    rp = torch.zeros(256, D).index_add_(0, pos, dh.float())
    rt = torch.zeros(40, D).index_add_(0, ty, dh.float())
    close(dwpe, rp, torch.float32, B, "dwpe")
    close(dwte, rt, torch.float32, M // NT, "dwte")


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stage", [1, 2])
def test_loss_fwd_bwd(dtype, stage):
    B, P, L, V = 5, 15, 89, 333
    Vpad = 384
    T = P + L
    gen = torch.Generator().manual_seed(stage)
    logits = torch.randn(B, T, V, generator=gen) * 2
    logits[1] += 6 * torch.nn.functional.one_hot(torch.randint(0, V, (T,), generator=gen), V)
    topic = torch.randint(0, V, (B, P), generator=gen)
    targ = torch.randint(0, V, (B, L), generator=gen)
    # make sample 1 nearly perfectly predicted so p -> 1 exercises the 1-p+eps branch
    lab = torch.cat([topic, targ], 1)
    logits[1, :-1] = logits[1, :-1] + 9 * torch.nn.functional.one_hot(lab[1, 1:], V)
    ratings = torch.tensor([1, 2, 3, 5, 4])
    lr = logits.clone().requires_grad_(True)
    ref = O.my_loss(lr, targ, ratings, stage, P)
    ref.backward()
    lm_ref = O.lm_loss_shifted(logits, lab)
    pad = torch.zeros(B * T, Vpad)
    pad[:, :V] = logits.view(-1, V)
    ld = pad.to(DEV)
    nll = torch.empty(B * T, device=DEV)
    lse = torch.empty(B * T, device=DEV)
    ce = torch.empty(B, device=DEV)
    coef = torch.empty(B, device=DEV)
    sc = torch.zeros(2, device=DEV)
    hip.loss_fwd(ld, Vpad, V, topic.to(DEV), targ.to(DEV), ratings.to(DEV), stage, False, B, P, L, float(B),
                 nll, lse, ce, coef, sc)
    assert abs(sc[0].item() - ref.item()) < 2e-5 * max(1, abs(ref.item())), (sc[0].item(), ref.item())
    assert abs(sc[1].item() - lm_ref.item()) < 2e-5 * max(1, abs(lm_ref.item()))
    close(lse.view(B, T), torch.logsumexp(logits, -1), torch.float32, 1, "lse")
    dl = torch.full((B * T, Vpad), float("nan"), device=DEV, dtype=dtype)
    hip.loss_bwd(ld, Vpad, V, topic.to(DEV), targ.to(DEV), lse, coef, 1.0, B, P, L, dl, Vpad, Vpad)
    assert float(dl[:, V:].float().abs().max()) == 0.0
    close(dl[:, :V].view(B, T, V), lr.grad, dtype, 1, "dlogits", scale=float(lr.grad.abs().max()))
    # inference branch: dummy zero labels (model.py:314)
    hip.loss_fwd(ld, Vpad, V, topic.to(DEV), targ.to(DEV), None, stage, True, B, P, L, float(B), nll, lse, ce, coef, sc)
    z = O.lm_loss_shifted(logits, torch.zeros(B, T, dtype=torch.long))
    assert abs(sc[1].item() - z.item()) < 2e-5 * abs(z.item())


def test_loss_on_bf16_logits_in_place():
    """bf16-stored logits (the bf16 trainer path): the loss of the ROUNDED logits is reproduced to fp32 accuracy
    (2e-5 rel; the only difference to the fp32 path is the storage rounding of the inputs), and the gradient
    written over the logits buffer equals the separately stored one bit for bit."""
    B, P, L, V, Vpad, stage = 4, 15, 45, 333, 384, 2
    T = P + L
    gen = torch.Generator().manual_seed(7)
    logits = (torch.randn(B, T, V, generator=gen) * 2).bfloat16()
    topic = torch.randint(0, V, (B, P), generator=gen)
    targ = torch.randint(0, V, (B, L), generator=gen)
    ratings = torch.tensor([1, 5, 3, 4])
    lr = logits.float().clone().requires_grad_(True)
    ref = O.my_loss(lr, targ, ratings, stage, P)
    ref.backward()
    pad = torch.zeros(B * T, Vpad, dtype=torch.bfloat16)
    pad[:, :V] = logits.view(-1, V)
    ld = pad.to(DEV)
    nll, lse = torch.empty(B * T, device=DEV), torch.empty(B * T, device=DEV)
    ce, coef, sc = torch.empty(B, device=DEV), torch.empty(B, device=DEV), torch.zeros(2, device=DEV)
    hip.loss_fwd(ld, Vpad, V, topic.to(DEV), targ.to(DEV), ratings.to(DEV), stage, False, B, P, L, float(B), nll, lse, ce, coef, sc)
    assert abs(sc[0].item() - ref.item()) < 2e-5 * max(1, abs(ref.item())), (sc[0].item(), ref.item())
    close(lse.view(B, T), torch.logsumexp(logits.float(), -1), torch.float32, 1, "lse of bf16 logits")
    dl = torch.full((B * T, Vpad), float("nan"), device=DEV, dtype=torch.bfloat16)
    hip.loss_bwd(ld, Vpad, V, topic.to(DEV), targ.to(DEV), lse, coef, 1.0, B, P, L, dl, Vpad, Vpad)
    close(dl[:, :V].view(B, T, V), lr.grad, torch.bfloat16, 1, "dlogits", scale=float(lr.grad.abs().max()))
    hip.loss_bwd(ld, Vpad, V, topic.to(DEV), targ.to(DEV), lse, coef, 1.0, B, P, L, ld, Vpad, Vpad)
    assert torch.equal(ld, dl)


# ------------------------------------------------------------------ encoder pieces
@pytest.mark.parametrize("dtype", DTYPES)
def test_gru_cell(dtype):
    B, H = 7, 512
    gi = rnd(B, 3 * H, dtype=dtype, seed=1)
    gh = rnd(B, 3 * H, dtype=dtype, seed=2)
    hp = rnd(B, H, dtype=dtype, seed=3)
    dh = rnd(B, H, seed=4)
    gir, ghr, hpr = (t.float().requires_grad_(True) for t in (gi, gh, hp))
    i_r, i_z, i_n = gir.chunk(3, -1)
    h_r, h_z, h_n = ghr.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    href = (1 - z) * n + z * hpr
    href.backward(dh)
    h = torch.empty(B, H, device=DEV, dtype=dtype)
    save = torch.empty(4, B, H, device=DEV)
    hip.gru_cell_fwd(gi.to(DEV), gh.to(DEV), hp.to(DEV), h, save, B, H)
    close(h, href, dtype, 1, "gru h")
    dgi = torch.empty(B, 3 * H, device=DEV, dtype=dtype)
    dgh = torch.empty(B, 3 * H, device=DEV, dtype=dtype)
    dhp = torch.empty(B, H, device=DEV)
    hip.gru_cell_bwd(dh.to(DEV), save, hp.to(DEV), dgi, dgh, dhp, B, H)
    close(dgi, gir.grad, dtype, 1, "gru dgi")
    close(dgh, ghr.grad, dtype, 1, "gru dgh")
    close(dhp, hpr.grad, torch.float32, 1, "gru dh_prev")
    # first step: h_prev = None means zeros
    hip.gru_cell_fwd(gi.to(DEV), gh.to(DEV), None, h, save, B, H)
    close(h, ((1 - z) * n).detach(), dtype, 1, "gru h (h0=0)")
    # step 0 of the sequence: the recurrent pre-activation is one bias row for every b (row stride 0)
    row = gh[:1].contiguous()
    h0 = torch.empty(B, H, device=DEV, dtype=dtype)
    hip.gru_cell_fwd(gi.to(DEV), row.to(DEV), None, h0, save, B, H, ld_gh=0)
    hb = torch.empty(B, H, device=DEV, dtype=dtype)
    hip.gru_cell_fwd(gi.to(DEV), row.expand(B, 3 * H).contiguous().to(DEV), None, hb, save, B, H)
    assert torch.equal(h0, hb)
    # fused backward: dh_t assembled in the kernel from rows (storage type) + carry + the carry product's slabs
    rows = rnd(B, H, dtype=dtype, seed=9).to(DEV)
    carry = rnd(B, H, seed=10).to(DEV)
    part = rnd(3, B, H, seed=11).to(DEV)
    dtot = rows.float() + carry + part.sum(0)
    hip.gru_cell_fwd(gi.to(DEV), gh.to(DEV), hp.to(DEV), h, save, B, H)
    ref = [torch.empty(B, 3 * H, device=DEV, dtype=dtype), torch.empty(B, 3 * H, device=DEV, dtype=dtype), torch.empty(B, H, device=DEV)]
    hip.gru_cell_bwd(dtot, save, hp.to(DEV), ref[0], ref[1], ref[2], B, H)
    got = [torch.empty_like(ref[0]), torch.empty_like(ref[1]), carry.clone()]
    hip.gru_cell_bwd_fused(rows, H, got[2], part, 3, save, hp.to(DEV), got[0], got[1], got[2], B, H)
    for a, b_, nm in zip(got, ref, ("dgi", "dgh", "dh_prev")):
        close(a, b_, dtype if nm != "dh_prev" else torch.float32, 1, "gru fused " + nm)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [5, 2, 8])
def test_alpha_attention(dtype, S):
    B, H, heads = 4, 512, 4
    qkv = rnd(B * S, 3 * H, dtype=dtype, seed=S, scale=0.7)
    dctx = rnd(B, S, H, dtype=dtype, seed=S + 1)
    pri = O.gaussian_priors(S)
    qr = qkv.float().requires_grad_(True)
    q, k, v = (t.view(B, S, heads, H // heads).permute(0, 2, 1, 3) for t in qr.view(B, S, 3 * H).split(H, -1))
    sc = q @ k.transpose(-1, -2) / math.sqrt(H // heads)
    pr = torch.softmax(sc, -1)
    klr = ((pri * (pri.log() - pr.log())).sum((0, 1, 3)) / B).mean()
    cr = (pr @ v).permute(0, 2, 1, 3).reshape(B, S, H)
    dkl = 0.37
    ((cr * dctx.float()).sum() + dkl * klr).backward()
    ctx = torch.empty(B * S, H, device=DEV, dtype=dtype)
    probs = torch.empty(B, heads, S, S, device=DEV)
    kl = torch.zeros(1, device=DEV)
    hip.alpha_attn_fwd(qkv.to(DEV), pri.to(DEV), ctx, probs, kl, B, S, H, heads)
    close(ctx.view(B, S, H), cr, dtype, S, "alpha ctx")
    close(probs, pr, torch.float32 if dtype == torch.float32 else dtype, 1, "alpha probs")
    assert abs(kl.item() - klr.item()) < (2e-5 if dtype == torch.float32 else 2e-2) * max(1, abs(klr.item()))
    dqkv = torch.empty(B * S, 3 * H, device=DEV, dtype=dtype)
    hip.alpha_attn_bwd(qkv.to(DEV), pri.to(DEV), probs, dctx.to(DEV), dkl, dqkv, B, S, H, heads)
    close(dqkv, qr.grad, dtype, S * 4, "alpha dqkv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [5, 2])
def test_beta_fuser(dtype, S):
    B, H = 6, 512
    topic = rnd(B, H, dtype=dtype, seed=1)
    img = rnd(B, S, H, dtype=dtype, seed=2)
    txt = rnd(B, S, H, dtype=dtype, seed=3)
    aw = rnd(S, H, seed=4, scale=0.1)
    ab = rnd(S, seed=5)
    do = rnd(B, S, H, dtype=dtype, seed=6)
    tr, ir, xr = (t.float().requires_grad_(True) for t in (topic, img, txt))
    awr, abr = aw.clone().requires_grad_(True), ab.clone().requires_grad_(True)
    outs = []
    for i in range(S):
        src = torch.stack([tr, ir[:, i], xr[:, i]], 1)
        a = torch.softmax((src @ awr[i]) + abr[i], -1)
        outs.append((a.unsqueeze(1) @ src).squeeze(1))
    oref = torch.stack(outs, 1)
    (oref * do.float()).sum().backward()
    o = torch.empty(B * S, H, device=DEV, dtype=dtype)
    a = torch.empty(B, S, 3, device=DEV)
    args = (topic.to(DEV), img.to(DEV).view(B * S, H), txt.to(DEV).view(B * S, H), aw.to(DEV))
    hip.beta_fuse_fwd(*args, ab.to(DEV), o, a, B, S, H)
    close(o.view(B, S, H), oref, dtype, 1, "beta o")
    dt_ = torch.zeros(B, H, device=DEV)
    di = torch.empty(B * S, H, device=DEV, dtype=dtype)
    dx = torch.empty(B * S, H, device=DEV, dtype=dtype)
    daw = torch.zeros(S, H, device=DEV)
    dab = torch.zeros(S, device=DEV)
    hip.beta_fuse_bwd(*args, a, do.to(DEV).view(B * S, H), dt_, di, dx, daw, dab, B, S, H)
    close(dt_, tr.grad, torch.float32 if dtype == torch.float32 else dtype, S, "beta dtopic")
    close(di.view(B, S, H), ir.grad, dtype, 1, "beta dimg")
    close(dx.view(B, S, H), xr.grad, dtype, 1, "beta dtxt")
    close(daw, awr.grad, torch.float32 if dtype == torch.float32 else dtype, B, "beta datt_w")
    assert float(dab.abs().max()) < 1e-4


# ------------------------------------------------------------------ optimizer
def test_adamw_clip_and_casts():
    n = 100003
    p = rnd(n, seed=1)
    g = rnd(n, seed=2, scale=3.0)
    pr = [p.clone()]
    gr = [g.clone()]
    norm = O.clip_grad_norm(gr, 1.0)
    st = {}
    O.adamw_hf_step(pr, gr, st, 1e-3, 1)
    g2 = rnd(n, seed=3, scale=1e-3)
    gr2 = [g2.clone()]
    O.clip_grad_norm(gr2, 1.0)
    O.adamw_hf_step(pr, gr2, st, 5e-4, 2)
    pd, gd = p.to(DEV), g.to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    pc = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    ns = torch.zeros(1, device=DEV)
    hip.sumsq(gd, n, ns)
    assert abs(math.sqrt(ns.item()) - norm.item()) < 1e-4 * norm.item()
    hip.adamw(pd, gd, m, v, pc, n, 1e-3, 0.9, 0.999, 1e-6, 0.0, 1, ns, 1.0)
    ns.zero_()
    g2d = g2.to(DEV)
    hip.sumsq(g2d, n, ns)
    hip.adamw(pd, g2d, m, v, pc, n, 5e-4, 0.9, 0.999, 1e-6, 0.0, 2, ns, 1.0)
    close(pd, pr[0], torch.float32, 1, "adamw params")
    assert torch.equal(pc.float().cpu(), pd.cpu().to(torch.bfloat16).float())
    x = rnd(37, 50, seed=4).to(DEV)
    y = torch.empty(37, 64, device=DEV, dtype=torch.bfloat16)
    hip.cast_pad_rows(x, 50, y, 64, 37, 50)
    assert torch.equal(y[:, :50], x.to(torch.bfloat16)) and float(y[:, 50:].float().abs().max()) == 0
    z = torch.empty(37 * 64, device=DEV)
    hip.cast_to_f32(y, z, 37 * 64)
    assert torch.equal(z.view(37, 64), y.float())
    a = rnd(1000, seed=5).to(DEV)
    b = rnd(1000, seed=6).to(DEV)
    ref = a + 0.5 * b
    hip.axpy_f32(a, b, 0.5, 1000)
    close(a, ref, torch.float32, 1, "axpy")


@pytest.mark.parametrize("dtype", DTYPES)
def test_decode_attention_step(dtype):
    """KV-cached single-token attention: appends this token's K/V at `pos`, attends over keys 0..pos under
    the key mask; the split variant takes the c_attn product as fp32 slabs + bias and must agree exactly."""
    B, nH, dh, Tmax, pos = 3, 2, 64, 40, 17
    D = nH * dh
    kc = rnd(B, nH, Tmax, dh, dtype=dtype, seed=41).to(DEV)
    vc = rnd(B, nH, Tmax, dh, dtype=dtype, seed=42).to(DEV)
    keep = torch.ones(B, Tmax, dtype=torch.int32)
    keep[1, 3:7] = 0
    keep = keep.to(DEV)
    posd = torch.tensor([pos], dtype=torch.int32, device=DEV)
    slabs = rnd(3, B, 3 * D, seed=43, scale=0.5).to(DEV)
    bias = rnd(3 * D, seed=44, scale=0.1).to(DEV)
    qkv = (slabs.sum(0) + bias).to(dtype)
    # reference on the rounded qkv
    q, k, v = (qkv.float()[:, i * D:(i + 1) * D].view(B, nH, dh) for i in range(3))
    kr, vr = kc.float().clone(), vc.float().clone()
    kr[:, :, pos], vr[:, :, pos] = k, v
    sc = torch.einsum("bhd,bhtd->bht", q, kr[:, :, :pos + 1]) * 0.125
    sc = sc.masked_fill(keep[:, None, :pos + 1] == 0, float("-inf"))
    ref = torch.einsum("bht,bhtd->bhd", torch.softmax(sc, -1), vr[:, :, :pos + 1]).reshape(B, D)
    outs = []
    for split in (False, True):
        k1, v1 = kc.clone(), vc.clone()
        out = torch.empty(B, D, device=DEV, dtype=dtype)
        if split:
            hip.decode_attn_split(slabs, 3, bias, k1, v1, keep, posd, out, B, nH, dh, Tmax)
        else:
            hip.decode_attn(qkv, k1, v1, keep, posd, out, B, nH, dh, Tmax)
        close(out, ref, dtype, dh, "decode attention split=%s" % split)
        assert torch.equal(k1[:, :, pos].reshape(B, D), qkv[:, D:2 * D]) and torch.equal(v1[:, :, pos].reshape(B, D), qkv[:, 2 * D:])
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_batch(dtype):
    """Batched weight transpose (the [out,in] copies of Conv1D weights): exact, ragged 64x64 tiles."""
    shapes = [(768, 2304), (72, 40), (3072, 768), (8, 200)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += r * c
    src = rnd(total, dtype=dtype, seed=21).to(DEV)
    dst = torch.zeros(total, dtype=dtype, device=DEV)
    desc = torch.tensor([[o, r, c, o] for o, (r, c) in zip(offs, shapes)], dtype=torch.int64, device=DEV)
    hip.transpose_batch(src, dst, desc, len(shapes), max(r for r, _ in shapes), max(c for _, c in shapes))
    for o, (r, c) in zip(offs, shapes):
        assert torch.equal(dst[o:o + r * c].view(c, r), src[o:o + r * c].view(r, c).t())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,splits", [(768, 2304, 4744, 7), (200, 136, 1000, 3), (3072, 768, 2048, 5), (128, 128, 64, 4)])
def test_weight_gradient_slabs(M, N, K, splits, dtype):
    """Weight gradients as K-split slabs (plain stores) + mmtg_slab_sum: same product as the fp32-atomic
    epilogue (tolerance: fp32 summation order only, 2e-5 relative to the largest entry), bit-identical
    between runs, accumulate / overwrite semantics, slabs past the end of K hold zeros.  bf16: the single-stage LDS-DMA kernel;
    fp32 (round 6): the exact-fp32 register-staged kernel -- the cross-check mode's weight gradients without atomics."""
    A = rnd(K, M, dtype=dtype, seed=31).to(DEV)
    B = rnd(K, N, dtype=dtype, seed=32).to(DEV)
    ref = (A.double().t() @ B.double()).float()
    scale = ref.abs().max().item()
    atom = torch.zeros(M, N, device=DEV)
    hip.gemm(A, B, atom, M, N, K, transA=True, epi=hip.EPI_ATOMIC, splits=splits)
    outs = []
    for _ in range(2):
        part = torch.full((splits, M, N), float("nan"), device=DEV)
        hip.gemm(A, B, part, M, N, K, transA=True, epi=hip.EPI_SPLIT, out_f32=True, splits=splits)
        assert torch.isfinite(part).all()
        acc = torch.ones(M, N, device=DEV)
        hip.slab_sum(part, splits, M * N, acc, M * N, accumulate=True)
        over = torch.full((M, N), 7.0, device=DEV)
        hip.slab_sum(part, splits, M * N, over, M * N, accumulate=False)
        assert torch.equal(over, part.sum(0)) or (over - part.sum(0)).abs().max().item() <= 2e-6 * scale
        assert (acc - 1 - over).abs().max().item() <= 1e-6 * scale
        outs.append(over)
    assert torch.equal(outs[0], outs[1])
    assert (outs[0] - ref).abs().max().item() <= 2e-5 * scale
    assert (outs[0] - atom).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("config", [0, 1])
@pytest.mark.parametrize("K,splits", [(4744, 2), (1000, 3), (15104, 2), (640, 1), (2048, 5)])
def test_wgrad_group_in_kernel_reduction(K, splits, config):
    """mmtg_wgrad_group: several weight-gradient products (ragged tile edges, leading dimensions wider than the extents, ragged
    last K tile) in one launch, K splits reduced inside the kernel by the last-arriving wave of every quadrant.  Against the
    fp32 product of the same bf16 inputs (fp32 summation order only), bit-identical between runs, equal to an explicit
    sum of the per-split products in split order, overwrite / accumulate semantics, counters left zeroed."""
    shapes = [(768, 384, 768, 384), (200, 136, 208, 144), (384, 768, 384, 768), (128, 128, 128, 128)]   # M, N, lda, ldb
    ops = []
    for i, (M, N, lda, ldb) in enumerate(shapes):
        A = rnd(K, lda, dtype=torch.bfloat16, seed=40 + i).to(DEV)
        B = rnd(K, ldb, dtype=torch.bfloat16, seed=50 + i).to(DEV)
        ops.append((A, B, M, N, lda, ldb))
    tiles, nws, ncnt = hip.wgrad_group_sizes([(M, N) for (_, _, M, N, _, _) in ops], splits, config)
    ws = torch.full((nws,), float("nan"), device=DEV)
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=DEV)
    kq = 128 if config else 64
    kper = -(-(-(-K // splits)) // kq) * kq
    outs = []
    for rep in range(3):
        Cs = [torch.full((M, N + 8), 7.0, device=DEV) for (_, _, M, N, _, _) in ops]      # ldc = N + 8: the pad columns stay 7
        probs = [(A, B, C_, M, N, lda, ldb, N + 8) for (A, B, M, N, lda, ldb), C_ in zip(ops, Cs)]
        hip.wgrad_group(probs, K, splits, ws, cnt, accumulate=False, config=config)
        assert int(cnt.abs().sum()) == 0
        outs.append([c.clone() for c in Cs])
        for (A, B, M, N, lda, ldb), C_ in zip(ops, Cs):
            assert (C_[:, N:] == 7.0).all()
            ref = A[:, :M].float().t() @ B[:, :N].float()
            scale = ref.abs().max().item()
            assert (C_[:, :N] - ref).abs().max().item() <= 2e-5 * scale
            # the reduction order is the split order: equal to summing per-split fp32 products in that order up to the
            # kernel's own in-tile accumulation order (tolerance), and independent of who arrived last (bit-equal runs)
            acc = torch.zeros(M, N, device=DEV)
            for s0 in range(0, K, kper):
                acc += A[s0:s0 + kper, :M].float().t() @ B[s0:s0 + kper, :N].float()
            assert (C_[:, :N] - acc).abs().max().item() <= 2e-5 * scale
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)
    # accumulate = 1 adds to what is there
    Cs = [torch.ones(M, N + 8, device=DEV) for (_, _, M, N, _, _) in ops]
    probs = [(A, B, C_, M, N, lda, ldb, N + 8) for (A, B, M, N, lda, ldb), C_ in zip(ops, Cs)]
    hip.wgrad_group(probs, K, splits, ws, cnt, accumulate=True, config=config)
    for C_, o, (_, _, M, N, _, _) in zip(Cs, outs[0], ops):
        scale = o[:, :N].abs().max().item()
        assert (C_[:, :N] - 1.0 - o[:, :N]).abs().max().item() <= 1e-6 * scale
        assert (C_[:, N:] == 1.0).all()


@pytest.mark.parametrize("config", [0, 1])
@pytest.mark.parametrize("K,splits", [(1000, 3), (200, 1), (4744, 2)])
def test_wgrad_group_ragged_k_never_reads_past_the_operands(K, splits, config):
    """A ragged last K slice with NaN-filled memory DIRECTLY behind both operands (same allocation, so the addresses are
    mapped): rows at or past K must be requested out of range per lane -- a kernel that relied on the buffer descriptor's
    range check for the K-tile offset it carries in the scalar offset would add NaNs to the gradient."""
    M, N = 384, 256
    bufA = torch.full((K + 256, M), float("nan"), dtype=torch.bfloat16, device=DEV)
    bufB = torch.full((K + 256, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    bufA[:K] = rnd(K, M, dtype=torch.bfloat16, seed=3).to(DEV)
    bufB[:K] = rnd(K, N, dtype=torch.bfloat16, seed=4).to(DEV)
    A, B = bufA[:K], bufB[:K]
    tiles, nws, ncnt = hip.wgrad_group_sizes([(M, N)], splits, config)
    ws = torch.zeros(nws, device=DEV)
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=DEV)
    C_ = torch.zeros(M, N, device=DEV)
    hip.wgrad_group([(A, B, C_, M, N, M, N, N)], K, splits, ws, cnt, accumulate=False, config=config)
    ref = A.float().t() @ B.float()
    assert torch.isfinite(C_).all()
    assert (C_ - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("config", [0, 1])
def test_wgrad_group_repeated_launches_stay_bit_equal(config):
    """Race screen at the training shapes: the four weight gradients of a GPT-2-base block (432 / 108 tiles x 2 K halves, every
    wave tile reduced by whichever wave arrives last) launched 40 times back to back -- every result bit-equal to the first."""
    K, D = 15104, 768
    g = torch.Generator(device=DEV).manual_seed(3)
    mk = lambda n: (torch.randn(K, n, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    m2, du, gact, dy, ctx, dy2, a1, dqkv = mk(D), mk(4 * D), mk(4 * D), mk(D), mk(D), mk(D), mk(D), mk(3 * D)
    shapes = [(m2, du, D, 4 * D), (gact, dy, 4 * D, D), (ctx, dy2, D, D), (a1, dqkv, D, 3 * D)]
    tiles, nws, ncnt = hip.wgrad_group_sizes([(M, N) for (_, _, M, N) in shapes], 2, config)
    assert tiles == (108 if config else 432)
    ws = torch.empty(nws, device=DEV)
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=DEV)
    first = None
    for it in range(40):
        Cs = [torch.empty(M, N, device=DEV) for (_, _, M, N) in shapes]
        hip.wgrad_group([(A, B, C_, M, N) for (A, B, M, N), C_ in zip(shapes, Cs)], K, 2, ws, cnt, config=config)
        if first is None:
            first = Cs
            for (A, B, M, N), C_ in zip(shapes, Cs):
                ref = A.float().t() @ B.float()
                assert (C_ - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        else:
            for a, b in zip(first, Cs):
                assert torch.equal(a, b), it
    assert int(cnt.abs().sum()) == 0



@pytest.mark.parametrize("M", [256, 200, 32])
def test_decode_gemm_ln_fold_and_in_kernel_reduce(M):
    """mmtg_decode_gemm + mmtg_ln_fold_weights (the fused decode step): (a) the folded operands against their definition;
    (b) mode 0 -- LN(x) W^T + b applied algebraically from row-statistics partials -- against the explicit LayerNorm in fp32 on
    the same bf16 operands (bf16 output rounding + the bf16 rounding of gamma (.) W), with GELU and with fp32 output; (c) mode 1
    slabs + folded bias = mode 0; (d) mode 2: split-K reduced in the kernel + bias + residual, bit-equal between runs and
    between split counts' own repeats, counters left zeroed, and its statistics partials = sums over the STORED bf16 rows."""
    K, N = 768, 2304
    NP = hip.DG_NP
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(M, K, generator=g) * 2.0 + torch.randn(M, 1, generator=g)).to(torch.bfloat16)
    x[:, 5] *= 20.0                                        # an outlier channel, as GPT-2's residual stream has
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    gamma, beta = 1.0 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    bias = 0.1 * torch.randn(N, generator=g)
    xd, Wd = x.to(DEV), W.to(DEV)
    Wf = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
    c, bf = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    hip.ln_fold_weights(Wd, gamma.to(DEV), beta.to(DEV), bias.to(DEV), Wf, c, bf, N, K)
    Wf_ref = (gamma[None, :] * W.float()).to(torch.bfloat16)
    assert torch.equal(Wf.cpu(), Wf_ref)
    close(c, Wf_ref.float().sum(1), torch.float32, K, "colsum")
    close(bf, bias + W.float() @ beta, torch.float32, K, "folded bias")
    # statistics partials of x as stored (what the producing kernels emit)
    xf = x.float()
    st = torch.zeros(M, NP, 2)
    st[:, :K // 32, 0] = xf.view(M, K // 32, 32).sum(2)
    st[:, :K // 32, 1] = (xf * xf).view(M, K // 32, 32).sum(2)
    std = st.to(DEV)
    mu, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    ln = (xf - mu) / torch.sqrt(var + 1e-5)
    ref = (ln * gamma + beta) @ W.float().t() + bias
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.decode_gemm(0, xd, Wf, out, M, N, K, bias=bf, colsum=c, stats_in=std, np_in=K // 32)
    close(out, ref, torch.bfloat16, K, "LN-fold")
    out32 = torch.empty(M, N, device=DEV)
    hip.decode_gemm(0, xd, Wf, out32, M, N, K, bias=bf, colsum=c, stats_in=std, np_in=NP, out_f32=True)
    exact = ((xf - mu) / torch.sqrt(var + 1e-5)) @ Wf_ref.float().t() + bf.cpu()        # the algebra on the folded operands
    assert (out32.cpu() - exact).abs().max().item() <= 2e-3 * exact.abs().max().item()
    outg = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.decode_gemm(0, xd, Wf, outg, M, N, K, bias=bf, colsum=c, stats_in=std, np_in=NP, act=hip.EPI_GELU)
    close(outg, O.gelu_new(ref), torch.bfloat16, K, "LN-fold + GELU")
    for S in (2, 3):
        slabs = torch.full((S, M, N), float("nan"), device=DEV)
        hip.decode_gemm(1, xd, Wf, slabs, M, N, K, colsum=c, stats_in=std, np_in=NP, out_f32=True, splits=S)
        assert (slabs.sum(0) + bf - out32).abs().max().item() <= 1e-3 * exact.abs().max().item()
    # ---- mode 2
    K2, N2 = 3072, 768
    A = (torch.randn(M, K2, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    W2 = (torch.randn(N2, K2, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    b2 = (0.1 * torch.randn(N2, generator=g)).to(DEV)
    resid = x.to(DEV)
    ref2 = A.float() @ W2.float().t() + b2 + resid.float()
    tiles = -(-M // 64) * (N2 // 64)
    for S in (1, 4, 8):
        ws = torch.full((tiles * S * 4096,), float("nan"), device=DEV)
        cnt = torch.zeros(tiles * 4, dtype=torch.int32, device=DEV)
        outs = []
        for rep in range(3):
            xo = torch.full((M, N2), 7.0, dtype=torch.bfloat16, device=DEV)
            so = torch.full((M, NP, 2), float("nan"), device=DEV)
            hip.decode_gemm(2, A, W2, xo, M, N2, K2, bias=b2, resid=resid, stats_out=so, splits=S, ws=ws, counters=cnt)
            assert int(cnt.abs().sum()) == 0
            outs.append((xo, so))
        close(outs[0][0], ref2, torch.bfloat16, K2, "reduce S=%d" % S)
        npo = N2 // 32
        for xo, so in outs[1:]:
            assert torch.equal(xo, outs[0][0]) and torch.equal(so[:, :npo], outs[0][1][:, :npo])
        assert torch.isnan(outs[0][1][:, npo:]).all()               # partials past N / 32 are not touched
        xs = outs[0][0].float()
        assert (outs[0][1][:, :npo, 0] - xs.view(M, npo, 32).sum(2)).abs().max().item() <= 1e-3
        assert (outs[0][1][:, :npo, 1] - (xs * xs).view(M, npo, 32).sum(2)).abs().max().item() <= 1e-2 * (xs * xs).view(M, npo, 32).sum(2).abs().max().item()



@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("M", [256, 200, 64, 3])
def test_decode_mlp_one_launch_vs_the_two_launch_pair(M, plain):
    """mmtg_decode_mlp (round 6): c_fc (LN-fold + GELU) -> mlp.c_proj (+ bias + residual + statistics) as ONE launch whose hidden
    dimension is split over the XCDs, against (a) the explicit fp32 arithmetic on the same bf16 operands and (b) the mode-0 + mode-2
    mmtg_decode_gemm pair it replaces (same rounding points: bf16 hidden activations, bf16 output; only the order of the fp32
    partial sums differs).  Both hand-off modes; ragged row counts; repeated launches bit-equal (slice-ordered reduction), counters
    re-armed, no error reported, statistics partials = sums over the STORED bf16 rows (16 partials of 48 columns)."""
    D, HID, NP = 768, 3072, hip.DG_NP
    if plain:
        from mmtg_amd.decode import _placement_ok
        if not _placement_ok(torch.device(DEV, torch.cuda.current_device())):
            pytest.skip("workgroups b and b + 8 k do not share an XCD on this box: the L2 hand-off is not taken (the launch would report it)")
    g = torch.Generator().manual_seed(21 + M)
    x = (torch.randn(M, D, generator=g) * 2.0 + torch.randn(M, 1, generator=g)).to(torch.bfloat16)
    x[:, 5] *= 20.0
    W1 = (torch.randn(HID, D, generator=g) * 0.03).to(torch.bfloat16)
    W2 = (torch.randn(D, HID, generator=g) * 0.03).to(torch.bfloat16)
    gamma, beta = 1.0 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    b1, b2 = 0.1 * torch.randn(HID, generator=g), 0.1 * torch.randn(D, generator=g)
    xd, W2d, b2d = x.to(DEV), W2.to(DEV), b2.to(DEV)
    W1f = torch.empty(HID, D, dtype=torch.bfloat16, device=DEV)
    c1, b1f = torch.empty(HID, device=DEV), torch.empty(HID, device=DEV)
    hip.ln_fold_weights(W1.to(DEV), gamma.to(DEV), beta.to(DEV), b1.to(DEV), W1f, c1, b1f, HID, D)
    xf = x.float()
    st = torch.zeros(M, NP, 2)
    st[:, :D // 32, 0] = xf.view(M, D // 32, 32).sum(2)
    st[:, :D // 32, 1] = (xf * xf).view(M, D // 32, 32).sum(2)
    std = st.to(DEV)
    # (b) the two launches
    G0 = torch.empty(M, HID, dtype=torch.bfloat16, device=DEV)
    hip.decode_gemm(0, xd, W1f, G0, M, HID, D, bias=b1f, colsum=c1, stats_in=std, np_in=D // 32, act=hip.EPI_GELU)
    tiles = -(-M // 64) * (D // 64)
    ws0 = torch.empty(tiles * 4 * 4096, device=DEV)
    cnt0 = torch.zeros(tiles * 4, dtype=torch.int32, device=DEV)
    C0 = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    so0 = torch.zeros(M, NP, 2, device=DEV)
    hip.decode_gemm(2, G0, W2d, C0, M, D, HID, bias=b2d, resid=xd, stats_out=so0, splits=4, ws=ws0, counters=cnt0)
    # the fused launch
    ws = torch.full((hip.decode_mlp_ws_floats(M),), float("nan"), device=DEV)
    sync = torch.zeros(hip.decode_mlp_sync_words(), dtype=torch.int64, device=DEV)
    outs = []
    for rep in range(3):
        G = torch.full((M, HID), 3.0, dtype=torch.bfloat16, device=DEV)
        C1 = torch.full((M, D), 7.0, dtype=torch.bfloat16, device=DEV)
        so = torch.full((M, NP, 2), float("nan"), device=DEV)
        hip.decode_mlp(xd, std, D // 32, 1e-5, W1f, c1, b1f, W2d, b2d, G, C1, so, ws, sync, M, D, plain=plain)
        torch.cuda.synchronize()
        assert int(sync.abs().sum()) == 0, sync.tolist()             # counters re-armed, no error word
        outs.append((G, C1, so))
    G, C1, so = outs[0]
    for G_, C_, so_ in outs[1:]:
        assert torch.equal(G_, G) and torch.equal(C_, C1) and torch.equal(so_[:, :16], so[:, :16])
    # the hidden activations go through the same arithmetic in both forms
    assert (G.float() - G0.float()).abs().max().item() <= 2e-2 * G0.float().abs().max().item()
    assert (G == G0).float().mean().item() > 0.99
    # (a) fp32 arithmetic on the stored bf16 hidden activations
    ref = G.float().cpu() @ W2.float().t() + b2 + xf
    close(C1, ref, torch.bfloat16, HID, "fused MLP output")
    assert bool(((C1.float() - C0.float()).abs() <= 2.0 ** -6 * C0.float().abs() + 1e-2).all())          # within ~2 bf16 ulps of the pair
    mu, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    full = O.gelu_new(((xf - mu) / torch.sqrt(var + 1e-5) * gamma + beta) @ W1.float().t() + b1)
    close(G, full, torch.bfloat16, D, "fused MLP hidden activations")
    xs = C1.float()
    assert (so[:, :16, 0] - xs.view(M, 16, 48).sum(2)).abs().max().item() <= 1e-3 * max(1.0, xs.abs().max().item())
    sq = (xs * xs).view(M, 16, 48).sum(2)
    assert (so[:, :16, 1] - sq).abs().max().item() <= 1e-3 * sq.abs().max().item()
    assert torch.isnan(so[:, 16:]).all()                             # partials past 16 are not touched


def test_decode_mlp_census_reports_the_placement_the_plain_handoff_needs():
    """256 workgroups, one per CU: the census the plain hand-off relies on (workgroups b and b + 8 k on one XCD, 32 per XCD).  The
    test records what the box does; the decoder only takes the plain mode when it holds."""
    c = hip.decode_mlp_census(DEV)
    assert c.shape == (8, 8) and int(c.sum()) == 256
    from mmtg_amd.decode import _placement_ok
    ok = bool(((c == 32).sum(1) == 1).all() and ((c == 32).sum(0) == 1).all())
    assert _placement_ok(torch.device(DEV, torch.cuda.current_device())) == ok
    print("XCD census (rows: workgroup id % 8, columns: XCC_ID):", c.tolist())


# ------------------------------------------------------------------ generation
def test_logits_process_argmax():
    B, V, G = 6, 500, 40
    gen = torch.Generator().manual_seed(0)
    logits = torch.randn(B, 512, generator=gen) * 3
    generated = torch.randint(3, V, (B, G), generator=gen)
    generated[0, :10] = 7
    generated[1, -1] = 0           # sticky PAD
    generated[2, 5] = 102
    lens = torch.tensor([G, G, G, 1, 17, G], dtype=torch.int32)
    logits[0, 7] = 50.0            # heavily penalised: 50 / 1.5^10
    logits[4, 100] = 99.0          # banned id must not win
    nxt = torch.empty(B, dtype=torch.long, device=DEV)
    hip.logits_process_argmax(logits.to(DEV), 512, V, generated.to(DEV), G, lens.to(DEV), 1.1, 1.5, nxt, B)
    for b in range(B):
        gl = generated[b, :lens[b]]
        if gl[-1].item() == 0:
            want = 0
        else:
            want = int(torch.argmax(O.process_logits(logits[b, :V], gl, 1.1, 1.5)).item())
        assert nxt[b].item() == want, (b, nxt[b].item(), want)


@pytest.mark.parametrize("top_k,top_p", [(30, 0.0), (1, 0.0), (5, 0.0), (10, 0.7), (0, 0.9), (30, 0.3), (0, 0.0), (400, 0.0)])
def test_logits_process_sample(top_k, top_p):
    """Stochastic selection kernel against the oracle (whose filter is pinned by the reference's golden KATs):
    (a) the filtered processed logits equal the oracle's bit for bit (same kept set, same values: the radix-select
    thresholds reproduce `logits < kth` and the shifted cumulative-probability rule); (b) the id drawn for a uniform
    u is the inverse CDF of the filtered softmax in index order -- checked against a float64 CDF with a 1e-5 band
    around the interval ends; (c) sticky PAD, bans, penalty per occurrence."""
    B, V, Vpad, G = 9, 500, 512, 40
    gen = torch.Generator().manual_seed(top_k * 7 + int(top_p * 100))
    logits = torch.zeros(B, Vpad)
    logits[:, :V] = torch.randn(B, V, generator=gen) * 3
    logits[2, 10] = logits[2, 11]                    # a tie
    logits[2, 12] = logits[2, 10]
    generated = torch.randint(3, V, (B, G), generator=gen)
    generated[0, :10] = 7
    generated[1, -1] = 0                              # sticky PAD
    lens = torch.tensor([G, G, G, 1, 17, G, 5, G, 3], dtype=torch.int32)
    u = torch.tensor([0.0, 0.5, 0.123, 0.999, 0.37, 0.61, 0.05, 0.88, 0.999999], dtype=torch.float32)
    nxt = torch.full((B,), -7, dtype=torch.long, device=DEV)
    filt = torch.full((B, Vpad), float("nan"), device=DEV)
    hip.logits_process_sample(logits.to(DEV), Vpad, V, generated.to(DEV), G, lens.to(DEV), 1.1, 1.5, top_k, top_p,
                              u.to(DEV), nxt, B, filtered=filt)
    nxt, filt = nxt.cpu(), filt.cpu()
    for b in range(B):
        gl = generated[b, :lens[b]]
        if int(gl[-1]) == 0:
            assert int(nxt[b]) == 0
            continue
        proc = O.process_logits(logits[b, :V], gl, 1.1, 1.5)
        want = O.top_k_top_p_filtering(proc.clone(), top_k, top_p)
        assert torch.equal(filt[b, :V], want), (b, (filt[b, :V] != want).nonzero().flatten()[:8])
        p64 = torch.softmax(want.double(), -1)
        cdf = torch.cumsum(p64, -1)
        t = int(nxt[b])
        assert want[t] > -float("inf") and t not in (1, 2, 100, 102)
        lo = float(cdf[t] - p64[t])
        assert lo - 1e-5 <= float(u[b]) <= float(cdf[t]) + 1e-5, (b, t, lo, float(u[b]), float(cdf[t]))
    if top_k == 1:      # a single survivor: every u picks the arg-max, i.e. the greedy kernel's choice
        ref = torch.empty(B, dtype=torch.long, device=DEV)
        hip.logits_process_argmax(logits.to(DEV), Vpad, V, generated.to(DEV), G, lens.to(DEV), 1.1, 1.5, ref, B)
        assert torch.equal(ref.cpu(), nxt)


def test_logits_process_sample_frequencies():
    """20 000 draws of one row (top_k = 8): empirical frequencies match the filtered softmax (chi-square with 7
    degrees of freedom < 24.3, the 0.1 % quantile) and nothing outside the top 8 is ever drawn."""
    V, Vpad, N = 300, 304, 20000
    g = torch.Generator().manual_seed(3)
    row = torch.zeros(Vpad)
    row[:V] = torch.randn(V, generator=g) * 2
    logits = row.repeat(N, 1).to(DEV)
    generated = torch.full((N, 1), 1, dtype=torch.long, device=DEV)
    lens = torch.ones(N, dtype=torch.int32, device=DEV)
    u = torch.rand(N, generator=g).to(DEV)
    nxt = torch.empty(N, dtype=torch.long, device=DEV)
    hip.logits_process_sample(logits, Vpad, V, generated, 1, lens, 1.0, 1.0, 8, 0.0, u, nxt, N)
    want = O.top_k_top_p_filtering(O.process_logits(row[:V], torch.tensor([1]), 1.0, 1.0), 8, 0.0)
    p = torch.softmax(want.double(), -1)
    counts = torch.bincount(nxt.cpu(), minlength=V).double()
    assert float(counts[p == 0].sum()) == 0
    kept = p > 0
    chi2 = float((((counts[kept] - N * p[kept]) ** 2) / (N * p[kept])).sum())
    assert int(kept.sum()) == 8 and chi2 < 24.3, chi2


def test_abi_error_behaviour():
    """SURVEY §8(b) error contract: a bad call returns a negative code with a thread-local message (raised as
    RuntimeError by the binding), launches nothing, and the library stays usable afterwards."""
    x = torch.zeros(8, 2048, device=DEV, dtype=torch.bfloat16)
    f = torch.zeros(2048, device=DEV)
    with pytest.raises(RuntimeError, match="1024"):                       # LayerNorm rows wider than the kernels serve
        hip.layernorm_fwd(x, x.clone(), f, f, torch.zeros(8, device=DEV), torch.zeros(8, device=DEV), 8, 2048)
    with pytest.raises(RuntimeError, match="head dim"):                   # attention is built for dh = 64
        hip.attn_fwd(torch.zeros(1, 4, 3 * 32, device=DEV, dtype=torch.bfloat16), torch.ones(1, 4, dtype=torch.int32, device=DEV),
                     torch.zeros(1, 4, 32, device=DEV, dtype=torch.bfloat16), torch.zeros(1, 1, 4, device=DEV), 1, 4, 1, 32)
    big = torch.zeros(1, 30000, device=DEV)
    with pytest.raises(RuntimeError, match="exceeds"):                    # sampler's LDS row image
        hip.logits_process_sample(big, 30000, 30000, torch.ones(1, 1, dtype=torch.long, device=DEV), 1,
                                  torch.ones(1, dtype=torch.int32, device=DEV), 1.0, 1.0, 5, 0.0,
                                  torch.zeros(1, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), 1)
    with pytest.raises(RuntimeError, match="temperature"):
        hip.logits_process_argmax(torch.zeros(1, 64, device=DEV), 64, 64, torch.ones(1, 1, dtype=torch.long, device=DEV), 1,
                                  torch.ones(1, dtype=torch.int32, device=DEV), 0.0, 1.0, torch.zeros(1, dtype=torch.long, device=DEV), 1)
    part = torch.zeros(2, 16, 16, device=DEV)
    with pytest.raises(RuntimeError, match="stride"):                     # slab stride smaller than the slab
        hip.slab_sum(part, 2, 8, torch.zeros(16, 16, device=DEV), 256)
    with pytest.raises(RuntimeError, match="split"):                      # split-K without a split-capable epilogue
        hip.gemm(x, x, torch.zeros(8, 8, device=DEV, dtype=torch.bfloat16), 8, 8, 2048, transB=True, splits=4)
    # still alive: a valid call after the failures
    a = rnd(64, 64, dtype=torch.bfloat16, seed=1).to(DEV)
    c = torch.empty(64, 64, device=DEV, dtype=torch.bfloat16)
    hip.gemm(a, a, c, 64, 64, 64, transB=True)
    close(c, a.float() @ a.float().t(), torch.bfloat16, 64, "gemm after errors")


# ------------------------------------------------------------------ gathered-operand products (multi-modal conditioning, fused)
def test_gemm_gather_forward_and_weight_gradient():
    """mmtg_gemm_gather against the same products on an explicitly gathered operand: the forward (table rows as the A
    operand's rows, K-contiguous) and the weight gradient (table rows as the reduction index of the K-strided B operand)
    are bit-identical to mmtg_gemm on table[rows] -- only the source addresses of the LDS-DMA differ -- and the
    TANH_ADD epilogue adds aux[aux_rows[m]] before the tanh (model.py:262-281: (E[id] + c) W1^T = E[id] W1^T + (c W1^T)[seg])."""
    dtype = torch.bfloat16
    Vt, E, H, M = 300, 2048, 512, 700          # table rows, embedding width, projector width, tokens (ragged: 700 = 5 x 128 + 60)
    table = rnd(Vt, E, dtype=dtype, seed=1, scale=0.05).to(DEV)
    w = rnd(H, E, dtype=dtype, seed=2, scale=0.03).to(DEV)
    bias = rnd(H, seed=3).to(DEV)
    g = torch.Generator().manual_seed(4)
    rows = torch.randint(0, Vt, (M,), generator=g, dtype=torch.int32).to(DEV)
    xg = table[rows.long()].contiguous()
    # forward, plain
    ref = torch.empty(M, H, device=DEV, dtype=dtype)
    hip.gemm(xg, w, ref, M, H, E, transB=True, ldb=E, bias=bias, flags=hip.GEMM_NO_WIDE | hip.GEMM_NO_OCC4)
    got = torch.full((M, H), float("nan"), device=DEV, dtype=dtype)
    hip.gemm_gather(0, table, w, got, M, H, E, rows, Vt, lda=E, ldb=E, bias=bias)
    assert torch.equal(got, ref)
    # forward, tanh(acc + bias + aux[aux_rows])
    nseg = 37
    aux = rnd(nseg + 1, H, dtype=dtype, seed=5).to(DEV)
    aux[nseg] = 0
    amap = torch.randint(0, nseg + 1, (M,), generator=g, dtype=torch.int32).to(DEV)
    got2 = torch.empty(M, H, device=DEV, dtype=dtype)
    hip.gemm_gather(0, table, w, got2, M, H, E, rows, Vt, lda=E, ldb=E, bias=bias, epi=hip.EPI_TANH_ADD, aux=aux, ldaux=H, aux_rows=amap)
    want = torch.tanh(xg.float() @ w.float().t() + bias + aux[amap.long()].float())
    close(got2, want.cpu(), dtype, E, "gather forward tanh_add")
    # weight gradient: slabs[s] = dA[k, :]^T . table[rows[k], :]
    Kt, splits = 1000, 3                       # ragged token count
    dA = rnd(Kt, H, dtype=dtype, seed=6).to(DEV)
    rk = torch.randint(0, Vt, (Kt,), generator=g, dtype=torch.int32).to(DEV)
    xk = table[rk.long()].contiguous()
    ref_s = torch.empty(splits, H, E, device=DEV, dtype=torch.float32)
    hip.gemm(dA, xk, ref_s, H, E, Kt, transA=True, transB=False, lda=H, ldb=E, ldc=E, epi=hip.EPI_SPLIT, out_f32=True, splits=splits)
    got_s = torch.full((splits, H, E), float("nan"), device=DEV, dtype=torch.float32)
    hip.gemm_gather(1, dA, table, got_s, H, E, Kt, rk, Vt, lda=H, ldb=E, ldc=E, epi=hip.EPI_SPLIT, splits=splits)
    assert torch.equal(got_s, ref_s)
    close(got_s.sum(0), dA.float().t() @ xk.float(), dtype, Kt, "gather weight gradient")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["LSTM", "RNN"])
def test_rnn_cells_fwd_bwd_vs_torch(kind, dtype):
    """mmtg_rnn_cell_fwd / _bwd (the LSTM and ReLU-RNN encoder cells of model.py:41-59) over a short sequence in the engine's
    row layout (rows b*S+t) against torch autograd on the same fp32 pre-activations."""
    B, S, H = 6, 3, 64
    G = 4 if kind == "LSTM" else 1
    code = hip.RNN_LSTM if kind == "LSTM" else hip.RNN_RELU
    gi = rnd(B * S, G * H, dtype=dtype, seed=61).to(DEV)
    gh = rnd(S, B, G * H, dtype=dtype, seed=62).to(DEV)
    dh = rnd(B * S, H, dtype=dtype, seed=63).to(DEV)
    part = rnd(S, B, H, seed=64).to(DEV)              # stands for d(a_{t+1}) W_hh
    h = torch.zeros(B * S, H, device=DEV, dtype=dtype)
    c = torch.zeros(S, B, H, device=DEV)
    save = torch.zeros(S, 5, B, H, device=DEV)
    for t in range(S):
        hip.rnn_cell_fwd(code, gi[t:], gh[t], c[t - 1] if t else None, h[t:], c[t], save[t], B, H, ld_gi=S * G * H, ld_h=S * H)
    # torch reference on the same pre-activations (a = gi + gh per step)
    a = (gi.float().view(B, S, G * H).transpose(0, 1) + gh.float()).detach().requires_grad_(True)      # [S, B, G H]
    hs, cp = [], torch.zeros(B, H, device=DEV)
    for t in range(S):
        if kind == "LSTM":
            ai, af, ag, ao = a[t].chunk(4, -1)
            cp = torch.sigmoid(af) * cp + torch.sigmoid(ai) * torch.tanh(ag)
            hs.append(torch.sigmoid(ao) * torch.tanh(cp))
        else:
            hs.append(torch.relu(a[t]))
    href = torch.stack(hs, 1).reshape(B * S, H)
    close(h, href, dtype, 4, kind + " forward")
    # backward: every step's dh = rows + part (no recurrence through h here: part plays its role), the cell state carries
    da = torch.zeros(B * S, G * H, device=DEV, dtype=dtype)
    dc = torch.zeros(B, H, device=DEV)
    for t in range(S - 1, -1, -1):
        last = t == S - 1
        hip.rnn_cell_bwd(code, dh[t:], S * H, None if last else part[t], save[t] if kind == "LSTM" else None,
                         c[t - 1] if kind == "LSTM" and t else None, h[t:], S * H, dc if kind == "LSTM" else None, not last,
                         da[t:], S * G * H, B, H)
    up = dh.float().view(B, S, H).clone()
    up[:, :S - 1] += part[:S - 1].transpose(0, 1)
    (href.view(B, S, H) * up).sum().backward()
    close(da.view(B, S, G * H).transpose(0, 1), a.grad, dtype, 8, kind + " backward")
