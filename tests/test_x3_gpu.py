"""Split-precision ("bf16x3") kernels on the MI355X against float64 restatements (round 5).

The x3 mode keeps fp32 storage and computes the GPT-2 / lm_head products (reference: the Conv1D / Linear layers behind
/root/reference/src/model.py:282-288, fp32 arithmetic) as  X_hi W_hi + X_lo W_hi + X_hi W_lo  over (hi | lo) bf16 plane pairs.
Tolerance of a product: 4e-5 of sqrt(K) * rms(A) * rms(B) (measured worst 2.2e-5 over 1e5-1e6 outputs) -- two orders below the bf16 mode's, one above exact fp32's.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmtg_amd import hip  # noqa: E402

DEV = "cuda"


def planes_of(x):
    rows, cols = x.shape
    return hip.split_planes(x.contiguous(), rows, cols, hip.Planes.empty(rows, cols, x.device))


def test_split_planes_reconstructs_to_2_pow_minus_17():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(333, 1024, device=DEV, generator=g) * torch.logspace(-6, 4, 1024, device=DEV)
    p = planes_of(x)
    hi = p.t[0].float()
    assert torch.equal(hi, x.bfloat16().float())                      # hi = bf16(x), round to nearest even
    rel = ((p.float() - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -17, rel
    # a strided source / destination
    big = torch.randn(64, 512, device=DEV, generator=g)
    out = torch.zeros(2, 64, 256, device=DEV, dtype=torch.bfloat16)
    hip.split_planes(big[:, 128:], 64, 128, hip.Planes(out[0, :, 64:], 64, 128, ld=256, plane=64 * 256), lds=512)
    got = out[0, :, 64:192].float() + out[1, :, 64:192].float()
    assert ((got - big[:, 128:256]).abs() <= 2.0 ** -17 * big[:, 128:256].abs()).all()
    assert out[:, :, :64].abs().sum().item() == 0 and out[:, :, 192:].abs().sum().item() == 0


def _ref_epi(acc, epi, bias, aux, drop=None):
    v = acc if bias is None else acc + bias.double()
    k = 0.7978845608028654
    if epi == hip.EPI_GELU:
        return 0.5 * v * (1 + torch.tanh(k * (v + 0.044715 * v ** 3))), v
    if epi == hip.EPI_TANH:
        return torch.tanh(v), None
    if epi == hip.EPI_RESID:
        return v + aux.double(), None
    if epi == hip.EPI_DGELU:
        a = aux.double()
        t = torch.tanh(k * (a + 0.044715 * a ** 3))
        return v * (0.5 * (1 + t) + 0.5 * a * (1 - t * t) * k * (1 + 3 * 0.044715 * a * a)), None
    if epi == hip.EPI_DTANH:
        return v * (1 - aux.double() ** 2), None
    return v, None


@pytest.mark.parametrize("M,N,K", [(300, 520, 256), (1000, 768, 768), (77, 3072, 128), (513, 264, 1152)])
@pytest.mark.parametrize("epi", [hip.EPI_NONE, hip.EPI_GELU, hip.EPI_TANH, hip.EPI_RESID, hip.EPI_DGELU, hip.EPI_DTANH])
def test_gemm_x3_vs_float64(M, N, K, epi):
    g = torch.Generator(device=DEV).manual_seed(M + N + K + epi)
    A = torch.randn(M, K, device=DEV, generator=g)
    Bm = torch.randn(N, K, device=DEV, generator=g) * 0.05
    bias = torch.randn(N, device=DEV, generator=g) if epi not in (hip.EPI_DGELU, hip.EPI_DTANH) else None
    aux = torch.randn(M, N, device=DEV, generator=g) * (0.9 if epi == hip.EPI_DTANH else 1.0) if epi in (hip.EPI_RESID, hip.EPI_DGELU, hip.EPI_DTANH) else None
    C = torch.full((M, N), float("nan"), device=DEV)
    pl = hip.Planes.empty(M, N, DEV)
    aux2 = torch.full((M, N), float("nan"), device=DEV) if epi == hip.EPI_GELU else None
    bands = torch.zeros((M + 63) // 64, N, device=DEV) if epi == hip.EPI_DGELU else None
    hip.gemm_x3(planes_of(A), planes_of(Bm), C, M, N, K, planes=pl, bias=bias, epi=epi, aux=aux,
                aux2=aux2 if epi == hip.EPI_GELU else bands)
    acc = A.double() @ Bm.double().t()
    ref, pre = _ref_epi(acc, epi, bias, aux)
    scale = float(np.sqrt(K)) * A.pow(2).mean().sqrt().item() * Bm.pow(2).mean().sqrt().item()
    err = (C.double() - ref).abs().max().item()
    amp = 1.0 + (aux.abs().max().item() if epi == hip.EPI_DGELU else aux.pow(2).max().item() if epi == hip.EPI_DTANH else 0.0)   # |d epi / d acc|
    assert err < 4e-5 * scale * amp, (err, scale, amp)
    # the plane-pair output is the fp32 result split
    assert ((pl.float() - C).abs() <= 2.0 ** -17 * C.abs() + 1e-30).all()
    if epi == hip.EPI_GELU:
        assert (aux2.double() - pre).abs().max().item() < 4e-5 * scale
    if epi == hip.EPI_DGELU:        # column sums of the output per 64-row band
        want = torch.stack([C[i:i + 64].sum(0) for i in range(0, M, 64)])
        assert (bands - want).abs().max().item() < 1e-3 * (1 + want.abs().max().item())
    # fp32-only and planes-only outputs agree with the combined call bit for bit
    C2 = torch.empty_like(C)
    hip.gemm_x3(planes_of(A), planes_of(Bm), C2, M, N, K, bias=bias, epi=epi, aux=aux,
                aux2=torch.empty_like(C) if epi == hip.EPI_GELU else None)
    assert torch.equal(C2, C)
    if epi != hip.EPI_GELU:
        pl2 = hip.Planes.empty(M, N, DEV)
        hip.gemm_x3(planes_of(A), planes_of(Bm), None, M, N, K, planes=pl2, bias=bias, epi=epi, aux=aux)
        assert torch.equal(pl2.t, pl.t)


def test_gemm_x3_is_far_closer_than_one_bf16_pass_and_deterministic():
    g = torch.Generator(device=DEV).manual_seed(3)
    M, N, K = 2048, 2304, 768
    A = torch.randn(M, K, device=DEV, generator=g)
    Bm = torch.randn(N, K, device=DEV, generator=g) * 0.02
    ref = A.double() @ Bm.double().t()
    C = torch.empty(M, N, device=DEV)
    hip.gemm_x3(planes_of(A), planes_of(Bm), C, M, N, K)
    C1 = torch.empty(M, N, device=DEV)
    hip.gemm(A.bfloat16(), Bm.bfloat16(), C1, M, N, K, transB=True, ldb=K, out_f32=True)
    e3 = (C.double() - ref).abs().max().item()
    e1 = (C1.double() - ref).abs().max().item()
    assert e3 < e1 / 100, (e3, e1)
    Cb = torch.empty_like(C)
    hip.gemm_x3(planes_of(A), planes_of(Bm), Cb, M, N, K)
    assert torch.equal(C, Cb)


def test_gemm_x3_residual_dropout_matches_the_fp32_kernels_mask():
    """The counter-hash dropout of the RESID epilogue is the same stream in every GEMM kernel: same seed, same mask."""
    g = torch.Generator(device=DEV).manual_seed(4)
    M, N, K = 384, 768, 256
    A = torch.randn(M, K, device=DEV, generator=g)
    Bm = torch.randn(N, K, device=DEV, generator=g) * 0.05
    aux = torch.randn(M, N, device=DEV, generator=g)
    C = torch.empty(M, N, device=DEV)
    hip.gemm_x3(planes_of(A), planes_of(Bm), C, M, N, K, epi=hip.EPI_RESID, aux=aux, drop_p=0.25, drop_seed=1234)
    C32 = torch.empty(M, N, device=DEV)
    hip.gemm(A, Bm, C32, M, N, K, transB=True, ldb=K, epi=hip.EPI_RESID, aux=aux, ldaux=N, drop_p=0.25, drop_seed=1234)
    assert (C - C32).abs().max().item() < 4e-5 * np.sqrt(K) * 0.05 * 4
    dropped = ((C - aux).abs() < 1e-12).float().mean().item()
    assert 0.2 < dropped < 0.3


def test_layernorm_fwd_x3_vs_torch():
    g = torch.Generator(device=DEV).manual_seed(5)
    for rows, cols in ((1000, 768), (37, 512), (64, 1024)):
        x = torch.randn(rows, cols, device=DEV, generator=g) * 3 + 0.5
        gm = torch.randn(cols, device=DEV, generator=g)
        bt = torch.randn(cols, device=DEV, generator=g)
        out = hip.Planes.empty(rows, cols, DEV)
        mu = torch.empty(rows, device=DEV)
        rs = torch.empty(rows, device=DEV)
        hip.layernorm_fwd_x3(x, out, gm, bt, mu, rs, rows, cols, 1e-5)
        ref = torch.nn.functional.layer_norm(x.double(), (cols,), gm.double(), bt.double(), 1e-5)
        assert (out.float().double() - ref).abs().max().item() < 2e-5 * (1 + ref.abs().max().item())
        assert (mu.double() - x.double().mean(1)).abs().max().item() < 1e-5


@pytest.mark.parametrize("config", [2, 6])        # 2: three passes over (A | B) stages; 6: one loop over combined four-plane stages
@pytest.mark.parametrize("K,splits", [(1024, 1), (3000, 2), (4096, 4)])
def test_wgrad_group_x3_vs_float64(K, splits, config):
    g = torch.Generator(device=DEV).manual_seed(K)
    shapes = ((768, 3072), (256, 768), (136, 264))
    probs, refs = [], []
    for (Mi, Ni) in shapes:
        X = torch.randn(K, Mi, device=DEV, generator=g)
        dY = torch.randn(K, Ni, device=DEV, generator=g) * 0.01
        Cg = torch.full((Mi, Ni), 7.0, device=DEV)
        probs.append((planes_of(X), planes_of(dY), Cg, Mi, Ni))
        refs.append((X.double().t() @ dY.double(), float(np.sqrt(K)) * 0.01))
    tiles, nws, ncnt = hip.wgrad_group_sizes(shapes, splits, 0)
    ws = torch.empty(nws, device=DEV) if splits > 1 else None
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=DEV)
    hip.wgrad_group(probs, K, splits, ws, cnt, accumulate=False, config=config)
    for (pr, (ref, scale)) in zip(probs, refs):
        assert (pr[2].double() - ref).abs().max().item() < 4e-5 * scale
    assert int(cnt.abs().sum().item()) == 0
    first = [pr[2].clone() for pr in probs]
    hip.wgrad_group(probs, K, splits, ws, cnt, accumulate=True, config=config)     # accumulate: exactly twice the first result
    for pr, f in zip(probs, first):
        assert torch.equal(pr[2], f + f)


def test_adamw_writes_the_weight_plane_pair():
    g = torch.Generator(device=DEV).manual_seed(9)
    n = 4096 + 8
    p = torch.randn(n, device=DEV, generator=g)
    gr = torch.randn(n, device=DEV, generator=g) * 1e-2
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    pl = torch.zeros(2, n, device=DEV, dtype=torch.bfloat16)
    hip.adamw(p, gr, m, v, pl[0], n, 1e-3, 0.9, 0.999, 1e-6, 0.0, 1, None, 1.0, p_lo=pl[1])
    assert torch.equal(pl[0].float(), p.bfloat16().float())
    assert ((pl[0].float() + pl[1].float() - p).abs() <= 2.0 ** -17 * p.abs()).all()


@pytest.mark.parametrize("B,T,drop", [(3, 236, 0.0), (2, 100, 0.0), (2, 236, 0.1), (1, 300, 0.1),
                                      # ragged edges of the 32-query / 64-query / 128-key tiles, one-token and maximum-length sequences
                                      (2, 1, 0.0), (1, 31, 0.1), (2, 33, 0.0), (1, 64, 0.0), (1, 65, 0.1), (1, 128, 0.0), (2, 129, 0.1), (1, 512, 0.0)])
def test_attention_x3_matches_the_exact_fp32_kernels(B, T, drop):
    """The split-precision attention (three bf16 passes per product, operands as (hi | lo) plane pairs) against the exact-fp32 kernels it
    replaces in the bf16x3 mode: same masks (causal + key padding), same dropout stream -- forward context / LSE and the backward's
    d(qkv) plane pair + c_attn bias gradient."""
    nH, dh = 12, 64
    D = nH * dh
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + T)
    qkv = torch.randn(B * T, 3 * D, device=DEV, generator=g) * 0.8
    keep = torch.ones(B, T, dtype=torch.int32, device=DEV)
    if T > 16:
        keep[0, T - 7:] = 0                               # padded tail keys on row 0
        keep[B - 1, 5] = 0
    dout = torch.randn(B * T, D, device=DEV, generator=g) * 0.1
    seed = 4242
    # exact fp32 reference kernels
    out32 = torch.empty(B * T, D, device=DEV)
    lse32 = torch.empty(B, nH, T, device=DEV)
    hip.attn_fwd(qkv, keep, out32, lse32, B, T, nH, dh, drop_p=drop, drop_seed=seed)
    delta = torch.empty(B * T, nH, device=DEV)
    dq32 = torch.empty(B * T, D, device=DEV)
    dqkv32 = torch.zeros(B * T, 3 * D, device=DEV)
    db32 = torch.zeros(3 * D, device=DEV)
    ws = torch.empty(hip.attn_bwd_bias_rows(B, T, hip.F32), 3 * D, device=DEV)
    hip.attn_bwd(qkv, keep, out32, dout, lse32, delta, dq32, dqkv32, B, T, nH, dh, drop_p=drop, drop_seed=seed, dbias=db32, dbias_ws=ws)
    # split-precision kernels
    out = torch.full((B * T, D), float("nan"), device=DEV)
    outp = hip.Planes.empty(B * T, D, DEV)
    lse = torch.empty(B, nH, T, device=DEV)
    qkvp = hip.split_planes(qkv, B * T, 3 * D, hip.Planes.empty(B * T, 3 * D, DEV))     # what the c_attn product hands over
    hip.attn_fwd_x3(qkvp, keep, out, outp, lse, B, T, nH, dh, drop_p=drop, drop_seed=seed)
    assert (out - out32).abs().max().item() < 2e-5 * max(1.0, out32.abs().max().item())
    assert (lse - lse32).abs().max().item() < 2e-5 * max(1.0, lse32.abs().max().item())
    assert ((outp.float() - out).abs() <= 2.0 ** -17 * out.abs() + 1e-30).all()
    dqp = hip.Planes.empty(B * T, 3 * D, DEV)
    db = torch.zeros(3 * D, device=DEV)
    ws3 = torch.empty(hip.attn_bwd_x3_ws(B, T, D), device=DEV)
    doutp = hip.split_planes(dout, B * T, D, hip.Planes.empty(B * T, D, DEV))           # ... and the c_proj dgrad
    dqs = torch.full((hip.attn_bwd_x3_dq_floats(B, T, D),), float("nan"), device=DEV)      # (never zeroed: every element read is written first)
    hip.attn_bwd_x3(qkvp, keep, out, doutp, lse, torch.empty_like(delta), dqs, dqp, B, T, nH, dh, drop_p=drop, drop_seed=seed, dbias=db, dbias_ws=ws3)
    scale = dqkv32.abs().max().item()
    assert (dqp.float() - dqkv32).abs().max().item() < 4e-5 * scale, ((dqp.float() - dqkv32).abs().max().item(), scale)
    assert (db - db32).abs().max().item() < 1e-4 * max(1.0, db32.abs().max().item())
    # no atomics anywhere in the split-precision backward: a second run gives the same bits
    dqp2, db2 = hip.Planes.empty(B * T, 3 * D, DEV), torch.zeros(3 * D, device=DEV)
    hip.attn_bwd_x3(qkvp, keep, out, doutp, lse, torch.empty_like(delta), dqs, dqp2, B, T, nH, dh, drop_p=drop, drop_seed=seed, dbias=db2, dbias_ws=ws3)
    assert torch.equal(dqp2.t, dqp.t) and torch.equal(db2, db)
    # round 6: the partial bias rows left for the batched column sums (dbias = None + a workspace) give the same bits
    dqp3, db3 = hip.Planes.empty(B * T, 3 * D, DEV), torch.zeros(3 * D, device=DEV)
    ws4 = torch.full((hip.attn_bwd_x3_ws(B, T, D),), float("nan"), device=DEV)
    hip.attn_bwd_x3(qkvp, keep, out, doutp, lse, torch.empty_like(delta), dqs, dqp3, B, T, nH, dh, drop_p=drop, drop_seed=seed, dbias=None, dbias_ws=ws4)
    nkv, nq = B * (-(-T // 128)), -(-(B * T) // 16)
    # (the items of a batch run concurrently and must not share output columns: k / v parts from the key-block rows -- whose q columns
    #  are zero -- and the q part from the band rows)
    assert float(ws4[:nkv * 3 * D].view(nkv, 3 * D)[:, :D].abs().max()) == 0.0
    hip.colsum_batch([(ws4.data_ptr() + 4 * D, db3.data_ptr() + 4 * D, 3 * D, nkv, 2 * D), (ws4.data_ptr() + 4 * nkv * 3 * D, db3.data_ptr(), D, nq, D)])
    assert torch.equal(dqp3.t, dqp.t) and torch.equal(db3, db)


def test_layernorm_bwd_x3_partial_rows_plus_the_batched_sum_equal_the_one_call_form():
    """mmtg_layernorm_bwd_x3_partial (first stage only) + mmtg_colsum_batch over its partial rows = mmtg_layernorm_bwd_x3, bit for bit:
    d(x), the masked plane pair, d gamma, d beta and the fused column sum."""
    rows, cols = 3001, 768
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(rows, cols, device=DEV, generator=g) * 2
    dy, dres = torch.randn(rows, cols, device=DEV, generator=g), torch.randn(rows, cols, device=DEV, generator=g)
    gam = 1 + 0.1 * torch.randn(cols, device=DEV, generator=g)
    mean, var = x.mean(-1), x.var(-1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    res = []
    for partial in (False, True):
        dx, pl = torch.empty_like(x), hip.Planes.empty(rows, cols, DEV)
        dg, db, dcs = (torch.full((cols,), v, device=DEV) for v in (1.0, -1.0, 2.0))
        ws = torch.full((int(hip.lib().mmtg_layernorm_bwd_ws(rows, cols)),), float("nan"), device=DEV)
        if partial:
            nb = hip.layernorm_bwd_x3_partial(dy, x, gam, mean, rstd, dres, dx, rows, cols, pl, ws, drop_p=0.1, drop_seed=77, want_colsum=True)
            assert 0 < nb <= 1024
            hip.colsum_batch([(ws.data_ptr() + 4 * q * cols, t.data_ptr(), 3 * cols, nb, cols) for q, t in enumerate((dg, db, dcs))])
        else:
            hip.layernorm_bwd_x3(dy, x, gam, mean, rstd, dres, dx, dg, db, rows, cols, pl, drop_p=0.1, drop_seed=77, dcolsum=dcs, ws=ws)
        res.append((dx, pl.t.clone(), dg, db, dcs))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    assert float((res[0][4] - 2.0).abs().max()) > 0


@pytest.mark.parametrize("B,T", [(2, 129), (1, 236), (1, 400)])
def test_whole_head_bf16_backward_regenerates_the_element_dropout_mask(B, T):
    """MMTG_ATTN_ELEM_MASK (round 6, the bf16x3f mode's backward): the bf16 whole-head attention backward kernels, given the fp32 /
    split-precision kernels' forward (context, LSE, dropout seed), differentiate THAT forward -- their d(qkv) agrees with the exact-fp32
    backward to bf16 accuracy -- whereas with their own 12-bit word masks (no flag) the same call differentiates a different dropout
    realisation and is far off.  p = 0.3 so that the difference is unmistakable."""
    nH, dh, drop, seed = 12, 64, 0.3, 991
    D = nH * dh
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * D, device=DEV, generator=g) * 0.8).bfloat16()
    keep = torch.ones(B, T, dtype=torch.int32, device=DEV)
    keep[0, T - 7:] = 0
    dout = (torch.randn(B * T, D, device=DEV, generator=g) * 0.1).bfloat16()
    q32, d32 = qkv.float(), dout.float()
    out32, lse32 = torch.empty(B * T, D, device=DEV), torch.empty(B, nH, T, device=DEV)
    hip.attn_fwd(q32, keep, out32, lse32, B, T, nH, dh, drop_p=drop, drop_seed=seed)
    delta, dq32 = torch.empty(B * T, nH, device=DEV), torch.empty(B * T, D, device=DEV)
    ref = torch.zeros(B * T, 3 * D, device=DEV)
    hip.attn_bwd(q32, keep, out32, d32, lse32, delta, dq32, ref, B, T, nH, dh, drop_p=drop, drop_seed=seed)
    outb = out32.bfloat16()
    got = {}
    for flags in (hip.ATTN_ELEM_MASK, 0):
        dqkv = torch.zeros(B * T, 3 * D, device=DEV, dtype=torch.bfloat16)
        hip.attn_bwd(qkv, keep, outb, dout, lse32, torch.empty_like(delta), torch.empty_like(dq32), dqkv, B, T, nH, dh, drop_p=drop, drop_seed=seed,
                     flags=flags)
        got[flags] = float((dqkv.float() - ref).norm() / ref.norm())
    assert got[hip.ATTN_ELEM_MASK] < 0.03, got
    assert got[0] > 0.25, got          # (guards against a vacuous pass: the kernels' own masks are a different realisation)
