#!/usr/bin/env python3
"""A few launches of the kernels that carry the bf16 training step, for `rocprofv3 --pmc` passes (tools/gpu_pmc_util_r6.sh): the
products of a GPT-2 block at M = 15104 as the engine launches them (c_attn, c_fc + GELU, attn.c_proj + residual + dropout, mlp.c_proj
+ residual, the dGELU product with column sums, the c_attn / c_fc input gradients), one grouped weight-gradient launch (four products,
in-kernel reduction), the whole-head attention forward / backward (B = 64, T = 236, dropout), LayerNorm forward / backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

B, T, nH = 64, 236, 12
M, D = B * T, 768
dev = "cuda"
t = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
e = lambda *s: torch.empty(*s, device=dev, dtype=torch.bfloat16)
x, wq, w1, wp = t(M, D), t(3 * D, D), t(4 * D, D), t(D, D)
h, w2, w2n, w1n = t(M, 4 * D), t(D, 4 * D), t(4 * D, D), t(D, 4 * D)
dy, res, dqkv = t(M, D), t(M, D), t(M, 3 * D)
cq, c1, pre, c2, du, da = e(M, 3 * D), e(M, 4 * D), e(M, 4 * D), e(M, D), e(M, 4 * D), e(M, D)
b1, b2, bq = torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)
bands = torch.zeros((M + 63) // 64, 4 * D, device=dev)
shapes = ((D, 4 * D), (4 * D, D), (D, D), (D, 3 * D))
tiles, nws, ncnt = hip.wgrad_group_sizes(shapes, 2, 0)
ws, cnt = torch.empty(nws, device=dev), torch.zeros(ncnt, device=dev, dtype=torch.int32)
ops = [(x, t(M, 4 * D)), (h, dy), (x, dy), (x, dqkv)]
outs = [torch.zeros(a, b, device=dev) for a, b in shapes]
probs = [(A, Bm, C, a, b) for (A, Bm), C, (a, b) in zip(ops, outs, shapes)]
qkv = t(M, 3 * D)
keep = torch.ones(B, T, dtype=torch.int32, device=dev)
out, dout = e(M, D), t(M, D)
lse, delta = torch.empty(B, nH, T, device=dev), torch.empty(M, nH, device=dev)
dq32 = torch.empty(M, D, device=dev)
brows = torch.empty(hip.attn_bwd_bias_rows(B, T, hip.BF16), 3 * D, device=dev)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
mu, rs = torch.zeros(M, device=dev), torch.ones(M, device=dev)
ln, dx, dxm = e(M, D), e(M, D), e(M, D)
dg, db, dcs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
lnws = torch.empty(hip.lib().mmtg_layernorm_bwd_ws(M, D), device=dev)
for _ in range(3):
    hip.gemm(x, wq, cq, M, 3 * D, D, transB=True, bias=bq)
    hip.gemm(x, w1, c1, M, 4 * D, D, transB=True, bias=b1, epi=hip.EPI_GELU, aux2=pre)
    hip.gemm(x, wp, c2, M, D, D, transB=True, bias=b2, epi=hip.EPI_RESID, aux=res, ldaux=D, drop_p=0.1, drop_seed=3)
    hip.gemm(h, w2, c2, M, D, 4 * D, transB=True, bias=b2, epi=hip.EPI_RESID, aux=res, ldaux=D, drop_p=0.1, drop_seed=4)
    hip.gemm(dy, w2n, du, M, 4 * D, D, transB=True, epi=hip.EPI_DGELU, aux=pre, ldaux=4 * D, aux2=bands)      # dy W2[in,out]^T: the dGELU product
    hip.gemm(du, w1n, da, M, D, 4 * D, transB=True)                                                            # the c_fc input gradient
    hip.wgrad_group(probs, M, 2, ws, cnt)
    hip.attn_fwd(qkv, keep, out, lse, B, T, nH, 64, drop_p=0.1, drop_seed=1)
    hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, 64, drop_p=0.1, drop_seed=1, dbias=torch.zeros(3 * D, device=dev), dbias_ws=brows)
    hip.layernorm_fwd(x, ln, gam, bet, mu, rs, M, D)
    hip.layernorm_bwd(dy, x, gam, mu, rs, res, dx, dg, db, M, D, dx_masked=dxm, drop_p=0.1, drop_seed=2, dcolsum=dcs, ws=lnws)
torch.cuda.synchronize()
print("ok")
