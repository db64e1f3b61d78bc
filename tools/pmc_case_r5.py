#!/usr/bin/env python3
"""A few launches of the kernels that carry a bf16x3 (split-precision) training step, for `rocprofv3 --pmc` passes
(tools/gpu_pmc_util_r5.sh): the eight-phase X3 products of a GPT-2 block (qkv, fc1 + GELU, attention c_proj + residual, fc2 +
residual, dGELU + column sums), one grouped X3 weight-gradient launch, the split-precision attention forward / backward at
B = 64, T = 236 with dropout, the plane-writing LayerNorm forward / backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

B, T, nH = 64, 236, 12
M, D = B * T, 768
dev = "cuda"
f = lambda *s: torch.randn(*s, device=dev) * 0.5
P = lambda x: hip.split_planes(x, x.shape[0], x.shape[1], hip.Planes.empty(x.shape[0], x.shape[1], dev))
x, wq, w1, wp = P(f(M, D)), P(f(3 * D, D)), P(f(4 * D, D)), P(f(D, D))
h, w2, w2t = P(f(M, 4 * D)), P(f(D, 4 * D)), P(f(4 * D, D))
dy, res = P(f(M, D)), f(M, D)
cq, pre, c2 = torch.empty(M, 3 * D, device=dev), torch.empty(M, 4 * D, device=dev), torch.empty(M, D, device=dev)
gp, dup = hip.Planes.empty(M, 4 * D, dev), hip.Planes.empty(M, 4 * D, dev)
b1, b2, bq = torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)
bands = torch.zeros((M + 63) // 64, 4 * D, device=dev)
shapes = ((D, 4 * D), (4 * D, D), (D, D), (D, 3 * D))
tiles, nws, ncnt = hip.wgrad_group_sizes(shapes, 2, 0)
ws, cnt = torch.empty(nws, device=dev), torch.zeros(ncnt, device=dev, dtype=torch.int32)
ops = [(x, P(f(M, 4 * D))), (h, dy), (x, dy), (x, P(f(M, 3 * D)))]
outs = [torch.zeros(a, b, device=dev) for a, b in shapes]
probs = [(A, Bm, C, a, b) for (A, Bm), C, (a, b) in zip(ops, outs, shapes)]
qkv = f(M, 3 * D)
qkvp = P(qkv)
keep = torch.ones(B, T, dtype=torch.int32, device=dev)
out, outp, dout = torch.empty(M, D, device=dev), hip.Planes.empty(M, D, dev), f(M, D) * 0.1
lse, delta = torch.empty(B, nH, T, device=dev), torch.empty(M, nH, device=dev)
doutp = P(dout)
dq32, dqp = torch.empty(hip.attn_bwd_x3_dq_floats(B, T, D), device=dev), hip.Planes.empty(M, 3 * D, dev)
aws = torch.empty(hip.attn_bwd_x3_ws(B, T, D), device=dev)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
mu, rs = torch.zeros(M, device=dev), torch.ones(M, device=dev)
xf, dyf = f(M, D), f(M, D)
lnp, dxp = hip.Planes.empty(M, D, dev), hip.Planes.empty(M, D, dev)
dx, dg, db = torch.empty(M, D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
lnws = torch.empty(hip.lib().mmtg_layernorm_bwd_ws(M, D), device=dev)
for _ in range(3):
    hip.gemm_x3(x, wq, cq, M, 3 * D, D, bias=bq)
    hip.gemm_x3(x, w1, None, M, 4 * D, D, planes=gp, ldc=4 * D, bias=b1, epi=hip.EPI_GELU, aux2=pre)
    hip.gemm_x3(x, wp, c2, M, D, D, bias=b2, epi=hip.EPI_RESID, aux=res, ldaux=D, drop_p=0.1, drop_seed=3)
    hip.gemm_x3(h, w2, c2, M, D, 4 * D, bias=b2, epi=hip.EPI_RESID, aux=res, ldaux=D, drop_p=0.1, drop_seed=4)
    hip.gemm_x3(dy, w2t, None, M, 4 * D, D, planes=dup, ldc=4 * D, epi=hip.EPI_DGELU, aux=pre, ldaux=4 * D, aux2=bands)
    hip.wgrad_group(probs, M, 2, ws, cnt, config=6)
    hip.attn_fwd_x3(qkvp, keep, out, outp, lse, B, T, nH, 64, drop_p=0.1, drop_seed=1)
    hip.attn_bwd_x3(qkvp, keep, out, doutp, lse, delta, dq32, dqp, B, T, nH, 64, drop_p=0.1, drop_seed=1, dbias=torch.zeros(3 * D, device=dev), dbias_ws=aws)
    hip.layernorm_fwd_x3(xf, lnp, gam, bet, mu, rs, M, D)
    hip.layernorm_bwd_x3(dyf, xf, gam, mu, rs, None, dx, dg, db, M, D, dxp, drop_p=0.1, drop_seed=2, ws=lnws)
torch.cuda.synchronize()
print("ok")
