#!/usr/bin/env python3
"""A few launches of the kernels that carry a round-3 training step, for `rocprofv3 --pmc` passes (tools/gpu_pmc_util_r3.sh):
the four eight-phase K-contiguous products of a GPT-2 block (qkv, fc1 + GELU, attention c_proj + residual, fc2 + residual),
the dGELU product (single-stage kernel), one grouped weight-gradient launch, the whole-head attention forward / backward at
B = 64, T = 236 with dropout, and a LayerNorm backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

B, T, nH = 64, 236, 12
M, D = B * T, 768
dev = "cuda"
t = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
x, wq, w1, wp = t(M, D), t(3 * D, D), t(4 * D, D), t(D, D)
h, w2 = t(M, 4 * D), t(D, 4 * D)
dy, res = t(M, D), t(M, D)
cq = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
c1, pre = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16), torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
c2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
b1, b2, bq = torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)
bands = torch.zeros((M + 63) // 64, 4 * D, device=dev)
# grouped weight gradients of one block
shapes = ((D, 4 * D), (4 * D, D), (D, D), (D, 3 * D))
tiles, nws, ncnt = hip.wgrad_group_sizes(shapes, 2, 0)
ws, cnt = torch.empty(nws, device=dev), torch.zeros(ncnt, device=dev, dtype=torch.int32)
ops = [(x, t(M, 4 * D)), (h, dy), (x, dy), (x, t(M, 3 * D))]
outs = [torch.zeros(a, b, device=dev) for a, b in shapes]
probs = [(A, Bm, C, a, b) for (A, Bm), C, (a, b) in zip(ops, outs, shapes)]
# attention
qkv = t(B, T, 3 * D)
keep = torch.ones(B, T, dtype=torch.int32, device=dev)
out, dout = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16), t(B, T, D)
lse, delta = torch.empty(B, nH, T, device=dev), torch.empty(B, nH, T, device=dev)
dq32, dqkv = torch.empty(M, D, device=dev), torch.empty(B, T, 3 * D, device=dev, dtype=torch.bfloat16)
# layernorm backward
gam = torch.ones(D, device=dev)
mu, rs = torch.zeros(M, device=dev), torch.ones(M, device=dev)
dx, dg, db = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
lnws = torch.empty(hip.lib().mmtg_layernorm_bwd_ws(M, D), device=dev)
for _ in range(3):
    hip.gemm(x, wq, cq, M, 3 * D, D, transB=True, bias=bq)
    hip.gemm(x, w1, c1, M, 4 * D, D, transB=True, bias=b1, epi=hip.EPI_GELU, aux2=pre)
    hip.gemm(x, wp, c2, M, D, D, transB=True, bias=b2, epi=hip.EPI_RESID, aux=res, ldaux=D)
    hip.gemm(h, w2, c2, M, D, 4 * D, transB=True, bias=b2, epi=hip.EPI_RESID, aux=res, ldaux=D)
    hip.gemm(dy, w2.t().contiguous() if False else t(4 * D, D), c1, M, 4 * D, D, transB=True, epi=hip.EPI_DGELU, aux=pre, ldaux=4 * D, aux2=bands)
    hip.wgrad_group(probs, M, 2, ws, cnt)
    hip.attn_fwd(qkv, keep, out, lse, B, T, nH, 64, drop_p=0.1, drop_seed=1)
    hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, 64, drop_p=0.1, drop_seed=1)
    hip.layernorm_bwd(dy, x, gam, mu, rs, None, dx, dg, db, M, D, ws=lnws)
torch.cuda.synchronize()
print("ok")
