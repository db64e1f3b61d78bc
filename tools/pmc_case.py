#!/usr/bin/env python3
"""A few launches of the three dominant GEMM shapes of a training step, for `rocprofv3 --pmc` passes
(tools/gpu_pmc_diag.sh): fc1 forward (NT, single-stage kernel), fc2 forward (NT, 192x128 tiles),
fc weight gradient (TN slabs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

M, D = 64 * 236, 768
dev = "cuda"
t = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
x, w1 = t(M, D), t(4 * D, D)
h, w2 = t(M, 4 * D), t(D, 4 * D)
dy = t(M, D)
c1, pre = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16), torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
c2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
part = torch.empty(5, 4 * D, D, device=dev)
b1, b2 = torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev)
for _ in range(3):
    hip.gemm(x, w1, c1, M, 4 * D, D, transB=True, bias=b1, epi=hip.EPI_GELU, aux2=pre)
    hip.gemm(h, w2, c2, M, D, 4 * D, transB=True, bias=b2, epi=hip.EPI_RESID, aux=dy, ldaux=D)
    hip.gemm(h, dy, part, 4 * D, D, M, transA=True, epi=hip.EPI_SPLIT, out_f32=True, splits=5)
torch.cuda.synchronize()
