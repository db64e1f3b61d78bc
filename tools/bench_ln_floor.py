#!/usr/bin/env python3
"""N1 (fused LayerNorm + QKV / fc1 in training) settled by measurement: the LayerNorm forward of the residual stream
(15104 x 768 bf16: 23 MB read + 23 MB written + statistics) against plain device copies of the same bytes -- the floor any
"statistics from the producing epilogue + a normalise pass" variant has, because the normalised rows must exist in HBM for
the weight gradient whatever computes the statistics."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

DEV = "cuda"
M, D = 15104, 768
x = torch.randn(M, D, device=DEV).to(torch.bfloat16)
y = torch.empty_like(x)
g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
mu, rs = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
fill = torch.empty(1 << 28, device=DEV)


def t(fn, cold, iters=20):
    fn(); fn()
    tot = 0.0
    for _ in range(iters):
        if cold:
            fill.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return 1e3 * tot / iters


mb = 2 * M * D * 2 / 1e6
for cold in (False, True):
    a = t(lambda: hip.layernorm_fwd(x, y, g, b, mu, rs, M, D), cold)
    c = t(lambda: y.copy_(x), cold)
    d = t(lambda: hip.dropout_apply(x, y, M * D, 0.0, 1), cold)
    print("%-5s layernorm_fwd %6.2f us (%.2f TB/s)   torch copy %6.2f us (%.2f TB/s)   mmtg_dropout_apply(p=0) copy %6.2f us (%.2f TB/s)   [%.1f MB each]"
          % ("cold" if cold else "warm", a, mb / a, c, mb / c, d, mb / d, mb))
