#!/usr/bin/env python3
"""Long-K weight-gradient layout (A [K, M], B [K, N], both K-strided) at full chip load: per-K-tile time of the tile
configurations of the two-stage LDS-DMA ring against the single-stage 128x128 kernel.  MMTG_GEMM_TN_BIG selects (read once per
process): unset = 128x128 single stage, four workgroups per CU; 1 = 256x256, 16 waves of 64x64; 2 = 256x256, 8 waves of 128x64;
3 = 256x128, 8 waves of 64x64."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

DEV = "cuda"
M = N = 4096
K = 7552
A = (torch.randn(K, M, device=DEV) * 0.5).to(torch.bfloat16)
B = (torch.randn(K, N, device=DEV) * 0.5).to(torch.bfloat16)
C = torch.zeros(M, N, device=DEV)
fill = torch.empty(1 << 28, device=DEV)


def run():
    hip.gemm(A, B, C, M, N, K, transA=True, epi=hip.EPI_ATOMIC, splits=1)


run(); run()
torch.cuda.synchronize()
ref = None
if os.environ.get("CHECK"):
    C.zero_(); run(); torch.cuda.synchronize()
    ref = A[:, :256].float().t() @ B[:, :256].float()
    print("max err vs fp32 on the first tile: %.3e (scale %.1f)" % ((C[:256, :256] - ref).abs().max().item(), ref.abs().max().item()))
for cold in (False, True):
    tot = 0.0
    for _ in range(10):
        if cold:
            fill.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = 1e3 * tot / 10
    print("MMTG_GEMM_TN_BIG=%s %s: %.1f us, %.0f TFLOP/s, %.2f us per 64-deep K tile of a 256x256 output block" %
          (os.environ.get("MMTG_GEMM_TN_BIG", "-"), "cold" if cold else "warm", us, 2.0 * M * N * K / us / 1e6, us / (K / 64)))
