#!/usr/bin/env python3
"""Timing of the attention kernels at the training shape (B=64, T=236, 12 heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip
B, T, nH, dh = 64, 236, 12, 64
D = nH * dh
dt = torch.bfloat16
qkv = (torch.randn(B, T, 3 * D, device="cuda") * 0.5).to(dt)
keep = torch.ones(B, T, dtype=torch.int32, device="cuda")
out = torch.empty(B, T, D, device="cuda", dtype=dt)
dout = (torch.randn(B, T, D, device="cuda") * 0.5).to(dt)
lse = torch.empty(B, nH, T, device="cuda")
delta = torch.empty(B, nH, T, device="cuda")
dq32 = torch.empty(B * T, D, device="cuda")
dqkv = torch.empty(B, T, 3 * D, device="cuda", dtype=dt)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for p in (0.0, 0.1):
    f = timeit(lambda: hip.attn_fwd(qkv, keep, out, lse, B, T, nH, dh, drop_p=p, drop_seed=1))
    bw = timeit(lambda: hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=p, drop_seed=1))
    print("ablate=%s drop=%.1f fwd %.1f us  bwd(+delta) %.1f us" % (os.environ.get("MMTG_ATTN_ABLATE", "0"), p, f, bw))
    # as the trainer calls it: delta from the preceding dgrad product's ROWDOT epilogue, the c_attn bias gradient from the kernels
    dbias = torch.zeros(3 * D, device="cuda")
    ws = torch.empty(hip.attn_bwd_bias_rows(B, T, hip.BF16), 3 * D, device="cuda")
    b1 = timeit(lambda: hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=p, drop_seed=1, delta_ready=True))
    b2 = timeit(lambda: hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=p, drop_seed=1, delta_ready=True,
                                     dbias=dbias, dbias_ws=ws))
    print("           drop=%.1f bwd without delta %.1f us; with the bias-gradient rows (trainer's call) %.1f us" % (p, b1, b2))
