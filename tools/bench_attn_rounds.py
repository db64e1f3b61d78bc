#!/usr/bin/env python3
"""Attention launch time against the number of (batch row, head) workgroups: 512 resident slots (two 8-wave workgroups per CU on
256 CUs) -- how much of the 768-workgroup training launch is the half-empty second round?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip
T, nH, dh = 236, 12, 64
D = nH * dh
dt = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (16, 21, 32, 42, 43, 48, 56, 64, 85, 86, 128):
    qkv = (torch.randn(B, T, 3 * D, device="cuda") * 0.5).to(dt)
    keep = torch.ones(B, T, dtype=torch.int32, device="cuda")
    out = torch.empty(B, T, D, device="cuda", dtype=dt)
    dout = (torch.randn(B, T, D, device="cuda") * 0.5).to(dt)
    lse = torch.empty(B, nH, T, device="cuda"); delta = torch.empty(B, nH, T, device="cuda")
    dq32 = torch.empty(B * T, D, device="cuda"); dqkv = torch.empty(B, T, 3 * D, device="cuda", dtype=dt)
    f = timeit(lambda: hip.attn_fwd(qkv, keep, out, lse, B, T, nH, dh, drop_p=0.1, drop_seed=1))
    bw = timeit(lambda: hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=0.1, drop_seed=1, delta_ready=True))
    print("B %3d  workgroups %4d (%.2f rounds of 512)  fwd %6.1f us (%.3f us/wg)  bwd %6.1f us (%.3f us/wg)" % (B, B * nH, B * nH / 512.0, f, f / (B * nH), bw, bw / (B * nH)))
