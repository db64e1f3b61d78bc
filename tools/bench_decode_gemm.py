#!/usr/bin/env python3
"""Decode-step products (M = batch 256, K-contiguous bf16 weights): tile configuration x K-split sweep.

  python tools/bench_decode_gemm.py            (sets MMTG_SKINNY_CFG per run; prints us per product incl. nothing else)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmtg_amd import hip

M = int(os.environ.get("M", "256"))
dev, dt = "cuda", torch.bfloat16
shapes = [("c_attn", 2304, 768), ("attn c_proj", 768, 768), ("c_fc", 3072, 768), ("mlp c_proj", 768, 3072), ("lm head", 13440, 768)]
names = {0: "256x32", 1: "64x64", 2: "128x64", 3: "64x128", 4: "128x32", 5: "64x32"}


def timeit(fn, n=40):
    """us per call from a replayed hipGraph of n calls (eager Python launches are host-bound below ~8 us)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


print("%-12s %-8s %s" % ("product", "tiles", "  ".join("s=%-2d" % s for s in (1, 2, 3, 4, 6, 8, 12))))
for name, N, K in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    for cfg in (0, 1, 2, 3, 4, 5):
        os.environ["MMTG_SKINNY_CFG"] = str(cfg)
        row = []
        for s in (1, 2, 3, 4, 6, 8, 12):
            if K // s < 128 or (K // s) % 64:
                row.append("   - ")
                continue
            part = torch.empty(s * M * N, device=dev, dtype=torch.float32)
            try:
                t = timeit(lambda: hip.gemm(A, W, part, M, N, K, transB=True, ldb=K, ldc=N, epi=hip.EPI_SPLIT, out_f32=True, splits=s))
                row.append("%5.1f" % t)
            except Exception as e:
                row.append("  err")
        print("%-12s %-8s %s" % (name, names[cfg], " ".join(row)))
