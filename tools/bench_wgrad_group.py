#!/usr/bin/env python3
"""The four weight gradients of a GPT-2-base block at the training shape (15104 tokens): four slab launches + four ordered sums
(round 2) against ONE grouped launch with the in-kernel reduction (mmtg_wgrad_group), warm and HBM-cold (a 1 GiB fill between
launches), per launch in microseconds.  MMTG_WGRAD_FENCE=1 in the environment selects the fence-pair variant of the kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip
from mmtg_amd.engine import _wgrad_splits

DEV = "cuda"
K, D = int(os.environ.get("TOKENS", "15104")), int(os.environ.get("WIDTH", "768"))
g = torch.Generator(device=DEV).manual_seed(3)
mk = lambda n: (torch.randn(K, n, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
m2, du, gact, dy, ctx, dy2, a1, dqkv = mk(D), mk(4 * D), mk(4 * D), mk(D), mk(D), mk(D), mk(D), mk(3 * D)
shapes = [(m2, du, D, 4 * D), (gact, dy, 4 * D, D), (ctx, dy2, D, D), (a1, dqkv, D, 3 * D)]
Cs = [torch.empty(M, N, device=DEV) for (_, _, M, N) in shapes]
tiles = sum((M // 128) * (N // 128) for (_, _, M, N) in shapes)
flops = sum(2.0 * M * N * K for (_, _, M, N) in shapes)
fill = torch.empty(1 << 28, device=DEV)      # 1 GiB
slabs = torch.empty(12 * 3072 * 768, device=DEV)


def baseline():
    for (A, B, M, N), C_ in zip(shapes, Cs):
        s = _wgrad_splits(M, N, K, True)
        hip.gemm(A, B, slabs, M, N, K, transA=True, epi=hip.EPI_SPLIT, out_f32=True, splits=s)
        hip.slab_sum(slabs, s, M * N, C_, M * N, accumulate=False)


def grouped(S, config=0):
    _, nws, ncnt = hip.wgrad_group_sizes([(M, N) for (_, _, M, N) in shapes], S, config)
    ws = torch.empty(nws, device=DEV)
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=DEV)
    probs = [(A, B, C_, M, N) for (A, B, M, N), C_ in zip(shapes, Cs)]
    return lambda: hip.wgrad_group(probs, K, S, ws, cnt, config=config)


def timeit(fn, cold, iters=12):
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


print("tokens %d width %d: %d tiles, %.1f GFLOP per block; MMTG_WGRAD_FENCE=%s" % (K, D, tiles, flops / 1e9, os.environ.get("MMTG_WGRAD_FENCE")))
for cold in (False, True):
    t = timeit(baseline, cold)
    print("%-5s 4 slab launches + 4 ordered sums          %7.1f us  %6.0f TFLOP/s" % ("cold" if cold else "warm", t, flops / t / 1e6))
    for S in [int(x) for x in os.environ.get("SPLITS", "1,2,3").split(",")]:
        t = timeit(grouped(S), cold)
        print("%-5s grouped launch, %d K split(s), %4d workgroups %7.1f us  %6.0f TFLOP/s" % ("cold" if cold else "warm", S, tiles * S, t, flops / t / 1e6))
    if not os.environ.get("MMTG_WGRAD_FENCE"):
        for S in [int(x) for x in os.environ.get("SPLITS8", "1,2,3").split(",")]:
            t8 = hip.wgrad_group_sizes([(M, N) for (_, _, M, N) in shapes], S, 1)[0]
            t = timeit(grouped(S, 1), cold)
            print("%-5s grouped eight-phase 256x256, %d K split(s), %4d workgroups %7.1f us  %6.0f TFLOP/s" % ("cold" if cold else "warm", S, t8 * S, t, flops / t / 1e6))
