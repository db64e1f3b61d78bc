#!/usr/bin/env python3
"""Grouped split-precision weight gradients (M = 15104 tokens): the four products of a GPT-2 block, the tied embedding's, the
projector's -- config 2 (three passes, four workgroups per CU) against config 6 (combined four-plane stages, two per CU) over K splits.
    python3 tools/bench_wgrad_x3.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mmtg_amd import hip

K = 15104
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
PL = lambda r, c: hip.split_planes(torch.randn(r, c, device=dev, generator=g) * 0.1, r, c, hip.Planes.empty(r, c, dev))


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


groups = {"block (768x3072, 3072x768, 768x768, 768x2304)": ((768, 3072), (3072, 768), (768, 768), (768, 2304)),
          "tied embedding (13440x768)": ((13440, 768),),
          "projector (768x512, 512x2048)": ((768, 512), (512, 2048))}
for name, shapes in groups.items():
    ops = {}
    probs = []
    for (Mi, Ni) in shapes:
        A = ops.setdefault(("a", Mi), PL(K, Mi))
        B = ops.setdefault(("b", Ni), PL(K, Ni))
        probs.append((A, B, torch.zeros(Mi, Ni, device=dev), Mi, Ni))
    fl = sum(2.0 * K * Mi * Ni for (Mi, Ni) in shapes) * 3
    line = []
    for cfg in (2, 6):
        for splits in ((1, 2, 3, 4) if cfg == 2 else (2, 3, 4, 5, 6, 8, 12, 14)):
            tiles, nws, ncnt = hip.wgrad_group_sizes(shapes, splits, 0)
            ws = torch.empty(nws, device=dev) if splits > 1 else None
            cnt = torch.zeros(ncnt, dtype=torch.int32, device=dev)
            t = timed(lambda: hip.wgrad_group(probs, K, splits, ws, cnt, accumulate=False, config=cfg))
            line.append("cfg %d x%d: %.0f us (%.0f TF)" % (cfg, splits, t, fl / t / 1e6))
    print("%-50s %d tiles | %s" % (name, tiles, "  ".join(line)))
