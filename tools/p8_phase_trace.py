#!/usr/bin/env python3
"""Per-phase shader-clock stamps of ONE K tile (K tile 8, workgroup 0, waves 0 and 4 = the two wave-row groups) of the eight-phase
kernel, K-contiguous (NT) against K-strided (the weight-gradient form): where do the K-strided form's extra cycles per K tile go?
Needs the diagnostic build:  MMTG_EXTRA_DEFS=-DMMTG_P8_PHASE_TRACE python -m mmtg_amd.build --force
    python tools/p8_phase_trace.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mmtg_amd import hip

dev = "cuda"
t = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def trace(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = torch.zeros(1 << 14, 6, device=dev, dtype=torch.int64)
    hip.gemm_trace(buf)
    fn()
    torch.cuda.synchronize()
    hip.gemm_trace(None)
    r = buf.cpu().numpy()
    n = int(os.environ.get("NWG", "0")) or int((r[:, 3] != 0).sum())
    wg = r[:n]
    loop = (wg[:, 2] - wg[:, 1]) * 0.01
    nk = wg[:, 4] & 0xFFFFF
    print("%s: %d workgroups, K loop per K tile median %.2f us" % (label, n, float(np.median(loop / np.maximum(nk, 1)))))
    names = ["P1 (a0,b0)", "P2 (a0,b1)", "P3 (a1,b1)", "P4 (a1,b0)"]
    for wave in (0, 4):
        flat = r[n + 4 * wave:n + 4 * wave + 4].reshape(-1)[:20].astype(np.int64)
        if flat[0] == 0:
            print("  (no phase stamps: not the diagnostic build)")
            return
        d = lambda a, b: int((flat[b] - flat[a]) & 0xFFFFFFFF)
        print("  wave %d (wave-row group %d), cycles: loads issued | wait at first barrier + LDS | MFMAs issued | second barrier" % (wave, wave // 4))
        tot = 0
        for ph in range(4):
            b = 4 * ph + (1 if ph else 0)      # stamp indices: phase 1 = 0..4, phase 2 = 4..8, ...
            s0 = 4 * ph
            a_, b_, c_, d_, e_ = s0, s0 + 1, s0 + 2, s0 + 3, s0 + 4
            print("    %-12s %6d | %6d | %6d | %6d   = %6d" % (names[ph], d(a_, b_), d(b_, c_), d(c_, d_), d(d_, e_), d(a_, e_)))
            tot += d(a_, e_)
        print("    K tile: %d cycles" % tot)


M = 15104
# NT: 15104 x 3072 x 768 (256-row tiles)
A, B, C_ = t(M, 768), t(3072, 768), torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
os.environ.setdefault("MMTG_GEMM_P8_ROWS", "256")
trace(lambda: hip.gemm(A, B, C_, M, 3072, 768, transB=True), "NT 15104 x 3072 x 768")
# K-strided (weight gradient): 768 x 3072, K = 15104 tokens, 7 slabs on the eight-phase K-strided form
X, dY = t(M, 768), t(M, 3072)
part = torch.empty(7, 768, 3072, device=dev)
trace(lambda: hip.gemm(X, dY, part, 768, 3072, M, transA=True, epi=hip.EPI_SPLIT, out_f32=True, splits=7, flags=hip.GEMM_P8), "TN 768 x 3072, K = 15104, 7 slabs")
