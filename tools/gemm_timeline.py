#!/usr/bin/env python3
"""In-kernel timeline of one LDS-DMA GEMM launch (mmtg_gemm_trace): where a workgroup's time goes.

Per workgroup the kernel stamps s_memrealtime (100 MHz) at entry, after its first K tile has landed,
at the end of the K loop and at exit (output stores drained).  Prints the launch's wall time, the
distribution of those segments and how the workgroups were scheduled over time (rounds).

  python tools/gemm_timeline.py M N K [NT|NN|TN] [epi] [flags] [splits]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmtg_amd import hip

M, N, K = (int(x) for x in sys.argv[1:4])
layout = sys.argv[4] if len(sys.argv) > 4 else "NT"
epi = int(sys.argv[5]) if len(sys.argv) > 5 else hip.EPI_NONE
flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
splits = int(sys.argv[7]) if len(sys.argv) > 7 else 1
dev, dt = "cuda", torch.bfloat16
tA, tB = layout[0] == "T", layout[1] == "T"


def t(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


A = t(K, M) if tA else t(M, K)
B = t(N, K) if tB else t(K, N)
atomic = layout == "TN"
if atomic:
    epi = hip.EPI_ATOMIC
C = torch.zeros(M, N, device=dev, dtype=torch.float32 if atomic else dt)
kw = {}
if epi == hip.EPI_GELU:
    kw["aux2"] = torch.empty(M, N, device=dev, dtype=dt)
if epi in (hip.EPI_RESID, hip.EPI_DGELU, hip.EPI_ROWDOT):
    kw["aux"] = t(M, N)
if epi == hip.EPI_ROWDOT:
    kw["aux2"] = torch.empty(M, N // 64, device=dev, dtype=torch.float32)
bias = None if epi in (hip.EPI_ATOMIC, hip.EPI_DGELU) else torch.zeros(N, device=dev)


def run():
    hip.gemm(A, B, C, M, N, K, transA=tA, transB=tB, epi=epi, splits=splits, bias=bias, flags=flags, **kw)


for _ in range(3):
    run()
torch.cuda.synchronize()
nwg = 1 << 16
buf = torch.zeros(nwg, 6, device=dev, dtype=torch.int64)
if os.environ.get("COLD"):   # operands behind a 1 GiB fill: HBM-cold, as inside the training step
    torch.empty(1 << 28, device=dev, dtype=torch.float32).fill_(1.0)
    torch.cuda.synchronize()
hip.gemm_trace(buf)
run()
torch.cuda.synchronize()
hip.gemm_trace(None)
r = buf.cpu().numpy()
r = r[r[:, 3] != 0]
t0 = r[:, 0].min()
us = lambda x: x * 0.01   # 100 MHz ticks -> us
start, first, loop_end, end = (us(r[:, i] - t0) for i in range(4))
nk = r[:, 4] & 0xFFFFF
loop_cycles = r[:, 4] >> 20          # eight-phase kernel only: shader-clock cycles of the K loop
xcc = (r[:, 5] >> 32) & 0xF
hw = r[:, 5] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("%s M=%d N=%d K=%d epi=%d flags=%d splits=%d: %d workgroups, kernel span %.1f us" % (layout, M, N, K, epi, flags, splits, len(r), end.max()))


def dist(name, x):
    q = np.percentile(x, [0, 10, 50, 90, 100])
    print("  %-34s min %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % ((name,) + tuple(q)))


dist("entry -> first K tile landed", first - start)
dist("K loop (per workgroup)", loop_end - first)
dist("K loop per K tile", (loop_end - first) / np.maximum(nk, 1))
dist("epilogue + store drain", end - loop_end)
if loop_cycles.max() > 0:
    ghz = loop_cycles / np.maximum((loop_end - first) * 1e3, 1e-9)     # cycles / ns
    q = np.percentile(ghz, [0, 10, 50, 90, 100])
    print("  %-34s min %7.3f  p10 %7.3f  p50 %7.3f  p90 %7.3f  max %7.3f GHz" % (("shader clock inside the K loop",) + tuple(q)))
    # MFMA-issue share of the loop: 16x16x32 bf16 = 16 cycles each on a SIMD's matrix pipe, two waves per SIMD
    rows = 256 if int(os.environ.get("MMTG_GEMM_P8_ROWS", "0") or 0) == 0 else int(os.environ["MMTG_GEMM_P8_ROWS"])
    print("  %-34s %.1f %% (at %d-row tiles: %d MFMAs of 16 cycles per wave and K tile, two waves per SIMD)" % (
        "matrix-pipe cycles / loop cycles", 100.0 * np.median(nk * (rows // 2 // 16) * 8 * 2 * 16 / np.maximum(loop_cycles, 1)), rows, (rows // 2 // 16) * 8))
dist("workgroup lifetime", end - start)
# scheduling: start-time histogram in 2 us bins
span = end.max()
bins = np.arange(0, span + 2, 2.0)
h, _ = np.histogram(start, bins)
print("  workgroup starts per 2 us bin:", " ".join(str(int(x)) for x in h))
h, _ = np.histogram(end, bins)
print("  workgroup exits  per 2 us bin:", " ".join(str(int(x)) for x in h))
# residency: how many workgroups alive over time
ts = np.linspace(0, span, 21)
alive = [(int(((start <= x) & (end > x)).sum())) for x in ts]
print("  alive workgroups at 5%% steps:", " ".join(map(str, alive)))
slots = {}
for i in range(len(r)):
    slots.setdefault((int(xcc[i]), int(se[i]), int(cu[i])), []).append(i)
per = np.array([len(v) for v in slots.values()])
print("  distinct (xcc,se,cu) ids seen: %d, workgroups per id min/mean/max %d/%.1f/%d" % (len(slots), per.min(), per.mean(), per.max()))
