#!/usr/bin/env python3
"""Per-wave timeline of the whole-head attention backward kernels (mmtg_attn_trace): where a workgroup's time goes.

  python tools/attn_bwd_timeline.py [drop_p]
Stamps (s_memrealtime, 100 MHz) per wave: entry, loads issued, first chunk landed (keep-bit matrix built meanwhile), the
streamed first tile done, every tile done and stored, exit (after the bias-gradient flush).  Rows 0 .. B nH - 1: the dK / dV
kernel (4 waves under dropout), rows B nH ..: the dQ kernel (8 waves)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmtg_amd import hip

B, T, nH, dh = 64, 236, 12, 64
D = nH * dh
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
dt = torch.bfloat16
qkv = (torch.randn(B, T, 3 * D, device="cuda") * 0.5).to(dt)
keep = torch.ones(B, T, dtype=torch.int32, device="cuda")
out = torch.empty(B, T, D, device="cuda", dtype=dt)
dout = (torch.randn(B, T, D, device="cuda") * 0.5).to(dt)
lse = torch.empty(B, nH, T, device="cuda")
delta = torch.empty(B, nH, T, device="cuda")
dq32 = torch.empty(B * T, D, device="cuda")
dqkv = torch.empty(B, T, 3 * D, device="cuda", dtype=dt)
dbias = torch.zeros(3 * D, device="cuda")
ws = torch.empty(hip.attn_bwd_bias_rows(B, T, hip.BF16), 3 * D, device="cuda")
hip.attn_fwd(qkv, keep, out, lse, B, T, nH, dh, drop_p=p, drop_seed=1)
call = lambda: hip.attn_bwd(qkv, keep, out, dout, lse, delta, dq32, dqkv, B, T, nH, dh, drop_p=p, drop_seed=1, delta_ready=False,
                            dbias=dbias, dbias_ws=ws)
for _ in range(3):
    call()
torch.cuda.synchronize()
buf = torch.zeros(2 * B * nH * 16, 8, device="cuda", dtype=torch.int64)
hip.attn_trace(buf)
call()
torch.cuda.synchronize()
hip.attn_trace(None)
r_all = buf.cpu().numpy().reshape(2, B * nH, 16, 8)
for name, r in (("dK / dV kernel", r_all[0]), ("dQ kernel", r_all[1])):
    nw = int(r[0, :, 7].sum())
    r = r[:, :nw]
    t0 = r[:, :, 0].min()
    us = lambda x: (x - t0) / 100.0
    print("== %s (%d waves per workgroup): first entry -> last exit %.1f us" % (name, nw, us(r[:, :, 5].max())))
    ent = us(r[:, 0, 0])
    first = ent < 2.0
    print("   workgroups entering in the first 2 us: %d of %d; entry p50 %.1f p75 %.1f max %.1f" % (first.sum(), len(ent), np.median(ent), np.percentile(ent, 75), ent.max()))
    for nm, sel in (("first round", first), ("later", ~first)):
        if sel.sum() == 0:
            continue
        rr = r[sel]
        seg = lambda a, b: (rr[:, :, b] - rr[:, :, a]) / 100.0
        life = (rr[:, :, 5].max(1) - rr[:, :, 0].min(1)) / 100.0
        print("   %s (%d workgroups): lifetime median %.1f  max %.1f us" % (nm, sel.sum(), np.median(life), life.max()))
        for lab, a, b in (("issue loads", 0, 1), ("mask + first chunk", 1, 2), ("streamed tile", 2, 3), ("other tiles + stores", 3, 4), ("bias flush + exit", 4, 5)):
            print("      %-22s per wave: %s" % (lab, " ".join("%5.2f" % np.median(seg(a, b)[:, w]) for w in range(nw))))
