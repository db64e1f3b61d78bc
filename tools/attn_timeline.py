#!/usr/bin/env python3
"""Per-wave timeline of the whole-head attention forward kernel (mmtg_attn_trace): where a workgroup's time goes.

  python tools/attn_timeline.py [drop_p]
Stamps (s_memrealtime, 100 MHz): entry, loads issued, first chunk landed (after the barrier), long tile done,
long tile stored, exit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmtg_amd import hip

B, T, nH, dh = 64, 236, 12, 64
D = nH * dh
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
qkv = (torch.randn(B, T, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
keep = torch.ones(B, T, dtype=torch.int32, device="cuda")
out = torch.empty(B, T, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, nH, T, device="cuda")
for _ in range(3):
    hip.attn_fwd(qkv, keep, out, lse, B, T, nH, dh, drop_p=p, drop_seed=1)
torch.cuda.synchronize()
buf = torch.zeros(B * nH * 8, 8, device="cuda", dtype=torch.int64)
hip.attn_trace(buf)
hip.attn_fwd(qkv, keep, out, lse, B, T, nH, dh, drop_p=p, drop_seed=1)
torch.cuda.synchronize()
hip.attn_trace(None)
r = buf.cpu().numpy().reshape(B * nH, 8, 8)
t0 = r[:, :, 0].min()
us = lambda x: (x - t0) / 100.0
print("launch wall (first entry -> last exit): %.1f us" % us(r[:, :, 5].max()))
ent = us(r[:, 0, 0])
print("workgroup entry times: min %.1f  median %.1f  p75 %.1f  max %.1f us" % (ent.min(), np.median(ent), np.percentile(ent, 75), ent.max()))
first = ent < 2.0
print("workgroups entering in the first 2 us: %d of %d" % (first.sum(), len(ent)))
for nm, sel in (("first round", first), ("later", ~first)):
    if sel.sum() == 0:
        continue
    rr = r[sel]
    seg = lambda a, b: (rr[:, :, b] - rr[:, :, a]) / 100.0
    wg_life = (rr[:, :, 5].max(1) - rr[:, :, 0].min(1)) / 100.0
    print("%s (%d workgroups): lifetime median %.1f  max %.1f us" % (nm, sel.sum(), np.median(wg_life), wg_life.max()))
    print("   issue loads      median %.2f  max %.2f" % (np.median(seg(0, 1)), seg(0, 1).max()))
    print("   first chunk wait median %.2f  max %.2f" % (np.median(seg(1, 2)), seg(1, 2).max()))
    print("   long tile sweep  per wave: " + " ".join("%.1f" % np.median(seg(2, 3)[:, w]) for w in range(8)))
    print("   long tile store  per wave: " + " ".join("%.2f" % np.median(seg(3, 4)[:, w]) for w in range(8)))
    print("   short tile       per wave: " + " ".join("%.1f" % np.median(seg(4, 5)[:, w]) for w in range(8)))
xc = r[:, 0, 6]
print("XCC ids seen:", sorted(set(int(x) for x in xc)))
