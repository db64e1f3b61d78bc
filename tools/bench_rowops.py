#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip
M, D = 64 * 236, 768
dt = torch.bfloat16
x = torch.randn(M, D, device="cuda").to(dt); dy = torch.randn(M, D, device="cuda").to(dt); dres = torch.randn(M, D, device="cuda").to(dt)
y = torch.empty_like(x); dx = torch.empty_like(x); dxm = torch.empty_like(x)
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
mu = torch.empty(M, device="cuda"); rs = torch.empty(M, device="cuda")
dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda"); dc = torch.zeros(D, device="cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
hip.layernorm_fwd(x, y, g, b, mu, rs, M, D)
print("ln fwd          %.1f us" % timeit(lambda: hip.layernorm_fwd(x, y, g, b, mu, rs, M, D)))
print("ln bwd plain    %.1f us" % timeit(lambda: hip.layernorm_bwd(dy, x, g, mu, rs, None, dx, dg, db, M, D)))
print("ln bwd +dres    %.1f us" % timeit(lambda: hip.layernorm_bwd(dy, x, g, mu, rs, dres, dx, dg, db, M, D)))
print("ln bwd +colsum  %.1f us" % timeit(lambda: hip.layernorm_bwd(dy, x, g, mu, rs, dres, dx, dg, db, M, D, dcolsum=dc)))
print("ln bwd +mask    %.1f us" % timeit(lambda: hip.layernorm_bwd(dy, x, g, mu, rs, dres, dx, dg, db, M, D, dx_masked=dxm, drop_p=0.1, drop_seed=3, dcolsum=dc)))
q = torch.randn(M, 3 * D, device="cuda").to(dt); cs = torch.zeros(3 * D, device="cuda")
print("colsum 2304     %.1f us" % timeit(lambda: hip.colsum(q, M, 3 * D, cs)))
