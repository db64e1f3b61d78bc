#!/usr/bin/env python3
"""Round 6: the one-launch MLP of the decode token step (mmtg_decode_mlp) against the two launches it replaces
(mmtg_decode_gemm mode 0 + mode 2), stand-alone, graph-replayed over 12 layers' worth of DIFFERENT weights (113 MB: past the
L2s, inside the Infinity Cache -- as inside the token step), and its in-kernel timeline (12 real-time stamps per workgroup).

    python tools/decode_mlp_timeline.py [M=256]

Prints us per MLP for: pair, fused (write-through hand-off), fused (L2 hand-off); then per stage the median / p90 / max over the
256 workgroups of one launch (100 MHz real-time counter: 10 ns resolution)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mmtg_amd import hip  # noqa: E402

DEV = "cuda"
STAGES = ["start -> first K tile landed", "phase 1 K loop (c_fc, K = 768)", "GELU epilogue + G stores acknowledged",
          "hand-off wait (8 workgroups, one XCD)", "G rows -> LDS", "phase 2 (mlp.c_proj K slice, from LDS)",
          "partial published (write-through, acknowledged)", "arrival counters bumped (2 returning atomics)",
          "last arrivers: strips summed over the 8 slices, stored"]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    hot = len(sys.argv) > 2 and sys.argv[2] == "hot"                 # every layer reads layer 0's weights (inside the L2s)
    D, HID, NP, L = 768, 3072, hip.DG_NP, 12
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, D, generator=g) * 2.0).to(torch.bfloat16).to(DEV)
    xf = x.float()
    st = torch.zeros(M, NP, 2, device=DEV)
    st[:, :D // 32, 0] = xf.view(M, D // 32, 32).sum(2)
    st[:, :D // 32, 1] = (xf * xf).view(M, D // 32, 32).sum(2)
    Ws = []
    for l in range(L):
        W1 = (torch.randn(HID, D, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
        W2 = (torch.randn(D, HID, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
        c1, b1 = torch.randn(HID, generator=g).to(DEV) * 0.01, torch.randn(HID, generator=g).to(DEV) * 0.1
        b2 = (0.1 * torch.randn(D, generator=g)).to(DEV)
        Ws.append((W1, c1, b1, W2, b2))
    if hot:
        Ws = [Ws[0]] * L
        print("HOT weights: all 12 launches of a replay read the same 9.4 MB")
    G = torch.empty(M, HID, dtype=torch.bfloat16, device=DEV)
    C1 = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    so = torch.zeros(M, NP, 2, device=DEV)
    tiles = -(-M // 64) * (D // 64)
    ws0 = torch.empty(tiles * 4 * 4096, device=DEV)
    cnt0 = torch.zeros(tiles * 4, dtype=torch.int32, device=DEV)
    ws = torch.empty(hip.decode_mlp_ws_floats(M), device=DEV)
    sync = torch.zeros(hip.decode_mlp_sync_words(), dtype=torch.int64, device=DEV)

    def pair(l):
        W1, c1, b1, W2, b2 = Ws[l]
        hip.decode_gemm(0, x, W1, G, M, HID, D, bias=b1, colsum=c1, stats_in=st, np_in=D // 32, act=hip.EPI_GELU)
        hip.decode_gemm(2, G, W2, C1, M, D, HID, bias=b2, resid=x, stats_out=so, splits=4, ws=ws0, counters=cnt0)

    def fused(l, plain, trace=None):
        W1, c1, b1, W2, b2 = Ws[l]
        hip.decode_mlp(x, st, D // 32, 1e-5, W1, c1, b1, W2, b2, G, C1, so, ws, sync, M, D, plain=plain, trace=trace)

    def timed(fn, reps=20):
        for l in range(L):
            fn(l)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for l in range(L):
                fn(l)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / (reps * L)

    print("census (rows: workgroup id %% 8, columns: XCC_ID): %s" % hip.decode_mlp_census(DEV).tolist())
    print("M = %d: us per MLP (graph replay of 12 layers' weights)" % M)
    print("  two launches (c_fc | mlp.c_proj x4)   %.2f" % timed(pair))
    for plain in (False, True):
        print("  one launch, %-13s hand-off   %.2f   (error word %d)" % ("L2" if plain else "write-through", timed(lambda l: fused(l, plain)),
                                                                        int(sync[65].item())))
        sync.zero_()
    for plain in (False, True):
        tr = torch.zeros(256 * 12, dtype=torch.int64, device=DEV)
        for rep in range(3):
            for l in range(L):
                fused(l, plain, tr if (rep == 2 and l == L - 1) else None)
        torch.cuda.synchronize()
        t = tr.view(256, 12).double().cpu() / 100.0                  # us
        live = t[:, 8] > 0                                          # (workgroups past row M never reach the later stamps)
        t = t[live]
        t0 = t[:, 0].min()
        print("timeline, %s hand-off (%d workgroups; us, median / p90 / max over workgroups):" % ("L2" if plain else "write-through", int(live.sum())))
        print("  %-52s %6.2f / %6.2f / %6.2f" % ("workgroup start after the first one", (t[:, 0] - t0).median(), (t[:, 0] - t0).quantile(0.9), (t[:, 0] - t0).max()))
        for i, name in enumerate(STAGES):
            d = t[:, i + 1] - t[:, i]
            print("  %-52s %6.2f / %6.2f / %6.2f" % (name, d.median(), d.quantile(0.9), d.max()))
        tot = t[:, 9] - t[:, 0]
        print("  %-52s %6.2f / %6.2f / %6.2f   (first start -> last end %.2f)" % ("workgroup lifetime", tot.median(), tot.quantile(0.9), tot.max(), t[:, 9].max() - t0))


if __name__ == "__main__":
    main()
