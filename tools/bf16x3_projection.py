#!/usr/bin/env python3
"""What would a split-precision ("bf16x3") training mode cost?  (error side: tools/bf16x3_study.py.)

The three passes hi hi + hi lo + lo hi of X W^T are ONE bf16 product over a three times longer contraction:
[X_hi | X_hi | X_lo] [W_hi | W_lo | W_hi]^T, fp32 accumulate -- so the mode's products run on the kernels that exist (eight-phase NT,
grouped weight gradients) at K' = 3 K with fp32 outputs.  This tool TIMES exactly those launches for every product of a GPT-2
block and the LM head at the training shape (random bf16 operands, HBM-cold like inside the step), adds the streaming cost of
splitting each fp32 operand into its bf16 triple (read 4 B + write 6 B per element, priced at the 5 TB/s this library's streaming
kernels reach) and the measured non-GEMM time of the f32 mode (attention, LayerNorm, loss, AdamW in fp32: profiles/r02_v5_bench_f32_mode.json),
and prints the projected step next to the f32 mode's.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

dev = "cuda"
M, D, V, L = 64 * 236, 768, 13440, 12
fill = torch.empty(1 << 28, device=dev)
t = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, n=6):
    fn(); fn()
    tot = 0.0
    for _ in range(n):
        fill.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3


def nt(Mm, N, K):
    A, B, C_ = t(Mm, 3 * K), t(N, 3 * K), torch.empty(Mm, N, device=dev)
    us = timeit(lambda: hip.gemm(A, B, C_, Mm, N, 3 * K, transB=True, out_f32=True))
    return us


rows = []
per_layer = 0.0
for name, N, K in (("c_attn fwd", 3 * D, D), ("attn c_proj fwd", D, D), ("c_fc fwd", 4 * D, D), ("mlp c_proj fwd", D, 4 * D),
                   ("mlp c_proj dgrad", 4 * D, D), ("c_fc dgrad", D, 4 * D), ("attn c_proj dgrad", D, D), ("c_attn dgrad", D, 3 * D)):
    us = nt(M, N, K)
    per_layer += us
    rows.append((name, us, 3 * 2.0 * M * N * K / us / 1e6))
# the block's four weight gradients as one grouped launch over 3 M "tokens"
K3 = 3 * M
mk = lambda n: t(K3, n)
shapes = [(mk(D), mk(4 * D), D, 4 * D), (mk(4 * D), mk(D), 4 * D, D), (mk(D), mk(D), D, D), (mk(D), mk(3 * D), D, 3 * D)]
Cs = [torch.empty(a, b, device=dev) for (_, _, a, b) in shapes]
S = 2
_, nws, ncnt = hip.wgrad_group_sizes([(a, b) for (_, _, a, b) in shapes], S, 0)
ws, cnt = torch.empty(nws, device=dev), torch.zeros(ncnt, dtype=torch.int32, device=dev)
probs = [(A, B, C_, a, b) for (A, B, a, b), C_ in zip(shapes, Cs)]
us = timeit(lambda: hip.wgrad_group(probs, K3, S, ws, cnt))
per_layer += us
rows.append(("block weight gradients (grouped)", us, 3 * 2.0 * M * 12 * D * D / us / 1e6))
del shapes, Cs, probs
head = nt(M, V, D) + nt(M, D, V)
A, B = t(3 * M, V), t(3 * M, D)
Cw = torch.empty(V, D, device=dev)
_, nws, ncnt = hip.wgrad_group_sizes([(V, D)], 1, 0)
ws, cnt = torch.empty(max(nws, 1), device=dev), torch.zeros(ncnt, dtype=torch.int32, device=dev)
head += timeit(lambda: hip.wgrad_group([(A, B, Cw, V, D)], 3 * M, 1, ws, cnt))
for name, us, tf in rows:
    print("%-36s %8.1f us   %6.0f TFLOP/s of bf16 work" % (name, us, tf))
gemm_ms = (L * per_layer + head) / 1e3
# operand splits: every product's activation operand (and the weight gradients' two), fp32 -> bf16 triple: 10 bytes per element
# per block: forward A operands (ln_1 out, ctx, ln_2 out, GELU out) 7 D, dgrad A operands (d out, d u, d attn-proj out, d qkv) 9 D, and
# the weight gradients' eight operands split along the TOKEN dimension (a different image: [hi; hi; lo] rows) 16 D; head: h_f twice, d logits twice
elems = L * M * (7 * D + 9 * D + 16 * D) + 2 * M * D + 2 * M * V
split_ms = elems * 10 / 5e12 * 1e3
other_ms = 14.2      # the f32 mode's non-GEMM time (fp32 attention 10.1, LayerNorm 2.0, loss 0.5, AdamW 0.7, misc 0.9)
step = gemm_ms + split_ms + other_ms
print("per block %.1f us, head %.1f us -> products %.2f ms per step; operand splits (%.1f G elements x 10 B at 5 TB/s) %.2f ms; fp32 non-GEMM %.1f ms" % (per_layer, head, gemm_ms, elems / 1e9, split_ms, other_ms))
print("projected bf16x3 step %.1f ms = %.0f train tokens/s; f32 mode 103.0 ms = 146 600 tokens/s -> %.2fx (adoption bar: 2.5x)" % (step, M / step * 1e3, 103.0 / step))
