#!/usr/bin/env python3
"""Per-shape timing of the split-precision products of a training step (M = 15104 rows), with the output variants the engine uses,
beside the bf16 product of the same shape at K' = 3K (the loop the x3 kernel runs, with a bf16 epilogue): what the fp32 / plane-pair
epilogues cost on top of the three passes.
    python3 tools/gemm_x3_shapes.py [rows]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mmtg_amd import hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 15104
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


# name, N, K, epilogue, outputs ("c" fp32, "p" planes), aux
cases = [("c_attn fwd", 2304, 768, hip.EPI_NONE, "c", False),
         ("attn.c_proj fwd (+resid)", 768, 768, hip.EPI_RESID, "c", True),
         ("c_fc fwd (GELU, u + planes)", 3072, 768, hip.EPI_GELU, "p", True),
         ("mlp.c_proj fwd (+resid)", 768, 3072, hip.EPI_RESID, "c", True),
         ("d mlp.c_proj (dGELU, planes)", 3072, 768, hip.EPI_DGELU, "p", True),
         ("d c_fc", 768, 3072, hip.EPI_NONE, "c", False),
         ("d attn.c_proj (planes + c)", 768, 768, hip.EPI_NONE, "cp", False),
         ("d c_attn", 768, 2304, hip.EPI_NONE, "c", False)]
print("rows %d" % M)
for name, N, K, epi, outs, has_aux in cases:
    A32 = torch.randn(M, K, device=dev, generator=g) * 0.5
    B32 = torch.randn(N, K, device=dev, generator=g) * 0.05
    A, B = hip.Planes.empty(M, K, dev), hip.Planes.empty(N, K, dev)
    hip.split_planes(A32, M, K, A)
    hip.split_planes(B32, N, K, B)
    bias = torch.zeros(N, device=dev)
    C = torch.empty(M, N, device=dev)
    P = hip.Planes.empty(M, N, dev)
    aux = torch.randn(M, N, device=dev, generator=g) if has_aux else None
    res = {}
    variants = {"as used": outs, "fp32 C only": "c", "planes only": "p"}
    for vn, o in variants.items():
        if epi == hip.EPI_GELU and o == "c":
            continue
        kw = dict(aux2=aux) if epi == hip.EPI_GELU else dict(aux=aux)      # GELU: aux2 receives the fp32 pre-activation
        res[vn] = timed(lambda: hip.gemm_x3(A, B, C if "c" in o else None, M, N, K, planes=P if "p" in o else None, bias=bias, epi=epi, **kw))
    # the bf16 product of the same shape at K' = 3K, plain epilogue
    A3 = torch.randn(M, 3 * K, device=dev, generator=g).bfloat16()
    B3 = torch.randn(N, 3 * K, device=dev, generator=g).bfloat16()
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_b = timed(lambda: hip.gemm(A3, B3, Cb, M, N, 3 * K, transA=False, transB=True))
    fl = 2.0 * M * N * K * 3
    print("%-32s N %4d K %4d: %s | bf16 K'=3K %.1f us (%.0f TF)" % (
        name, N, K, "  ".join("%s %.1f us (%.0f TF)" % (k, v, fl / v / 1e6) for k, v in res.items()), t_b, fl / t_b / 1e6))
