#!/usr/bin/env python3
"""Per-shape timing of the GEMMs of one full-config training step (HIP events, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmtg_amd import hip

dev = "cuda"
M = int(os.environ.get("TOKENS", 64 * 236))
D, V = int(os.environ.get("DMODEL", "768")), 13440
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32

def t(*s): return (torch.randn(*s, device=dev) * 0.5).to(dt)

COLD = bool(os.environ.get("COLD"))
_flush = torch.empty(1 << 28, device=dev, dtype=torch.float32) if COLD else None   # 1 GiB > L2 + MALL

def timeit(fn, n=20):
    if COLD:   # every launch behind a 1 GiB fill: operands come from HBM, as inside a training step
        tot = 0.0
        for _ in range(8):
            _flush.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        return tot / 8 * 1e3
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us

rows = []
FLAGS = int(os.environ.get("FLAGS", "0"))
SLAB = int(os.environ.get("SLAB", "0"))   # weight gradients as K-split slabs + slab_sum (2: time the GEMM alone)
def case(name, Mm, N, K, tA, tB, epi=hip.EPI_NONE, splits=1, out_f32=False):
    A = t(K, Mm) if tA else t(Mm, K)
    B = t(N, K) if tB else t(K, N)
    if SLAB and epi == hip.EPI_ATOMIC:
        part = torch.empty(splits, Mm, N, device=dev, dtype=torch.float32)
        C = torch.zeros(Mm, N, device=dev, dtype=torch.float32)
        def fn():
            hip.gemm(A, B, part, Mm, N, K, transA=True, transB=False, epi=hip.EPI_SPLIT, splits=splits, out_f32=True, flags=FLAGS)
            if SLAB == 1: hip.slab_sum(part, splits, Mm * N, C, Mm * N, accumulate=True)
        us = timeit(fn)
        tf = 2.0 * Mm * N * K / us / 1e6
        print("%-28s M=%6d N=%6d K=%6d %9.1f us %8.1f TF/s" % (name + " slab", Mm, N, K, us, tf), flush=True)
        return
    C = torch.zeros(Mm, N, device=dev, dtype=torch.float32 if (out_f32 or epi == hip.EPI_ATOMIC) else dt)
    kw = {}
    if epi == hip.EPI_GELU: kw["aux2"] = torch.empty(Mm, N, device=dev, dtype=dt)
    if epi in (hip.EPI_RESID, hip.EPI_DGELU): kw["aux"] = t(Mm, N)
    bias = None if epi in (hip.EPI_ATOMIC, hip.EPI_DGELU) else torch.zeros(N, device=dev)
    us = timeit(lambda: hip.gemm(A, B, C, Mm, N, K, transA=tA, transB=tB, epi=epi, splits=splits, out_f32=out_f32, bias=bias, flags=FLAGS, **kw))
    tf = 2.0 * Mm * N * K / us / 1e6
    rows.append((name, Mm, N, K, us, tf))
    print("%-28s M=%6d N=%6d K=%6d %9.1f us %8.1f TF/s" % (name, Mm, N, K, us, tf), flush=True)

from mmtg_amd.engine import _wgrad_splits as ws
if os.environ.get("PMC"):
    # one launch per layout at comparable work, for rocprofv3 --pmc
    timeit_n = 2
    def once(name, Mm, N, K, tA, tB, epi=hip.EPI_NONE, splits=1):
        A = t(K, Mm) if tA else t(Mm, K)
        B = t(N, K) if tB else t(K, N)
        C = torch.zeros(Mm, N, device=dev, dtype=torch.float32 if epi == hip.EPI_ATOMIC else dt)
        for _ in range(2):
            hip.gemm(A, B, C, Mm, N, K, transA=tA, transB=tB, epi=epi, splits=splits, flags=FLAGS)
        torch.cuda.synchronize()
    once("NT", M, D, 4 * D, False, True)
    once("NN", M, D, 4 * D, False, False)
    once("TN", 4 * D, D, M, True, False, hip.EPI_ATOMIC, 3)
    once("TN fc1", D, 4 * D, M, True, False, hip.EPI_ATOMIC, 3)
    FLAGS = 256   # tile_n-fastest item order
    once("TN fc1 row order", D, 4 * D, M, True, False, hip.EPI_ATOMIC, 3)
    sys.exit(0)
if os.environ.get("TNSWEEP"):
    for K in (64, 256, 1024, 4096, 15104):
        case("TN 3072x768 s=1 K-sweep", 4 * D, D, K, True, False, hip.EPI_ATOMIC, 1)
    wide = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12) if os.environ.get("SWEEPWIDE") else None
    for s_ in wide or (1, 2, 3, 4, 6, 8):
        case("TN 3072x768 K=15104 s=%d" % s_, 4 * D, D, M, True, False, hip.EPI_ATOMIC, s_)
    for s_ in wide or (1, 2, 3, 4):
        case("TN 13440x768 K=15104 s=%d" % s_, V, D, M, True, False, hip.EPI_ATOMIC, s_)
    for s_ in wide or (3, 4, 5, 8):
        case("TN 768x2304 K=15104 s=%d" % s_, D, 3 * D, M, True, False, hip.EPI_ATOMIC, s_)
    for s_ in wide or ():
        case("TN 768x768 K=15104 s=%d" % s_, D, D, M, True, False, hip.EPI_ATOMIC, s_)
    sys.exit(0)
if os.environ.get("KSWEEP"):
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        case("NT N=3072 K-sweep", M, 4 * D, K, False, True)
    for K in (64, 256, 768, 3072):
        case("NT N=768 K-sweep", M, D, K, False, True)
    for K in (64, 768):
        case("NT gelu N=3072 K-sweep", M, 4 * D, K, False, True, hip.EPI_GELU)
    sys.exit(0)
if os.environ.get("NTSET"):
    # the forward / dgrad products as the trainer launches them (NT through the [out,in] weight copies) + calibration squares
    case("NT qkv plain", M, 3 * D, D, False, True)
    case("NT fc1 plain", M, 4 * D, D, False, True)
    case("NT fc1 gelu", M, 4 * D, D, False, True, hip.EPI_GELU)
    case("NT fc2-dgrad dgelu", M, 4 * D, D, False, True, hip.EPI_DGELU)
    case("NT fc2 resid K=3072", M, D, 4 * D, False, True, hip.EPI_RESID)
    case("NT fc1-dgrad K=3072", M, D, 4 * D, False, True)
    case("NT qkv-dgrad K=2304", M, D, 3 * D, False, True)
    case("NT proj resid", M, D, D, False, True, hip.EPI_RESID)
    case("NT lm head", M, V, D, False, True)
    case("NT square 4096", 4096, 4096, 4096, False, True)
    case("NT square 8192", 8192, 8192, 8192, False, True)
    case("NT 15104x3072x3072", M, 4 * D, 4 * D, False, True)
    sys.exit(0)
case("fwd qkv (NN)", M, 3 * D, D, False, False)
case("fwd attn proj (NN,resid)", M, D, D, False, False, hip.EPI_RESID)
case("fwd fc1 (NN,gelu)", M, 4 * D, D, False, False, hip.EPI_GELU)
case("fwd fc2 (NN,resid)", M, D, 4 * D, False, False, hip.EPI_RESID)
case("fwd lm head (NT,f32 out)", M, V, D, False, True, out_f32=True)
case("fwd proj1 (NT)", M, 512, 2048, False, True, hip.EPI_TANH)
case("dgrad fc2 (NT,dgelu)", M, 4 * D, D, False, True, hip.EPI_DGELU)
case("dgrad fc1 (NT)", M, D, 4 * D, False, True)
case("dgrad proj (NT)", M, D, D, False, True)
case("dgrad qkv (NT)", M, D, 3 * D, False, True)
case("dgrad lm head (NN)", M, D, V, False, False)
if os.environ.get("TNSET"):
    # weight gradients as the trainer launches them: K-split slabs (+ the ordered slab sum), split count per kernel family
    from mmtg_amd.engine import _wgrad_splits_p8 as ws8
    SLAB = int(os.environ.get("SLAB", "1"))
    for nm, a, b in (("wgrad fc2", 4 * D, D), ("wgrad fc1", D, 4 * D), ("wgrad proj", D, D), ("wgrad qkv", D, 3 * D)):
        sp = (ws8(a, b, M) if os.environ.get("MMTG_GEMM_P8T", "0") != "0" else None) or ws(a, b, M, True)
        case(nm + " (TN s=%d)" % sp, a, b, M, True, False, hip.EPI_ATOMIC, sp)
    sys.exit(0)
for nm, a, b in (("wgrad fc2", 4 * D, D), ("wgrad fc1", D, 4 * D), ("wgrad proj", D, D), ("wgrad qkv", D, 3 * D), ("wgrad lm head", V, D), ("wgrad proj1", 512, 2048)):
    occ4 = not (FLAGS & hip.GEMM_NO_OCC4)
    case(nm + " (TN,atomic s=%d)" % ws(a, b, M, occ4), a, b, M, True, False, hip.EPI_ATOMIC, ws(a, b, M, occ4))
    case(nm + " (TN,atomic s=1)", a, b, M, True, False, hip.EPI_ATOMIC, 1)
tot = sum(r[4] for r in rows)
print("sum %.1f us" % tot)
