"""Bit-equality of the eight-phase 256x256 kernel (MMTG_GEMM_BIG=2) with the default kernels: run once with MODE=ref
(default kernels, writes /tmp/p8_ref.pt), once with MMTG_GEMM_BIG=2 (compares).  Repeats every case to screen for races."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmtg_amd import hip
torch.manual_seed(0)
dev = "cuda"
cases = [(15104, 3072, 768, hip.EPI_NONE), (15104, 768, 3072, hip.EPI_RESID), (15104, 3072, 768, hip.EPI_GELU), (15104, 13440, 768, hip.EPI_NONE),
         (1100, 304, 256, hip.EPI_NONE), (2048, 512, 2048, hip.EPI_TANH), (4096, 4096, 4096, hip.EPI_NONE), (1024, 256, 128, hip.EPI_NONE),
         (15104, 768, 768, hip.EPI_DGELU), (1500, 1000, 384, hip.EPI_GELU)]
ref = {} if os.environ.get("MODE") == "ref" else torch.load("/tmp/p8_ref.pt")
bad = 0
for ci, (M, N, K, epi) in enumerate(cases):
    g = torch.Generator(device=dev); g.manual_seed(ci)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) if epi != hip.EPI_DGELU else None
    kw = {}
    if epi == hip.EPI_GELU: kw["aux2"] = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    if epi in (hip.EPI_RESID, hip.EPI_DGELU): kw["aux"] = (torch.randn(M, N, device=dev, generator=g)).bfloat16()
    outs = []
    for rep in range(1 if ref == {} or os.environ.get("MODE") == "ref" else 6):
        C = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
        if "aux2" in kw: kw["aux2"].fill_(3.0)
        hip.gemm(A, B, C, M, N, K, transB=True, epi=epi, bias=bias, **kw)
        torch.cuda.synchronize()
        outs.append((C.clone(), kw["aux2"].clone() if "aux2" in kw else None))
    if os.environ.get("MODE") == "ref":
        ref[ci] = (outs[0][0].cpu(), None if outs[0][1] is None else outs[0][1].cpu())
    else:
        for rep, (C, a2) in enumerate(outs):
            ok = torch.equal(C.cpu(), ref[ci][0]) and (a2 is None or torch.equal(a2.cpu(), ref[ci][1]))
            if not ok:
                bad += 1
                d = (C.cpu().float() - ref[ci][0].float()).abs()
                print("MISMATCH case", ci, (M, N, K, epi), "rep", rep, "max abs", float(d.max()), "n", int((d > 0).sum()), flush=True)
        print("case", ci, (M, N, K, epi), "ok" if not bad else "checked", flush=True)
if os.environ.get("MODE") == "ref":
    torch.save(ref, "/tmp/p8_ref.pt")
    print("reference outputs written")
else:
    print("p8 check:", "ALL BIT-EQUAL" if bad == 0 else "%d mismatches" % bad)
