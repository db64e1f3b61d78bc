"""Calibration only (not a product path): what the vendor library's bf16 kernels reach on this model's product shapes.
Run under rocprofv3 --kernel-trace --stats to see which macro tiles it picks (kernel names) next to the HIP-event times."""
import torch
shapes = [("fc1 fwd NT", 15104, 3072, 768, "nt"), ("fc2 fwd NT", 15104, 768, 3072, "nt"), ("qkv fwd NT", 15104, 2304, 768, "nt"),
          ("proj fwd NT", 15104, 768, 768, "nt"), ("lm head NT", 15104, 13440, 768, "nt"), ("dgrad lm head NN", 15104, 768, 13440, "nn"),
          ("wgrad fc TN", 3072, 768, 15104, "tn"), ("wgrad qkv TN", 768, 2304, 15104, "tn"), ("wgrad proj TN", 768, 768, 15104, "tn"),
          ("square 8192", 8192, 8192, 8192, "nt"), ("square 4096", 4096, 4096, 4096, "nt")]
dev = "cuda"
for name, M, N, K, lay in shapes:
    if lay == "nt":
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a, b.t())
    elif lay == "nn":
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a, b)
    else:
        a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a.t(), b)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    print("%-20s M=%6d N=%6d K=%6d  %8.1f us  %7.1f TF/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
