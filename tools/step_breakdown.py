#!/usr/bin/env python3
"""In-situ time of every C-ABI call of one full-config training step (torch events around each
binding call; GEMMs keyed by layout / shape / epilogue).  Answers "which launches does the step
actually spend its time in" -- isolated per-shape timings (tools/bench_gemm.py) run warm and
back-to-back and are optimistic.

  python tools/step_breakdown.py [steps]
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmtg_amd import MMTG, hip, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.trainer import MMTGTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
S, V, B = 5, 13317, 64
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
torch.manual_seed(0)
model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).train()
trainer = MMTGTrainer(model, lr=1e-5, alpha=0.2, warmup_steps=10, total_steps=100000, distributed=False)
batches = []
for i in range(2):
    nb = synth.make_batch(B, mcfg, dcfg, V, seed=i)
    batches.append({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()})
for i in range(4):
    trainer.step(batches[i % 2], stage=3)
torch.cuda.synchronize()

log = []
EPI = ["none", "gelu", "tanh", "resid", "dgelu", "dtanh", "atomic", "rowdot", "slab"]


def wrap(name, fn):
    def f(*a, **k):
        if name == "gemm":
            M, N, K = a[3:6]
            key = "gemm %s%s M=%d N=%d K=%d %s s=%d" % ("T" if k.get("transA") else "N", "T" if k.get("transB") else "N", M, N, K,
                                                      EPI[k.get("epi", 0)], k.get("splits", 1))
        else:
            key = name
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        log.append((key, e0, e1))
        return r
    return f


skip = {"lib", "lib_path", "exported_symbols", "dt", "torch_dtype", "drop_thresh", "prof_enable", "prof_read", "gemm_trace"}
for name in dir(hip):
    fn = getattr(hip, name)
    if callable(fn) and not isinstance(fn, type) and not name.startswith("_") and name not in skip and getattr(fn, "__module__", "") == hip.__name__:
        setattr(hip, name, wrap(name, fn))

t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(steps):
    trainer.step(batches[i % 2], stage=3)
t1.record()
torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in log:
    tot[key][0] += 1
    tot[key][1] += e0.elapsed_time(e1)
wall = t0.elapsed_time(t1) / steps
print("instrumented step: %.3f ms; sum of bracketed calls %.3f ms" % (wall, sum(v[1] for v in tot.values()) / steps))
print("%-58s %6s %9s %9s" % ("call", "n/step", "us/call", "ms/step"))
for key, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-58s %6.1f %9.1f %9.3f" % (key, n / steps, 1e3 * ms / n, ms / steps))
