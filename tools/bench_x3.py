#!/usr/bin/env python3
"""Train step of a compute mode (bf16x3 / f32 / bf16) at the full configuration: ms per step + the per-category launch profile.
    python3 tools/bench_x3.py [mode] [rows] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from mmtg_amd import MMTG, hip, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.trainer import MMTGTrainer

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
V = 13317
mcfg, dcfg = make_model_cfgs(seq_len=5), data_config(seq_len=5)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
dev = torch.device("cuda:0")
model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype=mode, token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).train()
tr = MMTGTrainer(model, lr=1e-5, alpha=0.2, warmup_steps=10, total_steps=100000)
batches = [{k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.make_batch(B, mcfg, dcfg, V, seed=i).items()} for i in range(2)]
T = dcfg.topic_prompt_length + batches[0]["targets"].shape[1]
for i in range(3):
    out = tr.step(batches[i % 2], stage=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    out = tr.step(batches[i % 2], stage=3)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("%s rows %d T %d: %.3f ms/step, %.0f tokens/s, loss %.4f" % (mode, B, T, 1e3 * el / steps, B * T * steps / el, float(out["loss"])))
hip.prof_enable(True)
for i in range(2):
    tr.step(batches[i % 2], stage=3)
torch.cuda.synchronize()
hip.prof_enable(False)
for k, v in hip.prof_read().items():
    if v["launches"]:
        print("  %-10s %4d launches %8.3f ms/step  %7.1f TFLOP/s  %6.2f TB/s" % (k, v["launches"] // 2, v["ms"] / 2,
              v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0, v["bytes"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0))
