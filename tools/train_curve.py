#!/usr/bin/env python3
"""Memorisation run at the full configuration (12L/768, V=13317, 64 rows x 236 positions, bf16, dropout 0.1 ON):
120 clip+AdamW steps over four fixed synthetic batches with the reference's schedule (linear warm-up, linear decay);
prints MyLoss / KL every 10 steps.  A training-works check of the whole fused step, not a benchmark.
    python tools/train_curve.py [steps]        MODE=bf16|bf16x3|bf16x3f|f32 picks the compute mode (default bf16)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.trainer import MMTGTrainer

S, V, B = 5, 13317, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
every = max(10, steps // 12)
# ENC="LSTM,2,RNN,2": image type / layers, text type / layers (the encoder channel variants of model.py:41-59; default: released GRUs)
enc = {}
if os.environ.get("ENC"):
    it, il, tt, tl = os.environ["ENC"].split(",")
    enc = dict(image_type=it, image_layers=int(il), text_type=tt, text_layers=int(tl))
mcfg, dcfg = make_model_cfgs(seq_len=S, **enc), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype=os.environ.get("MODE", "bf16"), token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to("cuda")
model.train()
batches = [{k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_batch(B, mcfg, dcfg, V, seed=40 + i).items()} for i in range(4)]
tr = MMTGTrainer(model, lr=2e-4, alpha=0.2, warmup_steps=12, total_steps=steps)
t0 = time.time()
for i in range(steps):
    out = tr.step(batches[i % 4], stage=3)
    if i % every == 0 or i == steps - 1:
        print("step %3d  lr %.2e  MyLoss %.4f  lm_loss %.4f  kl %.5f" % (i, tr.current_lr(), float(out["loss"]), float(out["lm_loss"]), float(out["kl"])), flush=True)
torch.cuda.synchronize()
print("[%s] %d steps in %.2f s; all parameters finite: %s" % (os.environ.get("MODE", "bf16"), steps, time.time() - t0, bool(torch.isfinite(model.engine().master).all())))
