#!/usr/bin/env python3
"""Where a training run first turns non-finite: the memorisation run of tools/train_curve.py (full configuration, dropout on), checking
the loss and every gradient tensor each step; prints the first offending tensors.   MODE=..., PDROP=..., python tools/nan_hunt.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.trainer import MMTGTrainer

S, V, B = 5, 13317, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mode = os.environ.get("MODE", "bf16x3f")
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
if os.environ.get("PDROP"):
    pd = float(os.environ["PDROP"])
    gcfg.update(embd_pdrop=pd, attn_pdrop=pd, resid_pdrop=pd)
model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype=mode, token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to("cuda").train()
batches = [{k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_batch(B, mcfg, dcfg, V, seed=40 + i).items()} for i in range(4)]
tr = MMTGTrainer(model, lr=2e-4, alpha=0.2, warmup_steps=12, total_steps=3000)
eng = model.engine()
for i in range(steps):
    out = tr.step(batches[i % 4], stage=3)
    loss, gn = float(out["loss"]), float(tr.grad_norm())
    bad = not (np.isfinite(loss) and np.isfinite(gn))
    if i % 20 == 0 or bad:
        print("[%s pdrop %s] step %3d loss %.4f kl %.5f grad norm %.4f" % (mode, os.environ.get("PDROP", "0.1"), i, loss, float(out["kl"]), gn), flush=True)
    if bad:
        g = eng.grad
        n = 0
        for k, (off, shape, cnt) in eng.layout.entries.items():
            t = g[off:off + cnt]
            if not bool(torch.isfinite(t).all()):
                print("   non-finite gradient:", k, "nan", int(torch.isnan(t).sum()), "inf", int(torch.isinf(t).sum()))
                n += 1
                if n > 12:
                    break
        print("   master finite:", bool(torch.isfinite(eng.master).all()))
        break
else:
    print("[%s] %d steps finite" % (mode, steps))
