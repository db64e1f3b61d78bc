#!/usr/bin/env python3
"""Where the persistent decode token step spends its time: per stage, workgroup 0's own work and the wait at the barrier behind it
(mmtg_decode_persist_trace), at a given prefix length; eager launches, full 12-layer model, batch 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mmtg_amd import MMTG, hip, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.decode import GreedyDecoder
dev = "cuda"
S, V, B, Ln = 5, 13317, int(os.environ.get("B", "256")), int(os.environ.get("LEN", "100"))
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
model = MMTG(mcfg, dcfg, V, gpt2_config=gpt2_config(n_layer=12, vocab_size=V), compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).eval()
nb = synth.make_batch(B, mcfg, dcfg, V, seed=7)
tb = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items() if k not in ("rating", "targets")}
dec = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=False)
dec.generate(tb, Ln, temperature=1.1, repitition_penalty=1.5)
n = dec.begin(tb, Ln, temperature=1.1, repitition_penalty=1.5)
buf = torch.zeros(2 * 61 + 1, dtype=torch.int64, device=dev)
names = ["c_attn", "attention", "attn c_proj", "c_fc + GELU", "mlp c_proj"]
for pos in range(n):
    last = pos == n - 2
    if last:
        torch.cuda.synchronize()
        hip.decode_persist_trace(buf)
    wh = dec.step_at(pos)
    if last:
        torch.cuda.synchronize()
        hip.decode_persist_trace(None)
        t = buf.cpu().numpy().astype(np.float64) * 0.01        # us
        ns = 61 if wh else 60
        t0 = t[2 * ns]
        work = {k: [] for k in names + ["head"]}
        wait = {k: [] for k in names + ["head"]}
        prev = t0
        for s in range(ns):
            nm = "head" if s == 60 else names[s % 5]
            work[nm].append(t[2 * s] - prev)
            wait[nm].append(t[2 * s + 1] - t[2 * s])
            prev = t[2 * s + 1]
        print("persistent token step at prefix length %d, batch %d: %.1f us from kernel entry to the last barrier (workgroup 0)" % (pos + 1, B, prev - t0))
        for nm in names + ["head"]:
            if work[nm]:
                print("  %-12s x %2d   own work %6.2f us   barrier wait %6.2f us   stage %6.2f us" % (nm, len(work[nm]), np.mean(work[nm]), np.mean(wait[nm]), np.mean(work[nm]) + np.mean(wait[nm])))
