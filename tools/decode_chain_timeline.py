#!/usr/bin/env python3
"""Per-workgroup timeline of one chained decode launch (attn.c_proj -> c_fc -> mlp.c_proj of a block; mmtg_decode_persist_trace):
entry, wait passed, exit per stage; eager launches, full 12-layer model, batch 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mmtg_amd import MMTG, hip, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.decode import GreedyDecoder
dev = "cuda"
S, V, B, Ln = 5, 13317, int(os.environ.get("B", "256")), int(os.environ.get("LEN", "60"))
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
model = MMTG(mcfg, dcfg, V, gpt2_config=gpt2_config(n_layer=12, vocab_size=V), compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).eval()
nb = synth.make_batch(B, mcfg, dcfg, V, seed=7)
tb = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items() if k not in ("rating", "targets")}
dec = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=False)
assert dec.chain, "chained launches are off: run with MMTG_DECODE_CHAIN=1"
dec.generate(tb, Ln, temperature=1.1, repitition_penalty=1.5)
n = dec.begin(tb, Ln, temperature=1.1, repitition_penalty=1.5)
nwg = 576
buf = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
for pos in range(n):
    last = pos == n - 2
    if last:
        torch.cuda.synchronize()
        hip.decode_persist_trace(buf)          # every chained launch of the step writes the same rows: the LAST block's survive
    dec.step_at(pos)
    if last:
        torch.cuda.synchronize()
        hip.decode_persist_trace(None)
        r = buf.cpu().numpy().reshape(nwg, 4).astype(np.float64)
        t0 = r[:, 1].min()
        us = lambda x: (x - t0) * 0.01
        print("chained launch of the last block at prefix length %d, batch %d: first entry -> last exit %.1f us" % (pos + 1, B, us(r[:, 3].max())))
        for st, nm in enumerate(("attn c_proj (reduce)", "c_fc + GELU", "mlp c_proj (reduce)")):
            q = r[r[:, 0] == st]
            print("  %-22s %3d workgroups: entry %5.1f..%5.1f  wait passed %5.1f..%5.1f (median %5.1f)  exit %5.1f..%5.1f (median %5.1f) us" % (
                nm, len(q), us(q[:, 1].min()), us(q[:, 1].max()), us(q[:, 2].min()), us(q[:, 2].max()), np.median(us(q[:, 2])),
                us(q[:, 3].min()), us(q[:, 3].max()), np.median(us(q[:, 3]))))
