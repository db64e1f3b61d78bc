#!/usr/bin/env python3
"""Debug aid: persistent vs per-launch fused decode step, logits of every model call compared bit for bit (teacher-forced on the
per-launch decoder's ids so that the prefixes stay equal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.decode import GreedyDecoder
dev = "cuda"
S, V = 5, 13317
B, Ln, L = int(os.environ.get("B", "256")), int(os.environ.get("LEN", "100")), int(os.environ.get("LAYERS", "12"))
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
model = MMTG(mcfg, dcfg, V, gpt2_config=gpt2_config(n_layer=L, vocab_size=V), compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).eval()
nb = synth.make_batch(B, mcfg, dcfg, V, seed=7)
tb = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items() if k not in ("rating", "targets")}
recs = {}
ref_ids = None
for persist in ("0", "1", "1"):
    os.environ["MMTG_DECODE_PERSIST"] = persist
    dec = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=bool(int(os.environ.get("GRAPH", "1"))))
    rec = []
    def tap(j, with_head, picked, logits, rec=rec):
        if with_head:
            rec.append((j, logits.clone(), dec.h.clone(), dec.st[0].clone()))
    ids = dec.generate(tb, Ln, temperature=1.1, repitition_penalty=1.5, tap=tap, teacher=ref_ids)
    if ref_ids is None:
        ref_ids = ids.clone()
        ref = rec
        continue
    bad = 0
    for (j, lg, h, st), (j2, lg2, h2, st2) in zip(ref, rec):
        d = (lg - lg2).abs()
        if float(d.max()) != 0.0:
            rows = torch.nonzero(d.amax(1) > 0).flatten().tolist()
            print("call j=%d: logits differ, max %.4g, rows %s%s" % (j, float(d.max()), rows[:12], "..." if len(rows) > 12 else ""))
            bad += 1
            if bad >= 8:
                break
    print("persist=%s: %d model calls compared, %d differ; err flag %d" % (persist, len(rec), bad, int(dec.err.item()) if hasattr(dec, "err") else -1), flush=True)
    del dec
