#!/usr/bin/env python3
"""Decode soak at the benchmarked shape (batch 256, 128 positions, greedy): N generations alternating between two prompt batches on ONE
decoder -- every generation of a batch must return the ids of that batch's first generation (graph replay, KV cache, prefill, LayerNorm
folds and the position slots carry nothing over from the other batch's generation in between).
    python tools/decode_soak.py [generations] [mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.decode import GreedyDecoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
S, V, B, Ln = 5, 13317, 256, 128
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, gpt2_config=gcfg, compute_dtype=mode, token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to("cuda").eval()
batches = []
for seed in (7, 8):
    nb = synth.make_batch(B, mcfg, dcfg, V, seed=seed)
    batches.append({k: torch.from_numpy(np.asarray(v)).cuda() for k, v in nb.items() if k not in ("rating", "targets")})
dec = GreedyDecoder(model, max_batch=B, max_len=Ln)
first = [None, None]
bad = 0
t0 = time.time()
for g in range(n):
    w = g & 1
    ids = dec.generate(batches[w], Ln, temperature=1.1, repitition_penalty=1.5).clone()
    if first[w] is None:
        first[w] = ids
    elif not torch.equal(ids, first[w]):
        bad += 1
        print("generation %d (batch %d): %d ids differ from that batch's first generation" % (g, w, int((ids != first[w]).sum())), flush=True)
torch.cuda.synchronize()
el = time.time() - t0
print("[%s] %d generations of %d x %d ids in %.1f s (%.0f tokens/s incl. the host checks); generations that differed: %d; "
      "the two batches' texts differ in %d of %d ids" % (mode, n, B, Ln, el, n * B * Ln / el, bad, int((first[0] != first[1]).sum()), first[0].numel()))
sys.exit(1 if bad else 0)
