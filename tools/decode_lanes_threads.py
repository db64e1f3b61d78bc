#!/usr/bin/env python3
"""Decode: the rows of a batch never interact -- do several INDEPENDENT decoders (row blocks), each replaying its own captured
token step on its own stream from its own host thread, overlap on the GPU?  (Round 2 measured fork / join branches INSIDE one
captured graph: the runtime ran them one after another.  Separate graphs on separate streams are a different object.)

    python tools/decode_lanes_threads.py [--batch 256] [--len 128] [--lanes 1,2,4]
Prints microseconds per token step of the whole batch and tokens/s per lane count, and checks that the ids do not depend on it.
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.decode import GreedyDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--len", type=int, default=128)
ap.add_argument("--lanes", default="1,2,4")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = "cuda"
S, V, B, Ln = 5, 13317, a.batch, a.len
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, gpt2_config=gcfg, compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).eval()
nb = synth.make_batch(B, mcfg, dcfg, V, seed=7)
batch = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items() if k not in ("rating", "targets")}
ref = None
for lanes in (int(x) for x in a.lanes.split(",")):
    rows = B // lanes
    decs = [GreedyDecoder(model, max_batch=rows, max_len=Ln, lanes=1) for _ in range(lanes)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    parts = [{k: v[i * rows:(i + 1) * rows] for k, v in batch.items()} for i in range(lanes)]
    for d, s, p in zip(decs, streams, parts):          # warm-up: captures every lane's graphs on its own stream
        with torch.cuda.stream(s):
            d.generate(p, Ln, temperature=1.1, repitition_penalty=1.5)
    torch.cuda.synchronize()
    best = None
    for rep in range(a.reps):
        n_steps = 0
        for d, s, p in zip(decs, streams, parts):      # once-per-generation work, lane after lane (shared engine workspaces)
            with torch.cuda.stream(s):
                n_steps = d.begin(p, Ln, temperature=1.1, repitition_penalty=1.5)
        torch.cuda.synchronize()

        def loop(d, s):
            with torch.cuda.stream(s):
                for pos in range(d.first_pos, n_steps):
                    d.step_at(pos)

        t0 = time.perf_counter()
        th = [threading.Thread(target=loop, args=(d, s)) for d, s in zip(decs, streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    ids = torch.cat([d.seq[:, 15:15 + 1 + Ln] for d in decs], 0).cpu()
    if ref is None:
        ref = ids
    same = bool(torch.equal(ids, ref))
    print("lanes %d x %3d rows: %8.1f us per token step of the whole batch, %9.0f tokens/s, ids equal to 1 lane: %s"
          % (lanes, rows, 1e6 * best / n_steps, B * Ln / best, same), flush=True)
    del decs
    torch.cuda.empty_cache()
