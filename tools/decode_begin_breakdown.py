#!/usr/bin/env python3
"""What a generation spends before its first token step (GreedyDecoder.begin, batch 256): wall time (host + GPU, synchronised) of the
weight-copy refresh, the LayerNorm folds, the host-side state reset, the encoder + prompt prefill forward and the K / V copies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.decode import GreedyDecoder

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S, V, B, Ln = 5, 13317, 256, 128
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, gpt2_config=gcfg, compute_dtype=mode, token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to("cuda").eval()
nb = synth.make_batch(B, mcfg, dcfg, V, seed=7)
batch = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in nb.items() if k not in ("rating", "targets")}
dec = GreedyDecoder(model, max_batch=B, max_len=Ln)
dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
eng = dec.eng


def timed(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def refresh():
    eng.invalidate_copies()
    eng.refresh_copies()


print("[%s] begin() whole                        %.3f ms" % (mode, timed(lambda: dec.begin(batch, Ln, temperature=1.1, repitition_penalty=1.5))))
print("  weight copies (cast + transposes)      %.3f ms" % timed(refresh))
print("  LayerNorm folds (25 launches)          %.3f ms" % timed(dec._refresh_folds))
print("  prompt prefill (encoder + 12 blocks over %d rows + K / V copies)  %.3f ms" % (B * (eng.sh.P + 1), timed(lambda: dec._prefill(batch))))
pb = dict(batch)
pb["targets"] = dec.seq[:, eng.sh.P:eng.sh.P + 1]
print("    of it the engine forward            %.3f ms" % timed(lambda: eng.forward(pb, train_flag=False, training=False, per_row_infer=True, need_logits=False)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    eng.forward(pb, train_flag=False, training=False, per_row_infer=True, need_logits=False)
e1.record()
torch.cuda.synchronize()
print("    the same by GPU events (back to back) %.3f ms" % (e0.elapsed_time(e1) / 5))
