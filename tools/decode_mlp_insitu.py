#!/usr/bin/env python3
"""Round 6: the one-launch decode MLP (mmtg_decode_mlp) INSIDE the token step: its in-kernel timeline (12 real-time stamps per
workgroup, tools/decode_mlp_timeline.py prints the stand-alone one) taken from the last block's launch of eager token steps at
batch 256, 12 layers -- which stage costs more between attn.c_proj and the next block's c_attn than between two MLP launches.

    python tools/decode_mlp_insitu.py [plain|sc1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MMTG_DECODE_MLP"] = "1"
os.environ["MMTG_DECODE_MLP_HANDOFF"] = sys.argv[1] if len(sys.argv) > 1 else "sc1"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmtg_amd import MMTG, synth  # noqa: E402
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs  # noqa: E402
from mmtg_amd.decode import GreedyDecoder  # noqa: E402
from decode_mlp_timeline import STAGES  # noqa: E402

S, V, B, Ln = 5, 13317, 256, 128
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, gpt2_config=gcfg, compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to("cuda").eval()
nb = synth.make_batch(B, mcfg, dcfg, V, seed=7)
batch = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in nb.items() if k not in ("rating", "targets")}
dec = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=False)
assert dec.mlp
dec.generate(batch, 8, temperature=1.1, repitition_penalty=1.5)
n = dec.begin(batch, Ln, temperature=1.1, repitition_penalty=1.5)
tr = torch.zeros(256 * 12, dtype=torch.int64, device="cuda")
rows = []
for pos in range(dec.first_pos, dec.first_pos + 40):
    dec.mlp_trace = tr if pos >= dec.first_pos + 30 else None
    dec.step_at(pos)
    if dec.mlp_trace is not None:
        torch.cuda.synchronize()
        rows.append(tr.view(256, 12).double().cpu() / 100.0)
dec.check_mlp_error()
acc = None
for t in rows:
    d = torch.stack([t[:, i + 1] - t[:, i] for i in range(9)] + [t[:, 9] - t[:, 0], (t[:, 9].max() - t[:, 0].min()).expand(256)], 1)
    acc = d if acc is None else acc + d
acc /= len(rows)
print("in-step timeline of the last block's mmtg_decode_mlp, %s hand-off, mean of %d token steps (us, median / p90 / max over workgroups):"
      % (os.environ["MMTG_DECODE_MLP_HANDOFF"], len(rows)))
for i, name in enumerate(STAGES + ["workgroup lifetime", "first start -> last end"]):
    c = acc[:, i]
    print("  %-52s %6.2f / %6.2f / %6.2f" % (name, c.median(), c.quantile(0.9), c.max()))
