import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from mmtg_amd import MMTG, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.trainer import MMTGTrainer
S, V, B = 5, 13317, 64
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0); model.to("cuda"); model.train()
tr = MMTGTrainer(model, lr=1e-5, alpha=0.2)
b = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_batch(B, mcfg, dcfg, V, seed=4).items()}
for _ in range(5): tr.step(b, stage=3)
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(b, stage=3); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print("enqueue (host) %.2f ms per step; step incl. GPU %.2f ms" % (1e3 * np.median(enq), 1e3 * np.median(tot)))
