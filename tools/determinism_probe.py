"""Two trainers from identical state, same batch, same dropout seed: which gradient tensors differ bit for bit after one
backward?  (tests/test_ddp_gpu.py relies on the GPT-2 block matrices being reproducible.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ddp_worker import build as build_small


def build(dtype, pdrop, dev):
    """FULL=1: the benchmarked configuration (12 layers, V = 13317); otherwise the 2-layer model of tests/ddp_worker.py."""
    if not os.environ.get("FULL"):
        return build_small(dtype, pdrop, dev)
    from mmtg_amd import MMTG, synth
    from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
    S, V = 5, 13317
    mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
    gcfg = gpt2_config(n_layer=12, vocab_size=V, embd_pdrop=pdrop, attn_pdrop=pdrop, resid_pdrop=pdrop)
    model = MMTG(mcfg, dcfg, V, gpt2_config=gcfg, compute_dtype=dtype, token_table=synth.make_token_table(V, seed=2))
    model.reset_parameters(seed=0)
    model.to(dev).train()
    return model, mcfg, dcfg, V

from mmtg_amd import synth
from mmtg_amd.trainer import MMTGTrainer
dev = torch.device("cuda", 0)
rows = int(os.environ.get("ROWS", "64" if os.environ.get("FULL") else "12"))
for trial in range(int(os.environ.get("TRIALS", "8"))):
    grads = []
    for rep in range(2):
        model, mcfg, dcfg, V = build(os.environ.get("DTYPE", "bf16"), 0.1, dev)
        tr = MMTGTrainer(model, lr=0.0, alpha=0.2, distributed=False)
        tr.eng.drop_seed = 4242
        nb = synth.make_batch(rows, mcfg, dcfg, V, seed=7)
        batch = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()}
        if rep:   # perturb allocator / cache history between the two runs
            junk = torch.randn(64 << 20, device=dev)
        tr.step(batch, stage=int(os.environ.get("STAGE", "1")))
        torch.cuda.synchronize()
        grads.append(tr.eng.grad.detach().cpu().clone())
        lay = model.layout
    bad = []
    for k, (off, shape, n) in lay.entries.items():
        a, b = grads[0][off:off + n], grads[1][off:off + n]
        if not torch.equal(a, b):
            bad.append((k, float((a - b).abs().max()), float(a.abs().max()), int((a != b).sum()), n))
    mats = [x for x in bad if ".h." in x[0] and x[0].endswith(".weight") and ".ln_" not in x[0]]
    print("trial", trial, "differing tensors:", len(bad), "of which GPT-2 block matrices:", len(mats))
    for x in mats:
        print("   ", x)
    if os.environ.get("NAMES") and trial == 0:
        for x in bad:
            print("    differs:", x[0], "max|d| %.3g of %.3g, %d / %d elements" % x[1:])
