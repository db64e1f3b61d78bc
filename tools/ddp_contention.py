#!/usr/bin/env python3
"""Compute-side price of the gradient exchange on ONE GPU (VERDICT r3 next #7a).

At world size > 1 the RCCL ring kernels hold CUs for the whole all-reduce beside the backward, and mmtg_amd.ddp tells the
GEMM tile rule to plan on 32 CUs fewer (mmtg_gemm_cu_budget(-32)).  That rule was a guess.  Here the full-size training step
runs on one GPU while a stand-in kernel on a side stream holds `--held` CU-sized slots (a workgroup that declares the whole
160 KB of LDS and sleeps: nothing else fits on its CU) for the length of a step, under GEMM budgets 0 / -16 / -32 / -48.
Prints ms per step for every (held CUs, budget) pair: the row "held = 32" says which budget the reducer should set, the
column "budget = 0" what ignoring the collectives costs.

    python tools/ddp_contention.py [--steps 10] [--held 0,16,32] [--budgets 0,-16,-32,-48]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmtg_amd import MMTG, hip, synth
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.trainer import MMTGTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--held", default="0,16,32")
ap.add_argument("--budgets", default="0,-16,-32,-48")
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
dev = "cuda"
S, V = 5, 13317
mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
gcfg = gpt2_config(n_layer=12, vocab_size=V)
model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype="bf16", token_table=synth.make_token_table(V, seed=2))
model.reset_parameters(seed=0)
model.to(dev).train()
tr = MMTGTrainer(model, lr=1e-5, alpha=0.2, warmup_steps=10, total_steps=100000)
batches = [{k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.make_batch(a.batch, mcfg, dcfg, V, seed=i).items()} for i in range(2)]
side = torch.cuda.Stream()


def run(n, held, step_ms):
    """n steps; with held > 0 every step runs beside a stand-in that holds `held` CUs for ~the step's length, launched first."""
    for i in range(n):
        if held:
            side.wait_stream(torch.cuda.current_stream())
            hip.debug_occupy(held, step_ms * 1e3, stream=side)
        tr.step(batches[i % 2], stage=3)
        if held:
            torch.cuda.current_stream().wait_stream(side)


def timed(n, held, step_ms):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(n, held, step_ms)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


run(5, 0, 0)
base = timed(a.steps, 0, 0)
print("full-size bf16 training step, %d rows x 236 positions, one GPU; baseline %.3f ms per step" % (a.batch, base))
budgets = [int(x) for x in a.budgets.split(",")]
print("%-10s" % "held CUs" + "".join("%14s" % ("budget %d" % b) for b in budgets))
for held in (int(x) for x in a.held.split(",")):
    row = []
    for b in budgets:
        hip.gemm_cu_budget(b)
        run(2, held, base * 0.9)
        # the stand-in sleeps 0.9 x the undisturbed step: it is over before the step it runs beside, so steps do not queue behind it
        row.append(timed(a.steps, held, base * 0.9))
    hip.gemm_cu_budget(0)
    print("%-10d" % held + "".join("%14.3f" % x for x in row), flush=True)
