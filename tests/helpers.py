"""Shared helpers for the parity tests: rebuild the exact weights / batch a
fixture was generated from (seeds live in the fixture's meta record)."""
import json
import os

import numpy as np
import torch

from mmtg_amd import synth
from mmtg_amd.configs import data_config, make_model_cfgs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(fx["meta"]))
    mcfg = make_model_cfgs(seq_len=meta["S"], **meta.get("enc", {}))      # "enc": encoder types / depths of the variant fixtures
    gcfg = meta["gpt2_cfg"]
    dcfg = data_config(seq_len=meta["S"])
    weights = synth.make_weights(mcfg, gcfg, seed=meta["weight_seed"])
    table = synth.make_token_table(meta["V"], seed=meta["table_seed"])
    batch = synth.make_batch(meta["B"], mcfg, dcfg, meta["V"], seed=meta["batch_seed"])
    return fx, meta, mcfg, gcfg, dcfg, weights, table, batch


def batch_to_torch(batch, device="cpu"):
    return {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in batch.items()}


def sample_like_fixture(a, idx):
    return np.asarray(a, np.float32).reshape(-1)[idx]
