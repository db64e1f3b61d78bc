"""KV-cached batched greedy decode (hipGraph replay) against the reference's own
sample_sequence outputs (golden ids) and against the engine's no-cache path."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import batch_to_torch, load_case  # noqa: E402
from mmtg_amd import MMTG, sample_sequence  # noqa: E402
from mmtg_amd.decode import GreedyDecoder  # noqa: E402

DEV = "cuda"


def build(dtype):
    fx, meta, mcfg, gcfg, dcfg, weights, table, batch = load_case("tiny_s5")
    model = MMTG(mcfg, dcfg, meta["V"], train_flag=False, gpt2_config=gcfg, token_table=table, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model.to(DEV).eval()
    return fx, batch, model


@pytest.mark.parametrize("use_graph", [False, True])
def test_kv_cache_decode_matches_reference_ids(use_graph):
    fx, batch, model = build("f32")
    dp = json.loads(str(fx["decode_params"]))
    tb = {k: v for k, v in batch_to_torch(batch, DEV).items() if k not in ("rating", "targets")}
    dec = GreedyDecoder(model, max_batch=3, use_graph=use_graph)
    for length in (30, 220):
        ids = dec.generate(tb, length, temperature=dp["temperature"], repitition_penalty=dp["repitition_penalty"]).cpu().numpy()
        assert ids.shape == (3, 1 + length)
        for row in (0, 1):
            key = f"greedy_len{length}_row{row}"
            if key in fx.files:
                got = GreedyDecoder.reference_return(ids[row].tolist(), length)
                assert got == fx[key].tolist(), (length, row)
        # forced cadence on every row
        for j in range(2, 1 + length):
            if (j + 1) % 22 == 0:
                assert (ids[:, j] == 2).all()
            if (j + 1) % 22 == 1:
                assert (ids[:, j] == 1).all()
    # row 2 has no golden: compare with the (already pinned) no-cache path of the same engine
    start = {k: np.asarray(v[2]) for k, v in batch.items() if k != "rating"}
    start["targets"] = np.asarray([1])
    ref = sample_sequence(model, start, 220, None, temperature=dp["temperature"], top_k=1, top_p=0.0,
                          repitition_penalty=dp["repitition_penalty"], device=DEV)
    assert GreedyDecoder.reference_return(ids[2].tolist(), 220) == ref


def test_bf16_decode_runs_and_respects_the_rules():
    fx, batch, model = build("bf16")
    tb = {k: v for k, v in batch_to_torch(batch, DEV).items() if k not in ("rating", "targets")}
    dec = GreedyDecoder(model, max_batch=3)
    ids = dec.generate(tb, 128, temperature=1.1, repitition_penalty=1.5).cpu().numpy()
    assert ids.shape == (3, 129) and (ids[:, 0] == 1).all()
    free = [j for j in range(1, 129) if (j + 1) % 22 not in (0, 1)]
    assert not np.isin(ids[:, free], [1, 2, 100, 102]).any()          # banned ids never sampled
    # sticky PAD (generate.py:137-138): a PAD is followed by a PAD until the next forced token
    for r in range(3):
        for j in free:
            if ids[r, j] == 0 and (j + 1) in free:
                assert ids[r, j + 1] == 0
    ids2 = dec.generate(tb, 128, temperature=1.1, repitition_penalty=1.5).cpu().numpy()
    assert (ids == ids2).all()                                         # graph replay is deterministic
