"""Input pipeline: the MyDataset layout against the golden vectors produced by running the
reference's class (tools/make_golden.py --only-dataset -> tests/golden/dataset.npz), the binary format and the
batch loader's row order / stage filter.  CPU only; integer layouts are compared exactly."""
import json
import os

import numpy as np
import pytest
import torch

from mmtg_amd.configs import data_config
from mmtg_amd.data import BinaryDataset, DeviceLoader, MyDataset, pack_binary, stage_filter

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz")


class FixtureTokenizer:
    """Replays the reference tokenizer's recorded output for the fixture's strings (no vocabulary file needed)."""

    def __init__(self, meta):
        self._tok, self._vocab = meta["tokenize"], meta["vocab"]
        self.pad_token, self.sep_token, self.pad_token_id = meta["pad_token"], meta["sep_token"], meta["pad_token_id"]

    def tokenize(self, s):
        return list(self._tok[s])

    def convert_tokens_to_ids(self, toks):
        if isinstance(toks, str):
            return self._vocab[toks]
        return [self._vocab[t] for t in toks]


def _records(meta):
    rng = np.random.default_rng(meta["emb_seed"])
    recs = []
    for r in meta["records"]:
        rec = {"topic": r["topic"], "topic_emb": rng.standard_normal(2048).astype(np.float32).tolist(),
               "lyrics": r["lyrics"], "rating": r["rating"]}
        for i in range(5):
            for ch in ("text", "img", "r"):
                rec["%s_%d" % (ch, i)] = "x"
                rec["%s_%d_emb" % (ch, i)] = rng.standard_normal(2048).astype(np.float32).tolist()
        recs.append(rec)
    return recs


@pytest.fixture(scope="module")
def fixture():
    fx = np.load(GOLDEN)
    meta = json.loads(str(fx["meta_json"]))
    return fx, meta, _records(meta)


def test_dataset_layout_matches_reference(fixture):
    fx, meta, recs = fixture
    ds = MyDataset(recs, FixtureTokenizer(meta), data_config(), if_train=True)
    assert len(ds) == 2
    for n in range(2):
        item = ds[n]
        for k in ("topic_ids", "tpw_attention_mask", "tpw_type_ids", "targets", "attention_mask", "type_ids"):
            ref = fx["item%d_%s" % (n, k)]
            assert item[k].shape == ref.shape and np.array_equal(item[k], ref), (n, k)
        assert item["rating"] == int(fx["item%d_rating" % n])
        for k in ("topic_emb", "img_embs", "r_embs"):
            assert tuple(item[k].shape) == tuple(fx["item%d_%s_shape" % (n, k)])
        assert np.array_equal(item["img_embs"][3], np.asarray(recs[n]["img_3_emb"]))
        assert np.array_equal(item["r_embs"][0], np.asarray(recs[n]["r_0_emb"]))
    # released shapes: 15 prompt ids, 2*5 sentences of 22 slots + [SEP]
    assert ds[0]["topic_ids"].shape == (15,) and ds[0]["targets"].shape == (221,)
    # a long sentence is cut at 20 tokens, a long topic at 15 ids; inference items carry no rating
    assert int(ds[0]["attention_mask"][6 * 22:7 * 22].sum()) == 22
    assert int(ds[1]["tpw_attention_mask"].sum()) == 15
    assert "rating" not in MyDataset(recs, FixtureTokenizer(meta), data_config(), if_train=False)[0]


def test_binary_format_round_trip(fixture, tmp_path):
    fx, meta, recs = fixture
    ds = MyDataset(recs * 3, FixtureTokenizer(meta), data_config())
    path = pack_binary(ds, str(tmp_path / "bin"))
    bd = BinaryDataset(path)
    assert len(bd) == 6
    for i in (0, 1, 5):
        a, b = ds[i], bd[i]
        assert set(a) == set(b)
        for k in a:
            if k == "rating":
                assert a[k] == b[k]
            elif "emb" in k:
                assert np.array_equal(np.asarray(a[k], np.float32), b[k])
            else:
                assert np.array_equal(a[k], b[k]) and b[k].dtype == np.int64
    rows = bd.rows([4, 1, 3])
    assert np.array_equal(rows["targets"][0], ds[4]["targets"]) and np.array_equal(rows["targets"][1], ds[1]["targets"])
    assert list(rows["rating"]) == [ds[4]["rating"], ds[1]["rating"], ds[3]["rating"]]
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"format": "other"}, f)
    with pytest.raises(ValueError):
        BinaryDataset(path)


def test_loader_batches_filter_and_shards(fixture, tmp_path):
    fx, meta, recs = fixture
    base = MyDataset(recs, FixtureTokenizer(meta), data_config())
    items = []
    for i in range(12):                      # ratings 1..5 cycling, distinguishable rows
        it = dict(base[i % 2])
        it["rating"] = 1 + i % 5
        it["topic_ids"] = it["topic_ids"].copy()
        it["topic_ids"][0] = 1000 + i
        items.append(it)

    class ListDS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]

    ld = DeviceLoader(ListDS(), batch_size=4, device="cpu", shuffle=False, stage=3)
    batches = list(ld)
    assert len(batches) == len(ld) == 3
    assert [int(b["topic_ids"][0, 0]) for b in batches] == [1000, 1004, 1008]
    assert batches[0]["img_embs"].dtype == torch.float32 and batches[0]["targets"].dtype == torch.int64
    assert tuple(batches[1]["img_embs"].shape) == (4, 5, 2048)
    # stage 1 keeps rating < 2 then rating > 4, in that order (train.py:178-179); other rows are dropped before the copy
    ld1 = DeviceLoader(ListDS(), batch_size=6, device="cpu", shuffle=False, stage=1)
    b = next(iter(ld1))
    assert b["rating"].tolist() == [1, 1, 5] and b["topic_ids"][:, 0].tolist() == [1000, 1005, 1004]
    assert stage_filter([3, 1, 5, 2, 4], 2).tolist() == [1, 3, 2, 4]
    # two ranks see disjoint rows; shuffling is seeded per epoch and identical across ranks
    r0 = DeviceLoader(ListDS(), batch_size=3, device="cpu", shuffle=True, seed=5, rank=0, world=2)
    r1 = DeviceLoader(ListDS(), batch_size=3, device="cpu", shuffle=True, seed=5, rank=1, world=2)
    s0 = sorted(int(v) for b in r0 for v in b["topic_ids"][:, 0])
    s1 = sorted(int(v) for b in r1 for v in b["topic_ids"][:, 0])
    assert len(s0) == len(s1) == 6 and not set(s0) & set(s1) and sorted(s0 + s1) == list(range(1000, 1012))
    assert len([int(v) for b in r0 for v in b["topic_ids"][:, 0]]) == 6      # second epoch: reshuffled, same share size


def test_loader_yields_empty_batches_when_distributed(fixture):
    """A batch the stage filter empties is skipped by a single process (train.py:184-185) but must be yielded, with
    zero rows, when the loader serves one rank of a data-parallel group: every rank has to enter the trainer's
    step (and its all-reduces) the same number of times."""
    fx, meta, recs = fixture
    base = MyDataset(recs, FixtureTokenizer(meta), data_config())
    items = []
    for i in range(8):
        it = dict(base[i % 2])
        it["rating"] = 3 if i < 4 else 5           # rank 0's first batch (rows 0, 2) holds rating 3 only
        items.append(it)

    class ListDS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]

    single = list(DeviceLoader(ListDS(), batch_size=4, device="cpu", shuffle=False, stage=1))
    assert [int(b["rating"].shape[0]) for b in single] == [4]          # the all-rating-3 batch was dropped
    counts = []
    for rank in (0, 1):
        ld = DeviceLoader(ListDS(), batch_size=2, device="cpu", shuffle=False, stage=1, rank=rank, world=2)
        bs = list(ld)
        counts.append([int(b["rating"].shape[0]) for b in bs])
        assert all(tuple(b["img_embs"].shape[1:]) == (5, 2048) and b["targets"].dtype == torch.int64 for b in bs)
    assert counts == [[0, 2], [0, 2]]                                   # same number of steps on both ranks
