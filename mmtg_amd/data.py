"""Input side of the training path.

* ``MyDataset`` -- drop-in for the reference's dataset class (src/MyDataset.py:13-118): same constructor,
  same per-item dict of numpy arrays (``topic_ids/tpw_attention_mask/tpw_type_ids`` of the 15-token prompt,
  ``targets/attention_mask/type_ids`` of the 2*S sentences + [SEP], the 2048-d WenLan vectors, ``rating``).
  The number of experience steps is read from the record (``img_<i>_emb`` keys) instead of the reference's
  hard-coded ``range(5)``.
* ``pack_binary`` / ``BinaryDataset`` -- a flat on-disk format replacing the list-of-dict pickle
  (README.md:47-65): one ``.npy`` per field, memory-mapped, so a worker touches only the rows it serves and
  no 11 x 2048 Python float lists are converted per item.
* ``DeviceLoader`` -- collates rows into pinned host buffers and copies them to HBM on a side stream, one
  batch ahead of the consumer; optionally applies the curriculum stage filter (train.py:178-186) before the
  copy so filtered rows never cross PCIe.

Host-side plumbing only: no arithmetic of the model lives here.
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np
import torch

START_TOKEN, EOS_TOKEN = "[#START#]", "[#EOS#]"
TOPIC_PREFIX = "主题词："          # "Topic words: " (MyDataset.py:66)
_STRIP = (" ", "\n", "\t", "\r", "\xa0", "　")          # MyDataset.py:92-93

FIELDS = ("topic_ids", "tpw_attention_mask", "tpw_type_ids", "topic_emb", "img_embs", "r_embs",
          "targets", "attention_mask", "type_ids")


def _steps_of(record):
    n = 0
    while "img_%d_emb" % n in record:
        n += 1
    return n


class MyDataset(torch.utils.data.Dataset):
    """Same contract as src/MyDataset.py: ``MyDataset(file_path, tokenizer, data_config, if_train=True)``;
    ``file_path`` is the reference's pickle (a list of dicts) or an already loaded list."""

    def __init__(self, file_path, tokenizer, data_config, if_train=True):
        super().__init__()
        if isinstance(file_path, (list, tuple)):
            self.data = list(file_path)
        else:
            with open(file_path, "rb") as f:
                self.data = pickle.load(f)
        self._tokenizer = tokenizer
        self._max_topic_length = data_config["topic_prompt_length"]
        self._max_sent_length = data_config["max_sent_length"]
        self.if_train = if_train

    def __len__(self):
        return len(self.data)

    # ---- prompt: "主题词：" + topic words -> P ids, mask 1 / type 1 on real tokens, 0 / 0 on padding
    def convert_topic(self, topic_words):
        tok, P = self._tokenizer, self._max_topic_length
        ids = tok.convert_tokens_to_ids(tok.tokenize(TOPIC_PREFIX + topic_words))[:P]
        n = len(ids)
        pad = P - n
        return ids + [tok.pad_token_id] * pad, [1] * n + [0] * pad, [1] * n + [0] * pad

    # ---- lyrics: per sentence [#START#] w_1..w_n PAD.. [#EOS#] in a fixed slot of max_sent_length + 2,
    #      sentence pair k carries type k+1 (the fifth pair type 1 again, MyDataset.py:99-102), [SEP] closes
    def convert_lyrics2ids(self, lyrics):
        tok, msl = self._tokenizer, self._max_sent_length
        tokens, mask, types = [], [], []
        for s, sent in enumerate(lyrics):
            for ch in _STRIP:
                sent = sent.replace(ch, "")
            words = tok.tokenize(sent)[:msl]
            pair = s // 2
            tid = 1 if pair == 4 else pair + 1
            n, pad = len(words), msl - len(words)
            tokens += [START_TOKEN] + words + [tok.pad_token] * pad + [EOS_TOKEN]
            mask += [1] + [1] * n + [0] * pad + [1]
            types += [0] + [tid] * n + [0] * pad + [0]
        tokens.append(tok.sep_token)
        mask.append(1)
        types.append(0)
        return tok.convert_tokens_to_ids(tokens), mask, types

    def __getitem__(self, idx):
        rec = self.data[idx]
        S = _steps_of(rec)
        topic_ids, tmask, ttype = self.convert_topic(rec["topic"])
        targets, amask, types = self.convert_lyrics2ids(rec["lyrics"])
        item = {
            "topic_ids": np.asarray(topic_ids),
            "tpw_attention_mask": np.asarray(tmask),
            "tpw_type_ids": np.asarray(ttype),
            "topic_emb": np.asarray(rec["topic_emb"]),
            "img_embs": np.asarray([rec["img_%d_emb" % i] for i in range(S)]),
            "r_embs": np.asarray([rec["r_%d_emb" % i] for i in range(S)]),
            "targets": np.asarray(targets),
            "attention_mask": np.asarray(amask),
            "type_ids": np.asarray(types),
        }
        if self.if_train:
            item["rating"] = rec["rating"]
        return item


# ------------------------------------------------------------------ binary format
def pack_binary(dataset, out_dir, emb_dtype=np.float32):
    """Materialise every item of a ``MyDataset`` once into ``out_dir``: ``<field>.npy`` ([N, ...], ids int64,
    embeddings ``emb_dtype``) + ``meta.json``.  One pass over the pickle; afterwards no tokenizer and no pickle
    are needed to train."""
    os.makedirs(out_dir, exist_ok=True)
    n = len(dataset)
    first = dataset[0]
    fields = list(FIELDS) + (["rating"] if "rating" in first else [])
    arrays = {}
    for k in fields:
        v = np.asarray(first[k])
        dt = emb_dtype if v.dtype.kind == "f" else np.int64
        arrays[k] = np.lib.format.open_memmap(os.path.join(out_dir, k + ".npy"), mode="w+", dtype=dt, shape=(n,) + v.shape)
    for i in range(n):
        item = first if i == 0 else dataset[i]
        for k in fields:
            arrays[k][i] = np.asarray(item[k])
    for a in arrays.values():
        a.flush()
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump({"rows": n, "fields": fields, "format": "mmtg-binary-1"}, f)
    return out_dir


class BinaryDataset(torch.utils.data.Dataset):
    """Memory-mapped view of a ``pack_binary`` directory; items have the ``MyDataset`` layout."""

    def __init__(self, path, if_train=True):
        with open(os.path.join(path, "meta.json")) as f:
            self.meta = json.load(f)
        if self.meta.get("format") != "mmtg-binary-1":
            raise ValueError("not an mmtg binary dataset: %s" % path)
        self.fields = [k for k in self.meta["fields"] if if_train or k != "rating"]
        self.arrays = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r") for k in self.fields}
        for k, a in self.arrays.items():
            if a.shape[0] != self.meta["rows"]:
                raise ValueError("field %s has %d rows, meta says %d" % (k, a.shape[0], self.meta["rows"]))

    def __len__(self):
        return self.meta["rows"]

    def __getitem__(self, idx):
        item = {k: np.asarray(a[idx]) for k, a in self.arrays.items()}
        if "rating" in item:
            item["rating"] = int(item["rating"])
        return item

    def rows(self, idx):
        """Batch of rows (index array) as numpy arrays -- the fast path of ``DeviceLoader``."""
        idx = np.asarray(idx)
        order = np.argsort(idx, kind="stable")          # ascending file order for the page cache
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        return {k: np.asarray(a[idx[order]])[inv] for k, a in self.arrays.items()}


# ------------------------------------------------------------------ pinned, asynchronous host -> HBM
def stage_filter(ratings, stage):
    """Kept row indices, in the reference's order, of the curriculum stage filter (train.py:178-183): the same
    rule as ``trainer.curriculum_filter`` (stage 1: rating < 2 then rating > 4; stage 2: < 3 then > 3; stage 3:
    all rows), applied on the host so that dropped rows are never copied to the GPU."""
    from .trainer import curriculum_filter
    return curriculum_filter(torch.from_numpy(np.asarray(ratings)), stage).numpy()


class DeviceLoader:
    """Iterates ``(device batch dict)`` over a dataset: rows are collated into two alternating sets of pinned
    buffers and copied with ``non_blocking=True`` on a private stream while the previous batch trains; the
    consumer's stream waits on the copy's event only.  ``stage`` in (1, 2) applies ``stage_filter`` on the host
    first (filtered rows are never copied).

    A pinned slot is rewritten only after the HOST has seen its previous copy complete (the training loop never
    synchronises, so the host can run several batches ahead of the DMA engine).  With ``world > 1`` a batch the filter
    empties is still yielded (zero rows): every rank must enter ``MMTGTrainer.step`` the same number of times or the
    gradient all-reduce deadlocks; a single process skips it, as train.py:184-185 does."""

    def __init__(self, dataset, batch_size, device="cuda", shuffle=True, seed=0, stage=3, drop_last=True,
                 rank=0, world=1):
        self.ds, self.bs, self.dev = dataset, batch_size, torch.device(device)
        self.shuffle, self.seed, self.stage, self.drop_last = shuffle, seed, stage, drop_last
        self.rank, self.world = rank, world
        self.epoch = 0
        self._pinned = [None, None]
        self._copied = [None, None]     # per slot: event of the last H2D copy that read the pinned buffers
        self._stream = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None

    def __len__(self):
        per_rank = len(self.ds) // self.world
        return per_rank // self.bs if self.drop_last else -(-per_rank // self.bs)

    def _collate(self, idx):
        if hasattr(self.ds, "rows"):
            return self.ds.rows(idx)
        items = [self.ds[int(i)] for i in idx]
        return {k: np.stack([np.asarray(it[k]) for it in items]) for k in items[0]}

    def _to_pinned(self, slot, batch):
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()    # normally long complete: the copy was issued two batches ago
            self._copied[slot] = None
        pin = self._pinned[slot]
        if pin is None or any(pin[k].shape[0] < v.shape[0] or pin[k].shape[1:] != v.shape[1:] for k, v in batch.items()):
            pin = {}
            for k, v in batch.items():
                dt = torch.float32 if v.dtype.kind == "f" else torch.int64
                t = torch.empty((self.bs,) + v.shape[1:], dtype=dt)
                pin[k] = t.pin_memory() if self._stream is not None else t
            self._pinned[slot] = pin
        out = {}
        for k, v in batch.items():
            dst = pin[k][: v.shape[0]]
            # numpy memcpy into the pinned buffer's own view (torch's multi-threaded CPU copy_ costs tens of
            # milliseconds for a 2.6 MB tensor when the OpenMP pool is oversubscribed)
            np.copyto(dst.numpy(), v, casting="same_kind")
            out[k] = dst
        return out

    def _issue(self, slot, idx):
        batch = self._collate(idx)
        if self.stage in (1, 2) and "rating" in batch:
            keep = stage_filter(batch["rating"], self.stage)
            batch = {k: v[keep] for k, v in batch.items()}
        if next(iter(batch.values())).shape[0] == 0:
            if self.world == 1:
                return None
            empty = {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float32 if v.dtype.kind == "f" else torch.int64)
                     for k, v in batch.items()}
            return {k: v.to(self.dev) for k, v in empty.items()}, None
        host = self._to_pinned(slot, batch)
        if self._stream is None:
            return {k: v.clone() for k, v in host.items()}, None
        with torch.cuda.stream(self._stream):
            dev = {k: v.to(self.dev, non_blocking=True) for k, v in host.items()}
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._copied[slot] = ev
        return dev, ev

    def __iter__(self):
        n = len(self.ds)
        order = np.arange(n)
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(order)
        self.epoch += 1
        order = order[self.rank::self.world][: (n // self.world)]
        starts = list(range(0, len(order), self.bs))
        if self.drop_last and starts and len(order) - starts[-1] < self.bs:
            starts.pop()
        pending = None
        for i, s in enumerate(starts):
            nxt = self._issue(i & 1, order[s:s + self.bs])
            if pending is not None:
                yield self._ready(pending)
            pending = nxt
        if pending is not None:
            yield self._ready(pending)

    def _ready(self, pending):
        dev, ev = pending
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)
            for t in dev.values():
                t.record_stream(torch.cuda.current_stream(self.dev))
        return dev
