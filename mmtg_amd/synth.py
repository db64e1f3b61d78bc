"""Deterministic synthetic weights and batches of the released data shape.

No datasets or checkpoints exist offline, so tests, fixtures and ``bench.py``
all draw from here.  Everything is generated with ``numpy.random.default_rng``
(bit-reproducible across machines) so that the fixture generator running next
to the reference, the oracle and the HIP path all see identical numbers without
any weight file being committed.

Layouts follow the reference:
  * state-dict keys / shapes: SURVEY.md section 5 (probe of src/model.py:331-354)
  * batch layout: src/MyDataset.py:34-118 (prompt 15 ids; 2*S sentences of
    ``[#START#] w.. PAD.. [#EOS#]`` + ``[SEP]``; type ids k+1 per sentence pair
    with the 5th pair wrapping to 1; mask 0 on PAD only)
"""
from __future__ import annotations

import numpy as np

from .configs import rnn_param_shapes

PAD, START, EOS, UNK, CLS, SEP, MASK = 0, 1, 2, 100, 101, 102, 103
FIRST_WORD_ID = 104


def param_spec(model_cfgs, gpt2_cfg):
    """Ordered ``[(key, shape, kind)]`` of every parameter of MMTG.

    ``kind`` selects the synthetic init: 'w' dense weight, 'b' bias,
    'g' LayerNorm gain, 'e' embedding table.
    """
    S = model_cfgs["seq_len"]
    E = model_cfgs["topic"]["input_dim"]
    H = model_cfgs["topic"]["hidden_dim"]
    D = gpt2_cfg["n_embd"]
    V = gpt2_cfg["vocab_size"]
    NP = gpt2_cfg["n_positions"]
    spec = [("encoder.topic_fc.weight", (H, E), "w"),
            ("encoder.topic_fc.bias", (H,), "b")]
    for ch in ("image", "text"):
        for layer in rnn_param_shapes(model_cfgs, ch):
            spec += [(f"encoder.rnns_{ch}.{nm}", shp, "w" if nm.startswith("weight") else "b") for nm, shp in layer]
    for i in (1, 2, 3):
        spec += [(f"ln_layer{i}.weight", (H,), "g"), (f"ln_layer{i}.bias", (H,), "b")]
    for mod in ("img", "text"):
        for nm in ("query", "key", "value"):
            spec += [(f"{mod}_inner_atten_layer.{nm}.weight", (H, H), "w"),
                     (f"{mod}_inner_atten_layer.{nm}.bias", (H,), "b")]
    for i in range(S):
        spec += [(f"mm_atten_layer.att_matrices.{i}.weight", (1, H), "w"),
                 (f"mm_atten_layer.att_matrices.{i}.bias", (1,), "b")]
    spec += [("mm_atten_layer.out_linear.weight", (E, H), "w"),
             ("mm_atten_layer.out_linear.bias", (E,), "b"),
             ("decoder.projector_layer1.weight", (H, E), "w"),
             ("decoder.projector_layer1.bias", (H,), "b"),
             ("decoder.projector_layer2.weight", (D, H), "w"),
             ("decoder.projector_layer2.bias", (D,), "b"),
             ("decoder.gpt2.transformer.wte.weight", (V, D), "e"),
             ("decoder.gpt2.transformer.wpe.weight", (NP, D), "e")]
    for l in range(gpt2_cfg["n_layer"]):
        p = f"decoder.gpt2.transformer.h.{l}."
        spec += [(p + "ln_1.weight", (D,), "g"), (p + "ln_1.bias", (D,), "b"),
                 (p + "attn.c_attn.weight", (D, 3 * D), "w"),
                 (p + "attn.c_attn.bias", (3 * D,), "b"),
                 (p + "attn.c_proj.weight", (D, D), "w"),
                 (p + "attn.c_proj.bias", (D,), "b"),
                 (p + "ln_2.weight", (D,), "g"), (p + "ln_2.bias", (D,), "b"),
                 (p + "mlp.c_fc.weight", (D, 4 * D), "w"),
                 (p + "mlp.c_fc.bias", (4 * D,), "b"),
                 (p + "mlp.c_proj.weight", (4 * D, D), "w"),
                 (p + "mlp.c_proj.bias", (D,), "b")]
    spec += [("decoder.gpt2.transformer.ln_f.weight", (D,), "g"),
             ("decoder.gpt2.transformer.ln_f.bias", (D,), "b")]
    return spec


def make_weights(model_cfgs, gpt2_cfg, seed=0):
    """Synthetic state dict ``{key: float32 ndarray}`` (lm_head tied to wte).

    Scales are chosen so that activations stay O(1) and logits O(1..5): a
    parity test on all-tiny logits would prove nothing.
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape, kind in param_spec(model_cfgs, gpt2_cfg):
        x = rng.standard_normal(shape, dtype=np.float32)
        if kind == "w":
            fan = shape[0] + shape[-1] if len(shape) == 2 else shape[0]
            x *= np.float32(np.sqrt(2.0 / fan) * 1.5)
        elif kind == "b":
            x *= np.float32(0.05)
        elif kind == "g":
            x = np.float32(1.0) + np.float32(0.1) * x
        elif kind == "e":
            x *= np.float32(0.08)
        out[key] = np.ascontiguousarray(x, dtype=np.float32)
    out["decoder.gpt2.lm_head.weight"] = out["decoder.gpt2.transformer.wte.weight"]
    return out


def make_token_table(vocab_size, emb=2048, seed=1, scale=0.05):
    """WenLan text-embedding table ``E[V, emb]`` (reference: a pickled dict
    ``{id: list[2048]}`` at vocab/token_id2emb_dict.pkl, src/model.py:215)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((vocab_size, emb), dtype=np.float32)
            * np.float32(scale))


def make_batch(B, model_cfgs, data_cfg, vocab_size, seed=0, min_len=5,
               pad_prompt=True, low_to_high=None):
    """Synthetic training batch with the MyDataset layout (numpy arrays).

    low_to_high: if given (the "K" knob of the stress config), the ratio of
    low-rating (<=3) to high-rating rows; otherwise ratings ~ U{1..5}.
    """
    rng = np.random.default_rng(seed)
    S = model_cfgs["seq_len"]
    E = data_cfg["wenlan_emb_size"]
    P = data_cfg["topic_prompt_length"]
    msl = data_cfg["max_sent_length"]
    L = 2 * S * (msl + 2) + 1
    lo_word = min(FIRST_WORD_ID, vocab_size - 1)

    topic_ids = np.zeros((B, P), np.int64)
    tpw_mask = np.zeros((B, P), np.int64)
    tpw_type = np.zeros((B, P), np.int64)
    for b in range(B):
        npad = int(rng.integers(0, 8)) if pad_prompt else 0
        n = P - min(npad, P - 1)
        topic_ids[b, :n] = rng.integers(lo_word, vocab_size, n)
        tpw_mask[b, :n] = 1
        tpw_type[b, :n] = 1

    targets = np.zeros((B, L), np.int64)
    amask = np.zeros((B, L), np.int64)
    types = np.zeros((B, L), np.int64)
    for b in range(B):
        pos = 0
        for sent in range(2 * S):
            pair = sent // 2
            tid = 1 if pair == 4 else pair + 1   # MyDataset.py:99-102 (i == 8)
            n = int(rng.integers(min(min_len, msl), msl + 1))
            targets[b, pos] = START
            amask[b, pos] = 1
            pos += 1
            targets[b, pos:pos + n] = rng.integers(lo_word, vocab_size, n)
            amask[b, pos:pos + n] = 1
            types[b, pos:pos + n] = tid
            pos += msl
            targets[b, pos] = EOS
            amask[b, pos] = 1
            pos += 1
        targets[b, pos] = SEP if SEP < vocab_size else EOS
        amask[b, pos] = 1
    if low_to_high is None:
        rating = rng.integers(1, 6, B).astype(np.int64)
    else:
        n_hi = max(1, int(round(B / (1.0 + low_to_high))))
        rating = np.concatenate([rng.integers(4, 6, n_hi),
                                 rng.integers(1, 4, B - n_hi)]).astype(np.int64)
        rng.shuffle(rating)
    return {
        "topic_ids": topic_ids,
        "tpw_attention_mask": tpw_mask,
        "tpw_type_ids": tpw_type,
        "topic_emb": rng.standard_normal((B, E), dtype=np.float32),
        "img_embs": rng.standard_normal((B, S, E), dtype=np.float32),
        "r_embs": rng.standard_normal((B, S, E), dtype=np.float32),
        "targets": targets,
        "attention_mask": amask,
        "type_ids": types,
        "rating": rating,
    }
