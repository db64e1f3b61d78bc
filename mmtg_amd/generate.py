"""Drop-in for the reference's generation helpers (src/generate.py:64-145).

``sample_sequence`` keeps the reference's signature (including the misspelt
``repitition_penalty``), its forced [#START#]/[#EOS#] cadence, sticky PAD and the
lagging return value.  The model call is the HIP engine; the greedy setting
(top_k=1, top_p=0) post-processes logits with the fused HIP kernel
(mmtg_logits_process_argmax); the stochastic setting uses mmtg_logits_process_sample (the reference's
penalty / temperature / bans / top-k / top-p, then one draw by inverse CDF with a uniform from torch's generator --
torch.multinomial's own stream cannot be reproduced outside torch; the distribution is the reference's).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import hip

SENT_SLOT = 22  # generate.py:118-121
BANNED = (1, 2, 100, 102)  # [#START#] [#EOS#] [UNK] [SEP] in src/vocab/vocab.txt


def top_k_top_p_filtering(logits, top_k=0, top_p=0.0, filter_value=-float("Inf")):
    """Top-k / nucleus filter on a 1-D logits tensor, in place (generate.py:64-94)."""
    assert logits.dim() == 1
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < kth] = filter_value
    if top_p > 0.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits[sorted_indices[remove]] = filter_value
    return logits


def _check_tokenizer(tokenizer):
    if tokenizer is None:
        return
    got = tuple(tokenizer.convert_tokens_to_ids(t) for t in ("[#START#]", "[#EOS#]", "[UNK]", "[SEP]"))
    if got != BANNED:
        raise ValueError("tokenizer special ids %s differ from the reference vocabulary's %s" % (got, BANNED))


def _cached_decoder(model, start_input, length):
    """The KV-cached, graph-replayed decoder for this call, or None when the call is not the one the reference's
    generate.py makes (an MMTG in inference mode on the GPU -- possibly inside the nn.DataParallel wrapper of generate.py:191 --
    started from the single [#START#] token, generate.py:207-209).  MMTG_SAMPLE_RERUN=1 forces the prefix re-run loop."""
    from .model import MMTG
    m = getattr(model, "module", model)
    if not isinstance(m, MMTG) or m.train_flag or not m._flat.is_cuda or os.environ.get("MMTG_SAMPLE_RERUN"):
        return None
    if m.training:
        # a model left in train() mode: the prefix re-run loop follows the module mode (dropout, torch.multinomial draws);
        # the cached decoder never applies dropout and draws from its own uniform stream, so it only serves eval() models
        return None
    t = np.asarray(start_input["targets"]).reshape(-1)
    sh = m.shapes
    if t.shape[0] != 1 or int(t[0]) != 1 or length < 1 or sh.P + length + 1 > sh.NP:
        return None
    from .decode import GreedyDecoder
    cache = m.__dict__.setdefault("_sample_decoders", {})
    key = (m._flat.data_ptr(), int(length))
    dec = cache.get(key)
    if dec is None:
        cache.clear()                       # one decoder (KV caches + graphs) at a time
        dec = cache[key] = GreedyDecoder(m, max_batch=1, max_len=int(length))
    return dec


def sample_sequence(model, start_input, length, tokenizer, temperature=1.0, top_k=30, top_p=0.0,
                    repitition_penalty=1.0, device="cuda"):
    """generate.py:97-145.  When the call is the reference's own (see _cached_decoder) the loop runs on the KV-cached decoder:
    one captured token step replayed per position instead of a full forward over the growing prefix per token; the returned
    list is what the reference loop returns (the sequence before the last model call's append).  Greedy ids are the same as
    the re-run loop's bit for bit in the f32 mode (tests/test_decode_gpu.py)."""
    _check_tokenizer(tokenizer)
    dec = _cached_decoder(model, start_input, length)
    if dec is not None:
        m = getattr(model, "module", model)
        batch = {k: torch.as_tensor(np.asarray(v)).unsqueeze(0) for k, v in start_input.items() if k not in ("targets", "rating")}
        ids = dec.generate(batch, int(length), temperature=temperature, repitition_penalty=repitition_penalty,
                           top_k=int(top_k), top_p=float(top_p))
        return dec.reference_return(ids[0].tolist(), int(length), sent=m.shapes.msl + 2)
    inputs = {}
    for k, v in start_input.items():
        if k == "targets":
            inputs[k] = torch.tensor(np.asarray(v), dtype=torch.long, device=device).unsqueeze(0)
        else:
            inputs[k] = torch.tensor(np.asarray(v), dtype=torch.float32, device=device).unsqueeze(0)
    generated = inputs["targets"]
    greedy = top_k == 1 and top_p == 0.0
    nxt = torch.zeros(1, dtype=torch.long, device=device)
    glen = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.no_grad():
        for i in range(length):
            if i > 0 and (i + 2) % SENT_SLOT == 0:
                inputs["targets"] = torch.cat((inputs["targets"], torch.tensor([[2]], device=device)), dim=-1)
                continue
            if i > 0 and (i + 2) % SENT_SLOT == 1:
                inputs["targets"] = torch.cat((inputs["targets"], torch.tensor([[1]], device=device)), dim=-1)
                continue
            _, _, outputs = model.forward(inputs)
            generated = inputs["targets"]
            if greedy:
                row = outputs[0, -1, :]
                glen.fill_(generated.shape[1])
                hip.logits_process_argmax(row, row.shape[0], min(row.shape[0], 13317), generated.contiguous(),
                                          generated.shape[1], glen, temperature, repitition_penalty, nxt, 1)
                next_token = nxt.view(1, 1).clone()
            else:
                row = outputs[0, -1, :]
                glen.fill_(generated.shape[1])
                u = torch.rand(1, device=device).clamp_(max=1.0 - 2.0 ** -24)
                hip.logits_process_sample(row, row.shape[0], min(row.shape[0], 13317), generated.contiguous(),
                                          generated.shape[1], glen, temperature, repitition_penalty, top_k, top_p, u, nxt, 1)
                next_token = nxt.view(1, 1).clone()
            inputs["targets"] = torch.cat((generated, next_token), dim=-1)
    return generated.tolist()[0]


# ------------------------------------------------------------------ the sampling product path (generate.py:205-235)
def postprocess_tokens(preds):
    """The reference's cut rules and detokenisation of one sampled token list (generate.py:222-235):
    cut after the 10th [#EOS#] when there are at least ten and no [SEP] before the last one, else after the first
    [SEP], else append one; then join, drop [SEP] / [PAD] / [#START#], turn [#EOS#] into the full-width comma and strip
    trailing commas.  (An all-special list leaves an empty string; the reference would raise IndexError there.)"""
    preds = list(preds)
    eos = [i for i, v in enumerate(preds) if v == "[#EOS#]"]
    if len(eos) >= 10 and "[SEP]" not in preds[:eos[-1]]:
        preds = preds[:eos[9] + 1] + ["[SEP]"]
    elif "[SEP]" in preds:
        preds = preds[:preds.index("[SEP]") + 1]
    else:
        preds = preds + ["[SEP]"]
    text = "".join(preds).replace("[SEP]", "").replace("[PAD]", "").replace("[#START#]", "").replace("[#EOS#]", "，")
    while text and text[-1] == "，":
        text = text[:-1]
    return text


def generate_samples(model, rows, tokenizer, n_samples=10, length=None, temperature=1.1, top_k=10, top_p=0.7,
                     repetition_penalty=1.5, max_batch=256, generator=None, decoder=None):
    """n_samples texts per prompt (generate.sh's loop, generate.py:205-235) on the batched, graph-replayed decoder:
    every prompt is replicated over n_samples rows, rows are decoded max_batch at a time with the device-side
    top-k / top-p sampler (one uniform per row and position from `generator`), each row is cut back to what
    sample_sequence would have returned (the sequence before the last model call's append) and post-processed.

    rows: list of dataset items (dicts with topic_ids / tpw_* / topic_emb / img_embs / r_embs, numpy or tensors) or a dict of
    stacked arrays.  Returns a list (one entry per prompt) of lists of n_samples strings."""
    from .decode import GreedyDecoder
    eng = model.engine()
    sh = eng.sh
    length = sh.max_seq_length if length is None else length
    keys = ("topic_ids", "tpw_attention_mask", "tpw_type_ids", "topic_emb", "img_embs", "r_embs")
    if isinstance(rows, dict):
        stacked = {k: torch.as_tensor(np.asarray(rows[k])) for k in keys}
    else:
        stacked = {k: torch.as_tensor(np.stack([np.asarray(r[k]) for r in rows])) for k in keys}
    n_prompts = stacked["topic_ids"].shape[0]
    rep = {k: v.repeat_interleave(n_samples, dim=0) for k, v in stacked.items()}
    total = n_prompts * n_samples
    bsz = min(max_batch, total)
    dec = decoder if decoder is not None and decoder.B == bsz else GreedyDecoder(model, max_batch=bsz, max_len=length)
    texts = []
    for lo in range(0, total, bsz):
        hi = min(total, lo + bsz)
        chunk = {k: v[lo:hi] for k, v in rep.items()}
        if hi - lo < bsz:          # the decoder's batch is fixed: pad the last chunk by repeating its first row
            pad = bsz - (hi - lo)
            chunk = {k: torch.cat([v, v[:1].expand(pad, *v.shape[1:])], 0) for k, v in chunk.items()}
        chunk = {k: v.to(eng.dev) for k, v in chunk.items()}
        ids = dec.generate(chunk, length, temperature=temperature, repitition_penalty=repetition_penalty,
                           top_k=top_k, top_p=top_p, generator=generator).cpu().numpy()
        for r in range(hi - lo):
            seq = GreedyDecoder.reference_return(ids[r].tolist(), length, sent=sh.msl + 2)
            texts.append(postprocess_tokens(tokenizer.convert_ids_to_tokens(seq)))
    return [texts[i * n_samples:(i + 1) * n_samples] for i in range(n_prompts)]
