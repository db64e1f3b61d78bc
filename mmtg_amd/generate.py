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


def sample_sequence(model, start_input, length, tokenizer, temperature=1.0, top_k=30, top_p=0.0,
                    repitition_penalty=1.0, device="cuda"):
    _check_tokenizer(tokenizer)
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
