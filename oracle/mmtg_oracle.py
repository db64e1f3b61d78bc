"""ORACLE -- test infrastructure only, never the product path.

A CPU restatement (plain PyTorch fp32 tensor algebra + autograd, no
``transformers`` / HF dependency, no nn.Module from the reference) of the MMTG
training + generation hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this file; the shipped
package ``mmtg_amd`` must never do so.

Pinning: the reference ships no tests and no golden vectors for this path
(SURVEY.md section 4), so parity is pinned by fixtures generated in the build
container by *executing the reference itself* (``tools/make_golden.py`` ->
``tests/golden/*.npz``); ``tests/test_oracle_golden.py`` checks every function
below against them.  GPT-2 arithmetic lives in the un-vendored third-party
dependency ``transformers==4.12.3`` (reference requirements.txt:2; 5.15.0 was
what executed when the fixtures were made); its published algorithm is
restated in :func:`gpt2_forward` and anchored on the reference's call sites
src/model.py:282-288 and 320-326.

Weights are a flat ``dict`` keyed by the reference's state-dict names
(SURVEY.md section 5).  Every function cites the reference lines it follows.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SENT_SLOT = 22  # generate.py:118-121 hard-codes max_sent_length + 2 = 22


# --------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------
class Shapes:
    """Sizes the reference spreads over configs.py / model_config.json."""

    def __init__(self, model_cfgs, data_cfg, gpt2_cfg):
        self.S = model_cfgs["seq_len"]
        self.E = model_cfgs["topic"]["input_dim"]
        self.H = model_cfgs["topic"]["hidden_dim"]
        self.heads = model_cfgs["SELF_ATT"]["attention_heads"]
        self.P = data_cfg["topic_prompt_length"]
        self.msl = data_cfg["max_sent_length"]
        self.max_seq_length = data_cfg["max_seq_length"]
        self.two_sents = 2 * (self.msl + 2)            # model.py:250
        self.D = gpt2_cfg["n_embd"]
        self.nH = gpt2_cfg["n_head"]
        self.L = gpt2_cfg["n_layer"]
        self.V = gpt2_cfg["vocab_size"]
        self.eps = gpt2_cfg.get("layer_norm_epsilon", 1e-5)


def weights_to_torch(w, requires_grad=False):
    out = {}
    for k, v in w.items():
        if k == "decoder.gpt2.lm_head.weight":
            continue
        t = torch.from_numpy(np.array(v, dtype=np.float32, copy=True))
        t.requires_grad_(requires_grad)
        out[k] = t
    out["decoder.gpt2.lm_head.weight"] = out["decoder.gpt2.transformer.wte.weight"]  # tied
    return out


def gaussian_priors(S):
    """q_i[j] = phi(j; mu=i, sigma=1) normalised over j in 0..S-1
    (model.py:116-120, with the hard-coded 5 generalised to S)."""
    j = np.arange(S, dtype=np.float64)
    rows = []
    for i in range(S):
        v = np.exp(-0.5 * (j - i) ** 2) / math.sqrt(2 * math.pi)
        rows.append(v / v.sum())
    return torch.tensor(np.stack(rows), dtype=torch.float32)


# --------------------------------------------------------------------------
# encoder + attention fuser
# --------------------------------------------------------------------------
def gru_forward(x, w_ih, w_hh, b_ih, b_hh):
    """One-layer GRU, h0 = 0, sequence-first (model.py:78-79; PyTorch gate
    order r,z,n:  n = tanh(W_in x + b_in + r*(W_hn h + b_hn)),
    h' = (1-z)*n + z*h)."""
    S, B, _ = x.shape
    Hh = w_hh.shape[1]
    h = x.new_zeros(B, Hh)
    gi_all = x @ w_ih.t() + b_ih
    outs = []
    for t in range(S):
        gi = gi_all[t]
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi.chunk(3, -1)
        h_r, h_z, h_n = gh.chunk(3, -1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1 - z) * n + z * h
        outs.append(h)
    return torch.stack(outs, 0)


def lstm_forward(x, w_ih, w_hh, b_ih, b_hh):
    """One nn.LSTM layer, h0 = c0 = 0 (model.py:46-47 / 55-56; gate order i,f,g,o:
    c' = f*c + i*g, h' = o*tanh(c'))."""
    S, B, _ = x.shape
    Hh = w_hh.shape[1]
    h = x.new_zeros(B, Hh)
    c = x.new_zeros(B, Hh)
    gi_all = x @ w_ih.t() + b_ih
    outs = []
    for t in range(S):
        a_i, a_f, a_g, a_o = (gi_all[t] + h @ w_hh.t() + b_hh).chunk(4, -1)
        c = torch.sigmoid(a_f) * c + torch.sigmoid(a_i) * torch.tanh(a_g)
        h = torch.sigmoid(a_o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 0)


def rnn_relu_forward(x, w_ih, w_hh, b_ih, b_hh):
    """One nn.RNN(nonlinearity="relu") layer, h0 = 0 (model.py:42-43 / 52-53)."""
    S, B, _ = x.shape
    h = x.new_zeros(B, w_hh.shape[1])
    gi_all = x @ w_ih.t() + b_ih
    outs = []
    for t in range(S):
        h = torch.relu(gi_all[t] + h @ w_hh.t() + b_hh)
        outs.append(h)
    return torch.stack(outs, 0)


_RNN_LAYER = {"GRU": gru_forward, "LSTM": lstm_forward, "RNN": rnn_relu_forward}


def rnn_stack_forward(w, prefix, x, masks=None):
    """The channel's nn.RNNBase stack (model.py:41-59): type and depth are read off the state dict (rows of weight_ih_l0
    per hidden unit = gates: 3 GRU, 4 LSTM, 1 RNN; layers = weight_ih_l* present).  Eval semantics by default; masks[l]
    ([S,B,H], already scaled by 1/keep) is nn.RNNBase's training-mode dropout on the output of layer l (all but the last)."""
    gates = w[prefix + "weight_ih_l0"].shape[0] // w[prefix + "weight_hh_l0"].shape[1]
    layer = _RNN_LAYER[{3: "GRU", 4: "LSTM", 1: "RNN"}[gates]]
    l = 0
    while prefix + "weight_ih_l%d" % l in w:
        x = layer(x, *(w[prefix + n + "_l%d" % l] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        l += 1
        if masks is not None and prefix + "weight_ih_l%d" % l in w:
            x = x * masks[l - 1]
    return x


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def encoder_forward(w, topic_emb, img, txt, rnn_masks=None):
    """MultiModalEncoder.forward (model.py:63-81) then ln_layer1..3
    (model.py:380-382).  img/txt are [S,B,E]."""
    p = "encoder."
    topic = (topic_emb @ w[p + "topic_fc.weight"].t() + w[p + "topic_fc.bias"]).unsqueeze(0)
    oi = rnn_stack_forward(w, p + "rnns_image.", img, None if rnn_masks is None else rnn_masks["image"])
    ot = rnn_stack_forward(w, p + "rnns_text.", txt, None if rnn_masks is None else rnn_masks["text"])
    raw = (topic, oi, ot)
    normed = (layer_norm(topic, w["ln_layer1.weight"], w["ln_layer1.bias"]),
              layer_norm(oi, w["ln_layer2.weight"], w["ln_layer2.bias"]),
              layer_norm(ot, w["ln_layer3.weight"], w["ln_layer3.bias"]))
    return raw, normed


def alpha_attention(w, prefix, x, heads, priors):
    """InnerModalAttentionLayer.forward (model.py:133-161).  x: [B,S,H].
    kl_i = KLDivLoss(batchmean)(log P[:,:,i,:], prior_i)
         = sum_{b,h,j} q_i[j] (log q_i[j] - log P[b,h,i,j]) / B ; returns mean_i."""
    B, S, Hd = x.shape
    dh = Hd // heads

    def lin(n):
        return x @ w[f"{prefix}.{n}.weight"].t() + w[f"{prefix}.{n}.bias"]

    def split(t):
        return t.view(B, S, heads, dh).permute(0, 2, 1, 3)

    q, k, v = split(lin("query")), split(lin("key")), split(lin("value"))
    scores = q @ k.transpose(-1, -2) / math.sqrt(dh)
    probs = torch.softmax(scores, -1)                       # [B,h,S,S]
    logp = probs.log()
    pri = priors.to(x.dtype)                                # [S,S]
    kl_rows = (pri * (pri.log() - logp.permute(0, 1, 2, 3))).sum((0, 1, 3)) / B   # [S]
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    return ctx, kl_rows.mean()


def beta_attention(w, topic, img, txt):
    """MultiModalAttentionLayer.forward (model.py:181-202).  topic [1,B,H],
    img/txt [S,B,H] -> [S,B,E].  Per step i its own Linear(H->1) scores the
    three sources; softmax over the 3; blend; shared out_linear."""
    S = img.shape[0]
    outs = []
    for i in range(S):
        wi = w[f"mm_atten_layer.att_matrices.{i}.weight"]       # [1,H]
        bi = w[f"mm_atten_layer.att_matrices.{i}.bias"]
        src = torch.stack([topic[0], img[i], txt[i]], 1)        # [B,3,H]
        sc = (src @ wi.t()).squeeze(-1) + bi                    # [B,3]
        a = torch.softmax(sc, -1).unsqueeze(1)                  # [B,1,3]
        o = (a @ src).squeeze(1)                                # [B,H]
        outs.append(o @ w["mm_atten_layer.out_linear.weight"].t()
                    + w["mm_atten_layer.out_linear.bias"])
    return torch.stack(outs, 0)


# --------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------
def condition_embeddings(table, topic_ids, input_ids, concat_output, two_sents):
    """WenLan lookup + experience add (model.py:254-268):
    X[b,P+p] = E[input_ids[b,p]] + c[b, p // two_sents]  for p < two_sents*S,
    X[b,j]   = E[topic_ids[b,j]]                          for j < P."""
    S = concat_output.shape[1]
    tp = table[topic_ids.long()]
    xi = table[input_ids.long()].clone()
    Lc = xi.shape[1]
    for k in range(S):
        lo, hi = two_sents * k, min(two_sents * (k + 1), Lc)
        if lo >= Lc:
            break
        xi[:, lo:hi] = xi[:, lo:hi] + concat_output[:, k:k + 1]
    return torch.cat([tp, xi], 1)


def projector(w, x):
    """model.py:279-281."""
    h = torch.tanh(x @ w["decoder.projector_layer1.weight"].t() + w["decoder.projector_layer1.bias"])
    return h @ w["decoder.projector_layer2.weight"].t() + w["decoder.projector_layer2.bias"]


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def gpt2_forward(w, sh, inputs_embeds, type_ids, attention_mask, collect=None):
    """GPT2LMHeadModel.forward(inputs_embeds=, token_type_ids=, attention_mask=)
    as transformers 4.12.3 defines it: h0 = x + wpe[0..T) + wte[type_ids];
    L x { LN -> c_attn (Conv1D, weight [in,out]) -> causal MHA, scale
    1/sqrt(dh), future keys and padded keys excluded (4.12.3 writes -1e4 /
    adds -10000, which underflow to exactly 0 after fp32 softmax) -> c_proj ->
    +res ; LN -> c_fc -> gelu_new -> c_proj -> +res } ; ln_f ; logits = h wte^T.
    Dropout sites are identity here (eval / p = 0)."""
    pre = "decoder.gpt2.transformer."
    B, T, D = inputs_embeds.shape
    nH, dh = sh.nH, D // sh.nH
    h = inputs_embeds + w[pre + "wpe.weight"][:T] + w[pre + "wte.weight"][type_ids.long()]
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    keep = causal[None, None] & (attention_mask.bool()[:, None, None, :])
    for l in range(sh.L):
        p = f"{pre}h.{l}."
        a = layer_norm(h, w[p + "ln_1.weight"], w[p + "ln_1.bias"], sh.eps)
        qkv = a @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
        q, k, v = (t.view(B, T, nH, dh).transpose(1, 2) for t in qkv.split(D, -1))
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        sc = sc.masked_fill(~keep, float("-inf"))
        pr = torch.softmax(sc, -1)
        ctx = (pr @ v).transpose(1, 2).reshape(B, T, D)
        h = h + ctx @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"]
        m = layer_norm(h, w[p + "ln_2.weight"], w[p + "ln_2.bias"], sh.eps)
        m = gelu_new(m @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
        h = h + m @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"]
        if collect is not None:
            collect[f"block{l}"] = h
    hf = layer_norm(h, w[pre + "ln_f.weight"], w[pre + "ln_f.bias"], sh.eps)
    if collect is not None:
        collect["ln_f"] = hf
    return hf @ w["decoder.gpt2.lm_head.weight"].t()


def lm_loss_shifted(logits, labels):
    """GPT-2's internal loss (labels= at model.py:286): mean CE of logits[:, :-1]
    against labels[:, 1:], pads counted (no ignore index)."""
    V = logits.shape[-1]
    return F.cross_entropy(logits[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1).long())


def inference_type_ids_and_mask(sh, input_ids, tpw_type_ids, tpw_att_mask, per_row=False):
    """Inference branch of GPT2_Decoder.forward (model.py:290-312): lyric
    position i gets type 0 if (i+1) % 22 in {0,1}; else 0 if the token is PAD,
    else [1..10,1][i // 22]; mask 0 where the token is PAD.  The reference
    reads row 0 only and repeats it over the batch (per_row=False)."""
    sent = sh.msl + 2
    max_sent_num = sh.max_seq_length // sent + 1
    tlist = list(range(1, max_sent_num)) + [1]
    B, n = input_ids.shape
    src = input_ids if per_row else input_ids[:1].expand(B, n)
    types = torch.zeros(B, n, dtype=torch.long)
    mask = torch.ones(B, n, dtype=torch.long)
    for i in range(n):
        is_pad = src[:, i].long() == 0
        if not ((i + 1) % sent == 0 or (i + 1) % sent == 1):
            types[:, i] = torch.where(is_pad, torch.zeros(B, dtype=torch.long),
                                      torch.full((B,), tlist[i // sent], dtype=torch.long))
        mask[:, i] = torch.where(is_pad, torch.zeros(B, dtype=torch.long),
                                 torch.ones(B, dtype=torch.long))
    return (torch.cat([tpw_type_ids.long(), types], 1),
            torch.cat([tpw_att_mask.long(), mask], 1))


def mmtg_forward(w, sh, table, batch, train_flag=True, collect=None, per_row_infer=False, rnn_masks=None):
    """MMTG.forward (model.py:356-400) -> (lm_loss, kl, logits[B,T,V])."""
    topic_emb = batch["topic_emb"].float()
    img = batch["img_embs"].float().transpose(0, 1)
    txt = batch["r_embs"].float().transpose(0, 1)
    raw, (tn, im, tx) = encoder_forward(w, topic_emb, img, txt, rnn_masks)
    priors = gaussian_priors(sh.S)
    ia, ikl = alpha_attention(w, "img_inner_atten_layer", im.transpose(0, 1), sh.heads, priors)
    ta, tkl = alpha_attention(w, "text_inner_atten_layer", tx.transpose(0, 1), sh.heads, priors)
    mm = beta_attention(w, tn, ia.transpose(0, 1), ta.transpose(0, 1))      # [S,B,E]
    if collect is not None:
        collect.update(enc_topic=raw[0], enc_image=raw[1], enc_text=raw[2], ln_topic=tn,
                       ln_image=im, ln_text=tx, img_inner=ia, img_inner_kl=ikl,
                       text_inner=ta, text_inner_kl=tkl, mm_out=mm)
    input_ids = batch["targets"]
    topic_ids = batch["topic_ids"]
    x = condition_embeddings(table, topic_ids, input_ids, mm.transpose(0, 1), sh.two_sents)
    if train_flag:                                                       # model.py:270-288
        type_ids = torch.cat([batch["tpw_type_ids"].long(), batch["type_ids"].long()], 1)
        mask = torch.cat([batch["tpw_attention_mask"].long(), batch["attention_mask"].long()], 1)
        labels = torch.cat([topic_ids.long(), input_ids.long()], 1)
    else:                                                                # model.py:290-326
        type_ids, mask = inference_type_ids_and_mask(
            sh, input_ids, batch["tpw_type_ids"], batch["tpw_attention_mask"], per_row_infer)
        labels = torch.zeros(x.shape[0], x.shape[1], dtype=torch.long)
    g = projector(w, x)
    if collect is not None:
        collect["proj_out"] = g
    logits = gpt2_forward(w, sh, g, type_ids, mask, collect)
    return lm_loss_shifted(logits, labels), (ikl + tkl).mean(), logits


class CachedForward:
    """KV-cached stand-in for ``forward_fn`` of ``sample_sequence`` (batch 1, inference branch).

    The reference re-runs the whole prefix for every token (generate.py:117-142, O(L^2)).  GPT-2 is causal and the
    inference branch's rebuilt type ids / key mask of a position depend on that position's own token only
    (model.py:296-312), so the logits of the LAST position are a function of the new positions' rows and the K / V rows
    of the earlier ones.  This object keeps those per layer and, at every call, runs only the positions it has not seen
    -- the SURVEY 8(d) "cached" CPU baseline.  Returns logits [1, n_new, V] (sample_sequence reads [0, -1, :]).
    Pinned by tests/test_oracle_golden.py: identical ids, logits within fp32 rounding of the uncached loop."""

    def __init__(self, w, sh, table):
        self.w, self.sh, self.table = w, sh, table
        self.n = 0                   # decoder positions already in the cache (prompt + lyrics)
        self.k = [None] * sh.L
        self.v = [None] * sh.L
        self.keep = None             # key mask of the cached positions
        self.c = None                # experience vectors [1, S, E]

    def __call__(self, batch):
        w, sh = self.w, self.sh
        pre = "decoder.gpt2.transformer."
        if self.c is None:
            topic_emb = batch["topic_emb"].float()
            img = batch["img_embs"].float().transpose(0, 1)
            txt = batch["r_embs"].float().transpose(0, 1)
            _, (tn, im, tx) = encoder_forward(w, topic_emb, img, txt)
            priors = gaussian_priors(sh.S)
            ia, _ = alpha_attention(w, "img_inner_atten_layer", im.transpose(0, 1), sh.heads, priors)
            ta, _ = alpha_attention(w, "text_inner_atten_layer", tx.transpose(0, 1), sh.heads, priors)
            self.c = beta_attention(w, tn, ia.transpose(0, 1), ta.transpose(0, 1)).transpose(0, 1)
        input_ids, topic_ids = batch["targets"], batch["topic_ids"]
        x = condition_embeddings(self.table, topic_ids, input_ids, self.c, sh.two_sents)
        type_ids, mask = inference_type_ids_and_mask(sh, input_ids, batch["tpw_type_ids"], batch["tpw_attention_mask"])
        T = x.shape[1]
        n0 = self.n
        g = projector(w, x[:, n0:])
        D = g.shape[-1]
        nH, dh = sh.nH, D // sh.nH
        h = g + w[pre + "wpe.weight"][n0:T] + w[pre + "wte.weight"][type_ids[:, n0:].long()]
        nn = T - n0
        keep_k = mask.bool()[:, None, None, :]                                   # [1,1,1,T]
        causal = (torch.arange(T)[None, :] <= (n0 + torch.arange(nn))[:, None])   # [nn, T]
        keep = causal[None, None] & keep_k
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            a = layer_norm(h, w[p + "ln_1.weight"], w[p + "ln_1.bias"], sh.eps)
            qkv = a @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
            q, k, v = (t.view(1, nn, nH, dh).transpose(1, 2) for t in qkv.split(D, -1))
            self.k[l] = k if self.k[l] is None else torch.cat([self.k[l], k], 2)
            self.v[l] = v if self.v[l] is None else torch.cat([self.v[l], v], 2)
            sc = (q @ self.k[l].transpose(-1, -2)) / math.sqrt(dh)
            pr = torch.softmax(sc.masked_fill(~keep, float("-inf")), -1)
            ctx = (pr @ self.v[l]).transpose(1, 2).reshape(1, nn, D)
            h = h + ctx @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"]
            m = layer_norm(h, w[p + "ln_2.weight"], w[p + "ln_2.bias"], sh.eps)
            m = gelu_new(m @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
            h = h + m @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"]
        self.n = T
        hf = layer_norm(h[:, -1:], w[pre + "ln_f.weight"], w[pre + "ln_f.bias"], sh.eps)
        return hf @ w["decoder.gpt2.lm_head.weight"].t()


# --------------------------------------------------------------------------
# loss / optimiser / schedule  (loss.py:45-74, train.py:137-148,192-197)
# --------------------------------------------------------------------------
def my_loss(logits, targets, ratings, stage, P):
    """MyLoss.forward: y = rating>4 (stage 1) / rating>3 (stages 2,3);
    CE_b = mean_t CE(logits[b,P:-1], targets[b,1:]) pads counted;
    p = 1/exp(CE_b); l_b = -y log(p+1e-10) - (1-y) log(1-p+1e-10); mean_b."""
    y = (ratings > (4 if stage == 1 else 3)).to(logits.dtype)
    sl = logits[:, P:-1]
    tg = targets[:, 1:].long()
    B, n, V = sl.shape
    ce = F.cross_entropy(sl.reshape(-1, V), tg.reshape(-1), reduction="none").view(B, n).mean(1)
    p = 1.0 / torch.exp(ce)
    near0 = 1e-10
    return (-y * torch.log(p + near0) - (1 - y) * torch.log(1 - p + near0)).mean()


def curriculum_filter(ratings, stage):
    """Row filter + reorder of the train loop (train.py:178-183)."""
    r = torch.as_tensor(ratings)
    if stage == 1:
        return torch.cat([torch.where(r < 2)[0], torch.where(r > 4)[0]])
    if stage == 2:
        return torch.cat([torch.where(r < 3)[0], torch.where(r > 3)[0]])
    return torch.arange(len(r))


def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (train.py:194): coef = max/(norm+1e-6), clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_hf_step(params, grads, state, lr, step, betas=(0.9, 0.999), eps=1e-6, wd=0.0):
    """transformers.AdamW as train.py:137 builds it (eps 1e-6, wd 0,
    correct_bias=True): m,v EMAs; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)."""
    b1, b2 = betas
    for i, (p, g) in enumerate(zip(params, grads)):
        st = state.setdefault(i, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        p.addcdiv_(st["m"], st["v"].sqrt().add_(eps), value=-step_size)
        if wd > 0:
            p.add_(p, alpha=-lr * wd)


def linear_schedule(step, warmup, total):
    """get_linear_schedule_with_warmup multiplier (train.py:146-148)."""
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total - step) / max(1, total - warmup))


def train_step(w, sh, table, batch, ratings, stage, alpha, lr, step, opt_state, clip=1.0):
    """One iteration of the hot loop (train.py:188-197) on autograd weights.
    Returns (total_loss, my_loss, kl, pre-clip grad norm)."""
    uniq = []
    seen = set()
    for k, t in w.items():
        if id(t) not in seen and t.requires_grad:
            seen.add(id(t))
            uniq.append((k, t))
    for _, t in uniq:
        t.grad = None
    _, kl, logits = mmtg_forward(w, sh, table, batch, train_flag=True)
    loss = my_loss(logits, batch["targets"], ratings, stage, sh.P)
    total = loss + alpha * kl
    total.backward()
    grads = [t.grad for _, t in uniq]
    gn = clip_grad_norm(grads, clip)
    with torch.no_grad():
        adamw_hf_step([t for _, t in uniq], grads, opt_state, lr, step)
    return total.detach(), loss.detach(), kl.detach(), gn


# --------------------------------------------------------------------------
# generation (generate.py:64-145)
# --------------------------------------------------------------------------
def top_k_top_p_filtering(logits, top_k=0, top_p=0.0, filter_value=-float("inf")):
    """generate.py:64-94 on a 1-D tensor (modifies and returns it)."""
    assert logits.dim() == 1
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][-1]
        logits[logits < kth] = filter_value
    if top_p > 0.0:
        sl, si = torch.sort(logits, descending=True)
        cum = torch.cumsum(torch.softmax(sl, -1), -1)
        rm = cum > top_p
        rm[1:] = rm[:-1].clone()
        rm[0] = False
        logits[si[rm]] = filter_value
    return logits


def process_logits(logits, generated, temperature, repetition_penalty,
                   banned=(1, 2, 100, 102), skip=(0, 102)):
    """generate.py:127-136: divide once PER OCCURRENCE of every generated id
    (a set() of 0-d tensors does not dedupe), sign-agnostic; then temperature;
    then ban [#START#], [#EOS#], [UNK], [SEP]."""
    logits = logits.clone()
    for tok in generated.tolist():
        if tok in skip:
            continue
        logits[tok] = logits[tok] / repetition_penalty
    logits = logits / temperature
    for b in banned:
        if b < logits.numel():
            logits[b] = -float("inf")
    return logits


def sample_sequence(forward_fn, start_input, length, temperature=1.0, top_k=30, top_p=0.0,
                    repitition_penalty=1.0, greedy=True, generator=None, trace=None):
    """generate.py:97-145.  ``forward_fn(inputs) -> logits[B,T,V]`` plays
    model.forward.  Forced [#EOS#]/[#START#] cadence, sticky PAD, and the
    returned list lags the last appended token(s) exactly as the reference's
    does.  greedy=True replaces multinomial by lowest-index argmax (identical
    whenever top_k=1 leaves a single finite logit)."""
    inputs = {}
    for k, v in start_input.items():
        t = torch.as_tensor(np.asarray(v))
        inputs[k] = (t.long() if k == "targets" else t.float()).unsqueeze(0)
    generated = inputs["targets"]
    with torch.no_grad():
        for i in range(length):
            if i > 0 and (i + 2) % SENT_SLOT == 0:
                inputs["targets"] = torch.cat([inputs["targets"], torch.tensor([[2]])], -1)
                continue
            if i > 0 and (i + 2) % SENT_SLOT == 1:
                inputs["targets"] = torch.cat([inputs["targets"], torch.tensor([[1]])], -1)
                continue
            logits = forward_fn(inputs)[0, -1, :]
            if trace is not None:
                trace.append(logits.clone())
            generated = inputs["targets"]
            nl = process_logits(logits, generated[0], temperature, repitition_penalty)
            if generated[0, -1].item() == 0:
                nxt = 0
            else:
                fl = top_k_top_p_filtering(nl, top_k=top_k, top_p=top_p)[:13317]
                if greedy:
                    nxt = int(torch.argmax(fl).item())
                else:
                    nxt = int(torch.multinomial(torch.softmax(fl, -1), 1, generator=generator).item())
            inputs["targets"] = torch.cat([generated, torch.tensor([[nxt]])], -1)
    return generated.tolist()[0]
