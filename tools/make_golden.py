#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (it needs /root/reference); the GPU box never
sees the reference -- it gets the small ``.npz`` fixtures written here, which
hold inputs' recipe (seeds) and the reference's outputs.  Nothing from the
reference's source text is stored.

How the reference is made importable:
  shim 1  ``GPT2LMHeadModel.from_pretrained`` needs the HF hub -> patched to
          build ``GPT2LMHeadModel(GPT2Config.from_json_file(config/model_config.json))``
          in eval mode (what from_pretrained returns), eager attention.
  shim 2  ``./vocab/token_id2emb_dict.pkl`` is not in the repo -> a synthetic
          ``{id: list[2048]}`` is pickled into a scratch cwd.
  patch   S != 5: the hard-coded 5-step Gaussian priors
          (src/model.py:116-120) are replaced attribute-wise by S-step ones.

Weights/batches come from ``mmtg_amd.synth`` (numpy default_rng, seeds recorded
in each fixture) so every consumer regenerates them bit-identically.

Usage:  python tools/make_golden.py [--out tests/golden] [--skip-full]
"""
import argparse
import json
import os
import pickle
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("HF_HUB_OFFLINE", "1")
sys.dont_write_bytecode = True

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF_SRC = "/root/reference/src"

import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy import stats  # noqa: E402

from mmtg_amd import synth  # noqa: E402
from mmtg_amd.configs import make_model_cfgs, gpt2_config  # noqa: E402


class StubTokenizer:
    """The four special-token lookups generate.py performs (ids per
    src/vocab/vocab.txt: [#START#]=1 [#EOS#]=2 [UNK]=100 [SEP]=102)."""
    _ids = {"[#START#]": 1, "[#EOS#]": 2, "[UNK]": 100, "[SEP]": 102, "[PAD]": 0}

    def convert_tokens_to_ids(self, tok):
        return self._ids[tok]


def import_reference(scratch, gpt2_cfg, table):
    """chdir into a scratch dir holding config/ and vocab/, import the modules."""
    os.makedirs(os.path.join(scratch, "config"), exist_ok=True)
    os.makedirs(os.path.join(scratch, "vocab"), exist_ok=True)
    with open(os.path.join(scratch, "config", "model_config.json"), "w") as f:
        json.dump(gpt2_cfg, f)
    with open(os.path.join(scratch, "vocab", "token_id2emb_dict.pkl"), "wb") as f:
        pickle.dump({i: table[i].tolist() for i in range(table.shape[0])}, f)
    os.chdir(scratch)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import transformers
    from transformers import GPT2Config, GPT2LMHeadModel

    def _offline_from_pretrained(cls, name, *a, **k):
        cfg = GPT2Config.from_json_file("config/model_config.json")
        cfg._attn_implementation = "eager"
        m = GPT2LMHeadModel(cfg)
        m.eval()
        return m

    GPT2LMHeadModel.from_pretrained = classmethod(_offline_from_pretrained)
    import model as ref_model
    import loss as ref_loss
    import generate as ref_generate
    return ref_model, ref_loss, ref_generate


def build_reference(ref_model, mcfg, dcfg_ref, weights, V):
    m = ref_model.MMTG(mcfg, dcfg_ref, V, train_flag=False)
    S = mcfg["seq_len"]
    if S != 5:
        for layer in (m.img_inner_atten_layer, m.text_inner_atten_layer):
            pri = []
            for i in range(S):
                v = stats.norm.pdf(np.arange(0, S, 1), i, 1)
                pri.append(torch.tensor([x / sum(v) for x in v], dtype=torch.float32))
            layer.normal_dists = pri
    sd = m.state_dict()
    new = {}
    for k, v in sd.items():
        if k in weights:
            assert tuple(v.shape) == weights[k].shape, (k, v.shape, weights[k].shape)
            new[k] = torch.from_numpy(weights[k].copy())
        else:
            new[k] = v  # non-parameter buffers (attn.bias on old transformers)
    missing = [k for k in weights if k not in sd]
    assert not missing, missing
    m.load_state_dict(new)
    m.eval()
    return m


def to_torch(batch):
    out = {}
    for k, v in batch.items():
        t = torch.from_numpy(np.asarray(v))
        # MyDataset hands float64 embeddings (np.asarray of python lists); the
        # model casts with .float() (src/model.py:371-373)
        out[k] = t.double() if t.dtype == torch.float32 else t
    return out


def sample_vec(a, n=32):
    a = np.asarray(a, np.float32).reshape(-1)
    idx = np.unique(np.concatenate([np.arange(min(n, a.size)),
                                    np.linspace(0, a.size - 1, n).astype(np.int64)]))
    return idx.astype(np.int64), a[idx]


def adamw_hf_step(params, lr, step, state, betas=(0.9, 0.999), eps=1e-6, wd=0.0):
    """transformers.AdamW semantics (the optimizer train.py:137 builds):
    bias-corrected step size, eps added to sqrt(v) BEFORE bias correction is
    folded in, decoupled weight decay (0 here)."""
    b1, b2 = betas
    for p in params:
        if p.grad is None:
            continue
        g = p.grad
        st = state.setdefault(id(p), {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = st["v"].sqrt().add_(eps)
        step_size = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
        p.data.addcdiv_(st["m"], denom, value=-step_size)
        if wd > 0:
            p.data.add_(p.data, alpha=-lr * wd)


def greedy_margins(raw, temperature, penalty, length, sent_slot=22):
    """Replay of the reference's logits processing (generate.py:127-136, restated in the oracle's process_logits,
    itself pinned by the id lists) over the raw logits the reference produced per call: for every call the top-2 margin of
    the processed logits (what a reduced-precision path must resolve to pick the same token) and the chosen id."""
    from oracle import mmtg_oracle as O
    targets = [1]
    margins, chosen, j = [], [], 0
    for i in range(length):
        if i > 0 and (i + 2) % sent_slot == 0:
            targets.append(2)
            continue
        if i > 0 and (i + 2) % sent_slot == 1:
            targets.append(1)
            continue
        pl = O.process_logits(torch.from_numpy(raw[j]), torch.tensor(targets), temperature, penalty)[:13317]
        j += 1
        if targets[-1] == 0:
            nxt, mg = 0, np.inf
        else:
            top = torch.topk(pl, 2).values
            nxt, mg = int(torch.argmax(pl)), float(top[0] - top[1])
        margins.append(mg)
        chosen.append(nxt)
        targets.append(nxt)
    assert j == len(raw)
    return np.asarray(margins, np.float32), np.asarray(chosen, np.int64), targets


def case_model(name, out_dir, S, n_layer, V, B, seed, full_logits=True,
               with_grads=True, with_decode=True, decode_rows=(0, 1), store_rawlogits=True, enc=None, decode_lengths=None):
    mcfg = make_model_cfgs(seq_len=S, **(enc or {}))
    gcfg = gpt2_config(n_layer=n_layer, vocab_size=V, n_positions=256 if n_layer <= 2 else 1024,
                       embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    table = synth.make_token_table(V, seed=seed + 1)
    scratch = tempfile.mkdtemp(prefix="mmtg_golden_")
    ref_model, ref_loss, ref_generate = import_reference(scratch, gcfg, table)
    dcfg_ref = ref_model.data_config()
    dcfg_ref.max_seq_length = 2 * S * (dcfg_ref.max_sent_length + 2)
    weights = synth.make_weights(mcfg, gcfg, seed=seed)
    model = build_reference(ref_model, mcfg, dcfg_ref, weights, V)
    model.train_flag = True  # take the training branch of GPT2_Decoder.forward
    batch_np = synth.make_batch(B, mcfg, dcfg_ref, V, seed=seed + 2)
    batch = to_torch(batch_np)

    fx = {"meta": json.dumps({
        "S": S, "n_layer": n_layer, "V": V, "B": B, "weight_seed": seed,
        "table_seed": seed + 1, "batch_seed": seed + 2,
        "torch": torch.__version__, "gpt2_cfg": gcfg, "ratings": batch_np["rating"].tolist(), "enc": enc or {},
        "source": "reference src/model.py + src/loss.py executed on CPU fp32"})}

    # ---- intermediates via forward hooks ---------------------------------
    grabbed = {}

    def hook(key):
        def fn(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            grabbed[key] = o.detach().clone()
            if isinstance(out, tuple) and key.endswith("inner"):
                grabbed[key + "_kl"] = out[1].detach().clone()
        return fn

    hs = [model.encoder.register_forward_hook(
              lambda m, i, o: grabbed.update(enc_topic=o[0].detach().clone(),
                                             enc_image=o[1].detach().clone(),
                                             enc_text=o[2].detach().clone())),
          model.ln_layer1.register_forward_hook(hook("ln_topic")),
          model.ln_layer2.register_forward_hook(hook("ln_image")),
          model.ln_layer3.register_forward_hook(hook("ln_text")),
          model.img_inner_atten_layer.register_forward_hook(hook("img_inner")),
          model.text_inner_atten_layer.register_forward_hook(hook("text_inner")),
          model.mm_atten_layer.register_forward_hook(hook("mm_out")),
          model.decoder.projector_layer2.register_forward_hook(hook("proj_out")),
          model.decoder.gpt2.transformer.ln_f.register_forward_hook(hook("ln_f"))]
    for li, blk in enumerate(model.decoder.gpt2.transformer.h):
        hs.append(blk.register_forward_hook(hook(f"block{li}")))

    for p in model.parameters():
        p.grad = None
    lm_loss, kl, logits = model.forward(batch)
    for h in hs:
        h.remove()

    tstride = 7
    for k, v in grabbed.items():
        a = v.numpy().astype(np.float32)
        if a.ndim == 3 and a.shape[1] > 64:  # [B,T,D] decoder tensors: strided tokens
            a = a[:, ::tstride]
        fx["int_" + k] = a
    fx["tstride"] = np.int64(tstride)
    fx["lm_loss"] = np.float32(lm_loss.item())
    fx["kl"] = np.float32(kl.item())
    lg = logits.detach().numpy().astype(np.float32)
    if full_logits:
        fx["logits"] = lg
    else:
        rng = np.random.default_rng(seed + 3)
        n = 256
        bi = rng.integers(0, B, n)
        ti = rng.integers(0, lg.shape[1], n)
        vi = rng.integers(0, V, n)
        fx["logit_idx"] = np.stack([bi, ti, vi], 1).astype(np.int64)
        fx["logit_val"] = lg[bi, ti, vi]
        fx["logit_top5"] = np.argsort(-lg, axis=-1)[:, :, :5].astype(np.int32)
        fx["logit_top5_val"] = np.take_along_axis(lg, fx["logit_top5"].astype(np.int64), -1)
        fx["logit_lse"] = torch.logsumexp(logits.detach(), -1).numpy().astype(np.float32)

    # ---- MyLoss per curriculum stage (src/loss.py:45-74) ------------------
    crit = ref_loss.MyLoss(dcfg_ref, mcfg)
    ratings = batch["rating"]
    for stage in (1, 2, 3):
        fx[f"myloss_stage{stage}"] = np.float32(
            crit(logits.detach(), batch["targets"], ratings, stage).item())

    # ---- gradients + one optimizer step (train.py:188-197 semantics) ------
    if with_grads:
        alpha, stage, lr = 0.2, 2, 1e-3
        loss = crit(logits, batch["targets"], ratings, stage)
        total = loss.mean() + alpha * kl.mean()
        total.backward()
        fx["train_total_loss"] = np.float32(total.item())
        named = [(k, p) for k, p in model.named_parameters()]
        gn = torch.nn.utils.clip_grad_norm_([p for _, p in named], 1.0)
        fx["grad_total_norm"] = np.float32(gn.item())  # pre-clip global norm
        fx["grad_keys"] = np.array([k for k, _ in named])
        for k, p in named:
            g = p.grad.detach().numpy()
            idx, val = sample_vec(g)
            fx["gidx_" + k] = idx
            fx["gval_" + k] = val           # values AFTER clipping (what AdamW sees)
            fx["gnorm_" + k] = np.float32(np.linalg.norm(g.astype(np.float64)))
        state = {}
        adamw_hf_step([p for _, p in named], lr, 1, state)
        for k, p in named:
            idx, val = sample_vec(p.detach().numpy())
            fx["pval_" + k] = val
        fx["train_hparams"] = json.dumps({"alpha": alpha, "stage": stage, "lr": lr,
                                          "clip": 1.0, "eps": 1e-6, "wd": 0.0})
        # restore weights for the decode case
        model.load_state_dict({**model.state_dict(),
                               **{k: torch.from_numpy(weights[k].copy()) for k in weights}})

    # ---- greedy decode through sample_sequence (generate.py:97-145) -------
    if with_decode:
        model.train_flag = False
        rec = []
        orig_forward = model.forward

        def rec_forward(inputs):
            out = orig_forward(inputs)
            rec.append(out[2][0, -1, :].detach().clone().numpy())
            return out

        model.forward = rec_forward
        keys = [k for k in batch_np if k != "rating"]
        for length in decode_lengths or (30, 220 if S == 5 else 2 * S * 22):
            for row in decode_rows:
                rec.clear()
                start = {k: np.asarray(batch_np[k][row]) for k in keys}
                start["targets"] = np.asarray([1])
                ids = ref_generate.sample_sequence(
                    model, start, length, StubTokenizer(), temperature=1.1,
                    top_k=1, top_p=0.0, repitition_penalty=1.5, device="cpu")
                fx[f"greedy_len{length}_row{row}"] = np.asarray(ids, np.int64)
                raw = np.stack(rec).astype(np.float32)
                if store_rawlogits:
                    fx[f"greedy_len{length}_row{row}_rawlogits"] = raw
                else:
                    # V = 13317: keep the per-call top-2 margin of the processed logits, the chosen ids and the raw
                    # top-8 (ids, values) instead of 12 MB of logits
                    mg, ch, seq = greedy_margins(raw, 1.1, 1.5, length)
                    assert seq[:len(ids)] == list(ids), "replayed processing disagrees with the reference's ids"
                    fx[f"greedy_len{length}_row{row}_margin"] = mg
                    fx[f"greedy_len{length}_row{row}_chosen"] = ch
                    t8 = np.argsort(-raw, axis=-1)[:, :8]
                    fx[f"greedy_len{length}_row{row}_top8"] = t8.astype(np.int32)
                    fx[f"greedy_len{length}_row{row}_top8_val"] = np.take_along_axis(raw, t8, -1)
        model.forward = orig_forward
        fx["decode_params"] = json.dumps({"temperature": 1.1, "top_k": 1, "top_p": 0.0,
                                          "repitition_penalty": 1.5})

    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    return ref_generate


def case_filtering(out_dir, ref_generate):
    """KATs for top_k_top_p_filtering (generate.py:64-94)."""
    rng = np.random.default_rng(7)
    fx = {}
    cases = [(0, 0.0), (1, 0.0), (5, 0.0), (10, 0.7), (0, 0.9), (30, 0.3), (200, 0.0)]
    logits = rng.standard_normal((len(cases), 97)).astype(np.float32) * 3
    logits[2, 10] = logits[2, 11]  # a tie inside the top-k boundary region
    fx["in"] = logits.copy()
    fx["top_k"] = np.array([c[0] for c in cases])
    fx["top_p"] = np.array([c[1] for c in cases], np.float32)
    outs = []
    for i, (k, p) in enumerate(cases):
        o = ref_generate.top_k_top_p_filtering(torch.from_numpy(logits[i].copy()), top_k=k, top_p=p)
        outs.append(o.numpy())
    fx["out"] = np.stack(outs)
    path = os.path.join(out_dir, "filtering.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path)


def case_dataset(out_dir):
    """MyDataset layout KAT (MyDataset.py:34-118): two synthetic records through the reference class with the
    reference's own vocabulary; the fixture keeps the records' strings, the tokenizer's output for every string
    (so the test needs no vocabulary file) and the resulting arrays."""
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    from transformers import BertTokenizer
    import MyDataset as ref_ds
    tok = BertTokenizer(os.path.join(REF_SRC, "vocab", "vocab.txt"))
    rng = np.random.default_rng(11)

    class Cfg:
        topic_prompt_length, max_sent_length = 15, 20

    lyr_a = ["我站在 夏天的风里", "想起你说过的话", "城市的灯一盏一盏亮起", "谁还在等", "雨落下来\n打湿了回忆", "我们都没有说再见",
             "时间把故事写成一首很长很长很长很长很长很长很长很长的歌谣没有尽头", "梦", "如果还能回到那个夏天", "我会记得抱紧你"]
    lyr_b = ["hello 世界", "好", "春天来了", "花开了", "鸟儿唱", "风轻轻", "云淡淡", "水清清", "山高高", "路长长"]
    recs = []
    for topic, lyr, rating in (("爱情 夏天 回忆", lyr_a, 5), ("这是一个非常非常非常长的主题词列表需要被截断", lyr_b, 2)):
        r = {"topic": topic, "topic_emb": rng.standard_normal(2048).astype(np.float32).tolist(), "lyrics": lyr, "rating": rating}
        for i in range(5):
            for ch in ("text", "img", "r"):
                r["%s_%d" % (ch, i)] = "x"
                r["%s_%d_emb" % (ch, i)] = rng.standard_normal(2048).astype(np.float32).tolist()
        recs.append(r)
    scratch = tempfile.mkdtemp(prefix="mmtg_ds_")
    pk = os.path.join(scratch, "data.pkl")
    with open(pk, "wb") as f:
        pickle.dump(recs, f)
    ds = ref_ds.MyDataset(pk, tok, Cfg(), if_train=True)
    fx = {}
    strings = {}
    for n, r in enumerate(recs):
        item = ds[n]
        for k, v in item.items():
            if "emb" in k:      # pass-through fields: shape and dtype only (the values are the record's own lists)
                fx["item%d_%s_shape" % (n, k)] = np.asarray(np.asarray(v).shape)
            else:
                fx["item%d_%s" % (n, k)] = np.asarray(v)
        strings["主题词：" + r["topic"]] = tok.tokenize("主题词：" + r["topic"])
        for sent in r["lyrics"]:
            for ch in (" ", "\n", "\t", "\r", "\xa0", "\u3000"):
                sent = sent.replace(ch, "")
            strings[sent] = tok.tokenize(sent)
    vocab = {}
    for toks in strings.values():
        for t in toks:
            vocab[t] = tok.convert_tokens_to_ids(t)
    for t in ("[#START#]", "[#EOS#]", tok.pad_token, tok.sep_token):
        vocab[t] = tok.convert_tokens_to_ids(t)
    meta = {"records": [{"topic": r["topic"], "lyrics": r["lyrics"], "rating": r["rating"]} for r in recs],
            "tokenize": strings, "vocab": vocab, "pad_token": tok.pad_token, "sep_token": tok.sep_token,
            "pad_token_id": tok.pad_token_id, "emb_seed": 11}
    fx["meta_json"] = np.array(json.dumps(meta, ensure_ascii=False))
    path = os.path.join(out_dir, "dataset.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def case_postprocess(out_dir, golden_dir):
    """Cut rules + detokenisation of generate.py:222-235.  That code lives inline in the reference's main(); it is
    located in the source file at run time (between the two statements that bracket it), dedented and EXECUTED on
    token lists -- nothing of it is stored here.  Inputs: the reference's own 220-position greedy id list (tiny_s5
    fixture) and hand-made lists that reach every branch; tokens come from the reference's vocabulary."""
    import textwrap
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    from transformers import BertTokenizer
    tok = BertTokenizer(os.path.join(REF_SRC, "vocab", "vocab.txt"))
    lines = open(os.path.join(REF_SRC, "generate.py"), encoding="utf-8").read().split("\n")
    a = next(i for i, l in enumerate(lines) if "all_idx_of_eos = [" in l)
    b = next(i for i, l in enumerate(lines) if "n_preds += [tmp]" in l)
    code = compile(textwrap.dedent("\n".join(lines[a:b])), "<reference generate.py:%d-%d>" % (a + 1, b), "exec")

    def ref_post(tokens):
        ns = {"preds": list(tokens)}
        exec(code, ns)
        return ns["tmp"]

    fx0 = np.load(os.path.join(golden_dir, "tiny_s5.npz"))
    cases = [fx0["greedy_len220_row0"].tolist(), fx0["greedy_len30_row0"].tolist()]
    w = [104 + 7 * i for i in range(40)]
    sent = lambda n, k: [1] + w[k:k + n] + [0] * (20 - n) + [2]
    cases.append(sum((sent(5 + i, i) for i in range(10)), []) + [102])                      # ten sentences then [SEP]
    cases.append(sum((sent(3, i) for i in range(4)), []) + [102] + sent(4, 9))              # early [SEP]: cut there
    cases.append(sum((sent(6, i) for i in range(3)), []))                                   # no [SEP], fewer than ten [#EOS#]
    cases.append(sum((sent(2, i) for i in range(12)), []))                                  # twelve [#EOS#], no [SEP]: cut at the 10th
    cases.append(sum((sent(2, i) for i in range(11)), [])[:-1] + [102, 2])                  # [SEP] before the last [#EOS#]
    cases.append([1, 105, 0, 0, 2, 2, 2])                                                   # trailing commas stripped
    ids = sorted(set(i for c in cases for i in c))
    vocab = {int(i): tok.convert_ids_to_tokens(int(i)) for i in ids}
    fx = {"n": np.int64(len(cases)), "vocab_json": np.array(json.dumps(vocab, ensure_ascii=False))}
    outs = []
    for n, c in enumerate(cases):
        fx["ids_%d" % n] = np.asarray(c, np.int64)
        outs.append(ref_post(tok.convert_ids_to_tokens(c)))
    fx["expected_json"] = np.array(json.dumps(outs, ensure_ascii=False))
    path = os.path.join(out_dir, "postprocess.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, [o[:24] for o in outs])


def case_variants(out_dir):
    """The encoder types / depths MultiModalEncoder accepts besides the released 1-layer GRUs (src/model.py:41-59): the reference's
    own nn.LSTM / nn.RNN(relu) / multi-layer nn.GRU executed on the same synthetic recipe (eval mode: no inter-layer dropout)."""
    case_model("tiny_lstm2_rnn2", out_dir, S=5, n_layer=2, V=160, B=3, seed=400, decode_rows=(0,), decode_lengths=(30,),
               enc=dict(image_type="LSTM", image_layers=2, text_type="RNN", text_layers=2))
    case_model("tiny_gru2_lstm1", out_dir, S=2, n_layer=2, V=160, B=4, seed=500, with_decode=False,
               enc=dict(image_type="GRU", image_layers=2, text_type="LSTM", text_layers=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only-dataset", action="store_true", help="regenerate tests/golden/dataset.npz only")
    ap.add_argument("--only-postprocess", action="store_true", help="regenerate tests/golden/postprocess.npz only")
    ap.add_argument("--only-variants", action="store_true", help="(re)generate the encoder-variant fixtures only")
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if args.only_postprocess:
        case_postprocess(out_dir, os.path.join(REPO, "tests", "golden"))
        return
    if args.only_variants:
        case_variants(out_dir)
        return
    case_dataset(out_dir)
    if args.only_dataset:
        return
    gen = case_model("tiny_s5", out_dir, S=5, n_layer=2, V=160, B=3, seed=100)
    case_filtering(out_dir, gen)
    case_model("tiny_s2", out_dir, S=2, n_layer=2, V=160, B=4, seed=200, with_decode=False)
    case_postprocess(out_dir, out_dir)
    case_variants(out_dir)
    if not args.skip_full:
        # (full size: sampled logits + top-5 + LSE, sampled gradients / gradient norms / parameters after one step,
        #  and one greedy sample_sequence run of 30 and of 220 positions with its per-call top-2 margins)
        case_model("full_12l", out_dir, S=5, n_layer=12, V=13317, B=2, seed=300,
                   full_logits=False, with_grads=True, with_decode=True, decode_rows=(0,), store_rawlogits=False)


if __name__ == "__main__":
    main()
