"""Host-side mirror of the reference's model surface (src/model.py).

Same class names, constructor / forward signatures, return tuple and state-dict
keys as the reference so that its drivers (src/train.py:106,188,212 and
src/generate.py:124,192) can switch over, but NO arithmetic happens here: the
nn.Module tree only *holds* parameters (as views into one flat fp32 buffer) and
``forward`` hands the batch to the HIP engine (mmtg_amd.engine).  On a machine
without the extension / without an MI355X ``forward`` raises -- there is no
PyTorch fallback.
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .configs import GPT2_BASE, GPT2_PDROP, rnn_param_shapes
from .engine import Engine, ParamLayout, Shapes


class _Holder(nn.Module):
    """Parameter container; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the computation runs in mmtg_amd.engine")


class MultiModalEncoder(_Holder):
    """topic_fc + rnns_image + rnns_text parameters (reference model.py:24-88)."""


class InnerModalAttentionLayer(_Holder):
    """query/key/value projections of the alpha attention (model.py:91-161)."""


class MultiModalAttentionLayer(_Holder):
    """att_matrices[i] + out_linear of the beta attention (model.py:164-202)."""


def _attach(root, dotted, param):
    mod = root
    parts = dotted.split(".")
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, _Holder())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], param)


def _load_gpt2_config(config_path, override):
    cfg = dict(GPT2_BASE)
    cfg.update(GPT2_PDROP)
    if config_path and os.path.exists(config_path):
        with open(config_path) as f:
            cfg.update(json.load(f))
    if override:
        cfg.update(override)
    return cfg


PACKED_TABLE_PATH = "./vocab/token_id2emb.safetensors"


def table_from_dict(table):
    """{id: list[emb]} (model.py:215) -> dense float32 [max id + 1, emb]; ids missing from the dict stay zero."""
    n = max(int(k) for k in table) + 1
    ids = np.fromiter((int(k) for k in table), dtype=np.int64, count=len(table))
    arr = np.zeros((n, len(next(iter(table.values())))), np.float32)
    arr[ids] = np.asarray(list(table.values()), np.float32)
    return arr


def pack_token_table(table, out_path=PACKED_TABLE_PATH, dtype=torch.bfloat16):
    """One-time conversion of the reference's ``vocab/token_id2emb_dict.pkl`` (a pickled ``{id: list[2048 floats]}``, 246 MB,
    ~14 s to unpickle) into a flat ``[V, 2048]`` tensor file (safetensors, 54.5 MB in bf16, memory-mapped on load).
    ``table``: the pickle's path, the dict itself, or a [V, emb] array."""
    from safetensors.torch import save_file
    if isinstance(table, (str, os.PathLike)):
        with open(table, "rb") as f:
            table = pickle.load(f)
    arr = table_from_dict(table) if isinstance(table, dict) else np.asarray(table, np.float32)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype).contiguous()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    save_file({"token_id2emb": t}, str(out_path), metadata={"format": "mmtg-token-table-1", "rows": str(t.shape[0]), "emb": str(t.shape[1])})
    return out_path


def load_token_table(path):
    """[V, emb] tensor of a ``pack_token_table`` file (stored dtype, normally bf16)."""
    from safetensors import safe_open
    with safe_open(str(path), framework="pt") as f:
        if (f.metadata() or {}).get("format") != "mmtg-token-table-1":
            raise ValueError("not a packed token table: %s" % path)
        return f.get_tensor("token_id2emb")


class GPT2_Decoder(_Holder):
    """projector + GPT-2 parameters and the WenLan table (reference model.py:205-223).

    The reference fetches the architecture from the HF hub (model.py:219) and
    ignores its own config file (:214); offline the architecture comes from
    ``config_path`` (default config/model_config.json, falling back to the
    released values) or ``gpt2_config=``.
    """

    def __init__(self, data_config, model_name="uer/gpt2-chinese-cluecorpussmall",
                 config_path="config/model_config.json", gpt2_config=None, token_table=None):
        super().__init__()
        self.data_config = data_config
        self.model_name = model_name
        self.config = _load_gpt2_config(config_path, gpt2_config)
        self.token_id2emb = None
        if token_table is None:
            # the packed tensor file (pack_token_table) is preferred over the reference's 246 MB pickle of Python lists
            if os.path.exists(PACKED_TABLE_PATH):
                token_table = load_token_table(PACKED_TABLE_PATH)
            elif os.path.exists("./vocab/token_id2emb_dict.pkl"):
                token_table = self.load_token_id2emb("./vocab/token_id2emb_dict.pkl")
        self._table = None
        if token_table is not None:
            self.set_token_table(token_table)

    def load_token_id2emb(self, path):
        with open(path, "rb") as f:
            return pickle.load(f)

    def set_token_table(self, table):
        """Accepts the reference's ``{id: list[2048]}`` dict, a [V, 2048] array / tensor (any float dtype; a bf16 tensor
        from ``load_token_table`` is kept as is), or the path of a packed table file."""
        if isinstance(table, (str, os.PathLike)):
            table = load_token_table(table)
        if isinstance(table, dict):
            self.token_id2emb = table
            table = table_from_dict(table)
        t = table if torch.is_tensor(table) else torch.as_tensor(np.asarray(table))
        self._table = t if t.dtype == torch.bfloat16 else t.float()


class MMTG(nn.Module):
    """Drop-in for the reference's ``MMTG`` (model.py:330-400).

    Extra keyword arguments (all optional, the positional contract is unchanged):
      gpt2_config   dict overriding config/model_config.json
      token_table   WenLan table (dict or [V,2048]); default ./vocab/token_id2emb_dict.pkl
      compute_dtype 'bf16' (default; bf16 storage, fp32 accumulate), 'f32' (exact fp32 MFMA), 'bf16x3' (fp32 storage,
                    the GPT-2 / lm_head products as three bf16 matrix-core passes over (hi | lo) split operands: the
                    fp32 mode's parity at a multiple of its speed) or 'bf16x3f' (round 6: bf16x3's forward -- logits, loss,
                    KL and greedy ids at the fp32 mode's parity -- with the backward as ONE bf16 pass over the hi planes the
                    forward stored: gradients at the bf16 mode's accuracy)
    """

    def __init__(self, model_cfgs, data_config, vocab_size, train_flag=False, gpt2_config=None,
                 token_table=None, compute_dtype=None, config_path="config/model_config.json"):
        super().__init__()
        self.model_cfgs = model_cfgs
        self.data_config = data_config
        self.vocab_size = vocab_size
        self.train_flag = train_flag
        for ch in ("image", "text"):         # reference model.py:41-59: RNN(relu) / LSTM / GRU, num_layers >= 1
            rnn_param_shapes(model_cfgs, ch)
        assert model_cfgs["topic"]["hidden_dim"] == model_cfgs["image"]["hidden_dim"] == model_cfgs["text"]["hidden_dim"], \
            "The hidden dim of topic, image and text must be equal."
        compute_dtype = compute_dtype or os.environ.get("MMTG_DTYPE", "bf16")
        self.compute_dtype = {"bf16": hip.BF16, "f32": hip.F32, "fp32": hip.F32, "bf16x3": hip.F32, "bf16x3f": hip.F32}[compute_dtype]
        self.x3 = compute_dtype in ("bf16x3", "bf16x3f")
        self.hybrid = compute_dtype == "bf16x3f"      # split-precision forward (the reference's outputs to 1e-3), ONE bf16 pass backward

        self.encoder = MultiModalEncoder()
        self.ln_layer1 = _Holder()
        self.ln_layer2 = _Holder()
        self.ln_layer3 = _Holder()
        self.img_inner_atten_layer = InnerModalAttentionLayer()
        self.text_inner_atten_layer = InnerModalAttentionLayer()
        self.mm_atten_layer = MultiModalAttentionLayer()
        self.decoder = GPT2_Decoder(data_config, config_path=config_path, gpt2_config=gpt2_config,
                                    token_table=token_table)
        self.gpt2_cfg = self.decoder.config
        self.layout = ParamLayout(model_cfgs, self.gpt2_cfg)
        self.shapes = Shapes(model_cfgs, data_config, self.gpt2_cfg)
        if self.x3 and self.shapes.D % 128:
            raise ValueError("compute_dtype='bf16x3' needs n_embd to be a multiple of 128 (got %d): the split-precision products walk "
                             "whole 128-wide K tiles and the mode has no mixed fallback; use compute_dtype='f32'" % self.shapes.D)
        self._flat = torch.zeros(self.layout.total, dtype=torch.float32)
        self._engine = None
        self._anchor = None
        self._params = {}
        for key in self.layout.keys:
            p = nn.Parameter(self.layout.view(self._flat, key))
            self._params[key] = p
            _attach(self, key, p)
        # lm_head is tied to wte (one Parameter, two names) as in GPT2LMHeadModel
        _attach(self, "decoder.gpt2.lm_head.weight", self._params["decoder.gpt2.transformer.wte.weight"])
        self.reset_parameters()
        if train_flag:
            path = model_cfgs.get("GPT2_PATH")
            if path and os.path.exists(path):
                # Load pre-trained GPT2 (keys relative to GPT2_Decoder; optional Lightning
                # 'state_dict' wrapper) -- reference model.py:345-354
                print("Loading pre-trained GPT2 model...")
                sd = torch.load(path, map_location="cpu")
                if "state_dict" in sd:
                    sd = dict(sd["state_dict"])
                self.load_state_dict({"decoder." + k: v for k, v in sd.items()}, strict=False)
                print("Pre-trained GPT2 model loaded.")

    # ------------------------------------------------------------------ init / device moves
    @torch.no_grad()
    def reset_parameters(self, seed=None):
        """Reference initialisation: xavier_normal for topic_fc / W_ih, orthogonal W_hh
        (model.py:83-88), PyTorch defaults elsewhere, GPT-2 N(0, 0.02)."""
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(seed)
        else:
            g.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        self._flat.zero_()
        std = self.gpt2_cfg.get("initializer_range", 0.02)
        H = self.shapes.H
        for key, p in self._params.items():
            if ".gpt2." in key:
                if key.endswith("ln_1.weight") or key.endswith("ln_2.weight") or key.endswith("ln_f.weight"):
                    p.fill_(1.0)
                elif key.endswith(".bias"):
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * std)
            elif key.startswith("ln_layer"):
                p.fill_(1.0) if key.endswith("weight") else p.zero_()
            elif key.endswith("topic_fc.weight") or key.endswith("weight_ih_l0"):
                fan_out, fan_in = p.shape
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (fan_in + fan_out)) ** 0.5)
            elif key.endswith("weight_hh_l0"):
                q, r = torch.linalg.qr(torch.randn(p.shape, generator=g))
                p.copy_(q * torch.sign(torch.diagonal(r)).unsqueeze(0))
            elif "rnns_" in key:   # GRU biases: U(-1/sqrt(H), 1/sqrt(H))
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / H ** 0.5)
            else:                   # nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                w = self._params[key[:-4] + "weight"] if key.endswith("bias") else p
                fan_in = w.shape[-1]
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / fan_in ** 0.5)
        self._invalidate()

    def _invalidate(self):
        if self._engine is not None:
            self._engine.invalidate_copies()

    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("MMTG keeps fp32 master parameters; choose the compute dtype with compute_dtype=")
        if new_flat is not self._flat:
            self._flat = new_flat
            for key, p in self._params.items():
                p.data = self.layout.view(self._flat, key)
                p.grad = None
            self._engine = None
        return self

    def set_token_table(self, table):
        self.decoder.set_token_table(table)
        if self._engine is not None:
            self._engine.set_table(self.decoder._table)

    # ------------------------------------------------------------------ state dict compatibility
    def load_state_dict(self, state_dict, strict=True):
        """Accepts checkpoints saved from nn.DataParallel ('module.' prefix, train.py:113,212)
        and from transformers 4.12.3 (extra attn.bias / attn.masked_bias buffers)."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("module."):
                k = k[len("module."):]
            if k.endswith(".attn.bias") or k.endswith(".attn.masked_bias"):
                continue
            sd[k] = v
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            if not (missing == ["decoder.gpt2.lm_head.weight"] and not unexpected):
                raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing, unexpected))
        with torch.no_grad():
            for k, v in sd.items():
                if k in self._params:
                    self._params[k].copy_(torch.as_tensor(v).to(self._flat.device, torch.float32))
                elif k == "decoder.gpt2.lm_head.weight" and "decoder.gpt2.transformer.wte.weight" not in sd:
                    self._params["decoder.gpt2.transformer.wte.weight"].copy_(torch.as_tensor(v).to(self._flat.device, torch.float32))
        self._invalidate()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def legacy_state_dict(self):
        """state_dict() plus the per-block causal-mask buffers transformers 4.12.3 persisted."""
        sd = dict(self.state_dict())
        NP = self.shapes.NP
        for l in range(self.shapes.L):
            p = f"decoder.gpt2.transformer.h.{l}.attn."
            sd[p + "bias"] = torch.tril(torch.ones(NP, NP, dtype=torch.uint8)).view(1, 1, NP, NP)
            sd[p + "masked_bias"] = torch.tensor(-1e4)
        return sd

    # ------------------------------------------------------------------ engine access
    def engine(self):
        if self._engine is None:
            if not self._flat.is_cuda:
                raise RuntimeError("MMTG.forward needs the model on an MI355X: call model.to('cuda'). "
                                   "The hot path is HIP-only (no CPU fallback).")
            self._engine = Engine(self.model_cfgs, self.data_config, self.gpt2_cfg, self._flat,
                                  self.decoder._table, self.compute_dtype, x3=self.x3, hybrid=self.hybrid)
            self._anchor = torch.zeros((), device=self._flat.device, requires_grad=True)
        return self._engine

    def zero_grad(self, set_to_none=True):
        if self._engine is not None and self._engine.grad is not None:
            self._engine.grad.zero_()
        for p in self._params.values():
            p.grad = None

    def _attach_grads(self):
        eng = self._engine
        for key, p in self._params.items():
            p.grad = eng.G(key)

    def _grads_attached(self):
        eng = self._engine
        if eng.grad is None:
            return False
        p = self._params["decoder.gpt2.transformer.ln_f.weight"]
        return p.grad is not None and p.grad.data_ptr() == eng.G("decoder.gpt2.transformer.ln_f.weight").data_ptr()

    # ------------------------------------------------------------------ forward
    def forward(self, batch):
        """-> (lm_loss, kl, logits[B,T,V]) exactly as model.py:356-400."""
        eng = self.engine()
        eng.invalidate_copies()   # an external optimizer may have stepped the fp32 masters
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._params.values())
        if need_grad:
            return _MMTGFunction.apply(self._anchor, self, batch)
        return self._run_forward(batch)

    def _run_forward(self, batch):
        eng = self._engine
        a = eng.forward(batch, train_flag=self.train_flag, training=self.training, per_row_infer=True)
        sc = eng.loss(None, label_zero=not self.train_flag)
        B, T = a["B"], a["T"]
        logits = a["logits"].view(B, T, -1)[:, :, :self.shapes.V]
        return sc[1], a["kl"][0], logits


class _MMTGFunction(torch.autograd.Function):
    """Autograd boundary of the drop-in path: forward/backward both run in the HIP engine;
    parameter gradients land in the engine's flat buffer and are exposed as ``param.grad`` views."""

    @staticmethod
    def forward(ctx, anchor, model, batch):
        ctx.model = model
        lm, kl, logits = model._run_forward(batch)
        ctx.act = model._engine.act
        return lm.clone(), kl.clone(), logits

    @staticmethod
    def backward(ctx, d_lm, d_kl, d_logits):
        model = ctx.model
        eng = model._engine
        if eng.act is not ctx.act:
            raise RuntimeError("MMTG: backward through a forward that is no longer the engine's latest one")
        if not model._grads_attached():
            eng.zero_grad()
        a = eng.act
        if d_logits is None:
            dl = eng.buf("dlogits", (a["M"], eng.layout.Vpad), zero=True)
        else:
            dl = eng.dlogits_from(d_logits)
        if d_lm is not None and bool((d_lm != 0).any()):
            raise NotImplementedError("gradient of GPT-2's internal LM loss: use MMTGTrainer(lm_weight=...) "
                                      "(the reference discards this loss, train.py:188)")
        dkl = 0.0 if d_kl is None else float(d_kl)
        eng.backward(dl, dkl)
        model._attach_grads()
        return None, None, None
