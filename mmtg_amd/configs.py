"""Constructor contract of the hot path: ``model_cfgs`` and ``data_config``.

Mirrors the surface of the reference's ``src/configs.py:14-54`` (a plain dict
plus a subscriptable object) so that ``MMTG(model_cfgs, data_config, ...)``
accepts exactly what the reference drivers pass.  Unlike the reference the
released shape (5 experience steps, 20-token sentences, 15-token prompt) is a
default, not a hard-coded constant: every size can be overridden.
"""
from __future__ import annotations

import copy


def make_model_cfgs(seq_len=5, wenlan_dim=2048, hidden=512, heads=4,
                    gpt2_path="./pretrained/GPT2_lyrics_ckpt_epoch00.ckpt",
                    dropout=0.1, image_type="GRU", image_layers=1, text_type="GRU", text_layers=1):
    """Build a ``model_cfgs`` dict with the reference's key layout.  Channel types: 'GRU' (released), 'LSTM', 'RNN'
    (reference src/model.py:41-59)."""
    chan = lambda t, n: {"type": t, "input_dim": wenlan_dim, "hidden_dim": hidden, "num_layers": n}
    return {
        "seq_len": seq_len,
        "topic": {"input_dim": wenlan_dim, "hidden_dim": hidden},
        "image": chan(image_type, image_layers),
        "text": chan(text_type, text_layers),
        "SELF_ATT": {"hidden_size": hidden, "attention_heads": heads},
        "MM_ATT": {"attention_dim": 1},
        "GPT2_PATH": gpt2_path,
        "dropout": dropout,
    }


RNN_GATES = {"GRU": 3, "LSTM": 4, "RNN": 1}     # rows of weight_ih / weight_hh per hidden unit (torch.nn.RNNBase)


def rnn_param_shapes(model_cfgs, ch):
    """[(suffix, shape)] of the recurrent parameters of channel ``ch`` ('image' / 'text') in torch.nn.RNNBase naming, layer by layer."""
    c = model_cfgs[ch]
    if c["type"] not in RNN_GATES:
        raise ValueError("encoder channel type %r (reference src/model.py:41-59 knows RNN, LSTM, GRU)" % (c["type"],))
    if int(c["num_layers"]) < 1:
        raise ValueError("num_layers must be >= 1")
    G, H, E = RNN_GATES[c["type"]], c["hidden_dim"], c["input_dim"]
    out = []
    for l in range(int(c["num_layers"])):
        out.append([("weight_ih_l%d" % l, (G * H, E if l == 0 else H)), ("weight_hh_l%d" % l, (G * H, H)),
                    ("bias_ih_l%d" % l, (G * H,)), ("bias_hh_l%d" % l, (G * H,))])
    return out


#: released configuration (reference src/configs.py:14-41)
model_cfgs = make_model_cfgs()


class data_config:
    """Subscriptable shape record (reference src/configs.py:43-54).

    ``cfg['name']`` returns the attribute, or ``None`` (after printing a
    notice) for unknown names -- the reference swallows missing keys the same
    way.
    """

    def __init__(self, topic_prompt_length=15, max_sent_length=20,
                 max_seq_length=None, wenlan_emb_size=2048, seq_len=5):
        self.topic_prompt_length = topic_prompt_length
        self.max_sent_length = max_sent_length
        if max_seq_length is None:
            max_seq_length = 2 * seq_len * (max_sent_length + 2)
        self.max_seq_length = max_seq_length
        self.wenlan_emb_size = wenlan_emb_size

    def __getitem__(self, key):
        try:
            return getattr(self, key)
        except AttributeError:
            print("No {} exists!".format(key))
            return None


#: default GPT-2 hyper-parameters (reference src/config/model_config.json:1-10)
GPT2_BASE = {
    "initializer_range": 0.02,
    "layer_norm_epsilon": 1e-05,
    "n_ctx": 250,
    "n_embd": 768,
    "n_head": 12,
    "n_layer": 12,
    "n_positions": 1024,
    "vocab_size": 13317,
}

#: pdrop of the three GPT-2 dropout sites (transformers GPT2Config defaults,
#: which the reference inherits: embd / attn / resid = 0.1 each)
GPT2_PDROP = {"embd_pdrop": 0.1, "attn_pdrop": 0.1, "resid_pdrop": 0.1}


def gpt2_config(**over):
    cfg = copy.deepcopy(GPT2_BASE)
    cfg.update(GPT2_PDROP)
    cfg.update(over)
    return cfg
