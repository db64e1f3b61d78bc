"""Explicit forward/backward engine of the MMTG hot path on MI355X.

No autograd, no tracing compiler: the training step is a fixed sequence of
C-ABI kernel launches (mmtg_amd.hip) over caller-owned HBM buffers.  The
forward keeps the activations the backward needs (288 GB of HBM3E: nothing is
recomputed except attention probabilities), the backward walks the layers in
reverse and deposits parameter gradients into ONE flat fp32 buffer whose
element order is the order in which gradients become final -- so contiguous
slices of it are the all-reduce buckets of the data-parallel path
(mmtg_amd.ddp) and can be launched while the backward is still running.

Reference semantics (file:line into /root/reference/src):
  MMTG.forward model.py:356-400, GPT2_Decoder.forward :225-327, the GPT-2
  arithmetic of transformers 4.12.3 behind model.py:282-288, MyLoss
  loss.py:45-74.  The oracle (oracle/mmtg_oracle.py) restates the same maths
  and is what the tests compare this engine against.
"""
from __future__ import annotations

import contextlib
import math

import numpy as np
import torch

from . import hip
from .configs import rnn_param_shapes

ALIGN = 64  # elements; keeps every pack 16-byte aligned in both f32 and bf16


# --------------------------------------------------------------------------
# flat parameter layout
# --------------------------------------------------------------------------
class ParamLayout:
    """Flat layout of all parameters in gradient-ready order.

    ``packs`` are contiguous runs (no inner padding) so that e.g. the three
    alpha-attention projections form one [3H, H] GEMM operand."""

    def __init__(self, model_cfgs, gpt2_cfg):
        S = model_cfgs["seq_len"]
        E = model_cfgs["topic"]["input_dim"]
        H = model_cfgs["topic"]["hidden_dim"]
        D = gpt2_cfg["n_embd"]
        V = gpt2_cfg["vocab_size"]
        NP = gpt2_cfg["n_positions"]
        L = gpt2_cfg["n_layer"]
        self.Vpad = (V + 127) // 128 * 128
        pre = "decoder.gpt2.transformer."
        packs = []  # (pack_name, [(key, shape)], pad_tail_elems)
        packs.append(("ln_f.w", [(pre + "ln_f.weight", (D,))], 0))
        packs.append(("ln_f.b", [(pre + "ln_f.bias", (D,))], 0))
        self.layer_first_pack = {}
        for l in range(L - 1, -1, -1):
            p = f"{pre}h.{l}."
            self.layer_first_pack[l] = len(packs)
            for nm, shp in (("mlp.c_proj.weight", (4 * D, D)), ("mlp.c_proj.bias", (D,)),
                            ("mlp.c_fc.weight", (D, 4 * D)), ("mlp.c_fc.bias", (4 * D,)),
                            ("ln_2.weight", (D,)), ("ln_2.bias", (D,)),
                            ("attn.c_proj.weight", (D, D)), ("attn.c_proj.bias", (D,)),
                            ("attn.c_attn.weight", (D, 3 * D)), ("attn.c_attn.bias", (3 * D,)),
                            ("ln_1.weight", (D,)), ("ln_1.bias", (D,))):
                packs.append((p + nm, [(p + nm, shp)], 0))
        packs.append(("wte", [(pre + "wte.weight", (V, D))], (self.Vpad - V) * D))
        packs.append(("wpe", [(pre + "wpe.weight", (NP, D))], 0))
        for nm, shp in (("decoder.projector_layer2.weight", (D, H)), ("decoder.projector_layer2.bias", (D,)),
                        ("decoder.projector_layer1.weight", (H, E)), ("decoder.projector_layer1.bias", (H,)),
                        ("mm_atten_layer.out_linear.weight", (E, H)), ("mm_atten_layer.out_linear.bias", (E,))):
            packs.append((nm, [(nm, shp)], 0))
        packs.append(("att_w", [(f"mm_atten_layer.att_matrices.{i}.weight", (1, H)) for i in range(S)], 0))
        packs.append(("att_b", [(f"mm_atten_layer.att_matrices.{i}.bias", (1,)) for i in range(S)], 0))
        for mod in ("text", "img"):
            m = f"{mod}_inner_atten_layer."
            packs.append((mod + "_qkv_w", [(m + n + ".weight", (H, H)) for n in ("query", "key", "value")], 0))
            packs.append((mod + "_qkv_b", [(m + n + ".bias", (H,)) for n in ("query", "key", "value")], 0))
        for i in (3, 2, 1):
            packs.append((f"ln_layer{i}.weight", [(f"ln_layer{i}.weight", (H,))], 0))
            packs.append((f"ln_layer{i}.bias", [(f"ln_layer{i}.bias", (H,))], 0))
        for ch in ("text", "image"):
            r = f"encoder.rnns_{ch}."
            for layer in reversed(rnn_param_shapes(model_cfgs, ch)):     # the top layer's gradients are ready first
                for nm, shp in layer:
                    packs.append((r + nm, [(r + nm, shp)], 0))
        packs.append(("encoder.topic_fc.weight", [("encoder.topic_fc.weight", (H, E))], 0))
        packs.append(("encoder.topic_fc.bias", [("encoder.topic_fc.bias", (H,))], 0))

        self.entries = {}      # key -> (offset, shape, numel)
        self.pack_range = {}   # pack name -> (offset, numel incl. tail pad)
        self.pack_order = []
        off = 0
        for name, members, tail in packs:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            start = off
            for key, shape in members:
                n = int(np.prod(shape))
                self.entries[key] = (off, tuple(shape), n)
                off += n
            off += tail
            self.pack_range[name] = (start, off - start)
            self.pack_order.append(name)
        self.total = (off + ALIGN - 1) // ALIGN * ALIGN
        self.keys = [k for _, members, _ in packs for k, _ in members]

    def view(self, flat, key):
        off, shape, n = self.entries[key]
        return flat[off:off + n].view(shape)

    def pack(self, flat, name):
        off, n = self.pack_range[name]
        return flat[off:off + n]

    def buckets(self, bucket_elems, split_after=()):
        """Contiguous [start, end) ranges of ~bucket_elems elements on pack boundaries,
        in gradient-ready order.  `split_after`: pack names after which a bucket ends whatever its size -- the points of the
        backward's TAIL at which a run of gradients is final (the tied embedding + wpe once the input-embedding backward has
        run, the projector / fuser after the fuser's backward): what is left for the exchange after the last kernel is then
        the encoder's gradients only, not everything since the last full bucket."""
        out, start = [], 0
        forced = set(split_after)
        for name in self.pack_order:
            o, n = self.pack_range[name]
            end = o + n
            if end - start >= bucket_elems or (name in forced and end > start):
                out.append((start, end))
                start = end
        if start < self.total:
            out.append((start, self.total))
        return out


class Shapes:
    def __init__(self, model_cfgs, data_cfg, gpt2_cfg):
        self.S = model_cfgs["seq_len"]
        self.E = model_cfgs["topic"]["input_dim"]
        self.H = model_cfgs["topic"]["hidden_dim"]
        self.heads = model_cfgs["SELF_ATT"]["attention_heads"]
        self.P = data_cfg["topic_prompt_length"]
        self.msl = data_cfg["max_sent_length"]
        self.max_seq_length = data_cfg["max_seq_length"]
        self.two_sents = 2 * (self.msl + 2)
        self.D = gpt2_cfg["n_embd"]
        self.nH = gpt2_cfg["n_head"]
        self.L = gpt2_cfg["n_layer"]
        self.V = gpt2_cfg["vocab_size"]
        self.NP = gpt2_cfg["n_positions"]
        self.eps = gpt2_cfg.get("layer_norm_epsilon", 1e-5)
        self.pdrop = (gpt2_cfg.get("embd_pdrop", 0.1), gpt2_cfg.get("attn_pdrop", 0.1), gpt2_cfg.get("resid_pdrop", 0.1))
        # encoder channels (model.py:39-59): cell type, layers; dropout between the layers of a channel in training (nn.RNNBase)
        self.rnn = {ch: (model_cfgs[ch]["type"], int(model_cfgs[ch]["num_layers"])) for ch in ("image", "text")}
        self.rnn_pdrop = float(model_cfgs.get("dropout", 0.0))
        for ch in ("image", "text"):
            rnn_param_shapes(model_cfgs, ch)         # validates type / num_layers
            if model_cfgs[ch]["input_dim"] != self.E or model_cfgs[ch]["hidden_dim"] != self.H:
                raise ValueError("The hidden dim of topic, image and text must be equal (model.py:36); this engine also takes one input dim")
        if self.D % self.nH or self.D // self.nH != 64:
            raise ValueError("the attention kernels are built for head dim 64 (n_embd=%d, n_head=%d)" % (self.D, self.nH))
        if model_cfgs["MM_ATT"]["attention_dim"] != 1:
            raise ValueError("MM_ATT.attention_dim must be 1 (the reference's broadcast at model.py:200 requires it)")


def gaussian_prior(S):
    """q_i[j] ~ N(j; i, 1) normalised over j < S (model.py:116-120, 5 generalised to S)."""
    j = np.arange(S, dtype=np.float64)
    rows = []
    for i in range(S):
        v = np.exp(-0.5 * (j - i) ** 2) / math.sqrt(2 * math.pi)
        v32 = v.astype(np.float64)
        rows.append((v32 / v32.sum()).astype(np.float32))
    return np.stack(rows)


import os as _os


def _dist_rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else int(_os.environ.get("RANK", "0"))


_NO_OCC4 = bool(_os.environ.get("MMTG_GEMM_NO_OCC4"))     # A/B switch, mirrors the library's
_WG_NUM = float(_os.environ.get("MMTG_WGRAD_NUM", "760"))     # tuning knobs of the slab weight-gradient split count
_WG_CAP = int(_os.environ.get("MMTG_WGRAD_CAP", "12"))
_NO_FEW_ROWS = bool(_os.environ.get("MMTG_NO_FEW_ROWS"))   # A/B switch: plain launches for the encoder-sized products
_WGRAD_SLAB = not _os.environ.get("MMTG_WGRAD_ATOMIC")     # A/B switch: fp32-atomic weight gradients everywhere
_NO_GATHER = bool(_os.environ.get("MMTG_NO_GATHER"))       # A/B switch: materialise the conditioned embeddings (round-1 path)
_PREFETCH = int(_os.environ.get("MMTG_PREFETCH", "0"))      # backward: Infinity-Cache prefetch of saved activations on a side stream (workgroups; 0 = off)


def _wgrad_splits(M, N, K, occ4=False, slots=512, t_iter=1.1, t_fixed=6.0):
    """Split-K factor of a weight-gradient GEMM (K = tokens).  The 128x128-tile kernel keeps two
    workgroups per CU (512 slots on 256 CUs); a grid of tiles*s workgroups runs in
    ceil(tiles*s/512) rounds of (K/(64 s)) K-iterations each.  Pick the s with the smallest
    estimate -- e.g. 144 tiles: s=3 (432 WGs, one round) beats s=4 (576 WGs, two rounds) by 30 %
    on the GPU (profiles/r01_gemm_tn_split_sweep.log)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if occ4 and not _NO_OCC4:
        # bf16: the single-stage kernel keeps FOUR workgroups per CU (1024 slots) and is throughput-bound,
        # so only the fill of the last round and the atomic volume matter.  Measured optima
        # (profiles/r01_v6_gemm_tn_split_sweep.log): 144 tiles -> 5, 108 -> 6..7, 630 -> 4, 36 -> 8+.
        smax = max(1, K // 256)
        s = round(_WG_NUM / tiles) if tiles < 512 else round(2520.0 / tiles)
        return int(max(1, min(smax, s, _WG_CAP)))
    best, best_t = 1, None
    for s in range(1, 33):
        if s > 1 and K // s < 256:
            break
        rounds = -(-tiles * s // slots)
        # + the fp32 atomic epilogue: 64 KB of adds per workgroup at the chip-wide ~1.3 TB/s atomic rate
        t = rounds * (K / (64.0 * s) * t_iter + t_fixed) + tiles * s * 65536 / 1.3e6
        if best_t is None or t < best_t * 0.97:
            best, best_t = s, t
    return best


_LAZY_ZERO = _os.environ.get("MMTG_FULL_ZERO_GRAD") is None   # zero only the accumulated gradients once a step shape is known (A/B switch)
_WTE_T = _os.environ.get("MMTG_NO_WTE_T") is None      # [D, Vpad] copy of wte for the LM head's dgrad (A/B switch)
_P8T = _os.environ.get("MMTG_GEMM_P8T", "0") != "0"      # eight-phase K-strided kernel for the slab weight gradients (opt-in: measured slower in situ)
_WGRAD_GROUP = _os.environ.get("MMTG_WGRAD_GROUP", "1") != "0"   # one grouped launch per GPT-2 block for its four weight gradients (A/B switch)
_WGRAD_GROUP_SPLITS = int(_os.environ.get("MMTG_WGRAD_GROUP_SPLITS", "0"))    # 0 = the one-round rule below
# fc1's epilogue saves gelu'(pre-activation) instead of the pre-activation (MMTG_GEMM_GELU_GRAD): OPT-IN.  Same box: 15.11 -> 15.03 ms
# per step, but the tiny, cancellation-dominated gradient of mm_atten_layer.att_matrices.3.weight moves from 1.05x to 1.18x of the
# oracle's norm on the 2-layer golden model (every other tensor within 1 %): outside the parity suite's 10 % bound, so not the default
_GELU_GRAD = _os.environ.get("MMTG_GELU_GRAD", "0") != "0"
# the bf16 decoder backward's small ordered column sums (LayerNorm second stages, dGELU bands, attention bias rows: 4 per block) summed
# by ONE batched launch per data-parallel hand-over point -- one per step without a bucket hook -- instead of a launch each (A/B switch)
_DEFER_SUMS = _os.environ.get("MMTG_DEFER_SUMS", "1") != "0"
# the grouped weight gradients of the LAST blocks the backward walks (blocks MMTG_WGRAD_TAIL - 1 .. 0) on a side stream beside the
# backward's tail -- the fuser / encoder backward: ~60 small dependent launches during which the GPU is mostly idle -- instead of inside
# the block loop; joined before the gradient norm.  Single-GPU steps with dropout on (their operands are the masked copies, which get
# buffers of their own: 209 MB per block at GPT-2 base); 0 = off.  Measured (same box, ms per step): 0: 14.91, 1-2: 14.9-15.0 (the
# co-running launches slow the tail as much as they hide), 4: 14.72, 6: 14.66, 8: 14.64, 12: 14.67; the same launches at the same
# place on the MAIN stream (MMTG_WGRAD_TAIL_MAIN=1): 14.98 -- the gain is the overlap, not the order.  bf16x3 (plane-pair operands,
# with or without dropout): 33.57 -> 33.07 at 8 (33.03 at 4, 33.29 at 12)
_WGRAD_TAIL = int(_os.environ.get("MMTG_WGRAD_TAIL", "8"))
# where the side stream forks: "loop" = right after the block loop (beside the embedding / projector backward too: 14.28 -> 14.10 ms on
# top of the above), "proj" = after the projector backward (A/B)
_TAIL_FORK_EARLY = _os.environ.get("MMTG_WGRAD_TAIL_FORK", "loop") == "loop"
_LMHEAD_GROUP = _os.environ.get("MMTG_LMHEAD_GROUP", "1") != "0"      # the tied embedding's weight gradient through the grouped kernel (A/B switch)
_LMHEAD_GROUP_SPLITS = int(_os.environ.get("MMTG_LMHEAD_GROUP_SPLITS", "0"))
# the grouped weight-gradient launches on a SIDE stream, one block behind the dgrad chain (MMTG_WGRAD_STREAM; see Engine.backward)
_WGRAD_STREAM = _os.environ.get("MMTG_WGRAD_STREAM", "0") != "0"
# x3 grouped weight gradients: config 6 = combined stages (all four planes of a K tile per 64 KB stage, two workgroups per CU: 512
# slots), MMTG_WGRAD_X3_COMBINED=0 = config 2, three passes over 32 KB stages at four workgroups per CU (A/B switch)
# Round 6 (MMTG_ENC_STREAMS=1, opt-in: measured NEUTRAL -- 14.73 / 14.72 ms per step without, 14.71 / 14.72 with, same box, all gradient /
# reproducibility tests green under it): the two encoder channels (image / text: independent chains of ~20 tiny dependent launches each
# way) side by side -- the text channel on a second stream with its own workspaces -- forward and backward
_ENC_STREAMS = _os.environ.get("MMTG_ENC_STREAMS", "0") != "0"
# bf16x3f: the encoder / fuser backward on the bf16 kernels as well (MMTG_HYBRID_ENC_BF16=0: exact fp32, the round's first version)
_HYBRID_ENC_BF16 = _os.environ.get("MMTG_HYBRID_ENC_BF16", "1") != "0"
_X3_WG_CFG = 6 if _os.environ.get("MMTG_WGRAD_X3_COMBINED", "1") != "0" else 2
_WGRAD_GROUP_CFG = int(_os.environ.get("MMTG_WGRAD_GROUP_CFG", "0"))          # 0: 128x128 tiles, four workgroups per CU; 1: 256x256 eight-phase


def _group_splits(tiles, K, slots=1024):
    """K splits of a grouped weight-gradient launch: tiles x splits workgroups in ONE round of the kernel's slots (four
    128x128 workgroups per CU / one 256x256 workgroup per CU), no K slice shorter than 1024 tokens."""
    if _WGRAD_GROUP_SPLITS > 0:
        return _WGRAD_GROUP_SPLITS
    return int(max(1, min(slots // max(1, tiles), K // 1024, 16)))


def _group_splits_x3(tiles, K):
    """K splits of a split-precision grouped launch.  Combined stages (config 6, two workgroups per CU = 512 slots), measured at
    K = 15104 (tools/bench_wgrad_x3.py): one full round when the tiles fit (projector: 88 tiles x 5 = 122 us against 172 at x 6),
    else about four rounds' worth (tied embedding, 630 tiles: x 3 = 783 us, x 2 = 845, x 1 = 1054); a block's 432 tiles run within
    2 % of their best at any count from 1 to 5."""
    if _X3_WG_CFG != 6:
        return _group_splits(tiles, K)
    if _WGRAD_GROUP_SPLITS > 0:
        return _WGRAD_GROUP_SPLITS
    s = 512 // max(1, tiles)
    if s == 0:
        s = 2048 // tiles
    return int(max(1, min(s, K // 1024, 16)))


def _wgrad_splits_p8(M, N, K, cus=256):
    """Split count for the eight-phase weight-gradient kernel (256x256 tiles, ONE workgroup per CU): tiles x splits fills one
    round of the CUs, every split a whole number of 128-deep K units, no empty trailing split.  None when the product is
    not eligible (mmtg_gemm's rule: K % 128 == 0, M, N >= 256) or has too many tiles to gain from splitting."""
    if not _P8T or K % 128 or M < 256 or N < 256:
        return None
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles > cus // 2:
        return None
    units = K // 128
    s0 = max(1, min(cus // tiles, units // 4))
    per = -(-units // s0)
    return -(-units // per)


def _round_capacity(n):
    """Next value of {1, 1.25, 1.5, 1.75} x 2^k at or above n (<= 25 % slack, O(log) distinct sizes per buffer so the
    caching allocator's freed blocks are reused instead of piling up)."""
    if n <= 1024:
        return 1024
    k = 1 << (int(n - 1).bit_length() - 1)      # largest power of two below n
    for q in (4, 5, 6, 7, 8):
        if k * q // 4 >= n:
            return k * q // 4
    return 2 * k


def mix_seed(seed, site):
    """32-bit hash of (step seed, dropout site): sites get unrelated counter-hash streams instead of one stream
    at small integer offsets (the kernels compute hash(element index * G + seed))."""
    x = (seed ^ (0x9E3779B9 * (site + 1))) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 0x846CA68B) & 0xFFFFFFFF
    return (x ^ (x >> 16)) & 0xFFFFFFFF


def initial_drop_seed(rank=0):
    """Start of the dropout counter: a hash of torch's seed (train.py:88-94 seeds torch) and the data-parallel rank,
    so replicas draw independent masks (the reference's DataParallel replicas do) and torch.manual_seed controls
    the sequence."""
    return mix_seed(torch.initial_seed() & 0xFFFFFFFF, 0x5EED + 7919 * rank)


# --------------------------------------------------------------------------
class Engine:
    """Owns the flat buffers and runs forward / loss / backward / optimizer."""

    def __init__(self, model_cfgs, data_cfg, gpt2_cfg, master, table, dtype=hip.BF16, x3=False, hybrid=False):
        hip.lib()  # fail loudly if the extension is not built
        if x3 and dtype != hip.F32:
            raise ValueError("the split-precision (bf16x3) products belong to fp32 storage")
        if hybrid and not x3:
            raise ValueError("the hybrid mode (bf16x3f) is the split-precision forward with a bf16 backward: it needs x3")
        if not master.is_cuda:
            raise RuntimeError("the MMTG engine runs on an MI355X (cuda) device only -- there is no CPU path")
        self.sh = Shapes(model_cfgs, data_cfg, gpt2_cfg)
        if x3 and self.sh.D % 128:
            # the split-precision products walk whole 128-wide K tiles of n_embd; an x3 engine has no mixed fallback for its training
            # step (its weight copies are plane pairs, its saved activations too), so a width it cannot serve is refused up front
            raise ValueError("compute_dtype='bf16x3' needs n_embd to be a multiple of 128 (got %d): use compute_dtype='f32'" % self.sh.D)
        self.layout = ParamLayout(model_cfgs, gpt2_cfg)
        assert master.numel() == self.layout.total and master.dtype == torch.float32
        self.dev = master.device
        self.dtype = dtype
        self.tdt = hip.torch_dtype(dtype)
        self.master = master
        self.grad = None
        self.wc = master if dtype == hip.F32 else torch.zeros(self.layout.total, device=self.dev, dtype=torch.bfloat16)
        # x3 (round 5): fp32 storage everywhere, the GPT-2 / LM-head products on the bf16 matrix cores as three passes over
        # (hi | lo) bf16 plane pairs (hip.gemm_x3).  wc2 = the plane pair of the whole flat parameter buffer.
        self.x3 = bool(x3)
        # hybrid ("bf16x3f", round 6): the forward of the x3 mode -- logits / loss / KL at the fp32 mode's parity -- and the backward of
        # the bf16 mode, ONE matrix-core pass per product, over the hi planes the forward stored (hi = bf16(x): exactly the tensor the
        # bf16 kernels take, same leading dimension) and the hi planes of the weights; the encoder / fuser stay exact fp32 both ways
        self.hybrid = bool(hybrid)
        self.wc2 = torch.zeros(2, self.layout.total, device=self.dev, dtype=torch.bfloat16) if self.x3 else None
        self.copies_fresh = dtype == hip.F32 and not self.x3
        self._init_transposed()
        self.set_table(table)
        self.prior = torch.from_numpy(gaussian_prior(self.sh.S)).to(self.dev)
        self.ws = {}
        self._ws_tag = ""
        self.act = None
        self.training = False
        self.drop_seed = initial_drop_seed(_dist_rank())
        self.wgrad_overwrite = False   # set by MMTGTrainer.step around its backward
        self._ow_desc, self._ow_rec = {}, None   # zero lists per step shape / the record in progress (zero_grad)
        self._lazy = None                        # ranges the last zero_grad left un-zeroed (the backward overwrites them)
        self.step_count = 0
        self.opt_m = None
        self.opt_v = None
        self.normsq = torch.zeros(1, device=self.dev)
        self.bucket_hook = None   # callable(pack_index) fired as packs of gradients become final
        self._sums, self._defer = [], False     # deferred column sums of the backward (see _defer_sum)
        self._tail_jobs, self._tail_side = [], None      # grouped weight gradients deferred to the backward's tail (_WGRAD_TAIL)
        self._pf_stream = None
        self._rowmaps = {}

    # ---------------------------------------------------------------- buffers / views
    def set_table(self, table):
        """WenLan token table E[V, emb] (reference: host dict token_id2emb, model.py:215,221-223)."""
        if table is None:
            self.table = None
            return
        t = torch.as_tensor(table)
        if t.shape[1] != self.sh.E:
            raise ValueError("token table width %d != wenlan_emb_size %d" % (t.shape[1], self.sh.E))
        if self.dtype == hip.F32:
            self.table = t.to(self.dev, torch.float32).contiguous()
        else:       # a packed bf16 table goes to the device as it is (54.5 MB, no fp32 detour)
            self.table = t.to(self.dev).to(torch.bfloat16).contiguous()

    def buf(self, name, shape, dtype=None, zero=False):
        """Named workspace.  One allocation per (name, dtype), sized for the largest request seen so far (rounded up
        on a coarse grid) and handed out as a ``[:numel].view(shape)`` slice: the curriculum filter (train.py:178-186)
        changes the row count on almost every step of stages 1 and 2, and a buffer set per exact shape would grow
        without bound (0.3 MB per token at the full configuration).  Every kernel takes its row count and leading
        dimensions explicitly, so a slice of a larger allocation is as good as an exact one."""
        dtype = self.tdt if dtype is None else dtype
        n = 1
        for d in shape:
            n *= int(d)
        key = (name + self._ws_tag, dtype)          # (_ws_tag: "" / "~s" while a second stream's chain is being enqueued: its own workspaces)
        t = self.ws.get(key)
        if t is None or t.numel() < n:
            cap = _round_capacity(n)
            t = torch.zeros(cap, device=self.dev, dtype=dtype) if zero else torch.empty(cap, device=self.dev, dtype=dtype)
            self.ws[key] = t
            return t[:n].view(shape)
        v = t[:n].view(shape)
        if zero:
            v.zero_()
        return v

    def P(self, key):   # fp32 master view
        return self.layout.view(self.master, key)

    def W(self, key):   # compute-dtype weight view
        return self.layout.view(self.wc, key)

    def G(self, key):   # fp32 gradient view
        return self.layout.view(self.grad, key)

    def Wp(self, pack):
        return self.layout.pack(self.wc, pack)

    def Pp(self, pack):
        return self.layout.pack(self.master, pack)

    def Gp(self, pack):
        return self.layout.pack(self.grad, pack)

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros(self.layout.total, device=self.dev, dtype=torch.float32)

    def zero_grad(self, shape_key=None):
        """Zero the flat gradient buffer.  With a shape key (the fused trainer: rows x positions of the step) only what is
        ACCUMULATED into is zeroed once a step of that shape has shown which tensors its backward OVERWRITES (the slab-sum
        block matrices, recorded by _wgrad: 340 of 497 MB at the full configuration) -- one launch over the complement."""
        self.ensure_grad()
        self._ow_rec = None
        self._lazy = None
        if shape_key is not None and _LAZY_ZERO:
            ent = self._ow_desc.get(shape_key)
            if ent is not None:
                desc, ranges = ent
                hip.zero_ranges(self.grad, desc, desc.shape[0])
                self._lazy = ranges          # these ranges were NOT zeroed: this step's backward must overwrite them
                return
            if len(self._ow_desc) < 64:
                self._ow_rec = (shape_key, [])          # record this step's overwritten ranges
        self.grad.zero_()

    def _finish_overwrite_record(self):
        """After a recorded backward: the complement of the overwritten ranges becomes the zero list of that step shape."""
        if self._ow_rec is None:
            return
        key, ranges = self._ow_rec
        self._ow_rec = None
        ranges = sorted(r for r in ranges if r[0] % 4 == 0 and r[1] % 4 == 0)
        comp, pos = [], 0
        for off, n in ranges:
            if off < pos:
                return                                # overlapping records: keep zeroing everything
            if off > pos:
                comp.append((pos, off - pos))
            pos = off + n
        if pos < self.layout.total:
            comp.append((pos, self.layout.total - pos))
        if not ranges or any(o % 4 or c % 4 for o, c in comp):
            return
        self._ow_desc[key] = (torch.tensor(comp, dtype=torch.int64, device=self.dev), frozenset(ranges))

    def _init_transposed(self):
        """bf16 mode: K-contiguous [out,in] copies of GPT-2's Conv1D [in,out] weights.  With them every
        forward product is the NT layout (both operands K-contiguous in LDS, one ds_read_b128 per
        fragment); the NN layout reads its K-strided weights with two transposed 8-byte LDS reads per
        fragment and measured 5-15 % slower per product (profiles/r01_v4_gemm_per_shape.log)."""
        self.wt = None
        self.wte_t = None
        self.wt_entries = {}
        if self.dtype == hip.F32 and not self.x3:
            return
        pre, desc, off = "decoder.gpt2.transformer.", [], 0
        for l in range(self.sh.L):
            for nm in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):
                key = "%sh.%d.%s.weight" % (pre, l, nm)
                soff, shape, n = self.layout.entries[key]
                assert soff % 8 == 0 and shape[0] % 8 == 0 and shape[1] % 8 == 0
                desc.append((soff, shape[0], shape[1], off))
                self.wt_entries[key] = (off, (shape[1], shape[0]), n)
                off += n
        if self.x3:
            # x3: the input gradient of a Linear layer, dy [M,out] W[out,in], needs the [in,out] copy as its K-contiguous operand
            key = "decoder.projector_layer2.weight"
            soff, shape, n = self.layout.entries[key]
            if soff % 8 == 0 and shape[0] % 8 == 0 and shape[1] % 8 == 0:
                desc.append((soff, shape[0], shape[1], off))
                self.wt_entries[key] = (off, (shape[1], shape[0]), n)
                off += n
        # (x3: one more leading dimension, the (hi | lo) plane)
        self.wt = torch.zeros(*((2, off) if self.x3 else (off,)), device=self.dev, dtype=torch.bfloat16)
        self.wt_total = off
        self.wt_desc = torch.tensor(desc, dtype=torch.int64, device=self.dev)
        self.wt_max = (max(d[1] for d in desc), max(d[2] for d in desc))
        # ... and a [D, Vpad] copy of the tied embedding matrix: the LM head's dgrad d_h = dlogits @ wte contracts over the
        # vocabulary, wte's ROW index -- through the copy it is a K-contiguous x K-contiguous product like every other
        # forward / dgrad product (eight-phase kernel).  Its own launch: a shared grid would be sized by its 13440 rows.
        woff = self.layout.pack_range["wte"][0]
        self.wte_t = (torch.zeros(*((2,) if self.x3 else ()), self.sh.D, self.layout.Vpad, device=self.dev, dtype=torch.bfloat16)
                      if (_WTE_T or self.x3) else None)
        self.wte_desc = torch.tensor([(woff, self.layout.Vpad, self.sh.D, 0)], dtype=torch.int64, device=self.dev)

    def Wt(self, key):   # [out, in] copy of a Conv1D weight (bf16 mode)
        if self.x3:
            raise RuntimeError("Engine.Wt: an x3 engine's transposed copies are (hi | lo) plane pairs -- use Wtx()")
        off, shape, n = self.wt_entries[key]
        return self.wt[off:off + n].view(shape)

    def _refresh_transposed(self):
        if self.wt is not None and self.x3:
            for pl in (0, 1):
                hip.transpose_batch(self.wc2[pl], self.wt[pl], self.wt_desc, self.wt_desc.shape[0], *self.wt_max)
                hip.transpose_batch(self.wc2[pl], self.wte_t[pl], self.wte_desc, 1, self.layout.Vpad, self.sh.D)
        elif self.wt is not None:
            hip.transpose_batch(self.wc, self.wt, self.wt_desc, self.wt_desc.shape[0], *self.wt_max)
            if self.wte_t is not None:
                hip.transpose_batch(self.wc, self.wte_t, self.wte_desc, 1, self.layout.Vpad, self.sh.D)

    def refresh_copies(self):
        """bf16 mode: re-derive the GEMM weight copies from the fp32 masters (an
        external optimizer may have updated them).  654 MB of traffic, ~0.15 ms.
        x3 mode: the (hi | lo) plane pair of the flat buffer and its transposed copies."""
        if self.copies_fresh:
            return
        if self.x3:
            hip.split_planes(self.master, 1, self.layout.total, hip.Planes(self.wc2, 1, self.layout.total))
            self._refresh_transposed()
            self.copies_fresh = True
        elif self.dtype != hip.F32:
            hip.cast_f32_to(self.master, self.wc, self.layout.total)
            self._refresh_transposed()
            self.copies_fresh = True

    def invalidate_copies(self):
        if self.dtype != hip.F32 or self.x3:
            self.copies_fresh = False

    # ---------------------------------------------------------------- split-precision (x3) operands
    def Wx(self, key):
        """Plane pair of a parameter as stored ([out,in] Linear / [in,out] Conv1D)."""
        off, shape, n = self.layout.entries[key]
        return hip.Planes(self.wc2[0, off:off + n], shape[0], shape[1], plane=self.layout.total)

    def Wtx(self, key):
        """Plane pair of the [out,in] copy of a Conv1D weight."""
        off, shape, n = self.wt_entries[key]
        return hip.Planes(self.wt[0, off:off + n], shape[0], shape[1], plane=self.wt_total)

    def Wpx(self, pack, rows, cols):
        off, n = self.layout.pack_range[pack]
        return hip.Planes(self.wc2[0, off:off + n], rows, cols, plane=self.layout.total)

    def pbuf(self, name, rows, cols):
        """Named (hi | lo) plane-pair workspace of an fp32 [rows, cols] activation."""
        t = self.buf(name, (2, rows, cols), torch.bfloat16)
        return hip.Planes(t, rows, cols)

    def _fwd_x3(self, xp, wkey, out, M, bias=None, planes=None, **kw):
        """x3 forward product through the [out,in] plane pair of a Conv1D weight: out / planes = epi(x W + bias)."""
        w = self.Wtx(wkey)
        hip.gemm_x3(xp, w, out, M, w.rows, w.cols, planes=planes, bias=bias, **kw)

    def _dgrad_x3(self, dyp, wkey, dx, M, planes=None, **kw):
        """x3 input gradient of a Conv1D layer: dy [M,out] W[in,out]^T -- the weight as stored is the K-contiguous operand."""
        w = self.Wx(wkey)
        hip.gemm_x3(dyp, w, dx, M, w.rows, w.cols, planes=planes, **kw)

    # ---------------------------------------------------------------- GEMM helpers
    def _gemm_few_rows(self, A, Bm, out, M, N, K, transB, ldb, bias=None, lda=None):
        """Encoder-sized products (M = B or B*S rows, K >= 1024): a handful of output tiles each walking a long K
        serially leaves most CUs idle, so the bf16 mode splits K into fp32 slabs (deterministic) and lets the
        finish kernel add the bias.  Returns False when the plain launch should be used."""
        if K < 1024 or K % 8 or N % 8 or out.dtype != self.tdt or _NO_FEW_ROWS:
            return False
        if self.dtype == hip.F32 and (K % 32 or (lda or K) % 4):
            return False
        # (fp32 kernel, round 5: the same K slabs through its MMTG_EPI_SPLIT epilogue -- 128x128 tiles; the encoder's 2048-deep
        #  products on 6-36 tiles were 2.7 ms of the bf16x3 step at 6 TFLOP/s)
        tiles = (N + 31) // 32 if (transB and M <= 256 and self.dtype == hip.BF16) else ((M + 127) // 128) * ((N + 127) // 128)
        splits = min(K // 256, 384 // tiles)
        if tiles > 48 or splits < 2:
            return False
        part = self.buf("few_rows_slabs", (8 * 320 * 2048,), torch.float32)
        if splits * M * N > part.numel():
            return False
        hip.gemm(A, Bm, part, M, N, K, transB=transB, lda=lda, ldb=ldb, ldc=N, epi=hip.EPI_SPLIT, out_f32=True, splits=splits)
        hip.splitk_finish(part, splits, M, N, out, bias=bias)
        return True

    def _fwd(self, x, wkey, out, M, kind, bias=None, lda=None, **kw):
        """out[M,out] = x[M,in] W (+bias).  kind: 'linear' = [out,in], 'conv1d' = [in,out]."""
        w = self.W(wkey)
        if kind == "linear":
            N, K = w.shape
            if not kw and self._gemm_few_rows(x, w, out, M, N, K, True, K, bias=bias, lda=lda):
                return
            hip.gemm(x, w, out, M, N, K, transB=True, lda=lda, ldb=K, bias=bias, **kw)
        elif wkey in self.wt_entries and not self.x3:   # bf16: the [out,in] copy -> NT layout
            # (an x3 engine's `wt` holds (hi | lo) PLANE PAIRS, read by _fwd_x3 only; its plain products -- the decoder's
            #  MMTG_DECODE_X3=0 / n_embd > 1024 fallback -- take the fp32 weight as stored through the exact-fp32 kernel below)
            K, N = w.shape
            hip.gemm(x, self.Wt(wkey), out, M, N, K, transB=True, lda=lda, ldb=K, bias=bias, **kw)
        else:
            K, N = w.shape
            hip.gemm(x, w, out, M, N, K, transB=False, lda=lda, ldb=N, bias=bias, **kw)

    def _dgrad(self, dy, wkey, dx, M, kind, **kw):
        w = self.W(wkey)
        if kind == "linear":
            K, N = w.shape   # dy [M,out=K] @ W[out,in] -> [M,in=N]
            if not kw and self._gemm_few_rows(dy, w, dx, M, N, K, False, N):
                return
            hip.gemm(dy, w, dx, M, N, K, transB=False, ldb=N, **kw)
        else:
            N, K = w.shape   # dy [M,out=K] @ W[in,out]^T -> [M,in=N]
            hip.gemm(dy, w, dx, M, N, K, transB=True, ldb=K, **kw)

    def _wgrad(self, x, dy, wkey, bkey, Mtok, kind, ldx=None, ldy=None, bias_rows=None):
        gw = self.G(wkey)
        if kind == "linear":    # gw[out,in] += dy^T x
            out_f, in_f = gw.shape
            A, B, Mg, Ng, lda, ldb = dy, x, out_f, in_f, ldy or out_f, ldx or in_f
        else:                   # Conv1D: gw[in,out] += x^T dy
            in_f, out_f = gw.shape
            A, B, Mg, Ng, lda, ldb = x, dy, in_f, out_f, ldx or in_f, ldy or out_f
        bf = self.dtype == hip.BF16
        splits = _wgrad_splits(Mg, Ng, Mtok, bf)
        tiles = ((Mg + 127) // 128) * ((Ng + 127) // 128)
        if bf and _WGRAD_SLAB and Ng % 8 == 0:
            splits = _wgrad_splits_p8(Mg, Ng, Mtok) or splits
        # (the exact-fp32 cross-check mode, round 6: its split weight gradients as slabs too -- any tile count, speed is not its
        #  point -- so that the mode's gradient is bit-reproducible like the others'; MMTG_WGRAD_ATOMIC=1 restores the atomics)
        exact = self.dtype == hip.F32 and not self.x3
        use_slab = _WGRAD_SLAB and 1 < splits and Ng % 8 == 0 and ((bf and tiles < 512) or exact)
        if self._lazy is not None and (self.layout.entries[wkey][0], Mg * Ng) in self._lazy and not (use_slab and self.wgrad_overwrite):
            gw.zero_()          # the lazy zero_grad skipped this tensor expecting an overwrite that is not happening now
        if use_slab:
            # K-split slabs with plain stores + an ordered sum instead of fp32 atomics (deterministic)
            # (one workspace, sized for the largest request so far: 113 MB at GPT-2 base, 268 MB at GPT-2 medium)
            part = self.buf("wgrad_slabs", (splits * Mg * Ng,), torch.float32)
            hip.gemm(A, B, part, Mg, Ng, Mtok, transA=True, transB=False, lda=lda, ldb=ldb, ldc=Ng,
                     epi=hip.EPI_SPLIT, out_f32=True, splits=splits)
            # (the fused trainer zeroes the gradients right before its single backward: the sum may then overwrite)
            hip.slab_sum(part, splits, Mg * Ng, gw, Mg * Ng, accumulate=not self.wgrad_overwrite)
            if self.wgrad_overwrite and self._ow_rec is not None:
                self._ow_rec[1].append((self.layout.entries[wkey][0], Mg * Ng))
            splits = 0
        if splits:
            hip.gemm(A, B, gw, Mg, Ng, Mtok, transA=True, transB=False, lda=lda, ldb=ldb, ldc=Ng,
                     epi=hip.EPI_ATOMIC, splits=splits)
        if bkey is not None:
            hip.colsum(dy, Mtok if bias_rows is None else bias_rows, out_f, self.G(bkey), ldx=ldy or out_f)

    # ---------------------------------------------------------------- forward
    def _cast_in(self, t, name):
        """Batch embeddings arrive f64/f32 (model.py:371-373 calls .float()); store in compute dtype."""
        t = t.to(self.dev, torch.float32).contiguous()
        if self.dtype == hip.F32:
            return t
        out = self.buf(name, t.shape)
        hip.cast_f32_to(t, out, t.numel())
        return out

    def _rnn_fwd(self, ch, x, B, training, seed, site):
        """One encoder channel (model.py:39-59, 78-79): `num_layers` stacked GRU / LSTM / ReLU-RNN layers over the S steps, h0 = 0,
        rows b*S+t.  Per layer: W_ih x + b_ih for all steps in ONE product, then per step the recurrent product W_hh h_{t-1} + b_hh
        and the cell kernel; between layers nn.RNNBase's dropout (training only; counter-hash mask, regenerated in the backward).
        Returns the per-layer activation records the backward needs."""
        sh = self.sh
        S, H = sh.S, sh.H
        kind, NL = sh.rnn[ch]
        G = hip.RNN_GATES[kind]
        r = f"encoder.rnns_{ch}."
        gh = self.buf("gh%d" % G, (B, G * H))
        layers, inp = [], x
        for l in range(NL):
            sfx = "_l%d" % l
            tag = ch if l == 0 else "%s%d" % (ch, l)
            gi = self.buf("gi_" + tag, (B * S, G * H))
            self._fwd(inp, r + "weight_ih" + sfx, gi, B * S, "linear", bias=self.P(r + "bias_ih" + sfx))
            h_all = self.buf("h_" + tag, (B * S, H))
            rec = {"inp": inp, "gi": gi, "h": h_all, "drop": None}
            if kind == "GRU":
                save = rec["save"] = self.buf("gru_save_" + tag, (S, 4, B, H), torch.float32)
                for t in range(S):
                    if t == 0:      # h_prev = 0: the recurrent product is b_hh itself (one row, stride 0), no launch
                        hip.gru_cell_fwd(gi[t:], self.W(r + "bias_hh" + sfx), None, h_all[t:], save[t], B, H,
                                         ld_gi=S * 3 * H, ld_hp=S * H, ld_h=S * H, ld_gh=0)
                        continue
                    self._fwd(h_all[t - 1:], r + "weight_hh" + sfx, gh, B, "linear", bias=self.P(r + "bias_hh" + sfx), lda=S * H)
                    hip.gru_cell_fwd(gi[t:], gh, h_all[t - 1:], h_all[t:], save[t], B, H,
                                     ld_gi=S * 3 * H, ld_hp=S * H, ld_h=S * H)
            else:
                lstm = kind == "LSTM"
                code = hip.RNN_LSTM if lstm else hip.RNN_RELU
                save = rec["save"] = self.buf("lstm_save_" + tag, (S, 5, B, H), torch.float32) if lstm else None
                c_all = rec["c"] = self.buf("lstm_c_" + tag, (S, B, H), torch.float32) if lstm else None
                for t in range(S):
                    if t == 0:
                        g_t, ld_gh = self.W(r + "bias_hh" + sfx), 0
                    else:
                        self._fwd(h_all[t - 1:], r + "weight_hh" + sfx, gh, B, "linear", bias=self.P(r + "bias_hh" + sfx), lda=S * H)
                        g_t, ld_gh = gh, None
                    hip.rnn_cell_fwd(code, gi[t:], g_t, c_all[t - 1] if lstm and t else None, h_all[t:], c_all[t] if lstm else None,
                                     save[t] if lstm else None, B, H, ld_gi=S * G * H, ld_h=S * H, ld_gh=ld_gh)
            layers.append(rec)
            inp = h_all
            if l + 1 < NL and training and sh.rnn_pdrop > 0.0:
                rec["drop"] = mix_seed(seed, 0x1000 + 64 * site + l)
                inp = self.buf("hd_" + tag, (B * S, H))
                hip.dropout_apply(h_all, inp, B * S * H, sh.rnn_pdrop, rec["drop"])
        return layers

    def forward(self, batch, train_flag=True, training=False, per_row_infer=True, need_logits=True, logits_f32=True,
                encode_only=False):
        """MMTG.forward (model.py:356-400).  Returns dict(logits_pad [M,Vpad] f32, B, T, ...);
        lm_loss / kl are device scalars in self.scalars after loss().  need_logits=False stops after the last block's c_attn (no ln_f, no
        LM head: the decoder's prompt prefill reads the blocks' K / V rows only); encode_only=True stops after the fuser."""
        sh = self.sh
        if self.table is None:
            raise RuntimeError("no WenLan token table set (vocab/token_id2emb_dict.pkl or set_token_table())")
        self.refresh_copies()
        self.training = training
        pe, pa, pr = sh.pdrop if training else (0.0, 0.0, 0.0)
        self.drop_seed = (self.drop_seed * 1664525 + 1013904223) & 0xFFFFFFFF
        seed = self.drop_seed
        S, E, H, D, P = sh.S, sh.E, sh.H, sh.D, sh.P
        img = batch["img_embs"]
        B = img.shape[0]
        if img.shape[1] != S:
            raise ValueError("batch has %d experience steps, model_cfgs['seq_len'] = %d" % (img.shape[1], S))
        if encode_only:          # decode: only the experience vectors c[B,S,E] are needed
            targets = topic_ids = None
            L = T = M = 0
        else:
            targets = batch["targets"].to(self.dev, torch.long).contiguous()
            topic_ids = batch["topic_ids"].to(self.dev, torch.long).contiguous()
            L = targets.shape[1]
            T = P + L
            M = B * T
            if T > sh.NP:
                raise ValueError("sequence length %d exceeds n_positions %d" % (T, sh.NP))
        a = {"B": B, "L": L, "T": T, "M": M, "seed": seed, "pdrop": (pe, pa, pr), "train_flag": train_flag, "gelu_grad": _GELU_GRAD,
             "targets": targets, "topic_ids": topic_ids}

        # ---------------- encoder: topic_fc + 2 GRUs (model.py:63-81) + ln_layer1..3 (:380-382)
        xt = self._cast_in(batch["topic_emb"], "xt")
        xi = self._cast_in(img, "xi").view(B * S, E)
        xr = self._cast_in(batch["r_embs"], "xr").view(B * S, E)
        t_raw = self.buf("t_raw", (B, H))
        self._fwd(xt, "encoder.topic_fc.weight", t_raw, B, "linear", bias=self.P("encoder.topic_fc.bias"))
        t_ln = self.buf("t_ln", (B, H))
        st = {}
        st["ln1"] = (self.buf("ln1_mu", (B,), torch.float32), self.buf("ln1_rs", (B,), torch.float32))
        hip.layernorm_fwd(t_raw, t_ln, self.P("ln_layer1.weight"), self.P("ln_layer1.bias"), *st["ln1"], B, H)
        enc = {}
        for ch, x, lnk, site in (("image", xi, "ln_layer2", 0), ("text", xr, "ln_layer3", 1)):
            with self._beside(ch == "text"):
                layers = self._rnn_fwd(ch, x, B, training, seed, site)
                h_all = layers[-1]["h"]
                h_ln = self.buf("hln_" + ch, (B * S, H))
                st[lnk] = (self.buf(lnk + "_mu", (B * S,), torch.float32), self.buf(lnk + "_rs", (B * S,), torch.float32))
                hip.layernorm_fwd(h_all, h_ln, self.P(lnk + ".weight"), self.P(lnk + ".bias"), *st[lnk], B * S, H)
            enc[ch] = (layers, h_ln)
        self._join_beside()

        # ---------------- alpha attention (model.py:133-161) on batch-first rows b*S+i
        kl = self.buf("kl", (1,), torch.float32, zero=True)
        alpha = {}
        for mod, ch in (("img", "image"), ("text", "text")):
            h_ln = enc[ch][1]
            qkv = self.buf("aqkv_" + mod, (B * S, 3 * H))
            hip.gemm(h_ln, self.Wp(mod + "_qkv_w"), qkv, B * S, 3 * H, H, transB=True, ldb=H, bias=self.Pp(mod + "_qkv_b"))
            ctx = self.buf("actx_" + mod, (B * S, H))
            probs = self.buf("aprobs_" + mod, (B, sh.heads, S, S), torch.float32)
            hip.alpha_attn_fwd(qkv, self.prior, ctx, probs, kl, B, S, H, sh.heads)
            alpha[mod] = (qkv, ctx, probs)

        # ---------------- beta fuser (model.py:181-202) -> c [B,S,E]
        o = self.buf("beta_o", (B * S, H))
        ba = self.buf("beta_a", (B, S, 3), torch.float32)
        hip.beta_fuse_fwd(t_ln, alpha["img"][1], alpha["text"][1], self.Pp("att_w"), self.Pp("att_b"), o, ba, B, S, H)
        c = self.buf("c", (B * S, E))
        self._fwd(o, "mm_atten_layer.out_linear.weight", c, B * S, "linear", bias=self.P("mm_atten_layer.out_linear.bias"))

        if encode_only:
            a.update(c=c, kl=kl)
            self.act = a
            return a
        # ---------------- decoder front end (model.py:251-281) + GPT-2 input embedding
        if train_flag:
            type_ids = torch.cat([batch["tpw_type_ids"].to(self.dev).long(), batch["type_ids"].to(self.dev).long()], 1)
            keep = torch.cat([batch["tpw_attention_mask"].to(self.dev), batch["attention_mask"].to(self.dev)], 1)
        else:
            type_ids, keep = self._infer_types_mask(batch, targets, per_row_infer)
        type_ids = type_ids.contiguous().view(-1)
        keep = (keep != 0).to(torch.int32).contiguous()
        h1 = self.buf("h1", (M, H))
        px3 = (self.x3 and E % 128 == 0 and H % 128 == 0 and D % 8 == 0 and "decoder.projector_layer2.weight" in self.wt_entries
               and _os.environ.get("MMTG_X3_PROJECTOR", "1") != "0")
        a["px3"] = px3
        gather = self.dtype == hip.BF16 and M > 256 and not _NO_GATHER
        if gather:
            # Fused conditioning (model.py:254-281): X[m] = E[id_m] + c[b, seg_m] is never materialised.  By linearity
            # X W1^T = E[id] W1^T + (c W1^T)[b, seg]: the projector product gathers its A rows straight from the WenLan table
            # (the row index is the LDS-DMA lane's source offset) and its epilogue adds row b*S + seg of the small product
            # c W1^T (row B*S = zeros for the prompt and the positions past the last segment) before the tanh.
            ids32 = torch.cat([topic_ids, targets], 1).to(torch.int32).contiguous().view(-1)
            rowmap = self._rowmaps.get((B, T))
            if rowmap is None:          # depends on the shapes only: built once per (B, T)
                lp = torch.arange(T, device=self.dev) - P
                seg = torch.div(lp, sh.two_sents, rounding_mode="floor")
                seg = torch.where((lp >= 0) & (seg < S), seg, torch.full_like(seg, -1))
                rowmap = torch.where(seg >= 0, torch.arange(B, device=self.dev)[:, None] * S + seg[None, :],
                                     torch.full((B, T), B * S, device=self.dev)).to(torch.int32).contiguous().view(-1)
                if len(self._rowmaps) > 256:
                    self._rowmaps.clear()
                self._rowmaps[(B, T)] = rowmap
            cW = self.buf("c_w1", (B * S + 1, H))
            self._fwd(c, "decoder.projector_layer1.weight", cW[:B * S], B * S, "linear")
            cW[B * S:].zero_()
            hip.gemm_gather(0, self.table, self.W("decoder.projector_layer1.weight"), h1, M, H, E, ids32, self.table.shape[0],
                            lda=E, ldb=E, bias=self.P("decoder.projector_layer1.bias"), epi=hip.EPI_TANH_ADD, aux=cW, ldaux=H,
                            aux_rows=rowmap)
            x = None
        else:
            ids32 = None
            x = self.buf("x_cond", (M, E))
            hip.embed_condition(self.table, topic_ids, targets, c, x, B, P, L, S, E, sh.two_sents, self.table.shape[0])
            if px3:
                # split-precision projector (model.py:279-281): X and tanh(X W1^T + b1) also as plane pairs (the products' operands)
                xp = hip.split_planes(x, M, E, self.pbuf("x_cond_p", M, E))
                h1p = self.pbuf("h1_p", M, H)
                hip.gemm_x3(xp, self.Wx("decoder.projector_layer1.weight"), h1, M, H, E, planes=h1p,
                            bias=self.P("decoder.projector_layer1.bias"), epi=hip.EPI_TANH)
                a.update(xp=xp, h1p=h1p)
            else:
                self._fwd(x, "decoder.projector_layer1.weight", h1, M, "linear", bias=self.P("decoder.projector_layer1.bias"),
                          epi=hip.EPI_TANH)
        hcur = self.buf("resid_0", (M, D))
        if px3:
            hip.gemm_x3(a["h1p"], self.Wx("decoder.projector_layer2.weight"), hcur, M, D, H, bias=self.P("decoder.projector_layer2.bias"))
        else:
            self._fwd(h1, "decoder.projector_layer2.weight", hcur, M, "linear", bias=self.P("decoder.projector_layer2.bias"))
        pre = "decoder.gpt2.transformer."
        hip.embed_add(hcur, self.W(pre + "wpe.weight"), self.W(pre + "wte.weight"), type_ids, hcur, M, T, D,
                      drop_p=pe, drop_seed=seed)

        # ---------------- GPT-2 blocks
        layers = []
        x3 = self.x3 and D % 128 == 0
        a["x3"] = x3
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            s = (mix_seed(seed, 3 * l + 1), mix_seed(seed, 3 * l + 2), mix_seed(seed, 3 * l + 3))   # attn, resid 1, resid 2
            mu1 = self.buf(f"l{l}_mu1", (M,), torch.float32)
            rs1 = self.buf(f"l{l}_rs1", (M,), torch.float32)
            mu2 = self.buf(f"l{l}_mu2", (M,), torch.float32)
            rs2 = self.buf(f"l{l}_rs2", (M,), torch.float32)
            if x3:
                # split-precision block: the LayerNorms, c_attn and the GELU write (hi | lo) plane pairs -- only products and the
                # split-precision attention read them -- the attention context is needed in fp32 by its backward's delta AND as
                # planes by c_proj
                a1 = self.pbuf(f"l{l}_a", M, D)
                hyb = self.hybrid
                # (hybrid: the LayerNorm kernels also store their input rows as bf16 -- what the bf16 LayerNorm backward reads)
                xb1 = self.buf(f"l{l}_xin_b", (M, D), torch.bfloat16) if hyb else None
                xb2 = self.buf(f"l{l}_xmid_b", (M, D), torch.bfloat16) if hyb else None
                hip.layernorm_fwd_x3(hcur, a1, self.P(p + "ln_1.weight"), self.P(p + "ln_1.bias"), mu1, rs1, M, D, sh.eps, xb=xb1)
                qkv = self.pbuf(f"l{l}_qkvp", M, 3 * D)
                self._fwd_x3(a1, p + "attn.c_attn.weight", None, M, bias=self.P(p + "attn.c_attn.bias"), planes=qkv, ldc=3 * D)
                if not need_logits and l == sh.L - 1:      # (the prefill wants K / V only: nothing of the last block beyond c_attn)
                    layers.append((hcur, mu1, rs1, a1, qkv))
                    break
                ctx = self.buf(f"l{l}_ctx", (M, D))
                lse = self.buf(f"l{l}_lse", (B, sh.nH, T), torch.float32)
                ctxp = self.pbuf(f"l{l}_ctxp", M, D)
                hip.attn_fwd_x3(qkv, keep, ctx, ctxp, lse, B, T, sh.nH, D // sh.nH, drop_p=pa, drop_seed=s[0])
                xmid = self.buf(f"l{l}_xmid", (M, D))
                self._fwd_x3(ctxp, p + "attn.c_proj.weight", xmid, M, bias=self.P(p + "attn.c_proj.bias"),
                             epi=hip.EPI_RESID, aux=hcur, ldaux=D, drop_p=pr, drop_seed=s[1])
                m2 = self.pbuf(f"l{l}_m", M, D)
                hip.layernorm_fwd_x3(xmid, m2, self.P(p + "ln_2.weight"), self.P(p + "ln_2.bias"), mu2, rs2, M, D, sh.eps, xb=xb2)
                # (hybrid: the saved pre-activation as bf16 rows, the bf16 dGELU epilogue's operand -- half the bytes, too)
                u = self.buf(f"l{l}_u_b" if hyb else f"l{l}_u", (M, 4 * D), torch.bfloat16 if hyb else None)
                gact = self.pbuf(f"l{l}_g", M, 4 * D)
                self._fwd_x3(m2, p + "mlp.c_fc.weight", None, M, bias=self.P(p + "mlp.c_fc.bias"), planes=gact, ldc=4 * D,
                             epi=hip.EPI_GELU, aux2=u, flags=hip.GEMM_AUX2_BF16 if hyb else 0)
                xout = self.buf(f"resid_{l + 1}", (M, D))
                self._fwd_x3(gact, p + "mlp.c_proj.weight", xout, M, bias=self.P(p + "mlp.c_proj.bias"),
                             epi=hip.EPI_RESID, aux=xmid, ldaux=D, drop_p=pr, drop_seed=s[2])
                layers.append((hcur, mu1, rs1, a1, qkv, ctx, lse, xmid, mu2, rs2, m2, u, gact, s, ctxp))
                if hyb:
                    a.setdefault("hyb_rows", []).append((xb1, xb2))
                hcur = xout
                continue
            a1 = self.buf(f"l{l}_a", (M, D))
            hip.layernorm_fwd(hcur, a1, self.P(p + "ln_1.weight"), self.P(p + "ln_1.bias"), mu1, rs1, M, D, sh.eps)
            qkv = self.buf(f"l{l}_qkv", (M, 3 * D))
            self._fwd(a1, p + "attn.c_attn.weight", qkv, M, "conv1d", bias=self.P(p + "attn.c_attn.bias"))
            if not need_logits and l == sh.L - 1:          # (the prefill wants K / V only: nothing of the last block beyond c_attn)
                layers.append((hcur, mu1, rs1, a1, qkv))
                break
            ctx = self.buf(f"l{l}_ctx", (M, D))
            lse = self.buf(f"l{l}_lse", (B, sh.nH, T), torch.float32)
            hip.attn_fwd(qkv, keep, ctx, lse, B, T, sh.nH, D // sh.nH, drop_p=pa, drop_seed=s[0])
            xmid = self.buf(f"l{l}_xmid", (M, D))
            self._fwd(ctx, p + "attn.c_proj.weight", xmid, M, "conv1d", bias=self.P(p + "attn.c_proj.bias"),
                      epi=hip.EPI_RESID, aux=hcur, ldaux=D, drop_p=pr, drop_seed=s[1])
            m2 = self.buf(f"l{l}_m", (M, D))
            hip.layernorm_fwd(xmid, m2, self.P(p + "ln_2.weight"), self.P(p + "ln_2.bias"), mu2, rs2, M, D, sh.eps)
            u = self.buf(f"l{l}_u", (M, 4 * D))
            gact = self.buf(f"l{l}_g", (M, 4 * D))
            # (u: the pre-activation -- or, with MMTG_GELU_GRAD=1, gelu_new'(pre-activation): all the backward ever takes from it;
            #  saving the derivative moves the backward's exp / rcp into this epilogue, which computes the sigmoid anyway)
            self._fwd(m2, p + "mlp.c_fc.weight", gact, M, "conv1d", bias=self.P(p + "mlp.c_fc.bias"),
                      epi=hip.EPI_GELU, aux2=u, flags=hip.GEMM_GELU_GRAD if _GELU_GRAD else 0)
            xout = self.buf(f"resid_{l + 1}", (M, D))
            self._fwd(gact, p + "mlp.c_proj.weight", xout, M, "conv1d", bias=self.P(p + "mlp.c_proj.bias"),
                      epi=hip.EPI_RESID, aux=xmid, ldaux=D, drop_p=pr, drop_seed=s[2])
            layers.append((hcur, mu1, rs1, a1, qkv, ctx, lse, xmid, mu2, rs2, m2, u, gact, s, None))
            hcur = xout
        if not need_logits:      # the decoder's prompt prefill: only the blocks' K / V rows (layers[l][4]) are wanted
            a.update(c=c, kl=kl, type_ids=type_ids, keep=keep, layers=layers, x_last=hcur, logits=None)
            self.act = a
            return a
        muf = self.buf("lnf_mu", (M,), torch.float32)
        rsf = self.buf("lnf_rs", (M,), torch.float32)
        Vp = self.layout.Vpad
        # fp32 logits for the reference-shaped surface (MMTG.forward returns them); the fused trainer of the
        # bf16 mode keeps them in bf16 like every other activation (logits_f32=False)
        l32 = logits_f32 or self.dtype == hip.F32
        logits = self.buf("logits" if l32 else "logits_c", (M, Vp), torch.float32 if l32 else self.tdt)
        if x3:
            hf = self.pbuf("hf", M, D)
            if self.hybrid:
                a["x_last_b"] = self.buf("x_last_b", (M, D), torch.bfloat16)
            hip.layernorm_fwd_x3(hcur, hf, self.P(pre + "ln_f.weight"), self.P(pre + "ln_f.bias"), muf, rsf, M, D, sh.eps, xb=a.get("x_last_b"))
            hip.gemm_x3(hf, self.Wpx("wte", Vp, D), logits, M, Vp, D)
        else:
            hf = self.buf("hf", (M, D))
            hip.layernorm_fwd(hcur, hf, self.P(pre + "ln_f.weight"), self.P(pre + "ln_f.bias"), muf, rsf, M, D, sh.eps)
            hip.gemm(hf, self.Wp("wte"), logits, M, Vp, D, transB=True, ldb=D, out_f32=l32)
        a.update(xt=xt, t_raw=t_raw, t_ln=t_ln, st=st, enc=enc, alpha=alpha, kl=kl, o=o, ba=ba, c=c, x=x, ids32=ids32, h1=h1,
                 type_ids=type_ids, keep=keep, layers=layers, x_last=hcur, muf=muf, rsf=rsf, hf=hf, logits=logits)
        self.act = a
        return a

    def _infer_types_mask(self, batch, targets, per_row):
        """Inference branch of GPT2_Decoder.forward (model.py:290-312), vectorised: lyric
        position i has type 0 if (i+1) % sent in {0,1} or the token is PAD, else
        [1..max_sent_num-1, 1][i // sent]; mask 0 on PAD.  The reference decides from row 0
        only (batch 1); per_row=True applies the rule to every row (batched decode)."""
        sh = self.sh
        sent = sh.msl + 2
        n = targets.shape[1]
        Bn = targets.shape[0]
        src = targets if per_row else targets[:1].expand(Bn, n)
        tlist = list(range(1, sh.max_seq_length // sent + 1)) + [1]
        idx = torch.arange(n, device=self.dev)
        slot = torch.tensor([tlist[min(i // sent, len(tlist) - 1)] for i in range(n)], device=self.dev, dtype=torch.long)
        edge = ((idx + 1) % sent == 0) | ((idx + 1) % sent == 1)
        is_pad = src == 0
        types = torch.where(edge[None, :] | is_pad, torch.zeros_like(src), slot[None, :].expand(Bn, n))
        mask = (~is_pad).long()
        return (torch.cat([batch["tpw_type_ids"].to(self.dev).long(), types], 1),
                torch.cat([batch["tpw_attention_mask"].to(self.dev).long(), mask], 1))

    # ---------------------------------------------------------------- loss
    def loss(self, ratings=None, stage=3, batch_den=None, label_zero=False):
        """MyLoss (loss.py:45-74) + GPT-2's internal LM loss on the forward's logits.
        scalars[0] = MyLoss, scalars[1] = LM loss; also fills coef for loss_backward()."""
        a, sh = self.act, self.sh
        B, L, M = a["B"], a["L"], a["M"]
        nll = self.buf("nll", (M,), torch.float32)
        lse = self.buf("lse_rows", (M,), torch.float32)
        ce = self.buf("sample_ce", (B,), torch.float32)
        coef = self.buf("coef", (B,), torch.float32)
        sc = self.buf("loss_scalars", (2,), torch.float32)
        r = None if ratings is None else ratings.to(self.dev, torch.long).contiguous()
        hip.loss_fwd(a["logits"], self.layout.Vpad, sh.V, a["topic_ids"], a["targets"], r, stage, label_zero,
                     B, sh.P, L, float(B if batch_den is None else batch_den), nll, lse, ce, coef, sc)
        a.update(nll=nll, lse_rows=lse, coef=coef, scalars=sc)
        return sc

    def loss_backward(self, gscale=1.0, lm_coef=0.0):
        """d(gscale * MyLoss + lm_scale * LM loss)/d logits into the engine's dlogits buffer."""
        a, sh = self.act, self.sh
        if a.get("x3") and self.hybrid:
            # hybrid: fp32 logits in, the gradient as bf16 rows (the bf16 backward's LM-head operand)
            dl = self.buf("dlogits_b", (a["M"], self.layout.Vpad), torch.bfloat16)
            hip.loss_bwd(a["logits"], self.layout.Vpad, sh.V, a["topic_ids"], a["targets"], a["lse_rows"], a["coef"],
                         gscale, a["B"], sh.P, a["L"], dl, self.layout.Vpad, self.layout.Vpad, lm_coef=lm_coef)
            return dl
        if a.get("x3"):
            # x3: the gradient goes straight into the plane pair the LM head's split-precision products read
            dlp = self.pbuf("dlogits_p", a["M"], self.layout.Vpad)
            hip.loss_bwd_x3(a["logits"], self.layout.Vpad, sh.V, a["topic_ids"], a["targets"], a["lse_rows"], a["coef"],
                            gscale, a["B"], sh.P, a["L"], dlp, self.layout.Vpad, lm_coef=lm_coef)
            return dlp
        # compute-dtype logits are overwritten in place by their gradient
        dl = a["logits"] if a["logits"].dtype == self.tdt and self.dtype != hip.F32 else self.buf("dlogits", (a["M"], self.layout.Vpad))
        hip.loss_bwd(a["logits"], self.layout.Vpad, sh.V, a["topic_ids"], a["targets"], a["lse_rows"], a["coef"],
                     gscale, a["B"], sh.P, a["L"], dl, self.layout.Vpad, self.layout.Vpad, lm_coef=lm_coef)
        return dl

    def dlogits_from(self, g32):
        """External d(logits) [B,T,V] fp32 (drop-in autograd path) -> padded compute-dtype buffer."""
        a = self.act
        dl = self.buf("dlogits_b", (a["M"], self.layout.Vpad), torch.bfloat16) if (self.hybrid and a.get("x3")) else self.buf("dlogits", (a["M"], self.layout.Vpad))
        g32 = g32.contiguous().view(a["M"], self.sh.V)
        hip.cast_pad_rows(g32, self.sh.V, dl, self.layout.Vpad, a["M"], self.sh.V)
        return dl

    # ---------------------------------------------------------------- backward
    def _prefetch(self, *tensors):
        """Saved activations are cold by the time the backward reads them (4.5 GB of activations against a 256 MB
        Infinity Cache).  Called right before kernel k is enqueued: a side stream waits for that point and pulls the
        operands of the kernels after k into the cache while k runs (mmtg_prefetch).  MEASURED NEGATIVE (round 2, same box):
        16.41 ms per step without, 17.2-17.7 with 64 / 256 / 1024 prefetch workgroups -- the side-stream kernels take
        workgroup slots and L2 bandwidth from the product they run beside -- so it is off unless MMTG_PREFETCH=<workgroups>."""
        if not _PREFETCH:
            return
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=self.dev, priority=-1)
            self._pf_sink = torch.zeros(1, device=self.dev, dtype=torch.int32)
        ev = torch.cuda.Event()
        ev.record()
        self._pf_stream.wait_event(ev)
        for t in tensors:
            hip.prefetch(t, self._pf_sink, _PREFETCH, stream=self._pf_stream)

    @contextlib.contextmanager
    def _beside(self, on):
        """MMTG_ENC_STREAMS=1: enqueue the body on the second stream (forked from the current one) with workspaces of its own --
        every buffer it names through buf() carries a tag -- so that it runs beside what the current stream enqueues next;
        `_join_beside` makes the current stream wait for it.  Off (default) / on=False: a no-op."""
        if not (on and _ENC_STREAMS):
            yield
            return
        side = self._side_stream()
        side.wait_stream(torch.cuda.current_stream())
        self._ws_tag, self._beside_live = "~s", True
        try:
            with torch.cuda.stream(side):
                yield
        finally:
            self._ws_tag = ""

    def _join_beside(self):
        if getattr(self, "_beside_live", False):
            torch.cuda.current_stream().wait_stream(self._side)
            self._beside_live = False

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
            self._events = {}
        return self._side

    def _event(self, i):
        ev = self._events.get(i)
        if ev is None:
            ev = self._events[i] = torch.cuda.Event()
        return ev

    def _ready(self, pack):
        if self.bucket_hook is not None:
            self._flush_sums()          # the gradients the hook may hand to the exchange must be final: pending column sums first
            self.bucket_hook(pack)

    # ---- deferred column sums (mmtg_colsum_batch): the sources are per-site workspaces that nothing rewrites before the flush
    def _defer_sum(self, X, ldx, M, N, out, offset=0):
        """out[c] += sum_{r < M} X[offset + r * ldx + c], c < N, at the next flush (fp32 X, M <= 2048)."""
        self._sums.append((X.data_ptr() + 4 * offset, out.data_ptr(), int(ldx), int(M), int(N)))

    def _flush_sums(self):
        if self._sums:
            # the items of a flush run concurrently: no two may add into the same output elements
            spans = sorted((it[1], it[1] + 4 * it[4]) for it in self._sums)
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                if b0 < a1:
                    raise RuntimeError("deferred column sums with overlapping outputs in one batch")
            hip.colsum_batch(self._sums)
            self._sums = []

    def _ln_bwd_x3(self, site, dy, x, gamma_key, mean, rstd, dres, dx, rows, cols, dx_planes, drop_p=0.0, drop_seed=0, colsum_key=None):
        """_ln_bwd for the split-precision path (fp32 rows, the masked gradient written as a plane pair)."""
        g = gamma_key[:-len("weight")]
        if self._defer:
            ws = self.buf("ln_bwd_ws_" + site, (hip.lib().mmtg_layernorm_bwd_ws(rows, cols),), torch.float32)
            nb = hip.layernorm_bwd_x3_partial(dy, x, self.P(gamma_key), mean, rstd, dres, dx, rows, cols, dx_planes, ws,
                                              drop_p=drop_p, drop_seed=drop_seed, want_colsum=colsum_key is not None)
            self._defer_sum(ws, 3 * cols, nb, cols, self.G(gamma_key))
            self._defer_sum(ws, 3 * cols, nb, cols, self.G(g + "bias"), offset=cols)
            if colsum_key is not None:
                self._defer_sum(ws, 3 * cols, nb, cols, self.G(colsum_key), offset=2 * cols)
        else:
            lnws = self.buf("ln_bwd_ws", (hip.lib().mmtg_layernorm_bwd_ws(rows, max(self.sh.D, self.sh.H)),), torch.float32)
            hip.layernorm_bwd_x3(dy, x, self.P(gamma_key), mean, rstd, dres, dx, self.G(gamma_key), self.G(g + "bias"), rows, cols, dx_planes,
                                 drop_p=drop_p, drop_seed=drop_seed, dcolsum=None if colsum_key is None else self.G(colsum_key), ws=lnws)

    def _ln_bwd(self, site, dy, x, gamma_key, mean, rstd, dres, dx, rows, cols, dx_masked=None, drop_p=0.0, drop_seed=0, colsum_key=None):
        """LayerNorm backward of the bf16 decoder path: with deferred sums the first stage only, into this site's own partial-row
        workspace, the three gradients it ends in queued for the batched sum; otherwise the one-call form on the shared workspace."""
        g = gamma_key[:-len("weight")]
        if self._defer:
            ws = self.buf("ln_bwd_ws_" + site, (hip.lib().mmtg_layernorm_bwd_ws(rows, cols),), torch.float32)
            nb = hip.layernorm_bwd_partial(dy, x, self.P(gamma_key), mean, rstd, dres, dx, rows, cols, ws, dx_masked=dx_masked,
                                           drop_p=drop_p, drop_seed=drop_seed, want_colsum=colsum_key is not None)
            self._defer_sum(ws, 3 * cols, nb, cols, self.G(gamma_key))
            self._defer_sum(ws, 3 * cols, nb, cols, self.G(g + "bias"), offset=cols)
            if colsum_key is not None:
                self._defer_sum(ws, 3 * cols, nb, cols, self.G(colsum_key), offset=2 * cols)
        else:
            lnws = self.buf("ln_bwd_ws", (hip.lib().mmtg_layernorm_bwd_ws(rows, max(self.sh.D, self.sh.H)),), torch.float32)
            hip.layernorm_bwd(dy, x, self.P(gamma_key), mean, rstd, dres, dx, self.G(gamma_key), self.G(g + "bias"), rows, cols,
                              dx_masked=dx_masked, drop_p=drop_p, drop_seed=drop_seed,
                              dcolsum=None if colsum_key is None else self.G(colsum_key), ws=lnws)

    @contextlib.contextmanager
    def _as_bf16(self):
        """The decoder part of the hybrid mode's backward runs the bf16 mode's code path on an fp32-storage engine: for its
        duration the engine presents bf16 storage -- buffers default to bf16, W() / Wt() / the [D, Vpad] copy are the HI planes of
        the weight plane pairs (hi = bf16(w): the bf16 mode's own weight copy, bit for bit)."""
        saved = (self.dtype, self.tdt, self.x3, self.wc, self.wt, self.wte_t)
        self.dtype, self.tdt, self.x3 = hip.BF16, torch.bfloat16, False
        self.wc = self.wc2[0]
        self.wt = None if saved[4] is None else saved[4][0]
        self.wte_t = None if saved[5] is None else saved[5][0]
        try:
            yield
        finally:
            self.dtype, self.tdt, self.x3, self.wc, self.wt, self.wte_t = saved

    def _backward_hybrid(self, dlogits, dkl):
        """compute_dtype "bf16x3f": d(logits) through the decoder as ONE bf16 pass per product -- the bf16 mode's kernels over the hi
        planes the split-precision forward stored (LayerNorm outputs, qkv, attention context, GELU output, projector activations),
        the bf16 copies of the LayerNorm inputs and of the c_fc pre-activation it wrote beside them, and the hi planes of the
        weights -- then the encoder / fuser backward, on the bf16 kernels as well (MMTG_HYBRID_ENC_BF16=0: in exact fp32 as in the x3
        mode).  Gradients land in the same flat fp32 buffer."""
        a, sh = self.act, self.sh
        M, Vp = a["M"], self.layout.Vpad
        if isinstance(dlogits, hip.Planes):
            dlogits = dlogits.t[0]
        if dlogits.dtype != torch.bfloat16:
            dlb = self.buf("dlogits_b", (M, Vp), torch.bfloat16)
            hip.cast_f32_to(dlogits, dlb, M * Vp)
            dlogits = dlb
        hi = lambda pl: pl.t[0]
        layers = []
        for rec, (xb1, xb2) in zip(a["layers"], a["hyb_rows"]):
            (xin, mu1, rs1, a1, qkv, ctx, lse, xmid, mu2, rs2, m2, u, gact, s_, ctxp) = rec
            layers.append((xb1, mu1, rs1, hi(a1), hi(qkv), hi(ctxp), lse, xb2, mu2, rs2, hi(m2), u, hi(gact), s_, None))
        ab = dict(a)
        if a.get("px3"):
            h1b, xb = hi(a["h1p"]), hi(a["xp"])
        else:       # (projector widths the split-precision products do not tile: fp32 rows, cast once)
            h1b = self.buf("h1_b", tuple(a["h1"].shape), torch.bfloat16)
            hip.cast_f32_to(a["h1"], h1b, a["h1"].numel())
            xb = self.buf("x_cond_b", tuple(a["x"].shape), torch.bfloat16)
            hip.cast_f32_to(a["x"], xb, a["x"].numel())
        ab.update(x3=False, px3=False, gelu_grad=False, elem_mask=True, layers=layers, x_last=a["x_last_b"], hf=hi(a["hf"]), h1=h1b, x=xb, ids32=None)
        try:
            with self._as_bf16():
                self.act = ab
                seg = self._backward_decoder(ab, dlogits)
        finally:
            self.act = a
        if _HYBRID_ENC_BF16:
            # the encoder / fuser backward on the bf16 kernels too (the mode's gradients are bf16-accurate by contract): bf16 copies of
            # the ~14 small activations it reads (320 x 512 ... 320 x 2048), then the same code path under the bf16 presentation --
            # 0.8 ms of exact-fp32 products (14 TFLOP/s on 34 few-row launches) become ~0.2
            ae = self._encoder_record_bf16(a)
            with self._as_bf16():
                self._backward_encoder(ae, seg, dkl)
            return
        seg32 = self.buf("d_seg32", tuple(seg.shape), torch.float32)
        hip.cast_to_f32(seg, seg32, seg.numel())
        self._backward_encoder(a, seg32, dkl)

    def _encoder_record_bf16(self, a):
        """The activation record `_backward_encoder` reads, with every fp32 tensor it takes as a kernel operand replaced by a bf16
        copy (statistics, attention probabilities, cell saves and fuser weights stay fp32: the bf16 kernels read them as fp32 too)."""
        def cast(t, name):
            b = self.buf("hyb_" + name, tuple(t.shape), torch.bfloat16)
            hip.cast_f32_to(t, b, t.numel())
            return b
        ae = dict(a)
        for k in ("o", "t_ln", "t_raw", "xt"):
            ae[k] = cast(a[k], k)
        ae["alpha"] = {mod: (cast(q, "aqkv_" + mod), cast(c, "actx_" + mod), p) for mod, (q, c, p) in a["alpha"].items()}
        enc = {}
        for ch, (layers, h_ln) in a["enc"].items():
            recs = []
            for l, rec in enumerate(layers):
                r = dict(rec)
                r["h"] = cast(rec["h"], "h_%s%d" % (ch, l))
                r["inp"] = recs[l - 1]["h"] if (l > 0 and rec["inp"] is layers[l - 1]["h"]) else cast(rec["inp"], "inp_%s%d" % (ch, l))
                recs.append(r)
            enc[ch] = (recs, cast(h_ln, "hln_" + ch))
        ae["enc"] = enc
        return ae

    def backward(self, dlogits, dkl=0.0):
        """Back-propagate d(logits) [M,Vpad] (compute dtype) and d(kl) through the whole
        model; parameter gradients are ACCUMULATED into the flat fp32 buffer."""
        a = self.act
        self.ensure_grad()
        if self._lazy is not None and not self.wgrad_overwrite:
            # (a backward that accumulates after a zero_grad(shape_key) that relied on overwrites: zero what it skipped)
            self.grad.zero_()
            self._lazy = None
        if self.hybrid and a.get("x3"):
            self._backward_hybrid(dlogits, dkl)
        else:
            self._backward_encoder(a, self._backward_decoder(a, dlogits), dkl)
        if self._tail_side is not None:         # the weight gradients that ran beside the tail: final before anything reads the gradient
            torch.cuda.current_stream().wait_stream(self._tail_side)
            self._tail_side = None
        self._lazy = None

    def _backward_decoder(self, a, dlogits):
        """LM head -> GPT-2 blocks -> input embedding -> projector; returns the per-segment sums of d(h1_pre) [B*S, H] that the
        experience vectors' gradient is made of."""
        sh = self.sh
        B, T, M, L = a["B"], a["T"], a["M"], a["L"]
        S, E, H, D, P = sh.S, sh.E, sh.H, sh.D, sh.P
        pe, pa, pr = a["pdrop"]
        Vp = self.layout.Vpad
        pre = "decoder.gpt2.transformer."
        # ---- LM head (tied wte)
        dhf = self.buf("d_hf", (M, D))
        x3 = bool(a.get("x3"))
        if x3:
            # split-precision backward: d(logits) as a plane pair feeds both the LM head's dgrad (through the [D, Vpad] copy of the
            # tied embedding) and its weight gradient (grouped kernel, config 2: written, not accumulated with atomics)
            dlp = dlogits if isinstance(dlogits, hip.Planes) else hip.split_planes(dlogits, M, Vp, self.pbuf("dlogits_p", M, Vp))
            hip.gemm_x3(dlp, hip.Planes(self.wte_t, D, Vp), dhf, M, D, Vp)
            tiles = hip.wgrad_group_sizes(((Vp, D),), 1, 0)[0]
            hs = _LMHEAD_GROUP_SPLITS or _group_splits_x3(tiles, M)
            _, nws, ncnt = hip.wgrad_group_sizes(((Vp, D),), hs, 0)
            hws = self.buf("wgrad_group_ws_head", (nws,), torch.float32) if hs > 1 else None
            hcnt = self.ws.get(("wgrad_group_cnt", torch.int32))
            if hcnt is None or hcnt.numel() < ncnt:
                hcnt = self.ws[("wgrad_group_cnt", torch.int32)] = torch.zeros(ncnt, device=self.dev, dtype=torch.int32)
            hip.wgrad_group([(dlp, a["hf"], self.Gp("wte"), Vp, D, Vp, D, D)], M, hs, hws, hcnt, accumulate=not self.wgrad_overwrite, config=_X3_WG_CFG)
            if self.wgrad_overwrite and self._ow_rec is not None:
                self._ow_rec[1].append((self.layout.pack_range["wte"][0], Vp * D))
        elif getattr(self, "wte_t", None) is not None and Vp % 128 == 0 and not self.x3:      # (x3: wte_t is a plane pair)
            hip.gemm(dlogits, self.wte_t, dhf, M, D, Vp, transB=True, ldb=Vp)
        else:
            hip.gemm(dlogits, self.Wp("wte"), dhf, M, D, Vp, transB=False, ldb=D)
        # (Vpad rows: the pad columns of dlogits are zero, so the pad rows of the pack receive +0)
        if x3:
            pass
        elif _WGRAD_GROUP and _LMHEAD_GROUP and self.dtype == hip.BF16 and M >= 256 and D % 8 == 0 and Vp % 8 == 0:
            # the tied embedding's gradient through the grouped kernel too: no fp32 atomics (bit-reproducible), and the
            # gradient is WRITTEN, so the lazy zero_grad can skip its 41 MB (the type-embedding rows are added later)
            tiles = hip.wgrad_group_sizes(((Vp, D),), 1, 0)[0]
            hs = _LMHEAD_GROUP_SPLITS or _group_splits(tiles, M)
            _, nws, ncnt = hip.wgrad_group_sizes(((Vp, D),), hs, 0)
            # (a workspace of its own: under MMTG_WGRAD_STREAM this launch may still be pending on the side stream when the
            #  block launches size theirs -- a shared name could reallocate, i.e. free, the buffer under it)
            hws = self.buf("wgrad_group_ws_head", (nws,), torch.float32) if hs > 1 else None
            hcnt = self.ws.get(("wgrad_group_cnt", torch.int32))
            if hcnt is None or hcnt.numel() < ncnt:
                hcnt = self.ws[("wgrad_group_cnt", torch.int32)] = torch.zeros(ncnt, device=self.dev, dtype=torch.int32)
            side = self._side_stream() if (_WGRAD_STREAM and pr > 0) else None
            if side is not None:        # beside the LM head's dgrad, ln_f and the last block's chain (reads dlogits / hf only)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    hip.wgrad_group([(dlogits, a["hf"], self.Gp("wte"), Vp, D, Vp, D, D)], M, hs, hws, hcnt, accumulate=not self.wgrad_overwrite)
            else:
                hip.wgrad_group([(dlogits, a["hf"], self.Gp("wte"), Vp, D, Vp, D, D)], M, hs, hws, hcnt, accumulate=not self.wgrad_overwrite)
            if self.wgrad_overwrite and self._ow_rec is not None:
                self._ow_rec[1].append((self.layout.pack_range["wte"][0], Vp * D))
        else:
            splits = _wgrad_splits(Vp, D, M, self.dtype == hip.BF16)
            if self.dtype == hip.F32 and not self.x3 and _WGRAD_SLAB and splits > 1:
                # exact-fp32 mode: K-split slabs summed in order ON TOP of what the gradient already holds (no fp32 atomics)
                part = self.buf("wgrad_slabs", (splits * Vp * D,), torch.float32)
                hip.gemm(dlogits, a["hf"], part, Vp, D, M, transA=True, transB=False, lda=Vp, ldb=D, ldc=D,
                         epi=hip.EPI_SPLIT, out_f32=True, splits=splits)
                hip.slab_sum(part, splits, Vp * D, self.Gp("wte"), Vp * D, accumulate=True)
            else:
                hip.gemm(dlogits, a["hf"], self.Gp("wte"), Vp, D, M, transA=True, transB=False, lda=Vp, ldb=D, ldc=D,
                         epi=hip.EPI_ATOMIC, splits=splits)
        # Every LayerNorm backward on the residual stream also emits, in the same pass, the
        # dropout-masked gradient entering the previous residual branch and that branch's bias
        # gradient (column sum) -- see mmtg_layernorm_bwd.
        self._defer = _DEFER_SUMS and (self.dtype == hip.BF16 or x3)
        dx = self.buf("d_resid_a", (M, D))
        dx2 = self.buf("d_resid_b", (M, D))
        # One grouped launch per block for its four weight gradients (mmtg_wgrad_group): the mlp.c_proj product's dy must then
        # outlive the LayerNorm backward that produces the attention c_proj's dy, so the masked gradients alternate
        # between two buffers.
        group = (_WGRAD_GROUP and self.dtype == hip.BF16 and M >= 256 and D % 8 == 0) or x3
        dmask = self.buf("d_masked", (M, D)) if (pr > 0 and not x3) else None
        dmask_b = (self.buf("d_masked_b", (M, D)) if group else dmask) if (pr > 0 and not x3) else None
        if group:
            gshapes = ((D, 4 * D), (4 * D, D), (D, D), (D, 3 * D))
            gcfg = 1 if (_WGRAD_GROUP_CFG and D >= 256 and not x3) else 0
            gtiles = hip.wgrad_group_sizes(gshapes, 1, gcfg)[0]
            gsplits = _group_splits_x3(gtiles, M) if x3 else _group_splits(gtiles, M, 256 if gcfg else 1024)
            _, nws, ncnt = hip.wgrad_group_sizes(gshapes, gsplits, gcfg)
            gws = self.buf("wgrad_group_ws", (nws,), torch.float32) if gsplits > 1 else None
            gcnt = self.ws.get(("wgrad_group_cnt", torch.int32))
            if gcnt is None or gcnt.numel() < ncnt:      # zero once: every launch leaves the counters zeroed
                gcnt = self.ws[("wgrad_group_cnt", torch.int32)] = torch.zeros(ncnt, device=self.dev, dtype=torch.int32)
        lastp = f"{pre}h.{sh.L - 1}."
        # (MMTG_WGRAD_STREAM: the top block's masked gradient goes into the buffer set of that block's parity, see below)
        dmask_top = self.buf("d_masked_1", (M, D)) if (group and not x3 and _WGRAD_STREAM and pr > 0 and (sh.L - 1) & 1) else dmask
        # MMTG_WGRAD_TAIL: the last blocks' operand buffers are their own (nothing rewrites them before their launch at the tail)
        # (the split-precision blocks' operands are plane pairs the LayerNorm backward writes with or without dropout: name suffix per block)
        ntail = min(_WGRAD_TAIL, sh.L) if (group and not _WGRAD_STREAM and (pr > 0 or x3) and self.bucket_hook is None) else 0
        tsets = {} if x3 else {t: (self.buf("d_u_t%d" % t, (M, 4 * D)), self.buf("d_masked_t%d" % t, (M, D)),
                                   self.buf("d_masked_b_t%d" % t, (M, D)), self.buf("d_qkv_t%d" % t, (M, 3 * D))) for t in range(ntail)}
        xsfx = (lambda t: "_t%d" % t if t < ntail else "") if x3 else (lambda t: "")
        if ntail == sh.L and not x3:
            dmask_top = tsets[sh.L - 1][1]
        self._tail_jobs = []
        if x3:
            # (x3: every LayerNorm backward writes the masked gradient entering the previous residual branch as the plane pair
            #  that branch's split-precision products read -- no fp32 copy, no separate split pass)
            dyp = self.pbuf("d_masked_p" + xsfx(sh.L - 1), M, D)
            self._ln_bwd_x3("f", dhf, a["x_last"], pre + "ln_f.weight", a["muf"], a["rsf"], None, dx, M, D, dyp,
                            drop_p=pr, drop_seed=a["layers"][sh.L - 1][13][2], colsum_key=lastp + "mlp.c_proj.bias")
        else:
            self._ln_bwd("f", dhf, a["x_last"], pre + "ln_f.weight", a["muf"], a["rsf"], None, dx, M, D,
                         dx_masked=dmask_top, drop_p=pr, drop_seed=a["layers"][sh.L - 1][13][2], colsum_key=lastp + "mlp.c_proj.bias")
        self._ready("ln_f.b")
        du = None if x3 else self.buf("d_u", (M, 4 * D))
        dm = self.buf("d_m", (M, D))
        dctx = self.buf("d_ctx", (M, D))
        dqkv = self.buf("d_qkv", (M, 3 * D))
        # MMTG_WGRAD_STREAM: block l's grouped weight gradients run on a side stream while the main stream already walks block
        # l - 1 -- the operands the chain produces (du, the two masked residual gradients, dqkv) then live in two buffer sets that
        # alternate block by block, and the main stream waits for block l + 1's launch before it rewrites that block's set.
        stream_mode = group and _WGRAD_STREAM and pr > 0 and not x3
        if stream_mode:
            side = self._side_stream()
            odd = (self.buf("d_u_1", (M, 4 * D)), self.buf("d_masked_1", (M, D)), self.buf("d_masked_b_1", (M, D)),
                   self.buf("d_qkv_1", (M, 3 * D)))
            sets = ((du, dmask, dmask_b, dqkv), odd)
            wdone = {}
        da = self.buf("d_a", (M, D))
        delta = self.buf("attn_delta", (M, sh.nH), torch.float32)
        # (x3: one [M, D] buffer per block of 128 keys -- plain stores summed in block order, no atomics)
        dq32 = self.buf("attn_dq32", (hip.attn_bwd_x3_dq_floats(B, T, D),) if x3 else (M, D), torch.float32)
        for l in range(sh.L - 1, -1, -1):
            p = f"{pre}h.{l}."
            (xin, mu1, rs1, a1, qkv, ctx, lse, xmid, mu2, rs2, m2, u, gact, s, ctxp) = a["layers"][l]
            if stream_mode:
                du, dmask, dmask_b, dqkv = sets[l & 1]
            if l in tsets:
                du, dmask, dmask_b, dqkv = tsets[l]
            if x3:
                # ---- split-precision block backward: every gradient that feeds a product travels as a plane pair
                bands = self.buf("d_u_bands_%d" % l if self._defer else "d_u_bands", ((M + 63) // 64, 4 * D), torch.float32)
                dup = self.pbuf("d_u_p" + xsfx(l), M, 4 * D)
                self._dgrad_x3(dyp, p + "mlp.c_proj.weight", None, M, planes=dup, ldc=4 * D, epi=hip.EPI_DGELU, aux=u, ldaux=4 * D, aux2=bands)
                if self._defer and bands.shape[0] <= 2048:
                    self._defer_sum(bands, 4 * D, bands.shape[0], 4 * D, self.G(p + "mlp.c_fc.bias"))
                else:
                    hip.colsum(bands, bands.shape[0], 4 * D, self.G(p + "mlp.c_fc.bias"))
                self._dgrad_x3(dup, p + "mlp.c_fc.weight", dm, M)
                dy2p = self.pbuf("d_masked_b_p" + xsfx(l), M, D)
                self._ln_bwd_x3("%d_2" % l, dm, xmid, p + "ln_2.weight", mu2, rs2, dx, dx2, M, D, dy2p,
                                drop_p=pr, drop_seed=s[1], colsum_key=p + "attn.c_proj.bias")
                dqkvp = self.pbuf("d_qkv_p" + xsfx(l), M, 3 * D)
                # d(ctx) as a plane pair only (the attention backward is its one reader); the dgrad's epilogue also emits delta =
                # rowsum(d ctx * ctx) per head: no separate pass over ctx / d ctx
                dctxp = self.pbuf("d_ctx_p", M, D)
                self._dgrad_x3(dy2p, p + "attn.c_proj.weight", None, M, planes=dctxp, ldc=D, epi=hip.EPI_ROWDOT, aux=ctx, ldaux=D, aux2=delta)
                # (deferred sums: the partial bias rows -- k / v parts per key block, q part per 16-row band -- stay in this block's own scratch)
                nkv, nq = B * (-(-T // 128)), -(-M // 16)
                brows = self._defer and nkv <= 2048 and nq <= 2048
                bws = self.buf("attn_dbias_x3_%d" % l if brows else "attn_dbias_x3", (hip.attn_bwd_x3_ws(B, T, D),), torch.float32)
                hip.attn_bwd_x3(qkv, a["keep"], ctx, dctxp, lse, delta, dq32, dqkvp, B, T, sh.nH, D // sh.nH, drop_p=pa, drop_seed=s[0],
                                dbias=None if brows else self.G(p + "attn.c_attn.bias"), delta_ready=True, dbias_ws=bws)
                if brows:
                    # (two items must not share output columns -- they run in one launch: the key-block rows carry the k and v parts,
                    #  their q columns are zero; the band rows carry the q part)
                    self._defer_sum(bws, 3 * D, nkv, 2 * D, self.G(p + "attn.c_attn.bias")[D:], offset=D)
                    self._defer_sum(bws, D, nq, D, self.G(p + "attn.c_attn.bias"), offset=nkv * 3 * D)
                self._dgrad_x3(dqkvp, p + "attn.c_attn.weight", da, M)
                keys = (p + "mlp.c_fc.weight", p + "mlp.c_proj.weight", p + "attn.c_proj.weight", p + "attn.c_attn.weight")
                probs = [(m2, dup, self.G(keys[0]), D, 4 * D), (gact, dyp, self.G(keys[1]), 4 * D, D),
                         (ctxp, dy2p, self.G(keys[2]), D, D), (a1, dqkvp, self.G(keys[3]), D, 3 * D)]
                if l < ntail:
                    self._tail_jobs.append((probs, M, gsplits, gws, gcnt, not self.wgrad_overwrite, _X3_WG_CFG))       # launched at the tail
                else:
                    hip.wgrad_group(probs, M, gsplits, gws, gcnt, accumulate=not self.wgrad_overwrite, config=_X3_WG_CFG)
                if self.wgrad_overwrite and self._ow_rec is not None:
                    self._ow_rec[1].extend((self.layout.entries[k][0], self.layout.entries[k][2]) for k in keys)
                if l > 0:
                    dyp = self.pbuf("d_masked_p" + xsfx(l - 1), M, D)          # the planes block l - 1 reads (its own when it is deferred)
                    self._ln_bwd_x3("%d_1" % l, da, xin, p + "ln_1.weight", mu1, rs1, dx2, dx, M, D, dyp,
                                    drop_p=pr, drop_seed=a["layers"][l - 1][13][2], colsum_key=f"{pre}h.{l - 1}.mlp.c_proj.bias")
                else:
                    self._ln_bwd("0_1", da, xin, p + "ln_1.weight", mu1, rs1, dx2, dx, M, D)
                self._ready(p + "ln_1.bias")
                continue
            # x_out = x_mid + drop(gact W2 + b2): dy = dx * mask (already produced, with its bias gradient)
            dy = dmask if pr > 0 else dx
            # (the dGELU epilogue also emits the column sums of du per 64-row band: a [M/64, 4D] reduction
            #  gives the c_fc bias gradient instead of a pass over du)
            bands = self.buf("d_u_bands_%d" % l if self._defer else "d_u_bands", ((M + 63) // 64, 4 * D), torch.float32)
            self._prefetch(m2, gact)            # while dGELU runs: the operands of the two weight gradients after it
            self._dgrad(dy, p + "mlp.c_proj.weight", du, M, "conv1d", epi=hip.EPI_DGELU, aux=u, ldaux=4 * D, aux2=bands,
                        flags=hip.GEMM_GELU_GRAD if a["gelu_grad"] else 0)
            if self._defer and bands.shape[0] <= 2048:
                self._defer_sum(bands, 4 * D, bands.shape[0], 4 * D, self.G(p + "mlp.c_fc.bias"))
            else:
                hip.colsum(bands, bands.shape[0], 4 * D, self.G(p + "mlp.c_fc.bias"))
            # (both consumers of du run while it is still in the Infinity Cache; the c_proj weight gradient, whose
            #  operands come from HBM either way, goes last -- it must precede the LayerNorm backward, which reuses dmask)
            self._dgrad(du, p + "mlp.c_fc.weight", dm, M, "conv1d")
            dy_fc2 = dy
            if not group:
                self._wgrad(m2, du, p + "mlp.c_fc.weight", None, M, "conv1d")
                self._prefetch(xmid, ctx)           # while the c_proj weight gradient runs: LayerNorm input, attention context
                self._wgrad(gact, dy, p + "mlp.c_proj.weight", None, M, "conv1d")
            self._ln_bwd("%d_2" % l, dm, xmid, p + "ln_2.weight", mu2, rs2, dx, dx2, M, D,
                         dx_masked=dmask_b, drop_p=pr, drop_seed=s[1], colsum_key=p + "attn.c_proj.bias")
            # x_mid = x_in + drop(ctx Wp + bp)
            dy = dmask_b if pr > 0 else dx2
            # bf16: the dgrad GEMM's epilogue also emits delta = rowsum(d ctx * ctx) per head
            fuse_delta = self.dtype == hip.BF16 and M > 256
            if fuse_delta:
                self._dgrad(dy, p + "attn.c_proj.weight", dctx, M, "conv1d", epi=hip.EPI_ROWDOT, aux=ctx, ldaux=D, aux2=delta)
            else:
                self._dgrad(dy, p + "attn.c_proj.weight", dctx, M, "conv1d")
            self._prefetch(qkv)                 # while the c_proj weight gradient runs: the attention backward's rows
            if not group:
                self._wgrad(ctx, dy, p + "attn.c_proj.weight", None, M, "conv1d")
            # (hybrid: the split-precision forward drew the attention-dropout mask element by element; the whole-head backward
            #  kernels regenerate THAT mask instead of their own word masks)
            # (exact-fp32 mode: the tiled kernel combines its waves' bias sums with LDS atomics -- arrival order -- so the c_attn bias
            #  gradient is an ordered column sum over d(qkv) there instead: one more pass in the cross-check mode, reproducible bits)
            exact = self.dtype == hip.F32 and not self.x3 and _WGRAD_SLAB
            # (deferred sums: the whole-head kernels leave their B partial bias rows in this block's own scratch for the batched sum)
            brows = hip.attn_bwd_dbias_rows(self.dtype, B, T) if self._defer else 0
            brows = brows if 0 < brows <= 2048 else 0
            bws = None if exact else self.buf("attn_dbias_rows_%d" % l if brows else "attn_dbias_rows",
                                               (hip.attn_bwd_bias_rows(B, T, self.dtype), 3 * D), torch.float32)
            hip.attn_bwd(qkv, a["keep"], ctx, dctx, lse, delta, dq32, dqkv, B, T, sh.nH, D // sh.nH,
                         drop_p=pa, drop_seed=s[0], delta_ready=fuse_delta, dbias=None if exact else self.G(p + "attn.c_attn.bias"),
                         flags=(hip.ATTN_ELEM_MASK if a.get("elem_mask") else 0) | (hip.ATTN_DBIAS_ROWS if brows else 0), dbias_ws=bws)
            if brows:
                self._defer_sum(bws, 3 * D, brows, 3 * D, self.G(p + "attn.c_attn.bias"))
            if exact:
                hip.colsum(dqkv, M, 3 * D, self.G(p + "attn.c_attn.bias"))
            self._prefetch(a1, xin)             # while the c_attn dgrad runs: its weight gradient's operand, LayerNorm input
            self._dgrad(dqkv, p + "attn.c_attn.weight", da, M, "conv1d")
            if l > 0:
                self._prefetch(a["layers"][l - 1][11])      # the next layer's saved pre-activation (dGELU)
            if group:
                # gw[in, out] (+)= x^T dy for the block's four Conv1D layers: 432 tiles x 2 K halves at GPT-2 base, reduced in
                # the kernel (must run before the LayerNorm backward below, which rewrites dmask / dx for the next block)
                keys = (p + "mlp.c_fc.weight", p + "mlp.c_proj.weight", p + "attn.c_proj.weight", p + "attn.c_attn.weight")
                probs = [(m2, du, self.G(keys[0]), D, 4 * D), (gact, dy_fc2, self.G(keys[1]), 4 * D, D),
                         (ctx, dy, self.G(keys[2]), D, D), (a1, dqkv, self.G(keys[3]), D, 3 * D)]
                if stream_mode:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        hip.wgrad_group(probs, M, gsplits, gws, gcnt, accumulate=not self.wgrad_overwrite, config=gcfg)
                        wdone[l] = self._event(l)
                        wdone[l].record(side)
                elif l in tsets:
                    self._tail_jobs.append((probs, M, gsplits, gws, gcnt, not self.wgrad_overwrite, gcfg))       # launched at the tail
                else:
                    hip.wgrad_group(probs, M, gsplits, gws, gcnt, accumulate=not self.wgrad_overwrite, config=gcfg)
                if self.wgrad_overwrite and self._ow_rec is not None:
                    self._ow_rec[1].extend((self.layout.entries[k][0], self.layout.entries[k][2]) for k in keys)
            else:
                self._wgrad(a1, dqkv, p + "attn.c_attn.weight", None, M, "conv1d")
            if stream_mode and (l + 1) in wdone:
                # block l + 1's weight gradients have read the set this LayerNorm backward starts to rewrite; its gradients are
                # final only now (the data-parallel bucket hook must not see them earlier)
                torch.cuda.current_stream().wait_event(wdone.pop(l + 1))
                self._ready(f"{pre}h.{l + 1}.ln_1.bias")
            if l > 0:
                self._ln_bwd("%d_1" % l, da, xin, p + "ln_1.weight", mu1, rs1, dx2, dx, M, D,
                             dx_masked=(sets[(l - 1) & 1][1] if stream_mode else tsets[l - 1][1] if (l - 1) in tsets else dmask),
                             drop_p=pr, drop_seed=a["layers"][l - 1][13][2],
                             colsum_key=f"{pre}h.{l - 1}.mlp.c_proj.bias")
            else:
                self._ln_bwd("0_1", da, xin, p + "ln_1.weight", mu1, rs1, dx2, dx, M, D)
            if not stream_mode:
                self._ready(p + "ln_1.bias")
        if stream_mode:
            torch.cuda.current_stream().wait_stream(side)        # block 0's launch (and the tied embedding's, long done)
            for l_ in sorted(wdone, reverse=True):
                self._ready(f"{pre}h.{l_}.ln_1.bias")
        elif _WGRAD_STREAM and pr > 0 and getattr(self, "_side", None) is not None:
            torch.cuda.current_stream().wait_stream(self._side)  # (only the tied embedding's gradient ran on the side stream)
        # (deferred column sums: everything the blocks queued since the last hand-over point -- the whole decoder's without a bucket hook --
        #  in one launch, here, with the embedding / projector / encoder backward still to come behind it)
        self._flush_sums()
        self._defer = False
        if _TAIL_FORK_EARLY:            # (fork here: beside the embedding / projector backward too; MMTG_WGRAD_TAIL_FORK=proj: after the projector)
            self._launch_tail_jobs()
        # ---- GPT-2 input embedding: h0 = drop(g + wpe + wte[type])
        nty = min(32, sh.V)
        hip.embed_add_bwd(dx, a["type_ids"], self.G(pre + "wpe.weight"), self.G(pre + "wte.weight"), M, T, D,
                          nty, drop_p=pe, drop_seed=a["seed"],
                          ws=self.buf("embed_bwd_ws", (int(hip.lib().mmtg_embed_add_bwd_ws(M, D, nty)),), torch.float32))
        self._ready("wpe")
        # ---- projector (model.py:279-281)
        dh1 = self.buf("d_h1", (M, H))
        px3 = bool(a.get("px3"))
        if px3:
            # split-precision projector backward: d h1_pre = (dx W2) (1 - h1^2) through the [in,out] plane copy of W2, and BOTH
            # weight gradients (W2: dx^T h1, W1: d h1_pre^T X) in one grouped launch on plane pairs
            dxp = hip.split_planes(dx, M, D, self.pbuf("d_resid0_p", M, D))
            dh1p = self.pbuf("d_h1_p", M, H)
            hip.gemm_x3(dxp, self.Wtx("decoder.projector_layer2.weight"), dh1, M, H, D, planes=dh1p, epi=hip.EPI_DTANH, aux=a["h1"], ldaux=H)
            pshapes = ((D, H), (H, E))
            ptiles = hip.wgrad_group_sizes(pshapes, 1, 0)[0]
            psplits = _group_splits_x3(ptiles, M)
            _, pnws, pncnt = hip.wgrad_group_sizes(pshapes, psplits, 0)
            pws = self.buf("wgrad_group_ws_proj", (pnws,), torch.float32) if psplits > 1 else None
            # (counters of its own: the blocks' deferred launches may be running on the side stream by now -- _WGRAD_TAIL)
            pcnt = self.ws.get(("wgrad_group_cnt_proj", torch.int32))
            if pcnt is None or pcnt.numel() < pncnt:
                pcnt = self.ws[("wgrad_group_cnt_proj", torch.int32)] = torch.zeros(pncnt, device=self.dev, dtype=torch.int32)
            pkeys = ("decoder.projector_layer2.weight", "decoder.projector_layer1.weight")
            for k in pkeys:      # (tensors the lazy zero_grad skipped are overwritten below; otherwise accumulate as ever)
                if self._lazy is not None and (self.layout.entries[k][0], self.layout.entries[k][2]) in self._lazy and not self.wgrad_overwrite:
                    self.G(k).zero_()
            hip.wgrad_group([(dxp, a["h1p"], self.G(pkeys[0]), D, H), (dh1p, a["xp"], self.G(pkeys[1]), H, E)], M, psplits, pws, pcnt,
                            accumulate=not self.wgrad_overwrite, config=_X3_WG_CFG)
            if self.wgrad_overwrite and self._ow_rec is not None:
                self._ow_rec[1].extend((self.layout.entries[k][0], self.layout.entries[k][2]) for k in pkeys)
            hip.colsum(dx, M, D, self.G("decoder.projector_layer2.bias"))
        else:
            self._dgrad(dx, "decoder.projector_layer2.weight", dh1, M, "linear", epi=hip.EPI_DTANH, aux=a["h1"], ldaux=H)
            self._wgrad(a["h1"], dx, "decoder.projector_layer2.weight", "decoder.projector_layer2.bias", M, "linear")
        # d c[b,k] = (sum over the segment's tokens of d h1_pre) W1   (the add is linear)
        seg = self.buf("d_seg", (B * S, H))
        hip.segment_sum(dh1, seg, B, P, L, S, H, sh.two_sents)
        if a["ids32"] is not None:
            # d W1 = d h1_pre^T X with X = E[id] + c[seg] never materialised: the table rows enter the K-strided operand of
            # the weight-gradient product by index (mmtg_gemm_gather mode 1), the c part is the small product seg^T c
            gw = self.G("decoder.projector_layer1.weight")
            splits = _wgrad_splits(H, E, M, True)
            part = self.buf("wgrad_slabs", (splits * H * E,), torch.float32)
            hip.gemm_gather(1, dh1, self.table, part, H, E, M, a["ids32"], self.table.shape[0], lda=H, ldb=E, ldc=E,
                            epi=hip.EPI_SPLIT, splits=splits)
            hip.slab_sum(part, splits, H * E, gw, H * E, accumulate=not self.wgrad_overwrite)
            hip.gemm(seg, a["c"], gw, H, E, B * S, transA=True, transB=False, lda=H, ldb=E, ldc=E, epi=hip.EPI_ATOMIC, splits=1)
            hip.colsum(dh1, M, H, self.G("decoder.projector_layer1.bias"))
        elif px3:
            hip.colsum(dh1, M, H, self.G("decoder.projector_layer1.bias"))      # (the weight gradient went with W2's above)
        else:
            self._wgrad(a["x"], dh1, "decoder.projector_layer1.weight", "decoder.projector_layer1.bias", M, "linear")
        if not _TAIL_FORK_EARLY:
            self._launch_tail_jobs()
        return seg

    def _launch_tail_jobs(self):
        """The deferred blocks' grouped weight gradients, in block order on ONE side stream (they share a workspace and counters; the
        in-loop launches are complete by now): behind everything enqueued so far, beside what the main stream enqueues next."""
        if not self._tail_jobs:
            return
        if _os.environ.get("MMTG_WGRAD_TAIL_MAIN"):          # (A/B: the same launches on the main stream, no overlap with the tail)
            for probs, m_, gs_, gws_, gcnt_, acc_, cfg_ in self._tail_jobs:
                hip.wgrad_group(probs, m_, gs_, gws_, gcnt_, accumulate=acc_, config=cfg_)
            self._tail_jobs = []
            return
        side = self._side_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for probs, m_, gs_, gws_, gcnt_, acc_, cfg_ in self._tail_jobs:
                hip.wgrad_group(probs, m_, gs_, gws_, gcnt_, accumulate=acc_, config=cfg_)
        self._tail_jobs = []
        self._tail_side = side

    def _backward_encoder(self, a, seg, dkl):
        """d c = seg W1 -> beta fuser -> alpha attention + LayerNorm + recurrent channels -> topic channel."""
        sh = self.sh
        B, M = a["B"], a["M"]
        S, E, H, D = sh.S, sh.E, sh.H, sh.D
        lnws = self.buf("ln_bwd_ws", (hip.lib().mmtg_layernorm_bwd_ws(M, max(D, H)),), torch.float32)
        dc = self.buf("d_c", (B * S, E))
        self._dgrad(seg, "decoder.projector_layer1.weight", dc, B * S, "linear")
        # ---- beta fuser
        do = self.buf("d_o", (B * S, H))
        self._dgrad(dc, "mm_atten_layer.out_linear.weight", do, B * S, "linear")
        self._wgrad(a["o"], dc, "mm_atten_layer.out_linear.weight", "mm_atten_layer.out_linear.bias", B * S, "linear")
        dtopic = self.buf("d_topic32", (B, H), torch.float32, zero=True)
        dci = self.buf("d_actx_img", (B * S, H))
        dct = self.buf("d_actx_text", (B * S, H))
        hip.beta_fuse_bwd(a["t_ln"], a["alpha"]["img"][1], a["alpha"]["text"][1], self.Pp("att_w"), a["ba"], do,
                          dtopic, dci, dct, self.Gp("att_w"), self.Gp("att_b"), B, S, H)
        self._ready("att_b")
        # ---- alpha attention + LayerNorm + GRU per modality (MMTG_ENC_STREAMS=1: the text channel's chain on the second stream)
        for mod, ch, lnk, dctx_a in (("text", "text", "ln_layer3", dct), ("img", "image", "ln_layer2", dci)):
            with self._beside(mod == "text"):
                dhp32 = self.buf("d_hp32", (B, H), torch.float32)
                tmp32 = self.buf("d_tmp32", (B, H), torch.float32)
                lnws_m = self.buf("ln_bwd_ws", (hip.lib().mmtg_layernorm_bwd_ws(M, max(D, H)),), torch.float32)
                qkv_a, _, probs = a["alpha"][mod]
                layers, h_ln = a["enc"][ch]
                h_all = layers[-1]["h"]
                dqkv_a = self.buf("d_aqkv", (B * S, 3 * H))
                hip.alpha_attn_bwd(qkv_a, self.prior, probs, dctx_a, dkl, dqkv_a, B, S, H, sh.heads)
                dhln = self.buf("d_hln", (B * S, H))
                if not self._gemm_few_rows(dqkv_a, self.Wp(mod + "_qkv_w"), dhln, B * S, H, 3 * H, False, H):
                    hip.gemm(dqkv_a, self.Wp(mod + "_qkv_w"), dhln, B * S, H, 3 * H, transB=False, ldb=H)
                hip.gemm(dqkv_a, h_ln, self.Gp(mod + "_qkv_w"), 3 * H, H, B * S, transA=True, transB=False, lda=3 * H,
                         ldb=H, ldc=H, epi=hip.EPI_ATOMIC, splits=1)
                hip.colsum(dqkv_a, B * S, 3 * H, self.Gp(mod + "_qkv_b"))
                dh_all = self.buf("d_hall", (B * S, H))
                hip.layernorm_bwd(dhln, h_all, self.P(lnk + ".weight"), *a["st"][lnk], None, dh_all,
                                  self.G(lnk + ".weight"), self.G(lnk + ".bias"), B * S, H, ws=lnws_m)
                self._rnn_bwd(ch, layers, dh_all, B, dhp32, tmp32)
        # ---- topic channel
        dt_ln = self.buf("d_tln", (B, H))
        hip.cast_f32_to(dtopic, dt_ln, B * H)
        dt_raw = self.buf("d_traw", (B, H))
        hip.layernorm_bwd(dt_ln, a["t_raw"], self.P("ln_layer1.weight"), *a["st"]["ln1"], None, dt_raw,
                          self.G("ln_layer1.weight"), self.G("ln_layer1.bias"), B, H, ws=lnws)
        self._wgrad(a["xt"], dt_raw, "encoder.topic_fc.weight", "encoder.topic_fc.bias", B, "linear")
        self._join_beside()
        self._ready("encoder.topic_fc.bias")

    def _rnn_bwd(self, ch, layers, dh_all, B, dhp32, tmp32):
        """BPTT of one encoder channel (rows b*S+t), top layer first.  dh_all [B*S, H]: gradient of the top layer's outputs."""
        sh = self.sh
        S, H = sh.S, sh.H
        kind, NL = sh.rnn[ch]
        G = hip.RNN_GATES[kind]
        r = f"encoder.rnns_{ch}."
        for l in range(NL - 1, -1, -1):
            sfx = "_l%d" % l
            rec = layers[l]
            h_all, save = rec["h"], rec["save"]
            dgi = self.buf("d_gi%d" % G, (B * S, G * H))
            if kind == "GRU":
                # every step's d(gh) is kept (time-major) so that the recurrent weight / bias gradients are ONE
                # product and ONE column sum after the loop instead of one per step
                dgh_tm = self.buf("d_gh_tm", (S, B, 3 * H))
                ks = 6 if (3 * H) % (6 * 64) == 0 else 0      # split-K slabs of the carry product (fp32 kernel too, round 5: 92 -> us per step)
                part = self.buf("d_hp_slabs", (max(ks, 1), B, H), torch.float32)
                for t in range(S - 1, -1, -1):
                    # total gradient wrt h_t = rows b*S+t of dh_all + carry dh_{t+1} z_{t+1} + d(gh_{t+1}) W_hh, assembled in the cell kernel
                    last = t == S - 1
                    dgh = dgh_tm[t]
                    hip.gru_cell_bwd_fused(dh_all[t:], S * H, None if last else dhp32, None if last else (part if ks else tmp32),
                                           0 if last else (ks or 1), save[t], None if t == 0 else h_all[t - 1:], dgi[t:], dgh, dhp32,
                                           B, H, ld_hp=S * H, ld_dgi=S * 3 * H)
                    if t > 0:
                        if ks:
                            hip.gemm(dgh, self.W(r + "weight_hh" + sfx), part, B, H, 3 * H, transB=False, ldb=H,
                                     epi=hip.EPI_SPLIT, out_f32=True, splits=ks)
                        else:
                            hip.gemm(dgh, self.W(r + "weight_hh" + sfx), tmp32, B, H, 3 * H, transB=False, ldb=H, out_f32=True)
                hip.colsum(dgh_tm, S * B, 3 * H, self.G(r + "bias_hh" + sfx))
                drec_tm = dgh_tm[1:]
            else:
                # LSTM / ReLU cell: d(gi) = d(gh) = d(pre-activation), one tensor
                lstm = kind == "LSTM"
                code = hip.RNN_LSTM if lstm else hip.RNN_RELU
                dc = self.buf("d_c32", (B, H), torch.float32) if lstm else None
                for t in range(S - 1, -1, -1):
                    last = t == S - 1
                    hip.rnn_cell_bwd(code, dh_all[t:], S * H, None if last else tmp32, save[t] if lstm else None,
                                     rec["c"][t - 1] if lstm and t else None, h_all[t:], S * H, dc, not last, dgi[t:], S * G * H, B, H)
                    if t > 0:
                        hip.gemm(dgi[t:], self.W(r + "weight_hh" + sfx), tmp32, B, H, G * H, transB=False, lda=S * G * H, ldb=H,
                                 out_f32=True)
                hip.colsum(dgi, B * S, G * H, self.G(r + "bias_hh" + sfx))
                drec_tm = None
                if S > 1:
                    drec_tm = self.buf("d_a_tm%d" % G, (S - 1, B, G * H))
                    drec_tm.copy_(dgi.view(B, S, G * H)[:, 1:].transpose(0, 1))     # data movement only
            if S > 1:
                hprev_tm = self.buf("h_prev_tm", (S - 1, B, H))
                hprev_tm.copy_(h_all.view(B, S, H)[:, :S - 1].transpose(0, 1))     # data movement only
                hip.gemm(drec_tm, hprev_tm, self.G(r + "weight_hh" + sfx), G * H, H, (S - 1) * B, transA=True, transB=False,
                         lda=G * H, ldb=H, ldc=H, epi=hip.EPI_ATOMIC, splits=1)
            self._wgrad(rec["inp"], dgi, r + "weight_ih" + sfx, r + "bias_ih" + sfx, B * S, "linear")
            if l > 0:
                # gradient of the layer below's outputs: d(gi) W_ih, through the same dropout mask as the forward
                dh_all = self.buf("d_hx%d" % (l & 1), (B * S, H))
                hip.gemm(dgi, self.W(r + "weight_ih" + sfx), dh_all, B * S, H, G * H, transB=False, ldb=H)
                if layers[l - 1]["drop"] is not None:
                    hip.dropout_apply(dh_all, dh_all, B * S * H, sh.rnn_pdrop, layers[l - 1]["drop"])

    # ---------------------------------------------------------------- optimizer (train.py:194-197)
    def grad_norm_sq(self):
        hip.sumsq(self.grad, self.layout.total, self.normsq)        # (written, in a fixed order: no memset, no atomics)
        return self.normsq

    def adamw_step(self, lr, max_norm=1.0, betas=(0.9, 0.999), eps=1e-6, wd=0.0, grad_scale=1.0, clip=True, count=None):
        """clip_grad_norm_(1.0) + transformers.AdamW(lr, eps=1e-6, wd=0) over the flat buffers,
        refreshing the bf16 weight copies in the same pass.  count: device scalar, the global row count when the
        flat buffer holds row SUMS (mmtg_adamw divides by it; zero rows = no update)."""
        if self.opt_m is None:
            self.opt_m = torch.zeros_like(self.master)
            self.opt_v = torch.zeros_like(self.master)
        self.step_count += 1
        ns = self.grad_norm_sq() if clip else None
        hip.adamw(self.master, self.grad, self.opt_m, self.opt_v,
                  self.wc2[0] if self.x3 else None if self.dtype == hip.F32 else self.wc,
                  self.layout.total, lr, betas[0], betas[1], eps, wd, self.step_count, ns, max_norm, grad_scale, count=count,
                  p_lo=self.wc2[1] if self.x3 else None)
        self._refresh_transposed()
        self.copies_fresh = True
