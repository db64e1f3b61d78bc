"""Batched greedy generation with a KV cache and a replayed hipGraph per token.

The reference decodes one sample at a time and re-runs the whole model on the growing
prefix for every token (src/generate.py:117-145, O(L^2), ~198 full forwards per sample).
Here every row of a batch advances in lock step, one token per step, against per-layer
K/V caches; the step is a fixed launch sequence whose only varying input -- the position --
lives in device memory, so it is captured once into a hipGraph and replayed.

Semantics kept from the reference: forced [#EOS#]/[#START#] cadence every 22 slots,
per-occurrence repetition penalty, temperature, banned ids, sticky PAD, inference-branch
type ids / key mask.  The type-id / mask rule is applied per row (the reference reads row 0
only because it never batches; identical at batch 1).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import hip


_PLACEMENT = {}


def _placement_ok(dev):
    """True when a 256-workgroup, one-per-CU launch puts workgroups b and b + 8 k on one XCD, 32 per XCD (census kernel, once per
    device and process): what the plain hand-off of mmtg_decode_mlp relies on (the launch itself re-checks every group)."""
    key = str(dev)
    if key not in _PLACEMENT:
        c = hip.decode_mlp_census(dev)
        _PLACEMENT[key] = bool(((c == 32).sum(1) == 1).all() and ((c == 32).sum(0) == 1).all() and int(c.sum()) == 256)
    return _PLACEMENT[key]


class GreedyDecoder:
    def __init__(self, model, max_batch, max_len=None, use_graph=True, lanes=None, _parent=None, _lane=0):
        self.model = model
        self.eng = model.engine()
        sh = self.eng.sh
        self.B = max_batch
        self.max_len = sh.max_seq_length if max_len is None else max_len      # lyric positions incl. the first [#START#]
        self.Tmax = sh.P + self.max_len + 1
        if self.Tmax > sh.NP:
            raise ValueError("prompt + max_len exceeds n_positions")
        self.use_graph = use_graph
        dev, tdt = self.eng.dev, self.eng.tdt
        B, D, H, E = self.B, sh.D, sh.H, sh.E
        # bf16: deterministic split-K products (K slices per product: c_attn, attn c_proj, c_fc, mlp c_proj)
        # fp32 (the parity mode): the same split-K schedule through the fp32 kernel's slab epilogue (128x128 tiles, so more
        # K slices: 6-24 workgroups per product otherwise)
        self.fast = not os.environ.get("MMTG_DECODE_PLAIN")
        bf = self.eng.dtype == hip.BF16
        # Lanes (opt-in, MMTG_DECODE_LANES / lanes=): the rows of a batch never interact, so the batch can be cut into
        # `lanes` row blocks whose launch chains run side by side on their own streams (fork / join inside the captured
        # graph).  Every row goes through the same arithmetic in the same order whatever the lane count, so the ids do
        # not depend on it (tests/test_decode_gpu.py).  Measured at batch 256 (profiles/r02_v5_decode_lanes_ab.txt):
        # 1 lane 758 us per token step, 2 lanes 825, 4 lanes 1452, 8 lanes 2785 -- hipGraph replay on this ROCm runs
        # the branches one after another and adds a cross-stream dependency per fork / join, so the default stays 1.
        if _parent is not None:
            lanes = 1
        elif lanes is None:
            lanes = int(os.environ.get("MMTG_DECODE_LANES", "1")) if (self.fast and B >= 128) else 1
        if lanes < 1 or B % lanes:
            raise ValueError("batch %d does not divide into %d lanes" % (B, lanes))
        self.lanes = lanes
        if _parent is None:
            self.seq = torch.zeros(B, self.Tmax, dtype=torch.long, device=dev)
            # the position lives in a PAIR of slots per lane: the kernels of step k read slot k % 2, the step's last kernel
            # (decode_select) writes k + 1 into the other one -- no separate "advance" launch, two captured graphs per setting
            self.pos_all = torch.zeros(2, lanes, dtype=torch.int32, device=dev)
            self.keep = torch.zeros(B, self.Tmax, dtype=torch.int32, device=dev)
            self.types = torch.zeros(B, dtype=torch.long, device=dev)
            self.tpw_type = torch.zeros(B, sh.P, dtype=torch.long, device=dev)
            self.tpw_mask = torch.zeros(B, sh.P, dtype=torch.long, device=dev)
            self.c = torch.zeros(B * sh.S, E, dtype=tdt, device=dev)
            self.kc = torch.zeros(sh.L, B, sh.nH, self.Tmax, 64, dtype=tdt, device=dev)
            self.vc = torch.zeros_like(self.kc)
            self.pos_pair = (self.pos_all[0, 0:1], self.pos_all[1, 0:1])
        else:       # a lane: row block [_lane*B, (_lane+1)*B) of the parent's state, private scratch
            lo, hi = _lane * B, (_lane + 1) * B
            self.seq, self.keep, self.types = _parent.seq[lo:hi], _parent.keep[lo:hi], _parent.types[lo:hi]
            self.tpw_type, self.tpw_mask = _parent.tpw_type[lo:hi], _parent.tpw_mask[lo:hi]
            self.c = _parent.c[lo * sh.S:hi * sh.S]
            self.kc = [_parent.kc[l, lo:hi] for l in range(sh.L)]
            self.vc = [_parent.vc[l, lo:hi] for l in range(sh.L)]
            self.pos_pair = (_parent.pos_all[0, _lane:_lane + 1], _parent.pos_all[1, _lane:_lane + 1])
        # (K slices of c_attn, attn c_proj, c_fc, mlp c_proj.  Fused step -- no finish launches, the reduction runs in the product's
        #  own tail -- measured at batch 256, us per token step: 2,4,1,4 -> 714-716; 2,3,1,4 -> 715; 2,2,1,4 -> 721; 3,4,1,4 -> 720;
        #  2,4,1,3 -> 726; 2,4,1,8 -> 728; 2,4,1,6 -> 739; 2,6,1,4 -> 746; 2,4,1,2 -> 751; 4,4,1,4 -> 755; profiles/r03_v8_*)
        fused_ok = self.eng.dtype == hip.BF16 and not os.environ.get("MMTG_DECODE_PLAIN") and os.environ.get("MMTG_DECODE_FUSED", "1") != "0"
        # Round 5, the split-precision step (compute_dtype="bf16x3"): the fused step's launch structure on (hi | lo) plane pairs --
        # fp32 residual stream / KV cache / logits, every product three bf16 matrix-core passes (mmtg_decode_gemm_x3).
        # MMTG_DECODE_X3=0 keeps the exact-fp32 kernels (the f32 mode's step) for the A/B.
        self.x3 = (bool(getattr(self.eng, "x3", False)) and self.fast and os.environ.get("MMTG_DECODE_X3", "1") != "0"
                   and D % 64 == 0 and D // 32 <= hip.DG_NP and H % 64 == 0 and E % 64 == 0 and self.eng.layout.Vpad % 4 == 0)
        self.splits = tuple(int(x) for x in os.environ.get(
            "MMTG_DECODE_SPLITS", "2,4,1,8" if self.x3 else ("2,4,1,4" if fused_ok else "2,4,1,8") if bf else "8,12,8,24").split(","))
        # (measured at batch 256, us per token step: 2,3,2,6 -> 1020; 4,6,3,12 -> 1248; 1,1,1,2 -> 1183; unsplit 1264;
        #  with the one-slice c_fc + fused GELU: 2,3,1,8 -> 940, 2,3,1,6 -> 943, 1,3,1,8 -> 963, 2,3,1,12 -> 1003)
        #  round 2, 64x64 tiles (graph-replayed per-product times, profiles/r02_decode_gemm_tiles.log): 2,4,1,8)
        self.children, self.side = [], []
        if lanes > 1:
            self.children = [GreedyDecoder(model, B // lanes, max_len, use_graph=False, _parent=self, _lane=i) for i in range(lanes)]
            self.side = [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
        else:
            z = lambda *s: torch.empty(*s, dtype=tdt, device=dev)
            self.x, self.h1 = z(B, E), z(B, H)
            self.h, self.h2 = z(B, D), z(B, D)
            self.a, self.qkv, self.ctx = z(B, D), z(B, 3 * D), z(B, D)
            self.u, self.g = z(B, 4 * D), z(B, 4 * D)
            self.mu = torch.empty(B, dtype=torch.float32, device=dev)
            self.rs = torch.empty(B, dtype=torch.float32, device=dev)
            self.logits = torch.empty(B, self.eng.layout.Vpad, dtype=torch.float32, device=dev)
            if self.fast:
                slab = max(self.splits[0] * 3 * D, self.splits[1] * D, self.splits[2] * 4 * D, self.splits[3] * D)
                self.part = torch.empty(slab * B, dtype=torch.float32, device=dev)
            if self.x3:
                L, Vp = sh.L, self.eng.layout.Vpad
                f32 = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
                PL = lambda r, c_: hip.Planes.empty(r, c_, dev)
                self.xp, self.h1p = PL(B, E), PL(B, H)
                self.hp, self.h2p = PL(B, D), PL(B, D)
                self.ctxp, self.gp = PL(B, D), PL(B, 4 * D)
                self.fq = [(PL(3 * D, D), f32(3 * D), f32(3 * D)) for _ in range(L)]          # gamma-folded c_attn plane pair, column sums, folded bias
                self.ffc = [(PL(4 * D, D), f32(4 * D), f32(4 * D)) for _ in range(L)]
                self.fh = (PL(Vp, D), f32(Vp), f32(Vp))
                self.st = (torch.zeros(B, hip.DG_NP, 2, dtype=torch.float32, device=dev),
                           torch.zeros(B, hip.DG_NP, 2, dtype=torch.float32, device=dev))
                # K slices of the two projector products (K = 2048 / 512): in-kernel reduction like attn.c_proj
                self.psplits = tuple(int(x) for x in os.environ.get("MMTG_DECODE_X3_PSPLITS", "8,2").split(","))
                rt = -(-B // 64)
                tiles = rt * max(D // 64, H // 64)
                self.rws = f32(tiles * max(self.splits[1], self.splits[3], *self.psplits) * 4096)
                self.rcnt = torch.zeros(tiles * 4, dtype=torch.int32, device=dev)
            # Round 3, fused step (5 graph nodes per block instead of 7): split-K products reduced in the kernel by the last-arriving
            # wave (+ bias + residual + LayerNorm statistics), LayerNorms applied algebraically in the consuming products
            # (mmtg_decode_gemm).  MMTG_DECODE_FUSED=0 keeps the round-2 products + finish launches.
            self.fused = (self.fast and bf and os.environ.get("MMTG_DECODE_FUSED", "1") != "0" and D % 64 == 0 and D // 32 <= hip.DG_NP
                          and self.eng.layout.Vpad % 4 == 0)
            if self.fused:
                L, Vp = sh.L, self.eng.layout.Vpad
                bf = lambda *s_: torch.empty(*s_, dtype=torch.bfloat16, device=dev)
                f32 = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
                self.fq = [(bf(3 * D, D), f32(3 * D), f32(3 * D)) for _ in range(L)]          # gamma-folded c_attn copy, column sums, folded bias
                self.ffc = [(bf(4 * D, D), f32(4 * D), f32(4 * D)) for _ in range(L)]
                self.fh = (bf(Vp, D), f32(Vp), f32(Vp))
                self.st = (torch.zeros(B, hip.DG_NP, 2, dtype=torch.float32, device=dev),
                           torch.zeros(B, hip.DG_NP, 2, dtype=torch.float32, device=dev))
                tiles = -(-B // 64) * (D // 64)
                self.rws = f32(tiles * max(self.splits[1], self.splits[3]) * 4096)
                self.rcnt = torch.zeros(tiles * 4, dtype=torch.int32, device=dev)
                self.folds_for = None
                # (MMTG_DECODE_EMBED_IN_PROJ=0: projector_layer2 and the embedding add as two launches, the round-3 v8 step)
                self.embed_in_proj = os.environ.get("MMTG_DECODE_EMBED_IN_PROJ", "1") != "0" and H % 8 == 0
                # Round 6: c_fc -> GELU -> mlp.c_proj of a block as ONE launch (mmtg_decode_mlp): the hidden dimension is split over the
                # XCDs, so the GEMM -> GEMM hand-off stays on one L2 and every weight crosses the fabric once; 4 graph nodes per
                # block instead of 5.  Built for n_embd = 768 and up to 256 rows, parity-tested, and OPT-IN (MMTG_DECODE_MLP=1):
                # measured at batch 256 it ties the two launches stand-alone (20.2 against 20.8 us; 17.6 against 19.2 at 128 rows)
                # and loses inside the token step (713-720 against 682-695 us): the boundary it removes costs 1.1-1.7 us
                # (profiles/r06_v2_graph_node_floor_by_launch_shape.txt) + a ~2 us first-tile ramp, the hand-off that replaces it
                # 0.96 us of store acknowledgement + 2.75 us of wait (arrival skew of the 8 producers) + 0.69 us of reload
                # (profiles/r06_v2_decode_mlp_in_step_timeline.txt), and the 8-way cross-XCD reduction has a longer tail than the
                # 4-way one.  MMTG_DECODE_MLP_HANDOFF=plain: through the L2 (checked per launch; slower in the step: its partial-line
                # stores fetch lines); default sc1 (write-through + agent-scope loads: valid under any workgroup placement).
                # Never for the row blocks of a multi-lane decoder: its 256 workgroups must be co-resident, which side-by-side
                # launches on several streams cannot promise.
                self.mlp = _parent is None and D == 768 and B <= 256 and os.environ.get("MMTG_DECODE_MLP") == "1"
                if self.mlp:
                    self.mlp_ws = f32(hip.decode_mlp_ws_floats(B))
                    self.mlp_sync = torch.zeros(hip.decode_mlp_sync_words(), dtype=torch.int64, device=dev)
                    self.mlp_plain = os.environ.get("MMTG_DECODE_MLP_HANDOFF", "sc1") == "plain"
                    self.mlp_trace = None
                    if self.mlp_plain and not _placement_ok(dev):
                        self.mlp_plain = False          # workgroups b and b + 8 k do not share an XCD here: write-through hand-off
        self.pos, self.pos_next = self.pos_pair
        # the prompt in one batched pass (round 5; MMTG_DECODE_PREFILL=0: P token steps, as rounds 1-4 ran it)
        self.prefill = _parent is None and os.environ.get("MMTG_DECODE_PREFILL", "1") != "0"
        self.first_pos = 0
        self.uniforms = None
        self.graphs = {}
        self.params = None

    def _conv1d(self, x, wkey, out, **kw):
        # bf16: Engine._fwd routes Conv1D weights through their [out,in] copies (NT layout), which is
        # what the 256x32 small-M GEMM configuration serves; the engine keeps the copies fresh.
        self.eng._fwd(x, wkey, out, self.B, "conv1d", **kw)

    # ------------------------------------------------------------------ one token
    def _step(self, with_head, parity=0):
        self.pos, self.pos_next = self.pos_pair[parity], self.pos_pair[1 - parity]
        if self.children:
            # fork: lane 0 stays on the current stream, the others run on side streams; join before the step ends
            cur = torch.cuda.current_stream()
            for st in self.side:
                st.wait_stream(cur)
            for i, ch in enumerate(self.children):
                ch.params = self.params
                if i == 0:
                    ch._step(with_head, parity)
                else:
                    with torch.cuda.stream(self.side[i - 1]):
                        ch._step(with_head, parity)
            for st in self.side:
                cur.wait_stream(st)
            return
        eng, sh, B = self.eng, self.eng.sh, self.B
        D, H, E = sh.D, sh.H, sh.E
        pre = "decoder.gpt2.transformer."
        sent = sh.msl + 2
        if getattr(self, "x3", False):
            self._step_x3(with_head)
            self._select(with_head)
            return
        hip.decode_embed(eng.table, self.seq, self.c, self.x, self.pos, self.tpw_type, self.tpw_mask, self.types,
                         self.keep, B, sh.P, sh.S, E, sh.two_sents, eng.table.shape[0], sent,
                         sh.max_seq_length // sent + 1)
        eng._fwd(self.x, "decoder.projector_layer1.weight", self.h1, B, "linear",
                 bias=eng.P("decoder.projector_layer1.bias"), epi=hip.EPI_TANH)
        fused = getattr(self, "fused", False)
        if fused and self.embed_in_proj:
            # projector_layer2 + the GPT-2 input embedding (wpe[pos] + wte[type]) + the first LayerNorm statistics in one launch
            w2 = eng.W("decoder.projector_layer2.weight")
            hip.decode_gemm(2, self.h1, w2, self.h, B, D, H, bias=eng.P("decoder.projector_layer2.bias"), stats_out=self.st[0],
                            emb_pos=eng.W(pre + "wpe.weight"), emb_type=eng.W(pre + "wte.weight"), type_ids=self.types, pos=self.pos, ldr=D)
        else:
            eng._fwd(self.h1, "decoder.projector_layer2.weight", self.h, B, "linear", bias=eng.P("decoder.projector_layer2.bias"))
            hip.decode_embed_add(self.h, eng.W(pre + "wpe.weight"), eng.W(pre + "wte.weight"), self.types, self.pos, self.h, B, D,
                                 stats=self.st[0] if fused else None)
        hcur, hnext = self.h, self.h2
        if fused:
            self._layers_fused(hcur, hnext, with_head)
        elif self.fast:
            self._layers_split(hcur, hnext, with_head)
        else:
            self._layers_plain(hcur, hnext, with_head)
        if with_head and not fused:
            hip.gemm(self.a, eng.Wp("wte"), self.logits, B, eng.layout.Vpad, D, transB=True, ldb=D, out_f32=True)
        self._select(with_head)

    def _select(self, with_head):
        """The step's last launch: logits processing + arg-max / draw + the forced cadence + the next position (generate.py:117-142)."""
        eng, sh, B = self.eng, self.eng.sh, self.B
        temperature, rep, top_k, top_p = self.params
        sent = sh.msl + 2
        if with_head:
            Vp = eng.layout.Vpad
            if top_k == 1 and top_p == 0.0:
                hip.decode_select(self.logits, Vp, min(sh.V, 13317), self.seq, self.pos, sh.P, sent, temperature, rep, B,
                                  pos_next=self.pos_next)
            else:        # stochastic: the draw of each position reads its row of the pre-filled uniforms
                hip.decode_sample(self.logits, Vp, min(sh.V, 13317), self.seq, self.pos, sh.P, sent, temperature, rep,
                                  top_k, top_p, self.uniforms, B, pos_next=self.pos_next)
        else:
            hip.decode_select(None, 0, 0, self.seq, self.pos, sh.P, sent, temperature, rep, B, pos_next=self.pos_next)

    def _step_x3(self, with_head):
        """The split-precision token step: embedding -> projector -> 12 blocks -> head, every product through mmtg_decode_gemm_x3
        on plane pairs.  Per block: c_attn (LN-fold, fp32 slabs summed by the attention kernel) -> attention over the fp32 KV cache
        (context as a plane pair) -> attn.c_proj (+ bias + fp32 residual, statistics; fp32 stream AND its plane pair) -> c_fc
        (LN-fold + GELU, plane pair) -> mlp.c_proj (as attn.c_proj); the head is an LN-fold product writing fp32 logits."""
        eng, sh, B = self.eng, self.eng.sh, self.B
        D, H, E = sh.D, sh.H, sh.E
        pre = "decoder.gpt2.transformer."
        sent = sh.msl + 2
        sq, sp, _, s2 = self.splits
        p1, p2 = self.psplits
        NP = D // 32
        hip.decode_embed_x3(eng.table, self.seq, self.c, self.xp, self.pos, self.tpw_type, self.tpw_mask, self.types,
                            self.keep, B, sh.P, sh.S, E, sh.two_sents, eng.table.shape[0], sent, sh.max_seq_length // sent + 1)
        hip.decode_gemm_x3(2, self.xp, eng.Wx("decoder.projector_layer1.weight"), B, H, E, Cp=self.h1p,
                           bias=eng.P("decoder.projector_layer1.bias"), act=hip.EPI_TANH, splits=p1, ws=self.rws, counters=self.rcnt)
        x, xo, xp, xop = self.h, self.h2, self.hp, self.h2p
        st, sto = self.st
        hip.decode_gemm_x3(2, self.h1p, eng.Wx("decoder.projector_layer2.weight"), B, D, H, C_=x, Cp=xp,
                           bias=eng.P("decoder.projector_layer2.bias"), stats_out=st, splits=p2, ws=self.rws, counters=self.rcnt,
                           emb_pos=eng.P(pre + "wpe.weight"), emb_type=eng.P(pre + "wte.weight"), type_ids=self.types, pos=self.pos, ldr=D)
        kper = -(-(-(-D // sq)) // 64) * 64
        nslab = -(-D // kper)
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            wf, c, bq = self.fq[l]
            hip.decode_gemm_x3(1, xp, wf, B, 3 * D, D, C_=self.part, colsum=c, stats_in=st, np_in=NP, eps=sh.eps, splits=sq)
            hip.decode_attn_split_x3(self.part, nslab, bq, self.kc[l], self.vc[l], self.keep, self.pos, self.ctxp, B, sh.nH, 64, self.Tmax)
            hip.decode_gemm_x3(2, self.ctxp, eng.Wtx(p + "attn.c_proj.weight"), B, D, D, C_=xo, Cp=xop, bias=eng.P(p + "attn.c_proj.bias"),
                               resid=x, stats_out=sto, splits=sp, ws=self.rws, counters=self.rcnt)
            wf, c, bfc = self.ffc[l]
            hip.decode_gemm_x3(0, xop, wf, B, 4 * D, D, Cp=self.gp, bias=bfc, colsum=c, stats_in=sto, np_in=NP, eps=sh.eps, act=hip.EPI_GELU)
            hip.decode_gemm_x3(2, self.gp, eng.Wtx(p + "mlp.c_proj.weight"), B, D, 4 * D, C_=x, Cp=xp, bias=eng.P(p + "mlp.c_proj.bias"),
                               resid=xo, stats_out=st, splits=s2, ws=self.rws, counters=self.rcnt)
        if with_head:
            wf, c, bh = self.fh
            hip.decode_gemm_x3(0, xp, wf, B, eng.layout.Vpad, D, C_=self.logits, bias=bh, colsum=c, stats_in=st, np_in=NP, eps=sh.eps)

    def _layers_plain(self, hcur, hnext, with_head):
        """One GPT-2 block per layer with the training-side kernels (fp32 parity mode)."""
        eng, sh, B = self.eng, self.eng.sh, self.B
        D = sh.D
        pre = "decoder.gpt2.transformer."
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            hip.layernorm_fwd(hcur, self.a, eng.P(p + "ln_1.weight"), eng.P(p + "ln_1.bias"), self.mu, self.rs, B, D, sh.eps)
            self._conv1d(self.a, p + "attn.c_attn.weight", self.qkv, bias=eng.P(p + "attn.c_attn.bias"))
            hip.decode_attn(self.qkv, self.kc[l], self.vc[l], self.keep, self.pos, self.ctx, B, sh.nH, 64, self.Tmax)
            self._conv1d(self.ctx, p + "attn.c_proj.weight", hnext, bias=eng.P(p + "attn.c_proj.bias"),
                         epi=hip.EPI_RESID, aux=hcur, ldaux=D)
            hip.layernorm_fwd(hnext, self.a, eng.P(p + "ln_2.weight"), eng.P(p + "ln_2.bias"), self.mu, self.rs, B, D, sh.eps)
            self._conv1d(self.a, p + "mlp.c_fc.weight", self.g, bias=eng.P(p + "mlp.c_fc.bias"),
                         epi=hip.EPI_GELU, aux2=self.u)
            self._conv1d(self.g, p + "mlp.c_proj.weight", hcur, bias=eng.P(p + "mlp.c_proj.bias"),
                         epi=hip.EPI_RESID, aux=hnext, ldaux=D)
        if with_head:
            hip.layernorm_fwd(hcur, self.a, eng.P(pre + "ln_f.weight"), eng.P(pre + "ln_f.bias"), self.mu, self.rs, B, D, sh.eps)

    def _split(self, x, wkey, out, splits, bias, **fin):
        """Deterministic split-K product for the batch-sized M of a decode step: `splits` K slices store
        fp32 partial products (N/32 x splits workgroups instead of N/32), splitk_finish sums them in order
        and applies bias / activation / residual (/ the next LayerNorm)."""
        self._slabs(x, wkey, splits)
        hip.splitk_finish(self.part, splits, self.B, self.eng.W(wkey).shape[1], out, bias=bias, **fin)

    def _slabs(self, x, wkey, splits):
        """`splits` fp32 partial products of x @ W into self.part (bf16: the [out,in] weight copy through the 64x64 LDS-DMA
        configuration; fp32: the Conv1D weight as stored through the fp32 kernel)."""
        if self.eng.dtype == hip.BF16:
            wt = self.eng.Wt(wkey)
            N, K = wt.shape
            hip.gemm(x, wt, self.part, self.B, N, K, transB=True, ldb=K, ldc=N, epi=hip.EPI_SPLIT, out_f32=True, splits=splits)
        else:
            w = self.eng.W(wkey)
            K, N = w.shape
            hip.gemm(x, w, self.part, self.B, N, K, ldb=N, ldc=N, epi=hip.EPI_SPLIT, out_f32=True, splits=splits)

    def _layers_split(self, hcur, hnext, with_head):
        """bf16 fast path: 4 split-K products + 4 finish kernels + attention per layer; every LayerNorm
        rides on the finish kernel of the product before it (self.a always holds the next LN output)."""
        eng, sh, B = self.eng, self.eng.sh, self.B
        D = sh.D
        pre = "decoder.gpt2.transformer."
        sq, sp, s1, s2 = self.splits
        hip.layernorm_fwd(hcur, self.a, eng.P(pre + "h.0.ln_1.weight"), eng.P(pre + "h.0.ln_1.bias"), self.mu, self.rs, B, D, sh.eps)
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            if os.environ.get("MMTG_DECODE_ATTN_SPLIT", "1") == "1":
                # c_attn: the attention kernel sums the split-K slabs itself (no finish launch)
                self._slabs(self.a, p + "attn.c_attn.weight", sq)
                hip.decode_attn_split(self.part, sq, eng.P(p + "attn.c_attn.bias"), self.kc[l], self.vc[l], self.keep, self.pos,
                                      self.ctx, B, sh.nH, 64, self.Tmax)
            else:
                self._split(self.a, p + "attn.c_attn.weight", self.qkv, sq, eng.P(p + "attn.c_attn.bias"))
                hip.decode_attn(self.qkv, self.kc[l], self.vc[l], self.keep, self.pos, self.ctx, B, sh.nH, 64, self.Tmax)
            self._split(self.ctx, p + "attn.c_proj.weight", hnext, sp, eng.P(p + "attn.c_proj.bias"),
                        epi=hip.EPI_RESID, aux=hcur, ldaux=D,
                        ln_gamma=eng.P(p + "ln_2.weight"), ln_beta=eng.P(p + "ln_2.bias"), ln_out=self.a, eps=sh.eps)
            if s1 == 1:     # one K slice: bias + GELU in the product's own epilogue, no finish launch
                self._conv1d(self.a, p + "mlp.c_fc.weight", self.g, bias=eng.P(p + "mlp.c_fc.bias"), epi=hip.EPI_GELU, aux2=self.u)
            else:
                self._split(self.a, p + "mlp.c_fc.weight", self.g, s1, eng.P(p + "mlp.c_fc.bias"), epi=hip.EPI_GELU)
            if l + 1 < sh.L:
                nxt = f"{pre}h.{l + 1}.ln_1."
            else:
                nxt = pre + "ln_f." if with_head else None
            ln = {} if nxt is None else dict(ln_gamma=eng.P(nxt + "weight"), ln_beta=eng.P(nxt + "bias"), ln_out=self.a, eps=sh.eps)
            self._split(self.g, p + "mlp.c_proj.weight", hcur, s2, eng.P(p + "mlp.c_proj.bias"),
                        epi=hip.EPI_RESID, aux=hnext, ldaux=D, **ln)

    def _refresh_folds(self):
        """gamma-folded weight copies, their column sums and folded biases for the LN-fold products (once per generation: the
        weights may have been stepped)."""
        eng, sh = self.eng, self.eng.sh
        D = sh.D
        pre = "decoder.gpt2.transformer."
        if getattr(self, "x3", False):
            for l in range(sh.L):
                p = f"{pre}h.{l}."
                wf, c, b = self.fq[l]
                hip.ln_fold_weights_x3(eng.Wtx(p + "attn.c_attn.weight"), eng.P(p + "ln_1.weight"), eng.P(p + "ln_1.bias"),
                                       eng.P(p + "attn.c_attn.bias"), wf, c, b, 3 * D, D)
                wf, c, b = self.ffc[l]
                hip.ln_fold_weights_x3(eng.Wtx(p + "mlp.c_fc.weight"), eng.P(p + "ln_2.weight"), eng.P(p + "ln_2.bias"),
                                       eng.P(p + "mlp.c_fc.bias"), wf, c, b, 4 * D, D)
            wf, c, b = self.fh
            hip.ln_fold_weights_x3(eng.Wpx("wte", eng.layout.Vpad, D), eng.P(pre + "ln_f.weight"), eng.P(pre + "ln_f.bias"), None, wf, c, b,
                                   eng.layout.Vpad, D)
            return
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            wf, c, b = self.fq[l]
            hip.ln_fold_weights(eng.Wt(p + "attn.c_attn.weight"), eng.P(p + "ln_1.weight"), eng.P(p + "ln_1.bias"),
                                eng.P(p + "attn.c_attn.bias"), wf, c, b, 3 * D, D)
            wf, c, b = self.ffc[l]
            hip.ln_fold_weights(eng.Wt(p + "mlp.c_fc.weight"), eng.P(p + "ln_2.weight"), eng.P(p + "ln_2.bias"),
                                eng.P(p + "mlp.c_fc.bias"), wf, c, b, 4 * D, D)
        wf, c, b = self.fh
        hip.ln_fold_weights(eng.Wp("wte"), eng.P(pre + "ln_f.weight"), eng.P(pre + "ln_f.bias"), None, wf, c, b, eng.layout.Vpad, D)

    def _layers_fused(self, hcur, hnext, with_head):
        """bf16 fused path: per block c_attn (LN-fold, split-K slabs summed by the attention kernel) -> attention -> attn c_proj
        (split-K reduced in the kernel + bias + residual + statistics) -> c_fc (LN-fold + GELU) -> mlp c_proj (as c_proj); the
        head is an LN-fold product writing fp32 logits.  The residual stream alternates between two buffers and so do its
        LayerNorm statistics.  (Round 4's persistent one-launch step and chained launches -- bit-equal, measured slower -- are kept as
        tools/experiments/decode_persistent_and_chained.patch.)"""
        eng, sh, B = self.eng, self.eng.sh, self.B
        D = sh.D
        pre = "decoder.gpt2.transformer."
        sq, sp, _, s2 = self.splits
        NP = D // 32
        kper = -(-(-(-D // sq)) // 64) * 64          # K slices are whole 64-deep tiles: the product writes ceil(D / kper) slabs
        nslab = -(-D // kper)
        x, xo = hcur, hnext
        st, sto = self.st
        mlp = getattr(self, "mlp", False)
        npx = NP            # statistics partials of the block's input rows (the projector's / a mode-2 product's: one per 32 columns)
        for l in range(sh.L):
            p = f"{pre}h.{l}."
            wf, c, bq = self.fq[l]
            hip.decode_gemm(1, x, wf, self.part, B, 3 * D, D, colsum=c, stats_in=st, np_in=npx, eps=sh.eps, out_f32=True, splits=sq)
            hip.decode_attn_split(self.part, nslab, bq, self.kc[l], self.vc[l], self.keep, self.pos,
                                  self.ctx, B, sh.nH, 64, self.Tmax)
            hip.decode_gemm(2, self.ctx, eng.Wt(p + "attn.c_proj.weight"), xo, B, D, D, bias=eng.P(p + "attn.c_proj.bias"), resid=x,
                            stats_out=sto, splits=sp, ws=self.rws, counters=self.rcnt)
            wf, c, bfc = self.ffc[l]
            if mlp:
                # c_fc (LN-fold + GELU) -> mlp.c_proj (+ bias + residual + statistics, 16 partials of 48 columns) in one launch
                hip.decode_mlp(xo, sto, NP, sh.eps, wf, c, bfc, eng.Wt(p + "mlp.c_proj.weight"), eng.P(p + "mlp.c_proj.bias"), self.g, x, st,
                               self.mlp_ws, self.mlp_sync, B, D, plain=self.mlp_plain, trace=self.mlp_trace)
                npx = 16
                continue
            hip.decode_gemm(0, xo, wf, self.g, B, 4 * D, D, bias=bfc, colsum=c, stats_in=sto, np_in=NP, eps=sh.eps, act=hip.EPI_GELU)
            hip.decode_gemm(2, self.g, eng.Wt(p + "mlp.c_proj.weight"), x, B, D, 4 * D, bias=eng.P(p + "mlp.c_proj.bias"), resid=xo,
                            stats_out=st, splits=s2, ws=self.rws, counters=self.rcnt)
        if with_head:
            wf, c, bh = self.fh
            hip.decode_gemm(0, x, wf, self.logits, B, eng.layout.Vpad, D, bias=bh, colsum=c, stats_in=st, np_in=npx, eps=sh.eps,
                            out_f32=True)

    def _run_step(self, with_head, parity):
        if not self.use_graph:
            self._step(with_head, parity)
            return
        key = (with_head, parity, self.params)
        g = self.graphs.get(key)
        if g is None:
            # warm-up outside capture (lazy LDS-attribute / module loading), then rewind the position
            saved = (self.pos_all.clone(), self.seq.clone(), self.keep.clone())
            self._step(with_head, parity)
            torch.cuda.synchronize()
            self.pos_all.copy_(saved[0]); self.seq.copy_(saved[1]); self.keep.copy_(saved[2])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step(with_head, parity)
            self.graphs[key] = g
            self.pos_all.copy_(saved[0]); self.seq.copy_(saved[1]); self.keep.copy_(saved[2])
        g.replay()

    # ------------------------------------------------------------------ public
    def describe(self):
        how = "one hipGraph replay per token step" if self.use_graph else "eager launches per token step"
        return how + (", %d row blocks of %d side by side" % (self.lanes, self.B // self.lanes) if self.lanes > 1 else "")

    def kernel_name(self):
        if getattr(self, "x3", False):
            return ("decode token step (bf16x3): decode_gemm_x3_kernel<64x64> weight streaming over (hi | lo) plane pairs, three bf16 matrix-core "
                    "passes per product (split-K reduced in the kernel, LayerNorm applied algebraically, fp32 residual stream) + decode_attn "
                    "streaming an fp32 KV cache, 5 graph nodes per block")
        if getattr(self, "fused", False) and getattr(self, "mlp", False):
            return ("decode token step: decode_gemm_kernel<64x64> weight streaming (c_attn LN-fold slabs, attn.c_proj split-K reduced in the "
                    "kernel) + decode_attn KV-cache streaming + decode_mlp_kernel (c_fc -> GELU -> mlp.c_proj in one launch, hidden dimension "
                    "split over the XCDs, %s hand-off), 4 graph nodes per block" % ("L2" if self.mlp_plain else "write-through"))
        if getattr(self, "fused", False):
            return ("decode token step: decode_gemm_kernel<64x64> weight streaming (split-K reduced in the kernel, LayerNorm applied "
                    "algebraically) + decode_attn KV-cache streaming, 5 graph nodes per block")
        return ("decode token step: split-K gemm_dma_kernel<64x64> weight streaming + decode_attn KV-cache streaming + finish kernels"
                if self.fast else "decode token step: training-side kernels per layer")

    @torch.no_grad()
    def begin(self, batch, length, temperature=1.0, repitition_penalty=1.0, top_k=1, top_p=0.0, generator=None):
        """Everything of a generation that happens once: the settings, the uniforms of a stochastic run, fresh weight copies /
        LayerNorm folds, the experience encoder, the prompt (prefilled in one batched pass: ``first_pos`` = P afterwards) and the
        reset position.  Returns the end position n; ``step_at(pos)`` for pos = first_pos .. n - 1 then runs the token steps (``generate`` does both; tools/decode_lanes_threads.py drives
        several decoders' steps from their own threads and streams)."""
        eng, sh = self.eng, self.eng.sh
        B = batch["img_embs"].shape[0]
        if B != self.B:
            raise ValueError("decoder was built for batch %d, got %d" % (self.B, B))
        if length > self.max_len:
            raise ValueError("length %d exceeds max_len %d" % (length, self.max_len))
        self.params = (float(temperature), float(repitition_penalty), int(top_k), float(top_p))
        if not (int(top_k) == 1 and float(top_p) == 0.0):
            if self.uniforms is None:
                self.uniforms = torch.empty(sh.P + self.max_len + 1, self.B, device=eng.dev, dtype=torch.float32)
            self.uniforms.uniform_(0.0, 1.0, generator=generator)
            self.uniforms.clamp_(max=1.0 - 2.0 ** -24)
            for i, ch in enumerate(self.children):      # a lane reads its own [position, row] block of the same draws
                blk = self.uniforms[:, i * ch.B:(i + 1) * ch.B]
                if ch.uniforms is None:
                    ch.uniforms = torch.empty_like(blk, memory_format=torch.contiguous_format)
                ch.uniforms.copy_(blk)
        if getattr(self, "mlp", False):
            self.mlp_sync.zero_()           # counters armed, error report cleared (check_mlp_error reads it after the generation)
        eng.invalidate_copies()
        self.eng.refresh_copies()
        for d in ([self] + self.children):          # (a lane has its own folded copies: they are part of its scratch)
            if getattr(d, "fused", False) or getattr(d, "x3", False):
                d._refresh_folds()
        self.seq.zero_()
        self.seq[:, :sh.P] = batch["topic_ids"].to(eng.dev).long()
        self.seq[:, sh.P] = 1                                   # [#START#]
        self.tpw_type.copy_(batch["tpw_type_ids"].to(eng.dev).long())
        self.tpw_mask.copy_(batch["tpw_attention_mask"].to(eng.dev).long())
        self.keep.zero_()
        self.pos_all.zero_()
        self.first_pos = 0
        if self.prefill and sh.P > 0:
            self._prefill(batch)
        else:
            a = eng.forward(batch, train_flag=False, training=False, encode_only=True)
            self.c.copy_(a["c"])
        return sh.P + length                                    # positions first_pos .. P+length-1 are consumed by token steps

    def _prefill(self, batch):
        """The P prompt positions in ONE batched pass instead of P token steps (a token step costs the same whatever it appends; the
        reference has no such distinction -- its every call re-runs the whole prefix, generate.py:124).  The engine's inference-branch
        forward (model.py:290-326) over [prompt, [#START#]] computes every block's K / V rows for all prompt positions at once
        (B x (P + 1) rows through the training-side kernels of the compute mode, nothing of the last block beyond its c_attn; causal,
        so the rows of the P prompt positions do not depend on the extra one -- the conditioning kernel wants at least one lyric
        position); they go into the caches, the key mask of the prompt is the prompt's attention mask, and the token
        steps start at position P."""
        eng, sh = self.eng, self.eng.sh
        P = sh.P
        pb = dict(batch)
        pb["targets"] = self.seq[:, P:P + 1]
        a = eng.forward(pb, train_flag=False, training=False, per_row_infer=True, need_logits=False)
        self.c.copy_(a["c"])
        B, T, nH = a["B"], a["T"], sh.nH
        for l, rec in enumerate(a["layers"]):
            qkv = rec[4]
            if isinstance(qkv, hip.Planes):        # bf16x3: c_attn writes a (hi | lo) plane pair [2, M, 3D]; the cache holds fp32 rows
                hi, lo = qkv.t[0].view(B, T, 3, nH, 64), qkv.t[1].view(B, T, 3, nH, 64)
                kv = hi[:, :P, 1:3].float() + lo[:, :P, 1:3].float()           # [B, P, 2, nH, 64]: only what the cache takes
                k, v = kv[:, :, 0], kv[:, :, 1]
            else:
                qkv = qkv.view(B, T, 3, nH, 64)
                k, v = qkv[:, :P, 1], qkv[:, :P, 2]
            self.kc[l][:, :, :P].copy_(k.permute(0, 2, 1, 3))
            self.vc[l][:, :, :P].copy_(v.permute(0, 2, 1, 3))
        self.keep[:, :P] = (self.tpw_mask != 0).to(torch.int32)
        self.pos_all.fill_(P)
        self.first_pos = P

    def step_at(self, pos):
        """Token step `pos` of the generation ``begin`` prepared; returns whether the step called the model's head."""
        sh = self.eng.sh
        j = pos + 1 - sh.P                                      # lyric index appended after this step
        forced = j < 1 or (j > 1 and (j + 1) % (sh.msl + 2) in (0, 1))
        self._run_step(with_head=not forced, parity=pos & 1)
        return not forced

    @torch.no_grad()
    def generate(self, batch, length, temperature=1.0, repitition_penalty=1.0, top_k=1, top_p=0.0, generator=None,
                 use_graph=None, teacher=None, tap=None):
        """batch: dict with topic_ids/tpw_* [B,P], topic_emb, img_embs, r_embs (no targets needed).
        Runs `length` iterations of the reference loop and returns the lyric ids
        [B, 1 + length] (column 0 is the initial [#START#]).  top_k = 1, top_p = 0 is the greedy setting;
        anything else samples on the device (generate.py:137-141): one uniform per row and position is drawn
        from `generator` (a CUDA torch.Generator; default: the global one) before the steps are replayed.
        Parity hooks (tests/test_decode_gpu.py): `teacher` [B, 1 + length] long -- after every step the token the step
        appended is replaced by teacher[:, j] wherever that is >= 0 (teacher forcing on a reference id list; the step's own
        pick is handed to `tap` first); `tap(j, with_head, picked [B], logits [B, Vpad] or None)` is called after every step."""
        n_steps = self.begin(batch, length, temperature, repitition_penalty, top_k, top_p, generator)
        eng, sh = self.eng, self.eng.sh
        saved_mode = self.use_graph
        if use_graph is not None:
            self.use_graph = use_graph
        try:
            if teacher is not None and self.first_pos > 0:          # (the prompt steps a prefill skipped would have placed column 0)
                col = teacher[:, 0].to(eng.dev)
                self.seq[:, sh.P] = torch.where(col >= 0, col, self.seq[:, sh.P])
            for pos in range(self.first_pos, n_steps):
                with_head = self.step_at(pos)
                j = pos + 1 - sh.P                                  # lyric index appended by this step
                if tap is not None:
                    tap(j, with_head, self.seq[:, pos + 1].clone(), self.logits if with_head and not self.children else None)
                if teacher is not None and 0 <= j <= length:
                    col = teacher[:, j].to(eng.dev)
                    self.seq[:, pos + 1] = torch.where(col >= 0, col, self.seq[:, pos + 1])
        finally:
            self.use_graph = saved_mode
        self.check_mlp_error()
        return self.seq[:, sh.P:sh.P + 1 + length].clone()

    def check_mlp_error(self):
        """Raise if a mmtg_decode_mlp launch of this generation reported a bounded wait that ran out or a hand-off group split over
        two XCDs (include/mmtg_hip.h): its results are undefined then.  One device read (the ids are read right after anyway)."""
        if getattr(self, "mlp", False):
            code = int(self.mlp_sync[65].item())
            if code:
                raise RuntimeError("mmtg_decode_mlp reported error %d (2: a wait ran into its 4 ms bound -- the 256 workgroups were not "
                                   "co-resident; 4: a hand-off group sat on two XCDs in the plain mode); set MMTG_DECODE_MLP=0" % code)

    @staticmethod
    def reference_return(ids_row, length, sent=22):
        """What the reference's sample_sequence returns for one row: the sequence as it stood
        before the append of the last iteration that called the model (generate.py:126,144)."""
        last_call = max(i for i in range(length) if not (i > 0 and (i + 2) % sent in (0, 1)))
        return list(ids_row[:1 + last_call])
