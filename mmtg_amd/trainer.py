"""Fused training step: the hot loop of the reference's src/train.py:177-200 as one
sequence of HIP kernel launches without host synchronisation (losses, the clip coefficient and the
global row count stay on the device).  The one exception is the in-trainer curriculum filter of
stages 1 / 2 on DEVICE ratings, which reads the selection back once per step; hand the ratings
over on the host as well (``batch["rating_host"]``) or filter in the loader (``DeviceLoader``)
and the step is sync-free in every stage.

    filter rows by curriculum stage  -> forward -> MyLoss (+ alpha * KL) -> backward
    -> bucketed RCCL all-reduce (overlapped) -> clip_grad_norm_(1.0) + AdamW + LR schedule

Optimizer semantics are transformers.AdamW as train.py:137 constructs it
(betas .9/.999, eps 1e-6, weight_decay 0, bias-corrected) with
get_linear_schedule_with_warmup (train.py:146-148).
"""
from __future__ import annotations

import os

import torch

from .ddp import GradReducer


def curriculum_filter(ratings, stage):
    """Row selection + order of train.py:178-183 (negatives first)."""
    r = torch.as_tensor(ratings)
    if stage == 1:
        return torch.cat([torch.where(r < 2)[0], torch.where(r > 4)[0]])
    if stage == 2:
        return torch.cat([torch.where(r < 3)[0], torch.where(r > 3)[0]])
    return torch.arange(len(r), device=r.device)


def linear_schedule(step, warmup, total):
    """LR multiplier of get_linear_schedule_with_warmup after `step` scheduler steps."""
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total - step) / max(1, total - warmup))


_LOGITS_F32 = bool(os.environ.get("MMTG_LOGITS_F32"))     # A/B switch: fp32 logits in the bf16 trainer too


class MMTGTrainer:
    def __init__(self, model, lr=1e-5, alpha=0.0, max_norm=1.0, betas=(0.9, 0.999), eps=1e-6,
                 weight_decay=0.0, warmup_steps=0, total_steps=None, distributed=False, bucket_mb=64.0,
                 lm_weight=0.0):
        self.model = model
        self.eng = model.engine()
        self.lr, self.alpha, self.max_norm = lr, alpha, max_norm
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.warmup, self.total = warmup_steps, total_steps
        self.sched_step = 0
        self.lm_weight = lm_weight
        self.reducer = GradReducer(self.eng.layout, bucket_mb) if distributed else None
        self._count = None
        self.measure_finish, self.finish_events = False, []
        if self.reducer is not None:
            self.eng.bucket_hook = lambda pack: self.reducer.on_pack_ready(self.eng.grad, pack)
        # Self-tuning CU reservation (round 5): a multi-rank GPU run whose environment does not pin MMTG_DDP_GEMM_CUS spends its
        # first steps under each candidate (ddp.BUDGET_CANDIDATES: 0 / 16 / 32 CUs left to the RCCL kernels; TUNE_STEPS real
        # optimisation steps each, the first of every candidate untimed), then all ranks agree on the fastest through one MAX
        # all-reduce of the timings (ddp.agree_on_budget) -- whether RCCL holds CUs beside the backward depends on the node
        # (profiles/r04_v1_ddp_cu_contention_one_gpu.txt priced a wrong guess at 5 % of the step either way).
        from . import ddp as _ddp
        self._tune = None
        self.budget_report = None
        if (self.reducer is not None and self.reducer.world > 1 and self.eng.dev.type == "cuda" and not _ddp.cu_budget_fixed()
                and os.environ.get("MMTG_DDP_TUNE", "1") != "0"):
            self._tune = {"cand": list(_ddp.BUDGET_CANDIDATES), "i": 0, "k": 0, "events": [[] for _ in _ddp.BUDGET_CANDIDATES]}
            _ddp.set_chosen_budget(self._tune["cand"][0])

    def current_lr(self):
        if self.total is None:
            return self.lr
        return self.lr * linear_schedule(self.sched_step, self.warmup, self.total)

    # ---- true resume (the reference saves the model only, train.py:212)
    def state_dict(self):
        """Optimizer moments (flat fp32, the engine's parameter order), AdamW step count, scheduler position
        and the dropout counter: together with ``model.state_dict()`` everything a bit-continuing run needs."""
        eng = self.eng
        return {"format": "mmtg-trainer-1", "layout_total": eng.layout.total,
                "exp_avg": None if eng.opt_m is None else eng.opt_m.detach().cpu().clone(),
                "exp_avg_sq": None if eng.opt_v is None else eng.opt_v.detach().cpu().clone(),
                "step_count": eng.step_count, "sched_step": self.sched_step, "drop_seed": eng.drop_seed,
                # the CU reservation the ranks agreed on (None: not tuned): a resumed run keeps it instead of spending nine steps
                # re-tuning under wall-clock timings (MMTG_DDP_GEMM_CUS pins it from the environment)
                "cu_budget": None if self.budget_report is None else self.budget_report.get("budget_chosen"),
                "hparams": {"lr": self.lr, "alpha": self.alpha, "max_norm": self.max_norm, "betas": tuple(self.betas),
                            "eps": self.eps, "weight_decay": self.wd, "warmup_steps": self.warmup,
                            "total_steps": self.total, "lm_weight": self.lm_weight}}

    def load_state_dict(self, sd):
        eng = self.eng
        if sd.get("format") != "mmtg-trainer-1" or sd.get("layout_total") != eng.layout.total:
            raise ValueError("trainer state of a different model layout (%s, %s parameters; this model has %d)"
                             % (sd.get("format"), sd.get("layout_total"), eng.layout.total))
        for name, key in (("opt_m", "exp_avg"), ("opt_v", "exp_avg_sq")):
            v = sd[key]
            setattr(eng, name, None if v is None else v.to(eng.dev, torch.float32).clone())
        eng.step_count, self.sched_step, eng.drop_seed = int(sd["step_count"]), int(sd["sched_step"]), int(sd["drop_seed"])
        if sd.get("cu_budget") is not None and self._tune is not None:
            from . import ddp as _ddp
            _ddp.set_chosen_budget(int(sd["cu_budget"]))
            self.budget_report = {"budget_chosen": int(sd["cu_budget"]), "source": "checkpoint"}
            self._tune = None

    def step(self, batch, stage=3, filter_rows=True):
        """One optimisation step; returns device scalars (no host sync) {'loss','lm_loss','kl'} of the LOCAL rows.
        Returns None when the stage filter leaves no rows on a single rank (train.py:184-185); in a data-parallel
        group a rank without rows still takes part in the exchange and in the (identical) optimizer step.

        Scaling.  The backward produces the SUM over local rows of the per-row gradients (loss coefficient n_local /
        n_local, KL weight alpha * n_local); the global row count travels as a DEVICE scalar -- all-reduced next to the
        gradient buckets when distributed, never read by the host -- and the clip + AdamW kernel divides by it.  Single
        GPU and data-parallel steps therefore run the same arithmetic, and unequal shards after curriculum filtering
        give the single-GPU global mean.

        One documented difference: when EVERY rank of a multi-rank group is filtered to zero rows, the AdamW kernel sees a
        global count of 0 and leaves parameters and moments untouched, but the host -- which never reads the count --
        still advances the Adam step (bias correction) and the LR schedule by one.  A single process `continue`s before
        either counter moves (train.py:184-185).  Shards of a shuffled global batch that are ALL empty after the stage
        filter do not occur with the released data (every rating occurs in every batch of 64+ rows); the price of the
        exact behaviour would be a host read of the all-reduced count every step."""
        tune_ev = None
        if self._tune is not None:
            tune_ev = torch.cuda.Event(enable_timing=True)
            tune_ev.record()
        out = self._step(batch, stage, filter_rows)
        # (only after a step that RETURNED: the book-keeping ends in a blocking all-reduce, which after an exception on one rank
        #  would turn that rank's error into a hang of the group)
        if tune_ev is not None:
            self._tune_after_step(tune_ev)
        return out

    TUNE_STEPS = 3

    def _tune_after_step(self, ev0):
        """Book-keeping of the CU-reservation tuning: one event pair per step; after TUNE_STEPS steps of the last candidate the
        ranks agree (the only host synchronisation of the tuning) and the choice is installed for the rest of the run."""
        from . import ddp as _ddp
        t = self._tune
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        t["events"][t["i"]].append((ev0, ev1))
        t["k"] += 1
        if t["k"] < self.TUNE_STEPS:
            return
        t["k"] = 0
        t["i"] += 1
        if t["i"] < len(t["cand"]):
            _ddp.set_chosen_budget(t["cand"][t["i"]])
            return
        torch.cuda.synchronize()
        ms = [sum(a.elapsed_time(b) for a, b in evs[1:]) / max(1, len(evs) - 1) for evs in t["events"]]
        choice, agreed = _ddp.agree_on_budget(ms, t["cand"], self.reducer.group, device=self.eng.dev)
        _ddp.set_chosen_budget(choice)
        self.budget_report = {"candidates": t["cand"], "ms_per_step_this_rank": [round(x, 3) for x in ms],
                              "ms_per_step_slowest_rank": [round(x, 3) for x in agreed], "budget_chosen": choice,
                              "steps_per_candidate": self.TUNE_STEPS}
        self._tune = None

    def _step(self, batch, stage, filter_rows):
        eng = self.eng
        if filter_rows and stage in (1, 2):
            # The selected row COUNT shapes every launch of the step, so the host has to know it: computed from a host copy
            # of the ratings when the batch carries one ("rating_host", a CPU tensor -- the loader has the ratings on the
            # host anyway; DeviceLoader goes further and filters before the copy) this costs no device synchronisation;
            # from device ratings alone it costs ONE device -> host read per step in stages 1 / 2 (torch.where).
            idx = curriculum_filter(batch["rating_host"] if "rating_host" in batch else batch["rating"], stage)
            batch = {k: v[idx.to(v.device, non_blocking=True)] for k, v in batch.items() if k != "rating_host"}
        elif "rating_host" in batch:
            batch = {k: v for k, v in batch.items() if k != "rating_host"}
        n_local = int(batch["rating"].shape[0]) if "rating" in batch else int(batch["targets"].shape[0])
        red = self.reducer if (self.reducer is not None and self.reducer.active) else None
        if n_local == 0 and (red is None or red.world == 1):
            return None             # nothing to exchange with: no counter moves, as train.py:184-185 `continue`s
        if self._count is None:
            self._count = torch.zeros(1, device=eng.dev, dtype=torch.float32)
        self._count.fill_(float(n_local))           # asynchronous fill; the value is known on the host
        if red is not None:
            red.start_count(self._count)            # SUM over ranks, in place, asynchronous
        # (a step without rows runs no backward: everything must be zero for the exchange)
        eng.zero_grad((n_local, int(batch["targets"].shape[1])) if n_local > 0 and "targets" in batch else None)
        out = None
        if n_local > 0:
            eng.forward(batch, train_flag=True, training=self.model.training, logits_f32=_LOGITS_F32)
            sc = eng.loss(batch["rating"], stage, batch_den=n_local)
            T = eng.act["T"]
            dl = eng.loss_backward(float(n_local), lm_coef=self.lm_weight / (T - 1) if self.lm_weight else 0.0)
            eng.wgrad_overwrite = True      # gradients were zeroed above and every weight is written once
            try:
                eng.backward(dl, dkl=self.alpha * n_local)
                eng._finish_overwrite_record()
            finally:
                eng.wgrad_overwrite = False
                eng._ow_rec = None
            out = {"loss": sc[0], "lm_loss": sc[1], "kl": eng.act["kl"][0]}
        if red is not None:
            if self.measure_finish:     # exposed exchange time: how long the compute stream sits in finish() (bench.py --gpus N)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                red.finish(eng.grad)
                e1.record()
                self.finish_events.append((e0, e1))
            else:
                red.finish(eng.grad)
        eng.adamw_step(self.current_lr(), self.max_norm, self.betas, self.eps, self.wd, count=self._count)
        self.sched_step += 1
        return out

    # ---- evaluation (train.py:241-268)
    @torch.no_grad()
    def evaluate_batch(self, batch, stage=3, filter_rows=True):
        """Forward + MyLoss (+ KL) of one validation batch, no update: the stage's row filter, the engine's forward in eval mode (no
        dropout) with compute-dtype logits -- no fp32 [B, T, V] tensor, no autograd boundary -- and the loss kernels.  Returns the
        DEVICE scalars {'loss': MyLoss mean, 'kl': kl, 'total': loss + alpha * kl} (no host sync), or None when the filter leaves
        no rows (train.py:252-253 `continue`s)."""
        eng = self.eng
        if filter_rows and stage in (1, 2):
            idx = curriculum_filter(batch["rating_host"] if "rating_host" in batch else batch["rating"], stage)
            batch = {k: v[idx.to(v.device, non_blocking=True)] for k, v in batch.items() if k != "rating_host"}
        elif "rating_host" in batch:
            batch = {k: v for k, v in batch.items() if k != "rating_host"}
        if int(batch["rating"].shape[0]) == 0:
            return None
        eng.forward(batch, train_flag=True, training=False, logits_f32=_LOGITS_F32)
        sc = eng.loss(batch["rating"], stage)
        loss, kl = sc[0], eng.act["kl"][0]
        return {"loss": loss, "kl": kl, "total": loss + self.alpha * kl}

    @torch.no_grad()
    def evaluate(self, valid_batches, stage=3):
        """The reference's evaluate() (train.py:241-268) over an iterable of validation batches: per batch
        total = MyLoss.mean() + alpha * kl.mean(), summed -- on the device, one host read at the end -- and divided by the NUMBER OF
        BATCHES (filtered-out ones included, as `valid_loss /= len(valid_dataset)` does).  Returns (valid_loss, kldiv_loss) floats;
        the model is in eval mode during the pass (train.py:242) and gets its previous mode back afterwards."""
        was_training = self.model.training
        self.model.eval()
        acc = torch.zeros(2, device=self.eng.dev, dtype=torch.float32)
        n = 0
        try:
            for batch in valid_batches:
                n += 1
                out = self.evaluate_batch(batch, stage)
                if out is None:
                    continue
                acc[0] += out["total"]
                acc[1] += self.alpha * out["kl"]
        finally:
            self.model.train(was_training)
        if n == 0:
            return 0.0, 0.0
        v = (acc / n).tolist()
        return v[0], v[1]

    def finish_wait_ms(self):
        """Mean time per step the compute stream waited in the reducer's finish() -- the part of the gradient exchange the
        backward did NOT hide -- over the steps taken with ``measure_finish`` on (synchronises; clears the record)."""
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.finish_events]
        self.finish_events = []
        return sum(ms) / len(ms) if ms else None

    def grad_norm(self):
        """Global gradient norm of the last step as clip_grad_norm_ saw it (device scalar): the flat buffer holds row
        sums, so sqrt(sum g^2) / global row count."""
        return torch.sqrt(self.eng.normsq[0]) / self._count[0]


def save_checkpoint(path, model, trainer=None, args=None, model_cfgs=None, reference_compatible=False):
    """The reference's checkpoint dict ``{'model', 'args', 'model_cfgs'}`` (train.py:212) plus, when a trainer is
    given, ``'trainer'`` for resuming.

    reference_compatible=True writes the state dict exactly as the reference's own run would have: keys carry the
    ``module.`` prefix of its nn.DataParallel wrapper (train.py:113,212) and every GPT-2 block has the persistent
    ``attn.bias`` / ``attn.masked_bias`` buffers of transformers 4.12.3, so the reference's strict
    ``load_state_dict`` into its DataParallel-wrapped model (generate.py:191-192) accepts the file.  The default
    writes ``model.state_dict()`` (what ``load_checkpoint`` and any unwrapped consumer read)."""
    if reference_compatible:
        sd = {"module." + k: v for k, v in model.legacy_state_dict().items()}
    else:
        sd = model.state_dict()
    ckpt = {"model": {k: v.detach().cpu() for k, v in sd.items()}, "args": args, "model_cfgs": model_cfgs}
    if trainer is not None:
        ckpt["trainer"] = trainer.state_dict()
    torch.save(ckpt, path)


def load_checkpoint(path, model, trainer=None):
    """Loads a checkpoint written by ``save_checkpoint`` or by the reference's train.py ('module.' prefixes and
    transformers-4.12.3 buffers accepted).  Returns the checkpoint dict."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt)
    if trainer is not None and ckpt.get("trainer") is not None:
        trainer.load_state_dict(ckpt["trainer"])
    return ckpt
