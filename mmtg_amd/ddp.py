"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (src/train.py:112-114:
per-step parameter broadcast, logits gather to GPU0, gradient reduce to GPU0) with
the only exchange the maths needs: a SUM all-reduce of the flat fp32 gradient buffer.

The engine lays gradients out in the order they become final, so a bucket is a
contiguous slice; ``on_ready(end_offset)`` is called from the backward as soon as
everything below ``end_offset`` is final and launches the finished buckets
asynchronously -- RCCL runs them on its own stream while the backward continues.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large (default
64 MB) so each ring step moves enough bytes per link to hide its latency.

The global row count (curriculum filtering leaves the ranks with unequal shards) is all-reduced the same way as a
one-element device tensor and consumed on the device by the clip + AdamW kernel: no host synchronisation.

Works unchanged on CPU tensors with the gloo backend (tests/test_host_cpu.py::test_bucketed_allreduce_world2_gloo);
executed on the GPU over RCCL by tests/test_ddp_gpu.py (world size 1 with MMTG_FORCE_DDP=1, and 2 ranks where two
GPUs are visible).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# The CU reservation is a process-global setting of the GEMM library; reducers come and go (a trainer rebuilt after a
# checkpoint resume, a test that builds several), so it is reference-counted here: the first live reducer sets it, the last
# one to close gives the CUs back -- a stale reducer collected late cannot clear a live one's reservation.
_budget_users = 0


# what the self-tuning picked for this process (None: not tuned -- the environment's value or the default applies)
_budget_chosen = None
BUDGET_CANDIDATES = (0, -16, -32)
# packs after which a bucket always ends (ParamLayout.buckets split_after): the tail of the backward
TAIL_SPLIT = tuple(x for x in os.environ.get("MMTG_DDP_TAIL_SPLIT", "wpe,att_b").split(",") if x)


def cu_budget_fixed():
    """True when the environment pins the reservation (MMTG_DDP_GEMM_CUS): no self-tuning then."""
    return "MMTG_DDP_GEMM_CUS" in os.environ


def cu_budget_setting():
    """The CU reservation a data-parallel run asks the GEMM tile rule for (< 0: CUs left to RCCL): the value the self-tuning
    agreed on, else MMTG_DDP_GEMM_CUS, else 32 CUs left to the collectives."""
    if _budget_chosen is not None and not cu_budget_fixed():
        return _budget_chosen
    return int(os.environ.get("MMTG_DDP_GEMM_CUS", "-32"))


def agree_on_budget(times_ms, candidates=BUDGET_CANDIDATES, group=None, device=None):
    """Every rank timed the same few steps under each candidate reservation; the group keeps the candidate whose SLOWEST rank
    was fastest (one MAX all-reduce of the timings: every rank computes the same arg-min -- first candidate on ties -- so the
    tile rule, and with it the arithmetic order of every product, stays identical across ranks).  Returns (choice, agreed ms)."""
    t = torch.tensor([float(x) for x in times_ms], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    agreed = [float(x) for x in t.tolist()]
    best = min(range(len(candidates)), key=lambda i: (agreed[i], i))
    return candidates[best], agreed


def set_chosen_budget(value):
    """Install the agreed reservation (process-global, like the library setting it drives)."""
    global _budget_chosen
    _budget_chosen = value
    if _budget_users > 0:
        from . import hip
        hip.gemm_cu_budget(cu_budget_setting())


def _budget_acquire():
    global _budget_users
    from . import hip
    if _budget_users == 0:
        hip.gemm_cu_budget(cu_budget_setting())
    _budget_users += 1


def _budget_release():
    global _budget_users
    from . import hip
    _budget_users = max(0, _budget_users - 1)
    if _budget_users == 0:
        hip.gemm_cu_budget(0)


class GradReducer:
    def __init__(self, layout, bucket_mb=64.0, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.layout = layout
        # MMTG_FORCE_DDP: run the collectives even at world size 1 (single-GPU self-test of the RCCL path)
        self.force = dist.is_initialized() and bool(os.environ.get("MMTG_FORCE_DDP"))
        self.buckets = layout.buckets(int(bucket_mb * 1024 * 1024 / 4), split_after=TAIL_SPLIT)
        self._budget_set = False
        if self.world > 1 and torch.cuda.is_available():
            # the RCCL kernels run beside the backward and hold their CUs for the whole collective; a workgroup of the
            # eight-phase GEMM kernel needs a CU to itself, so its tile rule must not plan on all 256 (MMTG_DDP_GEMM_CUS:
            # > 0 = CUs it may count on, < 0 = CUs to leave to the collectives; default 32 left).  GPU runs only: on CPU
            # tensors (gloo tests) there is no library to tell.  The setting is process-global and reference-counted
            # (_budget_acquire / _budget_release): close() of the LAST live reducer restores it.
            _budget_acquire()
            self._budget_set = True
        self.pack_end = {name: o + n for name, (o, n) in layout.pack_range.items()}
        self.reset()

    def close(self):
        """Give the CUs reserved for the collectives back to the GEMM tile rule (decode / inference in the same process)."""
        if self._budget_set:
            self._budget_set = False
            _budget_release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def active(self):
        """True when collectives run: more than one rank, or the forced single-rank self-test."""
        return self.world > 1 or self.force

    def reset(self):
        self.next_bucket = 0
        self.handles = []

    def start_count(self, count):
        """Asynchronous SUM all-reduce (in place) of the one-element device tensor holding this rank's row count."""
        if self.active:
            self.handles.append(dist.all_reduce(count, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def on_pack_ready(self, grad_flat, pack_name):
        self.on_ready(grad_flat, self.pack_end[pack_name])

    def on_ready(self, grad_flat, end_offset):
        """All gradient elements below end_offset are final: launch every complete bucket."""
        if not self.active:
            return
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][1] <= end_offset:
            s, e = self.buckets[self.next_bucket]
            self.handles.append(dist.all_reduce(grad_flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.next_bucket += 1

    def tail_bytes(self):
        """Bytes of the last bucket: what can only leave after the backward's last kernel (the exposed part's upper bound)."""
        s, e = self.buckets[-1]
        return 4 * (e - s)

    def finish(self, grad_flat):
        """Flush the remaining buckets and make the current stream wait for all of them (and for the row count)."""
        if self.active:
            self.on_ready(grad_flat, self.layout.total)
            for h in self.handles:
                h.wait()
        self.reset()


def shard_rows(n_rows, rank, world):
    """Contiguous row shard [lo, hi) of a global batch (sizes differ by at most 1)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
