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

Works unchanged on CPU tensors with the gloo backend (tests/test_ddp_cpu.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, layout, bucket_mb=64.0, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.layout = layout
        # MMTG_FORCE_DDP: run the collectives even at world size 1 (single-GPU self-test of the RCCL path)
        self.force = dist.is_initialized() and bool(__import__("os").environ.get("MMTG_FORCE_DDP"))
        self.buckets = layout.buckets(int(bucket_mb * 1024 * 1024 / 4))
        self.pack_end = {name: o + n for name, (o, n) in layout.pack_range.items()}
        self.reset()

    def reset(self):
        self.next_bucket = 0
        self.handles = []

    def on_pack_ready(self, grad_flat, pack_name):
        self.on_ready(grad_flat, self.pack_end[pack_name])

    def on_ready(self, grad_flat, end_offset):
        """All gradient elements below end_offset are final: launch every complete bucket."""
        if self.world == 1 and not self.force:
            return
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][1] <= end_offset:
            s, e = self.buckets[self.next_bucket]
            self.handles.append(dist.all_reduce(grad_flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.next_bucket += 1

    def finish(self, grad_flat):
        """Flush the remaining buckets and make the current stream wait for all of them."""
        if self.world > 1 or self.force:
            self.on_ready(grad_flat, self.layout.total)
            for h in self.handles:
                h.wait()
        self.reset()

    def global_count(self, n_local, device):
        """Sum of per-rank row counts (curriculum filtering makes them unequal)."""
        if self.world == 1:
            return int(n_local)
        t = torch.tensor([float(n_local)], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(round(t.item()))


def shard_rows(n_rows, rank, world):
    """Contiguous row shard [lo, hi) of a global batch (sizes differ by at most 1)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
