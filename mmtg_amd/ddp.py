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

MMTG_DDP_COMM=abi (opt-in, round 6): the buckets go through libmmtg_hip's own RCCL communicator (SURVEY.md section 8b's second
small ABI, csrc/comm.hip: mmtg_comm_init / mmtg_allreduce_bucket_async / mmtg_comm_join) -- one C call per bucket from the
backward, the fork from and the join with the compute stream done with HIP events inside the library -- and torch.distributed
stays the control plane only (rendezvous: it ships rank 0's unique id; the self-tuning's MAX all-reduce).

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


def grad_exchange_dtype():
    """MMTG_DDP_GRAD_DTYPE=bf16 (opt-in, round 6): the gradient buckets cross xGMI as bf16 -- cast, SUM all-reduce in bf16, cast
    back into the flat fp32 buffer -- half of the 436 MB per rank and step.  For the bf16 compute mode, whose gradients carry bf16
    rounding anyway; the default (fp32) is what the reference's arithmetic implies and what the parity tests hold."""
    v = os.environ.get("MMTG_DDP_GRAD_DTYPE", "f32").lower()
    if v in ("f32", "fp32", "float32"):
        return torch.float32
    if v in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError("MMTG_DDP_GRAD_DTYPE must be f32 or bf16, got %r" % v)


def comm_backend():
    """MMTG_DDP_COMM: "torch" (default: torch.distributed.all_reduce per bucket) or "abi" (libmmtg_hip's RCCL communicator)."""
    v = os.environ.get("MMTG_DDP_COMM", "torch").lower()
    if v not in ("torch", "abi"):
        raise ValueError("MMTG_DDP_COMM must be torch or abi, got %r" % v)
    return v


def abi_comm_acquire(group=None):
    """The process's RCCL communicator behind the C ABI, created on first use (collective: every rank of `group` calls it): rank 0
    draws the unique id, torch.distributed -- whatever its backend -- carries the 128 bytes to the others, every rank joins on its
    current device.  It lives until the process ends (abi_comm_release() for an orderly shutdown); a second acquire checks that
    the live communicator is this group's."""
    from . import hip
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    info = hip.comm_info()
    if info["world"]:
        if (info["rank"], info["world"]) != (rank, world):
            raise RuntimeError("the live communicator is rank %d of %d, this group wants rank %d of %d"
                               % (info["rank"], info["world"], rank, world))
        return info
    box = [hip.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    hip.comm_init(rank, world, box[0])
    import atexit
    atexit.register(abi_comm_release)
    return hip.comm_info()


def abi_comm_release():
    from . import hip
    hip.comm_destroy()


class _EventHandle:
    """What finish() waits on for a bucket launched through the C ABI on a host-owned side stream (the measured form)."""

    def __init__(self, done):
        self.done = done

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)


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
        self.xdtype = grad_exchange_dtype()
        # MMTG_DDP_COMM=abi: GPU runs only (the communicator is RCCL's); CPU tensors (gloo tests) keep torch.distributed
        self.abi = comm_backend() == "abi" and torch.cuda.is_available() and (self.world > 1 or self.force)
        self.comm_info = abi_comm_acquire(group) if self.abi else None
        self._side = None           # host-owned side stream of the measured form
        self._stage = {}            # bucket index -> bf16 staging tensor (MMTG_DDP_GRAD_DTYPE=bf16), reused step after step
        # first-contact instrumentation (bench.py --gpus N): per bucket, when the backward handed it to RCCL and how long the
        # compute stream then sat in its wait -- events on the compute stream, collected only while `measure` is on
        self.measure = False
        self.timeline = []
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
        self._pending = []          # (bucket index, staging tensor) of the buckets in flight in a narrower dtype
        self._marks = []            # measure: (bucket index, launch event)
        self._abi_pending = False   # buckets on the library's side stream since the last join

    def _event(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def start_count(self, count):
        """Asynchronous SUM all-reduce (in place) of the one-element device tensor holding this rank's row count."""
        if self.active:
            self.handles.append(self._launch(count))

    def on_pack_ready(self, grad_flat, pack_name):
        self.on_ready(grad_flat, self.pack_end[pack_name])

    def on_ready(self, grad_flat, end_offset):
        """All gradient elements below end_offset are final: launch every complete bucket."""
        if not self.active:
            return
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][1] <= end_offset:
            i = self.next_bucket
            s, e = self.buckets[i]
            buf = grad_flat[s:e]
            if self.xdtype != torch.float32:
                st = self._stage.get(i)
                if st is None or st.numel() != e - s or st.device != buf.device:
                    st = self._stage[i] = torch.empty(e - s, dtype=self.xdtype, device=buf.device)
                st.copy_(buf)               # the cast (round to nearest even), on the compute stream, before the collective reads it
                self._pending.append((i, st))
                buf = st
            if self.measure and grad_flat.is_cuda:
                self._marks.append((i, self._event()))
            self.handles.append(self._launch(buf))
            self.next_bucket += 1

    def _launch(self, buf):
        """One asynchronous in-place SUM all-reduce; returns what finish() waits on (None: the library's join covers it)."""
        if not (self.abi and buf.is_cuda):
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        from . import hip
        if not self.measure:
            hip.allreduce_bucket_async(buf)         # fork by event inside the library; finish() joins once
            self._abi_pending = True
            return None
        # measured form: the same collective on a side stream the host owns, so that every bucket has its own end event
        if self._side is None:
            self._side = torch.cuda.Stream(device=buf.device)
        self._side.wait_stream(torch.cuda.current_stream())
        hip.allreduce_bucket(buf, stream=self._side)
        done = torch.cuda.Event()
        done.record(self._side)
        return _EventHandle(done)

    def tail_bytes(self):
        """Bytes of the last bucket: what can only leave after the backward's last kernel (the exposed part's upper bound)."""
        s, e = self.buckets[-1]
        return 4 * (e - s)

    def finish(self, grad_flat):
        """Flush the remaining buckets and make the current stream wait for all of them (and for the row count)."""
        if self.active:
            self.on_ready(grad_flat, self.layout.total)
            timed = self.measure and grad_flat.is_cuda
            waits = [self._event()] if timed else None
            for h in self.handles:
                if h is not None:
                    h.wait()
                if timed:
                    waits.append(self._event())
            if self._abi_pending:
                from . import hip
                hip.comm_join()                 # the compute stream waits for the library's side stream (no host wait)
            for i, st in self._pending:         # narrow exchange: the reduced sums back into the flat fp32 buffer
                s, e = self.buckets[i]
                grad_flat[s:e].copy_(st)
            if timed:
                self.timeline.append((list(self._marks), waits, len(self.handles)))
        self.reset()

    def timeline_report(self):
        """Per bucket, averaged over the measured steps (synchronises; clears the record): `launch_ms` = when the backward handed the
        bucket to RCCL, relative to the step's first launch; `exposed_wait_ms` = how long the compute stream sat in that bucket's
        wait() inside finish() -- the part of its all-reduce the backward did not hide (the row-count all-reduce is handle 0 when
        present and is folded into the first bucket's wait)."""
        if not self.timeline:
            return None
        torch.cuda.synchronize()
        nb = len(self.buckets)
        launch, wait, n = [0.0] * nb, [0.0] * nb, 0
        for marks, waits, nh in self.timeline:
            if len(marks) != nb:
                continue
            n += 1
            t0 = marks[0][1]
            for i, ev in marks:
                launch[i] += t0.elapsed_time(ev)
            extra = nh - nb                      # leading non-bucket handles (the row count)
            for i in range(nb):
                a = waits[0] if i == 0 else waits[extra + i]
                wait[i] += a.elapsed_time(waits[extra + i + 1])
        self.timeline = []
        if not n:
            return None
        return {"steps": n, "exchange_dtype": str(self.xdtype).replace("torch.", ""),
                "bucket_mb": [round((e - s) * (4 if self.xdtype == torch.float32 else 2) / 2 ** 20, 1) for s, e in self.buckets],
                "launch_ms_after_first": [round(x / n, 3) for x in launch], "exposed_wait_ms": [round(x / n, 3) for x in wait]}


def shard_rows(n_rows, rank, world):
    """Contiguous row shard [lo, hi) of a global batch (sizes differ by at most 1)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
