"""Timing helpers shared by bench.py and its parts."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_PROFILING_RUN = False          # set by bench.main(): --no-check marks a profiling / counter pass (no re-warm launches in its statistics)


def _host_threads():
    """Cores this process may actually run on (the box advertises more logical CPUs than the job's
    affinity / cgroup grants; oversubscribing them stalls OpenMP)."""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    quota = ncpu
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    return max(1, min(ncpu, quota, 64))


def gpu_rewarm(dev, seconds=0.4, max_launches=400):
    """Keep the matrix cores busy for a moment before an optional object's warm-up: the CPU baselines leave the GPU idle for up to
    two minutes, and the first launches after that run at ramping clocks (one default run measured its first decode generation at
    ~400 ms instead of 90 with only the object's own one-generation warm-up in front of it).  Outside every timed region.
    NOT in profiling passes (--no-check, or MMTG_BENCH_NO_REWARM=1): its 4096^3 products dispatch as gemm_p8_kernel and would be
    averaged into the GEMM family's per-launch counters; and bounded by a launch count as well as by time (under --pmc every
    dispatch is serialised, a wall-clock bound alone would instrument an unbounded number of them)."""
    if _PROFILING_RUN or os.environ.get("MMTG_BENCH_NO_REWARM"):
        return
    from mmtg_amd import hip
    a = torch.randn(4096, 4096, device=dev).bfloat16()
    c = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds and done < max_launches:
        for _ in range(20):
            hip.gemm(a, a, c, 4096, 4096, 4096, transB=True)
        done += 20
        torch.cuda.synchronize()


def _timed(fn, world, dev):
    """barrier + synchronize on both sides of fn(); max over ranks."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


def _event_us(call, iters=20, warm=3):
    """Mean duration of call() in us: HIP events on the launch stream around `iters` back-to-back launches."""
    for _ in range(warm):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


import contextlib as _contextlib


@_contextlib.contextmanager
def launches_unshared():
    """For an instrumented pass (HIP events around every launch): every kernel alone on the GPU.  The product schedule runs the last 8
    blocks' grouped weight-gradient launches on a side stream beside the backward's small-kernel tail (engine._WGRAD_TAIL); there a
    launch's event-to-event time includes the time it shares the CUs, which says nothing about the kernel.  `value` / `ms_per_step`
    are always measured in the product schedule."""
    import mmtg_amd.engine as E
    saved = E._WGRAD_TAIL
    E._WGRAD_TAIL = 0
    try:
        yield
    finally:
        E._WGRAD_TAIL = saved


UNSHARED_NOTE = ("achieved / frac: HIP events around every launch of an instrumented pass in which every kernel runs alone "
                 "(MMTG_WGRAD_TAIL=0 for that pass); the product schedule -- what `value` is measured in -- runs 8 of the grouped "
                 "weight-gradient launches beside the backward's small-kernel tail, where event-to-event time includes the sharing "
                 "(frac_in_product_schedule)")
