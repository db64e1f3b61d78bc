"""The decode half of BASELINE.json's metric: batched greedy generation (configs[3]) and its HBM roofline."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .common import _timed, gpu_rewarm


def bench_decode(args, world, rank, dev, steps, warmup, with_cpu=True, cpu_fn=None):
    """Greedy decode tokens/s: every rank decodes its own batch (replicas only, no exchange).  Returns the result
    object (rank 0) or None.  cpu_fn: bench.py's cpu_decode_baseline (the only code that touches the oracle stays in bench.py)."""
    from mmtg_amd import MMTG, hip, synth
    from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
    from mmtg_amd.decode import GreedyDecoder
    S, V = 5, 13317
    mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
    gcfg = gpt2_config(n_layer=args.layers, vocab_size=V)
    model = MMTG(mcfg, dcfg, V, gpt2_config=gcfg, compute_dtype=args.dtype, token_table=synth.make_token_table(V, seed=2))
    model.reset_parameters(seed=0)
    model.to(dev).eval()
    B, Ln = args.decode_batch, args.decode_len
    nb = synth.make_batch(B, mcfg, dcfg, V, seed=7 + rank)
    batch = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items() if k not in ("rating", "targets")}
    if os.environ.get("MMTG_DECODE_PROF"):
        dec = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=False)
        dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
        hip.prof_enable(True)
        dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
        hip.prof_enable(False)
        pr = hip.prof_read()
        print({k: (v["launches"], round(v["ms"], 2)) for k, v in pr.items() if v["launches"]})
        return None
    dec = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=not getattr(args, "decode_eager", False))
    if not (args.no_roofline and getattr(args, "decode_eager", False)):      # (not in the counter passes: every dispatch is instrumented)
        gpu_rewarm(dev)
    # (counter-collection passes -- eager launches under rocprofv3 --pmc -- may ask for NO warm-up generation: every dispatch is
    #  counted, and the profiler's counter pass has died on runs of much more than 10 k dispatches)
    for _ in range(warmup if (warmup == 0 and getattr(args, "decode_eager", False)) else max(1, warmup)):
        ids = dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
    out_ids = [None]

    def run():
        for _ in range(steps):
            out_ids[0] = dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)

    el = _timed(run, world, dev)
    ids = out_ids[0]
    free = [j for j in range(1, Ln + 1) if (j + 1) % 22 not in (0, 1)]
    check = {"ids_shape": list(ids.shape), "all_rows_start_with_START": bool((ids[:, 0] == 1).all().item()),
             "banned_ids_sampled": int(torch.isin(ids[:, free], torch.tensor([1, 2, 100, 102], device=ids.device)).sum().item()),
             "distinct_ids": int(torch.unique(ids).numel())}
    roof = cpu = None
    step_us = n_token_steps = None
    once_ms = None
    counter_pass = args.no_roofline and getattr(args, "decode_eager", False)
    if rank == 0 and not counter_pass:
        # HIP events on the launch stream around the token steps of one more generation (after its once-per-generation part: weight
        # copies, encoder, the prompt's batched prefill): the token step's duration
        step_us, n_token_steps = token_step_us(dec, batch, Ln)
        once_ms = round(1e3 * el / steps - 1e-3 * step_us * n_token_steps, 3)
    elif rank == 0:
        # (--no-roofline --decode-eager, the counter-collection passes: no extra generation -- every dispatch is instrumented and the
        #  profiler dies on long runs; the figure below then includes the generation's once-only part)
        n_token_steps = dcfg.topic_prompt_length + Ln - dec.first_pos
        step_us = 1e6 * el / steps / n_token_steps
    if rank == 0 and not args.no_roofline:
        roof = decode_roofline(args, model, batch, B, Ln, dec, step_us)
    if rank == 0 and world == 1 and with_cpu and cpu_fn is not None and not args.no_cpu_baseline:
        cpu = cpu_fn(mcfg, dcfg, gcfg, V)
    if rank != 0:
        return None
    out = {"metric": "greedy-decode tokens/sec, full MMTG config", "value": round(B * world * Ln * steps / el, 1),
           "unit": "tokens/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * el / steps, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "batched greedy generation, KV cache, %s: batch %d, "
                                  "%d generated positions after a 15-token prompt (%s), GPT-2 %dL/768/12H V=%d"
                                  % (dec.describe(), B, Ln, "prefilled in one batched pass" if dec.first_pos else "fed as token steps",
                                     args.layers, V),
                      "us_per_token_step": round(step_us, 2), "token_steps_per_generation": n_token_steps,
                      # what a generation spends outside its token steps: fresh weight copies / LayerNorm folds, the encoder, the prompt
                      "once_per_generation_ms": once_ms,
                      "parallelism": "replicas x%d (no exchange)" % world},
           "check": check}
    if roof is not None:
        out["roofline"] = roof
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def decode_mean_kv_rows(first_pos, n_steps):
    """K / V rows a token step touches per layer and batch row, averaged over positions first_pos .. first_pos + n_steps - 1: `pos`
    cached rows read + the step's own row written."""
    return first_pos + (n_steps - 1) / 2.0 + 1.0


def token_step_us(dec, batch, Ln, eager=False):
    """(us per token step, token steps) of one generation: HIP events around the step loop only -- begin() (weight copies, LayerNorm
    folds, encoder, the prompt's prefill) runs before the first event."""
    saved = dec.use_graph
    if eager:
        dec.use_graph = False
    try:
        n = dec.begin(batch, Ln, temperature=1.1, repitition_penalty=1.5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for pos in range(dec.first_pos, n):
            dec.step_at(pos)
        e1.record()
        torch.cuda.synchronize()
    finally:
        dec.use_graph = saved
    return 1e3 * e0.elapsed_time(e1) / (n - dec.first_pos), n - dec.first_pos


def decode_roofline(args, model, batch, B, Ln, dec, step_us_events):
    """A decode token step (one hipGraph replay = the launch) against the HBM roofline.  Algorithmic bytes per token step
    (SURVEY 8(d)): every weight once (bf16) + the KV cache of the prefix read once + one new K/V row written per layer;
    duration = HIP events around a graph-replayed generation / its token steps."""
    from mmtg_amd import hip
    sh = model.shapes
    esz = 2 if args.dtype == "bf16" else 4
    D, L, V, H, E = sh.D, sh.L, sh.V, sh.H, sh.E
    w_bytes = esz * (L * 12 * D * D + V * D + E * H + H * D)
    kv_row = 2 * L * D * esz                                # K and V of one position, all layers
    # live per-launch timing of one generation's token steps through the library's profiling hooks (HIP events on the launch stream)
    saved = dec.use_graph
    dec.use_graph = False
    try:
        n_end = dec.begin(batch, Ln, temperature=1.1, repitition_penalty=1.5)
        first = dec.first_pos
        hip.prof_enable(True)
        for pos in range(first, n_end):
            dec.step_at(pos)
        hip.prof_enable(False)
    finally:
        dec.use_graph = saved
    steps_per_seq = n_end - first
    # the token step at position pos reads the pos cached rows of every layer and writes one: mean over the steps that run
    kv_bytes = B * kv_row * decode_mean_kv_rows(first, steps_per_seq)
    alg = w_bytes + kv_bytes
    pr = hip.prof_read()
    tot_ms = sum(v["ms"] for v in pr.values())
    step_us = step_us_events
    ach = alg / max(step_us, 1e-9) / 1e3            # GB/s
    traffic, tsrc = None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_decode_pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                m = json.load(fh)
        except Exception:
            continue
        if m.get("kernel_source_sha") == hip.source_sha() and B == 256 and m.get("dtype", "bf16") == args.dtype:
            if m.get("step") in ("fused", "bf16x3") and (getattr(dec, "fused", False) or getattr(dec, "x3", False)):
                # counter pass of the fused step, taken at the generation length the file names (the KV-cache share scales with it)
                traffic = m["hbm_bytes_per_token_step"]
                tsrc = os.path.relpath(f,
                    ROOT) + (" [FUSED step at --decode-len %d: %d bytes per token step against %d algorithmic at that length]"
                                                   % (m["decode_len"], m["hbm_bytes_per_token_step"],
                                                       m["algorithmic_bytes_per_token_step_at_this_length"]))
                break
            if m.get("step") == "fused":
                continue
            if m.get("step") == "unfused" and getattr(dec, "fused", False):
                # the counter passes only ran on the round-2 step (round 3): not this step's traffic
                traffic, tsrc = None, os.path.relpath(f,
                    ROOT) + " holds the UNFUSED step's %d bytes per token step; the fused step's counter pass crashes in the profiler" % m["hbm_bytes_per_token_step"]
            else:
                traffic, tsrc = m["hbm_bytes_per_token_step"], os.path.relpath(f, ROOT)
            break
    return {"bound": "hbm", "kernel": dec.kernel_name(), "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(ach / 8000.0, 4), "traffic": traffic,
            "traffic_source": (tsrc + (" (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over every kernel of the token step)" if traffic else "")) if tsrc else
                              "none: no profiles/r*_decode_pmc_traffic.json was taken on kernel sources sha %s" % hip.source_sha()[:12],
            "algorithmic_bytes_per_token_step": int(alg), "weights_bytes": int(w_bytes), "kv_bytes_mean": int(kv_bytes),
            "us_per_token_step_hip_events": round(step_us, 2),
            "eager_kernel_us_per_token_step": round(1e3 * tot_ms / steps_per_seq, 2),
            "launches_per_token_step": round(sum(v["launches"] for v in pr.values()) / steps_per_seq, 1),
            "per_category_ms_per_generation": {k: round(v["ms"], 3) for k, v in pr.items() if v["launches"]},
            "token_steps_per_generation": steps_per_seq, "first_token_step_position": first,
            "note": "achieved = algorithmic bytes of a token step (every weight once + the mean KV prefix of the positions the token steps run at) "
                    "/ HIP-event duration of a graph-replayed token step; the per-category times are an eager (un-captured, host-bound) replay of "
                    "the same token steps through the library's profiling hooks"}
