#!/usr/bin/env python3
"""Headline benchmark: training tokens/s of the full MMTG configuration on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = the reference's hot loop (src/train.py:177-200) on one synthetic batch of the
released shape: forward (encoder + fuser + conditioning + 12-layer GPT-2) -> MyLoss
(+ alpha*KL) -> backward -> bucketed RCCL gradient all-reduce -> clip + AdamW, with GPT-2's
three dropout sites active (model.train()).  Batches are resident in HBM before the timed
region.  Weak scaling: 64 rows per GPU (global 512 at 8 GPUs).

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs / HIP-event time, measured
                live with events on the launch stream during a second, instrumented pass over
                the same steps (the first pass is timed without events and gives `value`)
  cpu_baseline  the CPU oracle (PyTorch fp32 restatement, parity-pinned to the reference) doing
                the same train step on this box's host cores on a bounded sample (B=4 and B=32)
  check         evidence that the timed steps did the training maths: MyLoss before the first and
                after the last step on a held-out probe batch, parameter finiteness, kernel launches
                per step (outside the timed region)
  decode        the second half of BASELINE.json's metric, measured in the same run after the training
                region: batched greedy generation (batch 256, 128 positions) with its own roofline and
                cpu_baseline objects (python bench.py --mode decode prints it as a line of its own)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# (before the HIP runtime can initialise: kernel arguments in device memory, as `import mmtg_amd` sets it for any user of the
#  package -- mmtg_amd/__init__.py has the measurement)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# (multi-process GPU work on this pool: the host driver only supports dmabuf IPC -- without it RCCL fails with
#  `hipIpcGetMemHandle: invalid argument`; already exported by the harness, kept here for a bare launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bench_parts import common as _common  # noqa: E402
from bench_parts.common import _host_threads, _timed, gpu_rewarm  # noqa: E402
from bench_parts.decode import bench_decode  # noqa: E402
from bench_parts.launch import _claim_stdout, _dry_launch, _emit, _self_launch  # noqa: E402
from bench_parts.objects import _pmc_traffic, allreduce_probe, conditioning_probe, f32_object, medium_object  # noqa: E402


def algorithmic_flops_per_token(S, T, D, L, V, H=512, E=2048):
    """forward FLOPs per decoder token (causal-half attention), x3 for training."""
    layer = 2 * (D * 3 * D + D * D + 2 * D * 4 * D)
    attn = 2 * T * D          # QK^T + PV over the causal half
    head = 2 * D * V
    proj = 2 * (E * H + H * D)
    fwd = L * (layer + attn) + head + proj
    enc_per_sample = 2 * (E * H) + 2 * S * 2 * (E * 3 * H + H * 3 * H) + 2 * S * 2 * (H * 3 * H) + 2 * S * H * E
    return fwd, enc_per_sample


def cpu_baseline(mcfg, dcfg, gcfg, V, T, seconds_budget=100.0):
    """Oracle train step (fwd + MyLoss + bwd + clip + AdamW) on the host cores: B=4 and B=32, 3 warm-up + 5 timed
    steps each as SURVEY 8(d) asks, every leg cut short by a time budget (the sample string says what ran)."""
    from mmtg_amd import synth
    from oracle import mmtg_oracle as O
    threads = _host_threads()
    torch.set_num_threads(threads)
    weights = synth.make_weights(mcfg, gcfg, seed=1)
    table = torch.from_numpy(synth.make_token_table(V, seed=2))
    sh = O.Shapes(mcfg, dcfg, gcfg)
    legs = []
    for B, budget in ((4, 0.2 * seconds_budget), (32, 0.8 * seconds_budget)):
        batch = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_batch(B, mcfg, dcfg, V, seed=3).items()}
        w = O.weights_to_torch(weights, requires_grad=True)
        state = {}
        t_leg = time.perf_counter()
        warm = 0
        while warm < 3 and (warm == 0 or time.perf_counter() - t_leg < 0.4 * budget):
            O.train_step(w, sh, table, batch, batch["rating"], 3, 0.2, 1e-5, warm + 1, state)
            warm += 1
        t0 = time.perf_counter()
        n = 0
        while n < 5 and (n == 0 or time.perf_counter() - t_leg < budget):
            O.train_step(w, sh, table, batch, batch["rating"], 3, 0.2, 1e-5, warm + n + 1, state)
            n += 1
        el = time.perf_counter() - t0
        legs.append({"rows": B, "warmup_steps": warm, "timed_steps": n, "tokens_per_s": round(B * T * n / el, 2)})
    best = max(legs, key=lambda l: l["tokens_per_s"])
    return {"value": best["tokens_per_s"], "unit": "tokens/s", "cores": threads, "kind": "port", "host_logical_cpus": os.cpu_count(),
            "legs": legs,
            "sample": "oracle (CPU PyTorch fp32 restatement) full config 12L/768/V=%d, T=%d, train steps (fwd+MyLoss+bwd+clip+AdamW), "
                      "dropout off; B=4: %d warm-up + %d timed, B=32: %d warm-up + %d timed (3 + 5 asked, cut by a %d s budget); "
                      "value = the faster leg" % (V, T, legs[0]["warmup_steps"], legs[0]["timed_steps"], legs[1]["warmup_steps"],
                                                  legs[1]["timed_steps"], int(seconds_budget))}


def cpu_decode_baseline(mcfg, dcfg, gcfg, V, positions=220, seconds_budget=45.0):
    """Oracle greedy decoding on the host cores, batch 1, as SURVEY 8(d) defines the leg: `positions` (220) lyric positions after
    the 15-token prompt, BOTH ways -- as the reference runs it (generate.py:117-142: no KV cache, the whole prefix re-run for every
    token, O(L^2)) and with per-layer K / V kept (oracle.CachedForward).  The cached loop runs first and in full; the reference-shaped
    loop is cut at the time budget (the sample string says how far it got: its rate falls with the prefix length)."""
    from mmtg_amd import synth
    from oracle import mmtg_oracle as O
    threads = _host_threads()
    torch.set_num_threads(threads)
    weights = synth.make_weights(mcfg, gcfg, seed=1)
    table = torch.from_numpy(synth.make_token_table(V, seed=2))
    nb = synth.make_batch(1, mcfg, dcfg, V, seed=3)
    sh = O.Shapes(mcfg, dcfg, gcfg)
    w = O.weights_to_torch(weights, requires_grad=False)
    start = {k: np.asarray(v[0]) for k, v in nb.items() if k not in ("rating", "targets")}
    start["targets"] = np.asarray([1])
    kw = dict(temperature=1.1, top_k=1, top_p=0.0, repitition_penalty=1.5, greedy=True)

    def fwd(inputs):
        return O.mmtg_forward(w, sh, table, inputs, train_flag=False)[2]

    O.sample_sequence(fwd, start, 2, **kw)   # warm-up
    t0 = time.perf_counter()
    ids_c = O.sample_sequence(O.CachedForward(w, sh, table), start, positions, **kw)
    el_c = time.perf_counter() - t0
    # the reference-shaped loop, position by position under the budget (sample_sequence is deterministic: a longer run extends a shorter one)
    done, el_n, ids_n = 0, 0.0, None
    for n in (24, 64, 128, positions):
        n = min(n, positions)
        if n <= done:
            continue
        est = el_n * (n / max(done, 1)) ** 2 if done else 0.0          # O(L^2): time grows with the square of the length
        if done and el_n + est > seconds_budget:
            break
        t0 = time.perf_counter()
        ids_n = O.sample_sequence(fwd, start, n, **kw)
        el_n, done = time.perf_counter() - t0, n
    same = ids_n is not None and ids_c[:len(ids_n)] == ids_n
    return {"value": round(done / el_n, 2), "unit": "tokens/s", "cores": threads, "kind": "port", "host_logical_cpus": os.cpu_count(),
            "no_cache": {"positions": done, "seconds": round(el_n, 2), "tokens_per_s": round(done / el_n, 2)},
            "kv_cached": {"positions": positions, "seconds": round(el_c, 2), "tokens_per_s": round(positions / el_c, 2)},
            "ids_agree": bool(same),
            "sample": "oracle (CPU PyTorch fp32 restatement) greedy decoding, batch 1, full 12L/768/V=%d after the 15-token prompt: value = as "
                      "the reference runs it (no KV cache, prefix re-run per token) over %d of the %d positions asked (cut by a %d s budget); "
                      "kv_cached = the same loop with per-layer K / V kept, all %d positions" % (V, done, positions, int(seconds_budget),
                          positions)}






def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="rows per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3", "bf16x3f"])
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--config", default="base", choices=["base", "medium"],
                    help="medium: BASELINE configs[4] (GPT-2-medium 24L/1024/16H, S=8, T=512, rating skew K=32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode object of the default line")
    ap.add_argument("--no-check", action="store_true", help="skip the probe-loss evidence (profiling runs: keeps the kernel "
                                                            "statistics to the training steps only)")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--mode", default="train", choices=["train", "decode"],
                    help="decode: batched greedy generation (BASELINE configs[3]: batch 256, max_len 128)")
    ap.add_argument("--decode-batch", type=int, default=256)
    ap.add_argument("--decode-len", type=int, default=128)
    ap.add_argument("--decode-eager", action="store_true",
        help="decode without graph capture (counter-collection passes: every dispatch visible)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch contract only (no GPU): ranks rendezvous over gloo, rank 0 prints one JSON line")
    ap.add_argument("--no-f32", action="store_true", help="skip the f32 (parity-gate mode) object of the default line")
    ap.add_argument("--no-x3", action="store_true", help="skip the bf16x3 (split-precision parity mode) object of the default line")
    ap.add_argument("--no-medium", action="store_true", help="skip the configs[4] (GPT-2-medium, T = 512) object of the default line")
    ap.add_argument("--primary-only", action="store_true",
        help="the primary train measurement only: no decode / bf16x3 / f32 / medium objects (profiling passes)")
    args = ap.parse_args()
    if args.primary_only:
        args.no_decode = args.no_x3 = args.no_f32 = args.no_medium = True
    _common._PROFILING_RUN = bool(args.no_check)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one (before anything touches the GPU)
        raise SystemExit(_self_launch(sys.argv[1:], args.gpus))
    if args.dry_launch:
        _dry_launch(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # MMTG_BENCH_ONE_GPU_BACKEND=gloo (rehearsal only, never a measurement): every rank on cuda:0, the exchange through the host --
    # the N > 1 code of this file and of the trainer run on a one-GPU box (RCCL refuses two ranks on one device); the line says so
    one_gpu_backend = os.environ.get("MMTG_BENCH_ONE_GPU_BACKEND")
    if one_gpu_backend:
        local = 0
    dev = torch.device("cuda", local)
    force_ddp = bool(os.environ.get("MMTG_FORCE_DDP"))     # exercise the RCCL path on one GPU (self-test)
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu_backend:
            dist.init_process_group(one_gpu_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.cuda.set_device(local)

    from mmtg_amd import MMTG, hip, synth
    from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
    from mmtg_amd.trainer import MMTGTrainer

    if args.mode == "decode":
        out = bench_decode(args, world, rank, dev, args.steps, args.warmup, cpu_fn=cpu_decode_baseline)
        if rank == 0 and out is not None:
            _emit(out)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    V = 13317
    if args.config == "medium":
        S, msl, skew = 8, 29, 32.0
        mcfg = make_model_cfgs(seq_len=S)
        dcfg = data_config(seq_len=S, max_sent_length=msl)
        gcfg = gpt2_config(n_layer=24 if args.layers == 12 else args.layers, n_embd=1024, n_head=16, n_positions=512, n_ctx=512,
                           vocab_size=V)
        if args.batch == 64:
            args.batch = 32
    else:
        S, skew = 5, None
        mcfg = make_model_cfgs(seq_len=S)
        dcfg = data_config(seq_len=S)
        gcfg = gpt2_config(n_layer=args.layers, vocab_size=V)          # GPT-2 base (zh vocab), pdrop 0.1 x3
        if os.environ.get("MMTG_BENCH_PDROP"):                         # measurement switch (what dropout costs); not the benchmark
            pd = float(os.environ["MMTG_BENCH_PDROP"])
            gcfg.update(embd_pdrop=pd, attn_pdrop=pd, resid_pdrop=pd)
    torch.manual_seed(0)                                           # identical replicas on every rank
    model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype=args.dtype,
                 token_table=synth.make_token_table(V, seed=2))
    model.reset_parameters(seed=0)
    model.to(dev).train()
    ddp = world > 1 or force_ddp
    trainer = MMTGTrainer(model, lr=1e-5, alpha=0.2, warmup_steps=10, total_steps=100000,
                          distributed=ddp, bucket_mb=args.bucket_mb)
    B = args.batch
    stage = 3 if skew is None else 2
    batches = []
    for i in range(2):
        nb = synth.make_batch(B, mcfg, dcfg, V, seed=1000 * rank + i, low_to_high=skew)
        if skew is not None:        # the stage-2 filter is part of the step; keep every row in (ratings 1-2 / 4-5)
            nb["rating"] = np.where(np.asarray(nb["rating"]) == 3, 2, nb["rating"])
        batches.append({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()})
        if skew is not None:        # ratings on the host too: the in-step filter then needs no device read-back
            batches[-1]["rating_host"] = torch.from_numpy(np.asarray(nb["rating"]))
    T = dcfg.topic_prompt_length + batches[0]["targets"].shape[1]

    # did-work evidence, outside the timed region: MyLoss of a held-out probe batch (eval mode, no dropout) before the
    # first and after the last optimizer step
    probe_nb = synth.make_batch(min(B, 16), mcfg, dcfg, V, seed=99 + rank)
    probe = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in probe_nb.items()}

    def probe_loss():
        eng = model.engine()
        eng.forward(probe, train_flag=True, training=False, logits_f32=False)
        sc = eng.loss(probe["rating"], 3)
        return float(sc[0].item())

    loss_first = None if args.no_check else probe_loss()
    p0 = model._flat.detach().clone()

    def run(n):
        for i in range(n):
            trainer.step(batches[i % 2], stage=stage)

    gpu_rewarm(dev)              # a fresh box's GPU has been idle: W short warm-up steps alone may still run at ramping clocks
    run(args.warmup)
    tune_steps = 0
    while getattr(trainer, "_tune", None) is not None and tune_steps < 32:
        run(1)                  # (data-parallel runs: the CU-reservation tuning finishes inside the untimed warm-up)
        tune_steps += 1
    el = _timed(lambda: run(args.steps), world, dev)
    ms_step = 1e3 * el / args.steps
    tokens = B * world * T
    value = tokens * args.steps / el

    roof = None
    launches_per_step = None
    if not args.no_roofline:
        # (the same instrumented steps once more in the product schedule first: what the overlapped launches' events read there)
        hip.prof_enable(True)
        el3 = _timed(lambda: run(args.steps), world, dev)
        hip.prof_enable(False)
        gp = hip.prof_read()["gemm_f32" if args.dtype == "f32" else "gemm_bf16"]
        with _common.launches_unshared():
            run(1)
            hip.prof_enable(True)
            el2 = _timed(lambda: run(args.steps), world, dev)
            hip.prof_enable(False)
        prof = hip.prof_read()
        launches_per_step = sum(v["launches"] for v in prof.values()) // args.steps
        g = prof["gemm_f32" if args.dtype == "f32" else "gemm_bf16"]
        # (bf16x3: three bf16 passes per product -- algorithmic FLOPs priced against a third of the dense bf16 peak)
        peak = 2500.0 if args.dtype == "bf16" else 157.3 if args.dtype == "f32" else 2500.0 / 3.0
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        roof = {"bound": "mfma",
            "kernel": ("gemm_p8_kernel / gemm_occ4_kernel / gemm_dma_kernel <bf16> (all instantiations)" if args.dtype == "bf16"
                       else "gemm_kernel<f32>" if args.dtype == "f32" else "gemm_p8_kernel<X3> / wgrad_group_kernel<X3>"),
            "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "launches_per_step": g["launches"] // args.steps,
                "avg_launch_us": round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                "ms_per_step_instrumented": round(1e3 * el2 / args.steps, 3),
                "per_category_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]},
                "frac_in_product_schedule": round((gp["flops"] / (gp["ms"] * 1e-3) / 1e12 if gp["ms"] > 0 else 0.0) / peak, 4),
                "ms_per_step_instrumented_product_schedule": round(1e3 * el3 / args.steps, 3),
                "measured": _common.UNSHARED_NOTE}
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the figure is
        # the committed rocprofv3 --pmc measurement of this same program (tools/gpu_pmc_bench.sh) -- accepted only when
        # it was taken on the kernel sources this library was built from; algorithmic bytes (A + B + C once) beside it.
        roof["algorithmic_bytes_per_launch"] = round(g["bytes"] / max(1, g["launches"]))
        if args.dtype == "bf16" and B == 64 and args.layers == 12 and args.config == "base":
            m, src = _pmc_traffic(hip.source_sha())
            if m is not None:
                roof["traffic"] = m["hbm_bytes_per_launch"]
                roof["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch; kernel sources sha %s)" % hip.source_sha()[:12]
            else:
                roof["traffic_source"] = "none: no profiles/r*_bench_pmc_gemm_traffic.json was taken on kernel sources sha %s" % hip.source_sha()[:12]
        if args.config == "base":
            fwd_tok, enc = algorithmic_flops_per_token(S, T, gcfg["n_embd"], gcfg["n_layer"], V)
            step_flops = 3.0 * (fwd_tok * B * T + enc * B)
            roof["whole_step_tflops_per_gpu"] = round(step_flops / (ms_step * 1e-3) / 1e12, 2)

    model.eval()
    loss_last = None if args.no_check else probe_loss()
    moved = float((model._flat.detach() - p0).abs().max().item())
    check = {"probe_myloss_before": loss_first if loss_first is None else round(loss_first, 6),
             "probe_myloss_after": loss_last if loss_last is None else round(loss_last, 6),
             "optimizer_steps": trainer.sched_step, "params_finite": bool(torch.isfinite(model._flat).all().item()),
             "max_param_change": moved, "launches_per_step": launches_per_step,
             "note": "probe = MyLoss (stage 3) of a held-out synthetic batch in eval mode before the first and after the last "
                     "optimizer step of this process (lr warms up from 0 over 10 steps to 1e-5)"}

    conditioning = None
    if rank == 0 and not args.no_roofline and args.dtype == "bf16" and args.config == "base":
        conditioning = conditioning_probe(model, batches[0])
    ddp_info = None
    if ddp:
        # exposed exchange time: a few more steps with HIP events around the reducer's finish() (outside the timed region)
        trainer.measure_finish = True
        trainer.reducer.measure = True
        run(max(3, min(args.steps, 10)))
        finish_wait = trainer.finish_wait_ms()
        bucket_timeline = trainer.reducer.timeline_report()
        trainer.measure_finish = False
        trainer.reducer.measure = False
        from mmtg_amd.ddp import cu_budget_setting
        ddp_info = {"rccl_world": dist.get_world_size(), "backend": dist.get_backend(),
                    "cu_budget": cu_budget_setting() if world > 1 else 0,
                    "finish_wait_ms_per_step": None if finish_wait is None else round(finish_wait, 3),
                    "budget_chosen": cu_budget_setting() if world > 1 else 0, "budget_tuning": trainer.budget_report,
                    "extra_warmup_steps_for_tuning": tune_steps,
                    "buckets": len(trainer.reducer.buckets), "bucket_mb": args.bucket_mb,
                    "bucket_sizes_mb": [round(4 * (e - s_) / 2 ** 20, 1) for s_, e in trainer.reducer.buckets],
                    "tail_bucket_mb": round(trainer.reducer.tail_bytes() / 2 ** 20, 1),
                    "gradient_bytes": int(trainer.eng.layout.total * 4),
                    "exchange_dtype": str(trainer.reducer.xdtype).replace("torch.", ""),
                    # "abi" (MMTG_DDP_COMM=abi): buckets through libmmtg_hip's own RCCL communicator (mmtg_allreduce_bucket_async)
                    "comm": "abi" if trainer.reducer.abi else "torch", "comm_info": trainer.reducer.comm_info,
                    # first-contact instrumentation: per bucket, when the backward handed it to RCCL (ms after the first launch) and
                    # how long the compute stream sat in its wait inside finish() -- the part of that all-reduce nothing hid
                    "bucket_timeline": bucket_timeline,
                    "allreduce_ms_per_step_isolated": round(allreduce_probe(trainer, max(3, min(args.steps, 10)), world, dev), 3),
                    "note": "allreduce_ms_per_step_isolated = the step's bucketed SUM all-reduces (+ the row count) alone, nothing to "
                            "overlap with; inside the step they run on RCCL's stream beside the backward; finish_wait_ms_per_step = how "
                            "long the compute stream waited for them after the backward (HIP events around GradReducer.finish, this rank): "
                            "the exposed part of the exchange; cu_budget = CUs the GEMM tile rule leaves to (< 0) the RCCL kernels -- budget_tuning: the trainer "
                            "timed 3 steps under each of 0 / -16 / -32 during the warm-up and all ranks agreed on the fastest (MAX over ranks, "
                            "one all-reduce; MMTG_DDP_GEMM_CUS pins it instead); tail_bucket_mb = what can only leave after the backward's last "
                            "kernel (buckets end after wpe and after the fuser, so the tied embedding / projector leave before the encoder's backward)"}
        if one_gpu_backend:
            ddp_info["rehearsal"] = ("MMTG_BENCH_ONE_GPU_BACKEND=%s: all %d ranks share cuda:0 and exchange through the host -- a run of "
                                     "the N > 1 code path on a one-GPU box, NOT a throughput measurement" % (one_gpu_backend, world))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "base":
        cpu = cpu_baseline(mcfg, dcfg, gcfg, V, T)

    decode = None
    if not args.no_decode and args.config == "base" and args.layers == 12:
        del trainer
        model = None
        torch.cuda.empty_cache()
        decode = bench_decode(args, world, rank, dev, steps=5, warmup=2, cpu_fn=cpu_decode_baseline)
    # the optional objects below run AFTER the primary measurement and must never cost it: a failure becomes {"error": ...}
    def guarded(fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as e:      # noqa: BLE001 -- reported in the line, the primary numbers above are already taken
            torch.cuda.empty_cache()
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}

    f32 = x3 = x3f = None
    extras = rank == 0 and world == 1 and not ddp and args.config == "base" and args.layers == 12 and args.dtype == "bf16"
    if extras and not args.no_x3:
        trainer = model = None
        torch.cuda.empty_cache()
        x3 = guarded(f32_object, args, dev, mcfg, dcfg, gcfg, V, mode="bf16x3")
        torch.cuda.empty_cache()
        x3f = guarded(f32_object, args, dev, mcfg, dcfg, gcfg, V, mode="bf16x3f")
    if extras and not args.no_f32:
        trainer = model = None
        torch.cuda.empty_cache()
        f32 = guarded(f32_object, args, dev, mcfg, dcfg, gcfg, V)
    medium = None
    if extras and not args.no_medium:
        trainer = model = None
        torch.cuda.empty_cache()
        medium = guarded(medium_object, args, dev)

    if rank == 0:
        if args.config == "medium":
            workload = ("Scaled stress (BASELINE configs[4]): GPT-2-medium %dL/1024/16H V=%d, S=8 experience steps, T=15+497=512 "
                        "decoder positions, rating skew K=32 (low:high), curriculum stage 2 filter inside the step, dropout 0.1 on, "
                        "MyLoss + 0.2*KL, clip 1.0, AdamW" % (gcfg["n_layer"], V))
        else:
            workload = ("Full MMTG train step: S=5 experience steps, T=15+221=236 decoder positions, "
                        "GPT-2 %dL/768/12H V=%d, dropout 0.1 on, MyLoss stage 3 + 0.2*KL, clip 1.0, AdamW" % (args.layers, V))
        par = ("single GPU, no collective" if not ddp else
               "dp%d (1 process/GPU, RCCL bucketed all-reduce of the flat fp32 gradient overlapped with backward%s)"
               % (world, ", forced at world 1" if world == 1 else ""))
        out = {
            "metric": "train tokens/sec, full MMTG config (GPT-2-base-zh decoder, 5x(img+text) 2048-d WenLan embs)"
                      if args.config == "base" else "train tokens/sec, scaled stress config (GPT-2-medium decoder, 8 experience steps)",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (random-init weights, random 2048-d embeddings / token ids of the released shape)",
            "config": {"workload": workload, "rows_per_gpu": B, "global_rows": B * world, "seq_len": T, "parallelism": par},
            "roofline": roof, "cpu_baseline": cpu, "check": check,
        }
        if conditioning is not None:
            out["conditioning"] = conditioning
        if ddp_info is not None:
            out["ddp"] = ddp_info
        if decode is not None:
            out["decode"] = decode
        if x3 is not None:
            out["bf16x3"] = x3
        if x3f is not None:
            out["bf16x3f"] = x3f
        if f32 is not None:
            out["f32"] = f32
        if x3 is not None or f32 is not None:
            out["parity_modes"] = {
                "modes_timed_in_this_run": [m for m, o in (("bf16x3", x3), ("f32", f32)) if o is not None and "error" not in o],
                "gates_asserted_by": "tests/test_model_gpu.py PARITY_MODES + tests/test_decode_gpu.py (pytest -m gpu), not by this run",
                "bounded_parity_only": ["bf16"],
                "note": "north_star's numeric gates (greedy-decode ids bit-exact, logits within 1e-3 of the reference) are TEST results on the "
                        "reference-generated goldens for the modes f32 and bf16x3 -- this line only times those modes (objects above: the "
                        "parity-qualified train / decode throughput); the headline `value` and `decode` are the bf16 mode BASELINE "
                        "configs[1] names (tests: logits within 0.15, ids equal wherever the reference's top-2 margin exceeds 0.27)"}
        if medium is not None:
            out["medium"] = medium
        # the parity-qualified and decode figures as SCALARS inside the two objects the driver keeps verbatim (`roofline`, `config`)
        summ = {}
        if isinstance(decode, dict) and "value" in decode:
            dr = decode.get("roofline") or {}
            summ.update(decode_tokens_per_s=decode["value"], decode_us_per_token_step=decode["config"]["us_per_token_step"],
                        decode_frac=dr.get("frac"),
                        decode_traffic_ratio=(round(dr["traffic"] / dr["algorithmic_bytes_per_token_step"], 3)
                                              if dr.get("traffic") and dr.get("algorithmic_bytes_per_token_step") else None))
        if isinstance(x3f, dict) and "train" in x3f:
            summ.update(forward_parity_dtype="bf16x3f", forward_parity_train_tokens_per_s=x3f["train"]["value"],
                        forward_parity_train_ms_per_step=x3f["train"]["ms_per_step"])
        if isinstance(x3, dict) and "train" in x3:
            summ.update(parity_dtype="bf16x3", parity_train_tokens_per_s=x3["train"]["value"],
                parity_train_ms_per_step=x3["train"]["ms_per_step"],
                        parity_decode_tokens_per_s=x3.get("decode", {}).get("value"),
                        parity_decode_us_per_token_step=x3.get("decode", {}).get("us_per_token_step"))
        if isinstance(conditioning, dict) and isinstance(conditioning.get("unfused_f32_configs1"), dict):
            summ["conditioning_unfused_f32_frac_hbm"] = conditioning["unfused_f32_configs1"].get("frac_hbm")
            summ["conditioning_fused_bf16_frac_mfma"] = conditioning.get("frac_mfma")
        if summ:
            out["config"].update(summ)
            if isinstance(out.get("roofline"), dict):
                out["roofline"].update(summ)
        _emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
