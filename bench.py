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
                the same train step on this box's host cores on a bounded sample (B=4)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def algorithmic_flops_per_token(S, T, D, L, V, H=512, E=2048):
    """forward FLOPs per decoder token (causal-half attention), x3 for training."""
    layer = 2 * (D * 3 * D + D * D + 2 * D * 4 * D)
    attn = 2 * T * D          # QK^T + PV over the causal half
    head = 2 * D * V
    proj = 2 * (E * H + H * D)
    fwd = L * (layer + attn) + head + proj
    enc_per_sample = 2 * (E * H) + 2 * S * 2 * (E * 3 * H + H * 3 * H) + 2 * S * 2 * (H * 3 * H) + 2 * S * H * E
    return fwd, enc_per_sample


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


def cpu_baseline(mcfg, dcfg, gcfg, V, T, seconds_budget=25.0):
    """Oracle train step (fwd + MyLoss + bwd + clip + AdamW) on the host cores, B=4."""
    from mmtg_amd import synth
    from oracle import mmtg_oracle as O
    threads = _host_threads()
    torch.set_num_threads(threads)
    B = 4
    weights = synth.make_weights(mcfg, gcfg, seed=1)
    table = torch.from_numpy(synth.make_token_table(V, seed=2))
    batch = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_batch(B, mcfg, dcfg, V, seed=3).items()}
    sh = O.Shapes(mcfg, dcfg, gcfg)
    w = O.weights_to_torch(weights, requires_grad=True)
    state = {}
    O.train_step(w, sh, table, batch, batch["rating"], 3, 0.2, 1e-5, 1, state)   # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        O.train_step(w, sh, table, batch, batch["rating"], 3, 0.2, 1e-5, n + 2, state)
        n += 1
        el = time.perf_counter() - t0
        if n >= 3 or el > seconds_budget:
            break
    return {"value": round(B * T * n / el, 2), "unit": "tokens/s", "cores": threads, "kind": "port", "host_logical_cpus": os.cpu_count(),
            "sample": "oracle (CPU PyTorch fp32 restatement) full config 12L/768/V=%d, B=%d x T=%d, %d train steps "
                      "(fwd+MyLoss+bwd+clip+AdamW), dropout off" % (V, B, T, n)}


def cpu_decode_baseline(mcfg, dcfg, gcfg, V, positions=24):
    """Oracle greedy decoding on the host cores exactly as the reference does it (generate.py:117-142: no KV
    cache, the whole prefix is re-run for every token), one prompt, a bounded number of positions."""
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

    def fwd(inputs):
        return O.mmtg_forward(w, sh, table, inputs, train_flag=False)[2]

    O.sample_sequence(fwd, start, 2, temperature=1.1, top_k=1, top_p=0.0, repitition_penalty=1.5, greedy=True)   # warm-up
    t0 = time.perf_counter()
    O.sample_sequence(fwd, start, positions, temperature=1.1, top_k=1, top_p=0.0, repitition_penalty=1.5, greedy=True)
    el = time.perf_counter() - t0
    return {"value": round(positions / el, 2), "unit": "tokens/s", "cores": threads, "kind": "port", "host_logical_cpus": os.cpu_count(),
            "sample": "oracle (CPU PyTorch fp32 restatement) greedy decoding as the reference runs it (no KV cache, prefix "
                      "re-run per token), batch 1, %d positions after the 15-token prompt, full 12L/768/V=%d" % (positions, V)}


def bench_decode(args, world, rank, dev):
    """Greedy decode tokens/s: every rank decodes its own batch (replicas only, no exchange)."""
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
        return
    dec = GreedyDecoder(model, max_batch=B, max_len=Ln)
    for _ in range(max(1, args.warmup)):
        dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = dec.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    roof = cpu = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel of a decode step = the small-M split-K products streaming the weights (HBM bound): live
        # HIP-event time and algorithmic bytes (weights + activations + fp32 slabs once) of every GEMM launch of
        # one eager (un-captured) generation, as the library's profiling hooks count them
        deg = GreedyDecoder(model, max_batch=B, max_len=Ln, use_graph=False)
        deg.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
        hip.prof_enable(True)
        deg.generate(batch, Ln, temperature=1.1, repitition_penalty=1.5)
        hip.prof_enable(False)
        pr = hip.prof_read()
        g = pr["gemm_bf16" if args.dtype == "bf16" else "gemm_f32"]
        ach = g["bytes"] / max(g["ms"], 1e-9) / 1e6          # GB/s
        roof = {"bound": "hbm", "kernel": "gemm_dma_kernel<256x32, 4-deep ring> (split-K weight streaming)",
                "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                "launches_per_generation": g["launches"], "avg_launch_us": round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                "algorithmic_bytes_per_launch": round(g["bytes"] / max(1, g["launches"])),
                "per_category_ms_per_generation": {k: round(v["ms"], 3) for k, v in pr.items() if v["launches"]},
                "note": "a graph node costs >= 4.1 us here (profiles/r01_v8_decode_rocprofv3_kernel_stats.csv): the step is "
                        "bound by ~110 dependent launches, not by HBM"}
        del deg
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_decode_baseline(mcfg, dcfg, gcfg, V)
    if rank == 0:
        steps_per_seq = dcfg.topic_prompt_length + Ln
        out = {"metric": "greedy-decode tokens/sec, full MMTG config", "value": round(B * world * Ln * args.steps / el, 1),
               "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "batched greedy generation, KV cache + hipGraph replay per token: batch %d, "
                                      "%d generated positions after a 15-token prompt, GPT-2 %dL/768/12H V=%d"
                                      % (B, Ln, args.layers, V),
                          "us_per_token_step": round(1e6 * el / args.steps / steps_per_seq, 2),
                          "parallelism": "replicas x%d" % world}}
        if roof is not None:
            out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="rows per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--mode", default="train", choices=["train", "decode"],
                    help="decode: batched greedy generation (BASELINE configs[3]: batch 256, max_len 128)")
    ap.add_argument("--decode-batch", type=int, default=256)
    ap.add_argument("--decode-len", type=int, default=128)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_ddp = bool(os.environ.get("MMTG_FORCE_DDP"))     # exercise the RCCL path on one GPU (self-test)
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mmtg_amd import MMTG, hip, synth
    from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
    from mmtg_amd.trainer import MMTGTrainer

    if args.mode == "decode":
        return bench_decode(args, world, rank, dev)

    S, V = 5, 13317
    mcfg = make_model_cfgs(seq_len=S)
    dcfg = data_config(seq_len=S)
    gcfg = gpt2_config(n_layer=args.layers, vocab_size=V)          # GPT-2 base (zh vocab), pdrop 0.1 x3
    torch.manual_seed(0)                                           # identical replicas on every rank
    model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, compute_dtype=args.dtype,
                 token_table=synth.make_token_table(V, seed=2))
    model.reset_parameters(seed=0)
    model.to(dev).train()
    trainer = MMTGTrainer(model, lr=1e-5, alpha=0.2, warmup_steps=10, total_steps=100000,
                          distributed=world > 1 or force_ddp, bucket_mb=args.bucket_mb)
    B = args.batch
    batches = []
    for i in range(2):
        nb = synth.make_batch(B, mcfg, dcfg, V, seed=1000 * rank + i)
        batches.append({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()})
    T = dcfg.topic_prompt_length + batches[0]["targets"].shape[1]

    def run(n):
        for i in range(n):
            trainer.step(batches[i % 2], stage=3)

    def timed(n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    run(args.warmup)
    el = timed(args.steps)
    ms_step = 1e3 * el / args.steps
    tokens = B * world * T
    value = tokens * args.steps / el

    roof = None
    if not args.no_roofline:
        hip.prof_enable(True)
        el2 = timed(args.steps)
        hip.prof_enable(False)
        prof = hip.prof_read()
        g = prof["gemm_bf16" if args.dtype == "bf16" else "gemm_f32"]
        peak = 2500.0 if args.dtype == "bf16" else 157.3
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        roof = {"bound": "mfma", "kernel": ("gemm_occ4_kernel / gemm_dma_kernel <bf16> (all instantiations)" if args.dtype == "bf16" else "gemm_kernel<f32>"), "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "launches_per_step": g["launches"] // args.steps,
                "avg_launch_us": round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                "ms_per_step_instrumented": round(1e3 * el2 / args.steps, 3),
                "per_category_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}}
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the
        # figure is the committed rocprofv3 --pmc measurement of this same program (tools/gpu_pmc_bench.sh),
        # reported only for the configuration it was taken on; algorithmic bytes (A + B + C once) beside it.
        roof["algorithmic_bytes_per_launch"] = round(g["bytes"] / max(1, g["launches"]))
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_v9_bench_pmc_gemm_traffic.json")
        if args.dtype == "bf16" and B == 64 and args.layers == 12 and os.path.exists(pmc):
            with open(pmc) as fh:
                m = json.load(fh)
            roof["traffic"] = m["hbm_bytes_per_launch"]
            roof["traffic_source"] = "profiles/r01_v9_bench_pmc_gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)"
        fwd_tok, enc = algorithmic_flops_per_token(S, T, gcfg["n_embd"], gcfg["n_layer"], V)
        step_flops = 3.0 * (fwd_tok * B * T + enc * B)
        roof["whole_step_tflops_per_gpu"] = round(step_flops / (ms_step * 1e-3) / 1e12, 2)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(mcfg, dcfg, gcfg, V, T)

    if rank == 0:
        out = {
            "metric": "train tokens/sec, full MMTG config (GPT-2-base-zh decoder, 5x(img+text) 2048-d WenLan embs)",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (random-init weights, random 2048-d embeddings / token ids of the released shape)",
            "config": {"workload": "Full MMTG train step: S=5 experience steps, T=15+221=236 decoder positions, "
                                   "GPT-2 %dL/768/12H V=%d, dropout 0.1 on, MyLoss stage 3 + 0.2*KL, clip 1.0, AdamW" % (args.layers, V),
                       "rows_per_gpu": B, "global_rows": B * world, "seq_len": T,
                       "parallelism": "dp%d (1 process/GPU, RCCL bucketed all-reduce)" % world},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
