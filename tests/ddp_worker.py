"""Child process of tests/test_ddp_gpu.py: joins an RCCL ("nccl") process group BEFORE touching the GPU in any other
way, then runs the fused trainer over the bucketed all-reduce path and writes what the parent asserts on.

    python tests/ddp_worker.py <mode> <out.pt>        (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env)

modes
  world1   one rank, MMTG_FORCE_DDP=1: every bucket and the row count go through RCCL; the same two steps (stage-1
           filter, dropout on) are then run by a non-distributed trainer from the same state.
  world1_abi  one rank, MMTG_FORCE_DDP=1 MMTG_DDP_COMM=abi: the exchange through libmmtg_hip's own RCCL communicator.
  shards   WORLD_SIZE ranks, each a contiguous shard of one global batch (stage-1 filter applied per shard, so the
           shards are unequal); rank 0 also runs the whole batch alone.  Compares the reduced gradient.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(dtype, pdrop, dev, seed=100):
    from mmtg_amd import MMTG, synth
    from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
    S, V = 5, 160
    mcfg, dcfg = make_model_cfgs(seq_len=S), data_config(seq_len=S)
    gcfg = gpt2_config(n_layer=2, vocab_size=V, n_positions=256, embd_pdrop=pdrop, attn_pdrop=pdrop, resid_pdrop=pdrop)
    weights = synth.make_weights(mcfg, gcfg, seed=seed)
    model = MMTG(mcfg, dcfg, V, train_flag=True, gpt2_config=gcfg, token_table=synth.make_token_table(V, seed=seed + 1),
                 compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model.to(dev).train()
    return model, mcfg, dcfg, V


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    # MMTG_DDP_TEST_BACKEND=gloo + LOCAL_RANK=0 for every rank: several ranks SHARE one GPU and exchange through the host (RCCL
    # refuses two ranks on one device) -- the same trainer, reducer and kernels, so the N > 1 arithmetic runs on a single-GPU box
    backend = os.environ.get("MMTG_DDP_TEST_BACKEND", "nccl")
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # first GPU touch of this process
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(dev)
    from mmtg_amd import synth
    from mmtg_amd.trainer import MMTGTrainer
    res = {"rank": rank, "world": world, "backend": dist.get_backend()}
    try:
        if mode == "world1":
            assert world == 1 and os.environ.get("MMTG_FORCE_DDP")
            for dtype in ("bf16", "f32"):
                models = []
                for distributed in (True, False):
                    model, mcfg, dcfg, V = build(dtype, 0.1, dev)
                    tr = MMTGTrainer(model, lr=1e-3, alpha=0.2, distributed=distributed, bucket_mb=8.0)
                    if distributed:
                        assert tr.reducer is not None and tr.reducer.active and len(tr.reducer.buckets) > 3
                    tr.eng.drop_seed = 4242                     # same dropout masks in both runs
                    models.append((model, tr))
                nb = synth.make_batch(12, mcfg, dcfg, V, seed=7)
                batch = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()}
                flats = {True: [], False: []}
                launched = 0
                for step in range(2):
                    for (model, tr), distributed in zip(models, (True, False)):
                        if distributed:
                            n_all = []
                            orig = dist.all_reduce

                            def counting(t, *a, **k):
                                n_all.append(t.numel())
                                return orig(t, *a, **k)

                            dist.all_reduce = counting
                            try:
                                o = tr.step(batch, stage=1)
                            finally:
                                dist.all_reduce = orig
                            launched = len(n_all)
                            res["%s_allreduce_elems" % dtype] = n_all
                        else:
                            o = tr.step(batch, stage=1)
                        torch.cuda.synchronize()
                        flats[distributed].append((model._flat.detach().cpu().clone(), float(o["loss"]), float(tr.grad_norm()),
                                                   tr.eng.grad.detach().cpu().clone()))
                res[dtype] = flats
                res["%s_launched" % dtype] = launched
                res["layout_total"] = models[0][0].layout.total
                res["%s_layout" % dtype] = {k: models[0][0].layout.entries[k][:1] + (models[0][0].layout.entries[k][2],)
                                            for k in models[0][0].layout.entries}
        elif mode == "world1_bf16":
            # MMTG_DDP_GRAD_DTYPE=bf16 over RCCL at world 1 (forced): every bucket is cast to bf16, all-reduced by RCCL in bf16 and cast
            # back; the per-bucket timeline is collected (HIP events on the compute stream)
            assert world == 1 and os.environ.get("MMTG_FORCE_DDP") and os.environ.get("MMTG_DDP_GRAD_DTYPE") == "bf16"
            model, mcfg, dcfg, V = build("bf16", 0.0, dev)
            model.eval()
            nb = synth.make_batch(8, mcfg, dcfg, V, seed=11)
            batch = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()}
            tr = MMTGTrainer(model, lr=0.0, alpha=0.2, distributed=True, bucket_mb=8.0)
            assert tr.reducer.xdtype == torch.bfloat16 and tr.reducer.active
            tr.reducer.measure = True
            tr.step(batch, stage=3)
            tr.step(batch, stage=3)
            torch.cuda.synchronize()
            res["grad"] = tr.eng.grad.detach().cpu().clone()
            res["timeline"] = tr.reducer.timeline_report()
            plain, _, _, _ = build("bf16", 0.0, dev)
            plain.eval()
            tp = MMTGTrainer(plain, lr=0.0, alpha=0.2, distributed=False)
            tp.step(batch, stage=3)
            torch.cuda.synchronize()
            res["grad_plain"] = tp.eng.grad.detach().cpu().clone()
        elif mode == "world1_abi":
            # MMTG_DDP_COMM=abi at world 1 (forced): the buckets and the row count go through libmmtg_hip's own RCCL communicator
            # (csrc/comm.hip); torch.distributed -- gloo here -- is the control plane only and must see no all-reduce
            from mmtg_amd import hip
            assert world == 1 and os.environ.get("MMTG_FORCE_DDP") and os.environ.get("MMTG_DDP_COMM") == "abi"
            model, mcfg, dcfg, V = build("bf16", 0.0, dev)
            model.eval()
            nb = synth.make_batch(8, mcfg, dcfg, V, seed=11)
            batch = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()}
            tr = MMTGTrainer(model, lr=0.0, alpha=0.2, distributed=True, bucket_mb=8.0)
            assert tr.reducer.abi and tr.reducer.active and len(tr.reducer.buckets) > 3
            res["info"] = hip.comm_info()
            # the raw entry points: a SUM over one rank is the identity, in both storage types and both launch forms
            t = torch.randn(1 << 20, device=dev)
            t0 = t.clone()
            hip.allreduce_bucket(t)
            tb = torch.randn(1 << 20, device=dev).bfloat16()
            tb0 = tb.clone()
            hip.allreduce_bucket_async(tb)
            hip.comm_join()
            torch.cuda.synchronize()
            res["raw_identity"] = bool(torch.equal(t, t0) and torch.equal(tb, tb0))
            try:
                hip.comm_init(0, 1, hip.comm_unique_id())
                res["second_init_refused"] = False
            except RuntimeError as e:
                res["second_init_refused"] = "already holds a communicator" in str(e)
            try:
                hip.allreduce_bucket(torch.zeros(4, device=dev, dtype=torch.float16).view(torch.int16))
                res["bad_dtype_refused"] = False
            except (TypeError, RuntimeError):
                res["bad_dtype_refused"] = True
            n_torch = []
            orig = dist.all_reduce

            def counting(tt, *a, **k):
                n_torch.append(tt.numel())
                return orig(tt, *a, **k)

            dist.all_reduce = counting
            try:
                tr.step(batch, stage=3)                     # the product form: mmtg_allreduce_bucket_async + one join
                torch.cuda.synchronize()
                res["grad_async"] = tr.eng.grad.detach().cpu().clone()
                res["count"] = float(tr._count.item())
                tr.reducer.measure = True                   # the measured form: a host-owned side stream, one end event per bucket
                tr.step(batch, stage=3)
                tr.step(batch, stage=3)
                torch.cuda.synchronize()
            finally:
                dist.all_reduce = orig
            res["torch_allreduces"] = len(n_torch)
            res["grad_measured"] = tr.eng.grad.detach().cpu().clone()
            res["timeline"] = tr.reducer.timeline_report()
            plain, _, _, _ = build("bf16", 0.0, dev)
            plain.eval()
            tp = MMTGTrainer(plain, lr=0.0, alpha=0.2, distributed=False)
            tp.step(batch, stage=3)
            torch.cuda.synchronize()
            res["grad_plain"] = tp.eng.grad.detach().cpu().clone()
            res["count_plain"] = float(tp._count.item())
        elif mode == "shards":
            from mmtg_amd.ddp import shard_rows
            model, mcfg, dcfg, V = build("f32", 0.0, dev)
            model.eval()
            nb = synth.make_batch(16, mcfg, dcfg, V, seed=9, low_to_high=1.0)
            full = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in nb.items()}
            lo, hi = shard_rows(16, rank, world)
            mine = {k: v[lo:hi] for k, v in full.items()}
            tr = MMTGTrainer(model, lr=0.0, alpha=0.2, distributed=True, bucket_mb=8.0)
            o = tr.step(mine, stage=1)
            torch.cuda.synchronize()
            res["count"] = float(tr._count.item())
            res["grad"] = (tr.eng.grad / tr._count).cpu()
            res["n_local"] = int((mine["rating"] < 2).sum() + (mine["rating"] > 4).sum())
            if rank == 0:
                single, _, _, _ = build("f32", 0.0, dev)
                single.eval()
                ts = MMTGTrainer(single, lr=0.0, alpha=0.2, distributed=False)
                ts.step(full, stage=1)
                torch.cuda.synchronize()
                res["grad_single"] = (ts.eng.grad / ts._count).cpu()
                res["count_single"] = float(ts._count.item())
        else:
            raise SystemExit("unknown mode " + mode)
        res["ok"] = True
    finally:
        torch.save(res, out + ".r%d" % rank)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
