"""The data-parallel exchange (mmtg_amd.ddp, replacing nn.DataParallel of train.py:112-114) executed over RCCL on the
GPU.  Each test starts fresh child processes that join the "nccl" process group before any other GPU call."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, world, out, extra_env=None, timeout=600):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        if env.get("MMTG_DDP_TEST_BACKEND", "nccl") != "nccl":
            env["LOCAL_RANK"] = "0"          # every rank on the one GPU there is
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_worker.py"), mode, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-4000:]
    return [torch.load(out + ".r%d" % r, weights_only=False) for r in range(world)]


def test_forced_ddp_world1_equals_the_plain_trainer(tmp_path):
    """WORLD_SIZE=1, MMTG_FORCE_DDP=1: the gradient buckets and the row count really go through RCCL all-reduces
    (counted), and two optimizer steps (curriculum stage 1 filter, dropout on, identical mask seeds) leave the
    parameters where the non-distributed trainer leaves them.  A SUM over one rank is the identity and both paths
    run the same kernels in the same order, so every GRADIENT behind a deterministic chain (the GPT-2 block matrices:
    slab weight gradients) is BIT-equal after the first backward; gradients that end in fp32 atomics (LayerNorm
    columns, biases, embeddings, LM head: 26 tensors, tools/determinism_probe.py) differ in summation order between any
    two runs -- and through the clip coefficient (a function of the global norm) they reach every parameter's update
    in its last bits -- bounded here at 1e-6 of the parameter norm after one step and 2e-5 (f32) / 1e-4 (bf16: weight
    copies re-rounded) after two."""
    res = _launch("world1", 1, str(tmp_path / "w1"), {"MMTG_FORCE_DDP": "1"})[0]
    assert res.get("ok") and res["backend"] == "nccl"
    for dtype in ("bf16", "f32"):
        n_all = res["%s_allreduce_elems" % dtype]
        assert n_all[0] == 1 and len(n_all) >= 5 and sum(n_all[1:]) == res["layout_total"]      # count + every gradient element
        ddp, plain = res[dtype][True], res[dtype][False]
        for step, tol in ((0, 1e-6), (1, 1e-4 if dtype == "bf16" else 2e-5)):
            a, b = ddp[step][0], plain[step][0]
            rel = float((a - b).norm() / b.norm())
            assert rel < tol, (dtype, step, rel)
            assert abs(ddp[step][1] - plain[step][1]) <= (1e-6 if step == 0 else 1e-4) * abs(plain[step][1]), (dtype, step)
            assert abs(ddp[step][2] - plain[step][2]) <= (1e-4 if step == 0 else 2e-3) * abs(plain[step][2]), (dtype, step)
        if dtype == "bf16":
            lay = res["bf16_layout"]
            n = 0
            for key, (off, numel) in lay.items():
                if ".h." in key and key.endswith(".weight") and ".ln_" not in key:
                    assert torch.equal(ddp[0][3][off:off + numel], plain[0][3][off:off + numel]), key
                    assert float(ddp[0][3][off:off + numel].abs().max()) > 0
                    n += 1
            assert n == 8
        # the step really moved the parameters
        assert float((ddp[0][0] - ddp[1][0]).abs().max()) > 1e-4


def test_bf16_gradient_exchange_over_rccl_world1_and_bucket_timeline(tmp_path):
    """MMTG_DDP_GRAD_DTYPE=bf16 (round 6, opt-in) executed over RCCL (world 1, forced): every bucket crosses as bf16 -- cast, SUM
    all-reduce, cast back -- so the exchanged gradient equals the plain trainer's rounded to bf16 once (a SUM over one rank is the
    identity), element for element; the per-bucket timeline of the exchange (when the backward handed each bucket over, how long the
    compute stream waited for it) is reported for every bucket."""
    res = _launch("world1_bf16", 1, str(tmp_path / "w1b"), {"MMTG_FORCE_DDP": "1", "MMTG_DDP_GRAD_DTYPE": "bf16"})[0]
    assert res.get("ok") and res["backend"] == "nccl"
    g, ref = res["grad"], res["grad_plain"]
    assert torch.equal(g, ref.bfloat16().float())
    assert float((g - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
    tl = res["timeline"]
    assert tl and tl["exchange_dtype"] == "bfloat16" and tl["steps"] == 2
    nb = len(tl["bucket_mb"])
    assert nb > 3 and len(tl["launch_ms_after_first"]) == nb and len(tl["exposed_wait_ms"]) == nb
    assert tl["launch_ms_after_first"][0] == 0.0 and all(x >= 0.0 for x in tl["launch_ms_after_first"] + tl["exposed_wait_ms"])
    assert tl["launch_ms_after_first"] == sorted(tl["launch_ms_after_first"])          # buckets leave in gradient-ready order


def test_exchange_through_the_library_s_own_rccl_communicator_world1(tmp_path):
    """MMTG_DDP_COMM=abi (SURVEY.md section 8b's second small ABI, csrc/comm.hip) executed on the GPU at world 1 (forced), with
    torch.distributed on gloo as the control plane only: the raw entry points are the identity over one rank (fp32 and bf16, the
    explicit-stream and the fork / join forms), a second communicator in the process is refused, torch.distributed sees no
    all-reduce, and the trainer's gradient -- buckets launched from the backward with one C call each, joined once before the
    optimizer -- equals the plain trainer's bit for bit, in the product form and in the measured one (per-bucket end events)."""
    res = _launch("world1_abi", 1, str(tmp_path / "w1a"),
                  {"MMTG_FORCE_DDP": "1", "MMTG_DDP_COMM": "abi", "MMTG_DDP_TEST_BACKEND": "gloo"})[0]
    assert res.get("ok") and res["backend"] == "gloo"
    info = res["info"]
    assert info["rank"] == 0 and info["world"] == 1 and info["device"] == 0 and info["rccl_version"] > 20000
    assert res["raw_identity"] and res["second_init_refused"] and res["bad_dtype_refused"]
    assert res["torch_allreduces"] == 0
    assert res["count"] == res["count_plain"] > 0
    assert float(res["grad_plain"].abs().max()) > 0
    assert torch.equal(res["grad_async"], res["grad_plain"])
    assert torch.equal(res["grad_measured"], res["grad_plain"])
    tl = res["timeline"]
    nb = len(tl["bucket_mb"])
    assert tl["steps"] == 2 and nb > 3 and len(tl["exposed_wait_ms"]) == nb
    assert tl["launch_ms_after_first"] == sorted(tl["launch_ms_after_first"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_rank_gradient_equals_single_rank_on_the_concatenated_batch(tmp_path):
    """2 ranks, contiguous row shards of one 16-row batch, stage-1 filter per shard (unequal shards): the all-reduced
    gradient divided by the all-reduced row count equals the single-process gradient of the whole batch."""
    res = _launch("shards", 2, str(tmp_path / "w2"))
    assert all(r.get("ok") for r in res)
    assert res[0]["count"] == res[1]["count"] == res[0]["count_single"] == res[0]["n_local"] + res[1]["n_local"]
    for r in res:
        g, ref = r["grad"], res[0]["grad_single"]
        assert float((g - ref).norm() / ref.norm()) < 1e-4
    assert torch.equal(res[0]["grad"], res[1]["grad"])


def test_two_rank_gradient_equals_single_rank_two_processes_on_one_gpu(tmp_path):
    """The 2-rank arithmetic on a ONE-GPU box: two processes share cuda:0 and all-reduce through the host (gloo; RCCL refuses
    two ranks on one device).  Same trainer, same GradReducer (buckets in gradient-ready order, the device-scalar row count),
    same kernels as the RCCL run: contiguous row shards of one 16-row batch, stage-1 filter per shard (unequal shards) -- the
    all-reduced gradient divided by the all-reduced row count equals the single-process gradient of the whole batch, and both
    ranks hold the same bits."""
    res = _launch("shards", 2, str(tmp_path / "w2g"), {"MMTG_DDP_TEST_BACKEND": "gloo"})
    assert all(r.get("ok") for r in res) and res[0]["backend"] == "gloo"
    assert res[0]["count"] == res[1]["count"] == res[0]["count_single"] == res[0]["n_local"] + res[1]["n_local"]
    assert res[0]["n_local"] != res[1]["n_local"]
    for r in res:
        g, ref = r["grad"], res[0]["grad_single"]
        assert float((g - ref).norm() / ref.norm()) < 1e-4
    assert torch.equal(res[0]["grad"], res[1]["grad"])
