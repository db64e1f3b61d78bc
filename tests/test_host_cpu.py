"""CPU-only checks of the host logic: state-dict contract, flat layout, C-ABI symbol
table, loud failure without a GPU, curriculum filter / LR schedule, and the N>1
gradient exchange on the gloo backend (world_size 2)."""
import ctypes
import os
import sys
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_case
from mmtg_amd import MMTG, hip
from mmtg_amd.configs import data_config, gpt2_config, make_model_cfgs
from mmtg_amd.ddp import GradReducer, shard_rows
from mmtg_amd.engine import ParamLayout
from mmtg_amd.trainer import curriculum_filter, linear_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_model(S=5, L=2, V=160, **enc):
    mcfg = make_model_cfgs(seq_len=S, **enc)
    gcfg = gpt2_config(n_layer=L, vocab_size=V, n_positions=256)
    return MMTG(mcfg, data_config(seq_len=S), V, gpt2_config=gcfg), mcfg, gcfg


@pytest.mark.parametrize("case", ["tiny_s5", "tiny_lstm2_rnn2", "tiny_gru2_lstm1"])
def test_state_dict_keys_match_reference(case):
    """named_parameters() of the reference (recorded in the fixtures) = this model's state dict, also for the LSTM / ReLU-RNN /
    multi-layer encoder channels of model.py:41-59 (torch.nn.RNNBase names: weight_ih_l{k}, weight_hh_l{k}, bias_*_l{k})."""
    fx, meta, mcfg, gcfg, dcfg, weights, table, batch = load_case(case)
    model = MMTG(mcfg, dcfg, meta["V"], gpt2_config=gcfg)
    ref_keys = sorted(str(k) for k in fx["grad_keys"])       # named_parameters() of the reference
    sd = model.state_dict()
    assert sorted(k for k in sd if k != "decoder.gpt2.lm_head.weight") == ref_keys
    assert sd["decoder.gpt2.lm_head.weight"].data_ptr() == sd["decoder.gpt2.transformer.wte.weight"].data_ptr()
    for k, v in weights.items():
        assert tuple(sd[k].shape) == v.shape, k
    n = sum(p.numel() for p in model.parameters())
    assert n == sum(v.size for k, v in weights.items() if k != "decoder.gpt2.lm_head.weight")


def test_encoder_variants_layout_and_validation():
    """Channel type / depth are validated like the reference's constructor would fail (unknown type: no rnns_* attribute); the
    flat layout orders a channel's layers top first (gradient-ready order) and torch's own modules agree on names and shapes."""
    mcfg = make_model_cfgs(seq_len=3, image_type="LSTM", image_layers=3, text_type="RNN", text_layers=2)
    lay = ParamLayout(mcfg, gpt2_config(n_layer=1, vocab_size=160, n_positions=128))
    for ch, mod in (("image", torch.nn.LSTM(2048, 512, num_layers=3)), ("text", torch.nn.RNN(2048, 512, num_layers=2, nonlinearity="relu"))):
        for k, v in mod.state_dict().items():
            assert lay.entries["encoder.rnns_%s.%s" % (ch, k)][1] == tuple(v.shape), (ch, k)
    img = [k for k in lay.keys if k.startswith("encoder.rnns_image.weight_ih")]
    assert img == ["encoder.rnns_image.weight_ih_l2", "encoder.rnns_image.weight_ih_l1", "encoder.rnns_image.weight_ih_l0"]
    with pytest.raises(ValueError):
        ParamLayout(make_model_cfgs(image_type="Transformer"), gpt2_config(n_layer=1))
    with pytest.raises(ValueError):
        ParamLayout(make_model_cfgs(text_layers=0), gpt2_config(n_layer=1))


def test_full_model_parameter_count():
    # SURVEY Appendix C probe 2: 109 064 709 parameters for the released configuration
    lay = ParamLayout(make_model_cfgs(), gpt2_config())
    assert sum(n for _, _, n in lay.entries.values()) == 109064709
    assert lay.Vpad == 13440 and lay.total % 64 == 0
    offs = sorted((o, n) for o, _, n in lay.entries.values())
    for (o1, n1), (o2, _) in zip(offs, offs[1:]):
        assert o1 + n1 <= o2
    assert all(o % 64 == 0 for name, (o, n) in lay.pack_range.items())
    bk = lay.buckets(16 * 1024 * 1024)
    assert bk[0][0] == 0 and bk[-1][1] == lay.total and all(a[1] == b[0] for a, b in zip(bk, bk[1:]))


@pytest.mark.parametrize("enc", [{}, dict(image_type="LSTM", image_layers=2, text_type="RNN", text_layers=2)])
def test_load_state_dict_variants_and_roundtrip(tmp_path, enc):
    model, mcfg, gcfg = tiny_model(**enc)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    if enc:
        assert sd["encoder.rnns_image.weight_ih_l1"].shape == (2048, 512) and sd["encoder.rnns_text.weight_hh_l1"].shape == (512, 512)
    other, _, _ = tiny_model(**enc)
    wrapped = {"module." + k: v for k, v in sd.items()}                      # saved from nn.DataParallel
    wrapped["module.decoder.gpt2.transformer.h.0.attn.bias"] = torch.ones(1, 1, 4, 4, dtype=torch.uint8)
    wrapped["module.decoder.gpt2.transformer.h.0.attn.masked_bias"] = torch.tensor(-1e4)
    other.load_state_dict(wrapped)
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k
    torch.save({"model": other.state_dict(), "args": None, "model_cfgs": mcfg}, tmp_path / "ckpt.pth")
    third, _, _ = tiny_model(**enc)
    third.load_state_dict(torch.load(tmp_path / "ckpt.pth")["model"])
    assert all(torch.equal(third.state_dict()[k], sd[k]) for k in sd)
    legacy = other.legacy_state_dict()
    assert legacy["decoder.gpt2.transformer.h.1.attn.bias"].shape == (1, 1, 256, 256)
    with pytest.raises(RuntimeError, match="unexpected"):
        third.load_state_dict({**sd, "bogus.weight": torch.zeros(1)})
    # wte pad rows of the flat buffer stay zero after loading
    off, n = third.layout.pack_range["wte"]
    V, D = 160, 768
    assert float(third._flat[off + V * D: off + n].abs().max()) == 0.0


def test_reference_compatible_checkpoint_loads_strictly_into_the_reference_key_set(tmp_path):
    """save_checkpoint(reference_compatible=True) against what the reference does with the file (generate.py:191-192):
    its model is wrapped in nn.DataParallel first and then takes a STRICT load_state_dict.  The skeleton below has the
    reference's own parameter names and shapes (the fixture's named_parameters() list, produced by running the
    reference) plus the persistent attn.bias / attn.masked_bias buffers transformers 4.12.3 registers per block."""
    from mmtg_amd.trainer import load_checkpoint, save_checkpoint
    fx, meta, mcfg, gcfg, dcfg, weights, table, batch = load_case("tiny_s5")
    model = MMTG(mcfg, dcfg, meta["V"], gpt2_config=gcfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})

    class Skel(torch.nn.Module):
        pass

    def attach(root, dotted, t, buffer=False):
        mod = root
        parts = dotted.split(".")
        for name in parts[:-1]:
            if name not in mod._modules:
                mod.add_module(name, Skel())
            mod = mod._modules[name]
        if buffer:
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], t)

    skel = Skel()
    params = {}
    for k in (str(k) for k in fx["grad_keys"]):
        params[k] = torch.nn.Parameter(torch.zeros(weights[k].shape))
        attach(skel, k, params[k])
    attach(skel, "decoder.gpt2.lm_head.weight", params["decoder.gpt2.transformer.wte.weight"])      # tied
    NP = gcfg["n_positions"]
    for l in range(gcfg["n_layer"]):
        attach(skel, "decoder.gpt2.transformer.h.%d.attn.bias" % l, torch.zeros(1, 1, NP, NP, dtype=torch.uint8), buffer=True)
        attach(skel, "decoder.gpt2.transformer.h.%d.attn.masked_bias" % l, torch.tensor(0.0), buffer=True)
    wrapped = torch.nn.DataParallel(skel)
    path = tmp_path / "ref_compat.pth"
    save_checkpoint(path, model, args={"lr": 1e-5}, model_cfgs=mcfg, reference_compatible=True)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) >= {"model", "args", "model_cfgs"}
    res = wrapped.load_state_dict(ckpt["model"], strict=True)          # raises on any missing / unexpected key
    assert not res.missing_keys and not res.unexpected_keys
    for k, p in params.items():
        assert torch.equal(p.detach(), torch.from_numpy(weights[k])), k
    b = skel.decoder.gpt2.transformer.h._modules["0"].attn.bias
    assert b.dtype == torch.uint8 and bool(b[0, 0, 5, 5]) and not bool(b[0, 0, 5, 6])
    # the plain file is NOT loadable there (the claim is made only for reference_compatible=True) ...
    save_checkpoint(tmp_path / "plain.pth", model)
    with pytest.raises(RuntimeError):
        wrapped.load_state_dict(torch.load(tmp_path / "plain.pth", weights_only=False)["model"], strict=True)
    # ... and both files load back here
    for f in ("ref_compat.pth", "plain.pth"):
        other = MMTG(mcfg, dcfg, meta["V"], gpt2_config=gcfg)
        load_checkpoint(tmp_path / f, other)
        assert all(torch.equal(other.state_dict()[k], model.state_dict()[k]) for k in model.state_dict())


def test_parameters_are_views_of_one_flat_buffer():
    model, _, _ = tiny_model()
    base = model._flat.data_ptr()
    for k, p in model._params.items():
        off = model.layout.entries[k][0]
        assert p.data_ptr() == base + 4 * off, k
    m2 = model.to(torch.float32)
    assert m2 is model
    with pytest.raises(TypeError):
        model.to(torch.float16)


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model, _, _ = tiny_model()
    with pytest.raises(RuntimeError, match="MI355X"):
        model({"img_embs": torch.zeros(1, 5, 2048)})


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mmtg_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mmtg_[a-z0-9_]+)\s*\(", header)))
    assert declared == hip.exported_symbols()
    lib = ctypes.CDLL(hip.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.lib().mmtg_abi_version() == hip.ABI_VERSION == 12
    # ... and nothing else: the library is built with -fvisibility=hidden + a linker version script, so no internal C++
    # helper (mmtg_set_error, ProfScope, template instantiations, hipcc's __hip_cuid_*) leaks into the dynamic symbol table
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", hip.lib_path()], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert sorted(names) == declared, sorted(set(names) - set(declared))


def test_host_side_of_the_round4_abi_without_a_gpu():
    """The parts of the ABI that are host arithmetic run without a GPU: workspace sizes of the ordered reductions, the build-flag
    string, and argument validation (a bad call is rejected with the library's error message before any device call)."""
    L = hip.lib()
    assert L.mmtg_colsum_ws(236, 3072) == 0 and L.mmtg_colsum_ws(15104, 512) == 128 * 512          # tall inputs go through 128 row slices
    assert L.mmtg_embed_add_bwd_ws(15104, 768, 11) == 118 * 11 * 768
    assert L.mmtg_beta_fuse_bwd_ws(64, 5, 512) == 64 * 5 * 513
    assert 0 < L.mmtg_sumsq_ws(109064709) <= 4096
    # the error contract without a GPU: argument checks run on the host before any launch
    assert L.mmtg_build_flags() == b""                                    # the product build carries no diagnostic defines
    # the batched column sums validate their host-side item list before any launch (round 6)
    it = (hip.ColsumItem * 1)(hip.ColsumItem(0x7000000000, 0x7100000000, 64, 4096, 64))
    assert L.mmtg_colsum_batch(ctypes.addressof(it), 1, None) != 0 and b"colsum_batch: item 0" in L.mmtg_last_error()      # > 2048 rows
    assert L.mmtg_colsum_batch(ctypes.addressof(it), 0, None) == 0                                                        # nothing to do
    assert L.mmtg_attn_bwd_dbias_rows(hip.BF16, 64, 236) == 64 and L.mmtg_attn_bwd_dbias_rows(hip.F32, 64, 236) == 0
    bad = L.mmtg_split_planes(0x7000000000, 8, 4, 12, 0x7100000000, 16, 64, None)
    assert bad != 0 and b"split_planes" in L.mmtg_last_error()           # cols % 8 != 0


def test_comm_abi_without_a_communicator_fails_loudly(monkeypatch):
    """The data-parallel exchange's second small ABI (csrc/comm.hip) on a box without a GPU: the library loads
    without RCCL as a link-time dependency, reports "no communicator", and every collective entry point refuses with a message --
    nothing falls back to a host reduction.  MMTG_DDP_COMM is validated; on CPU tensors the reducer keeps torch.distributed."""
    import subprocess
    from mmtg_amd import ddp
    needed = subprocess.run(["readelf", "-d", hip.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed.lower()
    assert hip.comm_info()["world"] == 0 and hip.comm_info()["rank"] == -1
    t = torch.zeros(8)
    with pytest.raises(RuntimeError, match="no communicator"):
        hip._check(hip.lib().mmtg_allreduce_bucket(t.data_ptr(), 8, hip.F32, None), "allreduce_bucket")
    with pytest.raises(RuntimeError, match="no communicator"):
        hip._check(hip.lib().mmtg_allreduce_bucket_async(t.data_ptr(), 8, hip.F32, None), "allreduce_bucket_async")
    with pytest.raises(RuntimeError, match="no communicator"):
        hip._check(hip.lib().mmtg_comm_join(None), "comm_join")
    with pytest.raises(RuntimeError, match="rank 3 of 2"):
        hip.comm_init(3, 2, bytes(hip.COMM_ID_BYTES))
    with pytest.raises(ValueError):
        hip.comm_init(0, 1, b"short")
    hip.comm_destroy()                                   # a no-op without a communicator
    assert float(t.abs().max()) == 0.0
    monkeypatch.setenv("MMTG_DDP_COMM", "mpi")
    with pytest.raises(ValueError, match="MMTG_DDP_COMM"):
        ddp.comm_backend()
    monkeypatch.setenv("MMTG_DDP_COMM", "abi")
    assert ddp.comm_backend() == "abi"
    if not torch.cuda.is_available():
        model, _, _ = tiny_model()
        red = ddp.GradReducer(model.layout, bucket_mb=1.0)
        assert red.abi is False and red.comm_info is None


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mmtg_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle|import_module\(.oracle|__import__\(.oracle", src, re.M), fn


def test_curriculum_filter_and_schedule():
    r = torch.tensor([1, 5, 3, 2, 4, 5, 1])
    assert curriculum_filter(r, 1).tolist() == [0, 6, 1, 5]
    assert curriculum_filter(r, 2).tolist() == [0, 3, 6, 1, 4, 5]
    assert curriculum_filter(r, 3).tolist() == list(range(7))
    assert linear_schedule(0, 10, 100) == 0.0 and linear_schedule(5, 10, 100) == 0.5
    assert abs(linear_schedule(55, 10, 100) - 0.5) < 1e-12 and linear_schedule(100, 10, 100) == 0.0
    assert [shard_rows(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


# ---------------------------------------------------------------- world_size-2 gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = ParamLayout(make_model_cfgs(seq_len=2), gpt2_config(n_layer=2, vocab_size=160, n_positions=128))
        red = GradReducer(lay, bucket_mb=4.0)
        g = torch.Generator().manual_seed(100 + rank)
        grad = torch.randn(lay.total, generator=g)
        mine = grad.clone()
        # emulate the backward: packs become final in layout order; fire the hook at the engine's points
        fire = ["ln_f.b"] + [f"decoder.gpt2.transformer.h.{l}.ln_1.bias" for l in (1, 0)] + ["wpe", "att_b", "encoder.topic_fc.bias"]
        launched = []
        for pk in fire:
            red.on_pack_ready(grad, pk)
            launched.append(red.next_bucket)
        assert launched == sorted(launched) and launched[0] <= 1 and launched[-1] >= len(red.buckets) - 1
        cnt = torch.tensor([3.0 + rank])
        red.start_count(cnt)            # the row count rides with the buckets (no host-side exchange of its own)
        red.finish(grad)
        n = int(cnt.item())
        # the tail of the backward: buckets end right after "wpe" and "att_b" (ddp.TAIL_SPLIT), so those runs of gradients leave
        # when their own backward is done and only the encoder's gradients are left for after the last kernel
        ends = {e for _, e in red.buckets}
        assert red.pack_end["wpe"] in ends and red.pack_end["att_b"] in ends
        assert 0 < red.tail_bytes() <= 4 * (lay.total - red.pack_end["att_b"])
        # the CU-reservation tuning: the ranks time the candidates DIFFERENTLY (rank 0 finds 0 fastest, rank 1 finds -32 fastest);
        # one MAX all-reduce later both hold the same choice -- the candidate whose slowest rank was fastest
        from mmtg_amd.ddp import agree_on_budget
        times = [10.0, 12.0, 15.0] if rank == 0 else [16.0, 12.5, 11.0]
        choice, agreed = agree_on_budget(times, (0, -16, -32))
        assert agreed == [16.0, 12.5, 15.0] and choice == -16
        tie, _ = agree_on_budget([5.0, 5.0, 5.0], (0, -16, -32))
        assert tie == 0                 # ties go to the first candidate on every rank
        # opt-in narrow exchange (MMTG_DDP_GRAD_DTYPE=bf16, round 6): cast -> SUM all-reduce in bf16 -> cast back into the fp32 buffer
        os.environ["MMTG_DDP_GRAD_DTYPE"] = "bf16"
        try:
            red16 = GradReducer(lay, bucket_mb=4.0)
        finally:
            del os.environ["MMTG_DDP_GRAD_DTYPE"]
        assert red16.xdtype == torch.bfloat16 and red.xdtype == torch.float32
        g16 = mine.clone()
        for pk in fire:
            red16.on_pack_ready(g16, pk)
        red16.finish(g16)
        assert g16.dtype == torch.float32 and not red16._pending and not red16.handles
        torch.save((rank, mine, grad, len(red.buckets), n, launched, g16), os.path.join(outdir, 'r%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [torch.load(tmp_path / ("r%d.pt" % r)) for r in range(world)]
    total = res[0][1] + res[1][1]
    # the bf16 exchange: each rank's share rounded to bf16, summed, the sum rounded to bf16 (what a bf16 all-reduce of two ranks does)
    total16 = (res[0][1].bfloat16().float() + res[1][1].bfloat16().float()).bfloat16().float()
    for rank, mine, reduced, nb, n, launched, g16 in res:
        assert nb > 3            # really bucketed
        assert n == 7            # 3 + 4 rows across ranks
        assert torch.equal(reduced, total)
        assert torch.equal(g16, total16)
        assert (g16 - total).abs().max() <= 2.0 ** -7 * total.abs().max()          # within bf16 rounding of the fp32 exchange


def test_postprocess_cut_rules_vs_reference_goldens():
    """generate.py:222-235 (10th-[#EOS#] / first-[SEP] cut, detokenise, comma join) -- expected strings were produced by
    executing the reference's own block on these token lists (tools/make_golden.py::case_postprocess)."""
    import json
    from mmtg_amd.generate import postprocess_tokens
    fx = np.load(os.path.join(ROOT, "tests", "golden", "postprocess.npz"))
    vocab = {int(k): v for k, v in json.loads(str(fx["vocab_json"])).items()}
    expected = json.loads(str(fx["expected_json"]))
    assert int(fx["n"]) == len(expected) >= 8
    for n, want in enumerate(expected):
        toks = [vocab[int(i)] for i in fx["ids_%d" % n]]
        assert postprocess_tokens(toks) == want, n
    assert postprocess_tokens(["[#START#]", "[#EOS#]", "[PAD]"]) == ""       # (the reference raises IndexError on this one)


def test_packed_token_table_round_trip(tmp_path):
    """pack_token_table: the reference's {id: list[2048]} pickle (model.py:215) -> [V, 2048] bf16 tensor file; the decoder
    picks it up from the path or as a tensor, ids missing from the dict stay zero."""
    import pickle
    from mmtg_amd.model import GPT2_Decoder, load_token_table, pack_token_table
    rng = np.random.default_rng(3)
    d = {i: rng.standard_normal(2048).astype(np.float32).tolist() for i in (0, 1, 2, 5, 9)}
    pk = tmp_path / "token_id2emb_dict.pkl"
    with open(pk, "wb") as f:
        pickle.dump(d, f)
    out = pack_token_table(str(pk), str(tmp_path / "t.safetensors"))
    t = load_token_table(out)
    assert t.dtype == torch.bfloat16 and tuple(t.shape) == (10, 2048)
    for i in d:
        assert torch.equal(t[i], torch.tensor(d[i]).to(torch.bfloat16))
    assert float(t[3].abs().max()) == 0.0
    dec = GPT2_Decoder(data_config(), token_table=out)
    assert dec._table.dtype == torch.bfloat16 and torch.equal(dec._table, t)
    dec2 = GPT2_Decoder(data_config(), token_table=d)            # the dict itself still works (float32 kept)
    assert dec2._table.dtype == torch.float32 and torch.equal(dec2._table.to(torch.bfloat16), t)
    with pytest.raises(ValueError):
        from safetensors.torch import save_file
        save_file({"x": torch.zeros(2)}, str(tmp_path / "bad.safetensors"))
        load_token_table(tmp_path / "bad.safetensors")


def _run_bench(argv, env=None, timeout=240):
    import subprocess
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus N` with no launcher (how the driver starts every N): the parent -- before any GPU call -- starts N
    fresh ranks with the torch.distributed.run environment contract; --dry-launch has them rendezvous over gloo instead of
    running the workload.  N children, distinct ranks and local ranks, ONE JSON line on stdout, rc 0."""
    import json
    r = _run_bench(["--gpus", "3", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_launch"] and d["world"] == 3 and d["n_gpus"] == 3 and d["self_launched"]
    assert sorted(x[0] for x in d["ranks"]) == [0, 1, 2] and sorted(x[1] for x in d["ranks"]) == [0, 1, 2]
    assert len({x[2] for x in d["ranks"]}) == 3                      # three different processes
    assert r.stderr.count("[bench dry-launch] rank") == 3


def test_bench_self_launch_reports_a_dead_rank():
    """A rank that dies takes the job down with its return code (the parent kills exactly the children it started once one
    has failed) instead of leaving the others waiting in the rendezvous."""
    r = _run_bench(["--gpus", "2", "--dry-launch"], env={"MMTG_DRY_FAIL_RANK": "1", "MMTG_BENCH_KILL_GRACE": "2"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip()]     # no JSON line from a failed job


def test_grouped_weight_gradient_split_rule_and_workspace_sizes():
    """Engine-side rules of the grouped weight-gradient launch (no GPU): one round of the kernel's slots, no K slice shorter than
    1024 tokens; workspace / counter sizes per configuration as include/mmtg_hip.h states them."""
    from mmtg_amd.engine import _group_splits
    base = ((768, 3072), (3072, 768), (768, 768), (768, 2304))
    tiles, ws, cnt = hip.wgrad_group_sizes(base, 2, 0)
    assert (tiles, ws, cnt) == (432, 432 * 2 * 16384, 432 * 4)
    assert _group_splits(tiles, 15104) == 2                      # 864 workgroups <= 1024 slots
    assert _group_splits(tiles, 1500) == 1                       # short batches: no slice below 1024 tokens
    medium = ((1024, 4096), (4096, 1024), (1024, 1024), (1024, 3072))
    assert hip.wgrad_group_sizes(medium, 1, 0)[0] == 768 and _group_splits(768, 16384) == 1
    t8, ws8, c8 = hip.wgrad_group_sizes(base, 2, 1)
    assert (t8, ws8, c8) == (108, 108 * 2 * 65536, 108 * 8)
    assert _group_splits(t8, 15104, 256) == 2


def test_split_precision_mode_host_contract():
    """bf16x3 = fp32 storage + split-precision products: the mode string maps to the fp32 storage code with the x3 flag, plane pairs
    address (hi | lo) views of one allocation, and the engine refuses the flag on bf16 storage (no GPU needed)."""
    from mmtg_amd import MMTG, hip, synth
    from mmtg_amd.configs import data_config
    mcfg, dcfg = make_model_cfgs(seq_len=2), data_config(seq_len=2)
    gcfg = gpt2_config(n_layer=2, vocab_size=160, n_positions=128)
    m = MMTG(mcfg, dcfg, 160, gpt2_config=gcfg, token_table=synth.make_token_table(160, seed=1), compute_dtype="bf16x3")
    assert m.compute_dtype == hip.F32 and m.x3
    assert not MMTG(mcfg, dcfg, 160, gpt2_config=gcfg, token_table=synth.make_token_table(160, seed=1), compute_dtype="f32").x3
    with pytest.raises(KeyError):
        MMTG(mcfg, dcfg, 160, gpt2_config=gcfg, token_table=synth.make_token_table(160, seed=1), compute_dtype="bf16x2")
    with pytest.raises(RuntimeError):
        m.engine()                       # the hot path is HIP-only: no CPU engine in any mode
    # a width the split-precision products cannot tile (n_embd % 128 != 0) is refused when the model is built -- the mode has no
    # mixed fallback (round-5 advice: it used to reach the bf16 copies' code with plane pairs) -- while f32 / bf16 take it
    narrow = gpt2_config(n_layer=1, vocab_size=160, n_positions=128, n_embd=192, n_head=3)
    with pytest.raises(ValueError, match="multiple of 128"):
        MMTG(mcfg, dcfg, 160, gpt2_config=narrow, token_table=synth.make_token_table(160, seed=1), compute_dtype="bf16x3")
    assert MMTG(mcfg, dcfg, 160, gpt2_config=narrow, token_table=synth.make_token_table(160, seed=1), compute_dtype="f32").shapes.D == 192
    t = torch.arange(2 * 3 * 8, dtype=torch.float32).bfloat16().view(2, 3, 8)
    p = hip.Planes(t, 3, 8)
    assert (p.ld, p.plane) == (8, 24)
    assert torch.equal(p.float(), t[0].float() + t[1].float())
    sub = hip.Planes(t[0, :, 4:], 3, 4, ld=8, plane=24)        # a column block of both planes
    assert sub.t.data_ptr() == t.data_ptr() + 8 and sub.plane == 24
    # scratch sizes the mode's host code hands the library (include/mmtg_hip.h): one dQ buffer per block of 128 keys, the
    # attention bias-gradient rows; and the K-split rule of its grouped weight gradients (combined stages: 512 workgroup slots)
    assert hip.attn_bwd_x3_dq_floats(64, 236, 768) == 2 * 64 * 236 * 768
    assert hip.attn_bwd_x3_dq_floats(1, 128, 768) == 128 * 768 and hip.attn_bwd_x3_dq_floats(2, 129, 64) == 2 * 2 * 129 * 64
    assert hip.attn_bwd_x3_ws(64, 236, 768) == (64 * 2 + (64 * 236 + 15) // 16) * 3 * 768
    from mmtg_amd import engine as E
    if E._X3_WG_CFG == 6 and E._WGRAD_GROUP_SPLITS <= 0:
        assert E._group_splits_x3(432, 15104) == 1          # a GPT-2 block: one round of 432 workgroups
        assert E._group_splits_x3(88, 15104) == 5           # the projector: 440 workgroups
        assert E._group_splits_x3(630, 15104) == 3          # the tied embedding: ~4 rounds' worth
        assert E._group_splits_x3(88, 2048) == 2            # never a K slice under 1024 tokens
        assert E._group_splits_x3(4, 10 ** 6) == 16
