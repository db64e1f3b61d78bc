"""CPU-only checks of the host logic: state-dict contract, flat layout, C-ABI symbol
table, loud failure without a GPU, curriculum filter / LR schedule, and the N>1
gradient exchange on the gloo backend (world_size 2)."""
import ctypes
import os
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


def tiny_model(S=5, L=2, V=160):
    mcfg = make_model_cfgs(seq_len=S)
    gcfg = gpt2_config(n_layer=L, vocab_size=V, n_positions=256)
    return MMTG(mcfg, data_config(seq_len=S), V, gpt2_config=gcfg), mcfg, gcfg


def test_state_dict_keys_match_reference():
    fx, meta, mcfg, gcfg, dcfg, weights, table, batch = load_case("tiny_s5")
    model = MMTG(mcfg, dcfg, meta["V"], gpt2_config=gcfg)
    ref_keys = sorted(str(k) for k in fx["grad_keys"])       # named_parameters() of the reference
    sd = model.state_dict()
    assert sorted(k for k in sd if k != "decoder.gpt2.lm_head.weight") == ref_keys
    assert sd["decoder.gpt2.lm_head.weight"].data_ptr() == sd["decoder.gpt2.transformer.wte.weight"].data_ptr()
    for k, v in weights.items():
        assert tuple(sd[k].shape) == v.shape, k
    n = sum(p.numel() for p in model.parameters())
    assert n == sum(v.size for k, v in weights.items() if k != "decoder.gpt2.lm_head.weight")


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


def test_load_state_dict_variants_and_roundtrip(tmp_path):
    model, mcfg, gcfg = tiny_model()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    other, _, _ = tiny_model()
    wrapped = {"module." + k: v for k, v in sd.items()}                      # saved from nn.DataParallel
    wrapped["module.decoder.gpt2.transformer.h.0.attn.bias"] = torch.ones(1, 1, 4, 4, dtype=torch.uint8)
    wrapped["module.decoder.gpt2.transformer.h.0.attn.masked_bias"] = torch.tensor(-1e4)
    other.load_state_dict(wrapped)
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k
    torch.save({"model": other.state_dict(), "args": None, "model_cfgs": mcfg}, tmp_path / "ckpt.pth")
    third, _, _ = tiny_model()
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
    assert hip.lib().mmtg_abi_version() == 1


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
        red.finish(grad)
        n = red.global_count(3 + rank, "cpu")
        torch.save((rank, mine, grad, len(red.buckets), n, launched), os.path.join(outdir, 'r%d.pt' % rank))
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
    for rank, mine, reduced, nb, n, launched in res:
        assert nb > 3            # really bucketed
        assert n == 7            # 3 + 4 rows across ranks
        assert torch.equal(reduced, total)
