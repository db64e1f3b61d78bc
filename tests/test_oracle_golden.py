"""Pin the oracle (oracle/mmtg_oracle.py) to vectors produced by executing the
reference itself (tools/make_golden.py).  CPU only.

Tolerances: fp32 vs fp32 with different op order (own GPT-2 vs transformers,
own GRU vs nn.GRU): logits <= 2e-4 abs at |logit| ~ 8, scalars <= 1e-5 rel,
token ids exact.
"""
import json

import numpy as np
import pytest
import torch

from helpers import batch_to_torch, load_case, sample_like_fixture
from oracle import mmtg_oracle as O


def _setup(name, requires_grad=False):
    fx, meta, mcfg, gcfg, dcfg, weights, table, batch = load_case(name)
    sh = O.Shapes(mcfg, dcfg, gcfg)
    w = O.weights_to_torch(weights, requires_grad)
    return fx, meta, sh, w, torch.from_numpy(table), batch_to_torch(batch)


CASES = ["tiny_s5", "tiny_s2", "tiny_lstm2_rnn2", "tiny_gru2_lstm1"]     # the last two: encoder types / depths of model.py:41-59


@pytest.mark.parametrize("case", CASES)
def test_forward_and_intermediates(case):
    fx, meta, sh, w, table, batch = _setup(case)
    col = {}
    with torch.no_grad():
        lm, kl, logits = O.mmtg_forward(w, sh, table, batch, True, col)
    ts = int(fx["tstride"])
    for k, v in col.items():
        ref = fx["int_" + k]
        got = v.numpy()
        if got.ndim == 3 and got.shape[1] > 64:
            got = got[:, ::ts]
        np.testing.assert_allclose(got, ref, atol=3e-4, rtol=1e-4, err_msg=k)
    np.testing.assert_allclose(logits.numpy(), fx["logits"], atol=2e-4, rtol=0)
    assert abs(lm.item() - float(fx["lm_loss"])) < 1e-5 * max(1, abs(float(fx["lm_loss"])))
    assert abs(kl.item() - float(fx["kl"])) < 1e-5 * max(1, abs(float(fx["kl"])))
    for stage in (1, 2, 3):
        got = O.my_loss(logits, batch["targets"], batch["rating"], stage, sh.P).item()
        ref = float(fx[f"myloss_stage{stage}"])
        assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (stage, got, ref)


@pytest.mark.parametrize("case", CASES)
def test_grads_and_one_step(case):
    fx, meta, sh, w, table, batch = _setup(case, requires_grad=True)
    hp = json.loads(str(fx["train_hparams"]))
    state = {}
    total, loss, kl, gn = O.train_step(w, sh, table, batch, batch["rating"], hp["stage"],
                                       hp["alpha"], hp["lr"], 1, state, hp["clip"])
    assert abs(total.item() - float(fx["train_total_loss"])) < 2e-5 * abs(float(fx["train_total_loss"]))
    assert abs(gn.item() - float(fx["grad_total_norm"])) < 1e-4 * float(fx["grad_total_norm"])
    gnorm_total = float(fx["grad_total_norm"])
    for k in fx["grad_keys"]:
        k = str(k)
        if k == "decoder.gpt2.lm_head.weight":
            continue
        idx = fx["gidx_" + k]
        g = sample_like_fixture(w[k].grad.numpy(), idx)
        scale = max(float(fx["gnorm_" + k]) / gnorm_total, 1e-3)
        np.testing.assert_allclose(g, fx["gval_" + k], atol=2e-5 * scale + 1e-7, rtol=2e-3, err_msg=k)
        p = sample_like_fixture(w[k].detach().numpy(), idx)
        # Adam's first step moves every weight by ~lr*sign(g): compare the moved weights
        np.testing.assert_allclose(p, fx["pval_" + k], atol=2e-5, rtol=0, err_msg=k)


def test_full_shape_spot_checks():
    fx, meta, sh, w, table, batch = _setup("full_12l")
    with torch.no_grad():
        lm, kl, logits = O.mmtg_forward(w, sh, table, batch, True)
    idx = fx["logit_idx"]
    got = logits.numpy()[idx[:, 0], idx[:, 1], idx[:, 2]]
    np.testing.assert_allclose(got, fx["logit_val"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(torch.logsumexp(logits, -1).numpy(), fx["logit_lse"], atol=5e-4)
    top1 = logits.argmax(-1).numpy()
    # top-1 must agree wherever the reference's top-2 margin is not a near-tie
    margin = fx["logit_top5_val"][..., 0] - fx["logit_top5_val"][..., 1]
    ok = margin > 2e-3
    assert (top1[ok] == fx["logit_top5"][..., 0][ok]).all()
    assert abs(lm.item() - float(fx["lm_loss"])) < 1e-4
    for stage in (1, 2, 3):
        got = O.my_loss(logits, batch["targets"], batch["rating"], stage, sh.P).item()
        assert abs(got - float(fx[f"myloss_stage{stage}"])) < 1e-4 * max(1, abs(float(fx[f"myloss_stage{stage}"])))


def test_full_shape_gradients_and_greedy_calls():
    """Full depth (12 layers, V = 13317): the oracle's backward against the reference's sampled gradients / norms,
    and the oracle's logits at every call of the reference's 220-position greedy run (one causal forward over the
    reference's own sequence) against the stored raw top-8 values and chosen ids."""
    fx, meta, sh, w, table, batch = _setup("full_12l", requires_grad=True)
    hp = json.loads(str(fx["train_hparams"]))
    state = {}
    total, loss, kl, gn = O.train_step(w, sh, table, batch, batch["rating"], hp["stage"], hp["alpha"], hp["lr"], 1, state, hp["clip"])
    assert abs(total.item() - float(fx["train_total_loss"])) < 2e-5 * abs(float(fx["train_total_loss"]))
    ref_gn = float(fx["grad_total_norm"])
    assert abs(gn.item() - ref_gn) < 2e-4 * ref_gn
    for k in (str(k) for k in fx["grad_keys"]):
        if k == "decoder.gpt2.lm_head.weight":
            continue
        g = sample_like_fixture(w[k].grad.numpy(), fx["gidx_" + k])
        scale = max(float(np.abs(fx["gval_" + k]).max()), float(fx["gnorm_" + k]) / ref_gn / np.sqrt(w[k].numel()), 1e-7)
        assert float(np.abs(g - fx["gval_" + k]).max()) < 2e-3 * scale, k
    # greedy calls
    fx, meta, sh, w, table, batch = _setup("full_12l")
    ids = fx["greedy_len220_row0"].tolist()
    inp = {k: v[:1] for k, v in batch.items() if k != "rating"}
    inp["targets"] = torch.tensor(ids).view(1, -1)
    with torch.no_grad():
        lg = O.mmtg_forward(w, sh, table, inp, train_flag=False)[2][0]
    calls = [i for i in range(220) if not (i > 0 and (i + 2) % 22 in (0, 1))]
    t8, v8 = fx["greedy_len220_row0_top8"], fx["greedy_len220_row0_top8_val"]
    chosen, margin = fx["greedy_len220_row0_chosen"], fx["greedy_len220_row0_margin"]
    n = 0
    for c, i in enumerate(calls):
        if i >= len(ids):
            break
        row = lg[sh.P + i]
        np.testing.assert_allclose(row[t8[c].astype(np.int64)].numpy(), v8[c], atol=5e-4, rtol=0)
        if ids[i] != 0 and margin[c] > 2e-3:
            pl = O.process_logits(row, torch.tensor(ids[:i + 1]), 1.1, 1.5)
            assert int(torch.argmax(pl)) == int(chosen[c]), c
            n += 1
    assert n > 150


def test_filtering_kats():
    fx = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "filtering.npz"))
    for i in range(fx["in"].shape[0]):
        out = O.top_k_top_p_filtering(torch.from_numpy(fx["in"][i].copy()),
                                      int(fx["top_k"][i]), float(fx["top_p"][i]))
        np.testing.assert_array_equal(out.numpy(), fx["out"][i])


@pytest.mark.parametrize("length,row,case", [(30, 0, "tiny_s5"), (30, 1, "tiny_s5"), (220, 0, "tiny_s5"), (30, 0, "tiny_lstm2_rnn2")])
def test_greedy_decode_ids(length, row, case):
    fx, meta, sh, w, table, batch = _setup(case)
    dp = json.loads(str(fx["decode_params"]))
    start = {k: v[row].numpy() for k, v in batch.items() if k != "rating"}
    start["targets"] = np.asarray([1])
    trace = []

    def fwd(inputs):
        return O.mmtg_forward(w, sh, table, inputs, train_flag=False)[2]

    ids = O.sample_sequence(fwd, start, length, temperature=dp["temperature"], top_k=dp["top_k"],
                            top_p=dp["top_p"], repitition_penalty=dp["repitition_penalty"],
                            greedy=True, trace=trace)
    ref = fx[f"greedy_len{length}_row{row}"]
    raw = fx[f"greedy_len{length}_row{row}_rawlogits"]
    assert ids == ref.tolist()
    got = torch.stack(trace).numpy()
    assert got.shape == raw.shape
    np.testing.assert_allclose(got, raw, atol=3e-4, rtol=0)


@pytest.mark.parametrize("length,row,case", [(30, 1, "tiny_s5"), (220, 0, "tiny_s5"), (30, 0, "tiny_lstm2_rnn2")])
def test_kv_cached_oracle_loop_equals_the_reference_loop(length, row, case):
    """The KV-cached CPU loop (oracle.CachedForward: bench.py's "cached" cpu_baseline leg) against the reference's own
    id lists and raw per-call logits -- the same golden vectors that pin the prefix-re-running loop."""
    fx, meta, sh, w, table, batch = _setup(case)
    dp = json.loads(str(fx["decode_params"]))
    start = {k: v[row].numpy() for k, v in batch.items() if k != "rating"}
    start["targets"] = np.asarray([1])
    trace = []
    ids = O.sample_sequence(O.CachedForward(w, sh, table), start, length, temperature=dp["temperature"], top_k=dp["top_k"],
                            top_p=dp["top_p"], repitition_penalty=dp["repitition_penalty"], greedy=True, trace=trace)
    assert ids == fx[f"greedy_len{length}_row{row}"].tolist()
    np.testing.assert_allclose(torch.stack(trace).numpy(), fx[f"greedy_len{length}_row{row}_rawlogits"], atol=3e-4, rtol=0)


def test_curriculum_filter_and_schedule():
    r = torch.tensor([1, 5, 3, 2, 4, 5, 1])
    assert O.curriculum_filter(r, 1).tolist() == [0, 6, 1, 5]
    assert O.curriculum_filter(r, 2).tolist() == [0, 3, 6, 1, 4, 5]
    assert O.curriculum_filter(r, 3).tolist() == list(range(7))
    assert O.linear_schedule(0, 10, 100) == 0.0
    assert O.linear_schedule(5, 10, 100) == 0.5
    assert abs(O.linear_schedule(55, 10, 100) - 0.5) < 1e-12
    assert O.linear_schedule(100, 10, 100) == 0.0
