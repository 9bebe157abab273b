"""Pins the oracle (oracle/xgate_oracle.py) against fixtures recorded from the
REFERENCE itself (tools/gen_golden.py).  CPU only."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import paramgen as pg
from oracle import xgate_oracle as xo
from tools.gen_golden import CFG, WEIGHT_CLASS

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def setup(tag, ragged=False, grad=False, **over):
    cfg = dict(CFG[tag]); cfg.update(over)
    d = pg.make_dims(**cfg)
    P = xo.to_torch_params(pg.make_params(d), requires_grad=grad)
    x = xo.to_torch_inputs(pg.make_inputs(d, seed=0, ragged=ragged))
    return d, P, x


def sample_idx(name, shape, n=16):
    tot = int(np.prod(shape))
    return pg.hash_u32(7, zlib.crc32(name.encode()) & 0xFFFFFFFF, 3, n).astype(np.int64) % tot


def run_xe(tag, ragged, hoist=True):
    d, P, x = setup(tag, ragged, grad=True)
    running = xo.new_running(d)
    trace = []
    logp, cat, V = xo.forward_xe(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                                 x["seq"], x["seq_mask"], train=True, p=0.0, running=running,
                                 hoist=hoist, trace=trace)
    l_xe = xo.lm_criterion(logp, x["seq"], x["seq_mask"])
    l_cls = xo.cls_criterion(cat, x["cap_classes"], x["seq_mask"], x["class_mask"])
    loss = l_xe + WEIGHT_CLASS * l_cls
    loss.backward()
    return d, P, x, logp, cat, V, trace, running, l_xe, l_cls, loss


@pytest.mark.parametrize("tag,ragged", [("tiny", False), ("tiny", True), ("c1", False), ("c1", True), ("c5", False)])
def test_xe_forward_backward_matches_reference(tag, ragged):
    g = load(f"xe_{tag}{'_ragged' if ragged else ''}.npz")
    d, P, x, logp, cat, V, trace, running, l_xe, l_cls, loss = run_xe(tag, ragged)
    assert abs(l_xe.item() - g["loss_xe"]) < 2e-6 * max(1, abs(g["loss_xe"]))
    assert abs(l_cls.item() - g["loss_cls"]) < 2e-6 * max(1, abs(g["loss_cls"]))
    ns = g["logp_slice"].shape[2]
    np.testing.assert_allclose(logp.detach().numpy()[:, :, :ns], g["logp_slice"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(cat.detach().numpy(), g["cat_logp"], atol=2e-5, rtol=0)
    nr = g["V_slice"].shape[2]
    np.testing.assert_allclose(V.detach().numpy()[:, :, :nr], g["V_slice"], atol=2e-6, rtol=0)
    st = np.array([[s["h1"].numpy(), s["c1"].numpy(), s["h2"].numpy(), s["c2"].numpy()] for s in trace])
    np.testing.assert_allclose(st[..., :g["state_slice"].shape[-1]], g["state_slice"], atol=3e-6, rtol=0)
    for mod in ("rgb", "opfl"):
        pre = xo.ENC + f"visual_emb_{mod}.1."
        np.testing.assert_allclose(running[pre + "running_mean"].numpy(), g[f"bn_{mod}_running_mean"], atol=1e-6)
        np.testing.assert_allclose(running[pre + "running_var"].numpy(), g[f"bn_{mod}_running_var"], atol=1e-6)
    for name, prm in P.items():
        gr = prm.grad.numpy() if prm.grad is not None else np.zeros(prm.shape, np.float32)
        gn = np.sqrt((gr.astype(np.float64) ** 2).sum())
        ref_n = float(g["gnorm/" + name])
        assert abs(gn - ref_n) <= 2e-4 * ref_n + 2e-7, (name, gn, ref_n)
        np.testing.assert_allclose(gr.reshape(-1)[sample_idx(name, gr.shape)], g["gsamp/" + name],
                                   atol=1e-7 + 2e-4 * np.abs(g["gsamp/" + name]).max(), rtol=0, err_msg=name)
        if "gfull/" + name in g:
            np.testing.assert_allclose(gr, g["gfull/" + name], atol=2e-7 + 1e-4 * np.abs(g["gfull/" + name]).max(),
                                       rtol=0, err_msg=name)


def test_unhoisted_equals_hoisted():
    """v2a(V) recomputed per step (reference sub_modules.py:677) == hoisted."""
    a = run_xe("tiny", True, hoist=True)
    b = run_xe("tiny", True, hoist=False)
    np.testing.assert_allclose(a[3].detach().numpy(), b[3].detach().numpy(), atol=1e-6)


@pytest.mark.parametrize("name,tag,ragged", [("greedy_tiny.npz", "tiny", False), ("greedy_c1.npz", "c1", False),
                                              ("greedy_c1_ragged.npz", "c1", True)])
def test_greedy_token_for_token(name, tag, ragged):
    g = load(name)
    d, P, x = setup(tag, ragged)
    P = xo.to_torch_params(pg.make_params(d, logit_gain=float(g["logit_gain"])))      # the gain the fixture was recorded with
    assert float(g["min_margin"]) >= 1e-3 and abs(float(g["min_margin"]) - g["margin"][: g["seq"].shape[1]].min()) < 1e-7
    with torch.no_grad():
        seq, slp = xo.sample(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], d.L,
                             mode="greedy", train=False, running=xo.new_running(d))
    assert seq.shape == g["seq"].shape
    assert np.array_equal(seq.numpy(), g["seq"])
    np.testing.assert_allclose(slp.numpy(), g["seqLogprobs"], atol=2e-5)


def test_single_step_matches_reference():
    g = load("step_c1.npz")
    cfg = dict(CFG["c1"]); cfg["B"] = 4
    d = pg.make_dims(**cfg)
    P = xo.to_torch_params(pg.make_params(d))
    B, K, R, E = d.B, d.K, d.R, d.E
    xt = torch.from_numpy(pg.uniform("step.xt", (B, E), 5, -0.1, 0.1))
    V = torch.from_numpy(pg.uniform("step.V", (B, K, R), 5, 0.0, 1.0))
    pos = torch.from_numpy(pg.uniform("step.pos", (B, R), 5, -1.0, 1.0))
    st = [torch.from_numpy(pg.uniform(f"step.s{i}", (1, B, R), 5, -0.5, 0.5))[0] for i in range(4)]
    mk = torch.tensor([[1.0], [1.0], [0.0], [1.0]])
    out, state, alpha = xo.core_step(P, xt, mk, V, pos, [(st[0], st[1]), (st[2], st[3])])
    np.testing.assert_allclose(out.numpy(), g["out"], atol=2e-6)
    np.testing.assert_allclose(state[0][0].numpy(), g["h1"], atol=2e-6)
    np.testing.assert_allclose(state[0][1].numpy(), g["c1"], atol=2e-6)
    np.testing.assert_allclose(state[1][1].numpy(), g["c2"], atol=2e-6)
    np.testing.assert_allclose(alpha.numpy(), g["alpha"], atol=1e-6)
    # mask-hold row (sub_modules.py:762,765)
    np.testing.assert_array_equal(state[0][0].numpy()[2], st[0].numpy()[2])


@pytest.mark.parametrize("tag", ["tiny", "c1"])
def test_scst_replay_matches_reference(tag):
    g = load(f"scst_{tag}.npz")
    d, P, x = setup(tag, grad=True)
    forced = torch.from_numpy(g["seq"])
    seq, slp = xo.sample(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], d.L,
                         mode="replay", forced=forced, train=True, p=0.0, running=xo.new_running(d))
    assert np.array_equal(seq.numpy(), g["seq"])
    # finished rows gather at token 0 in replay; the reference gathered at the raw draw: compare where it counts
    m = np.concatenate([np.ones((d.B, 1), bool), g["seq"][:, :-1] > 0], 1)
    np.testing.assert_allclose(slp.detach().numpy()[m], g["seqLogprobs"][m], atol=2e-5)
    loss = xo.reward_criterion(slp, seq, torch.from_numpy(g["reward"]))
    assert abs(loss.item() - g["loss"]) < 2e-6
    loss.backward()
    for name, prm in P.items():
        gr = prm.grad.numpy() if prm.grad is not None else np.zeros(prm.shape, np.float32)
        gn = np.sqrt((gr.astype(np.float64) ** 2).sum())
        ref_n = float(g["gnorm/" + name])
        assert abs(gn - ref_n) <= 3e-4 * ref_n + 2e-7, (name, gn, ref_n)


@pytest.mark.parametrize("tag", ["tiny", "c1"])
def test_eval_mode_batchnorm(tag):
    g = load(f"evalbn_{tag}.npz")
    d, P, x = setup(tag, ragged=True)
    running = xo.new_running(d)
    for mod in ("rgb", "opfl"):
        pre = xo.ENC + f"visual_emb_{mod}.1."
        running[pre + "running_mean"] = torch.from_numpy(pg.uniform(f"rm.{mod}", (d.R,), 9, -0.3, 0.3))
        running[pre + "running_var"] = torch.from_numpy(pg.uniform(f"rv.{mod}", (d.R,), 9, 0.5, 2.0))
    with torch.no_grad():
        logp, cat, _ = xo.forward_xe(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                                     x["seq"], x["seq_mask"], train=False, running=running)
        loss = xo.lm_criterion(logp, x["seq"], x["seq_mask"])
    assert abs(loss.item() - g["loss"]) < 2e-6 * abs(g["loss"])
    np.testing.assert_allclose(logp.numpy()[:, :, :32], g["logp_slice"], atol=2e-5)


def test_dropout_hash_statistics_and_determinism():
    m1 = pg.keep_mask(11, 6, 3, (64, 512), 0.5)
    m2 = pg.keep_mask(11, 6, 3, (64, 512), 0.5)
    assert np.array_equal(m1, m2)
    assert set(np.unique(m1)) == {0.0, 2.0}
    assert abs((m1 > 0).mean() - 0.5) < 0.01
    assert not np.array_equal(m1, pg.keep_mask(11, 6, 4, (64, 512), 0.5))
    assert not np.array_equal(m1, pg.keep_mask(11, 7, 3, (64, 512), 0.5))
    assert np.all(pg.keep_mask(1, 1, 1, (8, 8), 0.0) == 1.0)


def test_scheduled_sampling_replay_matches_reference():
    """f-3: the reference's own draws (recorded input tokens per step) replayed through the oracle."""
    g = load("ss_tiny.npz")
    d, P, x = setup("tiny", ragged=True, grad=True)
    assert int(g["n_replaced"]) > 0
    logp, cat, _ = xo.forward_xe(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"],
                                 train=True, running=xo.new_running(d), forced_it=torch.from_numpy(g["it"]))
    loss = xo.lm_criterion(logp, x["seq"], x["seq_mask"])
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    loss.backward()
    for name, prm in P.items():
        gr = prm.grad.numpy() if prm.grad is not None else np.zeros(prm.shape, np.float32)
        gn = np.sqrt((gr.astype(np.float64) ** 2).sum())
        ref_n = float(g["gnorm/" + name])
        assert abs(gn - ref_n) <= 3e-4 * ref_n + 2e-7, (name, gn, ref_n)


def test_greedy_with_natural_eos_matches_reference():
    """Rows finish at steps 3..13 (two never do) and 44 distinct words are emitted: pins `unfinished`, the zeroing of
    finished rows and the state hold under xt_mask (SAModel.py:200-215) against the reference itself."""
    from tests.util import EOS_CASE, eos_params
    g = load("greedy_c1_eos.npz")
    d = pg.make_dims(**CFG["c1"])
    P = xo.to_torch_params(eos_params(d))
    x = xo.to_torch_inputs(pg.make_inputs(d, seed=EOS_CASE["input_seed"]))
    with torch.no_grad():
        seq, slp = xo.sample(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], d.L,
                             mode="greedy", train=False, running=xo.new_running(d))
    assert g["margin"][g["alive"]].min() >= 1e-3
    ends = [int(np.flatnonzero(g["seq"][b] == 0)[0]) for b in range(d.B) if (g["seq"][b] == 0).any()]
    assert len(ends) >= 4 and min(ends) >= 2 and len(np.unique(g["seq"])) > 30
    assert np.array_equal(seq.numpy(), g["seq"])
    np.testing.assert_allclose(slp.numpy(), g["seqLogprobs"], atol=3e-5)


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_three_iteration_adam_trajectory_matches_reference(tag):
    """clamp +-0.1 (myutils.py:79-85) + torch.optim.Adam defaults (starttrain.py:76,134-137) on the reference model, three
    iterations: the oracle's forward/backward + clip_gradient + adam_step follow the same losses and parameters."""
    from tests.util import ZERO_GRAD_PARAMS
    g = load(f"traj_{tag}.npz")
    d, P, x = setup(tag, ragged=True, grad=True)
    P0 = {k: t.detach().numpy().copy() for k, t in P.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(t) for k, t in P.items()}
    running = xo.new_running(d)
    losses = []
    for step in (1, 2, 3):
        for t in P.values():
            t.grad = None
        logp, cat, _ = xo.forward_xe(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"],
                                     train=True, running=running)
        loss = xo.lm_criterion(logp, x["seq"], x["seq_mask"]) + WEIGHT_CLASS * xo.cls_criterion(
            cat, x["cap_classes"], x["seq_mask"], x["class_mask"])
        loss.backward()
        losses.append(loss.item())
        grads = xo.clip_gradient({k: (P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])) for k in P}, 0.1)
        with torch.no_grad():
            for k in P:
                pn, m[k], v[k] = xo.adam_step(P[k].detach(), grads[k], m[k], v[k], step, float(g["lr"]))
                P[k].copy_(pn)
    np.testing.assert_allclose(losses, g["losses"], atol=5e-6)
    assert g["losses"][2] < g["losses"][0]
    for name in P:
        if name in ZERO_GRAD_PARAMS:       # true gradient exactly zero: Adam turns round-off noise into +-lr steps
            continue
        idx = g["pidx/" + name]
        got = P[name].detach().numpy().reshape(-1)[idx]
        np.testing.assert_allclose(got, g["psamp/" + name], atol=2e-6, err_msg=name)
        # three steps of at most lr each: the DISPLACEMENT itself must agree, not just the value
        np.testing.assert_allclose(got - P0[name].reshape(-1)[idx], g["dsamp/" + name], atol=2e-6, err_msg=name)
        assert abs(np.sqrt((P[name].detach().numpy().astype(np.float64) ** 2).sum()) - g["pnorm/" + name]) <= 1e-5 * max(
            1.0, g["pnorm/" + name]), name
    for mod in ("rgb", "opfl"):
        pre = xo.ENC + f"visual_emb_{mod}.1."
        # the Linear bias in front of BatchNorm has a true gradient of zero: Adam moves it by +-lr per step on round-off
        # noise, which shifts the batch MEAN (not the output, not the variance) by up to momentum * 3 * lr
        np.testing.assert_allclose(running[pre + "running_mean"].numpy(), g[f"bn_{mod}_running_mean"], atol=2e-4)
        np.testing.assert_allclose(running[pre + "running_var"].numpy(), g[f"bn_{mod}_running_var"], atol=1e-5)
