"""GPU parity at the BENCHMARKED sizes (BASELINE.json configs[1] and configs[2]): the HIP path through the C ABI vs
the oracle on the same seeded inputs.  These sizes select kernel paths the small cases never reach (4 m-tiles per skinny
job, 128x128 GEMM tiles with split-K, the 2m = 128-row paired rollout).  About 10 s of CPU oracle per case."""
import functools

import os
import numpy as np
import pytest
import torch

from oracle import paramgen as pg
from oracle import xgate_oracle as xo
from tests.util import CFG, ZERO_GRAD_PARAMS, assert_grads_close, make_model, oracle_grads, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    import __graft_entry__ as ge
    ge.build()
    torch.set_num_threads(min(16, torch.get_num_threads()))


@functools.lru_cache(maxsize=4)
def _params(V, gain, bias0):
    d = pg.make_dims(**dict(CFG["c1"], V=V))
    P = pg.make_params(d, logit_gain=gain)
    if bias0:
        P["logit.bias"] = P["logit.bias"].copy()
        P["logit.bias"][0] += bias0
    return P


# ------------------------------------------------------------------ configs[1]: B = 128 teacher-forced XE, fp32
@pytest.mark.parametrize("path,ragged", [("fused", False), ("surface", True)])
def test_config2_b128_xe_loss_logprobs_and_every_gradient_vs_oracle(path, ragged):
    from controllable_xgating_amd import LanguageModelCriterion
    d = pg.make_dims(**dict(CFG["c1"], B=128))
    Pn = _params(d.V, 8.0, 0.0)
    xn = pg.make_inputs(d, seed=0, ragged=ragged)
    P = xo.to_torch_params(Pn, requires_grad=True)
    xi = xo.to_torch_inputs(xn)
    running = xo.new_running(d)
    logp_o, _, _ = xo.forward_xe(P, xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"], xi["seq"],
                                 xi["seq_mask"], train=True, running=running)
    loss_o = xo.lm_criterion(logp_o, xi["seq"], xi["seq_mask"])
    loss_o.backward()
    model = make_model(d, P=Pn)
    x = to_dev(xn)
    if path == "fused":          # what bench.py times
        loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        logp = None
    else:                        # the reference's call sequence (starttrain.py:125-126)
        logp, _ = model(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        loss = LanguageModelCriterion()(logp, x["seq"], x["seq_mask"])
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_o.item()) < 1e-4, (loss.item(), loss_o.item())      # north_star: 1e-4 on the training loss
    if logp is not None:
        lo = logp_o.detach()
        np.testing.assert_allclose(logp.detach()[:, :, :64].cpu().numpy(), lo[:, :, :64].numpy(), atol=3e-4, rtol=0)
        tgt = torch.cat([xi["seq"][:, 1:], xi["seq"][:, :1]], 1).unsqueeze(2)
        np.testing.assert_allclose(logp.detach().cpu().gather(2, tgt).numpy(), lo.gather(2, tgt).numpy(), atol=3e-4, rtol=0)
        # every row is a normalised distribution
        np.testing.assert_allclose(torch.logsumexp(logp.detach(), 2).cpu().numpy(), 0.0, atol=1e-4)
    assert_grads_close(model, oracle_grads(P), skip=ZERO_GRAD_PARAMS)
    for mod in ("rgb", "opfl"):
        bn = getattr(model.two_spatial_encoder, f"visual_emb_{mod}")[1]
        pre = xo.ENC + f"visual_emb_{mod}.1."
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), running[pre + "running_mean"].numpy(), atol=1e-5)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), running[pre + "running_var"].numpy(), atol=1e-5)


# ------------------------------------------------------------------ configs[1] at the reference's default drop_prob_lm = 0.5
def test_config2_b128_xe_with_dropout_half_loss_and_every_gradient_vs_oracle():
    """B = 128 teacher-forced XE in train mode at p = 0.5 (myopts.py:37; mask sites sub_modules.py:123,128,45,69-71,767 and
    SAModel.py:49): the HIP kernels regenerate the oracle's integer-hash masks from the shared seed, so loss (1e-4), running
    statistics and EVERY gradient compare as in the p = 0 case -- at the benchmarked size, through the fused loss path that
    bench.py's secondary.xe_drop05 line times."""
    d = pg.make_dims(**dict(CFG["c1"], B=128))
    Pn = _params(d.V, 8.0, 0.0)
    xn = pg.make_inputs(d, seed=0, ragged=True)
    seed, p = 987654321, 0.5
    P = xo.to_torch_params(Pn, requires_grad=True)
    xi = xo.to_torch_inputs(xn)
    running = xo.new_running(d)
    logp_o, _, _ = xo.forward_xe(P, xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"], xi["seq"],
                                 xi["seq_mask"], train=True, p=p, seed=seed, running=running)
    loss_o = xo.lm_criterion(logp_o, xi["seq"], xi["seq_mask"])
    loss_o.backward()
    # the masks matter: the same weights and batch without dropout give another loss
    with torch.no_grad():
        logp_0, _, _ = xo.forward_xe(xo.to_torch_params(Pn), xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"],
                                     xi["seq"], xi["seq_mask"], train=True, running=xo.new_running(d))
        assert abs(xo.lm_criterion(logp_0, xi["seq"], xi["seq_mask"]).item() - loss_o.item()) > 1e-3
    model = make_model(d, P=Pn, p_drop=p)
    model.dropout_seed = seed
    x = to_dev(xn)
    loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_o.item()) < 1e-4, (loss.item(), loss_o.item())
    # (relu_flips: the surviving activations are doubled and half as many, so one ReLU derivative that flips between the two fp32
    #  evaluations shows at up to ~2 % of a gradient's largest entry -- tests/util.py:relu_flip_exposed; everything downstream of
    #  the ReLUs keeps the strict bounds)
    assert_grads_close(model, oracle_grads(P), skip=ZERO_GRAD_PARAMS, relu_flips=True)
    for mod in ("rgb", "opfl"):
        bn = getattr(model.two_spatial_encoder, f"visual_emb_{mod}")[1]
        pre = xo.ENC + f"visual_emb_{mod}.1."
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), running[pre + "running_mean"].numpy(), atol=1e-5)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), running[pre + "running_var"].numpy(), atol=1e-5)
    # a second call with another seed draws other masks
    model.dropout_seed = seed + 1
    with torch.no_grad():
        loss2 = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
    assert abs(loss2.item() - loss.item()) > 1e-5


# ------------------------------------------------------------------ configs[4]: bf16, hidden 1024, 40 frames, vocab 20k, B = 128
def test_config5_b128_hidden1024_bf16_and_split_bf16_vs_oracle():
    """BASELINE.json configs[4] at its FULL size (B = 128, K = 40, R = 1024, V = 20000): the kernel paths only this size
    takes (4-wave skinny workgroups, the fused-attention step form, the staged 33-48-frame attention backward, 256 x 128 bf16
    tiles, background products).  precision='bf16': loss within 1e-2 of the fp32 oracle (north_star's tolerance for this
    config) and every gradient norm within 5 %; precision='bf16x3' (split-bf16, fp32-class): loss within 1e-4."""
    d = pg.make_dims(**dict(CFG["c5"], B=128, L=20))
    Pn = pg.make_params(d)
    xn = pg.make_inputs(d, seed=0, ragged=True)
    P = xo.to_torch_params(Pn, requires_grad=True)
    xi = xo.to_torch_inputs(xn)
    logp_o, _, _ = xo.forward_xe(P, xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"], xi["seq"],
                                 xi["seq_mask"], train=True, running=xo.new_running(d))
    loss_o = xo.lm_criterion(logp_o, xi["seq"], xi["seq_mask"])
    loss_o.backward()
    g_o = oracle_grads(P)
    del logp_o
    from controllable_xgating_amd import SAModel, make_opt
    x = to_dev(xn)
    for precision, tol in (("bf16", 1e-2), ("bf16x3", 1e-4)):
        model = SAModel(make_opt(d, precision=precision))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in Pn.items()}, strict=False)
        model = model.cuda()
        model.train()
        loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_o.item()) < tol, (precision, loss.item(), loss_o.item())
        bad = []
        for name, prm in model.named_parameters():
            if name in ZERO_GRAD_PARAMS:
                continue
            gd, rd = prm.grad.double().cpu().reshape(-1), torch.from_numpy(g_o[name].astype(np.float64)).reshape(-1)
            gn, rn = float(gd.norm()), float(rd.norm())
            # direction as well as length: a wrong-signed or permuted block keeps its norm but not its cosine
            cos = float(torch.dot(gd, rd)) / max(gn * rn, 1e-300)
            if not abs(gn - rn) <= (5e-2 if precision == "bf16" else 2e-3) * rn + 1e-6:
                bad.append((name, "norm", gn, rn))
            # (the two embedding matrices sit in front of train-mode BatchNorm: their gradient is what is left after the batch
            #  mean and the x-hat projection are subtracted -- a cancellation that amplifies the bf16 rounding of everything
            #  downstream; measured 0.9983 / 0.9987 on MI355X, every other parameter >= 0.9995)
            pre_bn = name in ("two_spatial_encoder.visual_emb_rgb.0.weight", "two_spatial_encoder.visual_emb_opfl.0.weight")
            cos_min = (0.997 if pre_bn else 0.999) if precision == "bf16" else (0.99999 if pre_bn else 0.999999)
            if rn > 1e-9 and not cos >= cos_min:
                bad.append((name, "cosine", cos))
            # and the elements themselves -- every one of a bias / small matrix (bias gradients are column sums formed inside the
            # products from the bf16 mirror of dY in this mode: a sum with strong cancellation would show here), every 9973rd of a
            # large one -- to the bf16 / fp32 class of the product chains
            idx = torch.arange(0, gd.numel(), 9973 if gd.numel() > (1 << 16) else 1)
            # (lstmcore.gate.gate.0.bias: its column sums are taken from the fp32 dGP, but dGP = dposg * pos * g (1 - g) sums to
            #  nearly nothing over the T * B rows, which amplifies the bf16 rounding of the products that made dposg: 0.092 of the
            #  largest element measured while the vector's cosine passes the 0.999 bound above; every other bias is inside 3 %)
            canc = name == "lstmcore.gate.gate.0.bias"
            tol_e = ((1.2e-1 if canc else (6e-2 if pre_bn else 3e-2)) if precision == "bf16" else 2e-3) * float(rd.abs().max()) + 1e-7
            if float((gd[idx] - rd[idx]).abs().max()) > tol_e:
                bad.append((name, "elements", float((gd[idx] - rd[idx]).abs().max()), tol_e))
        assert not bad, (precision, bad)
        del model
        torch.cuda.empty_cache()


# ------------------------------------------------------------------ configs[2]: SCST, B = 64, seq_len 30
def _scst_case():
    d = pg.make_dims(**dict(CFG["c1"], B=64, L=30))
    Pn = _params(d.V, 1.0, 7.0)          # EOS mass ~ 5-10 % per step: rows finish anywhere between t = 1 and t = 30
    xn = pg.make_inputs(d, seed=0)
    u = pg.uniform("uni64", (d.L + 1, d.B), 91)
    reward = np.repeat(pg.uniform("reward64", (d.B, 1), 3, -1.0, 1.0), d.L, 1)      # myutils.py:76
    return d, Pn, xn, u, reward


def _oracle_replay(d, Pn, xn, forced, reward):
    P = xo.to_torch_params(Pn, requires_grad=True)
    xi = xo.to_torch_inputs(xn)
    s, lp = xo.sample(P, xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"], d.L, mode="replay",
                      forced=forced, train=True, running=xo.new_running(d))
    n = s.shape[1]
    loss = xo.reward_criterion(lp, s, torch.from_numpy(reward[:, :n]))
    loss.backward()
    return P, lp.detach().numpy(), loss.item()


def test_config3_scst_b64_l30_replay_of_oracle_sampled_tokens():
    """model.sample with the oracle's own multinomial draws forced: log-probs, RewardCriterion loss, every gradient."""
    from controllable_xgating_amd import RewardCriterion
    d, Pn, xn, u, reward = _scst_case()
    xi = xo.to_torch_inputs(xn)
    with torch.no_grad():
        s_o, _ = xo.sample(xo.to_torch_params(Pn), xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"], d.L,
                           mode="sample", uniforms=u, train=True, running=xo.new_running(d))
    lens = (s_o.numpy() > 0).sum(1)
    assert lens.min() < 5 and lens.max() > 20                     # ragged finishing steps
    P, lp_o, loss_o = _oracle_replay(d, Pn, xn, s_o, reward)
    model = make_model(d, P=Pn, train=True)
    assert not os.environ.get("XG_NO_OVERLAP") and model._run(True, seed=0).aux, "this case must run with the overlap streams on"
    x = to_dev(xn)
    seq, slp = model.sample(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                            {"sample_max": 0, "forced_tokens": s_o.cuda()})
    n = s_o.shape[1]
    assert np.array_equal(seq.cpu().numpy()[:, :n], s_o.numpy())
    m = np.concatenate([np.ones((d.B, 1), bool), s_o.numpy()[:, :-1] > 0], 1)
    np.testing.assert_allclose(slp.detach().cpu().numpy()[:, :n][m], lp_o[m], atol=3e-4)
    loss = RewardCriterion()(slp[:, :n], seq[:, :n], torch.from_numpy(reward[:, :n]).cuda())
    assert abs(loss.item() - loss_o) < 1e-4, (loss.item(), loss_o)
    loss.backward()
    torch.cuda.synchronize()
    assert_grads_close(model, oracle_grads(P), skip=ZERO_GRAD_PARAMS)
    # and the HIP sampler itself, from the same uniforms: the oracle's tokens up to CDF-boundary coin flips
    model2 = make_model(d, P=Pn, train=True)
    with torch.no_grad():
        s_h, _ = model2.sample(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                               {"sample_max": 0, "uniforms": torch.from_numpy(u).cuda()})
    s_h = s_h.cpu().numpy()
    rows_diff = sum(1 for b in range(d.B) if s_h.shape != tuple(s_o.shape) or not np.array_equal(s_h[b], s_o.numpy()[b]))
    assert rows_diff <= 2, rows_diff


def test_config3_scst_b64_l30_paired_rollout_vs_oracle():
    """model.sample_pair (ONE 2m = 128-row pass: sampled rows + greedy baseline): the sampled half's log-probs / loss /
    gradients equal the oracle's replay of the very tokens it drew, the greedy half equals the oracle's greedy rollout."""
    from controllable_xgating_amd import RewardCriterion
    d, Pn, xn, u, reward = _scst_case()
    model = make_model(d, P=Pn, train=True)
    # the backward below runs over the side streams (the (T - 1)-step view of the T-step workspace, xg_rollout_bwd): say so
    assert not os.environ.get("XG_NO_OVERLAP") and model._run(True, seed=0).aux, "this case must run with the overlap streams on"
    x = to_dev(xn)
    gen, slp, greedy, n = model.sample_pair(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                                            {"uniforms": torch.from_numpy(u).cuda()})
    n_s, n_g = (int(v) for v in n.cpu())
    gen_t, slp_t = gen[:, :n_s], slp[:, :n_s]
    loss = RewardCriterion()(slp_t, gen_t, torch.from_numpy(reward[:, :n_s]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    forced = gen_t.cpu()
    P, lp_o, loss_o = _oracle_replay(d, Pn, xn, forced, reward)
    m = np.concatenate([np.ones((d.B, 1), bool), forced.numpy()[:, :-1] > 0], 1)
    np.testing.assert_allclose(slp_t.detach().cpu().numpy()[m], lp_o[m], atol=3e-4)
    assert abs(loss.item() - loss_o) < 1e-4, (loss.item(), loss_o)
    assert_grads_close(model, oracle_grads(P), skip=ZERO_GRAD_PARAMS)
    xi = xo.to_torch_inputs(xn)
    with torch.no_grad():
        g_o, _, logps = xo.sample(xo.to_torch_params(Pn), xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"],
                                  d.L, mode="greedy", train=True, running=xo.new_running(d), return_logp=True)
    g_h, g_o = greedy[:, :n_g].cpu().numpy(), g_o.numpy()
    assert g_h.shape == g_o.shape, (g_h.shape, g_o.shape)
    for b in range(d.B):
        for t in range(g_o.shape[1]):
            if g_h[b, t] != g_o[b, t]:            # only a round-off-level top-2 margin may flip a greedy token
                top2 = np.sort(logps[t].numpy()[b])[-2:]
                assert top2[1] - top2[0] < 1e-3, (b, t, int(g_h[b, t]), int(g_o[b, t]))
                break
