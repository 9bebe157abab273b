"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference hot path.

Written from the mathematics of the reference, each
function citing the reference file:line it follows.  It is the checker for
the HIP path (tests/, smoke(), bench cpu_baseline); it is validated against the
imported reference by tools/gen_golden.py and tests/test_oracle_golden.py.
Gradients come from torch autograd over this restatement.

All functions take ``P``: a dict  state_dict-name -> torch.float32 tensor.
Dropout (train mode, p>0) uses the integer hash in oracle/paramgen.py so the HIP
kernels can reproduce the very same masks.  Site ids:
  0 emb_rgb  1 emb_opfl  2 gate_rgb(step=frame)  3 gate_opfl(step=frame)  4 fusion
  5 decoder gate(step=t)  6 lstm_1 h(step=t)  7 lstm_2 h(step=t)  8 classifier(step=t)
"""
from __future__ import annotations

import numpy as np
import torch

from . import paramgen as pg

ENC = "two_spatial_encoder."


def _drop(x, seed, site, step, p, train):
    if not train or p <= 0.0:
        return x
    m = torch.from_numpy(pg.keep_mask(seed, site, step, tuple(x.shape), p))
    return x * m


def _lin(x, P, name):
    return x @ P[name + ".weight"].t() + P[name + ".bias"]


# ------------------------------------------------------------------ encoder (CG block)
def batchnorm_train(x, gamma, beta, eps=1e-5):
    """nn.BatchNorm1d in train mode over rows (sub_modules.py:98,102,121): biased
    batch variance for normalisation.  Returns y, batch mean, biased var."""
    mean = x.mean(dim=0)
    var = ((x - mean) ** 2).mean(dim=0)
    y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, mean, var


def encoder_fwd(P, feats_rgb, feats_opfl, mask, train=True, p=0.0, seed=0,
                running=None, momentum=0.1, eps=1e-5):
    """EncoderLstm_two_fc.forward (sub_modules.py:118-159) with Gate (:42-47) and
    Fusion (:68-72).  ``running``: optional dict with running_mean/var tensors
    (names as in the state_dict) -- updated in place in train mode (unbiased
    var, momentum 0.1) and used for normalisation in eval mode."""
    B, K, _ = feats_rgb.shape
    R = P[ENC + "visual_emb_rgb.0.weight"].shape[0]
    embs = []
    for site, (mod, x) in enumerate((("rgb", feats_rgb), ("opfl", feats_opfl))):
        pre = ENC + f"visual_emb_{mod}."
        z = x.reshape(B * K, -1) @ P[pre + "0.weight"].t() + P[pre + "0.bias"]       # :121,:126
        if train:
            y, mean, var = batchnorm_train(z, P[pre + "1.weight"], P[pre + "1.bias"], eps)
            if running is not None:
                n = z.shape[0]
                with torch.no_grad():
                    running[pre + "1.running_mean"].mul_(1 - momentum).add_(momentum * mean)
                    running[pre + "1.running_var"].mul_(1 - momentum).add_(
                        momentum * var * (n / max(n - 1, 1)))
        else:
            rm, rv = running[pre + "1.running_mean"], running[pre + "1.running_var"]
            y = (z - rm) / torch.sqrt(rv + eps) * P[pre + "1.weight"] + P[pre + "1.bias"]
        y = torch.relu(y).reshape(B, K, R)
        y = _drop(y, seed, site, 0, p, train) * mask.unsqueeze(-1)                    # :123,:128
        embs.append(y)
    h = [torch.zeros(B, R), torch.zeros(B, R)]
    c = [torch.zeros(B, R), torch.zeros(B, R)]
    outs = [[], []]
    for i in range(K):                                                                # :132
        mk = mask[:, i].unsqueeze(-1)
        for j, mod in enumerate(("rgb", "opfl")):
            pre = ENC + f"lstmcell_{mod}."
            s = embs[j][:, i] @ P[pre + "weight_ih"].t() + P[pre + "bias_ih"] \
                + h[j] @ P[pre + "weight_hh"].t() + P[pre + "bias_hh"]
            ig, fg, gg, og = s.chunk(4, dim=1)                      # nn.LSTMCell order i,f,g,o
            cn = torch.sigmoid(fg) * c[j] + torch.sigmoid(ig) * torch.tanh(gg)
            hn = torch.sigmoid(og) * torch.tanh(cn)
            h[j] = hn * mk                                                            # :139-140 zeroing
            c[j] = cn * mk
        g_r = _drop(torch.relu(_lin(h[1], P, ENC + "gate_rgb.gate.0")), seed, 2, i, p, train)
        g_o = _drop(torch.relu(_lin(h[0], P, ENC + "gate_opfl.gate.0")), seed, 3, i, p, train)
        outs[0].append(g_r * h[0] + h[0])                                             # :151, :45
        outs[1].append(g_o * h[1] + h[1])                                             # :152
    y_r = torch.stack(outs[0], dim=1)
    y_o = torch.stack(outs[1], dim=1)
    cat = torch.cat([y_r, y_o], dim=-1)                                               # :69
    V = torch.relu(_lin(cat, P, ENC + "fusion.late_fusion.0"))
    V = _drop(V, seed, 4, 0, p, train)                                                # :155-158
    return V


# ------------------------------------------------------------------ decoder init
def init_hidden(P, V, mask):
    """SAModel.init_hidden (SAModel.py:58-65): masked mean, DETACHED (numpy round trip)."""
    with torch.no_grad():
        vbar = V.detach().sum(dim=1) / mask.sum(dim=1, keepdim=True)
    return [(_lin(vbar, P, "img_embed_h_1"), _lin(vbar, P, "img_embed_c_1")),
            (_lin(vbar, P, "img_embed_h_2"), _lin(vbar, P, "img_embed_c_2"))]


# ------------------------------------------------------------------ decoder step
def _cell(P, name, x1, x2, h, c, mk, p, seed, site, t, train):
    """two_inputs_lstmcell.forward (sub_modules.py:750-770); gate order i,f,o,g."""
    R = h.shape[1]
    s = _lin(x1, P, name + ".i2h") + _lin(x2, P, name + ".a2h") + _lin(h, P, name + ".h2h")
    ig = torch.sigmoid(s[:, 0:R])
    fg = torch.sigmoid(s[:, R:2 * R])
    og = torch.sigmoid(s[:, 2 * R:3 * R])
    gg = torch.tanh(s[:, 3 * R:4 * R])
    cn = fg * c + ig * gg
    cn = cn * mk + c * (1.0 - mk)                                                     # :762
    hn = og * torch.tanh(cn)
    hn = hn * mk + h * (1.0 - mk)                                                     # :765
    hn = _drop(hn, seed, site, t, p, train)                                           # :767
    return hn, cn


def attention(P, V, h1, h2, vproj=None):
    """sub_modules.py:677-680.  softmax over all K frames, unmasked."""
    pq = _lin(torch.cat([h1, h2], dim=1), P, "lstmcore.h2a").unsqueeze(1)
    q = _lin(V, P, "lstmcore.v2a") if vproj is None else vproj
    e = torch.tanh(pq + q) @ P["lstmcore.a2w.weight"].t() + P["lstmcore.a2w.bias"]   # (B,K,1)
    alpha = torch.softmax(e, dim=1)
    af = (alpha * V).sum(dim=1)
    return af, alpha.squeeze(-1)


def core_step(P, xt, mk, V, pos, state, p=0.0, seed=0, t=0, train=True, vproj=None):
    """LSTMCore_two_layer_gate.forward (sub_modules.py:671-687)."""
    (h1, c1), (h2, c2) = state
    af, alpha = attention(P, V, h1, h2, vproj)
    g = _drop(torch.relu(_lin(xt, P, "lstmcore.gate.gate.0")), seed, 5, t, p, train)
    posg = g * pos + pos                                                              # :682
    h1n, c1n = _cell(P, "lstmcore.lstm_1", xt, posg, h1, c1, mk, p, seed, 6, t, train)
    h2n, c2n = _cell(P, "lstmcore.lstm_2", h1n, af, h2, c2, mk, p, seed, 7, t, train)
    return h2n, [(h1n, c1n), (h2n, c2n)], alpha


def heads(P, out, p=0.0, seed=0, t=0, train=True):
    """SAModel.py:109-110."""
    logp = torch.log_softmax(_lin(out, P, "logit"), dim=1)
    hcls = _drop(torch.relu(_lin(out, P, "classifer.0")), seed, 8, t, p, train)
    cat = torch.log_softmax(_lin(hcls, P, "classifer.3"), dim=1)
    return logp, cat


# ------------------------------------------------------------------ teacher-forced forward
def forward_xe(P, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask,
               train=True, p=0.0, seed=0, running=None, hoist=True, trace=None,
               ss_prob=0.0, u_sel=None, u_tok=None, forced_it=None, it_trace=None):
    """SAModel.forward (SAModel.py:67-115).  ``hoist=False`` recomputes v2a(V) every step exactly as the
    reference does (:677).  Scheduled sampling (:89-99): at steps i >= 1 in train mode, rows with
    u_sel[i,b] < ss_prob feed a token drawn (inverse CDF with u_tok[i,b]) from exp(previous step's log-probs)
    instead of seq[b,i]; ``forced_it`` (T,B) replays recorded input tokens instead (golden test)."""
    V = encoder_fwd(P, feats_rgb, feats_opfl, feat_mask, train, p, seed, running)
    state = init_hidden(P, V, feat_mask)
    vproj = _lin(V, P, "lstmcore.v2a") if hoist else None
    outs, cats = [], []
    for i in range(seq.shape[1]):
        if i >= 1 and int(seq[:, i].sum()) == 0:                                      # :103
            break
        it = seq[:, i].clone()
        if forced_it is not None:
            it = forced_it[i].clone()
        elif train and i >= 1 and ss_prob > 0.0:
            lp = outs[-1].detach().numpy()
            for b in range(it.shape[0]):
                if float(u_sel[i, b]) < ss_prob:
                    it[b] = sample_token(lp[b], float(u_tok[i, b]))
        if it_trace is not None:
            it_trace.append(it.clone())
        xt = P["embed.weight"][it]
        mk = seq_mask[:, i].unsqueeze(1)
        out, state, alpha = core_step(P, xt, mk, V, pos_feats, state, p, seed, i, train, vproj)
        logp, cat = heads(P, out, p, seed, i, train)
        outs.append(logp)
        cats.append(cat)
        if trace is not None:
            trace.append(dict(h1=state[0][0].detach(), c1=state[0][1].detach(),
                              h2=state[1][0].detach(), c2=state[1][1].detach(),
                              alpha=alpha.detach()))
    return torch.stack(outs, dim=1), torch.stack(cats, dim=1), V


# ------------------------------------------------------------------ rollouts
def sample_token(logp_row: np.ndarray, u: float, temperature: float = 1.0) -> int:
    """Inverse-CDF draw from exp(logp/temperature) (unnormalised, like
    torch.multinomial at SAModel.py:190-194): first index whose running float64
    sum exceeds u * total."""
    w = np.exp(logp_row.astype(np.float64) / temperature)
    cdf = np.cumsum(w)
    return int(min(np.searchsorted(cdf, u * cdf[-1], side="right"), len(w) - 1))


def sample(P, feats_rgb, feats_opfl, feat_mask, pos_feats, L, mode="greedy",
           uniforms=None, forced=None, temperature=1.0, train=False, p=0.0, seed=0,
           running=None, return_logp=False):
    """SAModel.sample (SAModel.py:163-219).  mode: 'greedy' (:186), 'sample'
    (inverse-CDF with supplied uniforms (L+1,B)), 'replay' (forced tokens (B,n)).
    Returns seq (B,n) int64, seqLogprobs (B,n) (torch, differentiable), and
    optionally the per-step logp list."""
    V = encoder_fwd(P, feats_rgb, feats_opfl, feat_mask, train, p, seed, running)
    B = V.shape[0]
    state = init_hidden(P, V, feat_mask)
    vproj = _lin(V, P, "lstmcore.v2a")
    seqs, slps, logps = [], [], []
    logp = None
    unfinished = None
    for t in range(L + 1):
        if t == 0:
            it = torch.zeros(B, dtype=torch.int64)
        elif mode == "greedy":
            slp, it = torch.max(logp.detach(), 1)
        else:
            if mode == "replay":
                if t - 1 >= forced.shape[1]:
                    break
                it = forced[:, t - 1].clone()
            else:
                lp = logp.detach().numpy()
                it = torch.tensor([sample_token(lp[b], float(uniforms[t, b]), temperature)
                                   for b in range(B)], dtype=torch.int64)
            slp = logp.gather(1, it.unsqueeze(1)).squeeze(1)                          # :195
        xt = P["embed.weight"][it]                                                    # :198
        if t >= 1:
            unfinished = (it > 0) if t == 1 else unfinished & (it > 0)                # :200-204
            if mode != "replay" and int(unfinished.sum()) == 0:
                break
            if mode != "replay":
                it = it * unfinished.to(it.dtype)                                     # :208
            seqs.append(it)
            slps.append(slp)
        mk = torch.ones(B, 1) if t == 0 else unfinished.float().unsqueeze(1)          # :212-215
        out, state, _ = core_step(P, xt, mk, V, pos_feats, state, p, seed, t, train, vproj)
        logp = torch.log_softmax(_lin(out, P, "logit"), dim=1)                        # :217
        logps.append(logp)
    if not seqs:                      # every row finished at t = 1 (the reference's torch.cat would raise here)
        res = (torch.zeros(B, 0, dtype=torch.int64), torch.zeros(B, 0))
    else:
        res = (torch.stack(seqs, 1), torch.stack(slps, 1))
    return res + (logps,) if return_logp else res


# ------------------------------------------------------------------ criteria
def lm_criterion(logp, target, mask):
    """LanguageModelCriterion (SAModel.py:225-234): target rolled left by one."""
    tgt = torch.cat([target[:, 1:], target[:, :1]], dim=1)
    out = -logp.gather(2, tgt.unsqueeze(2)).squeeze(2) * mask
    return out.sum() / mask.sum()


def cls_criterion(cat_logp, target, mask, class_mask=None):
    """ClassiferCriterion (SAModel.py:240-253): target NOT rolled."""
    out = -cat_logp.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    if class_mask is None:
        return out.sum() / mask.sum()
    return (out * class_mask).sum() / (mask * class_mask).sum()


def reward_criterion(slp, seq, reward):
    """RewardCriterion (SAModel.py:259-267)."""
    m = (seq > 0).float()
    m = torch.cat([torch.ones(m.shape[0], 1), m[:, :-1]], dim=1)
    return (-slp * reward * m).sum() / m.sum()


def clip_gradient(grads, clip):
    """myutils.clip_gradient (myutils.py:79-85): elementwise clamp."""
    return {k: g.clamp(-clip, clip) for k, g in grads.items()}


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam defaults (starttrain.py:76): L2 weight decay added to the
    gradient, bias-corrected.  Pure function on tensors; returns new (p,m,v)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    return p - (lr / bc1) * m / denom, m, v


# ------------------------------------------------------------------ helpers
def to_torch_params(np_params, requires_grad=False):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad)
            for k, v in np_params.items()}


def to_torch_inputs(x):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in x.items()}


def new_running(d):
    r = {}
    for mod in ("rgb", "opfl"):
        pre = ENC + f"visual_emb_{mod}.1."
        r[pre + "running_mean"] = torch.zeros(d.R)
        r[pre + "running_var"] = torch.ones(d.R)
    return r
