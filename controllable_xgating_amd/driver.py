"""Train / eval driver counterpart: the CALL SEQUENCE of the reference's py2-only
``caption_src/starttrain.py:84-241`` and ``caption_src/myutils.py:41-85`` restated on top of the HIP model surface.
and ``caption_src/eval_utils.py:18-84`` (eval_split).  Data loading, TensorBoard and the coco-caption metrics are out of scope:
batches are handed in as dicts of CUDA tensors and the SCST reward scorer is a caller-supplied callable
(BASELINE.json config 3 stubs CIDEr).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .model import ClassiferCriterion, LanguageModelCriterion, RewardCriterion
from .train import ClipAdam, GradSync, allreduce_gradients


def lr_for_epoch(opt, epoch):
    """starttrain.py:88-94: step decay lr * rate ** int((epoch - start) / every) once epoch > start >= 0."""
    start = getattr(opt, "learning_rate_decay_start", -1)
    if epoch > start and start >= 0:
        frac = int((epoch - start) / opt.learning_rate_decay_every)
        return opt.learning_rate * (opt.learning_rate_decay_rate ** frac)
    return opt.learning_rate


def ss_prob_for_epoch(opt, epoch, current=0.0):
    """starttrain.py:96-100: scheduled-sampling ramp (kept at its last value before the start epoch)."""
    start = getattr(opt, "scheduled_sampling_start", -1)
    if epoch > start and start >= 0:
        frac = int((epoch - start) / opt.scheduled_sampling_increase_every)
        return min(opt.scheduled_sampling_increase_prob * frac, opt.scheduled_sampling_max_prob)
    return current


def sc_flag_for_epoch(opt, epoch):
    """starttrain.py:102-106."""
    after = getattr(opt, "self_critical_after", -1)
    return after != -1 and epoch >= after


def get_self_critical_reward(model, feat1, feat2, feat_mask, pos_feat, gen_result, scorer, greedy_res=None):
    """myutils.get_self_critical_reward (myutils.py:41-77): greedy baseline rollout (model stays in whatever mode it
    is in -- the reference never leaves train mode, starttrain.py:68), scores = scorer(sampled (m,n), greedy (m,n'))
    -> (2m,) array, reward = score[:m] - score[m:], repeated over the n positions (:75-76).  ``greedy_res`` may be
    handed in when the baseline rollout already ran (scst_rollouts)."""
    if greedy_res is None:
        with torch.no_grad():
            greedy_res, _ = model.sample(feat1, feat2, feat_mask, pos_feat, {"sample_max": 1})
    gen = gen_result.detach().cpu().numpy()
    greedy = greedy_res.detach().cpu().numpy()
    scores = np.asarray(scorer(gen, greedy), dtype=np.float64)
    m = gen.shape[0]
    diff = scores[:m] - scores[m:]
    return np.repeat(diff[:, np.newaxis], gen.shape[1], 1)


def decode_sequence(ix_to_word, seq):
    """myutils.decode_sequence (myutils.py:88-102): one string per row, the words of the tokens up to (not including) the
    first token <= 0, joined by single spaces.  ``ix_to_word`` maps int -> str (the reference also accepts str keys from
    its json vocabulary; both are tried)."""
    rows = seq.detach().cpu().tolist() if torch.is_tensor(seq) else np.asarray(seq).tolist()
    out = []
    for row in rows:
        words = []
        for ix in row:
            if ix <= 0:
                break
            words.append(ix_to_word[ix] if ix in ix_to_word else ix_to_word[str(ix)])
        out.append(" ".join(words))
    return out


def eval_split(model, crit, classify_crit, batches, ix_to_word, eval_kwargs=None, gts_of=None, scorer=None):
    """eval_utils.eval_split (eval_utils.py:18-84) on the HIP model surface.  ``batches`` yields dicts with the
    collate fields (data.collate: feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, cap_classes, class_mask as
    CUDA tensors, plus ``image_ids``).  Per batch: eval-mode forward -> language + weight_class * category loss
    (:40-46), then ``model.sample(..., eval_kwargs)`` (greedy, or beam search when eval_kwargs['beam_size'] > 1: :48),
    decoded with the vocabulary (:55).  Returns (mean loss, predictions, lang_stats) like the reference (:84); the
    coco-caption metrics are out of scope: ``lang_stats = scorer(captions, references)`` when a scorer and ``gts_of``
    (image id -> references) are supplied and eval_kwargs['language_eval'] == 1, else None.  The model is put back into
    train mode at the end, as the reference does (:83).  ``verbose`` defaults to True as in the reference (:18).
    Note (SAModel.forward): the HIP forward always runs all T = seq.size(1) steps; the reference leaves its loop at the first
    all-zero token column (SAModel.py:103), which its own collate never produces."""
    kw = dict(eval_kwargs or {})
    weight_class = kw.get("weight_class", 0.0)
    model.eval()
    loss_sum, loss_evals = 0.0, 1e-8                                               # :28-29
    predictions, gts = [], []
    with torch.no_grad():
        for b in batches:
            out, category = model(b["feats_rgb"], b["feats_opfl"], b["feat_mask"], b["pos_feats"], b["seq"], b["seq_mask"])
            loss = float(crit(out, b["seq"], b["seq_mask"])) + \
                weight_class * float(classify_crit(category, b["cap_classes"], b["seq_mask"], b["class_mask"]))
            loss_sum += loss
            loss_evals += 1
            seq, seq_logprobs = model.sample(b["feats_rgb"], b["feats_opfl"], b["feat_mask"], b["pos_feats"], kw)
            sents = decode_sequence(ix_to_word, seq)
            lp = seq_logprobs.detach().cpu()
            for k, sent in enumerate(sents):
                image_id = b["image_ids"][k] if "image_ids" in b else len(predictions)
                predictions.append({"image_id": image_id, "caption": sent, "seqLogprob": lp[k].numpy()})
                if gts_of is not None:
                    gts.append(gts_of[image_id])
    if kw.get("verbose", True):                                                    # :18 (default True, like the reference)
        for x in predictions[:10]:
            print("image %s: %s" % (x["image_id"], x["caption"]))
    lang_stats = None
    if kw.get("language_eval", 0) == 1 and scorer is not None and gts_of is not None:
        # :66-76: references are the 'tokenized' field of each ground-truth entry; entries without one are dropped
        # (plain strings are taken as they are: callers that already hold tokenized references)
        gts_ = [[(i["tokenized"] if isinstance(i, dict) else i) for i in x if not isinstance(i, dict) or "tokenized" in i]
                for x in gts]
        lang_stats = scorer([x["caption"] for x in predictions], gts_)
    model.train()
    return loss_sum / loss_evals, predictions, lang_stats


def _second_bn_update(bns, r0):
    """The reference runs the encoder twice per SCST iteration (two sample() calls), i.e. two momentum updates of the
    BatchNorm running statistics with the same batch statistics s: with keep = 1 - momentum,
    r2 = keep r1 + momentum s = (1 + keep) r1 - keep r0."""
    with torch.no_grad():
        keep = 1.0 - bns[0].momentum
        cur = [t for m in bns for t in (m.running_mean, m.running_var)]
        old = [t for pair in r0 for t in pair]
        torch._foreach_mul_(cur, 1.0 + keep)                        # two multi-tensor launches for the four buffers
        torch._foreach_add_(cur, old, alpha=-keep)
        for m in bns:
            if m.num_batches_tracked is not None:
                m.num_batches_tracked += 1


def scst_rollouts(model, feat1, feat2, feat_mask, pos_feat, overlap=True, mode=None, uniforms=None, trim=True):
    """The two rollouts of one SCST iteration (starttrain.py:131 + myutils.py:45-48): the sampled rollout (keeps its
    activations for the policy-gradient backward) and the greedy baseline.  They are independent given the batch:
      mode "batched" (default): ONE pass over 2m rows (rows [0,m) sample, rows [m,2m) greedy; SAModel.sample_pair) --
          every per-step launch streams the decoder weights once for both rollouts;
      mode "streams": two m-row rollouts on two streams;   mode "sequential": the reference's order.
    trim=False (batched mode): returns (gen (m,L), slp (m,L), greedy (m,L), n (2,) device int32) with NO host sync -- feed
    them to RewardCriterion(..., n=n[0]); the reference's trimmed views are gen[:, :n[0]] etc.
    All three give the reference's results: both rollouts see the same batch statistics (BatchNorm's input does not
    depend on dropout, and statistics of a repeated batch equal those of the batch), and the running statistics receive
    the reference's TWO momentum updates (the second one is reconstructed exactly)."""
    if mode is None:
        mode = "batched" if overlap else "sequential"
    s_opt = {"sample_max": 0}
    if uniforms is not None:
        s_opt["uniforms"] = uniforms
    if mode == "sequential" or not model.training:
        gen, slp = model.sample(feat1, feat2, feat_mask, pos_feat, s_opt)
        with torch.no_grad():
            greedy, _ = model.sample(feat1, feat2, feat_mask, pos_feat, {"sample_max": 1})
        return gen, slp, greedy
    bns = [model.two_spatial_encoder.visual_emb_rgb[1], model.two_spatial_encoder.visual_emb_opfl[1]]
    if mode == "batched":                                           # (the BatchNorm bookkeeping of the pair is sample_pair's)
        gen, slp, greedy, n = model.sample_pair(feat1, feat2, feat_mask, pos_feat, s_opt)
        if not trim:                                                # no host sync at all: full-width tensors + device-side n
            return gen, slp, greedy, n
        ns = n.cpu()                                                # ONE host sync for both rollouts
        n_s, n_g = int(ns[0]), int(ns[1])
        return gen[:, :n_s], slp[:, :n_s], greedy[:, :n_g]
    r0 = [(m.running_mean.clone(), m.running_var.clone()) for m in bns]
    main = torch.cuda.current_stream()
    from .train import shared_stream
    side = shared_stream("rollout")              # (one side stream per process: see train.shared_stream)
    side.wait_stream(main)
    with torch.cuda.stream(side), torch.no_grad():
        g_seq, _, g_n = model.sample(feat1, feat2, feat_mask, pos_feat, {"sample_max": 1, "async": True, "bn_update": False})
    s_seq, s_slp, s_n = model.sample(feat1, feat2, feat_mask, pos_feat, dict(s_opt, **{"async": True}))
    main.wait_stream(side)
    for t in (g_seq, g_n):                                          # allocated under the side stream, consumed on main
        t.record_stream(main)
    _second_bn_update(bns, r0)                                      # the baseline's (second) running-stat update
    ns = torch.stack([s_n.reshape(()), g_n.reshape(())]).cpu()      # ONE host sync for both rollouts
    n_s, n_g = int(ns[0]), int(ns[1])
    return s_seq[:, :n_s], s_slp[:, :n_s], g_seq[:, :n_g]


class Trainer:
    """One object = the body of ``train(opt)`` (starttrain.py:19-242) without the data loader."""

    def __init__(self, model, opt, reward_scorer=None):
        self.model, self.opt, self.scorer = model, opt, reward_scorer
        self.crit, self.classify_crit, self.rl_crit = LanguageModelCriterion(), ClassiferCriterion(), RewardCriterion()
        self.optimizer = ClipAdam(model, lr=opt.learning_rate, weight_decay=getattr(opt, "weight_decay", 0.0),
                                  grad_clip=getattr(opt, "grad_clip", 0.1),
                                  overlap=getattr(opt, "overlap_update", True) and next(model.parameters()).is_cuda,
                                  fused_zero=getattr(opt, "fused_zero_grad", True))
        self.iteration, self.epoch = 1, 0
        self.sc_flag = False
        self.best_val_score = None
        self.patience = 0
        # the fused loss path (model.xe_loss: no (m,T,V) log-prob tensor, no gradient of it) is the benchmarked one and the
        # default; opt.fused_xe_loss = False gives the reference's call sequence model() + two criteria, same numbers
        self.fused = getattr(opt, "fused_xe_loss", True)
        self.grad_sync = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1 and next(model.parameters()).is_cuda:
            self.grad_sync = GradSync(model)        # all-reduce of the non-encoder gradients under the encoder backward

    def start_epoch(self, epoch):
        """the update_lr_flag block, starttrain.py:85-107"""
        self.epoch = epoch
        self.opt.current_lr = lr_for_epoch(self.opt, epoch)
        self.optimizer.set_lr(self.opt.current_lr)
        self.model.ss_prob = ss_prob_for_epoch(self.opt, epoch, self.model.ss_prob)
        self.sc_flag = sc_flag_for_epoch(self.opt, epoch)

    def train_batch(self, b):
        """starttrain.py:123-137.  b: dict with feat1, feat2, feat_mask, pos_feat, cap, cap_mask, cap_classes, class_mask."""
        model, opt = self.model, self.opt
        self.optimizer.zero_grad()                                                           # :123
        info = {}
        if not self.sc_flag:
            wc = getattr(opt, "weight_class", 0.0)
            if self.fused:
                loss = model.xe_loss(b["feat1"], b["feat2"], b["feat_mask"], b["pos_feat"], b["cap"], b["cap_mask"],
                                     b.get("cap_classes"), b.get("class_mask"), wc)
            else:
                out, category = model(b["feat1"], b["feat2"], b["feat_mask"], b["pos_feat"], b["cap"], b["cap_mask"])   # :125
                loss_language = self.crit(out, b["cap"], b["cap_mask"])                      # :126
                loss_classify = self.classify_crit(category, b["cap_classes"], b["cap_mask"], b["class_mask"])       # :127
                loss = loss_language + wc * loss_classify                                    # :129
                info.update(loss_language=loss_language, loss_classify=loss_classify)
        else:
            mode = getattr(opt, "scst_rollout_mode", None)
            if mode in (None, "batched") and model.training:
                # full-width rollout outputs + the device-side early-exit widths: the only host sync of the iteration is the
                # scorer's own copy of the tokens (the reference syncs every decoder step, SAModel.py:206)
                gen_f, slp_f, greedy_f, n = scst_rollouts(model, b["feat1"], b["feat2"], b["feat_mask"], b["pos_feat"],
                                                          mode="batched", trim=False)                      # :131 + myutils.py:45
                tok = torch.cat([gen_f, greedy_f]).cpu()                                                    # (one copy, one sync)
                ns = n.cpu()
                n_s, n_g = int(ns[0]), int(ns[1])
                m = gen_f.shape[0]
                gen_result, greedy = tok[:m, :n_s], tok[m:, :n_g]
                reward = get_self_critical_reward(model, b["feat1"], b["feat2"], b["feat_mask"], b["pos_feat"], gen_result,
                                                  self.scorer, greedy_res=greedy)            # :132
                rew = torch.zeros(m, slp_f.shape[1], dtype=torch.float32)
                rew[:, :n_s] = torch.from_numpy(np.ascontiguousarray(reward)).float()
                loss = self.rl_crit(slp_f, gen_f, rew.to(slp_f.device), n=n[:1])              # :133
            else:
                gen_result, sample_logprobs, greedy = scst_rollouts(model, b["feat1"], b["feat2"], b["feat_mask"], b["pos_feat"],
                                                                    mode=mode)               # :131 + myutils.py:45
                reward = get_self_critical_reward(model, b["feat1"], b["feat2"], b["feat_mask"], b["pos_feat"], gen_result,
                                                  self.scorer, greedy_res=greedy)            # :132
                loss = self.rl_crit(sample_logprobs, gen_result,
                                    torch.from_numpy(reward).float().to(sample_logprobs.device))  # :133
            info["avg_reward"] = float(np.mean(reward[:, 0])) if reward.size else 0.0
        if self.grad_sync is not None:
            self.grad_sync.arm()
        self.optimizer.arm()                                                                 # (update overlapped with the backward's tail)
        loss.backward()                                                                      # :134
        allreduce_gradients(model)                                                           # data parallel only
        self.optimizer.step()                                                                # :136-137 (clamp + Adam)
        self.iteration += 1
        info["loss"] = loss
        return info

    # ------------------------------------------------------------ checkpoints (starttrain.py:201-231, :29-66)
    def save_checkpoint(self, path, val_score=None, tag=""):
        os.makedirs(path, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(path, "model%s.pth" % tag))
        infos = dict(iter=self.iteration, epoch=self.epoch, best_val_score=self.best_val_score, opt=vars(self.opt),
                     val_score=val_score)
        torch.save(infos, os.path.join(path, "infos%s.pkl" % tag))

    def update_best(self, path, current_score):
        """best-checkpoint + patience bookkeeping, starttrain.py:194-238; returns True when training should stop."""
        if self.best_val_score is None or current_score > self.best_val_score:
            self.best_val_score = current_score
            self.patience = 0
            self.save_checkpoint(path, current_score, tag="-best")
            return False
        self.patience += 1
        return self.patience >= getattr(self.opt, "patience", 1 << 30)

    @staticmethod
    def resume(model, path, tag="-best", strict=True):
        """--start_from: state_dict only (the reference does not save optimizer state)."""
        model.load_state_dict(torch.load(os.path.join(path, "model%s.pth" % tag)), strict=strict)
        return torch.load(os.path.join(path, "infos%s.pkl" % tag), weights_only=False)
