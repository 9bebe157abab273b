"""Beam search on top of the single-step HIP entry point.

Restates the control flow of the reference's ``SAModel.sample_beam`` (caption_src/SAModel.py:129-161) and
``CaptionModel.beam_search`` (caption_src/CaptionModel.py:22-128): UNK (index 1) suppressed by -1000 each step
(:94), candidates = top ``beam_size`` words of each live beam (only beam 0 at t = 0, :41-42), stable sort by
cumulative log-prob (:50), finished beams (token 0 or last step) copied to ``done_beams`` and their running sum
set to -1000 (:110-118), best finished beam returned.

Unlike the reference (one video at a time, batch = beam_size, full-vocabulary sort on the CPU) all videos advance
together: rows = (video, beam), ONE ``xg_step_fwd`` launch group per time step for the whole batch, a device top-k,
and a tiny host-side merge per video.  The selection rule is the reference's, so results are identical.
"""
from __future__ import annotations

import torch


def sample_beam(model, feats, feat_masks, pos_feats, opt={}):
    beam_size = opt.get("beam_size", 5)
    B, K, R = feats.shape
    L = model.seq_length
    assert beam_size <= model.vocab_size, "lets assume this for now"          # SAModel.py:134
    dev = feats.device
    rows = B * beam_size
    feats_rep = feats.detach().float().repeat_interleave(beam_size, 0).contiguous()
    mask_rep = feat_masks.detach().float().repeat_interleave(beam_size, 0).contiguous()
    pos_rep = pos_feats.detach().float().repeat_interleave(beam_size, 0).contiguous()

    state = model.init_hidden(feats_rep, mask_rep)                              # SAModel.py:147
    it = torch.zeros(rows, dtype=torch.int64, device=dev)                       # <bos>
    logprobs, state = model.get_logprobs_state(it, feats_rep, pos_rep, state)   # SAModel.py:150-154

    beam_seq = torch.zeros(B, L, beam_size, dtype=torch.int64)
    beam_seq_logprobs = torch.zeros(B, L, beam_size)
    beam_sum = torch.zeros(B, beam_size)
    done = [[] for _ in range(B)]

    for t in range(L):
        lp = logprobs.float().clone()
        lp[:, 1] -= 1000.0                                                      # CaptionModel.py:94
        ys, ix = torch.topk(lp, beam_size, dim=1)                               # == first beam_size columns of the sort (:39)
        ys, ix = ys.cpu().view(B, beam_size, beam_size), ix.cpu().view(B, beam_size, beam_size)
        src = torch.empty(rows, dtype=torch.int64)
        for k in range(B):
            nrows = 1 if t == 0 else beam_size                                  # :41-42
            cands = []
            for c in range(beam_size):                                          # :43-49
                for q in range(nrows):
                    local = float(ys[k, q, c])
                    cands.append((int(ix[k, q, c]), q, float(beam_sum[k, q]) + local, local))
            cands.sort(key=lambda v: -v[2])                                     # stable, like sorted() at :50
            if t >= 1:
                prev_seq = beam_seq[k, :t].clone()
                prev_lp = beam_seq_logprobs[k, :t].clone()
            for vix in range(beam_size):                                        # :60-73
                c, q, p, r = cands[vix]
                if t >= 1:
                    beam_seq[k, :t, vix] = prev_seq[:, q]
                    beam_seq_logprobs[k, :t, vix] = prev_lp[:, q]
                src[k * beam_size + vix] = k * beam_size + q
                beam_seq[k, t, vix] = c
                beam_seq_logprobs[k, t, vix] = r
                beam_sum[k, vix] = p
            for vix in range(beam_size):                                        # :108-118
                if int(beam_seq[k, t, vix]) == 0 or t == L - 1:
                    done[k].append(dict(seq=beam_seq[k, :, vix].clone(), logps=beam_seq_logprobs[k, :, vix].clone(),
                                        p=float(beam_sum[k, vix])))
                    beam_sum[k, vix] = -1000.0
        if t == L - 1:
            break
        src_d = src.to(dev)
        state = [tuple(s.index_select(1, src_d) for s in layer) for layer in state]   # :66-69
        it = beam_seq[:, t, :].reshape(-1).to(dev)
        logprobs, state = model.get_logprobs_state(it, feats_rep, pos_rep, state)      # :125

    seq = torch.zeros(L, B, dtype=torch.int64)
    seq_logprobs = torch.zeros(L, B)
    model.done_beams = []
    for k in range(B):
        best = sorted(done[k], key=lambda v: -v["p"])[:beam_size]               # :127
        model.done_beams.append(best)
        seq[:, k] = best[0]["seq"]
        seq_logprobs[:, k] = best[0]["logps"]
    return seq.transpose(0, 1), seq_logprobs.transpose(0, 1)                    # SAModel.py:161
