"""Batch conventions either side of the hot path: how the reference turns per-video arrays into the
tensors ``SAModel.forward`` receives.  Restated from caption_src/data_io.py:27-37 (frame subsampling / padding),
:213 (frame mask from non-zero rows), :215-217 (POS vector = last row of the tagger's states) and :330-374
(collate: BOS column, masks, class targets).  Host-side numpy/torch only -- no HDF5 / pickle readers (the dataset files
are not part of the reference tree)."""
from __future__ import annotations

import numpy as np
import torch


def get_sub_frames(frames: np.ndarray, K: int) -> np.ndarray:
    """data_io.py:27-37: fewer than K frames -> zero-pad at the end; otherwise K frames at np.linspace(0, n, K) indices."""
    n = len(frames)
    if n < K:
        return np.concatenate((frames, np.zeros([K - n, frames.shape[1]], dtype=frames.dtype)), axis=0)
    index = np.linspace(0, n, K, endpoint=False, dtype=int)
    return frames[index]


def make_video_item(feat_rgb: np.ndarray, feat_opfl: np.ndarray, pos_states: np.ndarray, K: int):
    """data_io.py:204-217 -> feat1 (K,F1), feat2 (K,F2), feat_mask (1,K), pos_feat (R,)."""
    f1 = torch.from_numpy(np.ascontiguousarray(get_sub_frames(feat_rgb, K))).float()
    f2 = torch.from_numpy(np.ascontiguousarray(get_sub_frames(feat_opfl, K))).float()
    mask = (torch.sum(f1.view(f1.size(0), -1), dim=1, keepdim=True) != 0).float().transpose(1, 0)   # :213
    pos = torch.from_numpy(np.ascontiguousarray(pos_states[-1])).float()                             # :215-217
    return f1, f2, mask, pos


def collate(items):
    """data_io.py:330-374.  items: list of dicts with keys cap (list of word ids >= 2), cap_class, class_mask (lists),
    feat1, feat2, feat_mask (1,K), pos_feat.  Sorted by caption length (longest first, :331); returns the eight tensors
    of starttrain.py:114-121: caps (m,L+1) int64 with column 0 = BOS = 0, caps_mask, cap_classes, class_masks, feats1,
    feats2, feat_mask (m,K), pos_feat."""
    items = sorted(items, key=lambda it: len(it["cap"]), reverse=True)
    max_len = len(items[0]["cap"])
    m = len(items)
    caps = torch.zeros(m, max_len + 1, dtype=torch.int64)
    caps_mask = torch.zeros(m, max_len + 1)
    cap_classes = torch.zeros(m, max_len + 1, dtype=torch.int64)
    class_masks = torch.zeros(m, max_len + 1)
    for i, it in enumerate(items):
        n = len(it["cap"])
        caps[i, 1:n + 1] = torch.as_tensor(it["cap"], dtype=torch.int64)                  # :344
        caps_mask[i, :n + 1] = 1                                                           # :346
        nc = len(it["cap_class"])
        cap_classes[i, :nc] = torch.as_tensor(it["cap_class"], dtype=torch.int64)         # :356
        ncm = len(it["class_mask"])
        class_masks[i, :ncm] = torch.as_tensor(it["class_mask"], dtype=torch.float32)     # :359
        class_masks[i, ncm] = 1                                                            # :360
    feats1 = torch.stack([it["feat1"] for it in items], 0)
    feats2 = torch.stack([it["feat2"] for it in items], 0)
    feat_mask = torch.cat([it["feat_mask"] for it in items], 0)
    pos_feat = torch.stack([it["pos_feat"] for it in items], 0)
    return caps, caps_mask, cap_classes, class_masks, feats1, feats2, feat_mask, pos_feat
