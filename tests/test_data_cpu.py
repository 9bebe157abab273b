"""Batch conventions checked against the reference's own helpers where they are importable here
(data_io.get_sub_frames / collate_fn are plain functions) and against hand-worked layouts.  CPU only; the reference is
imported only when /root/reference exists (the GPU box and later containers skip that half)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from controllable_xgating_amd.data import collate, get_sub_frames, make_video_item

REF = "/root/reference/caption_src"


def _items(rng):
    items = []
    for n_frames, cap in ((40, [5, 9, 3, 7]), (11, [4, 2]), (26, [8, 6, 2, 9, 11, 3])):
        f1 = rng.random((n_frames, 6)).astype(np.float32); f2 = rng.random((n_frames, 4)).astype(np.float32)
        pos = rng.random((len(cap) + 1, 5)).astype(np.float32)
        a, b, m, p = make_video_item(f1, f2, pos, 26)
        items.append(dict(cap=cap, cap_class=[c % 3 for c in cap], class_mask=[1] * len(cap), feat1=a, feat2=b, feat_mask=m,
                          pos_feat=p, raw=(f1, f2, pos)))
    return items


def test_subsample_pad_and_mask():
    rng = np.random.default_rng(0)
    long = rng.random((40, 6)).astype(np.float32)
    short = rng.random((11, 6)).astype(np.float32)
    s = get_sub_frames(long, 26)
    assert s.shape == (26, 6) and np.array_equal(s[0], long[0]) and np.array_equal(s[13], long[int(13 * 40 / 26)])
    p = get_sub_frames(short, 26)
    assert np.array_equal(p[:11], short) and not p[11:].any()
    f1, f2, mask, pos = make_video_item(short, rng.random((11, 4)).astype(np.float32), rng.random((3, 5)).astype(np.float32), 26)
    assert mask.shape == (1, 26) and mask[0, :11].all() and not mask[0, 11:].any()


def test_collate_layout():
    items = _items(np.random.default_rng(1))
    caps, caps_mask, cap_classes, class_masks, feats1, feats2, feat_mask, pos_feat = collate(items)
    assert caps.shape == (3, 7) and caps[:, 0].eq(0).all()                       # BOS column
    assert caps[0].tolist() == [0, 8, 6, 2, 9, 11, 3]                           # sorted longest first
    assert caps_mask.sum(1).tolist() == [7.0, 5.0, 3.0]                          # BOS + n words
    assert class_masks[1].tolist() == [1, 1, 1, 1, 1, 0, 0]                      # n ones + the extra 1 (data_io.py:360)
    assert feats1.shape == (3, 26, 6) and feat_mask.shape == (3, 26) and pos_feat.shape == (3, 5)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_against_reference_helpers():
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.path.insert(0, REF)
    sys.argv = ["x"]
    import data_io as ref
    rng = np.random.default_rng(2)
    for n in (5, 26, 27, 100):
        fr = rng.random((n, 8)).astype(np.float32)
        assert np.array_equal(ref.get_sub_frames(fr, 26), get_sub_frames(fr, 26))
    items = _items(rng)
    batch = [("vid%d_0" % i, it["cap"], it["cap_class"], it["class_mask"], it["feat1"], it["feat2"], it["feat_mask"],
              it["pos_feat"], np.zeros((1, 3))) for i, it in enumerate(items)]
    out = ref.collate_fn(list(batch))
    mine = collate(items)
    for a, b in zip((out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[8]), mine):
        assert torch.equal(a.float(), b.float())
