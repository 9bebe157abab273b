"""Batch conventions either side of the hot path: how the reference turns per-video arrays into the
tensors ``SAModel.forward`` receives.  Restated from caption_src/data_io.py:27-37 (frame subsampling / padding),
:213 (frame mask from non-zero rows), :215-217 (POS vector = last row of the tagger's states) and :330-374
(collate: BOS column, masks, class targets), and the on-disk side: the pickled split / caption / vocabulary / POS-category
files and the per-video HDF5 feature stores (data_io.py:51-126 word -> category ids, :128-219 and :222-328 the train / test
datasets).  Host-side numpy/torch only.  ``h5py`` is optional: ``open_feature_store`` imports it lazily; every reader takes
any mapping ``vid -> array-like`` (an ``h5py.File`` is one), so the logic is testable without it.  The POS generator that
writes ``postagsequence.hdf5`` (``pos_src/``) stays out of scope."""
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


# ---------------------------------------------------------------------------------------------------- on-disk side
def load_pkl(path):
    """data_io.py:19-25."""
    import pickle
    with open(path, "rb") as f:
        return pickle.load(f)


def open_feature_store(path):
    """``h5py.File(path, 'r')`` (data_io.py: feats.hdf5[vid] -> (n_frames, D); postagsequence.hdf5[vid]['states'] ->
    (L + 1, 512)).  Raises ImportError with a clear message when h5py is not installed."""
    try:
        import h5py
    except ImportError as e:                      # pragma: no cover - depends on the image
        raise ImportError("h5py is needed to read the reference's .hdf5 feature files; any mapping vid -> array works too") from e
    return h5py.File(path, "r")


# POS tag -> category id (data_io.py:60-100); every tag not listed is category 1; ids 0 / 1 are <EOS> / unknown.
# 'WRR' is the reference's own spelling in caption_src/data_io.py:64 (pos_src/ has 'WRB'): words tagged WRB are category 1 on
# the training path, and parity with caption_src is the contract (tests/test_data_cpu.py pins the table tag by tag).
_CATEGORY_OF_TAG = {}
for _cid, _tags in ((2, "VB VBD VBP VBG VBN VBZ"), (3, "NN NNS NNP"), (4, "JJ JJR JJS"), (5, "RB RBS RBR WRR EX"), (6, "CC"),
                    (7, "PRP PRP$ WP POS WP$"), (8, "IN TO"), (9, "DT WDT PDT"), (10, "RP MD"), (11, "CD"), (12, "SYM : `` # $"),
                    (13, "UH")):
    for _t in _tags.split():
        _CATEGORY_OF_TAG[_t] = _cid
N_CATEGORIES = 14                                 # data_io.get_nclasses (:125-126)


def word_categories(category_words, words):
    """data_io.filt_word_category (:51-114): ``category_words`` {POS tag: [words]} (category.pkl), ``words`` the vocabulary.
    Returns {word: category id} with '<EOS>' -> 0, '<UNK>' -> 1, words without a tag -> 1.  A word listed under several tags
    takes the LAST one in the file's iteration order, as in the reference (:56-58)."""
    tag_of = {}
    for tag, wordlist in category_words.items():
        for wd in wordlist:
            tag_of[wd] = tag
    out = {wd: (_CATEGORY_OF_TAG.get(tag_of[wd], 1) if wd in tag_of else 1) for wd in words}
    out["<EOS>"] = 0
    out["<UNK>"] = 1
    return out


def vocab_size(worddict):
    """data_io.get_nwords (:122-123): the pickled dictionary starts at index 2; 0 = <EOS> / BOS / pad, 1 = unknown."""
    return len(worddict) + 2


class CaptionDataset:
    """data_io.custom_dset_train / custom_dset_test (:128-219, :222-328) without torch's Dataset base: the split list
    ``['vid1_0', 'vid1_2', ...]``, captions ``{vid: [{'caption', 'tokenized', ...}]}``, vocabulary ``{word: id >= 2}`` and POS
    categories, plus three per-video stores (rgb features, optical-flow features, POS tagger states).  Captions longer than
    ``seq_length`` tokens (or empty) are dropped (:158); ``test=True`` keeps the first caption of every video only (:266-275).
    Item = the dict ``collate`` takes (+ 'id' and 'gts', the video's numbered reference captions, longest first, zero padded)."""

    def __init__(self, split_ids, caps, worddict, category_words, feats_rgb, feats_opfl, pos_states, K, seq_length, test=False):
        self.K = K
        wtoi = dict(worddict)
        wtoi["<EOS>"] = 0
        wtoi["UNK"] = 1                           # (sic, :139) -- unknown words map to 1 through the `else 1` below
        self.wtoi = wtoi
        self.category = word_categories(category_words, wtoi)
        ids, items = [], []
        for ID in split_ids:
            vid, capid = ID.split("_")
            cap = caps[vid][int(capid)]
            token = cap["tokenized"].split()
            if not 0 < len(token) <= seq_length:
                continue
            cats = [self.category.get(wd, 1) for wd in token]
            ids.append(ID)
            items.append(dict(caption=cap["caption"], tokenized=cap["tokenized"], numbered=[wtoi.get(wd, 1) for wd in token],
                              category=cats, category_mask=[1 if 0 <= c < N_CATEGORIES else 0 for c in cats]))
        if test:
            seen, keep = set(), []
            for i, ID in enumerate(ids):
                vid = ID.split("_")[0]
                if vid not in seen:
                    seen.add(vid)
                    keep.append(i)
            ids, items = [ids[i] for i in keep], [items[i] for i in keep]
        self.gts = []
        for ID in ids:
            vid = ID.split("_")[0]
            numbered = sorted(([wtoi.get(wd, 1) for wd in c["tokenized"].split()] for c in caps[vid]), key=len, reverse=True)
            arr = np.zeros([len(numbered), len(numbered[0])], dtype=int)
            for r, row in enumerate(numbered):
                arr[r, :len(row)] = row
            self.gts.append(arr)
        self.ids, self.caps = ids, items
        self.feats_rgb, self.feats_opfl, self.pos_states = feats_rgb, feats_opfl, pos_states

    def itow(self):
        """index -> word (custom_dset_*.get_itow)."""
        return {v: k for k, v in self.wtoi.items()}

    def __len__(self):
        return len(self.caps)

    def __getitem__(self, index):
        ID, cap = self.ids[index], self.caps[index]
        vid = ID.split("_")[0]
        f1, f2, mask, pos = make_video_item(np.asarray(self.feats_rgb[vid][:]), np.asarray(self.feats_opfl[vid][:]),
                                            np.asarray(self.pos_states[vid]["states"][:]), self.K)
        return dict(id=ID, cap=cap["numbered"], cap_class=cap["category"], class_mask=cap["category_mask"], feat1=f1, feat2=f2,
                    feat_mask=mask, pos_feat=pos, gts=self.gts[index])

    @classmethod
    def from_files(cls, split_pkl, cap_pkl, worddict_pkl, category_pkl, feats_rgb_h5, feats_opfl_h5, pos_h5, K, seq_length, test=False):
        """The reference's file layout (data_io.py:131-137 arguments; .hdf5 paths are opened with h5py)."""
        return cls(load_pkl(split_pkl), load_pkl(cap_pkl), load_pkl(worddict_pkl), load_pkl(category_pkl),
                   open_feature_store(feats_rgb_h5), open_feature_store(feats_opfl_h5), open_feature_store(pos_h5), K, seq_length, test)
