"""TEST INFRASTRUCTURE ONLY -- deterministic weights / inputs / dropout masks.

Counter-based integer hashing (murmur3 finaliser) so that the build container
(where the reference is imported to make the golden fixtures) and the GPU box
regenerate bit-identical weights and inputs without shipping 144 MB of
parameters.  The dropout hash ``keep_mask`` restates, in numpy uint32
arithmetic, the integer hash the HIP kernels use
(``controllable_xgating_amd/csrc/xg_common.h: xg_keep``) so dropout parity is
testable at p > 0.

Shapes follow the reference's ``state_dict`` (SURVEY.md Appendix B;
reference caption_src/SAModel.py:14-50, caption_src/sub_modules.py:86-110,
648-669, 738-745).
"""
from __future__ import annotations

import collections
import zlib

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    """murmur3 32-bit finaliser on a uint64 array holding 32-bit values."""
    h = h & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def hash_u32(seed: int, site: int, step: int, n: int, offset: int = 0):
    """32-bit hash of (seed, site, step, idx) for idx in [offset, offset+n).

    Must stay identical to xg_hash() in csrc/xg_common.h.
    """
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    a = (idx + np.uint64(0x9E3779B9) * np.uint64((step + 1) & 0xFFFFFFFF)) & M32
    h = fmix32(a)
    k = np.uint64((seed ^ ((site * 0x632BE5AB) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = fmix32(h ^ k)
    return h.astype(np.uint32)


def keep_mask(seed: int, site: int, step: int, shape, p: float):
    """Inverted-dropout multiplier (0 or 1/(1-p)) as float32, shape `shape`."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return np.ones(shape, dtype=np.float32)
    thresh = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    h = hash_u32(seed, site, step, n)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(h >= thresh, scale, np.float32(0.0)).astype(np.float32).reshape(shape)


def uniform(name: str, shape, seed: int, lo: float = 0.0, hi: float = 1.0):
    """float32 U[lo,hi) array determined only by (name, shape, seed)."""
    n = int(np.prod(shape))
    site = zlib.crc32(name.encode()) & 0xFFFFFFFF
    h = hash_u32(seed, site, 0, n).astype(np.uint64)
    u = (h >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def randint(name: str, shape, seed: int, lo: int, hi: int):
    n = int(np.prod(shape))
    site = zlib.crc32(name.encode()) & 0xFFFFFFFF
    h = hash_u32(seed, site, 1, n).astype(np.uint64)
    return (lo + (h % np.uint64(hi - lo))).astype(np.int64).reshape(shape)


# --------------------------------------------------------------------------- dims
Dims = collections.namedtuple(
    "Dims", "B K R A E V C L F1 F2 H")  # H = classifier hidden (128)


def make_dims(B=8, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20,
              F1=1536, F2=1024, H=128):
    return Dims(B, K, R, A, E, V, C, L, F1, F2, H)


def param_shapes(d: Dims):
    """state_dict names -> shapes."""
    R, A, E, V, C = d.R, d.A, d.E, d.V, d.C
    s = collections.OrderedDict()
    enc = "two_spatial_encoder."
    s[enc + "visual_emb_rgb.0.weight"] = (R, d.F1)
    s[enc + "visual_emb_rgb.0.bias"] = (R,)
    s[enc + "visual_emb_rgb.1.weight"] = (R,)
    s[enc + "visual_emb_rgb.1.bias"] = (R,)
    s[enc + "visual_emb_opfl.0.weight"] = (R, d.F2)
    s[enc + "visual_emb_opfl.0.bias"] = (R,)
    s[enc + "visual_emb_opfl.1.weight"] = (R,)
    s[enc + "visual_emb_opfl.1.bias"] = (R,)
    for m in ("rgb", "opfl"):
        s[enc + f"lstmcell_{m}.weight_ih"] = (4 * R, R)
        s[enc + f"lstmcell_{m}.weight_hh"] = (4 * R, R)
        s[enc + f"lstmcell_{m}.bias_ih"] = (4 * R,)
        s[enc + f"lstmcell_{m}.bias_hh"] = (4 * R,)
    for m in ("rgb", "opfl"):
        s[enc + f"gate_{m}.gate.0.weight"] = (R, R)
        s[enc + f"gate_{m}.gate.0.bias"] = (R,)
    s[enc + "fusion.late_fusion.0.weight"] = (R, 2 * R)
    s[enc + "fusion.late_fusion.0.bias"] = (R,)
    for n in ("img_embed_h_1", "img_embed_c_1", "img_embed_h_2", "img_embed_c_2"):
        s[n + ".weight"] = (R, R)
        s[n + ".bias"] = (R,)
    s["lstmcore.gate.gate.0.weight"] = (R, E)
    s["lstmcore.gate.gate.0.bias"] = (R,)
    for cell, in1 in (("lstm_1", E), ("lstm_2", R)):
        s[f"lstmcore.{cell}.i2h.weight"] = (4 * R, in1)
        s[f"lstmcore.{cell}.i2h.bias"] = (4 * R,)
        s[f"lstmcore.{cell}.a2h.weight"] = (4 * R, R)
        s[f"lstmcore.{cell}.a2h.bias"] = (4 * R,)
        s[f"lstmcore.{cell}.h2h.weight"] = (4 * R, R)
        s[f"lstmcore.{cell}.h2h.bias"] = (4 * R,)
    s["lstmcore.v2a.weight"] = (A, R)
    s["lstmcore.v2a.bias"] = (A,)
    s["lstmcore.h2a.weight"] = (A, 2 * R)
    s["lstmcore.h2a.bias"] = (A,)
    s["lstmcore.a2w.weight"] = (1, A)
    s["lstmcore.a2w.bias"] = (1,)
    s["embed.weight"] = (V, E)
    s["logit.weight"] = (V, R)
    s["logit.bias"] = (V,)
    s["classifer.0.weight"] = (d.H, R)
    s["classifer.0.bias"] = (d.H,)
    s["classifer.3.weight"] = (C, d.H)
    s["classifer.3.bias"] = (C,)
    return s


def make_params(d: Dims, seed: int = 1024, logit_gain: float = 8.0):
    """Procedural weights: U(-1/sqrt(fan_in), +1/sqrt(fan_in)) like nn.Linear's
    default bound (reference relies on torch defaults, SAModel.py:38-49), embed
    U(-0.1,0.1) (SAModel.py:54).  ``logit.weight`` is scaled up by
    ``logit_gain`` so greedy top-1/top-2 margins are healthy (SURVEY.md 7.3
    item 4).  BatchNorm affine weights are U(0.5,1.5) so they matter.
    Returns OrderedDict name -> float32 ndarray.
    """
    out = collections.OrderedDict()
    for name, shape in param_shapes(d).items():
        if name.endswith(".1.weight"):          # BN gamma
            out[name] = uniform(name, shape, seed, 0.5, 1.5)
            continue
        if name.endswith(".1.bias"):            # BN beta
            out[name] = uniform(name, shape, seed, -0.2, 0.2)
            continue
        if name == "embed.weight":
            out[name] = uniform(name, shape, seed, -0.1, 0.1)
            continue
        if len(shape) == 2:
            fan_in = shape[1]
        else:                                   # bias: fan_in of its weight
            wshape = param_shapes(d)[name[:-4] + "weight"] if name.endswith("bias") and \
                (name[:-4] + "weight") in param_shapes(d) else None
            if wshape is None:                  # LSTMCell bias_ih / bias_hh
                fan_in = d.R
            else:
                fan_in = wshape[1]
        b = 1.0 / np.sqrt(float(fan_in))
        if name == "logit.weight":
            b *= logit_gain
        out[name] = uniform(name, shape, seed, -b, b)
    return out


def make_inputs(d: Dims, seed: int = 0, ragged: bool = False):
    """Synthetic batch per SURVEY.md 8(d): non-negative features, pos U(-1,1),
    seq[:,0]=0 (BOS), words UniformInt[2,V).  ``ragged`` gives the G4 variant:
    descending caption lengths and padded trailing frames on 3 videos.
    """
    B, K, L = d.B, d.K, d.L
    T = L + 1
    x = {}
    x["feats_rgb"] = uniform("feats_rgb", (B, K, d.F1), seed)
    x["feats_opfl"] = uniform("feats_opfl", (B, K, d.F2), seed)
    x["pos_feats"] = uniform("pos_feats", (B, d.R), seed, -1.0, 1.0)
    feat_mask = np.ones((B, K), dtype=np.float32)
    seq = np.zeros((B, T), dtype=np.int64)
    seq[:, 1:] = randint("seq", (B, L), seed, 2, d.V)
    seq_mask = np.ones((B, T), dtype=np.float32)
    if ragged:
        base = [20, 17, 12, 9, 7, 5, 3, 1]
        for b in range(B):
            n = max(1, int(round(base[b % len(base)] * L / 20.0)))   # words in caption b
            if b == 0:
                n = L                                   # longest fills the width
            seq[b, n + 1:] = 0                          # EOS / pad = 0
            seq_mask[b, n + 1:] = 0.0                   # BOS + n words (data_io.py:346)
        for b in (1, 3, 6):
            if b < B:
                npad = min(6, K - 1)
                feat_mask[b, K - npad:] = 0.0
                x["feats_rgb"][b, K - npad:] = 0.0      # padded rows are zero (data_io.py:27-37)
                x["feats_opfl"][b, K - npad:] = 0.0
    x["feat_mask"] = feat_mask
    x["seq"] = seq
    x["seq_mask"] = seq_mask
    x["cap_classes"] = randint("cap_classes", (B, T), seed, 0, d.C)
    x["class_mask"] = np.ones((B, T), dtype=np.float32)
    return x
