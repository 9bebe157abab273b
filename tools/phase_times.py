#!/usr/bin/env python3
"""Unprofiled phase timing of one training iteration at config 2 with stream events (no rocprof host overhead)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from controllable_xgating_amd import SAModel, make_opt
from controllable_xgating_amd.train import ClipAdam

cfg = dict(B=128, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])).cuda(); model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, "cuda")
opt = ClipAdam(model)
args = (x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])

def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def enc():
    with torch.no_grad(): model.encode(x["feats_rgb"], x["feats_opfl"], x["feat_mask"])
def fwd():
    with torch.no_grad(): model.xe_loss(*args)
def fwdbwd():
    opt.zero_grad(); model.xe_loss(*args).backward()
def full():
    opt.zero_grad(); model.xe_loss(*args).backward(); opt.step()
def rollout_greedy():
    with torch.no_grad(): model.sample(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], {"sample_max": 1})
print("encoder fwd            %.3f ms" % timeit(enc))
print("forward (enc+dec+loss) %.3f ms" % timeit(fwd))
print("forward + backward     %.3f ms" % timeit(fwdbwd))
print("full iteration         %.3f ms" % timeit(full))
print("greedy rollout B=128   %.3f ms" % timeit(rollout_greedy, 5))
