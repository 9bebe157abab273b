#!/usr/bin/env python3
"""Host-side enqueue cost of one training iteration (the GPU queue is empty when each measurement starts)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from controllable_xgating_amd import SAModel, make_opt
from controllable_xgating_amd.train import ClipAdam

cfg = dict(bench.CFG2) if hasattr(bench, "CFG2") else dict(B=128, K=26, L=20, V=20000, R=512, F1=1536, F2=1024, C=14)
dev = torch.device("cuda:0")
opt = make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])
model = SAModel(opt).to(dev); model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], seed=0, device=dev)
optim = ClipAdam(model, lr=4e-4, grad_clip=0.1)
def step(timing=None):
    t0 = time.perf_counter()
    optim.zero_grad()
    t1 = time.perf_counter()
    loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    optim.step()
    t4 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(5):
    step()
torch.cuda.synchronize()
rows = []
for _ in range(5):
    step(rows)
    torch.cuda.synchronize()
for r in rows:
    print("host enqueue: zero_grad %.3f ms | forward %.3f ms | backward %.3f ms | optimizer %.3f ms | total %.3f ms" % tuple([v * 1e3 for v in r] + [sum(r) * 1e3]))

# ---- raw C-call durations (host side) inside one iteration
from controllable_xgating_amd import _native as nv
lib = nv.lib()
class Timed:
    def __init__(self, f, name): self.f, self.name, self.t = f, name, []
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.f(*a); self.t.append(time.perf_counter() - t0); return r
wrapped = {}
for name in ("xg_xe_loss_fwd", "xg_xe_loss_bwd", "xg_clip_adam"):
    wrapped[name] = Timed(getattr(lib, name), name)
class LibProxy:
    def __getattr__(self, n): return wrapped.get(n) or getattr(lib, n)
proxy = LibProxy()
nv.lib = lambda: proxy
import controllable_xgating_amd.model as M, controllable_xgating_amd.train as TR
for _ in range(3):
    step(); torch.cuda.synchronize()
for n, w in wrapped.items():
    print("C call %-16s host time: %s ms" % (n, ", ".join("%.3f" % (v * 1e3) for v in w.t)))
