#!/usr/bin/env python3
"""In-kernel phase timeline of skinny launches INSIDE the training iteration (library built with -DSK_TRACE, XG_LIBRARY=...):
the last launch of the iteration that matches (jobs, grid width, threads) stays in the trace buffer.
SK_SHAPES="jobs,gx,threads ..." e.g. "1,256,256" = launch B of the reverse-time loop, "2,256,256" = launch A / the encoder's
backward recurrence, "1,128,256" = chain 1."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from controllable_xgating_amd import SAModel, make_opt, _native as nv
from controllable_xgating_amd.train import ClipAdam
cfg = dict(B=128, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])).cuda()
model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, "cuda")
optim = ClipAdam(model, lr=4e-4, grad_clip=0.1, overlap=True, fused_zero=True)
def step():
    optim.zero_grad()
    loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
    optim.arm(); loss.backward(); optim.step()
L = nv.lib()
for _ in range(3):
    step()
shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("SK_SHAPES", "1,256,256 2,256,256 1,128,256").split()]
for (nj, gx, th) in shapes:
    assert L.xg_debug_sk_trace_filter3(nj, gx, th) == 0 and L.xg_debug_sk_trace_clear() == 0
    torch.cuda.synchronize()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 4096 * 8
    buf = (C.c_longlong * n)()
    assert L.xg_debug_sk_trace(buf, n) == 0
    h = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
    live = h[:, 5] > 0
    print("== launches with %d jobs x %d workgroups x %d threads (in the iteration)" % (nj, gx, th))
    if not live.any():
        print("   (no workgroup recorded)"); continue
    t0 = h[live, 0].min()
    us = lambda v: (v - t0) * 0.01
    print("   WGs recorded %d ; span %.2f us" % (live.sum(), us(h[live, 5].max())))
    for y in range(nj):
        rows = h[y * gx:(y + 1) * gx]; rows = rows[rows[:, 5] > 0]
        if not len(rows): continue
        seq = rows[:, [0, 6, 1, 2, 3, 7, 4, 5]].astype(np.float64)
        d = np.diff(seq, axis=1) * 0.01
        print("   job %d: %3d tiles | entry %.2f..%.2f | exit %.2f..%.2f (median %.2f) | mean phases: desc %.2f prologue %.2f first %.2f "
              "kloop(w0) %.2f skew(last wave) %.2f reduce %.2f epilogue %.2f" % (
                  y, len(rows), us(rows[:, 0].min()), us(rows[:, 0].max()), us(rows[:, 5].min()), us(rows[:, 5].max()),
                  us(np.median(rows[:, 5])), *d.mean(0)))
