#!/usr/bin/env python3
"""In-kernel stamps (-DSK_TRACE build) of the FIRST launch of a rollout step at 128 rows (4 jobs x 256 workgroups: S2' | p | gate | zero)
inside a paired SCST rollout: with the token choice as the gate tiles' prologue (default) or as its own launch (XG_NO_STEP_SELECT=1,
needs a -DXG_DIAG -DSK_TRACE build).  XG_LIBRARY=controllable_xgating_amd/lib/libxgate_hip_sktrace.so python tools/r6/sel_trace.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from controllable_xgating_amd import SAModel, make_opt, _native as nv
from controllable_xgating_amd.driver import scst_rollouts
cfg = dict(B=64, K=26, R=512, A=1536, E=468, V=20000, C=14, L=30, F1=1536, F2=1024)
model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])).cuda()
model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, "cuda")
L = nv.lib()
nj, gx = 4, 256
assert L.xg_debug_sk_trace_filter(nj, gx) == 0
for _ in range(3):
    with torch.no_grad():
        scst_rollouts(model, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], trim=False)
torch.cuda.synchronize()
assert L.xg_debug_sk_trace_clear() == 0
with torch.no_grad():
    scst_rollouts(model, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], trim=False)
torch.cuda.synchronize()
n = 4096 * 8
buf = (C.c_longlong * n)()
assert L.xg_debug_sk_trace(buf, n) == 0
h = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
live = (h[:, 5] > h[:, 0]) & (h[:, 6] >= h[:, 0])
h = np.where(live[:, None], h, 0)
t0 = h[live, 0].min()
us = lambda v: (v - t0) * 0.01
print("WGs recorded %d ; span %.2f us" % (live.sum(), us(h[live, 5].max())))
for y in range(nj):
    rows = h[y * gx:(y + 1) * gx]
    rows = rows[rows[:, 5] > 0]
    if not len(rows):
        continue
    seq = rows[:, [0, 6, 1, 2, 3, 7, 4, 5]].astype(np.float64)
    d = np.diff(seq, axis=1) * 0.01
    print("job %d: %3d tiles | entry %.2f..%.2f | exit %.2f..%.2f (median %.2f) | mean phases: desc(+select) %.2f prologue %.2f first %.2f "
          "kloop(w0) %.2f skew(last wave) %.2f reduce %.2f epilogue %.2f" % (
              y, len(rows), us(rows[:, 0].min()), us(rows[:, 0].max()), us(rows[:, 5].min()), us(rows[:, 5].max()),
              us(np.median(rows[:, 5])), *d.mean(0)))
    if len(rows) == 64:
        print("   gate tiles, desc(+select) phase by tile (us):", np.round(d[:, 0], 2).tolist())
