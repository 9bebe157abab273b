#!/usr/bin/env python3
"""In-kernel phase timeline of the fast skinny kernel for one launch shape of the decoder step (library built with
-DSK_TRACE -DSK_TRACE_NJOBS=<jobs of the launch to look at>): XG_EXTRA_FLAGS="-DSK_TRACE -DSK_TRACE_NJOBS=4" python tools/sk_trace_run.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from controllable_xgating_amd import SAModel, make_opt, _native as nv
cfg = dict(B=128, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])).cuda()
model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, "cuda")
t = bench.measure_step_group(model, x, reps=50)
torch.cuda.synchronize()
print("step group %.2f us" % (t * 1e6))
n = 1024 * 8
buf = (C.c_longlong * n)()
assert nv.lib().xg_debug_sk_trace(buf, n) == 0
h = np.array(buf[:], dtype=np.int64).reshape(1024, 8)
GX = int(os.environ.get("SK_GX", "256"))
live = h[:, 5] > 0
t0 = h[live, 0].min()
us = lambda v: (v - t0) * 0.01
print("WGs recorded %d ; span %.2f us" % (live.sum(), us(h[live, 5].max())))
names = ["entry", "prologue", "first chunk", "k loop", "reduce", "epilogue"]
for y in range(4):
    rows = h[y * GX:(y + 1) * GX]
    rows = rows[rows[:, 5] > 0]
    if not len(rows):
        continue
    d = np.diff(rows[:, :6], axis=1) * 0.01
    print("job %d: %3d tiles | entry %.2f..%.2f | exit %.2f..%.2f | mean phases: prologue %.2f first %.2f kloop %.2f reduce %.2f epilogue %.2f" % (
        y, len(rows), us(rows[:, 0].min()), us(rows[:, 0].max()), us(rows[:, 5].min()), us(rows[:, 5].max()), *d.mean(0)))
