#!/usr/bin/env python3
"""In-kernel phase timeline of the fast skinny kernel for the three launches of the decoder step (library built with -DSK_TRACE:
python -c "import __graft_entry__ as g; g.build_variant('sktrace', ['-DSK_TRACE'])"; XG_LIBRARY=.../libxgate_hip_sktrace.so
python tools/sk_trace_run.py).  The launch to look at is chosen at run time by (jobs, grid x)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from controllable_xgating_amd import SAModel, make_opt, _native as nv
B = int(os.environ.get("SK_B", "128"))
cfg = dict(B=B, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
prec = os.environ.get("SK_PREC", "fp32")
model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"], precision=prec)).cuda()
model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, "cuda")
L = nv.lib()
names = ["desc", "prologue", "first", "kloop", "reduce", "epilogue"]
# rollout-form step at 128 rows: L1 = [S2' | gate | p | zero] (4 jobs, 256 wide), L2 = [attn | cell 1] (2 jobs, 256), L3 = [cell 2 | copy] (2 jobs, 256)
shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("SK_SHAPES", "4,256 2,256").split()]
for (nj, gx) in shapes:
    assert L.xg_debug_sk_trace_filter(nj, gx) == 0
    assert L.xg_debug_sk_trace_clear() == 0
    t = bench.measure_step_group(model, x, reps=50)
    torch.cuda.synchronize()
    print("== launches with %d jobs x %d workgroups ; step group %.2f us (%s)" % (nj, gx, t * 1e6, prec))
    n = 4096 * 8
    buf = (C.c_longlong * n)()
    assert L.xg_debug_sk_trace(buf, n) == 0
    h = np.array(buf[:], dtype=np.int64).reshape(4096, 8)
    live = h[:, 5] > 0
    if not live.any():
        print("   (no workgroup recorded)")
        continue
    t0 = h[live, 0].min()
    us = lambda v: (v - t0) * 0.01
    print("   WGs recorded %d ; span %.2f us" % (live.sum(), us(h[live, 5].max())))
    for y in range(nj):
        rows = h[y * gx:(y + 1) * gx]
        rows = rows[rows[:, 5] > 0]
        if not len(rows):
            continue
        # order of the stamps in time: 0 entry, 6 descriptor decoded, 1 before the first operand request, 2 first chunk staged,
        # 3 wave 0 leaves the K loop, 7 LAST wave leaves it, 4 partial tiles reduced, 5 epilogue done
        seq = rows[:, [0, 6, 1, 2, 3, 7, 4, 5]].astype(np.float64)
        d = np.diff(seq, axis=1) * 0.01
        print("   job %d: %3d tiles | entry %.2f..%.2f | exit %.2f..%.2f (median %.2f) | mean phases: desc %.2f prologue %.2f first %.2f "
              "kloop(w0) %.2f skew(last wave) %.2f reduce %.2f epilogue %.2f" % (
                  y, len(rows), us(rows[:, 0].min()), us(rows[:, 0].max()), us(rows[:, 5].min()), us(rows[:, 5].max()),
                  us(np.median(rows[:, 5])), *d.mean(0)))
