#!/usr/bin/env python3
"""In-kernel stamps of vocab_part16_kernel (-DVP_TRACE build) inside a paired SCST rollout at 128 rows.
XG_LIBRARY=controllable_xgating_amd/lib/libxgate_hip_vptrace.so python tools/r6/vp_trace.py"""
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
for _ in range(3):
    with torch.no_grad():
        scst_rollouts(model, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], trim=False)
torch.cuda.synchronize()
n = 512 * 8
buf = (C.c_longlong * n)()
assert L.xg_debug_vp_trace(buf, n) == 0
raw = np.array(buf[:], dtype=np.int64).reshape(512, 8)[:250]
h = raw[:, :5].astype(np.float64)
cyc = (raw[:, 6] - raw[:, 5]).astype(np.float64)            # shader clocks over the K loop (s_memtime)
kl = (raw[:, 2] - raw[:, 1]) * 0.01
print("K loop: %.0f shader clocks (mean) = %.1f per MFMA (1280 per wave; 32 = the matrix pipe's rate) at an effective %.2f GHz" % (cyc.mean(), cyc.mean() / 1280, cyc.mean() / kl.mean() / 1e3))
t0 = h[:, 0].min()
d = np.diff(h, axis=1) * 0.01
print("vocab_part16_kernel, 250 workgroups (last launch of the rollout); us, mean / min / max over workgroups")
print("entry after the first: %.2f / %.2f / %.2f" % (((h[:, 0] - t0) * 0.01).mean(), 0.0, ((h[:, 0] - t0) * 0.01).max()))
for i, name in enumerate(["prologue (2 slabs requested, first staged)", "K loop (16 slabs)", "tile -> LDS", "statistics + stores"]):
    print("%-45s %.2f / %.2f / %.2f" % (name, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
print("exit after the first entry: %.2f / %.2f / %.2f" % (((h[:, 4] - t0) * 0.01).mean(), ((h[:, 4] - t0) * 0.01).min(), ((h[:, 4] - t0) * 0.01).max()))
