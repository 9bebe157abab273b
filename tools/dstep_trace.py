"""In-kernel timeline of the dataflow step kernel (xg_dstep.hip built with -DDS_TRACE -DXG_DIAG into
lib/libxgate_hip_dstrace.so by `python tools/dstep_trace.py build`): per job, when its workgroups start / pass their waits /
finish, relative to the first workgroup's start (us)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import __graft_entry__ as ge
    print(ge.build_variant("dstrace", ["-DXG_DIAG", "-DDS_TRACE"]))
    sys.exit(0)
os.environ["XG_LIBRARY"] = os.path.join(ROOT, "controllable_xgating_amd", "lib", "libxgate_hip_dstrace.so")
os.environ["XG_DSTEP"] = "1"
import ctypes as C
import numpy as np
import torch
import bench
from controllable_xgating_amd import SAModel, make_opt, _native as nv

B = int(os.environ.get("DS_B", "128"))
model = SAModel(make_opt(None)).cuda(); model.train()
x = bench.synth_inputs(B, 26, 20, 20000, 512, 1536, 1024, 14, 0, "cuda")
t = bench.measure_step_group(model, x, reps=50)
torch.cuda.synchronize()
ntm = (B + 31) // 32
n_p, n_gate, n_cell = ntm * 48, ntm * 16, ntm * 64
o = [0, n_gate, n_gate + n_p, n_gate + n_p + B, n_gate + n_p + B + n_cell, n_gate + n_p + B + 2 * n_cell]
jobs = [("gate", o[0], o[1]), ("p", o[1], o[2]), ("attn", o[2], o[3]), ("cell1", o[3], o[4]), ("cell2", o[4], o[5])]
total = jobs[-1][2]
buf = (C.c_longlong * (total * 8))()
assert nv.lib().xg_debug_ds_trace(buf, total * 8) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(total, 8).astype(np.float64)
t0 = a[:, 0].min()
a = (a - t0) / 100.0          # 100 MHz -> us
print("step group %.2f us (event timing); last workgroup ends at %.2f us" % (t * 1e6, a[:, 7].max()))
def q(v):
    return "min %6.2f  med %6.2f  max %6.2f" % (v.min(), np.median(v), v.max())
for name, lo, hi in jobs:
    s = a[lo:hi]
    print("%-6s n=%4d  start: %s | end: %s | dur med %.2f" % (name, hi - lo, q(s[:, 0]), q(s[:, 7]), np.median(s[:, 7] - s[:, 0])))
    if name in ("cell1", "attn"):
        print("         wait: reached %s | passed %s | K/score done %s" % (q(s[:, 1]), q(s[:, 2]), q(s[:, 3])))
    if name == "cell2":
        print("         wait c1: reached %s | passed %s" % (q(s[:, 1]), q(s[:, 2])))
        print("         wait att: reached %s | passed %s | K done %s" % (q(s[:, 3]), q(s[:, 4]), q(s[:, 5])))
    if name == "p":
        print("         wave-0 K done %s | all waves reduced %s" % (q(s[:, 3]), q(s[:, 4])))
        print("         stores issued %s | wave-0 stores acked %s" % (q(s[:, 5]), q(s[:, 6])))
