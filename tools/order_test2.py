"""Diagnosis: iteration time of the XE training step for a sequence of model objects inside one process."""
import sys, os, gc, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
import __graft_entry__ as ge; ge.build()
from controllable_xgating_amd import SAModel, make_opt
from controllable_xgating_amd.train import ClipAdam
dev = torch.device("cuda", 0)
x = bench.synth_inputs(128, 26, 20, 20000, 512, 1536, 1024, 14, 0, dev)

def mk(overlap=True):
    m = SAModel(make_opt(None)).to(dev); m.train()
    o = ClipAdam(m, lr=4e-4, grad_clip=0.1, overlap=overlap, fused_zero=True)
    return m, o

def timeit(m, o, n=10):
    def step():
        o.zero_grad()
        loss = m.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        o.arm(); loss.backward(); o.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)

mode = sys.argv[1]
if mode == "newmodels":
    for i in range(4):
        m, o = mk(); print("model", i, timeit(m, o), flush=True); del m, o; gc.collect()
elif mode == "keepalive":
    keep = []
    for i in range(4):
        m, o = mk(); print("model", i, timeit(m, o), flush=True); keep.append((m, o))
elif mode == "nooverlap":
    for i in range(4):
        m, o = mk(False); print("model", i, timeit(m, o), flush=True); del m, o; gc.collect()
elif mode == "samemodel":
    m, o = mk()
    for i in range(4):
        print("run", i, timeit(m, o), flush=True)
elif mode == "newopt":
    m, o = mk()
    for i in range(4):
        print("run", i, timeit(m, o), flush=True); o = ClipAdam(m, lr=4e-4, grad_clip=0.1, overlap=True, fused_zero=True)
elif mode == "noaux":
    for i in range(4):
        m, o = mk(); m._aux_handle = lambda: None; print("model", i, timeit(m, o), flush=True); del m, o; gc.collect()
