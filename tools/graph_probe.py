"""Diagnosis: which part of the XE iteration survives HIP-graph capture (each variant in its own process)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = ["fwd", "fwd_noaux", "fwdbwd", "fwdbwd_noaux", "full_nooverlap", "full_noaux", "full"]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import faulthandler; faulthandler.enable()
    import torch, bench
    from controllable_xgating_amd import SAModel, make_opt
    from controllable_xgating_amd.train import ClipAdam, shared_stream
    v = sys.argv[2]
    dev = torch.device("cuda", 0)
    m = SAModel(make_opt(None)).to(dev); m.train()
    if "noaux" in v:
        m._aux_handle = lambda: None
    x = bench.synth_inputs(128, 26, 20, 20000, 512, 1536, 1024, 14, 0, dev)
    o = ClipAdam(m, lr=4e-4, grad_clip=0.1, overlap=("nooverlap" not in v and "noaux" not in v), fused_zero=True, device_state=True)
    def it():
        if v.startswith("fwd") and not v.startswith("fwdbwd"):
            with torch.no_grad():
                return m.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        o.zero_grad()
        loss = m.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        if v.startswith("full"):
            o.arm()
        loss.backward()
        if v.startswith("full"):
            o.step(); m._packed_ptr()
        return loss
    side = shared_stream("graph")
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            it()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            loss = it()
    torch.cuda.current_stream().wait_stream(side)
    import time
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    te = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("OK %s: loss %.5f  %.3f ms per replay (host %.3f ms)" % (v, float(loss.item()), t * 100, te * 100))
else:
    for v in (sys.argv[1:] or VARIANTS):
        r = subprocess.run(["timeout", "120", sys.executable, os.path.abspath(__file__), "child", v], capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if l.startswith("OK")]
        print(v, "rc", r.returncode, out[-1] if out else (r.stderr.strip().splitlines() or ["?"])[-1][:300], flush=True)
        if r.returncode != 0:
            tb = [l for l in r.stderr.splitlines() if "File" in l or "Error" in l or "error" in l][:6]
            print("   ", "\n    ".join(tb))
