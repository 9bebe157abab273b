"""Time xg_gemm / xg_gemm_mode on the iteration's big shapes and report the max error vs an fp64 reference.
XG_GEMM_FORCE="tile,splitk" overrides the fp32 kernel's heuristic.  usage: gemm_bench.py [one] [mode]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [  # name, ta, tb, M, N, K, acc
    ("logits fwd NT", 0, 1, 2688, 20000, 512, 0),
    ("dW_logit TN", 1, 0, 20000, 512, 2688, 1),
    ("dH NN K=20000", 0, 0, 2688, 512, 20000, 0),
    ("wgrad TN 2048x512 K=2688", 1, 0, 2048, 512, 2688, 1),
    ("enc embed NT 3328x512 K=1536", 0, 1, 3328, 512, 1536, 0),
    ("PRE NT 3328x2048 K=512", 0, 1, 3328, 2048, 512, 0),
    ("vproj NT 3328x1536 K=512", 0, 1, 3328, 1536, 512, 0),
    ("dX NN 3328x512 K=2048", 0, 0, 3328, 512, 2048, 0),
    ("rollout logits NT 128x20000 K=512", 0, 1, 128, 20000, 512, 0),
    ("rollout logits NT 64x20000 K=512", 0, 1, 64, 20000, 512, 0),
    ("dH NN 1920x512 K=20000", 0, 0, 1920, 512, 20000, 0),
    ("dH half NN 1408x512 K=20000", 0, 0, 1408, 512, 20000, 0),
    ("dH half NN 1280x512 K=20000", 0, 0, 1280, 512, 20000, 0),
]
if os.environ.get("XG_GEMM_SHAPES"):
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in os.environ["XG_GEMM_SHAPES"].split(","))]
def run(mode):
    import torch
    from controllable_xgating_amd import _native as nv
    L = nv.lib()
    out = {}
    for name, ta, tb, M, N, K, acc in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
        Cc = torch.zeros(M, N, device="cuda")
        def call():
            assert L.xg_gemm_mode(None, mode, ta, tb, M, N, K, nv.ptr(A), A.shape[1], nv.ptr(B), B.shape[1], nv.ptr(Cc), N, None, 0, acc) == 0
        Cc.zero_(); call()
        sl = slice(0, min(256, M))
        ref = ((A.t() if ta else A)[sl].double() @ (B.t() if tb else B).double())
        err = float((Cc[sl].double() - ref).abs().max() / ref.abs().max())
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out[name] = (round(us, 1), round(2.0 * M * N * K / us / 1e6, 1), float("%.2g" % err))
    print(("mode %d %s" % (mode, os.environ.get("XG_GEMM_FORCE", "auto"))).ljust(14), json.dumps(out))
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    else:
        for m in (0, 3, 1):
            subprocess.run([sys.executable, __file__, "one", str(m)])
