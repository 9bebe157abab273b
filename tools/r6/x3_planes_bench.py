#!/usr/bin/env python3
"""Split-bf16 products with operands pre-split in memory (xg_gemm_bf16x3_planes) against the on-the-fly split (xg_gemm_mode 3) and the
exact fp32 kernels (mode 0), on the large NT / NN products of configs[1] (hidden 512) and configs[4] (hidden 1024):
us, TFLOP/s-equivalent, max error against fp64 relative to the largest result."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from controllable_xgating_amd import _native as nv
L = nv.lib()
SHAPES = [  # name, ta, tb, M, N, K
    ("logits late NT 1792x20000 K=512", 0, 1, 1792, 20000, 512),
    ("logits all NT 3584x20000 K=512", 0, 1, 3584, 20000, 512),
    ("dH late NN 1792x512 K=20000", 0, 0, 1792, 512, 20000),
    ("dH all NN 3584x512 K=20000", 0, 0, 3584, 512, 20000),
    ("enc emb NT 3328x512 K=1536", 0, 1, 3328, 512, 1536),
    ("PRE NT 3328x2048 K=512", 0, 1, 3328, 2048, 512),
    ("enc dX NN 3328x512 K=2048", 0, 0, 3328, 512, 2048),
    ("v2a NT 3328x1536 K=512", 0, 1, 3328, 1536, 512),
    ("logits NT 2688x20000 K=1024", 0, 1, 2688, 20000, 1024),
    ("dH NN 2688x1024 K=20000", 0, 0, 2688, 1024, 20000),
]
if os.environ.get("XG_GEMM_SHAPES"):
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in os.environ["XG_GEMM_SHAPES"].split(","))]
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, ta, tb, M, N, K in SHAPES:
    torch.manual_seed(1)
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    Cc = torch.zeros(M, N, device="cuda")
    A3 = torch.empty((3,) + tuple(A.shape), dtype=torch.bfloat16, device="cuda"); B3 = torch.empty((3,) + tuple(B.shape), dtype=torch.bfloat16, device="cuda")
    assert L.xg_split_bf16x3(None, nv.ptr(A), nv.ptr(A3), A.numel(), A.numel()) == 0
    assert L.xg_split_bf16x3(None, nv.ptr(B), nv.ptr(B3), B.numel(), B.numel()) == 0
    assert torch.equal(A3.float().sum(0), A) and torch.equal(B3[0].float() + (B3[1].float() + B3[2].float()), B)      # the split is exact
    sl = slice(0, min(256, M))
    ref = ((A.t() if ta else A)[sl].double() @ (B.t() if tb else B).double())
    res = {}
    def planes():
        assert L.xg_gemm_bf16x3_planes(None, ta, tb, M, N, K, nv.ptr(A3), A.numel(), A.shape[1], nv.ptr(B3), B.numel(), B.shape[1],
                                       nv.ptr(Cc), N, None, 0, 0) == 0
    def mode(m):
        def f():
            assert L.xg_gemm_mode(None, m, ta, tb, M, N, K, nv.ptr(A), A.shape[1], nv.ptr(B), B.shape[1], nv.ptr(Cc), N, None, 0, 0) == 0
        return f
    for tag, fn in (("fp32", mode(0)), ("split on the fly", mode(3)), ("split planes", planes)):
        Cc.zero_(); fn()
        err = float((Cc[sl].double() - ref).abs().max() / ref.abs().max())
        us = bench(fn)
        res[tag] = "%.1f us %.0f TF err %.1e" % (us, 2.0 * M * N * K / us / 1e6, err)
    sa = bench(lambda: L.xg_split_bf16x3(None, nv.ptr(A), nv.ptr(A3), A.numel(), A.numel()))
    print(name.ljust(34), json.dumps(res), "| split(A) %.1f us" % sa, flush=True)
