"""bf16 GEMM (gemm_mode 1) with fp32 operands converted on the fly vs bf16 operand copies in memory (xg_gemm_bf16_operands),
on the shapes of BASELINE.json configs[4] (hidden 1024, 40 frames, batch 128): us, TFLOP/s, error vs fp64."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from controllable_xgating_amd import _native as nv
L = nv.lib()
SHAPES = [  # name, ta, tb, M, N, K
    ("logits NT 2688x20000 K=1024", 0, 1, 2688, 20000, 1024),
    ("dW_logit TN 20000x1024 K=2688", 1, 0, 20000, 1024, 2688),
    ("dH NN 2688x1024 K=20000", 0, 0, 2688, 1024, 20000),
    ("PRE NT 5120x4096 K=1024", 0, 1, 5120, 4096, 1024),
    ("enc dW_hh TN 4096x1024 K=5120", 1, 0, 4096, 1024, 5120),
    ("enc dX NN 5120x1024 K=4096", 0, 0, 5120, 1024, 4096),
    ("dec wgrad TN 4096x1024 K=2688", 1, 0, 4096, 1024, 2688),
    ("emb NT 5120x1024 K=1536", 0, 1, 5120, 1024, 1536),
    ("emb dW TN 1024x1536 K=5120", 1, 0, 1024, 1536, 5120),
    ("emb dW TN 1024x1024 K=5120", 1, 0, 1024, 1024, 5120),
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
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    Cc = torch.zeros(M, N, device="cuda")
    A16 = torch.empty(A.shape, dtype=torch.bfloat16, device="cuda"); B16 = torch.empty(B.shape, dtype=torch.bfloat16, device="cuda")
    assert L.xg_cvt_bf16(None, nv.ptr(A), nv.ptr(A16), A.numel()) == 0 and L.xg_cvt_bf16(None, nv.ptr(B), nv.ptr(B16), B.numel()) == 0
    assert torch.equal(A16, A.bfloat16()) and torch.equal(B16, B.bfloat16())
    sl = slice(0, min(256, M))
    ref = ((A.t() if ta else A)[sl].double() @ (B.t() if tb else B).double())
    res = {}
    cases = (("fp32 operands", None, None), ("A bf16", A16, None), ("B bf16", None, B16), ("both bf16", A16, B16))
    if len(sys.argv) > 1 and sys.argv[1] == "both":
        cases = cases[3:]
    for tag, a16, b16 in cases:
        def call():
            assert L.xg_gemm_bf16_operands(None, ta, tb, M, N, K, nv.ptr(A), nv.ptr(a16), A.shape[1], nv.ptr(B), nv.ptr(b16), B.shape[1],
                                           nv.ptr(Cc), N, None, 0, 0) == 0
        Cc.zero_(); call()
        err = float((Cc[sl].double() - ref).abs().max() / ref.abs().max())
        us = bench(call)
        res[tag] = "%.1f us %.0f TF err %.1e" % (us, 2.0 * M * N * K / us / 1e6, err)
    cv = bench(lambda: L.xg_cvt_bf16(None, nv.ptr(A), nv.ptr(A16), A.numel()))
    print(name.ljust(34), json.dumps(res), "| cvt(A) %.1f us" % cv, flush=True)
