"""What the vendor fp32 GEMM reaches on this iteration's large shapes (torch.mm -> hipBLASLt / rocBLAS), as a yardstick for
xg_gemm.hip's kernels (profiles/r03_gemm_bench.txt).  Not used by the product path."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
shapes = [  # (name, M, N, K, transA, transB)   C[M,N] = op(A) op(B)
    ("logits  NT 2688x20000x512", 2688, 20000, 512, False, True),
    ("dH      NN 2688x512x20000", 2688, 512, 20000, False, False),
    ("dWlogit TN 20000x512x2688", 20000, 512, 2688, True, False),
    ("emb     NT 3328x512x1536", 3328, 512, 1536, False, True),
    ("W_ih x  NT 3328x2048x512", 3328, 2048, 512, False, True),
    ("dX      NN 3328x512x2048", 3328, 512, 2048, False, False),
    ("dW_ih   TN 2048x512x3328", 2048, 512, 3328, True, False),
    ("vproj   NT 3328x1536x512", 3328, 1536, 512, False, True),
    ("rollout NT 128x20000x512", 128, 20000, 512, False, True),
]
for name, M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    a = A.t() if ta else A
    b = B.t() if tb else B
    for _ in range(5): torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:32s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF")
