"""Where a slab's time goes in xg_gemm.hip's fp32 kernels: builds of the library with parts of the K loop compiled out
(-DGEMM_NO_GLOBAL: no global loads in the loop, -DGEMM_NO_LDS_STORE: no LDS staging stores, -DGEMM_NO_SYNC: no workgroup barrier,
-DGEMM_NO_EPI: no result stores in the persistent kernel) timed on the same shapes (results are wrong by construction).
usage: gemm_ablate.py build | run"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANTS = {
    "abl_full": ["-DXG_DIAG"],
    "abl_noglobal": ["-DXG_DIAG", "-DGEMM_NO_GLOBAL"],
    "abl_nostore": ["-DXG_DIAG", "-DGEMM_NO_GLOBAL", "-DGEMM_NO_LDS_STORE"],
    "abl_mfma": ["-DXG_DIAG", "-DGEMM_NO_GLOBAL", "-DGEMM_NO_LDS_STORE", "-DGEMM_NO_SYNC"],
    "abl_mfma_noepi": ["-DXG_DIAG", "-DGEMM_NO_GLOBAL", "-DGEMM_NO_LDS_STORE", "-DGEMM_NO_SYNC", "-DGEMM_NO_EPI"],
}
if sys.argv[1] == "build":
    import __graft_entry__ as g
    for n, f in VARIANTS.items():
        print(g.build_variant(n, f))
else:
    for n in VARIANTS:
        env = dict(os.environ, XG_LIBRARY=os.path.join(ROOT, "controllable_xgating_amd", "lib", "libxgate_hip_%s.so" % n),
                   XG_GEMM_SHAPES=os.environ.get("XG_GEMM_SHAPES", "logits fwd,wgrad,embed,dX NN 3328"))
        print("==", n, flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ubench", "gemm_bench.py"), "one", "0"], env=env)
