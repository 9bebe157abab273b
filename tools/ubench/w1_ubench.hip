// Times xg_gemm.hip's one-workgroup-per-CU kernel on the mid-size shapes, with in-kernel stamps (-DW1_TRACE): prologue, slab loop
// (shader cycles per slab against the MFMA cycles it holds), epilogue.  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DXG_DIAG -DW1_TRACE tools/ubench/w1_ubench.hip -o /tmp/w1_ubench
#include "../../controllable_xgating_amd/csrc/xg_gemm.hip"
#include <vector>
int xgk_gemm_bf16(hipStream_t, int, bool, bool, int, int, int, const float*, int, const float*, int, float*, int, const float*, bool, bool) { return 0; }
int xgk_colsum3(hipStream_t, const float*, int, int, int, float*, float*, float*) { return 0; }
int xgk_gemm_bf16x(hipStream_t, int, bool, bool, int, int, int, const float*, const unsigned short*, int, const float*, const unsigned short*, int,
                   float*, int, const float*, bool, bool) { return 0; }
static float* dalloc(size_t n) { float* p; (void)hipMalloc(&p, n * 4); std::vector<float> h(n); unsigned x = 12345u;
    for (auto& f : h) { x = x * 1664525u + 1013904223u; f = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; } (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; }
int main() {
    struct S { const char* name; bool ta, tb; int M, N, K; bool acc; } shapes[] = {
        {"wgrad TN 2048x512 K=2688", true, false, 2048, 512, 2688, true},
        {"enc embed NT 3328x512 K=1536", false, true, 3328, 512, 1536, false},
        {"PRE NT 3328x2048 K=512", false, true, 3328, 2048, 512, false},
        {"vproj NT 3328x1536 K=512", false, true, 3328, 1536, 512, false},
        {"dX NN 3328x512 K=2048", false, false, 3328, 512, 2048, false}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (auto& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        float *A = dalloc(na), *B = dalloc(nb), *C = dalloc(nc);
        const int lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
        for (int it = 0; it < 3; ++it) {
            (void)hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) xgk_gemm(0, 0, s.ta, s.tb, s.M, s.N, s.K, A, lda, B, ldb, C, s.N, nullptr, false, s.acc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) {
                printf("%-32s %8.1f us  %6.1f TF\n", s.name, ms * 100, 2.0 * s.M * s.N * s.K / (ms * 1e-4) / 1e12);
#ifdef W1_TRACE
                static long long h[256 * 8]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(w1_trace_buf), sizeof(h));
                long long t0 = 1LL << 62; for (int w = 0; w < 256; ++w) if (h[w * 8] && h[w * 8] < t0) t0 = h[w * 8];
                for (int w : {0, 1, 7, 8, 100, 200}) {
                    const long long* q = h + w * 8;
                    printf("   wg %3d: start %.2f  loop from %.2f to %.2f  end %.2f us;  loop = %lld shader clk (%.2f GHz)\n", w, (q[0] - t0) * 0.01,
                           (q[1] - t0) * 0.01, (q[2] - t0) * 0.01, (q[3] - t0) * 0.01, q[5] - q[4], (q[5] - q[4]) / ((q[2] - q[1]) * 10.0));
                }
#endif
            }
        }
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
    }
    return 0;
}
