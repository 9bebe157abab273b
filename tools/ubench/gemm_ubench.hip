// Times xgk_gemm on the three big shapes.  Build with -DGEMM_NO_GLOBAL / -DGEMM_NO_LDS_STORE / -DGEMM_NO_SYNC to see which
// part of the slab loop costs the matrix pipe its time (results are wrong in those builds; only the clock matters).
#include "../../controllable_xgating_amd/csrc/xg_gemm.hip"
#include <vector>
int xgk_gemm_bf16(hipStream_t, int, bool, bool, int, int, int, const float*, int, const float*, int, float*, int, const float*, bool, bool) { return 0; }
static float* dalloc(size_t n, float v) { float* p; (void)hipMalloc(&p, n * 4); std::vector<float> h(n, v);
    if (getenv("GEMM_RANDOM")) { unsigned x = 12345u; for (auto& f : h) { x = x * 1664525u + 1013904223u; f = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f); } } (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; }
int main() {
    struct S { const char* name; bool ta, tb; int M, N, K; bool acc; } shapes[] = {
        {"logits fwd NT 2688x20000 K=512", false, true, 2688, 20000, 512, false},
        {"dW_logit TN 20000x512 K=2688", true, false, 20000, 512, 2688, true},
        {"dH NN 2688x512 K=20000", false, false, 2688, 512, 20000, false}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (auto& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        float *A = dalloc(na, 0.01f), *B = dalloc(nb, 0.02f), *C = dalloc(nc, 0.f);
        const int lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
        for (int it = 0; it < 3; ++it) {
            (void)hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) xgk_gemm(0, 0, s.ta, s.tb, s.M, s.N, s.K, A, lda, B, ldb, C, s.N, nullptr, false, s.acc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) {
                printf("%-36s %8.1f us  %6.1f TF", s.name, ms * 100, 2.0 * s.M * s.N * s.K / (ms * 1e-4) / 1e12);
#ifdef GEMM_CLK
                long long h[4]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gemm_clk_buf), sizeof(h));
                printf("   shader clock inside a workgroup: %.2f GHz (%lld clk / %lld ticks of 10 ns)", h[0] / (h[1] * 10.0), h[0], h[1]);
#endif
                printf("\n");
#ifdef PK_TRACE
                {
                    static long long h[512 * 40]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(pk_trace_buf), sizeof(h));
                    long long t0 = h[0]; for (int w = 0; w < 512; ++w) if (h[w * 40] && h[w * 40] < t0) t0 = h[w * 40];
                    { long long mx = 0, mn = 1LL << 62; int wmx = 0; for (int w = 0; w < 512; ++w) { long long last = 0; for (int i = 0; i < 38; ++i) if (h[w * 40 + i] > last) last = h[w * 40 + i]; if (last > mx) { mx = last; wmx = w; } if (last && last < mn) mn = last; }
                      printf("  first WG end %.1f us, last WG end %.1f us (wg %d)\n", (mn - t0) * 0.01, (mx - t0) * 0.01, wmx); }
                    for (int w : {0, 1, 8, 100, 255, 256, 511}) { { long long last = 0; for (int i = 0; i < 38; ++i) if (h[w * 40 + i] > last) last = h[w * 40 + i]; printf("  [%.3f GHz]", (h[w * 40 + 39] - h[w * 40 + 38]) / ((last - h[w * 40]) * 10.0)); } printf("  wg %3d:", w); for (int i = 0; i < 30; ++i) printf(" %.1f", (h[w * 40 + i] - t0) * 0.01); printf("\n"); }
                }
#endif
            }
        }
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
    }
    return 0;
}
