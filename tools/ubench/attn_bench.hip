// Times xgk_attn_fwd / xgk_attn_bwd on the decoder's shapes and prints an in-kernel timeline (XG_ATTN_TRACE).
#define XG_ATTN_TRACE 1
#include "../../controllable_xgating_amd/csrc/xg_attn.hip"
#include <cstdio>
#include <vector>
static float* dalloc(size_t n, float v) {
    float* p; (void)hipMalloc(&p, n * 4);
    std::vector<float> h(n, v); (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p;
}
int main() {
    const int B = 128, K = 26, R = 512, A = 1536, NSET = 6;
    float *P[NSET], *VP[NSET], *V[NSET];
    for (int q = 0; q < NSET; ++q) { P[q] = dalloc((size_t)B * A, 0.01f); VP[q] = dalloc((size_t)B * K * A, 0.02f); V[q] = dalloc((size_t)B * K * R, 0.5f); }
    float *w = dalloc(A, 0.01f), *alpha = dalloc(B * K, 0), *af = dalloc(B * R, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < 1000; ++r) xgk_attn_fwd(0, P[r % NSET], VP[r % NSET], V[r % NSET], w, alpha, af, B, K, R, A);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it == 2) printf("attn_fwd B=128 K=26 A=1536: %.2f us/launch\n", ms);
    }
    (void)hipDeviceSynchronize();
    static long long h[1024 * 8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(attn_trace_buf), sizeof(h));
    long long t0 = h[0], t5 = 0;
    for (int b = 0; b < B; ++b) { if (h[b * 8] < t0) t0 = h[b * 8]; if (h[b * 8 + 5] > t5) t5 = h[b * 8 + 5]; }
    printf("span first-entry -> last-exit %.2f us\n", (t5 - t0) * 0.01);
    for (int b : {0, 1, 63, 127})
        printf("  wg %3d: entry +%.2f | issue V/p/w loads %.2f | score rounds %.2f | barrier %.2f | softmax+context %.2f | reduce+store %.2f\n", b,
               (h[b * 8] - t0) * 0.01, (h[b * 8 + 1] - h[b * 8]) * 0.01, (h[b * 8 + 2] - h[b * 8 + 1]) * 0.01, (h[b * 8 + 3] - h[b * 8 + 2]) * 0.01,
               (h[b * 8 + 4] - h[b * 8 + 3]) * 0.01, (h[b * 8 + 5] - h[b * 8 + 4]) * 0.01);
    float *daf = dalloc(B * R, 0.01f), *de = dalloc(B * K, 0), *dp = dalloc((size_t)B * A, 0);
    { std::vector<float> ha(B * K, 1.0f / K); (void)hipMemcpy(alpha, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); }
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < 1000; ++r) xgk_attn_bwd(0, daf, R, P[r % NSET], VP[r % NSET], V[r % NSET], w, alpha, de, dp, B, K, R, A);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it == 2) printf("attn_bwd B=128 K=26 A=1536: %.2f us/launch\n", ms);
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(attn_trace_buf), sizeof(h));
    t0 = h[0]; t5 = 0;
    for (int b = 0; b < B; ++b) { if (h[b * 8] < t0) t0 = h[b * 8]; if (h[b * 8 + 5] > t5) t5 = h[b * 8 + 5]; }
    printf("span first-entry -> last-exit %.2f us\n", (t5 - t0) * 0.01);
    for (int b : {0, 1, 63, 127})
        printf("  wg %3d: entry +%.2f | issue loads %.2f | dalpha %.2f | barrier %.2f | softmax bwd %.2f | dp (tanh) %.2f\n", b,
               (h[b * 8] - t0) * 0.01, (h[b * 8 + 1] - h[b * 8]) * 0.01, (h[b * 8 + 2] - h[b * 8 + 1]) * 0.01, (h[b * 8 + 3] - h[b * 8 + 2]) * 0.01,
               (h[b * 8 + 4] - h[b * 8 + 3]) * 0.01, (h[b * 8 + 5] - h[b * 8 + 4]) * 0.01);
    return 0;
}
