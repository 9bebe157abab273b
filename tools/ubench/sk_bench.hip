// Times xgk_skinny on the decoder's per-step shapes.  Build variants with -DSK_NO_MFMA / -DSK_NO_LOAD to see
// which side of the pipeline bounds the kernel.
#include "../../controllable_xgating_amd/csrc/xg_step.hip"
#include <cstdio>
int xgk_gemm(hipStream_t, bool, bool, int, int, int, const float*, int, const float*, int, float*, int, const float*, bool, bool) { return 0; }
int xgk_get_gemm_mode() { return getenv("SK_BF16") ? 1 : 0; }
static float* dalloc(size_t n, float v) {
    float* p; hipMalloc(&p, n * 4);
    std::vector<float> h(n, v); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p;
}
#include <vector>
int main() {
    const int B = 128, R = 512, A = 1536;
    float *h1 = dalloc(B * R, 0.01f), *af = dalloc(B * R, 0.02f), *h2 = dalloc(B * R, 0.03f);
    // NSET rotating copies of the weights: consecutive launches never find their weights in L2 (as in the model)
    const int NSET = 8;
    float *W1s[NSET], *W2s[NSET], *W3s[NSET], *Was[NSET];
    for (int q = 0; q < NSET; ++q) { W1s[q] = dalloc(4 * R * R, 0.001f); W2s[q] = dalloc(4 * R * R, 0.001f); W3s[q] = dalloc(4 * R * R, 0.001f); Was[q] = dalloc((size_t)A * 2 * R, 0.001f); }
    float *W1 = W1s[0], *W2 = W2s[0], *W3 = W3s[0];
    int rot = 0;
    float *b = dalloc(4 * R, 0.f), *c = dalloc(B * R, 0.1f), *co = dalloc(B * R, 0), *ho = dalloc(B * R, 0), *g = dalloc(B * 4 * R, 0);
    float *Wa = Was[0], *P = dalloc(B * A, 0), *ds = dalloc(B * 4 * R, 0.01f), *dx = dalloc(B * R, 0);
    XgRun run{}; run.train = 0;
    auto cell2 = [&]() {
        SkArgs k{}; k.njobs = 1;
        SkJob& j = k.job[0];
        j.M = B; j.N = 4 * R; j.R = R; j.epi = SK_EPI_LSTM; j.order = 0; j.mask_mode = 0;
        j.c_prev = c; j.ldcp = R; j.h_prev = h2; j.ldhp = R; j.gates = g; j.ldg = 4 * R; j.c_out = co; j.ldco = R; j.h_out = ho; j.ldho = R;
        j.drop = xg_make_drop(&run, 0, 0);
        j.nseg = 3; j.seg[0] = SkSeg{h1, W1, R, R, R, 0}; j.seg[1] = SkSeg{af, W2, R, R, R, 0}; j.seg[2] = SkSeg{h2, W3, R, R, R, 0};
        j.bias[0] = b;
        return xgk_skinny(0, k);
    };
    auto pjob = [&]() {
        SkArgs k{}; k.njobs = 1;
        SkJob& j = k.job[0];
        j.M = B; j.N = A; j.C = P; j.ldc = A; j.epi = SK_EPI_STORE;
        j.nseg = 2; j.seg[0] = SkSeg{h1, Wa, R, 2 * R, R, 0}; j.seg[1] = SkSeg{h2, Wa + R, R, 2 * R, R, 0};
        return xgk_skinny(0, k);
    };
    auto nn = [&]() {
        SkArgs k{}; k.njobs = 3;
        for (int q = 0; q < 3; ++q) {
            SkJob& j = k.job[q];
            j.M = B; j.N = R; j.C = dx; j.ldc = R; j.epi = SK_EPI_STORE; j.nseg = 1;
            j.seg[0] = SkSeg{ds, q == 0 ? W1 : q == 1 ? W2 : W3, 4 * R, R, 4 * R, 1};
        }
        return xgk_skinny(0, k);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"cell2 (3 seg K=1536, LSTM epi, 256 WG)", "p (2 seg K=1024, 192 WG)", "bwd NN 3 jobs K=2048 (192 WG)"};
    for (int which = 0; which < 3; ++which) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            for (int r = 0; r < 1000; ++r) { rot = (rot + 1) % NSET; W1 = W1s[rot]; W2 = W2s[rot]; W3 = W3s[rot]; Wa = Was[rot]; if (which == 0) cell2(); else if (which == 1) pjob(); else nn(); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) printf("%-45s %.2f us/launch\n", names[which], ms * 1e3 / 1000);
        }
    }
#ifdef SK_TRACE
    for (int which = 0; which < 3; ++which) {
        hipDeviceSynchronize();
        for (int r = 0; r < 5; ++r) { rot = (rot + 1) % NSET; W1 = W1s[rot]; W2 = W2s[rot]; W3 = W3s[rot]; Wa = Was[rot]; if (which == 0) cell2(); else if (which == 1) pjob(); else nn(); }
        hipDeviceSynchronize();
        static long long h[4096 * 8];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(sk_trace_buf), sizeof(h));
        const int nwg = which == 0 ? 256 : 192;
        long long t0 = h[0], t5 = 0;
        for (int w = 0; w < nwg; ++w) { if (h[w * 8] < t0) t0 = h[w * 8]; if (h[w * 8 + 5] > t5) t5 = h[w * 8 + 5]; }
        printf("%s: span first-entry -> last-exit %.2f us\n", names[which], (t5 - t0) * 0.01);
        for (int w : {0, 1, 97, nwg - 1}) {
            printf("   wg %3d: entry +%.2f | prologue %.2f | first chunk %.2f | k loop %.2f | reduce %.2f | epilogue %.2f\n", w, (h[w * 8] - t0) * 0.01,
                   (h[w * 8 + 1] - h[w * 8]) * 0.01, (h[w * 8 + 2] - h[w * 8 + 1]) * 0.01, (h[w * 8 + 3] - h[w * 8 + 2]) * 0.01,
                   (h[w * 8 + 4] - h[w * 8 + 3]) * 0.01, (h[w * 8 + 5] - h[w * 8 + 4]) * 0.01);
        }
    }
#endif
    return 0;
}
