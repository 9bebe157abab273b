#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) mf(float* out, int n, long long* clk) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    long long t1 = clock64(), w1 = wall_clock64();
    if (acc[0] == 123.f) out[0] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
int main() {
    float* out; hipMalloc(&out, 4); long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int reps : {50, 500, 5000}) {
        for (int n : {96, 960}) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mf, dim3(256), dim3(512), 0, 0, out, n, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("reps %5d n %4d MFMA/wave (x2 waves/SIMD): %.2f us/launch ; in-kernel: %lld shader clk, %lld wall ticks(100MHz) -> %.2f GHz ; %.1f cyc/MFMA/SIMD\n",
                   reps, n, ms * 1e3 / reps, h[0], h[1], h[0] / (h[1] * 10.0), (double)h[0] / (2.0 * n));
        }
    }
}
