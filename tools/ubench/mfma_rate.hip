// Issue rate of the fp32 matrix instructions on gfx950, nothing but MFMAs in the loop: shader cycles per instruction for
// v_mfma_f32_32x32x2_f32 (16 passes) and v_mfma_f32_16x16x4_f32 (8 passes) with 1 / 2 / 4 / 8 independent accumulators, one or two
// waves per SIMD, every CU busy (power state as in a real product).   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ long long out_clk[8];
template <int NACC, bool BIG>
__global__ void __launch_bounds__(512) k(float a0, float b0, float* sink, int iters, int slot) {
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    if (a0 < 0.f) {          // "random" operands: a different value per lane, mantissa bits all over the place
        unsigned x = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u; x ^= x >> 13; x *= 1274126177u; x ^= x >> 16;
        a = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f;
        x = x * 1664525u + 1013904223u;
        b = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f;
    }
    f32x16 acc[NACC]; f32x4 acs[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) { for (int r = 0; r < 16; ++r) acc[q][r] = 0.f; for (int r = 0; r < 4; ++r) acs[q][r] = 0.f; }
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                if (BIG) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
                else acs[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acs[q], 0, 0, 0);
            }
    }
    const long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q) { for (int r = 0; r < 16; ++r) s += acc[q][r]; for (int r = 0; r < 4; ++r) s += acs[q][r]; }
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) out_clk[slot] = c1 - c0;
}
template <int NACC, bool BIG>
void run(int threads, const char* name, float a0 = 1.0f) {
    float* sink; (void)hipMalloc(&sink, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(threads), 0, 0, a0, 2.0f, sink, iters, 0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(threads), 0, 0, a0, 2.0f, sink, iters, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(out_clk), sizeof(h));
    const double n = (double)iters * 16;
    const double flop = n * (BIG ? 4096.0 : 2048.0) * (threads / 64) * 256;
    printf("%-22s %s acc %d, %d waves/SIMD: %6.1f shader clk per MFMA per wave, %6.1f TF (kernel %.1f us)\n", name, a0 < 0.f ? "random operands  " : "constant operands", NACC, threads / 256, h[0] / n,
           flop / (ms * 1e-3) / 1e12, ms * 1e3);
}
int main() {
    run<1, true>(256, "32x32x2_f32"); run<2, true>(256, "32x32x2_f32"); run<4, true>(256, "32x32x2_f32"); run<8, true>(256, "32x32x2_f32");
    run<4, true>(512, "32x32x2_f32"); run<8, true>(512, "32x32x2_f32");
    run<1, false>(256, "16x16x4_f32"); run<2, false>(256, "16x16x4_f32"); run<4, false>(256, "16x16x4_f32"); run<8, false>(256, "16x16x4_f32");
    run<4, false>(512, "16x16x4_f32");
    run<1, true>(256, "32x32x2_f32", -1.f); run<4, true>(256, "32x32x2_f32", -1.f); run<8, true>(256, "32x32x2_f32", -1.f); run<4, true>(512, "32x32x2_f32", -1.f);
    run<1, false>(256, "16x16x4_f32", -1.f); run<4, false>(256, "16x16x4_f32", -1.f); run<8, false>(256, "16x16x4_f32", -1.f); run<4, false>(512, "16x16x4_f32", -1.f);
    return 0;
}
