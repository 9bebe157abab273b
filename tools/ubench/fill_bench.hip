// L2 -> CU fill rate on MI355X as a function of waves per CU and loads in flight per wave: every wave streams 1 KB-coalesced
// 16-byte loads (the packed-weight access pattern of xg_step.hip) from a buffer that fits the L2s (or not), sums them, no MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o fill_bench fill_bench.hip ; run: ./fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(1024) fill_kernel(const f32x4* __restrict__ buf, size_t wg_stride4, int iters, size_t span4, float* out) {
    // workgroup b reads [b * wg_stride4, +span4) (float4 units) round after round; wave w, lane l: 1 KB per wave-load
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const f32x4* p = buf + (size_t)blockIdx.x * wg_stride4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        for (size_t o = (size_t)wave * 64 * UNROLL; o + 64 * UNROLL <= span4; o += (size_t)nw * 64 * UNROLL) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[o + u * 64 + lane];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

template <int UNROLL>
double run(const f32x4* buf, float* out, int wgs, int threads, size_t wg_stride4, size_t span4, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fill_kernel<UNROLL>, dim3(wgs), dim3(threads), 0, 0, buf, wg_stride4, 2, span4, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(fill_kernel<UNROLL>, dim3(wgs), dim3(threads), 0, 0, buf, wg_stride4, iters, span4, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * span4 * 16.0 * iters / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t total = (size_t)512 << 20;            // 512 MB buffer
    f32x4* buf; float* out;
    hipMalloc(&buf, total); hipMalloc(&out, 4); hipMemset(buf, 0, total);
    printf("# TB/s aggregate (256 workgroups, one per CU); span per workgroup; waves per CU x loads in flight per wave\n");
    for (size_t span_kb : {64, 256, 1024}) {            // 256 x span: 16 MB (L2), 64 MB (MALL), 256 MB (MALL edge)
        for (int waves : {4, 8, 16}) {
            const size_t span4 = span_kb * 1024 / 16;
            const int iters = (int)((size_t)(64 << 20) / (span_kb * 1024)) + 1;
            printf("span %4zu KB/WG  waves/CU %2d : ", span_kb, waves);
            printf("x2 %.1f  ", run<2>(buf, out, 256, waves * 64, span4, span4, iters));
            printf("x4 %.1f  ", run<4>(buf, out, 256, waves * 64, span4, span4, iters));
            printf("x8 %.1f  ", run<8>(buf, out, 256, waves * 64, span4, span4, iters));
            printf("x16 %.1f\n", run<16>(buf, out, 256, waves * 64, span4, span4, iters));
        }
    }
    // 2 workgroups per CU of 8 waves (the skinny kernels' shape), all CUs reading the SAME 16 MB (weights shared by the 4 m-tiles)
    for (int wgs : {256, 512, 1024}) {
        const size_t span4 = 256 * 1024 / 16;
        printf("shared: %4d WGs x 8 waves, each 256 KB of a 16 MB region read by 4 WGs: x4 %.1f  x8 %.1f\n", wgs,
               run<4>(buf, out, wgs, 512, span4 / 4, span4, 64), run<8>(buf, out, wgs, 512, span4 / 4, span4, 64));
    }
    return 0;
}
