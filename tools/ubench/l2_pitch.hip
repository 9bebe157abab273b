// Microbenchmark: L2->CU bandwidth of the skinny-GEMM operand pattern (each wave instruction = 8 rows x 128 B,
// 4 instructions = 32 rows x one 128-B line) as a function of the row pitch.  Working set stays L2/MALL resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: 32 rows x 128B per chunk (pitch-strided) ; 1: linear 4 KB per chunk
__global__ void __launch_bounds__(512) k(const float* __restrict__ base, int pitch_f, int nchunk, int rows_total, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = blockIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    // every WG reads the same 32-row panel set (like A re-read by many n-tiles) offset by (wg % 4) panels
    const int row0 = (wg % (rows_total / 32)) * 32;
    for (int c = 0; c < nchunk; ++c) {
        const int kline = wave * nchunk + c;          // contiguous k range per wave
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* p;
            if (MODE == 0) p = base + (size_t)(row0 + i * 8 + (lane >> 3)) * pitch_f + kline * 32 + (lane & 7) * 4;
            else p = base + (size_t)row0 * pitch_f + ((size_t)kline * 4 + i) * 256 + lane * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(p);
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

int main() {
    const int rows = 128, nchunk = 6, waves = 8;     // K = 8 * 6 * 32 = 1536 floats per row
    float* out; hipMalloc(&out, 4);
    for (int pitch : {1536, 1536 + 32, 1536 + 64, 2048, 2048 + 32, 512 * 3, 4096, 4096 + 32, 1600}) {
        float* buf; size_t n = (size_t)rows * pitch + (1 << 20); hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                for (int r = 0; r < 20; ++r) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, buf, pitch, nchunk, rows, out);
                    else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, buf, pitch, nchunk, rows, out);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it == 2) {
                    double bytes = 256.0 * waves * nchunk * 4096;
                    printf("pitch %5d floats mode %d: %.2f us/launch  %.1f GB/s L2->CU  (%.1f B/clk/CU @2.4GHz)\n", pitch, mode,
                           ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e9, bytes / 256 / (ms * 1e-3 / 20) / 2.4e9);
                }
            }
        }
        hipFree(buf);
    }
    return 0;
}
