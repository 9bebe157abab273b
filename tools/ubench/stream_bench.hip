// How fast can ONE launch pull a cold 12.6 MB weight matrix into the CUs?  (floor for the skinny step kernels)
//   mode 0: empty kernel;  mode 1: contiguous 16-B loads, all issued up front;  mode 2: the skinny kernel's pattern
//   (32 rows x 128 B pieces at a 2 KB row pitch, 6 dependent rounds);  mode 3: as 2 but all rounds issued up front.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_empty(const float*, float*) {}
template <int NL>
__global__ void __launch_bounds__(512) k_contig(const float* __restrict__ w, float* __restrict__ out, size_t per_wg) {
    const f32x4* p = reinterpret_cast<const f32x4*>(w + (blockIdx.x % 256) * per_wg) + threadIdx.x;
    f32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = p[i * 512];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}
// W is (N=2048, K=1536) row-major; WG = 32 rows; wave w takes k in [w*192, w*192+192): 6 chunks of 32 k
template <bool UPFRONT, int MT /* how many M tiles share each weight tile */>
__global__ void __launch_bounds__(512) k_rows(const float* __restrict__ w, float* __restrict__ out, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    { const int nwg = gridDim.x, q = nwg / 8, xcd = bid % 8, idx = bid / 8; bid = xcd * q + idx; }
    const int tn = bid / MT;
    const float* base = w + (size_t)(tn * 32 + (lane >> 3)) * K + wave * (K / 8) + ((lane & 7) << 2);
    float s = 0.f;
    const int NC = 6;
    if (UPFRONT) {
        f32x4 v[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c][i] = *reinterpret_cast<const f32x4*>(base + (size_t)i * 8 * K + c * 32);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[c][i][0] + v[c][i][3];
    } else {
        for (int c = 0; c < NC; ++c) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + (size_t)i * 8 * K + c * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[i][0] + v[i][3];
            if (s == 77.f) break;        // dependent rounds
        }
    }
    if (s == 123.456f) out[threadIdx.x] = s;
}
int main() {
    const int N = 2048, K = 1536, NSET = 10;
    const size_t n = (size_t)N * K;
    float* W[NSET]; float* out;
    std::vector<float> h(n, 0.001f);
    for (int q = 0; q < NSET; ++q) { hipMalloc(&W[q], n * 4); hipMemcpy(W[q], h.data(), n * 4, hipMemcpyHostToDevice); }
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"empty", "contig upfront 256 WG x 6 loads", "rows dependent 6 rounds, 64 WG-cols x1", "rows upfront, x1",
                           "rows dependent, x4 M-tiles (256 WG)", "rows upfront, x4 M-tiles (256 WG)", "contig upfront 1024 WG x 6 loads (4x re-read)"};
    for (int mode = 0; mode < 7; ++mode) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            for (int r = 0; r < 1000; ++r) {
                const float* w = W[r % NSET];
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, w, out); break;
                case 1: hipLaunchKernelGGL((k_contig<6>), dim3(256), dim3(512), 0, 0, w, out, n / 256); break;
                case 2: hipLaunchKernelGGL((k_rows<false, 1>), dim3(64), dim3(512), 0, 0, w, out, K); break;
                case 3: hipLaunchKernelGGL((k_rows<true, 1>), dim3(64), dim3(512), 0, 0, w, out, K); break;
                case 4: hipLaunchKernelGGL((k_rows<false, 4>), dim3(256), dim3(512), 0, 0, w, out, K); break;
                case 5: hipLaunchKernelGGL((k_rows<true, 4>), dim3(256), dim3(512), 0, 0, w, out, K); break;
                case 6: hipLaunchKernelGGL((k_contig<6>), dim3(1024), dim3(512), 0, 0, w, out, n / 256); break;
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) printf("%-50s %.2f us/launch  (%.2f TB/s of 12.6 MB)\n", names[mode], ms, n * 4 / (ms * 1e-6) / 1e12);
        }
    }
    return 0;
}
