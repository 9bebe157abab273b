// How many 256-thread workgroups are co-resident per CU as a function of dynamic LDS bytes?  512 WGs spin 30 us each;
// the number that start within the first 10 us is the resident set.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) spin(long long* t, float* sink) {
    extern __shared__ float sm[];
    const long long t0 = wall_clock64();
    sm[threadIdx.x] = (float)t0;
    __syncthreads();
    while (wall_clock64() - t0 < 3000) { }
    if (threadIdx.x == 0) { t[blockIdx.x] = t0; sink[blockIdx.x] = sm[5]; }
}
int main() {
    long long* t; float* s; (void)hipMalloc(&t, 8 * 1024); (void)hipMalloc(&s, 4 * 1024);
    for (int kb : {32, 48, 53, 56, 60, 64, 66, 68, 70, 72, 73, 74, 76, 78, 80, 84, 96, 128, 160}) {
        const int bytes = kb * 1024;
        if (hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { printf("%d KB: attr failed\n", kb); continue; }
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), bytes, 0, t, s);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(1024); (void)hipMemcpy(h.data(), t, 8 * 1024, hipMemcpyDeviceToHost);
        long long m = h[0]; for (auto v : h) if (v < m) m = v;
        int n = 0; for (auto v : h) if (v - m < 1000) ++n;
        printf("%3d KB (%6d B): %d of 1024 workgroups start in the first 10 us -> %.2f per CU\n", kb, bytes, n, n / 256.0);
    }
    return 0;
}
