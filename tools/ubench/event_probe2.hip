// A stream waits for an event that is NOT complete when the host calls hipStreamWaitEvent but IS complete by the time the GPU
// reaches the wait: how long does the waiting stream stall?   stream A: K1 (d1 us), record M.   stream B: n1 short kernels, wait M,
// n2 short kernels; per-kernel start / end stamps.   variants: d1 shorter / longer than B's work in front of the wait; extra
// event records on B in front of the wait (the fork pattern of xg_model.hip).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/event_probe2 tools/ubench/event_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long* stamp) {      // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1;
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    unsigned long long* st; CK(hipHostMalloc(&st, 256 * sizeof(unsigned long long)));
    hipEvent_t M, F[8];
    CK(hipEventCreateWithFlags(&M, hipEventDisableTiming));
    for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&F[i], hipEventDisableTiming));
    const int n1 = 30, n2 = 6;
    for (int variant = 0; variant < 4; ++variant)
        for (int d1 = 100; d1 <= 500; d1 += 200)
            for (int rep = 0; rep < 2; ++rep) {
                for (int i = 0; i < 256; ++i) st[i] = 0;
                CK(hipDeviceSynchronize());
                spin<<<grid, 64, 0, B>>>(1000, st + 0);              // B's first kernel: the time origin (10 us)
                CK(hipEventRecord(F[0], B)); CK(hipStreamWaitEvent(A, F[0], 0));
                spin<<<grid, 64, 0, A>>>(d1 * 100ull, st + 2);       // K1
                CK(hipEventRecord(M, A));
                if (variant == 3) spin<<<grid, 64, 0, A>>>(60000, nullptr);      // more work behind the mark on A
                for (int i = 0; i < n1; ++i) spin<<<grid, 64, 0, B>>>(1000, st + 4 + 2 * i);
                if (variant >= 1) { CK(hipEventRecord(F[1], B)); CK(hipStreamWaitEvent(C, F[1], 0)); spin<<<grid, 64, 0, C>>>(3000, nullptr); }
                if (variant >= 2) { CK(hipEventRecord(F[2], B)); CK(hipStreamWaitEvent(A, F[2], 0)); spin<<<grid, 64, 0, A>>>(3000, nullptr); }
                CK(hipStreamWaitEvent(B, M, 0));
                for (int i = n1; i < n1 + n2; ++i) spin<<<grid, 64, 0, B>>>(1000, st + 4 + 2 * i);
                CK(hipDeviceSynchronize());
                const double t0 = (double)st[0];
                printf("variant %d  K1 %3d us: K1 ran %6.1f .. %6.1f | B before the wait ends %6.1f | B after the wait starts %6.1f (stall %6.1f us) | gaps between B's kernels before the wait: avg %.1f us\n",
                       variant, d1, (st[2] - t0) / 100, (st[3] - t0) / 100, (st[4 + 2 * (n1 - 1) + 1] - t0) / 100, (st[4 + 2 * n1] - t0) / 100,
                       (st[4 + 2 * n1] - (double)st[4 + 2 * (n1 - 1) + 1]) / 100,
                       ((st[4 + 2 * (n1 - 1)] - (double)st[4]) / 100 - 10.0 * (n1 - 1)) / (n1 - 1));
            }
    return 0;
}
