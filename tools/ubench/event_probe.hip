// Does a stream that waits for an event recorded BETWEEN two kernels of another stream resume when the first kernel ends, or only
// when the other stream's queue has drained up to the point the host had reached when it called hipStreamWaitEvent?
//   stream A: K1 (d1 us)  record E  K2 (d2 us)        stream B: K0 (short)  wait E  K3 (short, stamps its start)
// variants: wait called before / after K2 is enqueued; event with / without hipEventDisableTiming.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/event_probe tools/ubench/event_probe.hip && /tmp/event_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long* stamp) {      // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    unsigned long long* st; CK(hipHostMalloc(&st, 64 * sizeof(unsigned long long)));
    for (int flags = 0; flags < 2; ++flags)
        for (int late = 0; late < 3; ++late) {
            hipEvent_t E; CK(hipEventCreateWithFlags(&E, flags ? hipEventDisableTiming : hipEventDefault));
            for (int rep = 0; rep < 3; ++rep) {
                for (int i = 0; i < 8; ++i) st[i] = 0;
                CK(hipDeviceSynchronize());
                spin<<<1, 64, 0, B>>>(100, nullptr);
                spin<<<1, 64, 0, A>>>(20000, st + 0);               // K1: 200 us
                CK(hipEventRecord(E, A));
                if (late == 0) CK(hipStreamWaitEvent(B, E, 0));     // wait enqueued before K2 exists
                spin<<<1, 64, 0, A>>>(60000, st + 2);               // K2: 600 us
                if (late == 2) usleep(100);                         // K2 surely in the hardware queue (K1 still running)
                if (late >= 1) CK(hipStreamWaitEvent(B, E, 0));
                spin<<<1, 64, 0, B>>>(100, st + 4);                 // K3
                CK(hipDeviceSynchronize());
                printf("event %-14s wait %-28s  K1 end %7.1f us   K2 end %7.1f us   K3 start %7.1f us (after K1's start)\n",
                       flags ? "DisableTiming" : "Default", late == 0 ? "before K2 enqueued" : late == 1 ? "after K2 enqueued" : "after K2 enqueued + 100 us",
                       (st[1] - st[0]) / 100.0, (st[3] - st[0]) / 100.0, (st[4] - st[0]) / 100.0);
            }
            CK(hipEventDestroy(E));
        }
    return 0;
}
