// Cost of a grid-wide barrier between co-resident workgroups on MI355X (the building block of a persistent recurrence kernel):
// G workgroups x 256 threads, one per CU; each round = [optional: read `bytes` per workgroup of a buffer another workgroup wrote
// in the previous round] + barrier.   variants: 0 = one monotonically increasing counter (atomic add, spin on a load);
// 1 = per-XCD counters + a top-level counter (workgroup i runs on XCD i % 8);  2 = flag array: every workgroup writes its own
// round number, everyone polls all G flags with one coalesced load per wave (no atomics);  3 (round 6) = the XCD-hierarchical barrier
// of MI355X_MICROARCH.md's price list ("barrier-xcd") with hand-placed fences: ONE lane per workgroup releases (buffer_wbl2 sc1 +
// s_waitcnt vmcnt(0)) before it arrives on its XCD's counter with a relaxed atomic, the last arriver of an XCD arrives on the top
// counter, polls it relaxed, then publishes the XCD's generation; every other workgroup polls its XCD's generation word relaxed
// (s_sleep between polls); ONE acquire (buffer_inv sc1) per workgroup behind the poll.  No __threadfence() by 256 threads, no
// acquire loads inside a spin.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/grid_barrier tools/ubench/grid_barrier.hip && tools/ubench/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

template <int VAR>
__global__ void __launch_bounds__(256) bar_kernel(unsigned* ctr, unsigned* flags, float* buf, int rounds, int floats_per_wg, float* sink, long long limit) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    const long long t_start = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (floats_per_wg > 0) {                  // dependent data: written by workgroup (wg + 1) % G in the previous round
            const float* src = buf + (size_t)((wg + 1) % G) * floats_per_wg;
            for (int i = tid * 4; i < floats_per_wg; i += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(src + i);
                acc += v.x + v.y + v.z + v.w;
            }
            float* dst = buf + (size_t)wg * floats_per_wg;
            for (int i = tid * 4; i < floats_per_wg; i += 1024) *reinterpret_cast<float4*>(dst + i) = make_float4(acc, r, wg, tid);
        }
        if (VAR != 3) __threadfence();            // release: this workgroup's stores before the arrival
        __syncthreads();
        if (VAR == 0) {
            if (tid == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)r * G;
                while (ld_acq(ctr) < target) { if (wall_clock64() - t_start > limit) break; __builtin_amdgcn_s_sleep(1); }
            }
        } else if (VAR == 1) {
            if (tid == 0) {
                const int x = wg & 7, nx = (G + 7 - x) / 8;          // workgroups on this XCD
                const unsigned old = __hip_atomic_fetch_add(ctr + 64 * (1 + x), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)r * nx) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)r * (G < 8 ? G : 8);
                while (ld_acq(ctr) < target) { if (wall_clock64() - t_start > limit) break; __builtin_amdgcn_s_sleep(1); }
            }
        } else if (VAR == 3) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int x = wg & 7, nx = (G + 7 - x) / 8, nxcd = G < 8 ? G : 8;
                const unsigned old = __hip_atomic_fetch_add(ctr + 64 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)r * nx) {                   // this XCD's last arriver: the top level
                    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * nxcd) {
                        if (wall_clock64() - t_start > limit) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __hip_atomic_store(ctr + 64 * (9 + x), (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (__hip_atomic_load(ctr + 64 * (9 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
                        if (wall_clock64() - t_start > limit) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            if (tid == 0) __hip_atomic_store(flags + wg, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 64) {
                bool ok = false;
                while (!ok) {
                    ok = true;
                    for (int i = tid; i < G; i += 64) ok = ok && ld_acq(flags + i) >= (unsigned)r;
                    ok = __all(ok);
                    if (wall_clock64() - t_start > limit) break;
                }
            }
        }
        __syncthreads();
        if (VAR != 3) __threadfence();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *ctr, *flags; float *buf, *sink;
    CK(hipMalloc(&ctr, 64 * 32 * sizeof(unsigned))); CK(hipMalloc(&flags, 1024 * sizeof(unsigned)));
    CK(hipMalloc(&buf, (size_t)256 * 65536 * sizeof(float))); CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long limit = 100000000LL * 5;     // 5 s of wall_clock64 (100 MHz): a workgroup that is not co-resident cannot hang the box
    for (int var = 0; var < 4; ++var)
        for (int G : {64, 128, 256})
            for (int kb : {0, 16, 64, 256}) {
                const int fl = kb * 256;          // floats per workgroup
                CK(hipMemset(ctr, 0, 64 * 32 * sizeof(unsigned))); CK(hipMemset(flags, 0, 1024 * sizeof(unsigned)));
                CK(hipMemset(buf, 0, (size_t)256 * 65536 * sizeof(float)));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (var == 0) bar_kernel<0><<<G, 256>>>(ctr, flags, buf, rounds, fl, sink, limit);
                else if (var == 1) bar_kernel<1><<<G, 256>>>(ctr, flags, buf, rounds, fl, sink, limit);
                else if (var == 2) bar_kernel<2><<<G, 256>>>(ctr, flags, buf, rounds, fl, sink, limit);
                else bar_kernel<3><<<G, 256>>>(ctr, flags, buf, rounds, fl, sink, limit);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("variant %d  G=%3d  %3d KB read+written per workgroup per round: %.2f us per round\n", var, G, kb, ms * 1e3 / rounds);
            }
    return 0;
}
