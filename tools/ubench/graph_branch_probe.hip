// Does a HIP graph with PARALLEL BRANCHES built from explicit kernel nodes (no fork-join stream capture) replay its branches
// concurrently on this ROCm, and what does a replay cost?  Three chains of `spin` kernels (64 workgroups each, ~20 us) with a
// cross edge every 10 links; built (a) directly with hipGraphAddKernelNode, (b) by adding the same nodes to a graph that is
// being stream-captured on ONE stream (hipStreamGetCaptureInfo_v2 / hipStreamUpdateCaptureDependencies) -- the form a library
// needs when its caller captures (torch.cuda.graph) and the library wants its side-stream work inside the same graph.
// Reference: the same kernels eagerly on three streams with events.
// build: hipcc --offload-arch=gfx950 -O3 -o graph_branch_probe graph_branch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); return 1; } } while (0)

__global__ void spin(long long cycles, int* out, int tag) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(4); }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out + tag, 1);
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int CH = 3, LINKS = 60;

static int add_nodes(hipGraph_t g, const std::vector<hipGraphNode_t>& roots, int* out, long long cyc, std::vector<hipGraphNode_t>& tails) {
    hipGraphNode_t last[CH];
    bool has[CH] = {false, false, false};
    for (int l = 0; l < LINKS; ++l)
        for (int c = 0; c < CH; ++c) {
            std::vector<hipGraphNode_t> deps;
            if (has[c]) deps.push_back(last[c]); else deps = roots;
            if (l % 10 == 0 && l > 0) deps.push_back(last[(c + 1) % CH]);      // cross edge: chain c waits for chain c+1's previous link
            int tag = c;
            void* args[] = {&cyc, &out, &tag};
            hipKernelNodeParams p{};
            p.func = reinterpret_cast<void*>(spin); p.gridDim = dim3(64); p.blockDim = dim3(64); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
            hipGraphNode_t n;
            CK(hipGraphAddKernelNode(&n, g, deps.data(), deps.size(), &p));
            last[c] = n; has[c] = true;
        }
    tails.assign(last, last + CH);
    return 0;
}

int main() {
    int* out; CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    const long long cyc = 2000;                 // wall_clock64 ticks at 100 MHz: 20 us
    hipStream_t s[CH]; for (auto& q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    hipEvent_t ev[CH][LINKS]; for (auto& r : ev) for (auto& e : r) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // ---- eager, three streams
    auto eager = [&]() -> int {
        for (int l = 0; l < LINKS; ++l)
            for (int c = 0; c < CH; ++c) {
                if (l % 10 == 0 && l > 0) CK(hipStreamWaitEvent(s[c], ev[(c + 1) % CH][l - 1], 0));
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s[c], cyc, out, c);
                CK(hipEventRecord(ev[c][l], s[c]));
            }
        return 0;
    };
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now_ms();
        if (eager()) return 1;
        const double t1 = now_ms();
        for (auto& q : s) CK(hipStreamSynchronize(q));
        const double t2 = now_ms();
        if (rep == 2) printf("eager 3 streams            : %.3f ms total, host enqueue %.3f ms (%d kernels; one chain alone = %.2f ms)\n", t2 - t0, t1 - t0, CH * LINKS, LINKS * 0.02);
    }
    // ---- (a) explicit graph
    {
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> tails;
        if (add_nodes(g, {}, out, cyc, tails)) return 1;
        hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 4; ++rep) {
            const double t0 = now_ms();
            CK(hipGraphLaunch(ex, s[0]));
            const double t1 = now_ms();
            CK(hipStreamSynchronize(s[0]));
            const double t2 = now_ms();
            if (rep == 3) printf("explicit nodes, 3 branches : %.3f ms total, host launch %.3f ms\n", t2 - t0, t1 - t0);
        }
    }
    // ---- (b) nodes added to a graph under single-stream capture
    {
        hipGraph_t g = nullptr;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s[0], cyc, out, 3);           // something captured the ordinary way
        hipStreamCaptureStatus st; unsigned long long id; const hipGraphNode_t* deps; size_t ndeps;
        CK(hipStreamGetCaptureInfo_v2(s[0], &st, &id, &g, &deps, &ndeps));
        std::vector<hipGraphNode_t> roots(deps, deps + ndeps), tails;
        if (add_nodes(g, roots, out, cyc, tails)) return 1;
        CK(hipStreamUpdateCaptureDependencies(s[0], tails.data(), tails.size(), hipStreamSetCaptureDependencies));
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s[0], cyc, out, 4);           // ... and something behind the manual nodes
        CK(hipStreamEndCapture(s[0], &g));
        hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 4; ++rep) {
            const double t0 = now_ms();
            CK(hipGraphLaunch(ex, s[0]));
            const double t1 = now_ms();
            CK(hipStreamSynchronize(s[0]));
            const double t2 = now_ms();
            if (rep == 3) printf("nodes inside a capture     : %.3f ms total, host launch %.3f ms\n", t2 - t0, t1 - t0);
        }
    }
    int h[8]; CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf("kernel counts per tag: %d %d %d | captured head %d tail %d\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
