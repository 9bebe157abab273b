// CU-masked streams on MI355X (hipExtStreamCreateWithCUMask): which mask bit is which (XCD, SE, CU), and what two kernels
// cost each other when they share every CU versus when each has its own half of every XCD.
//   part 1: census -- for a few masks, the set of (xcc, se, cu) a 4096-workgroup kernel actually ran on;
//   part 2: a latency-bound "chain" (dependent small launches, 512 workgroups x 256 threads: pointer-chasing loads + a few
//           MFMAs) beside a throughput "background" kernel (one 256-thread MFMA workgroup per CU, persistent for ~2 ms):
//           chain alone, chain beside the background unmasked, and with complementary masks.
// build: hipcc --offload-arch=gfx950 -O3 -o cumask_probe cumask_probe.hip ; run: ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <map>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void census_kernel(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// background: dense fp32 MFMA for `iters` rounds, one wave per SIMD
__global__ void __launch_bounds__(256) bg_kernel(float* out, int iters) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.f) out[0] = acc[1];
}

// chain link: every workgroup streams 64 KB (L2-resident weights), does 16 MFMAs per wave, writes 128 B
__global__ void __launch_bounds__(256) link_kernel(const f32x4* __restrict__ wts, float* out, size_t span4) {
    const f32x4* p = wts + ((size_t)blockIdx.x * 4096) % span4;
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[(r * 4 + u) * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u];
    }
    f32x16 acc = {0};
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s[0], s[1], acc, 0, 0, 0);
    if (threadIdx.x < 32) out[blockIdx.x * 32 + threadIdx.x] = acc[0] + s[2];
}

static int census(hipStream_t st, uint32_t* dev, std::vector<uint32_t>& host, int nwg, const char* name) {
    CK(hipMemsetAsync(dev, 0xff, sizeof(uint32_t) * 2 * nwg, st));
    hipLaunchKernelGGL(census_kernel, dim3(nwg), dim3(64), 0, st, dev, 2000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(host.data(), dev, sizeof(uint32_t) * 2 * nwg, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> per_xcc;      // xcc -> set of (se, sh, cu)
    for (int i = 0; i < nwg; ++i) {
        const uint32_t hw = host[2 * i], xcc = host[2 * i + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%-28s", name);
    for (auto& kv : per_xcc) { printf(" x%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total %d\n", total);
    if (total <= 64) {
        for (auto& kv : per_xcc) { printf("    xcc %d:", kv.first); for (int v : kv.second) printf(" se%d.sh%d.cu%d", v >> 8, (v >> 4) & 1, v & 0xf); printf("\n"); }
    }
    return 0;
}

static int make_stream(hipStream_t* st, const std::vector<uint32_t>& mask) {
    CK(hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int nwg = 4096;
    uint32_t* dev; CK(hipMalloc(&dev, sizeof(uint32_t) * 2 * nwg));
    std::vector<uint32_t> host(2 * nwg);
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    if (census(s0, dev, host, nwg, "unmasked")) return 1;
    struct M { const char* name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    auto bits = [](auto pred) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    masks.push_back({"bits 0-127", bits([](int i) { return i < 128; })});
    masks.push_back({"bits 128-255", bits([](int i) { return i >= 128; })});
    masks.push_back({"even bits", bits([](int i) { return i % 2 == 0; })});
    masks.push_back({"bits 0-7", bits([](int i) { return i < 8; })});
    masks.push_back({"bits 0-31", bits([](int i) { return i < 32; })});
    masks.push_back({"bits i%16<8", bits([](int i) { return i % 16 < 8; })});
    masks.push_back({"bits i%64<32", bits([](int i) { return i % 64 < 32; })});
    masks.push_back({"bits (i/8)%2==0", bits([](int i) { return (i / 8) % 2 == 0; })});
    masks.push_back({"bit 0 only", bits([](int i) { return i == 0; })});
    masks.push_back({"bit 1 only", bits([](int i) { return i == 1; })});
    masks.push_back({"bit 8 only", bits([](int i) { return i == 8; })});
    masks.push_back({"bit 32 only", bits([](int i) { return i == 32; })});
    for (auto& mk : masks) {
        hipStream_t st;
        if (make_stream(&st, mk.m)) return 1;
        if (census(st, dev, host, nwg, mk.name)) return 1;
        CK(hipStreamDestroy(st));
    }

    // ---- part 2: interference
    const size_t span4 = (size_t)(24 << 20) / 16;         // 24 MB of "weights": L2-resident across the chip
    f32x4* wts; float* out; CK(hipMalloc(&wts, span4 * 16)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(wts, 0, span4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain_us = [&](hipStream_t sc, hipStream_t sb, int bg_wgs, int links, int link_wgs) -> double {
        // warm
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(link_kernel, dim3(link_wgs), dim3(256), 0, sc, wts, out, span4);
        hipStreamSynchronize(sc);
        if (sb) hipLaunchKernelGGL(bg_kernel, dim3(bg_wgs), dim3(256), 0, sb, out, 20000);      // 20000*16*64 cycles = 9 ms at 2.2 GHz: outlasts the chain
        hipEventRecord(e0, sc);
        for (int i = 0; i < links; ++i) hipLaunchKernelGGL(link_kernel, dim3(link_wgs), dim3(256), 0, sc, wts, out, span4);
        hipEventRecord(e1, sc);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (sb) hipStreamSynchronize(sb);
        return ms * 1e3 / links;
    };
    auto bg_ms = [&](hipStream_t sb, int bg_wgs) -> double {
        hipLaunchKernelGGL(bg_kernel, dim3(bg_wgs), dim3(256), 0, sb, out, 100);
        hipStreamSynchronize(sb);
        hipEventRecord(e0, sb);
        hipLaunchKernelGGL(bg_kernel, dim3(bg_wgs), dim3(256), 0, sb, out, 2000);
        hipEventRecord(e1, sb);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    printf("\nchain link (512 wgs) alone, unmasked: %.2f us/link\n", chain_us(sa, nullptr, 0, 200, 512));
    printf("background alone, unmasked, 256 wgs x 2000 rounds: %.3f ms\n", bg_ms(sb, 256));
    printf("chain beside background (256 wgs), both unmasked: %.2f us/link\n", chain_us(sa, sb, 256, 200, 512));
    // candidate complementary masks
    std::vector<std::pair<const char*, std::pair<std::vector<uint32_t>, std::vector<uint32_t>>>> pairs;
    pairs.push_back({"lo/hi 128", {bits([](int i) { return i < 128; }), bits([](int i) { return i >= 128; })}});
    pairs.push_back({"even/odd", {bits([](int i) { return i % 2 == 0; }), bits([](int i) { return i % 2 == 1; })}});
    pairs.push_back({"i%16<8 / >=8", {bits([](int i) { return i % 16 < 8; }), bits([](int i) { return i % 16 >= 8; })}});
    pairs.push_back({"i%64<32 / >=32", {bits([](int i) { return i % 64 < 32; }), bits([](int i) { return i % 64 >= 32; })}});
    pairs.push_back({"chain 3/4, bg 1/4 (i%4)", {bits([](int i) { return i % 4 != 0; }), bits([](int i) { return i % 4 == 0; })}});
    pairs.push_back({"chain all, bg half(even)", {bits([](int i) { return true; }), bits([](int i) { return i % 2 == 0; })}});
    for (auto& pr : pairs) {
        hipStream_t mc, mb;
        if (make_stream(&mc, pr.second.first) || make_stream(&mb, pr.second.second)) return 1;
        int nb = 0; for (uint32_t wd : pr.second.second) nb += __builtin_popcount(wd);
        const double ca = chain_us(mc, nullptr, 0, 200, 512);
        const double ba = bg_ms(mb, nb);
        const double cb = chain_us(mc, mb, nb, 200, 512);
        printf("%-26s chain alone %.2f us/link, bg alone (%d wgs) %.3f ms, chain beside bg %.2f us/link\n", pr.first, ca, nb, ba, cb);
        CK(hipStreamDestroy(mc)); CK(hipStreamDestroy(mb));
    }
    return 0;
}
