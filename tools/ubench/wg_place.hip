// Which workgroups share a CU?  512 resident WGs (72 KB LDS each) record HW_ID and XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256) place(unsigned* out) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = 1.f;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    }
}
int main() {
    unsigned* o; (void)hipMalloc(&o, 8 * 512);
    (void)hipFuncSetAttribute((const void*)place, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(place, dim3(512), dim3(256), 72 * 1024, 0, o);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(1024); (void)hipMemcpy(h.data(), o, 8 * 512, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int w = 0; w < 512; ++w) {
        const unsigned id = h[2 * w], x = h[2 * w + 1] & 0xF;
        const unsigned cuid = (id >> 8) & 0xF, sh = (id >> 12) & 1, se = (id >> 13) & 7;
        if (w < 24 || (w >= 256 && w < 272)) printf("wg %3d: hw_id %08x xcc %u se %u sh %u cu %u\n", w, id, x, se, sh, cuid);
        cu[(x << 16) | (se << 8) | (sh << 4) | cuid].push_back(w);
    }
    printf("%zu distinct (xcc,se,sh,cu)\n", cu.size());
    int n = 0; for (auto& kv : cu) { if (n++ < 12) { printf("  %06x:", kv.first); for (int w : kv.second) printf(" %d", w); printf("\n"); } }
    return 0;
}
