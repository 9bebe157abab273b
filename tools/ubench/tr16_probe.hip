// What does ds_read_b64_tr_b16 return?  LDS[i] = i (16-bit); lane l reads at a per-lane byte address; print 4 x u16 per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;                         // element index (16-bit units) this lane points at
    if (mode == 0) elem = l * 4;                                  // lane-linear, 8 B apart
    else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;      // 16 rows of 64 elements; 16-lane group g reads cols 4g..4g+3
    else elem = (l & 3) * 64 + (l >> 2) * 4;                      // rows by l&3
    typedef __attribute__((address_space(3))) v4s lds_v4s;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; (void)hipMalloc(&d, 512);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        unsigned short h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); if (l % 4 == 3) printf("\n"); }
    }
    return 0;
}
