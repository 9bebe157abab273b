// Stand-alone reproducer attempt for the packed-FMA fault of docs/pkfma_hazard.md (round 6).
//
// In skf_kernel (xg_step.hip) the attention job's context loop -- one ds_read_b128 of four softmax weights, four global float2
// loads of V rows, four v_pk_fma_f32 with operand-select modifiers (the form the SLP vectorizer makes of `ax += s * v.x; ay += s * v.y`)
// -- gives run-to-run differences of single context elements when, and only when, the OTHER workgroup on its CU runs split-bf16 cell
// tiles (three bf16 plane images in LDS, 12 x v_mfma_f32_32x32x16_bf16 per 32-deep chunk); never beside fp32 or plain-bf16 tiles.
// This probe rebuilds that pairing outside the library: 512-thread workgroups with the library's LDS footprint (61440 B, two per CU),
// the first half of the grid in role A (the context loop, packed form), the second half in role B (selectable), and compares role A's
// result bit for bit with the same sums from a pinned v_fmac_f32 loop computed by the same workgroup.
//
//   hipcc --offload-arch=gfx950 -O3 -fslp-vectorize -o /tmp/pkfma_repro tools/ubench/pkfma_repro.hip && /tmp/pkfma_repro
//   knobs: argv[1] = launches per configuration (200), -DPK_NOP=<n> (s_nop n between the LDS read's wait and the first packed FMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int R = 512, K = 26, NROW = 13, NWV = 8, LDH = 40, PLH = 32 * LDH;
constexpr int LDS_FLOATS = 8 * (3 * PLH) / 2;          // 61440 bytes: the split-bf16 instantiation's staging area

struct Args {
    const float* V;        // (B, K, R)
    const float* ex;       // (B, K) softmax weights
    float* out_pk;         // (B, 2, R) packed-form sums of the two halves
    float* out_ref;        // (B, 2, R) pinned scalar-form sums
    const float* bsrc;     // role B operand stream (L2-resident)
    float* bsink;
    int B, role_b, iters_b, nrow, reps_a;
    unsigned* nbad;
};

__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xFFFF0000u), sb = rb - __uint_as_float(vb & 0xFFFF0000u);
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t lo; lo[0] = (__bf16)sa; lo[1] = (__bf16)sb;
    p2 = __builtin_bit_cast(unsigned, lo);
}

// role A: the attention job's last phase for video b, half `part` (rows k0 .. k0 + NROW)
__device__ __forceinline__ void role_a(const Args& a, int tile, float* smem) {
    const int b = tile >> 1, part = tile & 1, k0 = part * NROW, nrow = a.nrow;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sx = smem + 64 + 4;
    if (wave == 0 && lane < nrow) sx[lane] = a.ex[(size_t)b * K + k0 + lane];
    __syncthreads();
    const float* Vb = a.V + (size_t)b * K * R;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    unsigned bad = 0;
    for (int rep = 0; rep < a.reps_a; ++rep) {
    for (int c = threadIdx.x * 2; c < R; c += NWV * 128) {
        const float* vp = Vb + (size_t)k0 * R + c;
        // ---- packed form, pinned to the instructions the SLP vectorizer makes of the plain loop inside skf_kernel (ISA of the
        // -DSKF_ATTN_PKFMA -fslp-vectorize build): one ds_read_b128 for four weights, four float2 loads, v_pk_fma_f32 with the weight
        // broadcast by operand select (low half: op_sel_hi:[1,0,1]; high half: op_sel:[0,1,0]), one v_mov for the fourth weight
        f32x2 acc2 = {0.f, 0.f};
        int r = 0;
        for (; r + 4 <= nrow; r += 4) {
            const f32x2 v0 = *reinterpret_cast<const f32x2*>(vp + (size_t)(r + 0) * R);
            const f32x2 v1 = *reinterpret_cast<const f32x2*>(vp + (size_t)(r + 1) * R);
            const f32x2 v2 = *reinterpret_cast<const f32x2*>(vp + (size_t)(r + 2) * R);
            const f32x2 v3 = *reinterpret_cast<const f32x2*>(vp + (size_t)(r + 3) * R);
            const f32x4 w = *reinterpret_cast<const f32x4*>(sx + r);
            f32x2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
#ifdef PK_NOP
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop %0" :: "n"(PK_NOP) : "memory");
#endif
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(v0), "v"(w01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc2) : "v"(v1), "v"(w01));
            w01[0] = w23[1];                                                          // (the v_mov into the first pair's low register)
            asm volatile("" : "+v"(w01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(v2), "v"(w23));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(v3), "v"(w01));
        }
        for (; r < nrow; ++r) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(vp + (size_t)r * R);
            f32x2 wq = {sx[r], 0.f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(v), "v"(wq));
        }
        const float ax = acc2[0], ay = acc2[1];
        // ---- reference: the same sums, plain v_fmac_f32, pinned
        float bx = 0.f, by = 0.f;
        for (int q = 0; q < nrow; ++q) {
            const float2 v = *reinterpret_cast<const float2*>(vp + (size_t)q * R);
            const float s = sx[q];
            asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(bx), "+v"(by) : "v"(s), "v"(v.x), "v"(v.y));
        }
        bad += (ax != bx) + (ay != by);
        float* o = a.out_pk + ((size_t)b * 2 + part) * R + c;  o[0] = ax; o[1] = ay;
        float* p = a.out_ref + ((size_t)b * 2 + part) * R + c; p[0] = bx; p[1] = by;
    }
    }
    if (bad) atomicAdd(a.nbad, bad);
}

// role B: what a cell tile's waves do per 32-deep chunk.  1: fp32 (LDS image + 16 x 32x32x2 MFMA), 2: plain bf16 (2 MFMAs),
// 3: split-bf16 (three plane images, 12 x 32x32x16 bf16 MFMA)
__device__ __forceinline__ void role_b(const Args& a, int tile, float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    float* As = smem + wave * ((3 * PLH) / 2);
    const float* src = a.bsrc + (size_t)(tile & 63) * 4096 + wave * 512;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int it = 0; it < a.iters_b; ++it) {
        f32x4 ra[4], rb[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(src + ((it & 7) * 64 + i * 16 + (lane >> 2)) * 4 % 4096);
#pragma unroll
        for (int i = 0; i < 6; ++i) rb[i] = *reinterpret_cast<const f32x4*>(a.bsrc + ((size_t)(it & 15) * 6 + i) * 256 + lane * 4);
        if (a.role_b == 3) {
            unsigned short* lds = reinterpret_cast<unsigned short*>(As);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned a0, a1, a2, b0, b1, b2;
                split3_pair(ra[i][0], ra[i][1], a0, a1, a2);
                split3_pair(ra[i][2], ra[i][3], b0, b1, b2);
                unsigned short* q = lds + (i * 8 + (lane >> 3)) * LDH + ((lane & 7) << 2);
                *reinterpret_cast<uint2*>(q) = make_uint2(a0, b0);
                *reinterpret_cast<uint2*>(q + PLH) = make_uint2(a1, b1);
                *reinterpret_cast<uint2*>(q + 2 * PLH) = make_uint2(a2, b2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8 bp3[3], ap3[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    bp3[q] = __builtin_bit_cast(bf16x8, rb[3 * j + q]);
                    ap3[q] = *reinterpret_cast<const bf16x8*>(lds + q * PLH + l31 * LDH + half * 16 + j * 8);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[0], bp3[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[2], bp3[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[1], bp3[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[0], bp3[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[1], bp3[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[0], bp3[0], acc, 0, 0, 0);
            }
        } else if (a.role_b == 2) {
            unsigned short* lds = reinterpret_cast<unsigned short*>(As);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                bf16x2_t lo, hi;
                lo[0] = (__bf16)ra[i][0]; lo[1] = (__bf16)ra[i][1]; hi[0] = (__bf16)ra[i][2]; hi[1] = (__bf16)ra[i][3];
                *reinterpret_cast<uint2*>(lds + (i * 8 + (lane >> 3)) * LDH + ((lane & 7) << 2)) =
                    make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(lds + l31 * LDH + i * 16 + half * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, rb[i]), acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + (i * 8 + (lane >> 3)) * 36 + ((lane & 7) << 2)) = ra[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(As + l31 * 36 + half * 16 + i * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], rb[i][kk], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (acc[0] == 123.456f) a.bsink[threadIdx.x] = acc[1] + acc[7];
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) repro_kernel(Args a) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    const int na = 2 * a.B;
    if ((int)blockIdx.x < na) role_a(a, blockIdx.x, smem);
    else if (a.role_b) role_b(a, blockIdx.x - na, smem);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int B = 128;
    std::vector<float> hV((size_t)B * K * R), hex((size_t)B * K), hb(1 << 20);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : hV) v = rnd();
    for (auto& v : hex) v = 0.05f + rnd();
    for (auto& v : hb) v = rnd() - 0.5f;
    float *dV, *dex, *dpk, *dref, *dbs, *dsink; unsigned* dbad;
    CK(hipMalloc(&dbad, 4));
    CK(hipMalloc(&dV, hV.size() * 4)); CK(hipMalloc(&dex, hex.size() * 4)); CK(hipMalloc(&dpk, (size_t)B * 2 * R * 4));
    CK(hipMalloc(&dref, (size_t)B * 2 * R * 4)); CK(hipMalloc(&dbs, hb.size() * 4)); CK(hipMalloc(&dsink, 4096));
    CK(hipMemcpy(dV, hV.data(), hV.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dex, hex.data(), hex.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbs, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> pk((size_t)B * 2 * R), ref((size_t)B * 2 * R), first;
    const char* names[] = {"alone (no second workgroup on the CU)", "fp32 tiles (32x32x2 f32 MFMA)", "plain bf16 tiles", "split-bf16 tiles (3 planes, 12 bf16 MFMAs per chunk)"};
    printf("role A: %d workgroups (two per video), packed context loop vs pinned v_fmac_f32 loop, %d launches per configuration\n", 2 * B, launches);
    const int reps_a = argc > 2 ? atoi(argv[2]) : 300;
    for (int role = 0; role < 4; ++role) {
        for (int iters : {2000}) {
            long bad_ref = 0, launches_bad = 0;
            for (int l = 0; l < launches; ++l) {
                Args a{dV, dex, dpk, dref, dbs, dsink, B, role, iters, NROW, reps_a, dbad};
                CK(hipMemset(dbad, 0, 4));
                hipLaunchKernelGGL(repro_kernel, dim3(role ? 4 * B : 2 * B), dim3(512), 0, 0, a);
                CK(hipDeviceSynchronize());
                unsigned nb = 0;
                CK(hipMemcpy(&nb, dbad, 4, hipMemcpyDeviceToHost));
                bad_ref += nb; launches_bad += nb ? 1 : 0;
            }
            printf("second workgroup: %-58s: %ld launches of %d with a wrong element; wrong elements %ld of %.3g evaluated\n",
                   names[role], launches_bad, launches, bad_ref, (double)launches * reps_a * 2 * B * R);
        }
    }
    return 0;
}
