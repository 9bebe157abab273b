// Prototype of the packed-weight skinny kernel: weights pre-tiled into MFMA-fragment order so the B operand goes
// global -> VGPR with 1 KB-coalesced loads (no LDS), the A operand through a wave-private LDS image (MODE 0) or also
// fragment-ordered (MODE 1).  Shape = cell 2 of the decoder step: M = 128, N = 2048, K = 3 x 512.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMODE=1] [-DDEPTH=2] -o skp_bench skp_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MODE
#define MODE 0
#endif
#ifndef DEPTH
#define DEPTH 1
#endif
#ifndef WAVES
#define WAVES 8
#endif
constexpr int SKW = WAVES, SKT = SKW * 64, CK = 32, LDR = CK + 4, OPF = 32 * LDR;

struct Seg { const float* A; const float* Bp; int lda; int K; };
struct Args { Seg seg[3]; int nseg; int M, N; float* C; int ldc; };

// packed B: block (nt, kc) of 1024 floats: [i(4)][h(2)][n(32)][4] ; blocks of one n-tile are contiguous over kc
// packed A (MODE 1): block (mt, kc): [i(4)][h(2)][m(32)][4]
__device__ long long trace_buf[1024 * 8];
#ifdef TRACE
#define STAMP(i) do { if (threadIdx.x == 0) trace_buf[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
__global__ void __launch_bounds__(SKT) __attribute__((amdgpu_waves_per_eu(4, 4))) skp_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) float smem[SKW * 32 * 33 > SKW * OPF ? SKW * 32 * 33 : SKW * OPF];
    STAMP(0);
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntm = (a.M + 31) >> 5;
    const int tm = bid % ntm, tn = bid / ntm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    float* As = smem + wave * OPF;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int nc_total = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) if (s < a.nseg) nc_total += a.seg[s].K / CK;
    const int wc0 = (wave * nc_total) / SKW, wc1 = ((wave + 1) * nc_total) / SKW;
    int seg_start = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= a.nseg) break;
        const Seg sg = a.seg[s];
        const int nc = sg.K / CK;
        const int c0 = max(wc0, seg_start) - seg_start, c1 = min(wc1, seg_start + nc) - seg_start;
        seg_start += nc;
        if (c0 >= c1) continue;
        // B: this lane's 16-B piece i of chunk c: Bp + ((tn * nc + c) * 8 + i*2 + h) * 128 + l31*4   (floats)
        const float* bp = sg.Bp + ((size_t)tn * nc) * 1024 + (size_t)(half * 32 + l31) * 4;
#if MODE == 1
        const float* ap = sg.A + ((size_t)tm * nc) * 1024 + (size_t)(half * 32 + l31) * 4;
#else
        const int lrow = lane >> 3, lcol = (lane & 7) << 2;
        const float* ap[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = sg.A + (size_t)min(tm * 32 + i * 8 + lrow, a.M - 1) * sg.lda + lcol;
#endif
        f32x4 ra[DEPTH][4], rb[DEPTH][4];
        if (s == 0) STAMP(1);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int c = c0 + d;
            if (c < c1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rb[d][i] = *reinterpret_cast<const f32x4*>(bp + (size_t)c * 1024 + i * 256);
#if MODE == 1
                    ra[d][i] = *reinterpret_cast<const f32x4*>(ap + (size_t)c * 1024 + i * 256);
#else
                    ra[d][i] = *reinterpret_cast<const f32x4*>(ap[i] + (size_t)c * CK);
#endif
                }
            }
        }
        for (int cb = c0; cb < c1; cb += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int c = cb + d;
                if (c >= c1) break;
                f32x4 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[i] = rb[d][i];
#if MODE == 1
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = ra[d][i];
#else
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4*>(As + (i * 8 + (lane >> 3)) * LDR + ((lane & 7) << 2)) = ra[d][i];
#endif
                if (c + DEPTH < c1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        rb[d][i] = *reinterpret_cast<const f32x4*>(bp + (size_t)(c + DEPTH) * 1024 + i * 256);
#if MODE == 1
                        ra[d][i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(c + DEPTH) * 1024 + i * 256);
#else
                        ra[d][i] = *reinterpret_cast<const f32x4*>(ap[i] + (size_t)(c + DEPTH) * CK);
#endif
                    }
                }
#if MODE != 1
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const f32x4*>(As + l31 * LDR + half * 16 + i * 4);
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[i][kk], acc, 0, 0, 0);
#if MODE != 1
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#endif
            }
        }
    }
    STAMP(3);
    __syncthreads();
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();
    STAMP(4);
#pragma unroll
    for (int e = 0; e < 1024 / SKT; ++e) {
        const int idx = threadIdx.x + SKT * e;
        const int m = idx >> 5, c = idx & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SKW; ++w) v += red[w][m][c];
        const int row = tm * 32 + m, col = tn * 32 + c;
        if (row < a.M && col < a.N) a.C[(size_t)row * a.ldc + col] = v;
    }
    STAMP(5);
}

static float* dev(const std::vector<float>& h) {
    float* p; (void)hipMalloc(&p, h.size() * 4); (void)hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice); return p;
}
// host packers
static std::vector<float> packB(const std::vector<float>& W, int N, int K) {   // W (N,K) row-major
    std::vector<float> p((size_t)N * K);
    const int nc = K / 32;
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
        const int nt = n / 32, nn = n % 32, kc = k / 32, kk = k % 32, h = kk / 16, i = (kk % 16) / 4, q = kk % 4;
        p[((size_t)(nt * nc + kc) * 8 + i * 2 + h) * 128 + nn * 4 + q] = W[(size_t)n * K + k];
    }
    return p;
}
int main() {
    const int M = 128, N = 2048, Ks = 512, NSET = 8;
    srand(1);
    std::vector<float> hA[3], hW[3];
    float* dA[3]; float* dAp[3]; float* dW[NSET][3];
    for (int s = 0; s < 3; ++s) {
        hA[s].resize((size_t)M * Ks); hW[s].resize((size_t)N * Ks);
        for (auto& v : hA[s]) v = (rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hW[s]) v = (rand() % 2001 - 1000) * 1e-3f;
        dA[s] = dev(hA[s]);
        dAp[s] = dev(packB(hA[s], M, Ks));
        auto pk = packB(hW[s], N, Ks);
        for (int q = 0; q < NSET; ++q) dW[q][s] = dev(pk);
    }
    float* C; (void)hipMalloc(&C, (size_t)M * N * 4);
    auto launch = [&](int q) {
        Args a{};
        a.nseg = 3; a.M = M; a.N = N; a.C = C; a.ldc = N;
        for (int s = 0; s < 3; ++s) a.seg[s] = Seg{MODE == 1 ? dAp[s] : dA[s], dW[q][s], Ks, Ks};
        hipLaunchKernelGGL(skp_kernel, dim3((M / 32) * (N / 32)), dim3(SKT), 0, 0, a);
    };
    launch(0);
    (void)hipDeviceSynchronize();
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 200; ++t) {
        const int m = rand() % M, n = rand() % N;
        double r = 0;
        for (int s = 0; s < 3; ++s) for (int k = 0; k < Ks; ++k) r += (double)hA[s][(size_t)m * Ks + k] * hW[s][(size_t)n * Ks + k];
        maxerr = fmax(maxerr, fabs(r - hC[(size_t)m * N + n]));
    }
    printf("MODE %d DEPTH %d WAVES %d: max err vs host %.3g\n", MODE, DEPTH, WAVES, maxerr);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < 1000; ++r) launch(r % NSET);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it == 2) printf("cell2-shape packed kernel: %.2f us/launch\n", ms);
    }
#ifdef TRACE
    (void)hipDeviceSynchronize();
    for (int r = 0; r < 5; ++r) launch(r % NSET);
    (void)hipDeviceSynchronize();
    static long long h[1024 * 8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(trace_buf), sizeof(h));
    const int nwg = 256;
    long long t0 = h[0], t5 = 0;
    for (int w = 0; w < nwg; ++w) { if (h[w * 8] < t0) t0 = h[w * 8]; if (h[w * 8 + 5] > t5) t5 = h[w * 8 + 5]; }
    printf("span first-entry -> last-exit %.2f us\n", (t5 - t0) * 0.01);
    for (int w : {0, 1, 97, nwg - 1})
        printf("   wg %3d: entry +%.2f | prologue %.2f | k loop (incl first wait) %.2f | barrier+reduce %.2f | epilogue %.2f\n", w, (h[w * 8] - t0) * 0.01,
               (h[w * 8 + 1] - h[w * 8]) * 0.01, (h[w * 8 + 3] - h[w * 8 + 1]) * 0.01, (h[w * 8 + 4] - h[w * 8 + 3]) * 0.01,
               (h[w * 8 + 5] - h[w * 8 + 4]) * 0.01);
#endif
    return 0;
}
