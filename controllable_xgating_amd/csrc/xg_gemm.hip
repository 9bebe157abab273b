// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 157 TF peak).
//
// One kernel template covers the three operand layouts the decoder needs:
//   forward  Linear   Y[M,N]  = X[M,K]   * W[N,K]^T      A k-contiguous, B k-contiguous
//   data grad         dX[M,K] = dY[M,N]  * W[N,K]        A k-contiguous, B n-contiguous
//   weight grad       dW[N,K] = dY[M,N]^T * X[M,K]       A m-contiguous, B n-contiguous
// Tile BMxBNx32, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 32x32 MFMA tiles.
// Global -> registers -> LDS staging, double-buffered LDS, one barrier per 32-deep slab; the
// next slab's global loads are in flight while the current slab's MFMAs run.
//
// LDS images (conflict-free for the MFMA operand reads, MI355X LDS banking):
//   k-contiguous operand: [rows][36]   -> each lane reads ONE ds_read_b128 = 4 consecutive k
//                                         (row stride 36 floats = 9 16-B slots, odd -> the 16
//                                         lanes of a b128 group hit 16 distinct slots)
//   m-contiguous operand: [32][rows+4] -> each lane reads 4 ds_read_b32, 32 consecutive floats
//                                         per half-wave.
// The k index inside an 8-deep block is permuted (lane half h owns k = 4h..4h+3) identically
// for A and B, which leaves the dot product unchanged.
#include "xg_common.h"
#include "xg_kernels.h"
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BKS = 32;          // slab depth
constexpr int LDK = BKS + 4;     // row stride of a k-contiguous LDS image

template <int ROWS, bool KC>
struct TileGeom {
    static constexpr int lds_floats = KC ? ROWS * LDK : BKS * (ROWS + 4);
    static constexpr int nvec = ROWS * BKS / 4 / 256;   // float4 per thread per slab
};

// Load one ROWSx32 operand tile from global into registers (nvec float4 per thread).
//  KC : element (r,k) at P[r*ld + k]      (k contiguous)
// !KC : element (r,k) at P[k*ld + r]      (r contiguous)
template <int ROWS, bool KC, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int r0, int k0, int nrows, int K,
                                          f32x4 (&regs)[TileGeom<ROWS, KC>::nvec]) {
    constexpr int NV = TileGeom<ROWS, KC>::nvec;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int f = t + 256 * i;
        int r, k;
        if (KC) { r = f >> 3; k = (f & 7) << 2; }
        else    { k = f / (ROWS / 4); r = (f % (ROWS / 4)) << 2; }
        const int gr = r0 + r, gk = k0 + k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            if (gr < nrows) {
                const float* src = P + (size_t)gr * ld + gk;
                if (VEC) { if (gk < K) v = *reinterpret_cast<const f32x4*>(src); }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gk + j < K) v[j] = src[j];
                }
            }
        } else {
            if (gk < K) {
                const float* src = P + (size_t)gk * ld + gr;
                if (VEC) { if (gr < nrows) v = *reinterpret_cast<const f32x4*>(src); }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gr + j < nrows) v[j] = src[j];
                }
            }
        }
        regs[i] = v;
    }
}

// ---- fast path of the vector kernels: per-thread BYTE offsets are computed once (rows clamped into the operand, so every
// address is valid and nothing is predicated -- rows past the edge only feed output rows / columns that are never stored),
// and a slab's loads are `uniform base + 32-bit lane offset`: no address arithmetic and no exec masking in the K loop
// (each non-MFMA VALU instruction there is issue time the matrix pipe does not get back).  A partial last slab takes
// the predicated loader above.  Requires every offset < 2^31 (checked by the launcher: GemmArgs::fast).
template <int ROWS, bool KC>
__device__ __forceinline__ void tile_offsets(int ld, int r0, int nrows, uint32_t (&off)[TileGeom<ROWS, KC>::nvec]) {
    constexpr int NV = TileGeom<ROWS, KC>::nvec;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            const int r = f >> 3, k = (f & 7) << 2;
            off[i] = (uint32_t)(((size_t)min(r0 + r, nrows - 1) * ld + k) * sizeof(float));
        } else {
            const int k = f / (ROWS / 4), r = (f % (ROWS / 4)) << 2;
            off[i] = (uint32_t)(((size_t)k * ld + min(r0 + r, nrows - 4)) * sizeof(float));
        }
    }
}
template <int NV>
__device__ __forceinline__ void load_fast(const char* __restrict__ base, const uint32_t (&off)[NV], f32x4 (&regs)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) regs[i] = *reinterpret_cast<const f32x4*>(base + off[i]);
}

template <int ROWS, bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const f32x4 (&regs)[TileGeom<ROWS, KC>::nvec]) {
    constexpr int NV = TileGeom<ROWS, KC>::nvec;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            const int r = f >> 3, k = (f & 7) << 2;
            *reinterpret_cast<f32x4*>(lds + r * LDK + k) = regs[i];
        } else {
            const int k = f / (ROWS / 4), r = (f % (ROWS / 4)) << 2;
            *reinterpret_cast<f32x4*>(lds + k * (ROWS + 4) + r) = regs[i];
        }
    }
}

// Fragment for MFMA: 4 k-values (k = kb*8 + 4*half + j) of row `row`.
template <int ROWS, bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ lds, int row, int kb, int half) {
    if (KC) {
        return *reinterpret_cast<const f32x4*>(lds + row * LDK + kb * 8 + half * 4);
    } else {
        const float* p = lds + (kb * 8 + half * 4) * (ROWS + 4) + row;
        f32x4 v;
        v[0] = p[0];
        v[1] = p[ROWS + 4];
        v[2] = p[2 * (ROWS + 4)];
        v[3] = p[3 * (ROWS + 4)];
        return v;
    }
}

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, relu, accumulate;
    int splitk;   // > 1: the K slabs are divided over `splitk` workgroups per tile; partial tiles are added
                  // into C with hardware fp32 atomics (C pre-zeroed by the launcher unless accumulating)
    int fast;     // vector kernels: offset-based unpredicated loads for the full slabs (all byte offsets < 2^31)
};

#ifdef GEMM_CLK
__device__ long long gemm_clk_buf[4];
#endif
template <int BM, int BN, bool AKC, bool BKC, bool VEC>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
#ifdef GEMM_CLK
    const long long clk0 = clock64(), wall0 = wall_clock64();
#endif
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
    constexpr int MT = WM / 32, NT = WN / 32;    // MFMA tiles per wave
    constexpr int A_FL = TileGeom<BM, AKC>::lds_floats;
    constexpr int B_FL = TileGeom<BN, BKC>::lds_floats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = A_FL + B_FL;           // one LDS stage = A tile then B tile

    // XCD-aware tile order: consecutive tiles (sharing an A row panel) stay on one XCD's L2.
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int nwg = ntm * ntn * g.splitk;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ks = bid % g.splitk;          // K split index (fastest: the splits of one tile share an XCD)
    bid /= g.splitk;
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[TileGeom<BM, AKC>::nvec], rb[TileGeom<BN, BKC>::nvec];
    const int nslab_all = (g.K + BKS - 1) / BKS;
    const int s_begin = (int)(((long)ks * nslab_all) / g.splitk), s_end = (int)(((long)(ks + 1) * nslab_all) / g.splitk);
    const int nslab = s_end;
    constexpr int NVA = TileGeom<BM, AKC>::nvec, NVB = TileGeom<BN, BKC>::nvec;
    uint32_t offA[NVA], offB[NVB];
    const bool fast = VEC && g.fast;
    if (fast) {
        tile_offsets<BM, AKC>(g.lda, m0, g.M, offA);
        tile_offsets<BN, BKC>(g.ldb, n0, g.N, offB);
    }
    const size_t stepA = (AKC ? (size_t)BKS : (size_t)BKS * g.lda) * sizeof(float);   // bytes per slab
    const size_t stepB = (BKC ? (size_t)BKS : (size_t)BKS * g.ldb) * sizeof(float);
    const int nfull = g.K / BKS;                 // slabs below nfull are complete
    load_tile<BM, AKC, VEC>(g.A, g.lda, m0, s_begin * BKS, g.M, g.K, ra);
    load_tile<BN, BKC, VEC>(g.B, g.ldb, n0, s_begin * BKS, g.N, g.K, rb);
    store_tile<BM, AKC>(smem, ra);
    store_tile<BN, BKC>(smem + A_FL, rb);
    __syncthreads();

    for (int s = s_begin; s < nslab; ++s) {
        const int cur = (s - s_begin) & 1;
#ifndef GEMM_NO_GLOBAL
        if (s + 1 < nslab) {
            if (fast && s + 1 < nfull) {
                load_fast<NVA>(reinterpret_cast<const char*>(g.A) + (size_t)(s + 1) * stepA, offA, ra);
                load_fast<NVB>(reinterpret_cast<const char*>(g.B) + (size_t)(s + 1) * stepB, offB, rb);
            } else {
                load_tile<BM, AKC, VEC>(g.A, g.lda, m0, (s + 1) * BKS, g.M, g.K, ra);
                load_tile<BN, BKC, VEC>(g.B, g.ldb, n0, (s + 1) * BKS, g.N, g.K, rb);
            }
        }
#endif
        const float* as = smem + cur * STAGE;
        const float* bs = as + A_FL;
#pragma unroll
        for (int kb = 0; kb < BKS / 8; ++kb) {
            f32x4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = read_frag<BM, AKC>(as, wm * WM + i * 32 + l31, kb, half);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = read_frag<BN, BKC>(bs, wn * WN + j * 32 + l31, kb, half);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
#ifndef GEMM_NO_LDS_STORE
        if (s + 1 < nslab) {
            store_tile<BM, AKC>(smem + (cur ^ 1) * STAGE, ra);
            store_tile<BN, BKC>(smem + (cur ^ 1) * STAGE + A_FL, rb);
        }
#endif
#ifndef GEMM_NO_SYNC
        __syncthreads();
#endif
    }

#ifdef GEMM_CLK
    if (blockIdx.x == 100 && threadIdx.x == 0) { gemm_clk_buf[0] = clock64() - clk0; gemm_clk_buf[1] = wall_clock64() - wall0; }
#endif
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (g.bias && ks == 0) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.M) {
                    float* dst = g.C + (size_t)row * g.ldc + col;
                    float v = acc[i][j][r] + bv;
                    if (g.splitk > 1) { unsafeAtomicAdd(dst, v); continue; }
                    if (g.accumulate) v += *dst;
                    if (g.relu) v = fmaxf(v, 0.f);
                    *dst = v;
                }
            }
        }
}

template <int BM, int BN, bool AKC, bool BKC, bool VEC>
int launch(hipStream_t st, const GemmArgs& g) {
    const int ntm = xg_cdiv(g.M, BM), ntn = xg_cdiv(g.N, BN);
    if (g.splitk > 1 && !g.accumulate) {      // partial tiles are atomically added: start from zero
        if (g.ldc == g.N) { if (hipMemsetAsync(g.C, 0, sizeof(float) * (size_t)g.M * g.N, st) != hipSuccess) return XG_EHIP; }
        else if (hipMemset2DAsync(g.C, sizeof(float) * g.ldc, 0, sizeof(float) * g.N, g.M, st) != hipSuccess) return XG_EHIP;
    }
    const size_t lds = 2 * (TileGeom<BM, AKC>::lds_floats + TileGeom<BN, BKC>::lds_floats) * sizeof(float);
    if (lds > 65536) {
        static std::atomic<unsigned> optin{0};
        XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&gemm_kernel<BM, BN, AKC, BKC, VEC>), (int)lds));
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AKC, BKC, VEC>), dim3(ntm * ntn * g.splitk), dim3(256), lds, st, g);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

template <bool AKC, bool BKC>
int dispatch(hipStream_t st, GemmArgs g, bool vec) {
    const long t128 = (long)xg_cdiv(g.M, 128) * xg_cdiv(g.N, 128);
    const long t64 = (long)xg_cdiv(g.M, 64) * xg_cdiv(g.N, 64);
    const int nslab = xg_cdiv(g.K, BKS);
    g.splitk = 1;
    bool big = false;
    // measured on MI355X (tools/ubench/gemm_bench.py): 128x128 wins with >= ~400 tiles (x2 split when the reduction is
    // deep), and for FEW tiles under a very deep reduction (dH = dlogits W: 84 tiles, K = 20000) when split-K fills exactly
    // one round of 512 workgroups (87.6 -> 104 TF); otherwise 64x64 tiles with enough K splits to put ~1000-1500
    // workgroups in flight.  (A per-shape cost model over rounds x depth was tried and lost on the mid-size shapes.)
    if (t128 >= 1024) big = true;
    else if (t128 >= 384) { big = true; if (!g.relu && nslab >= 32) g.splitk = 2; }
    else if (!g.relu && t128 >= 32 && nslab >= 256 && 512 / t128 >= 2) {
        big = true;
        long sk = 512 / t128;
        if (sk > nslab / 16) sk = nslab / 16;
        g.splitk = (int)sk;
    }
    else if (!g.relu && t64 >= 4) {
        const long target = (AKC && BKC) ? 768 : 1536;
        long sk = target / t64;
        if (sk > nslab / 16) sk = nslab / 16;      // keep K >= 512 per split
        if (sk >= 2) g.splitk = (int)sk;
    }
    // tuning hook for tools/ubench/gemm_bench.py: XG_GEMM_FORCE="<tile>,<splitk>"
    static const char* force = getenv("XG_GEMM_FORCE");
    if (force) {
        int t = 0, sk = 1;
        if (sscanf(force, "%d,%d", &t, &sk) == 2) { big = t == 128; g.splitk = g.relu ? 1 : (sk < 1 ? 1 : sk); }
    }
    if (big) return vec ? launch<128, 128, AKC, BKC, true>(st, g) : launch<128, 128, AKC, BKC, false>(st, g);
    return vec ? launch<64, 64, AKC, BKC, true>(st, g) : launch<64, 64, AKC, BKC, false>(st, g);
}

}  // namespace

int xgk_gemm(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
             const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate) {
    if (M <= 0 || N <= 0) return XG_OK;
    if (K < 0 || !A || !B || !C) return XG_EINVAL;
    // large products may run on the bf16 matrix cores (split-bf16 or plain bf16); skinny / tiny ones stay fp32
    if ((mode == 1 || mode == 3) && M >= 256 && N >= 64 && K >= 64)
        return xgk_gemm_bf16(st, mode, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate);
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, relu ? 1 : 0, accumulate ? 1 : 0, 1, 0};
    const bool akc = !transA;   // A (M,K) row-major -> k contiguous
    const bool bkc = transB;    // B (N,K) row-major -> k contiguous
    {   // offset-based loads: the largest byte offset inside either operand must fit 31 bits, and the m/n-contiguous
        // clamp needs at least 4 rows
        const size_t ea = akc ? (size_t)M * lda : (size_t)K * lda, eb = bkc ? (size_t)N * ldb : (size_t)K * ldb;
        g.fast = (ea * 4 < (1u << 31)) && (eb * 4 < (1u << 31)) && M >= 4 && N >= 4;
    }
    // 16-byte vector loads need aligned bases, ld % 4 == 0 and the vectorised extent % 4 == 0
    bool vec = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && (lda % 4 == 0) && (ldb % 4 == 0);
    vec = vec && ((akc ? K : M) % 4 == 0) && ((bkc ? K : N) % 4 == 0);
    if (akc && bkc) return dispatch<true, true>(st, g, vec);
    if (akc && !bkc) return dispatch<true, false>(st, g, vec);
    if (!akc && !bkc) return dispatch<false, false>(st, g, vec);
    return dispatch<false, true>(st, g, vec);
}

extern "C" int xg_gemm_mode(void* stream, int mode, int transA, int transB, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, const float* bias, int relu, int accumulate) {
    if (mode != 0 && mode != 1 && mode != 3) return XG_EINVAL;
    if (mode == 0 || M <= 0 || N <= 0)
        return xgk_gemm((hipStream_t)stream, 0, transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu != 0, accumulate != 0);
    if (K < 0 || !A || !B || !C) return XG_EINVAL;
    return xgk_gemm_bf16((hipStream_t)stream, mode, transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias,
                         relu != 0, accumulate != 0);
}

extern "C" int xg_gemm(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                       const float* B, int ldb, float* C, int ldc, const float* bias, int relu, int accumulate) {
    return xgk_gemm((hipStream_t)stream, 0, transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias,
                    relu != 0, accumulate != 0);
}
