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
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BKS = 32;          // slab depth
constexpr int LDK = BKS + 4;     // row stride of a k-contiguous LDS image
constexpr int MCP = 4;           // padding of a k-row of an m-contiguous LDS image

template <int ROWS, bool KC>
struct TileGeom {
    static constexpr int lds_floats = KC ? ROWS * LDK : BKS * (ROWS + MCP);
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
            *reinterpret_cast<f32x4*>(lds + k * (ROWS + MCP) + r) = regs[i];
        }
    }
}

// Fragment for MFMA: 4 k-values (k = kb*8 + 4*half + j) of row `row`.
template <int ROWS, bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ lds, int row, int kb, int half) {
    if (KC) {
        return *reinterpret_cast<const f32x4*>(lds + row * LDK + kb * 8 + half * 4);
    } else {
        const float* p = lds + (kb * 8 + half * 4) * (ROWS + MCP) + row;
        f32x4 v;
        v[0] = p[0];
        v[1] = p[ROWS + MCP];
        v[2] = p[2 * (ROWS + MCP)];
        v[3] = p[3 * (ROWS + MCP)];
        return v;
    }
}

// sched_group_barrier pipeline of one block: M MFMAs with D LDS instructions and V global loads spread evenly between them
// (the block's first MFMAs depend on nothing issued in the block, so an MFMA group always comes first)
template <int M, int D, int V>
__device__ __forceinline__ void w1_interleave() {
    constexpr int MEM = D + V;
    if constexpr (M == 0) {
        if constexpr (V > 0) __builtin_amdgcn_sched_group_barrier(0x020, V, 0);
        if constexpr (D > 0) __builtin_amdgcn_sched_group_barrier(0x080, D, 0);
    } else if constexpr (MEM == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
    } else if constexpr (MEM >= M) {
        constexpr int K = (MEM + M - 1) / M, v = V < K ? V : K, d = K - v;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x020, v, 0);
        if constexpr (d > 0) __builtin_amdgcn_sched_group_barrier(0x080, d, 0);
        w1_interleave<M - 1, D - d, V - v>();
    } else {
        constexpr int mm = M / MEM;
        __builtin_amdgcn_sched_group_barrier(0x008, mm, 0);
        if constexpr (V > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); w1_interleave<M - mm, D, V - 1>(); }
        else { __builtin_amdgcn_sched_group_barrier(0x080, 1, 0); w1_interleave<M - mm, D - 1, V>(); }
    }
}

// One 32-deep slab of MFMAs out of the LDS images: per 8-deep block [fragment reads][4 MT NT MFMAs], scheduled by the compiler
// ([reads][s_waitcnt][MFMAs]: a wave's matrix pipe idles for one LDS round trip per block, which the other waves of the SIMD
// cover).  (Fragments requested one block ahead with a pinned order were measured in round 3 and are gone: no gain on
// k-contiguous operands, a loss on m-contiguous ones -- docs/EXPERIMENTS.md.)
template <int BM, int BN, bool AKC, bool BKC, int MT, int NT>
__device__ __forceinline__ void slab_mfma_pipelined(const float* __restrict__ as, const float* __restrict__ bs, int arow, int brow,
                                                    int half, f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int kb = 0; kb < BKS / 8; ++kb) {
        f32x4 fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = read_frag<BM, AKC>(as, arow + i * 32, kb, half);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = read_frag<BN, BKC>(bs, brow + j * 32, kb, half);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
    }
}

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, relu, accumulate;
    int splitk;   // > 1: the K slabs are divided over `splitk` workgroups per tile; partial tiles are added
                  // into C with hardware fp32 atomics (C pre-zeroed by the launcher unless accumulating)
    int fast;     // vector kernels: offset-based unpredicated loads for the full slabs (all byte offsets < 2^31)
    int gm;       // tile rows per group of the tile order (xgk_group_rows)
    int bg;       // XGK_GEMM_BG: the product runs BESIDE a latency-bound launch chain on another stream (see launch_pk)
    // weight-gradient layout only (A m-contiguous = dY^T): csum[q][m] += sum_k A(k, m) for up to three accumulators -- the bias
    // gradient(s) that belong to the same dY, formed from the A slabs the tn == 0 tiles stream anyway (no second pass over dY)
    float* csum[3];
    int alone;    // XGK_GEMM_ALONE (xg_kernels.h)
};

// sum of this thread's A-slab registers (m-contiguous operand: 4 consecutive rows m per register, the same 4 for every register
// of the thread) and, at the end of a reduction range, the workgroup's total per row -> atomics.  `red4` = 256 float4 of LDS.
template <int BM>
__device__ __forceinline__ void csum_flush(const f32x4& cs, f32x4* red4, int m0, int M, float* const (&out)[3]) {
    constexpr int RG = BM / 4;                   // row groups; thread t holds rows 4 (t % RG) .. + 3
    const int t = threadIdx.x;
    if (t < RG) {
        f32x4 v = cs;
#pragma unroll
        for (int j = 1; j < 256 / RG; ++j) v += red4[t + j * RG];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + 4 * t + q;
            if (row < M) {
#pragma unroll
                for (int o = 0; o < 3; ++o) if (out[o]) unsafeAtomicAdd(out[o] + row, v[q]);
            }
        }
    }
}

template <int BM, int BN, bool AKC, bool BKC, bool VEC>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
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
    int tm, tn;
    {   // grouped order: gm tile rows x ~64/gm tile columns are live together on an XCD (gm + 64/gm operand panels in its L2, not 1 + 64)
        const int per = g.gm * ntn, grp = bid / per, in = bid - grp * per;
        const int first = grp * g.gm, gsz = min(ntm - first, g.gm);
        tn = in / gsz;
        tm = first + (in - tn * gsz);
    }
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

    constexpr int NVA = TileGeom<BM, AKC>::nvec, NVB = TileGeom<BN, BKC>::nvec;
    f32x4 ra[NVA], rb[NVB];
    const int nslab_all = (g.K + BKS - 1) / BKS;
    const int s_begin = (int)(((long)ks * nslab_all) / g.splitk), s_end = (int)(((long)(ks + 1) * nslab_all) / g.splitk);
    const int nslab = s_end;
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
    const bool do_cs = !AKC && VEC && g.csum[0] != nullptr && tn == 0;      // (launcher: csum only with the vector kernels)
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    if (do_cs) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) cs += ra[i];
    }
    store_tile<BM, AKC>(smem, ra);
    store_tile<BN, BKC>(smem + A_FL, rb);
    __syncthreads();

    for (int s = s_begin; s < nslab; ++s) {
        const int cur = (s - s_begin) & 1;
        if (s + 1 < nslab) {
            if (fast && s + 1 < nfull) {
                load_fast<NVA>(reinterpret_cast<const char*>(g.A) + (size_t)(s + 1) * stepA, offA, ra);
                load_fast<NVB>(reinterpret_cast<const char*>(g.B) + (size_t)(s + 1) * stepB, offB, rb);
            } else {
                load_tile<BM, AKC, VEC>(g.A, g.lda, m0, (s + 1) * BKS, g.M, g.K, ra);
                load_tile<BN, BKC, VEC>(g.B, g.ldb, n0, (s + 1) * BKS, g.N, g.K, rb);
            }
        }
        const float* as = smem + cur * STAGE;
        const float* bs = as + A_FL;
        slab_mfma_pipelined<BM, BN, AKC, BKC, MT, NT>(as, bs, wm * WM + l31, wn * WN + l31, half, acc);
        if (s + 1 < nslab) {
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < NVA; ++i) cs += ra[i];
            }
            store_tile<BM, AKC>(smem + (cur ^ 1) * STAGE, ra);
            store_tile<BN, BKC>(smem + (cur ^ 1) * STAGE + A_FL, rb);
        }
        __syncthreads();
    }
    if (do_cs) {                                 // (the staging buffers are free: every wave is past its last slab)
        f32x4* red4 = reinterpret_cast<f32x4*>(smem);
        red4[threadIdx.x] = cs;
        __syncthreads();
        csum_flush<BM>(cs, red4, m0, g.M, g.csum);
    }

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

// ---- one workgroup per CU, deep slabs ("w1") -----------------------------------------------------------------------
// For products whose output divides into roughly ONE round of tiles for the 256 CUs (the mid-size shapes: 3328 x 512, 2048 x 512,
// 3328 x 1536 ...).  The kernels above need two or more workgroups per CU to cover each other's stalls -- per slab a wave waits
// for its global loads (issued one 32-deep slab = 0.5-2 us earlier), stores, meets the barrier and waits for the first
// fragments -- and with one wave per SIMD they run at 55-65 % of the matrix rate (wgrad TN 2048 x 512 x 2688 as 256 tiles of 64 x 64:
// 83.6 us, the vendor library's 256-tile kernel: 52.6).  Here ONE wave per SIMD has to keep its matrix pipe busy by itself:
//   * slabs are BK = 128 / 64 / 32 deep by tile area, 4096-8192 MFMA cycles each: the next slab's global loads (issued at the top
//     of the slab, one register set) have a whole slab to land, and there is one barrier per slab instead of four;
//   * inside a slab the operand fragments of 8-deep block kb + 1 are requested before the MFMAs of block kb (two fragment sets,
//     order pinned), and the LDS stores of the next slab are spread over the last quarter of the blocks (its stage is idle);
//   * what is left exposed per slab is the barrier and the first fragment round trip.
// k-contiguous images have a row stride of BK + 4 floats (an odd number of 16-byte slots: ds_read_b128 fragments are conflict-free).
template <int ROWS, int BK, bool KC>
struct TileW {
    static constexpr int ld = KC ? BK + 4 : ROWS + MCP;
    static constexpr int lds_floats = KC ? ROWS * (BK + 4) : BK * (ROWS + MCP);
    static constexpr int nvec = ROWS * BK / 4 / 256;        // float4 per thread per slab
    static_assert(ROWS * BK % 1024 == 0, "tile must divide over 256 threads");
};
template <int ROWS, int BK, bool KC>
__device__ __forceinline__ void w1_offsets(int ld, int r0, int nrows, uint32_t (&off)[TileW<ROWS, BK, KC>::nvec]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TileW<ROWS, BK, KC>::nvec; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            const int r = f / (BK / 4), k = (f % (BK / 4)) << 2;
            off[i] = (uint32_t)(((size_t)min(r0 + r, nrows - 1) * ld + k) * sizeof(float));
        } else {
            const int k = f / (ROWS / 4), r = (f % (ROWS / 4)) << 2;
            off[i] = (uint32_t)(((size_t)k * ld + min(r0 + r, nrows - 4)) * sizeof(float));
        }
    }
}
template <int ROWS, int BK, bool KC>
__device__ __forceinline__ void w1_store_one(float* __restrict__ lds, int i, const f32x4& v) {
    const int f = threadIdx.x + 256 * i;
    if (KC) {
        const int r = f / (BK / 4), k = (f % (BK / 4)) << 2;
        *reinterpret_cast<f32x4*>(lds + r * (BK + 4) + k) = v;
    } else {
        const int k = f / (ROWS / 4), r = (f % (ROWS / 4)) << 2;
        *reinterpret_cast<f32x4*>(lds + k * (ROWS + MCP) + r) = v;
    }
}
template <int ROWS, int BK, bool KC>
__device__ __forceinline__ f32x4 w1_frag(const float* __restrict__ lds, int row, int kb, int half) {
    if (KC) return *reinterpret_cast<const f32x4*>(lds + row * (BK + 4) + kb * 8 + half * 4);
    const float* p = lds + (kb * 8 + half * 4) * (ROWS + MCP) + row;
    f32x4 v;
    v[0] = p[0]; v[1] = p[ROWS + MCP]; v[2] = p[2 * (ROWS + MCP)]; v[3] = p[3 * (ROWS + MCP)];
    return v;
}

#ifdef W1_TRACE
__device__ long long w1_trace_buf[256 * 8];
#define W1_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256) { w1_trace_buf[blockIdx.x * 8 + (i)] = wall_clock64(); \
                                                                       if ((i) == 1) w1_trace_buf[blockIdx.x * 8 + 4] = clock64(); \
                                                                       if ((i) == 2) w1_trace_buf[blockIdx.x * 8 + 5] = clock64(); } } while (0)
#else
#define W1_STAMP(i) do { } while (0)
#endif
template <int BM, int BN, int BK, bool AKC, bool BKC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_w1_kernel(GemmArgs g) {
    W1_STAMP(0);
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32, NB = BK / 8;
    constexpr int A_FL = TileW<BM, BK, AKC>::lds_floats, B_FL = TileW<BN, BK, BKC>::lds_floats, STAGE = A_FL + B_FL;
    constexpr int NVA = TileW<BM, BK, AKC>::nvec, NVB = TileW<BN, BK, BKC>::nvec, NV = NVA + NVB;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {   // XCD-aware grouped tile order (as gemm_kernel)
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    {
        const int per = g.gm * ntn, grp = bid / per, in = bid - grp * per;
        const int first = grp * g.gm, gsz = min(ntm - first, g.gm);
        tn = in / gsz;
        tm = first + (in - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int arow = wm * WM + l31, brow = wn * WN + l31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint32_t offA[NVA], offB[NVB];
    w1_offsets<BM, BK, AKC>(g.lda, m0, g.M, offA);
    w1_offsets<BN, BK, BKC>(g.ldb, n0, g.N, offB);
    const size_t stepA = (AKC ? (size_t)BK : (size_t)BK * g.lda) * sizeof(float);
    const size_t stepB = (BKC ? (size_t)BK : (size_t)BK * g.ldb) * sizeof(float);
    const char* pa = reinterpret_cast<const char*>(g.A);
    const char* pb = reinterpret_cast<const char*>(g.B);
    const int ns = g.K / BK;                     // (launcher: K % BK == 0)
    // TWO register sets of global loads: slab x travels in set x & 1, requested at the top of slab x - 2 and stored into LDS at
    // the end of slab x - 1 -- 1.75 slabs for the loads to land (measured: 1.3-2.6 us under load, i.e. more than one slab)
    f32x4 ra[2][NVA], rb[2][NVB];
    float bv[NT];                                // bias of this lane's columns: requested now, used in the epilogue
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
        bv[j] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    }
    load_fast<NVA>(pa, offA, ra[0]);
    load_fast<NVB>(pb, offB, rb[0]);
    if (ns > 1) {
        load_fast<NVA>(pa + stepA, offA, ra[1]);
        load_fast<NVB>(pb + stepB, offB, rb[1]);
    }
#pragma unroll
    for (int i = 0; i < NVA; ++i) w1_store_one<BM, BK, AKC>(smem, i, ra[0][i]);
#pragma unroll
    for (int i = 0; i < NVB; ++i) w1_store_one<BN, BK, BKC>(smem + A_FL, i, rb[0][i]);
    __syncthreads();
    W1_STAMP(1);

    // One slab as straight-line code (SET = s & 1; LOAD: slab s + 2 exists and is requested here; STORE: slab s + 1 exists and goes
    // to LDS here).  Per 8-deep block the memory instructions -- this block's share of the global loads (first blocks), the
    // fragment reads of block kb + 1, this block's share of the LDS stores (last blocks) -- are INTERLEAVED with the block's MFMAs
    // (sched_group_barrier pipeline): a wave issues in order, and a cluster of 12-20 LDS / VMEM instructions between two MFMA
    // groups keeps the matrix pipe waiting while the four waves' requests queue up at the CU's one LDS pipe and one texture
    // addresser (measured: 300-370 cycles per block, 76-105 cycles per MFMA instead of 64, whatever else the schedule did).
    // A pipeline STAGE is BD 8-deep blocks (16 MFMAs or more: the fragments a stage needs are requested one stage ahead, and an LDS
    // round trip is 130-250 cycles -- with 4-MFMA stages the MFMAs caught up with their own operands at every block)
    constexpr int BD = MT * NT <= 2 ? (MT * NT == 1 ? 4 : 2) : 1;
    constexpr int NS = NB / BD;                               // stages per slab
    static_assert(NB % BD == 0 && NS >= 2, "");
    constexpr int WS = NS >= 8 ? NS / 4 : (NS >= 4 ? 2 : 1);  // stages over which a slab's LDS stores are spread (the last ones)
    constexpr int NLS = NS - WS;                              // stages that carry global loads (the first ones)
    constexpr int PERW = (NV + WS - 1) / WS, PERL = (NV + NLS - 1) / NLS;
    constexpr int FRA = AKC ? 1 : 2, FRB = BKC ? 1 : 2;       // LDS instructions per fragment (b128, or two ds_read2_b32)
    auto slab = [&](int s, auto set_tag, auto load_tag, auto store_tag) {
        constexpr int SET = decltype(set_tag)::value;          // == s & 1
        constexpr bool LOAD = decltype(load_tag)::value, STORE = decltype(store_tag)::value;
        const float* as = smem + SET * STAGE;
        const float* bs = as + A_FL;
        float* was = smem + (SET ^ 1) * STAGE;
        float* wbs = was + A_FL;
        const char* la = pa + (size_t)(s + 2) * stepA;
        const char* lb = pb + (size_t)(s + 2) * stepB;
        f32x4 fa[2][BD][MT], fb[2][BD][NT];
#pragma unroll
        for (int d = 0; d < BD; ++d) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[0][d][i] = w1_frag<BM, BK, AKC>(as, arow + i * 32, d, half);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[0][d][j] = w1_frag<BN, BK, BKC>(bs, brow + j * 32, d, half);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int c = st & 1, n = c ^ 1;
            if (LOAD && st < NLS) {               // this set's slab is in LDS: reuse the registers for slab s + 2
#pragma unroll
                for (int q = 0; q < PERL; ++q) {
                    const int i = st * PERL + q;
                    if (i < NVA) ra[SET][i] = *reinterpret_cast<const f32x4*>(la + offA[i]);
                    else if (i < NV) rb[SET][i - NVA] = *reinterpret_cast<const f32x4*>(lb + offB[i - NVA]);
                }
            }
            if (st + 1 < NS) {
#pragma unroll
                for (int d = 0; d < BD; ++d) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa[n][d][i] = w1_frag<BM, BK, AKC>(as, arow + i * 32, (st + 1) * BD + d, half);
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb[n][d][j] = w1_frag<BN, BK, BKC>(bs, brow + j * 32, (st + 1) * BD + d, half);
                }
            }
            if (STORE && st >= NLS) {             // the other set (slab s + 1, requested 1.75 slabs ago) goes to LDS
                const int w = st - NLS;
#pragma unroll
                for (int q = 0; q < PERW; ++q) {
                    const int i = w * PERW + q;
                    if (i < NVA) w1_store_one<BM, BK, AKC>(was, i, ra[SET ^ 1][i]);
                    else if (i < NV) w1_store_one<BN, BK, BKC>(wbs, i - NVA, rb[SET ^ 1][i - NVA]);
                }
            }
#pragma unroll
            for (int d = 0; d < BD; ++d)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][d][i][kk], fb[c][d][j][kk], acc[i][j], 0, 0, 0);
            // the pipeline of this stage (counts are upper bounds: a group that finds fewer instructions is simply shorter)
            constexpr int NM = 4 * BD * MT * NT, NFR = BD * (MT * FRA + NT * FRB);
            if (st < NLS) {
                if (st + 1 < NS) w1_interleave<NM, NFR, LOAD ? PERL : 0>();
                else w1_interleave<NM, 0, LOAD ? PERL : 0>();
            } else {
                if (st + 1 < NS) w1_interleave<NM, NFR + (STORE ? PERW : 0), 0>();
                else w1_interleave<NM, STORE ? PERW : 0, 0>();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    {
        using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
        using Y = std::true_type; using N = std::false_type;
        int s = 0;
        for (; s + 3 < ns; s += 2) {             // both slabs of the pair have a slab two ahead
            slab(s, T0{}, Y{}, Y{});
            slab(s + 1, T1{}, Y{}, Y{});
        }
        // the last two or three slabs (s is even here)
        if (s + 2 < ns) {                        // three left: s (loads s + 2), s + 1, s + 2
            slab(s, T0{}, Y{}, Y{});
            slab(s + 1, T1{}, N{}, Y{});
            slab(s + 2, T0{}, N{}, N{});
        } else if (s + 1 < ns) {                 // two left
            slab(s, T0{}, N{}, Y{});
            slab(s + 1, T1{}, N{}, N{});
        } else if (s < ns) {
            slab(s, T0{}, N{}, N{});
        }
    }
    W1_STAMP(2);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if (col >= g.N) continue;
            float old[16];
            if (g.accumulate) {                  // all 16 reads in flight before the first add
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    old[r] = row < g.M ? g.C[(size_t)row * g.ldc + col] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.M) {
                    float v = acc[i][j][r] + bv[j];
                    if (g.accumulate) v += old[r];
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    W1_STAMP(3);
}

// ---- persistent form of the 128x128 kernel ("pk") ----------------------------------------------------------------
// The one-tile-per-workgroup kernel above spends 20-25 % of a tile's lifetime outside its K loop when the reduction is
// short (logits: K = 512 = 16 slabs): the first slab's load latency, the 64 KB result tile and the dispatch of the next
// workgroup are paid once per tile by BOTH workgroups of a CU at the same moment (they start in phase and stay in phase),
// and the last round of tiles is partly empty (3297 tiles on 512 slots = 6.44 rounds, paid as 7).  Here the grid is the
// 512 resident workgroups (2 per CU, LDS-bound) and each one walks a list of work items (tile, slab range):
//   * whole tiles first, floor(T / G) per workgroup, in an XCD-contiguous order whose 64 concurrent tiles per XCD form an
//     8 x 8 block (8 A panels + 8 B panels live in that XCD's L2 instead of 1 + 64);
//   * then the remaining T mod G tiles as slab units shared evenly by all workgroups (stream-K): a workgroup that owns
//     only part of a tile's reduction adds its partial tile with fp32 atomics (the launcher zeroes those tiles).
// The slab pipeline runs ACROSS items: during the last slab of an item the first slab of the next one is already in
// flight, the result stores (fire-and-forget, no wait) drain under the next item's MFMAs.
struct PkArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, relu, accumulate;
    int ntm, ntn, nslab;
    int rounds;      // whole tiles per workgroup; tiles [0, rounds * G) in swizzled order
    int tail_wgs;    // workgroup positions [0, tail_wgs) share the slab units of tiles [rounds * G, T)
    int gm;          // tile rows per group of the swizzled order
    float* csum[3];  // see GemmArgs::csum
};

struct PkItem { int tile, s0, s1; };
#ifdef PK_TRACE
__device__ long long pk_trace_buf[512 * 40];
#define PK_STAMP(i) do { if (threadIdx.x == 0 && (i) < 38) pk_trace_buf[blockIdx.x * 40 + (i)] = wall_clock64(); \
                         if (threadIdx.x == 0 && (i) == 0) pk_trace_buf[blockIdx.x * 40 + 38] = clock64(); \
                         if (threadIdx.x == 0 && (i) >= 5) pk_trace_buf[blockIdx.x * 40 + 39] = clock64(); } while (0)
#else
#define PK_STAMP(i) do { } while (0)
#endif
#define PK_BARRIER() __syncthreads()

__device__ __forceinline__ void pk_tile_coords(int t, int ntm, int ntn, int GM, int& tm, int& tn) {
    const int per = GM * ntn, grp = t / per, in = t - grp * per;
    const int first = grp * GM, gsz = min(ntm - first, GM);
    tn = in / gsz;
    tm = first + (in - tn * gsz);
}

// rows x cols sub-matrix (row pitch ld, base 16-byte aligned, ld % 4 == 0) := 0; one workgroup per 1024 columns of a row
__global__ void __launch_bounds__(256) zero2d_kernel(float* __restrict__ p, int ld, int cols) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    float* q = p + (size_t)blockIdx.y * ld + c;
    if (c + 4 <= cols) *reinterpret_cast<f32x4*>(q) = f32x4{0.f, 0.f, 0.f, 0.f};
    else for (int j = 0; c + j < cols; ++j) q[j] = 0.f;
}

template <int BM, int BN, bool AKC, bool BKC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_pk_kernel(PkArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
    constexpr int A_FL = TileGeom<BM, AKC>::lds_floats, B_FL = TileGeom<BN, BKC>::lds_floats, STAGE = A_FL + B_FL;
    constexpr int NVA = TileGeom<BM, AKC>::nvec, NVB = TileGeom<BN, BKC>::nvec;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // position of this workgroup: XCD x owns the contiguous positions [px, px + cnt)
    const int G = gridDim.x;
    const int xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int px = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int cnt = xcd < xr ? xq + 1 : xq;
    const int pos = px + idx;
    const int T = g.ntm * g.ntn, dp_tiles = g.rounds * G;
    // work list state (wave-uniform)
    int round = 0;
    long u = 0, u1 = 0;
    if (pos < g.tail_wgs) {
        const long U = (long)(T - dp_tiles) * g.nslab;
        u = (U * pos) / g.tail_wgs;
        u1 = (U * (pos + 1)) / g.tail_wgs;
    }
    auto next_item = [&](PkItem& it) -> bool {
        if (round < g.rounds) {
            it.tile = px * g.rounds + round * cnt + idx;
            it.s0 = 0; it.s1 = g.nslab;
            ++round;
            return true;
        }
        if (u < u1) {
            const int tl = (int)(u / g.nslab);
            it.tile = dp_tiles + tl;
            it.s0 = (int)(u - (long)tl * g.nslab);
            const long left = u1 - u;
            it.s1 = (int)min((long)g.nslab, (long)it.s0 + left);
            u += it.s1 - it.s0;
            return true;
        }
        return false;
    };

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

    f32x4 ra[NVA], rb[NVB];
    uint32_t offA[NVA], offB[NVB];
    const size_t stepA = (AKC ? (size_t)BKS : (size_t)BKS * g.lda) * sizeof(float);
    const size_t stepB = (BKC ? (size_t)BKS : (size_t)BKS * g.ldb) * sizeof(float);
    const int nfull = g.K / BKS;

    PkItem cur;
    if (!next_item(cur)) return;
    int m0, n0;
    const bool want_cs = !AKC && g.csum[0] != nullptr;
    bool cs_on, csn_on = false;                  // this item's tile is in tile column 0: it also sums its A slabs per row
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, csn = {0.f, 0.f, 0.f, 0.f};
    f32x4* red4 = reinterpret_cast<f32x4*>(smem + 2 * STAGE);       // 4 KB behind the two stages (launcher adds it)
    {
        int tm, tn;
        pk_tile_coords(cur.tile, g.ntm, g.ntn, g.gm, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        cs_on = want_cs && tn == 0;
    }
    tile_offsets<BM, AKC>(g.lda, m0, g.M, offA);
    tile_offsets<BN, BKC>(g.ldb, n0, g.N, offB);
    // bias of the item's columns, fetched when the item starts (the epilogue must not wait on a load)
    float bv[NT], bvn[NT];
    auto load_bias = [&](const PkItem& it, int nn0, float (&b)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = nn0 + wn * WN + j * 32 + l31;
            b[j] = (g.bias && it.s0 == 0 && col < g.N) ? g.bias[col] : 0.f;
        }
    };
    load_bias(cur, n0, bv);
#pragma unroll
    for (int j = 0; j < NT; ++j) bvn[j] = 0.f;
    if (cur.s0 < nfull) {
        load_fast<NVA>(reinterpret_cast<const char*>(g.A) + (size_t)cur.s0 * stepA, offA, ra);
        load_fast<NVB>(reinterpret_cast<const char*>(g.B) + (size_t)cur.s0 * stepB, offB, rb);
    } else {
        load_tile<BM, AKC, true>(g.A, g.lda, m0, cur.s0 * BKS, g.M, g.K, ra);
        load_tile<BN, BKC, true>(g.B, g.ldb, n0, cur.s0 * BKS, g.N, g.K, rb);
    }
    PK_STAMP(0);
    if (cs_on) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) cs += ra[i];
    }
    store_tile<BM, AKC>(smem, ra);
    store_tile<BN, BKC>(smem + A_FL, rb);
    __syncthreads();
    PK_STAMP(1);
    int item_no = 0;
    (void)item_no;                               // (read by the -DPK_TRACE stamps only)

    auto slab_mfma = [&](int buf) {
        const float* as = smem + buf * STAGE;
        const float* bs = as + A_FL;
        slab_mfma_pipelined<BM, BN, AKC, BKC, MT, NT>(as, bs, wm * WM + l31, wn * WN + l31, half, acc);
    };

    int buf = 0;
    while (true) {                               // one item per trip; the accumulators live in AGPRs across its slab loop
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int s = cur.s0; s + 1 < cur.s1; ++s) {
            if (s + 1 < nfull) {
                load_fast<NVA>(reinterpret_cast<const char*>(g.A) + (size_t)(s + 1) * stepA, offA, ra);
                load_fast<NVB>(reinterpret_cast<const char*>(g.B) + (size_t)(s + 1) * stepB, offB, rb);
            } else {
                load_tile<BM, AKC, true>(g.A, g.lda, m0, (s + 1) * BKS, g.M, g.K, ra);
                load_tile<BN, BKC, true>(g.B, g.ldb, n0, (s + 1) * BKS, g.N, g.K, rb);
            }
            slab_mfma(buf);
            if (cs_on) {
#pragma unroll
                for (int i = 0; i < NVA; ++i) cs += ra[i];
            }
            store_tile<BM, AKC>(smem + (buf ^ 1) * STAGE, ra);
            store_tile<BN, BKC>(smem + (buf ^ 1) * STAGE + A_FL, rb);
            __syncthreads();
            buf ^= 1;
        }
        // last slab of the item: the first slab of the NEXT item is fetched under it
        PK_STAMP(2 + 4 * item_no);
        PkItem nx;
        const bool has = next_item(nx);
        int nm0 = m0, nn0 = n0;
        if (has) {
            int tm, tn;
            pk_tile_coords(nx.tile, g.ntm, g.ntn, g.gm, tm, tn);
            nm0 = tm * BM; nn0 = tn * BN;
            csn_on = want_cs && tn == 0;
            tile_offsets<BM, AKC>(g.lda, nm0, g.M, offA);
            tile_offsets<BN, BKC>(g.ldb, nn0, g.N, offB);
            load_bias(nx, nn0, bvn);
            if (nx.s0 < nfull) {
                load_fast<NVA>(reinterpret_cast<const char*>(g.A) + (size_t)nx.s0 * stepA, offA, ra);
                load_fast<NVB>(reinterpret_cast<const char*>(g.B) + (size_t)nx.s0 * stepB, offB, rb);
            } else {
                load_tile<BM, AKC, true>(g.A, g.lda, nm0, nx.s0 * BKS, g.M, g.K, ra);
                load_tile<BN, BKC, true>(g.B, g.ldb, nn0, nx.s0 * BKS, g.N, g.K, rb);
            }
        }
        slab_mfma(buf);
        if (has) {
            csn = f32x4{0.f, 0.f, 0.f, 0.f};
            if (csn_on) {
#pragma unroll
                for (int i = 0; i < NVA; ++i) csn += ra[i];
            }
            store_tile<BM, AKC>(smem + (buf ^ 1) * STAGE, ra);
            store_tile<BN, BKC>(smem + (buf ^ 1) * STAGE + A_FL, rb);
        }
        if (cs_on) red4[threadIdx.x] = cs;       // (read behind the barrier below)
        PK_STAMP(3 + 4 * item_no);
        {
            // result of item `cur`: whole tile -> stores (bias / ReLU; += C via atomics), part of a reduction -> atomics
            const bool part = cur.s0 != 0 || cur.s1 != g.nslab;
            const bool atom = part || g.accumulate;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = n0 + wn * WN + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const float v = acc[i][j][r] + bv[j];
                        if (row < g.M && col < g.N) {
                            float* dst = g.C + (size_t)row * g.ldc + col;
                            if (atom) unsafeAtomicAdd(dst, v);
                            else *dst = g.relu ? fmaxf(v, 0.f) : v;
                        }
                    }
                }
        }
        PK_STAMP(4 + 4 * item_no);
        PK_BARRIER();
        if (cs_on) csum_flush<BM>(cs, red4, m0, g.M, g.csum);
        PK_STAMP(5 + 4 * item_no);
        ++item_no;
        if (!has) break;
        cs = csn; cs_on = csn_on;
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = bvn[j];
        cur = nx; buf ^= 1; m0 = nm0; n0 = nn0;
    }
}

// launcher of the persistent kernel; returns XG_OK, or 1 when the shape should take the one-tile-per-workgroup kernels
template <int BM, int BN, bool AKC, bool BKC>
int launch_pk(hipStream_t st, const GemmArgs& a, long min_units_default) {
    static const int disabled = xg_diag_env("XG_GEMM_NO_PK") ? 1 : 0;
    if (disabled || !a.fast || a.M < BM || a.N < BN) return 1;
    PkArgs g{a.A, a.B, a.C, a.bias, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.relu, a.accumulate, 0, 0, 0, 0, 0, a.gm,
             {a.csum[0], a.csum[1], a.csum[2]}};
    g.ntm = xg_cdiv(a.M, BM); g.ntn = xg_cdiv(a.N, BN); g.nslab = xg_cdiv(a.K, BKS);
    const long T = (long)g.ntm * g.ntn;
    const long units = T * g.nslab;
    // measured (tools/ubench/gemm_bench.py): the persistent form wins 3-6 % when every workgroup has >= ~40 slabs of work
    // (vocabulary head: logits, dW_logit, dH) and loses 10-15 % to 64x64 tiles + split-K on the mid-size products
    static const long min_units_env = xg_diag_env("XG_PK_MIN") ? atol(xg_diag_env("XG_PK_MIN")) : -1;
    const long min_units = min_units_env >= 0 ? min_units_env : min_units_default;      // (XG_PK_MIN: tests force the kernel onto small shapes)
    if (units < 512L * min_units) return 1;
    // 2 workgroups per CU (73.7 KB of LDS each) -- or, for a background product, ONE per CU (LDS padded past half a CU so
    // that the dispatcher cannot pair them): 512 persistent workgroups own every register file and LDS for the whole
    // product (0.7 ms for dW_logit), and a recurrent chain on another stream then waits for leftovers -- its step took
    // 120 us instead of 53 beside dW_logit.  One workgroup per CU leaves 256 VGPRs per SIMD and 78 KB of LDS, exactly one
    // 8-wave (or two 4-wave) skinny workgroups, and costs the product itself ~10 %.
    static const int bg_off = xg_diag_env("XG_GEMM_NO_BG") ? 1 : 0;
    static const int bg_all = xg_diag_env("XG_GEMM_FORCE_BG") ? 1 : 0;     // tests: every product takes the background form
    const bool bg = (a.bg + bg_all > 0) && !bg_off;
    const int GMAX = bg ? 256 : 512;
    int G;
    const bool split = !a.relu && g.nslab >= 2;
    if (!split) {
        if (T < GMAX) return 1;
        // whole tiles only (a ReLU epilogue cannot be split): the fewest rounds, evened out over the workgroups; the ragged
        // last round goes through the tail logic as one whole tile per share
        G = (int)xg_cdiv64(T, xg_cdiv64(T, GMAX));
        g.rounds = (int)(T / G);
        g.tail_wgs = (int)(T - (long)g.rounds * G);
    } else {
        G = GMAX;
        g.rounds = (int)(T / G);
        const long tail = T - (long)g.rounds * G;
        const long U = tail * g.nslab;
        long tw = U / 4;                            // >= 4 slabs per share
        if (tw > G) tw = G;
        if (tail > 0 && tw < 1) tw = 1;
        g.tail_wgs = (int)tw;
    }
    // zero the tiles that receive partial sums (non-accumulating products only): every tail tile unless each share is a
    // whole tile.  The tail is the END of the swizzled order: inside the last group of tile rows it is a set of whole tile
    // columns plus at most one partial column -- clear the bounding rectangle (whole tiles inside it overwrite the zeros).
    const long dp = (long)g.rounds * G, tail = T - dp;
    const bool whole_shares = tail == g.tail_wgs;   // one tile per share (U / tail_wgs == nslab)
    if (tail > 0 && !whole_shares && !a.accumulate) {
        if (a.ldc % 4 != 0 || ((uintptr_t)a.C % 16) != 0) return 1;
        const int per = g.gm * g.ntn;
        const int grp = (int)(dp / per), first = grp * g.gm, gsz = (g.ntm - first) < g.gm ? (g.ntm - first) : g.gm;
        const int tn0 = (int)((dp - (long)grp * per) / gsz);
        const bool one_group = first + gsz >= g.ntm;              // tail confined to the last group
        const int r0 = first * BM, c0 = one_group ? tn0 * BN : 0;
        if (r0 < a.M && c0 < a.N) {      // (hipMemset2DAsync takes 23 us for these 15 MB; this kernel 5)
            const int rows = a.M - r0, cols = a.N - c0;
            hipLaunchKernelGGL(zero2d_kernel, dim3(xg_cdiv(cols, 1024), rows), dim3(256), 0, st, a.C + (size_t)r0 * a.ldc + c0, a.ldc, cols);
            XG_CHECK_LAUNCH();
        }
    }
    size_t lds = 2 * (TileGeom<BM, AKC>::lds_floats + TileGeom<BN, BKC>::lds_floats) * sizeof(float) + 4096;   // + csum scratch
    static std::atomic<unsigned> optin{0};
    XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&gemm_pk_kernel<BM, BN, AKC, BKC>), 84 * 1024));
    if (bg && lds < 82 * 1024) lds = 82 * 1024;
    hipLaunchKernelGGL((gemm_pk_kernel<BM, BN, AKC, BKC>), dim3(G), dim3(256), lds, st, g);
    XG_CHECK_LAUNCH();
    return XG_OK;
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

// ---- weight-gradient layout straight from memory ("td": transposed-direct) -------------------------------------------
// C[M,N] (+)= A^T B with A stored (K, M) and B stored (K, N), both row-major: the layout of every weight gradient dW = dY^T X.
// Here the operand layout of v_mfma_f32_16x16x4_f32 IS the memory layout: the instruction wants A[i = lane % 16][k = lane / 16]
// and B[k = lane / 16][j = lane % 16], i.e. for a fixed k one value per consecutive m (n) -- a row of the stored matrix.  A lane
// loads 16 bytes = 4 consecutive m of row k = 4 s + lane / 16 and uses the four values as the A operand of FOUR tiles whose rows
// interleave (tile i holds rows m0 + 4 a + i): one global_load_dwordx4 per operand per 4-deep step feeds 16 MFMAs of a 64 x 64
// wave tile.  No LDS image, no barrier and no ds_read in the K loop; 16-lane groups read 256 contiguous bytes.  The four waves of
// a workgroup take interleaved k-steps of the SAME tile (16 consecutive k-rows per round) and add their tiles through LDS at the
// end; a ring of D steps of operands per wave is in flight.  Measured against the LDS-staged kernels on the weight-gradient
// shapes in tools/ubench/gemm_bench.py.
struct TdArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc, accumulate;
    int ntm, ntn, ksplit, nrounds;   // nrounds = K / 16
    float* csum[3];
};
constexpr int TD_LDW = 68;           // row pitch of a wave's partial tile in LDS (floats)

template <int D>
__global__ void __launch_bounds__(256) gemm_td_kernel(TdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // 4 x 64 x TD_LDW partial tiles + 4 x 64 column sums
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a16 = lane & 15, kq = lane >> 4;
    const int G = gridDim.x;
    const int xq = G >> 3, xr = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int pos = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;     // XCD x owns contiguous positions
    const int nitems = g.ntm * g.ntn * g.ksplit;
    const bool want_cs = g.csum[0] != nullptr;
    const size_t stepA = (size_t)16 * g.lda * sizeof(float), stepB = (size_t)16 * g.ldb * sizeof(float);
    float* red = smem + 4 * 64 * TD_LDW;
    for (int item = pos; item < nitems; item += G) {
        const int part = item % g.ksplit, tile = item / g.ksplit;
        const int tn = tile % g.ntn, tm = tile / g.ntn;
        const int m0 = tm * 64, n0 = tn * 64;
        const int i0 = (int)(((long)part * g.nrounds) / g.ksplit), i1 = (int)(((long)(part + 1) * g.nrounds) / g.ksplit);
        const bool cs_on = want_cs && tn == 0;
        // this lane's 16 bytes of a k-row: columns clamped into the matrix (whole groups of 4: M % 4 == N % 4 == 0)
        const uint32_t offA = (uint32_t)(((size_t)kq * g.lda + min(m0 + 4 * a16, g.M - 4)) * sizeof(float));
        const uint32_t offB = (uint32_t)(((size_t)kq * g.ldb + min(n0 + 4 * a16, g.N - 4)) * sizeof(float));
        const char* pa = reinterpret_cast<const char*>(g.A) + (size_t)4 * wave * g.lda * sizeof(float);
        const char* pb = reinterpret_cast<const char*>(g.B) + (size_t)4 * wave * g.ldb * sizeof(float);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
        // ring of D steps: slot u holds step it + u; a slot is requested again (step + D) right after it is consumed.  The order
        // [16 MFMAs][2 loads] per step is pinned (sched_barrier): left alone the scheduler sinks all of a ring's loads behind its
        // last MFMA, which makes the ring one step deep.  (launcher: every part has at least D steps)
        f32x4 ra[D], rb[D];
        const char* qa = pa + (size_t)i0 * stepA;
        const char* qb = pb + (size_t)i0 * stepB;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            ra[u] = *reinterpret_cast<const f32x4*>(qa + u * stepA + offA);
            rb[u] = *reinterpret_cast<const f32x4*>(qb + u * stepB + offB);
        }
        qa += D * stepA; qb += D * stepB;            // -> step it + D of slot 0
        auto step = [&](int u, bool reload) {         // (the reload goes into the registers the MFMAs have just read: no second name)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][i], rb[u][j], acc[i][j], 0, 0, 0);
            if (cs_on) cs += ra[u];
            __builtin_amdgcn_sched_barrier(0);
            if (reload) {
                ra[u] = *reinterpret_cast<const f32x4*>(qa + u * stepA + offA);
                rb[u] = *reinterpret_cast<const f32x4*>(qb + u * stepB + offB);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        int it = i0;
        for (; it + 2 * D <= i1; it += D) {          // every reload of this ring is a step of this part
#pragma unroll
            for (int u = 0; u < D; ++u) step(u, true);
            qa += D * stepA; qb += D * stepB;
        }
        {   // the last D .. 2 D - 1 steps
            const int left = i1 - it - D;           // steps behind the ring that is in registers: 0 .. D - 1
#pragma unroll
            for (int u = 0; u < D; ++u) step(u, u < left);
#pragma unroll
            for (int u = 0; u < D; ++u) if (u < left) step(u, false);
        }
        // ---- the four waves' tiles meet in LDS.  D layout of the instruction: lane l, register r = element (4 (l / 16) + r, l % 16)
        // of the 16 x 16 tile; tile (i, j) holds rows m0 + 4 a + i, columns n0 + 4 c + j: a lane's (i, r) is a row, its j = 0..3
        // four consecutive columns.
        float* mine = smem + wave * 64 * TD_LDW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * kq + 4 * r + i;
                *reinterpret_cast<f32x4*>(mine + row * TD_LDW + 4 * a16) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            }
        if (cs_on) {
            // the four k-quarters of the wave (lanes l, l + 16, l + 32, l + 48), then one row of 64 sums per wave
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = cs[q];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (kq == 0) red[wave * 64 + 4 * a16 + q] = v;
            }
        }
        __syncthreads();
        {
            const bool atom = g.ksplit > 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = lane + 64 * u;
                const int row = 16 * wave + (q >> 4), col = (q & 15) << 2;
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * TD_LDW + col);
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) v += *reinterpret_cast<const f32x4*>(smem + (w2 * 64 + row) * TD_LDW + col);
                const int gr = m0 + row, gc = n0 + col;
                if (gr < g.M && gc < g.N) {
                    float* dst = g.C + (size_t)gr * g.ldc + gc;
                    if (atom) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, v[e]);
                    } else if (g.accumulate) {
                        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(dst) + v;
                    } else {
                        *reinterpret_cast<f32x4*>(dst) = v;
                    }
                }
            }
            if (cs_on && threadIdx.x < 64) {
                const int m = m0 + (int)threadIdx.x;
                if (m < g.M) {
                    const float v = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
#pragma unroll
                    for (int o = 0; o < 3; ++o) if (g.csum[o]) unsafeAtomicAdd(g.csum[o] + m, v);
                }
            }
        }
        __syncthreads();                          // the partial tiles are free again
    }
}

// returns 1 when the product is not this kernel's kind
int launch_td(hipStream_t st, const GemmArgs& a) {
    static const int off = xg_diag_env("XG_GEMM_NO_TD") ? 1 : 0;
    if (off || !a.fast || a.relu || a.bias || a.M % 4 || a.N % 4 || a.K % 16 || a.ldc % 4 || ((uintptr_t)a.C % 16) != 0) return 1;
    if (a.M < 64 || a.N < 64 || a.K < 16 * 16) return 1;      // (>= 2 rings of 8 steps)
    TdArgs g{a.A, a.B, a.C, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.accumulate, 0, 0, 1, a.K / 16, {a.csum[0], a.csum[1], a.csum[2]}};
    g.ntm = xg_cdiv(a.M, 64); g.ntn = xg_cdiv(a.N, 64);
    const long tiles = (long)g.ntm * g.ntn;
    if (tiles < 32) return 1;
    // Production rule: the vocabulary head's weight gradient (a background product, or >= 1024 tiles).  The mid-size weight
    // gradients are 18 % faster here when they run ALONE (51.6 vs 61.0 us at 2048 x 512 x 2688) but the iteration is 1 % slower
    // with them (5.95-5.99 vs 5.88-5.91 ms, three runs each way, tools/ubench/td_iter*.sh): they run beside the recurrent launch
    // chains, which are the critical path, and a product that keeps 16 wide loads per wave in flight lengthens every round trip of
    // the chain next to it.  XG_TD_ALL=1 (diag library) sends every eligible product here.
    static const int td_all = xg_diag_env("XG_TD_ALL") ? atoi(xg_diag_env("XG_TD_ALL")) : 0;
    if (!td_all && !a.bg && !a.alone && tiles < 1024) return 1;
    static const int ks_env = xg_diag_env("XG_TD_KS") ? atoi(xg_diag_env("XG_TD_KS")) : 0;
    int ks = 1;
    if (tiles < 192) { ks = (int)(256 / tiles); if (ks > g.nrounds / 16) ks = g.nrounds / 16; if (ks < 1) ks = 1; }
    if (ks_env > 0) ks = ks_env;
    if (ks > g.nrounds / 8) ks = g.nrounds / 8;      // every part: at least one ring
    g.ksplit = ks;
    if (ks > 1 && !a.accumulate) {
        if (a.ldc == a.N) { if (hipMemsetAsync(a.C, 0, sizeof(float) * (size_t)a.M * a.N, st) != hipSuccess) return XG_EHIP; }
        else if (hipMemset2DAsync(a.C, sizeof(float) * a.ldc, 0, sizeof(float) * a.N, a.M, st) != hipSuccess) return XG_EHIP;
    }
    static const int bg_off = xg_diag_env("XG_GEMM_NO_BG") ? 1 : 0;
    const bool bg = a.bg && !bg_off;
    const long items = tiles * ks;
    const int GMAX = bg ? 256 : 512;
    const long rounds = xg_cdiv64(items, GMAX);
    const int G = (int)xg_cdiv64(items, rounds);      // the fewest rounds, evened out over the workgroups
    size_t lds = (4 * 64 * TD_LDW + 256) * sizeof(float);
    if (bg && lds < 82 * 1024) lds = 82 * 1024;      // one workgroup per CU beside a launch chain (see launch_pk)
    static const int depth_all = xg_diag_env("XG_TD_DEPTH") ? atoi(xg_diag_env("XG_TD_DEPTH")) : 8;
    const int depth_env = depth_all;          // (XG_TD_DEPTH: tests run the 4- and 12-step rings too)
    if (depth_env == 4) {
        static std::atomic<unsigned> optin4{0};
        XG_TRY(xg_lds_optin(optin4, reinterpret_cast<const void*>(&gemm_td_kernel<4>), 84 * 1024));
        hipLaunchKernelGGL((gemm_td_kernel<4>), dim3(G), dim3(256), lds, st, g);
    } else if (depth_env == 12 && g.nrounds / ks >= 12) {
        static std::atomic<unsigned> optin12{0};
        XG_TRY(xg_lds_optin(optin12, reinterpret_cast<const void*>(&gemm_td_kernel<12>), 84 * 1024));
        hipLaunchKernelGGL((gemm_td_kernel<12>), dim3(G), dim3(256), lds, st, g);
    } else {
        static std::atomic<unsigned> optin{0};
        XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&gemm_td_kernel<8>), 84 * 1024));
        hipLaunchKernelGGL((gemm_td_kernel<8>), dim3(G), dim3(256), lds, st, g);
    }
    XG_CHECK_LAUNCH();
    return XG_OK;
}

// launcher of the one-workgroup-per-CU kernel: picks the tile (wave grid 2 x 2 of 32 x 32 MFMA tiles) whose ONE round over the 256 CUs
// wastes the least, returns 1 when no candidate covers the chip well enough (or the shape is not its kind)
template <int BM, int BN, int BK, bool AKC, bool BKC>
int launch_w1(hipStream_t st, const GemmArgs& g) {
    const int ntm = xg_cdiv(g.M, BM), ntn = xg_cdiv(g.N, BN);
    constexpr size_t lds = 2 * (TileW<BM, BK, AKC>::lds_floats + TileW<BN, BK, BKC>::lds_floats) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS stages do not fit a CU");
    static std::atomic<unsigned> optin{0};
    XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&gemm_w1_kernel<BM, BN, BK, AKC, BKC>), (int)lds));
    hipLaunchKernelGGL((gemm_w1_kernel<BM, BN, BK, AKC, BKC>), dim3(ntm * ntn), dim3(256), lds, st, g);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
struct W1Tile { int bm, bn, bk; };
constexpr W1Tile W1_TILES[] = {{64, 64, 128}, {64, 128, 64}, {128, 64, 64}, {128, 128, 64}, {128, 192, 32}, {192, 128, 32},
                               {128, 256, 32}, {256, 128, 32}};     // (with the slab depth: 4096-8192 MFMA cycles per slab)
template <bool AKC, bool BKC>
int dispatch_w1(hipStream_t st, const GemmArgs& g) {
    static const int off = xg_diag_env("XG_GEMM_NO_W1") ? 1 : 0;
    if (off || !g.fast || g.csum[0] || g.bg || g.K < 256) return 1;
    static const char* force = xg_diag_env("XG_W1_TILE");           // diagnosis: "bm,bn"
    int fbm = 0, fbn = 0;
    if (force) sscanf(force, "%d,%d", &fbm, &fbn);
    int best = -1;
    double best_eff = 0.0;
    for (int c = 0; c < (int)(sizeof(W1_TILES) / sizeof(W1_TILES[0])); ++c) {
        const W1Tile& t = W1_TILES[c];
        if (g.K % t.bk || g.M < t.bm / 2 || g.N < t.bn / 2) continue;
        const long tiles = (long)xg_cdiv(g.M, t.bm) * xg_cdiv(g.N, t.bn);
        if (force) { if (t.bm == fbm && t.bn == fbn) { best = c; best_eff = 1.0; } continue; }
        if (tiles > 256) continue;
        // useful fraction of the chip's matrix time in that one round; larger tiles stream fewer operand bytes per flop
        const double eff = (double)g.M * g.N / (256.0 * t.bm * t.bn);
        const double score = eff * (t.bm * t.bn >= 128 * 128 ? 1.03 : (t.bm * t.bn >= 64 * 128 ? 1.0 : 0.97));
        if (score > best_eff) { best_eff = score; best = c; }
    }
    if (best < 0 || best_eff < 0.70) return 1;
    GemmArgs a = g;
    a.splitk = 1;
    a.gm = xgk_group_rows(g.K);
    switch (best) {
        case 0: return launch_w1<64, 64, 128, AKC, BKC>(st, a);
        case 1: return launch_w1<64, 128, 64, AKC, BKC>(st, a);
        case 2: return launch_w1<128, 64, 64, AKC, BKC>(st, a);
        case 3: return launch_w1<128, 128, 64, AKC, BKC>(st, a);
        case 4: return launch_w1<128, 192, 32, AKC, BKC>(st, a);
        case 5: return launch_w1<192, 128, 32, AKC, BKC>(st, a);
        case 6: return launch_w1<128, 256, 32, AKC, BKC>(st, a);
        default: return launch_w1<256, 128, 32, AKC, BKC>(st, a);
    }
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
    static const char* force = xg_diag_env("XG_GEMM_FORCE");
    if (force) {
        int t = 0, sk = 1;
        if (sscanf(force, "%d,%d", &t, &sk) == 2) { big = t == 128; g.splitk = g.relu ? 1 : (sk < 1 ? 1 : sk); }
    }
    g.gm = xgk_group_rows(g.K);                     // the persistent kernel walks whole reductions
    if constexpr (!AKC && !BKC) {
        if (vec && !force) {                        // the weight-gradient layout: operands straight from memory
            const int rct = launch_td(st, g);
            if (rct != 1) return rct;
        }
    }
    if (vec && !force) {
        const int rc1 = dispatch_w1<AKC, BKC>(st, g);
        if (rc1 != 1) return rc1;
    }
    if (vec && !force) {
        const int rc = launch_pk<128, 128, AKC, BKC>(st, g, 40);
        if (rc != 1) return rc;
    }
    g.gm = xgk_group_rows(g.K / g.splitk);
    if (big) return vec ? launch<128, 128, AKC, BKC, true>(st, g) : launch<128, 128, AKC, BKC, false>(st, g);
    return vec ? launch<64, 64, AKC, BKC, true>(st, g) : launch<64, 64, AKC, BKC, false>(st, g);
}

}  // namespace

int xgk_gemm(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
             const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate) {
    return xgk_gemm_cs(st, mode, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate, nullptr, nullptr, nullptr);
}

// the product plus, for the weight-gradient layout (transA: A = dY stored (K rows, M columns)), the column sums of A added into up
// to three accumulators: cs[m] += sum_k A(k, m) -- the bias gradient(s) of the same dY.  Fused into the fp32 vector kernels (the
// tiles of tile column 0 sum the A slabs they stream anyway); every other route runs the product and a separate column reduction.
int xgk_gemm_cs(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
                const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate, float* cs1, float* cs2,
                float* cs3) {
    if (M <= 0 || N <= 0) return XG_OK;
    if (K < 0 || !A || !B || !C) return XG_EINVAL;
    const bool want_cs = cs1 != nullptr;
    if (want_cs && !transA) return XG_EINVAL;
    const int bg = (mode & XGK_GEMM_BG) ? 1 : 0, alone = (mode & XGK_GEMM_ALONE) ? 1 : 0;
    mode &= ~(XGK_GEMM_BG | XGK_GEMM_ALONE);
    // large products may run on the bf16 matrix cores (split-bf16 or plain bf16); skinny / tiny ones stay fp32
    // Split-bf16 (mode 3) keeps the weight-gradient layout (transA) in exact fp32 (round 5): alone the fp32 kernels are level or
    // ahead there (2048 x 512 x 2688: 60 against 65 us, dW_logit 504 against 530), and inside the iteration the 21 split-bf16
    // weight gradients ran at 131 us each beside the chains (63 alone) and held the encoder's backward and the update back by
    // 0.4 ms.  The other layouts stay split (logits 453 against 539 us, dH 381 against 514, PRE 58 against 80).
    // (diag: XG_X3_FP32 = mask of the classes routed to exact fp32 in mode 3 -- 1 TN, 2 NT, 4 NN; default 1)
    static const int x3_fp32 = xg_diag_env("XG_X3_FP32") ? atoi(xg_diag_env("XG_X3_FP32")) : 1;
    const bool x3_exact = mode == 3 && (x3_fp32 & (transA ? 1 : (transB ? 2 : 4)));
    // (... and shallow reductions: the gradients of the initial-state projections are R x R x B products -- 1024 x 1024 x 128 at hidden
    // 1024 -- that took 62 us each on the bf16 tile kernel, four slabs per tile)
    constexpr int bf16_mink = 256;
    if ((mode == 1 || mode == 3) && !x3_exact && M >= 256 && N >= 64 && K >= bf16_mink) {
        return xgk_gemm_bf16(st, mode, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate, cs1, cs2, cs3);
    }
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, relu ? 1 : 0, accumulate ? 1 : 0, 1, 0, 1, bg, {nullptr, nullptr, nullptr}, alone};
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
    if (want_cs) {
        if (vec) { g.csum[0] = cs1; g.csum[1] = cs2; g.csum[2] = cs3; }
        else XG_TRY(xgk_colsum3(st, A, lda, K, M, cs1, cs2, cs3));          // (scalar-load kernels: separate pass)
    }
    if (akc && bkc) return dispatch<true, true>(st, g, vec);
    if (akc && !bkc) return dispatch<true, false>(st, g, vec);
    if (!akc && !bkc) return dispatch<false, false>(st, g, vec);
    return dispatch<false, true>(st, g, vec);
}

int xgk_gemm_x(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, const unsigned short* A16,
               int lda, const float* B, const unsigned short* B16, int ldb, float* C, int ldc, const float* bias, bool relu,
               bool accumulate, float* cs1, float* cs2, float* cs3) {
    if ((mode & ~(XGK_GEMM_BG | XGK_GEMM_ALONE)) == 1 && (A16 || B16) && M >= 256 && N >= 64 && K >= 256) {
        if (cs1 && !transA) return XG_EINVAL;
        return xgk_gemm_bf16x(st, 1 | (mode & XGK_GEMM_BG), transA, transB, M, N, K, A, A16, lda, B, B16, ldb, C, ldc, bias, relu, accumulate, cs1, cs2, cs3);
    }
    return xgk_gemm_cs(st, mode, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate, cs1, cs2, cs3);
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
