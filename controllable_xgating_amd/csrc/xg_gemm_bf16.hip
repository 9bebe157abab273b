// Split-bf16 GEMM on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16, ~2.5 PF dense = 16x the fp32 MFMA rate).
//
// fp32 operands are split ON THE FLY, while a tile is staged from registers into LDS, into NP bf16 planes
//     x = x0 + x1 + x2,   x0 = trunc_bf16(x), x1 = trunc_bf16(x - x0), x2 = round_bf16(x - x0 - x1)
// (8 significand bits per plane, 24 in total: the split of an fp32 value is exact), and the product is
// assembled from the plane products whose weight is above fp32 round-off:
//     NP = 3:  a*b ~= sum_{i+j<=2} a_i*b_j   -- 6 bf16 MFMAs, fp32 accumulate: fp32-class accuracy at 16/6 = 2.7x
//              the fp32-MFMA rate (SURVEY.md 7.2 option a, "split-bf16");
//     NP = 1:  a*b ~= a_0*b_0 (round-to-nearest-even planes) -- plain bf16 compute, fp32 accumulate, for
//              BASELINE.json configs[4] (bf16, tolerance 1e-2).
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles; global -> registers (prefetched
// one slab ahead) -> split -> LDS planes -> fragments.  LDS images:
//   k-contiguous operand:  plane[rows][40] bf16 (80-B rows, 5 16-B slots: odd -> conflict-free ds_read_b128 fragments).
//   m-contiguous operand (weight / data gradients): 16-byte loads along m, kept in memory order as plane[32 k][rows + 16]
//   and transposed by the READ: two ds_read_b64_tr_b16 per fragment (tools/ubench/tr16_probe.hip pins what that
//   instruction returns).  History: round 1 gathered such fragments with 8 ds_read_u16 (LDS-issue bound); transposing by
//   thread assignment instead (one row, 16 consecutive k per thread) needs 32 dword wave-loads per thread per slab and is
//   bound by the texture-address unit (16 clk per wave-load whatever its width) -- both 190 us on dW_logit, this form 157.
//   Operands that are not 16-byte loadable fall back to the thread-assignment form (k-contiguous image).
// Same argument struct, XCD-aware tile order, split-K (fp32 atomics) and epilogue as xg_gemm.hip.
#include "xg_common.h"
#include "xg_kernels.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDKC = BK + 8;       // k-contiguous image: row stride in bf16 (80 B)

struct BArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, relu, accumulate, splitk;
    int gm;   // tile rows per group of the tile order (xg_kernels.h: xgk_group_rows)
    // operands that ALREADY ARE bf16 in memory (same shape / layout, lda / ldb in elements): loaded 16 bytes = 8 elements at a
    // time and stored into the LDS image as they are -- half the operand bytes from L2, no convert (NP = 1 kernels only)
    const unsigned short* A16; const unsigned short* B16;
    // weight-gradient layout only (A m-contiguous = dY^T): csum[q][m] += sum_k A(k, m) for up to three accumulators -- the bias
    // gradient(s) of the same dY, formed by the tn == 0 tiles from the A slabs they stream anyway (as xg_gemm.hip's fp32
    // kernels do; round 4: no separate column-reduction pass over dY in the bf16 / split-bf16 modes either)
    float* csum[3];
};

constexpr int LDMC = BM + 16;      // [k][m] image of an m-contiguous operand: row stride in bf16 (288 B: the 4 k rows of a
                                   // transposing read land 8 banks apart)
// an m-contiguous operand whose rows are 16-byte loadable keeps its memory order in LDS and is transposed by the READ
// (ds_read_b64_tr_b16); otherwise it is transposed by the thread assignment of dword loads into the k-contiguous image
template <bool KC, bool VEC> constexpr bool tr_image() { return !KC && VEC; }
template <bool KC, bool VEC = false> constexpr int plane_elems() { return tr_image<KC, VEC>() ? BK * LDMC : BM * LDKC; }
typedef short v4s __attribute__((ext_vector_type(4)));

// ---- fp32 -> bf16 planes
template <int NP>
__device__ __forceinline__ void split(float x, unsigned short (&h)[NP]) {
    if constexpr (NP == 1) {
        unsigned u = __float_as_uint(x);
        u += 0x7FFFu + ((u >> 16) & 1u);                 // round to nearest even
        h[0] = (unsigned short)(u >> 16);
    } else {
        unsigned u0 = __float_as_uint(x) & 0xFFFF0000u;
        const float r1 = x - __uint_as_float(u0);        // exact
        unsigned u1 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(u1);       // exact
        unsigned u2 = __float_as_uint(r2);
        u2 += 0x7FFFu + ((u2 >> 16) & 1u);
        h[0] = (unsigned short)(u0 >> 16);
        h[1] = (unsigned short)(u1 >> 16);
        h[NP - 1] = (unsigned short)(u2 >> 16);
    }
}

// Buffer descriptor over an operand (base made provably wave-uniform; no bounds: the callers clamp).  The m-contiguous
// loads use it as `SGPR row offset + per-lane column offset`: the row arithmetic stays on the scalar unit.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float* P) {
    const uint64_t a = reinterpret_cast<uint64_t>(P);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, -1, 0x00020000);
}

// ---- global -> registers: 4 float4 per thread per operand per slab (same index maps as xg_gemm.hip).
// Loads are UNCONDITIONAL (out-of-range rows / k are clamped to a valid address) so the compiler keeps all eight in
// flight under the MFMAs -- a predicated load makes it wait for each one (measured: 8 serialized round trips per
// slab); the out-of-range lanes are zeroed later, when the registers are split into LDS.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int r0, int k0, int nrows, int K, f32x4 (&regs)[4]) {
    const int t = threadIdx.x;
    if (tr_image<KC, VEC>()) {
        // element (r, k) at P[k * ld + r]: 16-byte loads along r (k = f >> 5, r = (f & 31) * 4), kept in that order in LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i;
            const int k = f >> 5, r = (f & 31) << 2;
            regs[i] = *reinterpret_cast<const f32x4*>(P + (size_t)min(k0 + k, K - 1) * ld + min(r0 + r, nrows - 4));
        }
        return;
    }
    if (!KC) {
        // element (r, k) at P[k * ld + r]: this thread's row r = t & 127, k = (t >> 7) * 16 + 4 i + j
        // (the k rows are wave-uniform: saying so keeps their address arithmetic on the scalar unit -- SGPR row base + one
        // per-lane offset -- instead of 16 64-bit vector multiply-adds per operand per slab)
        const unsigned cr = (unsigned)min(r0 + (t & 127), nrows - 1);
        const int kb = k0 + (__builtin_amdgcn_readfirstlane(t >> 7) << 4);
        const __amdgpu_buffer_rsrc_t rs = operand_rsrc(P);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                regs[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs, (int)(cr * 4u), (int)((unsigned)min(kb + 4 * i + j, K - 1) * (unsigned)ld * 4u), 0));
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i;
        const int r = f >> 3, k = (f & 7) << 2;
        const int gr = r0 + r, gk = k0 + k;
        if (VEC) {
            // the vectorised extent is a multiple of 4, so a float4 is entirely in or entirely out: clamp to the last one
            const float* src = P + (size_t)min(gr, nrows - 1) * ld + min(gk, K - 4);
            regs[i] = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) regs[i][j] = P[(size_t)min(gr, nrows - 1) * ld + min(gk + j, K - 1)];
        }
    }
}

// ---- an operand that is bf16 in memory: 2 x 16-byte loads per thread per slab (8 elements each), straight into the LDS image
//  KC: thread -> row f >> 2, k = 8 (f & 3);   !KC (m-contiguous, [k][m] image read back transposed): k = f >> 4, 8 rows from 8 (f & 15)
template <bool KC>
__device__ __forceinline__ void load_tile16(const unsigned short* __restrict__ P, int ld, int r0, int k0, int nrows, int K, uint4 (&regs)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            const int r = f >> 2, k = (f & 3) << 3;
            regs[i] = *reinterpret_cast<const uint4*>(P + (size_t)min(r0 + r, nrows - 1) * ld + min(k0 + k, K - 8));
        } else {
            const int k = f >> 4, r = (f & 15) << 3;
            regs[i] = *reinterpret_cast<const uint4*>(P + (size_t)min(k0 + k, K - 1) * ld + min(r0 + r, nrows - 8));
        }
    }
}
template <bool KC>
__device__ __forceinline__ void store_tile16(unsigned short* __restrict__ lds, const uint4 (&regs)[2], int r0, int k0, int nrows, int K,
                                             bool edge) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = t + 256 * i;
        int r, k;
        if (KC) { r = f >> 2; k = (f & 3) << 3; } else { k = f >> 4; r = (f & 15) << 3; }
        uint4 v = regs[i];
        if (edge) {
            unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = KC ? (r0 + r < nrows && k0 + k + j < K) : (r0 + r + j < nrows && k0 + k < K);
                if (!ok) wds[j >> 1] &= (j & 1) ? 0x0000FFFFu : 0xFFFF0000u;
            }
            v = uint4{wds[0], wds[1], wds[2], wds[3]};
        }
        *reinterpret_cast<uint4*>(lds + (KC ? r * LDKC + k : k * LDMC + r)) = v;
    }
}

// ---- registers -> (zero the out-of-range lanes) -> split -> LDS planes: 4 consecutive elements of the contiguous
// dimension = one 8-byte store per plane
template <int NP, bool KC, bool VEC>
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ lds, const f32x4 (&regs)[4], int r0, int k0, int nrows,
                                           int K, bool edge) {
    constexpr bool TR = tr_image<KC, VEC>();
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i;
        int r, k;                                  // 4 consecutive k (k .. k + 3) of row r -- or, TR, 4 consecutive rows at k
        if (KC) { r = f >> 3; k = (f & 7) << 2; }
        else if (TR) { k = f >> 5; r = (f & 31) << 2; }
        else    { r = t & 127; k = ((t >> 7) << 4) + 4 * i; }
        f32x4 v = regs[i];
        if (edge) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (TR ? (r0 + r + j < nrows && k0 + k < K) : (r0 + r < nrows && k0 + k + j < K)) ? v[j] : 0.f;
        }
        const int off = TR ? k * LDMC + r : r * LDKC + k;
        if constexpr (NP == 1) {
            // plain bf16: the hardware's packed round-to-nearest-even convert (v_cvt_pk_bf16_f32), one instruction per pair
            // instead of ~4 integer operations per element (269 -> 294 TF on the vocabulary products).  (A 64-deep,
            // double-buffered variant of this kernel was measured SLOWER, 267 TF: at bf16 rates a slab is 0.2 us of MFMA
            // against ~1.5 us of load latency, so what is missing is bytes in flight.  An LDS-DMA variant (fp32 slabs
            // global_load_lds -> two-stage ring, 2 workgroups per CU, rounding on the LDS -> fragment path) was also
            // measured: correct, but 64 KB in flight per CU against this kernel's 96 KB of staging registers -- logits
            // 187 vs 167 us, mid-size 67 vs 46 us, only the weight-gradient layout 5 % ahead; not kept.  What would
            // help is HALF the bytes: bf16 operand copies in memory.  Two register sets + two LDS buffers (two slabs of
            // loads in flight per workgroup, one barrier per slab, 2 workgroups per CU instead of 3) were measured too:
            // the per-workgroup slab time does not change (1.37 us on the weight-gradient layout either way), so fewer
            // resident workgroups just lose -- dW_logit 190 -> 281 us.  The kernel moves ~18 TB/s of fp32 operands from L2
            // to the CUs: it is bound by that fill rate (32 flop per operand byte at 128x128 tiles), not by latency.)
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            bf16x2_t lo, hi;
            lo[0] = (__bf16)v[0]; lo[1] = (__bf16)v[1]; hi[0] = (__bf16)v[2]; hi[1] = (__bf16)v[3];
            uint2 w;
            w.x = __builtin_bit_cast(unsigned, lo); w.y = __builtin_bit_cast(unsigned, hi);
            *reinterpret_cast<uint2*>(lds + off) = w;
            continue;
        }
        unsigned short h[4][NP];
#pragma unroll
        for (int j = 0; j < 4; ++j) split<NP>(v[j], h[j]);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            uint2 w;
            w.x = (unsigned)h[0][p] | ((unsigned)h[1][p] << 16);
            w.y = (unsigned)h[2][p] | ((unsigned)h[3][p] << 16);
            *reinterpret_cast<uint2*>(lds + p * plane_elems<KC, VEC>() + off) = w;
        }
    }
}

// ---- LDS -> MFMA fragment: 8 consecutive k (k0..k0+7) of row `row`
template <bool KC, bool VEC>
__device__ __forceinline__ bf16x8 read_frag(const unsigned short* __restrict__ plane, int row, int k0) {
    if (!tr_image<KC, VEC>()) return *reinterpret_cast<const bf16x8*>(plane + row * LDKC + k0);
    // [k][m] image: ds_read_b64_tr_b16 hands lane i of a 16-lane group element (i & 3) of the 8 bytes addressed by lanes
    // (i >> 2) + 4 j, j = 0..3 -- so the lane with in-group index s points at (k = kbase + (s >> 2), m = mbase + 4 (s & 3)) and
    // receives k = kbase .. kbase + 3 of row mbase + i.  Two reads = the 8 consecutive k of the MFMA fragment.
    typedef __attribute__((address_space(3))) v4s lds_v4s;
    const int s = threadIdx.x & 15;
    const unsigned short* p = plane + (k0 + (s >> 2)) * LDMC + (row & ~15) + ((s & 3) << 2);
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)p);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(p + 4 * LDMC));
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    return v;
}

template <int NP, bool AKC, bool BKC, bool VEC, bool A16 = false, bool B16 = false>
__global__ void __launch_bounds__(256) gemm_bs_kernel(BArgs g) {
    static_assert(!(A16 || B16) || (NP == 1 && VEC), "bf16 operands: plain-bf16 vector kernels only");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_bs[];
    unsigned short* As = smem_bs;
    unsigned short* Bs = smem_bs + NP * plane_elems<AKC, VEC>();

    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int nwg = ntm * ntn * g.splitk;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ks = bid % g.splitk;
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
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab_all = (g.K + BK - 1) / BK;
    const int s_begin = (int)(((long)ks * nslab_all) / g.splitk), s_end = (int)(((long)(ks + 1) * nslab_all) / g.splitk);
    const bool edge_a = m0 + BM > g.M, edge_b = n0 + BN > g.N;     // wave-uniform: interior tiles skip the masking
    auto slab_mfma = [&](const unsigned short* Asb, const unsigned short* Bsb) {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int k0 = kk * 16 + half * 8;
            bf16x8 fa[2][NP], fb[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i][p] = read_frag<AKC, VEC>(Asb + p * plane_elems<AKC, VEC>(), wm * 64 + i * 32 + l31, k0);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j][p] = read_frag<BKC, VEC>(Bsb + p * plane_elems<BKC, VEC>(), wn * 64 + j * 32 + l31, k0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // smallest terms first; only plane pairs with pa + pb <= NP - 1 are above fp32 round-off
#pragma unroll
                    for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
                        for (int pa = 0; pa <= sum; ++pa)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][sum - pa], acc[i][j], 0, 0, 0);
                }
        }
    };
    // column sums of the A slabs (see BArgs::csum): per thread the rows it loads -- 8 (bf16 operand), 4 (16-byte fp32 loads) or 1
    constexpr int CSN = AKC ? 1 : (A16 ? 8 : (VEC ? 4 : 1));
    const bool cs_on = !AKC && g.csum[0] != nullptr && tn == 0;
    float csv[CSN];
#pragma unroll
    for (int e = 0; e < CSN; ++e) csv[e] = 0.f;
    {
    f32x4 ra[A16 ? 1 : 4], rb[B16 ? 1 : 4];
    uint4 ha[2], hb[2];
    auto cs_add = [&](int s, bool edge) {        // (loads are clamped at the edges: duplicates there are masked out here)
        if constexpr (!AKC) {
            const int t = threadIdx.x;
            if constexpr (A16) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int f = t + 256 * i, k = f >> 4, r = (f & 15) << 3;
                    const unsigned wds[4] = {ha[i].x, ha[i].y, ha[i].z, ha[i].w};
                    const bool kok = !edge || s * BK + k < g.K;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = __uint_as_float((j & 1) ? (wds[j >> 1] & 0xFFFF0000u) : (wds[j >> 1] << 16));
                        if (kok && (!edge || m0 + r + j < g.M)) csv[j] += v;
                    }
                }
            } else if constexpr (VEC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = t + 256 * i, k = f >> 5, r = (f & 31) << 2;
                    const bool kok = !edge || s * BK + k < g.K;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (kok && (!edge || m0 + r + j < g.M)) csv[j] += ra[i][j];
                }
            } else {
                const int kb = (t >> 7) << 4;
                const bool rok = m0 + (t & 127) < g.M;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (rok && s * BK + kb + 4 * i + j < g.K) csv[0] += ra[i][j];
            }
        }
    };
    auto loadA = [&](int s) { if constexpr (A16) load_tile16<AKC>(g.A16, g.lda, m0, s * BK, g.M, g.K, ha); else load_tile<AKC, VEC>(g.A, g.lda, m0, s * BK, g.M, g.K, ra); };
    auto loadB = [&](int s) { if constexpr (B16) load_tile16<BKC>(g.B16, g.ldb, n0, s * BK, g.N, g.K, hb); else load_tile<BKC, VEC>(g.B, g.ldb, n0, s * BK, g.N, g.K, rb); };
    loadA(s_begin); loadB(s_begin);
    for (int s = s_begin; s < s_end; ++s) {
        const bool ktail = (s + 1) * BK > g.K;
        __syncthreads();                                   // everyone is done reading the previous slab
        if (cs_on) cs_add(s, edge_a || ktail);
        if constexpr (A16) store_tile16<AKC>(As, ha, m0, s * BK, g.M, g.K, edge_a || ktail);
        else store_tile<NP, AKC, VEC>(As, ra, m0, s * BK, g.M, g.K, edge_a || ktail);
        if constexpr (B16) store_tile16<BKC>(Bs, hb, n0, s * BK, g.N, g.K, edge_b || ktail);
        else store_tile<NP, BKC, VEC>(Bs, rb, n0, s * BK, g.N, g.K, edge_b || ktail);
        __syncthreads();
        if (s + 1 < s_end) { loadA(s + 1); loadB(s + 1); }   // next slab's loads fly under this slab's MFMAs
        slab_mfma(As, Bs);
    }
    }
    if constexpr (!AKC) {
        if (cs_on) {                                       // (wave-uniform: the whole workgroup is in tile column 0 or not)
            __syncthreads();                               // the staging area is free: [256][CSN] partial sums
            float* red = reinterpret_cast<float*>(smem_bs);
#pragma unroll
            for (int e = 0; e < CSN; ++e) red[threadIdx.x * CSN + e] = csv[e];
            __syncthreads();
            const int r = threadIdx.x;
            if (r < BM && m0 + r < g.M) {
                float v = 0.f;
                if constexpr (A16) { for (int j = 0; j < 16; ++j) v += red[((r >> 3) + 16 * j) * CSN + (r & 7)]; }
                else if constexpr (VEC) { for (int j = 0; j < 8; ++j) v += red[((r >> 2) + 32 * j) * CSN + (r & 3)]; }
                else { v = red[r] + red[r + 128]; }
#pragma unroll
                for (int o = 0; o < 3; ++o) if (g.csum[o]) unsafeAtomicAdd(g.csum[o] + m0 + r, v);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (g.bias && ks == 0) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
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

// ---- plain bf16, large tiles ("bx") -------------------------------------------------------------------------------
// gemm_bs_kernel<1,...> moves ~18 TB/s of fp32 operands from L2 to the CUs -- the chip's operand fill rate -- at
// 32 flop per operand byte (128 x 128 tiles): that, not the matrix cores, is its bound.  The same staging with TM x TN
// tiles and WGM x WGN waves (each wave (TM / WGM) x (TN / WGN) as 32 x 32 MFMA tiles) raises the reuse: 256 x 128 tiles,
// 8 waves, = 43 flop per byte.  Slab s+1's loads fly under slab s's MFMAs.
template <int ROWS, int THREADS, bool KC>
__device__ __forceinline__ void bx_load(const float* __restrict__ P, int ld, int r0, int k0, int nrows, int K,
                                        f32x4 (&regs)[ROWS * 8 / THREADS]) {
    constexpr int NV = ROWS * 8 / THREADS;                     // float4 per thread per operand per slab
    const int t = threadIdx.x;
    if (!KC) {                                                 // transposing assignment: row t % ROWS, NV * 4 consecutive k
        static_assert(THREADS % ROWS == 0 && (THREADS / ROWS) * NV * 4 == BK, "thread map of the m-contiguous operand");
        const unsigned cr = (unsigned)min(r0 + (t % ROWS), nrows - 1);
        const int kb = k0 + __builtin_amdgcn_readfirstlane(t / ROWS) * (NV * 4);
        const __amdgpu_buffer_rsrc_t rs = operand_rsrc(P);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                regs[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs, (int)(cr * 4u), (int)((unsigned)min(kb + 4 * i + j, K - 1) * (unsigned)ld * 4u), 0));
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int f = t + THREADS * i;
        const int r = f >> 3, k = (f & 7) << 2;
        regs[i] = *reinterpret_cast<const f32x4*>(P + (size_t)min(r0 + r, nrows - 1) * ld + min(k0 + k, K - 4));
    }
}
template <int ROWS, int THREADS, bool KC>
__device__ __forceinline__ void bx_store(unsigned short* __restrict__ lds, const f32x4 (&regs)[ROWS * 8 / THREADS], int r0, int k0,
                                         int nrows, int K, bool edge) {
    constexpr int NV = ROWS * 8 / THREADS;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int r, k;
        if (KC) { const int f = t + THREADS * i; r = f >> 3; k = (f & 7) << 2; }
        else    { r = t % ROWS; k = (t / ROWS) * (NV * 4) + 4 * i; }
        f32x4 v = regs[i];
        if (edge) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (r0 + r < nrows && k0 + k + j < K) ? v[j] : 0.f;
        }
        bf16x2_t lo, hi;
        lo[0] = (__bf16)v[0]; lo[1] = (__bf16)v[1]; hi[0] = (__bf16)v[2]; hi[1] = (__bf16)v[3];
        uint2 w;
        w.x = __builtin_bit_cast(unsigned, lo); w.y = __builtin_bit_cast(unsigned, hi);
        *reinterpret_cast<uint2*>(lds + r * LDKC + k) = w;
    }
}

template <int TM, int TN, int WGM, int WGN, bool AKC, bool BKC>
__global__ void __launch_bounds__(64 * WGM * WGN) gemm_bx_kernel(BArgs g) {
    constexpr int THREADS = 64 * WGM * WGN, WM = TM / WGM, WN = TN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int NVA = TM * 8 / THREADS, NVB = TN * 8 / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_bx[];
    unsigned short* As = smem_bx;
    unsigned short* Bs = smem_bx + TM * LDKC;

    const int ntm = (g.M + TM - 1) / TM, ntn = (g.N + TN - 1) / TN;
    const int nwg = ntm * ntn * g.splitk;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ks = bid % g.splitk;
    bid /= g.splitk;
    int tm, tn;
    {
        const int per = g.gm * ntn, grp = bid / per, in = bid - grp * per;
        const int first = grp * g.gm, gsz = min(ntm - first, g.gm);
        tn = in / gsz;
        tm = first + (in - tn * gsz);
    }
    const int m0 = tm * TM, n0 = tn * TN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN, half = lane >> 5, l31 = lane & 31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab_all = (g.K + BK - 1) / BK;
    const int s_begin = (int)(((long)ks * nslab_all) / g.splitk), s_end = (int)(((long)(ks + 1) * nslab_all) / g.splitk);
    const bool edge_a = m0 + TM > g.M, edge_b = n0 + TN > g.N;
    f32x4 ra[NVA], rb[NVB];
    bx_load<TM, THREADS, AKC>(g.A, g.lda, m0, s_begin * BK, g.M, g.K, ra);
    bx_load<TN, THREADS, BKC>(g.B, g.ldb, n0, s_begin * BK, g.N, g.K, rb);
    for (int s = s_begin; s < s_end; ++s) {
        const bool ktail = (s + 1) * BK > g.K;
        __syncthreads();
        bx_store<TM, THREADS, AKC>(As, ra, m0, s * BK, g.M, g.K, edge_a || ktail);
        bx_store<TN, THREADS, BKC>(Bs, rb, n0, s * BK, g.N, g.K, edge_b || ktail);
        __syncthreads();
        if (s + 1 < s_end) {
            bx_load<TM, THREADS, AKC>(g.A, g.lda, m0, (s + 1) * BK, g.M, g.K, ra);
            bx_load<TN, THREADS, BKC>(g.B, g.ldb, n0, (s + 1) * BK, g.N, g.K, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int k0 = kk * 16 + half * 8;
            bf16x8 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + (wm * WM + i * 32 + l31) * LDKC + k0);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * WN + j * 32 + l31) * LDKC + k0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
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

template <int TM, int TN, int WGM, int WGN, bool AKC, bool BKC>
int launch_bx(hipStream_t st, const BArgs& g) {
    const int ntm = xg_cdiv(g.M, TM), ntn = xg_cdiv(g.N, TN);
    if (g.splitk > 1 && !g.accumulate) {
        if (g.ldc == g.N) { if (hipMemsetAsync(g.C, 0, sizeof(float) * (size_t)g.M * g.N, st) != hipSuccess) return XG_EHIP; }
        else if (hipMemset2DAsync(g.C, sizeof(float) * g.ldc, 0, sizeof(float) * g.N, g.M, st) != hipSuccess) return XG_EHIP;
    }
    constexpr int lds = (TM + TN) * LDKC * (int)sizeof(unsigned short);
    hipLaunchKernelGGL((gemm_bx_kernel<TM, TN, WGM, WGN, AKC, BKC>), dim3(ntm * ntn * g.splitk), dim3(64 * WGM * WGN), lds, st, g);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
template <int NP, bool AKC, bool BKC, bool VEC, bool A16 = false, bool B16 = false>
int launch(hipStream_t st, const BArgs& g) {
    const int ntm = xg_cdiv(g.M, BM), ntn = xg_cdiv(g.N, BN);
    if (g.splitk > 1 && !g.accumulate) {
        if (g.ldc == g.N) { if (hipMemsetAsync(g.C, 0, sizeof(float) * (size_t)g.M * g.N, st) != hipSuccess) return XG_EHIP; }
        else if (hipMemset2DAsync(g.C, sizeof(float) * g.ldc, 0, sizeof(float) * g.N, g.M, st) != hipSuccess) return XG_EHIP;
    }
    const size_t lds = (size_t)NP * (plane_elems<AKC, VEC>() + plane_elems<BKC, VEC>()) * sizeof(unsigned short);
    if (lds > 65536) {
        static std::atomic<unsigned> optin{0};
        XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&gemm_bs_kernel<NP, AKC, BKC, VEC, A16, B16>), (int)lds));
    }
    hipLaunchKernelGGL((gemm_bs_kernel<NP, AKC, BKC, VEC, A16, B16>), dim3(ntm * ntn * g.splitk), dim3(256), lds, st, g);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

// plain bf16 with one or both operands already bf16 in memory
template <bool AKC, bool BKC>
int dispatch16(hipStream_t st, const BArgs& g) {
    if (g.A16 && g.B16) return launch<1, AKC, BKC, true, true, true>(st, g);
    if (g.A16) return launch<1, AKC, BKC, true, true, false>(st, g);
    return launch<1, AKC, BKC, true, false, true>(st, g);
}

template <int NP>
int dispatch(hipStream_t st, const BArgs& g, bool akc, bool bkc, bool vec) {
#define XG_BS(a, b) (vec ? launch<NP, a, b, true>(st, g) : launch<NP, a, b, false>(st, g))
    if (akc && bkc) return XG_BS(true, true);
    if (akc && !bkc) return XG_BS(true, false);
    if (!akc && !bkc) return XG_BS(false, false);
    return XG_BS(false, true);
#undef XG_BS
}

// fp32 -> bf16 (round to nearest even, v_cvt_pk_bf16_f32), 8 elements per thread: two 16-byte loads, one 16-byte store
__global__ void __launch_bounds__(256) cvt_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n8, size_t n) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
        bf16x2_t p0, p1, p2, p3;
        p0[0] = (__bf16)a[0]; p0[1] = (__bf16)a[1]; p1[0] = (__bf16)a[2]; p1[1] = (__bf16)a[3];
        p2[0] = (__bf16)b[0]; p2[1] = (__bf16)b[1]; p3[0] = (__bf16)b[2]; p3[1] = (__bf16)b[3];
        *reinterpret_cast<uint4*>(dst + i * 8) = uint4{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1),
                                                       __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
    } else {
        // tail (and the whole range when it is not 16-byte aligned: n8 = 0): one element per thread of the trailing workgroups
        const size_t j = n8 * 8 + (i - n8);
        if (j < n) {
            unsigned u = __float_as_uint(src[j]);
            u += 0x7FFFu + ((u >> 16) & 1u);
            dst[j] = (unsigned short)(u >> 16);
        }
    }
}

}  // namespace

int xgk_cvt_bf16(hipStream_t st, const float* src, unsigned short* dst, size_t n) {
    if (n == 0) return XG_OK;
    if (!src || !dst) return XG_EINVAL;
    const size_t n8 = (((uintptr_t)src % 16) || ((uintptr_t)dst % 16)) ? 0 : n / 8;      // unaligned: everything through the scalar tail
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)((n8 + (n - n8 * 8) + 255) / 256)), dim3(256), 0, st, src, dst, n8, n);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

extern "C" int xg_cvt_bf16(void* stream, const float* src, void* dst, int64_t n) {
    if (n < 0) return XG_EINVAL;
    return xgk_cvt_bf16((hipStream_t)stream, src, static_cast<unsigned short*>(dst), (size_t)n);
}
extern "C" int xg_gemm_bf16_operands(void* stream, int transA, int transB, int M, int N, int K, const float* A, const void* A16, int lda,
                                     const float* B, const void* B16, int ldb, float* C, int ldc, const float* bias, int relu,
                                     int accumulate) {
    if (M <= 0 || N <= 0) return XG_OK;
    if (K < 0 || !A || !B || !C) return XG_EINVAL;
    return xgk_gemm_bf16x((hipStream_t)stream, 1, transA != 0, transB != 0, M, N, K, A, static_cast<const unsigned short*>(A16), lda, B,
                          static_cast<const unsigned short*>(B16), ldb, C, ldc, bias, relu != 0, accumulate != 0, nullptr, nullptr, nullptr);
}

// planes: 1 = bf16 compute, 3 = split-bf16 (fp32-class accuracy).  Only called for products large enough to tile.
int xgk_gemm_bf16(hipStream_t st, int planes, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
                  const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate, float* cs1, float* cs2,
                  float* cs3) {
    return xgk_gemm_bf16x(st, planes, transA, transB, M, N, K, A, nullptr, lda, B, nullptr, ldb, C, ldc, bias, relu, accumulate, cs1, cs2, cs3);
}

// the same with optional bf16 copies of the operands (A16 / B16: same shape, layout and leading dimension as A / B, or null):
// where one exists and is 16-byte loadable the kernel reads it instead of converting the fp32 operand on the fly
int xgk_gemm_bf16x(hipStream_t st, int planes, bool transA, bool transB, int M, int N, int K, const float* A, const unsigned short* A16,
                   int lda, const float* B, const unsigned short* B16, int ldb, float* C, int ldc, const float* bias, bool relu,
                   bool accumulate, float* cs1, float* cs2, float* cs3) {
    // (column sums are a side output of the weight-gradient layout only: transA)
    if (cs1 && !transA) return XG_EINVAL;
    const bool bg = (planes & XGK_GEMM_BG) != 0;      // launched beside a latency-bound chain (xg_kernels.h)
    planes &= ~XGK_GEMM_BG;
    BArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, relu ? 1 : 0, accumulate ? 1 : 0, 1, 1, nullptr, nullptr, {cs1, cs2, cs3}};
    const bool akc = !transA, bkc = transB;
    bool vec = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && (lda % 4 == 0) && (ldb % 4 == 0);
    vec = vec && ((akc ? K : M) % 4 == 0) && ((bkc ? K : N) % 4 == 0);
    if (planes == 1 && vec) {      // 16-byte loads of 8 bf16: base, pitch and contiguous extent in multiples of 8 elements
        if (A16 && (uintptr_t)A16 % 16 == 0 && lda % 8 == 0 && (akc ? K : M) % 8 == 0 && (akc ? K : M) >= 8) g.A16 = A16;
        if (B16 && (uintptr_t)B16 % 16 == 0 && ldb % 8 == 0 && (bkc ? K : N) % 8 == 0 && (bkc ? K : N) >= 8) g.B16 = B16;
    }
    const long tiles = (long)xg_cdiv(M, BM) * xg_cdiv(N, BN);
    const int nslab = xg_cdiv(K, BK);
    if (!relu && tiles < 512) {                     // fill the chip by splitting deep reductions (3 workgroups per CU would fit,
                                                    // but 768 shares were measured slower: dX 52 -> 83 us, dW_logit 157 -> 169 us)
        // ... and never PAST the 512 slots (round 5): the rule used to round up, so 104 tiles became 5 x 104 = 520 workgroups -- a
        // second round for 8 of them -- and 416 tiles two rounds of 832.  Rounded down (split-bf16, alone): encoder embedding
        // 98 -> 63 us, dX 110 -> 74, PRE 104 -> 59, v2a(V) 94 -> 58 (tools/ubench/bs_skmax.sh)
        long sk = 512 / tiles;
        if (sk > nslab / 8) sk = nslab / 8;
        if (sk >= 2) g.splitk = (int)sk;
    }
    // plain bf16 with a k-contiguous A (forward and data-gradient layouts): 256 x 128 tiles, 43 flop per operand byte instead
    // of 32 -- logits 170 -> 145 us, PRE 79 -> 59 us; hidden-1024 iteration 9.19 -> 9.02 ms.  Measured and not used: the
    // weight-gradient layout on these tiles (dW_logit 186 -> 276 us) and 256 x 256 tiles (one workgroup per CU: 10.4 ms).
    static const bool no_bx = xg_diag_env("XG_NO_BX") != nullptr;
    if (planes == 1 && vec && akc && !no_bx && M >= 256 && !g.A16 && !g.B16) {
        const long t2 = (long)xg_cdiv(M, 256) * xg_cdiv(N, 128);
        g.splitk = 1;
        if (!relu && t2 < 256) {
            long sk = (256 + t2 - 1) / t2;
            if (sk > nslab / 8) sk = nslab / 8;
            if (sk >= 2) g.splitk = (int)sk;
        }
        g.gm = xgk_group_rows(2 * K / g.splitk);
        return bkc ? launch_bx<256, 128, 4, 2, true, true>(st, g) : launch_bx<256, 128, 4, 2, true, false>(st, g);
    }
    // both operands bf16 in memory: tiles by LDS-DMA, 64-deep slabs, one barrier per slab (xg_gemm_g16.hip)
    static const bool no_g16 = xg_diag_env("XG_NO_G16") != nullptr;
    if (g.A16 && g.B16 && !no_g16 && xgk_gemm_g16_ok(transA, transB, M, N, K, g.A16, lda, g.B16, ldb))
        return xgk_gemm_g16(st, transA, transB, M, N, K, g.A16, lda, g.B16, ldb, C, ldc, bias, relu, accumulate, bg ? -1 : 0, cs1, cs2, cs3);
    g.gm = xgk_group_rows(K / g.splitk);
    if (g.A16 || g.B16) {
        if (akc && bkc) return dispatch16<true, true>(st, g);
        if (akc && !bkc) return dispatch16<true, false>(st, g);
        if (!akc && !bkc) return dispatch16<false, false>(st, g);
        return dispatch16<false, true>(st, g);
    }
    return planes == 1 ? dispatch<1>(st, g, akc, bkc, vec) : dispatch<3>(st, g, akc, bkc, vec);
}
