// bf16 GEMM for operands that are BOTH bf16 in memory (gemm_mode 1 with weight shadows / activation mirrors): the tiles go
// global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction), never through registers.
//
//   C[M,N] (+)= op(A) op(B) (+ bias, relu), fp32 accumulate, v_mfma_f32_32x32x16_bf16.
//   128 x 128 tiles, a wave's share 64 x 64 = 2 x 2 MFMA tiles; a ring of NS LDS stages of TK-deep slabs, filled by DMA and drained
//   by COUNT (s_waitcnt vmcnt(N) + the raw s_barrier: __syncthreads() would drain the whole DMA queue), ONE barrier per slab; the
//   DMA instructions of the next slab sit between the MFMA groups of the slab being multiplied.  Three forms (xgk_gemm_g16 picks):
//     TK 64, NS 2, four waves  -- 64 KiB, two workgroups per CU: the default;
//     TK 32, NS 3, four waves  -- 48 KiB, three workgroups per CU: the weight-gradient layout with >= 1024 tiles;
//     TK 64, NS 4, EIGHT waves -- two groups of four share the tile and split every slab's depth, 128 KiB, one workgroup per CU:
//                                 weight gradients of at most one tile per CU.
//
// The predecessor (xg_gemm_bf16.hip: gemm_bs_kernel<1,..,A16,B16>) staged 32-deep slabs through registers into ONE LDS image with
// two barriers per slab: 516-576 TF alone on the hidden-1024 vocabulary shapes.  What an LDS-DMA costs is that the LDS image of
// a wave-instruction is lane-linear (base + 16 * lane): no padding.  Bank conflicts are removed by a swizzle that is applied
// on the SOURCE side (which 16 bytes of global memory a lane fetches) and undone by the fragment reads:
//   k-contiguous operand  (rows x 64 k, 128 B per row):  16-byte slot j of row r is stored in slot j ^ ((r >> 1) & 7).
//        A ds_read_b128 fragment read (32 rows x one slot per half-wave) is serviced in the 16-lane groups
//        {0-3,12-15,20-27} / {4-11,16-19,28-31}: their 8 even and 8 odd rows get 8 distinct slots of each 128-byte half of the
//        256-byte bank row -> conflict-free.
//   m-contiguous operand  (64 k x 128 m, 256 B per k row = one bank row; transposed by ds_read_b64_tr_b16):  the four 64-byte
//        quarters of k row k are stored XOR-ed with (k & 3): the four k rows of one transposing read land in four quarters.
// Edges: the DMA goes through a buffer descriptor whose extent is the operand's; a fetch past the end writes zeros.  Rows past
// the last row of a k-contiguous operand and k rows past K of an m-contiguous operand are therefore zero for free; columns
// past the end of an INNER row fetch the next row's head (finite data), which only ever meets unstored outputs or a zero row
// of the other operand -- the dispatcher (xgk_gemm_g16_ok) admits exactly the shapes for which that holds.
#include "xg_common.h"
#include "xg_kernels.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 128, TN = 128;
// TK = slab depth (64 or 32), NS = LDS stages of the ring (NS - 1 slabs in flight while one is multiplied)
template <int TK> constexpr int opb() { return 128 * TK * 2; }     // bytes of one operand's stage image
template <int TK> constexpr int stb() { return 2 * opb<TK>(); }    // bytes of one stage

struct GArgs {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, relu, accumulate, splitk, gm;
    float* csum[3];
    int pad_;
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g16_rsrc(const unsigned short* P, unsigned bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(P);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

// per-lane byte offsets of a wave's four DMA instructions into the operand, slab 0 (the range check of a raw buffer looks at
// the VECTOR offset only, so the whole offset lives there; a slab advance is one add per instruction)
template <bool KC, int TK, int NI>
__device__ __forceinline__ void g16_offsets(unsigned (&off)[NI], int ld, int r0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave * NI + i;
        if (KC) {       // TK = 64: instruction q = rows 8q .. 8q + 7; lane -> row 8q + (lane >> 3), stored slot lane & 7
                        // TK = 32: rows 16q .. 16q + 15 of 64 bytes; lane -> row 16q + (lane >> 2), stored slot lane & 3
            constexpr int LPR = TK / 8;                      // lanes (16-byte slots) per row
            const int r = q * (64 / LPR) + lane / LPR;
            const int x = TK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3);
            const int j = (lane % LPR) ^ x;
            off[i] = ((unsigned)(r0 + r) * (unsigned)ld + (unsigned)j * 8u) * 2u;
        } else {        // instruction q = k rows 4q .. 4q + 3; lane -> k row 4q + (lane >> 4), stored slot lane & 15
            const int k = q * 4 + (lane >> 4);
            const int j = (lane & 15) ^ ((k & 3) << 2);
            off[i] = ((unsigned)k * (unsigned)ld + (unsigned)(r0 + j * 8)) * 2u;
        }
    }
}

// KG = 2: eight waves -- two groups of four that share the tile and split every slab's depth between them (same tile, same operand
// traffic, twice the waves per workgroup: ONE workgroup per CU with a four-stage ring = three slabs in flight instead of one);
// the groups exchange the half of the accumulators they do not store through LDS at the end.
template <bool AKC, bool BKC, int TK, int NS, int KG>
__global__ void __launch_bounds__(256 * KG, 2) gemm_g16_kernel(GArgs g) {
    constexpr int NI = TK / 16 / KG, OPB = opb<TK>(), STB = stb<TK>(), P = NS - 1;      // NI: DMA instructions per wave, operand and slab
    static_assert(KG == 1 || (KG == 2 && TK == 64), "two wave groups: 64-deep slabs, two 16-deep blocks each");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_g16[];

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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w4 = wave & 3, kg = wave >> 2;
    const int wm = w4 >> 1, wn = w4 & 1, half = lane >> 5, l31 = lane & 31;

    const int nslab_all = (g.K + TK - 1) / TK;
    const int s_begin = (int)(((long)ks * nslab_all) / g.splitk), s_end = (int)(((long)(ks + 1) * nslab_all) / g.splitk);

    // ---- DMA side
    const __amdgpu_buffer_rsrc_t rsA = g16_rsrc(g.A, AKC ? ((unsigned)(g.M - 1) * g.lda + g.K) * 2u : ((unsigned)(g.K - 1) * g.lda + g.M) * 2u);
    const __amdgpu_buffer_rsrc_t rsB = g16_rsrc(g.B, BKC ? ((unsigned)(g.N - 1) * g.ldb + g.K) * 2u : ((unsigned)(g.K - 1) * g.ldb + g.N) * 2u);
    unsigned offA[NI], offB[NI];
    g16_offsets<AKC, TK, NI>(offA, g.lda, m0, wave, lane);
    g16_offsets<BKC, TK, NI>(offB, g.ldb, n0, wave, lane);
    const unsigned slabA = AKC ? TK * 2u : (unsigned)TK * (unsigned)g.lda * 2u;
    const unsigned slabB = BKC ? TK * 2u : (unsigned)TK * (unsigned)g.ldb * 2u;
#pragma unroll
    for (int i = 0; i < NI; ++i) { offA[i] += (unsigned)s_begin * slabA; offB[i] += (unsigned)s_begin * slabB; }
    // (Every workgroup walks its slabs from the first one.  Workgroups of one round run in step and the tiles that share an operand
    // panel then ask L2 for the same slab at the same moment; a tile-dependent start -- tried against the suspicion that 2 KB row
    // pitches pile every request of a moment onto two L2 channels -- made every shape 5-45 % SLOWER: tools/ubench/g16_rot.sh.)
    auto issue = [&](int buf) {
        unsigned char* base = smem_g16 + buf * STB + wave * (NI * 1024);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + i * 1024), 16, (int)offA[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + OPB + i * 1024), 16, (int)offB[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) { offA[i] += slabA; offB[i] += slabB; }
    };
    // the same slab one DMA instruction at a time (piece 0 .. 2 NI - 1: A's, then B's): placed BETWEEN the MFMA groups of the slab that
    // is being multiplied -- issued in one burst in front of them, a wave's eight instructions hold its instruction stream for their
    // whole issue time (the address unit takes them at 100-200 cycles apiece under load) before its first fragment read
    auto issue_piece = [&](int buf, int pc) {
        unsigned char* base = smem_g16 + buf * STB + wave * (NI * 1024);
        if (pc < NI) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(base + pc * 1024), 16, (int)offA[pc], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(base + OPB + (pc - NI) * 1024), 16, (int)offB[pc - NI], 0, 0, 0);
    };
    auto issue_advance = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) { offA[i] += slabA; offB[i] += slabB; }
    };

    // ---- fragment side: byte addresses inside an operand's stage image
    // k-contiguous: row * 128 + (((2 kk + half) ^ x) << 4), x = (row >> 1) & 7 = (l31 >> 1) & 7 for both 32-row tiles of the wave
    // m-contiguous: (kk * 16 + half * 8 [+ 4] + kq) * 256 + ((2 m) ^ (kq << 6)), kq = (lane & 15) >> 2, m = wave origin + i * 32 + (l31 & 16) + 4 (lane & 3)
    const int s16 = lane & 15, kq = s16 >> 2;
    const unsigned xa = TK == 64 ? (unsigned)((l31 >> 1) & 7) : (unsigned)((l31 >> 2) & 3);
    const unsigned fa0 = AKC ? (unsigned)(wm * 64 + l31) * (TK * 2u)
                             : (unsigned)(half * 8 + kq) * 256u + (((unsigned)(wm * 64 + (l31 & 16) + ((s16 & 3) << 2)) * 2u) ^ ((unsigned)kq << 6));
    const unsigned fb0 = BKC ? (unsigned)(wn * 64 + l31) * (TK * 2u)
                             : (unsigned)(half * 8 + kq) * 256u + (((unsigned)(wn * 64 + (l31 & 16) + ((s16 & 3) << 2)) * 2u) ^ ((unsigned)kq << 6));

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradients as a side output (weight-gradient layout: A m-contiguous): column sums of the A slabs of tile column 0
    const bool cs_on = !AKC && g.csum[0] != nullptr && tn == 0;
    float csv0 = 0.f, csv1 = 0.f;

    auto frag = [&](const unsigned char* img, bool kc, unsigned f0, int i, int kk) -> bf16x8 {
        if (kc) {
            const unsigned a = f0 + (unsigned)i * (32u * TK * 2u) + ((((unsigned)(2 * kk + half)) ^ xa) << 4);
            return *reinterpret_cast<const bf16x8*>(img + a);
        }
        typedef __attribute__((address_space(3))) v4s lds_v4s;
        // (i * 32 rows = 64 bytes: bit 6 of the in-row offset, which the XOR with kq << 6 may have set -> add, not or, after removing it)
        const unsigned inrow = ((f0 & 255u) ^ ((unsigned)kq << 6)) + (unsigned)i * 64u;
        const unsigned a = (f0 & ~255u) + (unsigned)kk * (16u * 256u) + (inrow ^ ((unsigned)kq << 6));
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(img + a));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(img + a + 4 * 256));
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        return v;
    };

    // ring of NS stages: slabs s + 1 .. s + P are in flight while slab s is multiplied.  The DMA queue is drained by COUNT
    // (a wave's 2 NI instructions per slab retire in order), and the barrier is the raw instruction: __syncthreads() would
    // drain the whole queue in front of every barrier.
    int nis = 0;                                                // slabs issued so far
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (s_begin + p < s_end) { issue(p); ++nis; }
    int buf = 0, ibuf = P % NS;
    for (int s = s_begin; s < s_end; ++s) {
        const int rem = nis - 1 - (s - s_begin);                // slabs that may stay in flight once slab s has landed
        if (rem >= 3) { if constexpr (P >= 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI * 3) : "memory"); }
        else if (rem == 2) { if constexpr (P >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI * 2) : "memory"); }
        else if (rem == 1) { if constexpr (P >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI * 1) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // slab s is in LDS for everybody; everybody is done with slab s - 1
        __builtin_amdgcn_sched_barrier(0);
        const bool more = s + P < s_end;                        // ... whose stage takes slab s + P
        const int jbuf = ibuf;
        if (more) ++nis;
        ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
        const unsigned char* Ai = smem_g16 + buf * STB;
        const unsigned char* Bi = Ai + OPB;
        if (cs_on) {
            // thread -> m pair (t & 63), k rows wave * TK / (4 KG) .. of the [TK k][128 m] image
            const unsigned mb = (unsigned)(threadIdx.x & 63) * 4u;
#pragma unroll
            for (int e = 0; e < TK / (4 * KG); ++e) {
                const unsigned k = (unsigned)(wave * (TK / (4 * KG)) + e);
                const unsigned w = *reinterpret_cast<const unsigned*>(Ai + k * 256u + (mb ^ ((k & 3u) << 6)));
                csv0 += __uint_as_float(w << 16);
                csv1 += __uint_as_float(w & 0xFFFF0000u);
            }
        }
        {
            // every fragment of the slab is requested before its first MFMA (a wave's LDS reads then run under its own MFMAs,
            // not only under the other waves'); the scheduler is told to keep that order
            constexpr int KK = TK / 16 / KG;                    // 16-deep blocks of the slab this wave multiplies
            const int kk0 = kg * KK;
            bf16x8 fa[KK][2], fb[KK][2];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[kk][i] = frag(Ai, AKC, fa0, i, kk0 + kk);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[kk][j] = frag(Bi, BKC, fb0, j, kk0 + kk);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PPK = 2 * NI / KK;                    // DMA pieces per 16-deep block
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (more) {
#pragma unroll
                    for (int q = 0; q < PPK; ++q) issue_piece(jbuf, kk * PPK + q);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) issue_advance();
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }

    if constexpr (!AKC) {
        if (cs_on) {                                            // (workgroup-uniform)
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem_g16);
            red[threadIdx.x * 2 + 0] = csv0;
            red[threadIdx.x * 2 + 1] = csv1;
            __syncthreads();
            const int r = threadIdx.x;
            if (r < TM && m0 + r < g.M) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4 * KG; ++w) v += red[((r >> 1) + 64 * w) * 2 + (r & 1)];
#pragma unroll
                for (int o = 0; o < 3; ++o) if (g.csum[o]) unsafeAtomicAdd(g.csum[o] + m0 + r, v);
            }
        }
    }
    auto store_rows = [&](int i, const f32x16 (&ai)[2]) {
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
                    float v = ai[j][r] + bv;
                    if (g.splitk > 1) { unsafeAtomicAdd(dst, v); continue; }
                    if (g.accumulate) v += *dst;
                    if (g.relu) v = fmaxf(v, 0.f);
                    *dst = v;
                }
            }
        }
    };
    if constexpr (KG == 1) {
        store_rows(0, acc[0]);
        store_rows(1, acc[1]);
    } else {
        // group kg stores the wave's 32-row block i = kg: it hands the other block's partial sums to the other group through LDS
        // ([group][wave][j][r][lane] floats, 32 KB per group) and adds what the other group left for its own block
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                        // every wave is done with the stage images
        float* X = reinterpret_cast<float*>(smem_g16);
        float* mine = X + kg * 8192 + w4 * 2048 + lane;
        const float* theirs = X + (1 - kg) * 8192 + w4 * 2048 + lane;
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64] = acc[1][j][r];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64] = acc[0][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][j][r] += theirs[(j * 16 + r) * 64];
            store_rows(0, acc[0]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[1][j][r] += theirs[(j * 16 + r) * 64];
            store_rows(1, acc[1]);
        }
    }
}

template <bool AKC, bool BKC, int TK, int NS, int KG>
int launch_g16(hipStream_t st, const GArgs& g, int extra_lds = 0) {
    const int ntm = xg_cdiv(g.M, TM), ntn = xg_cdiv(g.N, TN);
    if (g.splitk > 1 && !g.accumulate) {
        if (g.ldc == g.N) { if (hipMemsetAsync(g.C, 0, sizeof(float) * (size_t)g.M * g.N, st) != hipSuccess) return XG_EHIP; }
        else if (hipMemset2DAsync(g.C, sizeof(float) * g.ldc, 0, sizeof(float) * g.N, g.M, st) != hipSuccess) return XG_EHIP;
    }
    const int lds = NS * stb<TK>() + extra_lds;     // (extra_lds: diag -- a product beside a chain kept to one workgroup per CU)
    if (lds > 65536) {
        static std::atomic<unsigned> optin{0};
        XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&gemm_g16_kernel<AKC, BKC, TK, NS, KG>), lds));
    }
    hipLaunchKernelGGL((gemm_g16_kernel<AKC, BKC, TK, NS, KG>), dim3(ntm * ntn * g.splitk), dim3(256 * KG), lds, st, g);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
template <int TK, int NS, int KG = 1>
int launch_g16_layout(hipStream_t st, const GArgs& g, bool akc, bool bkc, int extra_lds = 0) {
    if (akc && bkc) return launch_g16<true, true, TK, NS, KG>(st, g, extra_lds);
    if (akc && !bkc) return launch_g16<true, false, TK, NS, KG>(st, g, extra_lds);
    if (!akc && !bkc) return launch_g16<false, false, TK, NS, KG>(st, g, extra_lds);
    return launch_g16<false, true, TK, NS, KG>(st, g, extra_lds);
}

}  // namespace

// Shapes the DMA kernel handles (see the header on edges): 16-byte loadable rows, 32-bit byte offsets, and a K tail only where
// one operand supplies zeros for it (an m-contiguous operand: its k rows past K are past the end of the buffer) while the
// other's overrun stays inside its own data (k-contiguous with lda == K: the next row's head; or m-contiguous as well).
bool xgk_gemm_g16_ok(bool transA, bool transB, int M, int N, int K, const unsigned short* A16, int lda, const unsigned short* B16, int ldb) {
    const bool akc = !transA, bkc = transB;
    if (!A16 || !B16 || M < 128 || N < 128 || K < 128) return false;
    if ((uintptr_t)A16 % 16 || (uintptr_t)B16 % 16 || lda % 8 || ldb % 8) return false;
    if ((akc ? K : M) % 8 || (bkc ? K : N) % 8) return false;
    const int64_t ea = (int64_t)((akc ? M : K) + 128) * lda * 2, eb = (int64_t)((bkc ? N : K) + 128) * ldb * 2;
    if (ea >= (int64_t)1 << 31 || eb >= (int64_t)1 << 31) return false;
    if (K % 64) {
        const bool a_zero = !akc, b_zero = !bkc;                  // zero k rows past K for free
        const bool a_safe = !akc || lda == K, b_safe = !bkc || ldb == K;
        if (!((a_zero && b_safe) || (b_zero && a_safe))) return false;
    }
    return true;
}

int xgk_gemm_g16(hipStream_t st, bool transA, bool transB, int M, int N, int K, const unsigned short* A16, int lda,
                 const unsigned short* B16, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate, int splitk,
                 float* cs1, float* cs2, float* cs3) {
    if (cs1 && !transA) return XG_EINVAL;
    const bool akc = !transA, bkc = transB;
    GArgs g{A16, B16, C, bias, M, N, K, lda, ldb, ldc, relu ? 1 : 0, accumulate ? 1 : 0, 1, 1, {cs1, cs2, cs3}, 0};
    // Configuration <wave groups, slab depth, ring stages> (tools/ubench/g16_cfg.sh / g16_burst.sh, products alone):
    //   64-deep slabs in two stages, four waves, two workgroups per CU ("642") everywhere but the weight-gradient layout, where
    //   - with at most one tile per CU, EIGHT waves per tile win (two groups of four that split every slab's depth; one workgroup
    //     per CU, four-stage ring: "844"): 4096 x 1024 x 5120 58-65 against 71-73 us, 4096 x 1024 x 2688 36-40 against 48-53,
    //     1024 x 1536 x 5120 40-43 against 48-54 -- and lose everywhere else (logits 241 against 200, dH 193 against 143);
    //   - with >= 1024 tiles, 32-deep slabs in three stages (three workgroups per CU, "323") are 10-20 % ahead (dW_logit 143
    //     against 173 us); wherever an operand is k-contiguous they lose (64-byte row pieces = half cache lines).
    //   One workgroup of four waves per CU with three 64-deep stages loses everywhere (277 against 188 us on the logits): what this
    //   kernel needs is waves to switch to, not bytes in flight.
    const long tiles = (long)xg_cdiv(M, TM) * xg_cdiv(N, TN);
    static const char* cfg = xg_diag_env("XG_G16_CFG");     // diag build: 642 / 643 / 322 / 323 / 324 / 325 / 843 / 844
    const bool tn_layout = !akc && !bkc;
    const int c = cfg ? atoi(cfg) : (tn_layout && tiles >= 1024 ? 323 : (tn_layout && tiles <= 256 ? 844 : 642));
    // Split of the reduction across workgroups (`splitk` <= 0: this kernel's own rule; the register-staged kernel's rule
    // filled 512 slots whenever the tiles did not).  A part's result is added with fp32 atomics behind a memset of C, and that
    // epilogue is expensive here: 5120 x 1024 x 1536 takes 34 us unsplit and 58-70 us in two parts, 5120 x 1024 x 4096 76 against
    // 85-114 us.  Measured rule (tools/ubench/g16_sk.sh): split only while tiles x parts stay ONE round of the slots (512 with two
    // workgroups per CU, 256 with one) and every part keeps a reduction of >= 2560 -- 2688 x 1024 x 20000: 3 parts (158 us; 1 / 2 / 4
    // parts: 322 / 198 / 228) -- or, while the tiles alone leave CUs empty, of >= 1024 (1024 x 1536 x 5120, 96 tiles, four waves:
    // 87 / 65 / 55 / 51 / 52 us in 1 / 2 / 3 / 4 / 5 parts).
    // (splitk == -1: the product was launched beside a latency-bound chain, XGK_GEMM_BG.  Keeping it to ONE workgroup per CU there
    //  was measured at the end of round 5 and is inside the noise: gone.)
    constexpr int extra_lds = 0;
    if (splitk > 0) g.splitk = splitk;
    else if (!relu) {
        const long slots = (c >= 800 || extra_lds) ? 256 : 512;
        const int deep = tiles < 256 ? 1024 : 2560;
        long sk = slots / tiles;
        if (sk > K / deep) sk = K / deep;
        g.splitk = sk < 1 ? 1 : (int)sk;
    }
    g.gm = xgk_group_rows(K / g.splitk / 2);          // (an operand panel is 128 x k_depth bf16 = half the bytes the rule was made for)
    switch (c) {
    case 323: return launch_g16_layout<32, 3>(st, g, akc, bkc);
    case 844: return launch_g16_layout<64, 4, 2>(st, g, akc, bkc);       // eight waves (two k groups), one workgroup per CU
#ifdef XG_DIAG
    case 322: return launch_g16_layout<32, 2>(st, g, akc, bkc);
    case 843: return launch_g16_layout<64, 3, 2>(st, g, akc, bkc);
    case 643: return launch_g16_layout<64, 3>(st, g, akc, bkc);
    case 324: return launch_g16_layout<32, 4>(st, g, akc, bkc);
    case 325: return launch_g16_layout<32, 5>(st, g, akc, bkc);
#endif
    default: return launch_g16_layout<64, 2>(st, g, akc, bkc, extra_lds);
    }
}
