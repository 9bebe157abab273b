// Skinny (M <= a few hundred rows) fp32 MFMA GEMM for the recurrent steps of the decoder / encoder.
//
// The per-timestep products are (B x K) x (K x N) with B = 128: too few output tiles to fill 256 CUs
// with an ordinary tiled GEMM, and each 32x32 MFMA tile needs K/2 dependent 64-cycle MFMAs.  So the
// parallelism comes from K: one workgroup = one 32x32 output tile, its 8 waves split the reduction
// dimension 8-way, stream their A / W fragments straight from L2 into registers (nothing is shared
// between waves, so LDS staging would be pure overhead), and the 8 partial tiles are reduced through
// LDS.  Several independent products ("jobs") ride in ONE launch so a decoder step is a handful of
// launches, and the LSTM cell arithmetic (reference caption_src/sub_modules.py:752-767) runs in the
// epilogue of the product that feeds it: tiles of the cell products are laid out as 8 hidden units
// x 4 gates so one tile holds everything a unit's cell update needs.
//
//   job:   C[M,N] (+)= sum_s A_s[M,K_s] * op(B_s) + sum biases        (up to 3 K-segments)
//   op(B): (N,K) row-major "k-contiguous" (nn.Linear weight, forward)  or  (K,N) row-major (data gradient)
#include "xg_common.h"
#include "xg_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SKW = 8;            // waves per workgroup (K split)
constexpr int SKT = SKW * 64;     // threads
constexpr int NPF = 4;            // 8-deep k blocks in flight per wave

template <bool VEC>
__device__ __forceinline__ f32x4 ld4_k(const float* __restrict__ row, int k, int K, bool ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        if (VEC) { if (k < K) v = *reinterpret_cast<const f32x4*>(row + k); }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = row[k + j];
        }
    }
    return v;
}
// B stored (K,N): element (k, n) at B[k*ldb + n]; 4 consecutive k for a fixed column
__device__ __forceinline__ f32x4 ld4_n(const float* __restrict__ col, int ldb, int k, int K, bool ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = col[(size_t)(k + j) * ldb];
    }
    return v;
}

template <bool VEC>
__global__ void __launch_bounds__(SKT) sk_kernel(SkArgs args) {
    __shared__ __attribute__((aligned(16))) float red[SKW][32][32];
    // ---- which job / tile (XCD-aware: the m-tiles that share a weight slice stay on one XCD's L2)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int ji = 0;
#pragma unroll
    for (int j = 1; j < SK_MAX_JOBS; ++j) if (j < args.njobs && bid >= args.job[j].tile0) ji = j;
    const SkJob& job = args.job[ji];
    const int tile = bid - job.tile0;
    const int ntm = (job.M + 31) >> 5;
    const int tm = tile % ntm, tn = tile / ntm;
    const int m0 = tm * 32, n0 = tn * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;

    // B row / output column owned by this lane.  LSTM tiles: 8 units x 4 gates.
    const int R = job.R;
    int ncol;
    if (job.epi == SK_EPI_LSTM) ncol = (l31 >> 3) * R + tn * 8 + (l31 & 7);
    else ncol = n0 + l31;
    const bool n_ok = job.epi == SK_EPI_LSTM ? (tn * 8 + (l31 & 7)) < R : ncol < job.N;
    const int mrow = m0 + l31;
    const bool m_ok = mrow < job.M;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // ---- K loop: this wave's share of the 8-deep blocks of every segment
    int nb_total = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) if (s < job.nseg) nb_total += (job.seg[s].K + 7) >> 3;
    const int wb0 = (wave * nb_total) / SKW, wb1 = ((wave + 1) * nb_total) / SKW;
    int seg_start = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= job.nseg) break;
        const SkSeg sg = job.seg[s];
        const int nb = (sg.K + 7) >> 3;
        const int b0 = max(wb0, seg_start) - seg_start, b1 = min(wb1, seg_start + nb) - seg_start;
        seg_start += nb;
        if (b0 >= b1) continue;
        const float* arow = sg.A + (size_t)mrow * sg.lda;
        const float* brow = sg.b_ncontig ? sg.B + ncol : sg.B + (size_t)ncol * sg.ldb;
        f32x4 fa[NPF], fb[NPF];
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int k = (b0 + j) * 8 + half * 4;
            const bool live = b0 + j < b1;
            fa[j] = ld4_k<VEC>(arow, k, sg.K, m_ok && live);
            fb[j] = sg.b_ncontig ? ld4_n(brow, sg.ldb, k, sg.K, n_ok && live) : ld4_k<VEC>(brow, k, sg.K, n_ok && live);
        }
        for (int i = b0; i < b1; i += NPF) {
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const f32x4 a = fa[j], b = fb[j];
                const int nxt = i + NPF + j;
                const int k = nxt * 8 + half * 4;
                const bool live = nxt < b1;
                fa[j] = ld4_k<VEC>(arow, k, sg.K, m_ok && live);
                fb[j] = sg.b_ncontig ? ld4_n(brow, sg.ldb, k, sg.K, n_ok && live) : ld4_k<VEC>(brow, k, sg.K, n_ok && live);
                if (i + j < b1) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
                }
            }
        }
    }
    // ---- reduce the SKW partial tiles through LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();

    if (job.epi == SK_EPI_STORE) {
#pragma unroll
        for (int e = 0; e < 1024 / SKT; ++e) {
            const int idx = threadIdx.x + SKT * e;
            const int m = idx >> 5, c = idx & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < SKW; ++w) v += red[w][m][c];
            const int row = m0 + m, col = n0 + c;
            if (row < job.M && col < job.N) {
                if (job.bias[0]) v += job.bias[0][col];
                if (job.bias[1]) v += job.bias[1][col];
                if (job.bias[2]) v += job.bias[2][col];
                float* dst = job.C + (size_t)row * job.ldc + col;
                if (job.accumulate) v += *dst;
                if (job.relu) v = fmaxf(v, 0.f);
                *dst = v;
            }
        }
    } else {
        // LSTM cell epilogue: thread -> (row m, unit u); its four gate pre-activations sit at columns u + 8*gate
        if (threadIdx.x < 256) {
            const int m = threadIdx.x >> 3, u = threadIdx.x & 7;
            const int b = m0 + m, j = tn * 8 + u;
            if (b < job.M && j < R) {
                float s[4];
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < SKW; ++w) v += red[w][m][gi * 8 + u];
                    const int col = gi * R + j;
                    if (job.bias[0]) v += job.bias[0][col];
                    if (job.bias[1]) v += job.bias[1][col];
                    if (job.bias[2]) v += job.bias[2][col];
                    if (job.add) v += job.add[(size_t)b * job.ldadd + col];
                    s[gi] = v;
                }
                const float so = job.order == XG_ORDER_IFOG ? s[2] : s[3];
                const float sg_ = job.order == XG_ORDER_IFOG ? s[3] : s[2];
                const float ig = xg_sigmoid(s[0]), fg = xg_sigmoid(s[1]), og = xg_sigmoid(so), gg = xg_tanh(sg_);
                const float cp = job.c_prev[(size_t)b * job.ldcp + j];
                const float mk = job.mask ? job.mask[(size_t)b * job.ldm] : 1.0f;
                float cn = fg * cp + ig * gg, hn;
                if (job.mask_mode == XG_MASK_HOLD) {
                    cn = cn * mk + cp * (1.0f - mk);
                    hn = og * xg_tanh(cn);
                    hn = hn * mk + job.h_prev[(size_t)b * job.ldhp + j] * (1.0f - mk);
                } else {
                    hn = og * xg_tanh(cn) * mk;
                    cn = cn * mk;
                }
                hn *= xg_keep(job.drop, (uint32_t)(b * R + j));
                if (job.gates) {
                    float* g = job.gates + (size_t)b * job.ldg;
                    g[j] = ig; g[R + j] = fg;
                    if (job.order == XG_ORDER_IFOG) { g[2 * R + j] = og; g[3 * R + j] = gg; }
                    else                            { g[2 * R + j] = gg; g[3 * R + j] = og; }
                }
                job.c_out[(size_t)b * job.ldco + j] = cn;
                job.h_out[(size_t)b * job.ldho + j] = hn;
            }
        }
    }
}

}  // namespace

int xgk_skinny(hipStream_t st, SkArgs& a) {
    if (a.njobs <= 0 || a.njobs > SK_MAX_JOBS) return XG_EINVAL;
    bool vec = true;
    int tiles = 0;
    for (int j = 0; j < a.njobs; ++j) {
        SkJob& jb = a.job[j];
        if (jb.M <= 0 || jb.N <= 0 || jb.nseg < 1 || jb.nseg > 3) return XG_EINVAL;
        jb.tile0 = tiles;
        const int ntm = xg_cdiv(jb.M, 32);
        int ntn;
        if (jb.epi == SK_EPI_LSTM) {
            if (jb.N != 4 * jb.R) return XG_EINVAL;
            ntn = xg_cdiv(jb.R, 8);
        } else {
            ntn = xg_cdiv(jb.N, 32);
        }
        tiles += ntm * ntn;
        for (int s = 0; s < jb.nseg; ++s) {
            const SkSeg& sg = jb.seg[s];
            if (!sg.A || !sg.B || sg.K <= 0) return XG_EINVAL;
            vec = vec && ((uintptr_t)sg.A % 16 == 0) && (sg.lda % 4 == 0) && (sg.K % 4 == 0);
            if (!sg.b_ncontig) vec = vec && ((uintptr_t)sg.B % 16 == 0) && (sg.ldb % 4 == 0);
        }
    }
    if (vec) hipLaunchKernelGGL((sk_kernel<true>), dim3(tiles), dim3(SKT), 0, st, a);
    else hipLaunchKernelGGL((sk_kernel<false>), dim3(tiles), dim3(SKT), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
