// Element-wise / reduction kernels of the CG encoder and the decoder cells (gfx950).
// All of them are HBM/L2-bound byte movers: one thread per element (or float4), coalesced along
// the feature dimension, 256-thread workgroups, grids sized to the element count.
#include "xg_common.h"
#include "xg_kernels.h"

namespace {

constexpr int TPB = 256;
inline dim3 grid1(int64_t n) { return dim3((unsigned)xg_cdiv64(n, TPB)); }

// ------------------------------------------------------------------ LSTM cell pointwise
// column blocks of s: order IFOG -> [i f o g], IFGO -> [i f g o]
__global__ void lstm_fwd_kernel(LstmFwdArgs a) {
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= a.B * a.R) return;
    const int b = idx / a.R, j = idx % a.R, R = a.R;
    const float* s = a.s + (size_t)b * a.lds_;
    float si = s[j], sf = s[R + j], s2 = s[2 * R + j], s3 = s[3 * R + j];
    if (a.add) {
        const float* ad = a.add + (size_t)b * a.ldadd;
        si += ad[j]; sf += ad[R + j]; s2 += ad[2 * R + j]; s3 += ad[3 * R + j];
    }
    const float so = a.order == XG_ORDER_IFOG ? s2 : s3;
    const float sg = a.order == XG_ORDER_IFOG ? s3 : s2;
    const float ig = xg_sigmoid(si), fg = xg_sigmoid(sf), og = xg_sigmoid(so), gg = xg_tanh(sg);
    const float cp = a.c_prev[(size_t)b * a.ldcp + j];
    const float m = a.mask ? a.mask[(size_t)b * a.ldm] : 1.0f;
    float cn = fg * cp + ig * gg;
    float hn;
    if (a.mask_mode == XG_MASK_HOLD) {
        cn = cn * m + cp * (1.0f - m);                       // sub_modules.py:762
        hn = og * xg_tanh(cn);
        const float hp = a.h_prev[(size_t)b * a.ldhp + j];
        hn = hn * m + hp * (1.0f - m);                       // sub_modules.py:765
    } else {
        hn = og * xg_tanh(cn) * m;                           // sub_modules.py:139-140
        cn = cn * m;
    }
    hn *= xg_keep(a.drop, (uint32_t)idx);                    // sub_modules.py:767
    if (a.gates) {
        float* g = a.gates + (size_t)b * a.ldg;
        g[j] = ig; g[R + j] = fg;
        if (a.order == XG_ORDER_IFOG) { g[2 * R + j] = og; g[3 * R + j] = gg; }
        else                          { g[2 * R + j] = gg; g[3 * R + j] = og; }
    }
    a.c_out[(size_t)b * a.ldco + j] = cn;
    a.h_out[(size_t)b * a.ldho + j] = hn;
}

__device__ __forceinline__ void lstm_bwd_body(const LstmBwdArgs& a) {
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= a.B * a.R) return;
    const int b = idx / a.R, j = idx % a.R, R = a.R;
    const float* g = a.gates + (size_t)b * a.ldg;
    const float ig = g[j], fg = g[R + j];
    const float og = a.order == XG_ORDER_IFOG ? g[2 * R + j] : g[3 * R + j];
    const float gg = a.order == XG_ORDER_IFOG ? g[3 * R + j] : g[2 * R + j];
    const float cp = a.c_prev[(size_t)b * a.ldcp + j];
    const float cn = a.c_out[(size_t)b * a.ldco + j];
    const float m = a.mask ? a.mask[(size_t)b * a.ldm] : 1.0f;
    float dh = a.dh_out[(size_t)b * a.lddh + j];
    if (a.dh_add) dh += a.dh_add[(size_t)b * a.lddha + j];
    dh *= xg_keep(a.drop, (uint32_t)idx);
    float dc = a.dc_out ? a.dc_out[(size_t)b * a.lddc + j] : 0.0f;
    float dht, dct, dcp;
    // tanh(c'): HOLD uses the post-mask cell (sub_modules.py:763); ZERO: m in {0,1} so m*tanh(c_out) == m*tanh(cn)
    const float tc = xg_tanh(cn);
    if (a.mask_mode == XG_MASK_HOLD) {
        if (a.dh_prev) a.dh_prev[(size_t)b * a.lddhp + j] = (1.0f - m) * dh;
        dht = m * dh;
        dc += dht * og * (1.0f - tc * tc);
        dcp = (1.0f - m) * dc;
        dct = m * dc;
    } else {
        dht = m * dh;
        dct = m * dc + dht * og * (1.0f - tc * tc);
        dcp = 0.0f;
    }
    const float d_o = dht * tc;
    dcp += dct * fg;
    const float d_f = dct * cp, d_i = dct * gg, d_g = dct * ig;
    float* ds = a.ds + (size_t)b * a.ldds;
    ds[j] = d_i * ig * (1.0f - ig);
    ds[R + j] = d_f * fg * (1.0f - fg);
    const float dso = d_o * og * (1.0f - og), dsg = d_g * (1.0f - gg * gg);
    if (a.order == XG_ORDER_IFOG) { ds[2 * R + j] = dso; ds[3 * R + j] = dsg; }
    else                          { ds[2 * R + j] = dsg; ds[3 * R + j] = dso; }
    a.dc_prev[(size_t)b * a.lddcp + j] = dcp;
}
__global__ void lstm_bwd_kernel(LstmBwdArgs a) { XG_CHAIN_PRIO(); lstm_bwd_body(a); }
__global__ void lstm_bwd2_kernel(LstmBwdArgs a, LstmBwdArgs b) {
    XG_CHAIN_PRIO();
    if (blockIdx.y == 0) lstm_bwd_body(a); else lstm_bwd_body(b);
}

// ------------------------------------------------------------------ gates
// NV consecutive columns per thread (4 when R and every leading dimension are multiples of 4 and the bases are 16-byte
// aligned: one 16-byte access per operand instead of four 4-byte ones)
template <int NV> struct VecT { typedef float type; };
template <> struct VecT<4> { typedef float4 type; };
template <int NV> __device__ __forceinline__ void ldv(const float* p, float (&v)[NV]) {
    if (NV == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else v[0] = p[0];
}
template <int NV> __device__ __forceinline__ void stv(float* p, const float (&v)[NV]) {
    if (NV == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else p[0] = v[0];
}
// (both gate kernels take TWO parameter sets, blockIdx.y picks one: the encoder's two cross gates are one launch)
struct GateFwdP { float* pre_g; int ldp; const float* t; int ldt; int t_mod; float* y; int ldy; XgDrop drop; };
template <int NV>
__global__ void gate_fwd_kernel(GateFwdP pa, GateFwdP pb, int rows, int R, int s_div, int s_mod, int b_div, int b_mod) {
    const GateFwdP& p = blockIdx.y ? pb : pa;
    const int64_t idx = ((int64_t)blockIdx.x * TPB + threadIdx.x) * NV;
    if (idx >= (int64_t)rows * R) return;
    const int r = (int)(idx / R), j = (int)(idx % R);
    XgDrop dr = p.drop;
    dr.step = p.drop.step + (uint32_t)((r / s_div) % s_mod);
    const uint32_t e = (uint32_t)((r / b_div) % b_mod) * (uint32_t)R + (uint32_t)j;
    float g[NV], tv[NV], yv[NV];
    ldv<NV>(p.pre_g + (size_t)r * p.ldp + j, g);
    if (p.y) ldv<NV>(p.t + (size_t)(p.t_mod > 0 ? r % p.t_mod : r) * p.ldt + j, tv);
#pragma unroll
    for (int q = 0; q < NV; ++q) g[q] *= xg_keep(dr, e + q);
    stv<NV>(p.pre_g + (size_t)r * p.ldp + j, g);
    if (p.y) {
#pragma unroll
        for (int q = 0; q < NV; ++q) yv[q] = g[q] * tv[q] + tv[q];                // sub_modules.py:45
        stv<NV>(p.y + (size_t)r * p.ldy + j, yv);
    }
}
struct GateBwdP { const float* dy; int lddy; const float* g; int ldg; const float* t; int ldt; int t_mod; float* dpre; int lddp;
                  float* dt; int lddt; int dt_acc; XgDrop drop; };
template <int NV>
__global__ void gate_bwd_kernel(GateBwdP pa, GateBwdP pb, int rows, int R) {
    const GateBwdP& p = blockIdx.y ? pb : pa;
    const int64_t idx = ((int64_t)blockIdx.x * TPB + threadIdx.x) * NV;
    if (idx >= (int64_t)rows * R) return;
    const int r = (int)(idx / R), j = (int)(idx % R);
    float d[NV], gv[NV], tv[NV], o[NV];
    ldv<NV>(p.dy + (size_t)r * p.lddy + j, d);
    ldv<NV>(p.g + (size_t)r * p.ldg + j, gv);
    ldv<NV>(p.t + (size_t)(p.t_mod > 0 ? r % p.t_mod : r) * p.ldt + j, tv);
    if (p.dpre) {
#pragma unroll
        for (int q = 0; q < NV; ++q) o[q] = gv[q] > 0.f ? d[q] * tv[q] * p.drop.scale : 0.f;   // g>0 <=> relu active & kept
        stv<NV>(p.dpre + (size_t)r * p.lddp + j, o);
    }
    if (p.dt) {
        float* qd = p.dt + (size_t)r * p.lddt + j;
        if (p.dt_acc) ldv<NV>(qd, o);
#pragma unroll
        for (int q = 0; q < NV; ++q) o[q] = (p.dt_acc ? o[q] : 0.f) + d[q] * (gv[q] + 1.0f);
        stv<NV>(qd, o);
    }
}

__global__ void relu_drop_fwd_kernel(float* x, int64_t n, XgDrop drop) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) x[i] *= xg_keep(drop, (uint32_t)i);
}
__global__ void relu_drop_bwd_kernel(float* dy, const float* src, const float* y, int64_t n, XgDrop drop) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) dy[i] = y[i] > 0.f ? src[i] * drop.scale : 0.f;
}

// ------------------------------------------------------------------ column reductions
// block = 64 columns x 4 row lanes; grid.y chunks of rows; atomic accumulate into out.
template <int MODE>   // 0: sum x ; 1: sum x*y ; 2: sum (x-mean)^2
__global__ void colreduce_kernel(const float* X, int ldx, const float* Y, int ldy, int N, int Cn, float* out,
                                 float* out2, float* out3, int rows_per_chunk) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(N, r0 + rows_per_chunk);
    float acc = 0.f;
    if (c < Cn) {
        const float mu = MODE == 2 ? Y[c] : 0.f;
        for (int r = r0 + rl; r < r1; r += 4) {
            const float x = X[(size_t)r * ldx + c];
            if (MODE == 0) acc += x;
            else if (MODE == 1) acc += x * Y[(size_t)r * ldy + c];
            else { const float d = x - mu; acc += d * d; }
        }
    }
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < Cn) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(out + c, s);
        if (out2) atomicAdd(out2 + c, s);      // the same column sums into up to three accumulators (the three bias
        if (out3) atomicAdd(out3 + c, s);      // vectors of a two-input LSTM cell share one gradient)
    }
}

template <int MODE>
int colreduce(hipStream_t st, const float* X, int ldx, const float* Y, int ldy, int N, int Cn, float* out,
              float* out2 = nullptr, float* out3 = nullptr) {
    if (N <= 0 || Cn <= 0) return XG_OK;
    const int rpc = 64;
    dim3 grid(xg_cdiv(Cn, 64), xg_cdiv(N, rpc));
    hipLaunchKernelGGL((colreduce_kernel<MODE>), grid, dim3(256), 0, st, X, ldx, Y, ldy, N, Cn, out, out2, out3, rpc);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

// ------------------------------------------------------------------ BatchNorm
__global__ void scale_kernel(float* x, float s, int n) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n) x[i] *= s;
}
__global__ void bn_running_kernel(const float* mean, const float* var, float* rmean, float* rvar, int N, int R,
                                  float mom) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= R) return;
    const float unb = var[i] * ((float)N / (float)max(N - 1, 1));
    rmean[i] = (1.f - mom) * rmean[i] + mom * mean[i];
    rvar[i] = (1.f - mom) * rvar[i] + mom * unb;
}
template <int NV>
__global__ void bn_apply_kernel(const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ var,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ rowmask, float* __restrict__ X, int N, int R, float eps,
                                XgDrop drop) {
    const int64_t idx = ((int64_t)blockIdx.x * TPB + threadIdx.x) * NV;
    if (idx >= (int64_t)N * R) return;
    const int r = (int)(idx / R), c = (int)(idx % R);
    float z[NV], mu[NV], vr[NV], ga[NV], be[NV], o[NV];
    ldv<NV>(Z + idx, z); ldv<NV>(mean + c, mu); ldv<NV>(var + c, vr); ldv<NV>(gamma + c, ga); ldv<NV>(beta + c, be);
    const float rm = rowmask[r];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float xh = (z[q] - mu[q]) / sqrtf(vr[q] + eps);
        o[q] = fmaxf(xh * ga[q] + be[q], 0.f) * xg_keep(drop, (uint32_t)(idx + q)) * rm;
    }
    stv<NV>(X + idx, o);
}
// in place dX -> dY (grad wrt BN output), plus column sums of dY and dY*xhat
__global__ void bn_bwd_reduce_kernel(float* __restrict__ dX, const float* __restrict__ X, const float* __restrict__ Z,
                                     const float* __restrict__ mean, const float* __restrict__ var,
                                     const float* __restrict__ rowmask, int N, int R, float eps, XgDrop drop,
                                     float* sum_dy, float* sum_dyxhat, int rows_per_chunk) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(N, r0 + rows_per_chunk);
    float a0 = 0.f, a1 = 0.f;
    if (c < R) {
        const float mu = mean[c], is = 1.0f / sqrtf(var[c] + eps);
        for (int rb = r0 + rl; rb < r1; rb += 16) {       // four rows per thread in flight before any is consumed
            float xv[4], dv[4], zv[4], mk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = min(rb + 4 * q, r1 - 1);
                const size_t i = (size_t)r * R + c;
                xv[q] = X[i]; dv[q] = dX[i]; zv[q] = Z[i]; mk[q] = rowmask[r];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rb + 4 * q;
                if (r < r1) {
                    // X = relu(.)*keep*rowmask ; X>0 <=> relu active, kept, row unmasked
                    const float dy = xv[q] > 0.f ? dv[q] * drop.scale * mk[q] : 0.f;
                    dX[(size_t)r * R + c] = dy;
                    a0 += dy;
                    a1 += dy * (zv[q] - mu) * is;
                }
            }
        }
    }
    red[0][rl][threadIdx.x & 63] = a0;
    red[1][rl][threadIdx.x & 63] = a1;
    __syncthreads();
    if (rl == 0 && c < R) {
        const int t = threadIdx.x;
        atomicAdd(sum_dy + c, red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t]);
        atomicAdd(sum_dyxhat + c, red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t]);
    }
}
template <int NV>
__global__ void bn_bwd_apply_kernel(float* __restrict__ dY, const float* __restrict__ Z, const float* __restrict__ mean,
                                    const float* __restrict__ var, const float* __restrict__ gamma,
                                    const float* __restrict__ sum_dy, const float* __restrict__ sum_dyxhat, int N, int R,
                                    float eps, int train, float* g_beta, float* g_gamma) {
    const int64_t idx = ((int64_t)blockIdx.x * TPB + threadIdx.x) * NV;
    if (idx >= (int64_t)N * R) return;
    const int c = (int)(idx % R);
    float d[NV], z[NV], mu[NV], vr[NV], ga[NV], s1[NV], s2[NV];
    ldv<NV>(dY + idx, d); ldv<NV>(Z + idx, z); ldv<NV>(mean + c, mu); ldv<NV>(var + c, vr); ldv<NV>(gamma + c, ga);
    ldv<NV>(sum_dy + c, s1); ldv<NV>(sum_dyxhat + c, s2);
    if (g_beta && idx < R) {                         // row 0's threads: d(beta) += sum dy, d(gamma) += sum dy xhat (two axpy launches less)
#pragma unroll
        for (int q = 0; q < NV; ++q) { g_beta[c + q] += s1[q]; g_gamma[c + q] += s2[q]; }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float is = 1.0f / sqrtf(vr[q] + eps);
        float v = d[q];
        if (train) {
            const float xh = (z[q] - mu[q]) * is;
            v = v - s1[q] / (float)N - xh * s2[q] / (float)N;
        }
        d[q] = v * ga[q] * is;
    }
    stv<NV>(dY + idx, d);
}

// ------------------------------------------------------------------ embedding, means, misc
__device__ __forceinline__ int64_t tok_at(const int64_t* tok, int i, int inner, int64_t s_inner, int64_t s_outer, int V) {
    int64_t t = tok[(int64_t)(i % inner) * s_inner + (int64_t)(i / inner) * s_outer];
    return t < 0 ? 0 : (t >= V ? V - 1 : t);
}
__global__ void embed_gather_kernel(const float* table, int E, const int64_t* tok, int inner, int64_t s_inner,
                                    int64_t s_outer, int V, float* out, int ldo) {
    const int i = blockIdx.x;
    const int64_t t = tok_at(tok, i, inner, s_inner, s_outer, V);
    for (int e = threadIdx.x; e < E; e += blockDim.x) out[(size_t)i * ldo + e] = table[(size_t)t * E + e];
}
// one launch in front of a stand-alone decoder step: embedding rows of the B tokens + a copy of the recurrent state
__global__ void step_prep_kernel(const float* table, int E, const int64_t* tok, int V, float* xt, int B,
                                 const float* state, float* state_copy, int64_t nstate) {
    if ((int)blockIdx.x < B) {
        const int i = blockIdx.x;
        const int64_t t = tok_at(tok, i, B, 1, 0, V);
        for (int e = threadIdx.x; e < E; e += blockDim.x) xt[(size_t)i * E + e] = table[(size_t)t * E + e];
    } else {
        const int64_t i = ((int64_t)(blockIdx.x - B) * blockDim.x + threadIdx.x) * 4;
        if (i + 3 < nstate) *reinterpret_cast<float4*>(state_copy + i) = *reinterpret_cast<const float4*>(state + i);
        else for (int64_t j = i; j < nstate; ++j) state_copy[j] = state[j];
    }
}
// xg_rollout_compact: one launch for every tensor.  A workgroup moves 4096 floats of one entry.
constexpr int CPB = 4096;
__global__ void __launch_bounds__(256) compact_kernel(CompactArgs a) {
    int ei = 0;
    for (int i = 1; i < a.n; ++i) if ((int)blockIdx.x >= a.start[i]) ei = i;
    const CompactEntry& e = a.e[ei];
    const int64_t base = (int64_t)(blockIdx.x - a.start[ei]) * CPB;
    const bool vec = (e.dst_block % 4 == 0) && (e.src_pitch % 4 == 0);
    if (vec) {
#pragma unroll
        for (int u = 0; u < CPB / (256 * 4); ++u) {
            const int64_t i = base + ((int64_t)u * 256 + threadIdx.x) * 4;
            if (i < e.total) {
                const int64_t blk = i / e.dst_block, off = i - blk * e.dst_block;
                *reinterpret_cast<float4*>(e.dst + i) = *reinterpret_cast<const float4*>(e.src + blk * e.src_pitch + off);
            }
        }
    } else {
        for (int u = 0; u < CPB / 256; ++u) {
            const int64_t i = base + (int64_t)u * 256 + threadIdx.x;
            if (i < e.total) {
                const int64_t blk = i / e.dst_block, off = i - blk * e.dst_block;
                e.dst[i] = e.src[blk * e.src_pitch + off];
            }
        }
    }
}
__global__ void embed_scatter_kernel(float* dtable, int E, const int64_t* tok, int inner, int64_t s_inner,
                                     int64_t s_outer, int V, const float* dX, int ldx) {
    const int i = blockIdx.x;
    const int64_t t = tok_at(tok, i, inner, s_inner, s_outer, V);
    for (int e = threadIdx.x; e < E; e += blockDim.x) atomicAdd(dtable + (size_t)t * E + e, dX[(size_t)i * ldx + e]);
}
__global__ void masked_mean_kernel(const float* V, const float* mask, float* out, int B, int K, int R) {
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= B * R) return;
    const int b = idx / R, r = idx % R;
    float s = 0.f, ms = 0.f;
    for (int k = 0; k < K; ++k) { s += V[((size_t)b * K + k) * R + r]; ms += mask[b * K + k]; }
    out[idx] = s / ms;                                       // SAModel.py:59-61
}
__global__ void axpy_kernel(float* y, const float* x, float a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
__global__ void fill_kernel(float* y, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) y[i] = v;
}
__global__ void copy2d_kernel(float* dst, int ldd, const float* src, int lds, int rows, int cols, int add) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float v = src[(size_t)r * lds + c];
    float* d = dst + (size_t)r * ldd + c;
    *d = add ? *d + v : v;
}

}  // namespace

int xgk_lstm_fwd(hipStream_t st, const LstmFwdArgs& a) {
    hipLaunchKernelGGL(lstm_fwd_kernel, grid1((int64_t)a.B * a.R), dim3(TPB), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_lstm_bwd(hipStream_t st, const LstmBwdArgs& a) {
    hipLaunchKernelGGL(lstm_bwd_kernel, grid1((int64_t)a.B * a.R), dim3(TPB), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_lstm_bwd2(hipStream_t st, const LstmBwdArgs& a, const LstmBwdArgs& b) {
    const int64_t n = (int64_t)(a.B * a.R > b.B * b.R ? a.B * a.R : b.B * b.R);
    hipLaunchKernelGGL(lstm_bwd2_kernel, dim3((unsigned)xg_cdiv64(n, TPB), 2), dim3(TPB), 0, st, a, b);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
static bool gate_fwd_v4(const GateFwdP& p, int R) {
    return R % 4 == 0 && p.ldp % 4 == 0 && (!p.y || (p.ldt % 4 == 0 && p.ldy % 4 == 0)) && ((uintptr_t)p.pre_g % 16 == 0) &&
           ((uintptr_t)p.t % 16 == 0) && ((uintptr_t)p.y % 16 == 0);
}
static int gate_fwd_launch(hipStream_t st, const GateFwdP& a, const GateFwdP& b, int n, int rows, int R, int s_div, int s_mod, int b_div,
                           int b_mod) {
    const bool v4 = gate_fwd_v4(a, R) && (n < 2 || gate_fwd_v4(b, R));
    if (v4) { dim3 g = grid1((int64_t)rows * R / 4); g.y = n;
              hipLaunchKernelGGL((gate_fwd_kernel<4>), g, dim3(TPB), 0, st, a, b, rows, R, s_div, s_mod, b_div, b_mod); }
    else { dim3 g = grid1((int64_t)rows * R); g.y = n;
           hipLaunchKernelGGL((gate_fwd_kernel<1>), g, dim3(TPB), 0, st, a, b, rows, R, s_div, s_mod, b_div, b_mod); }
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_gate_fwd(hipStream_t st, float* pre_g, int ldp, const float* t, int ldt, int t_mod, float* y, int ldy,
                 int rows, int R, XgDrop drop, int s_div, int s_mod, int b_div, int b_mod) {
    if (!y && drop.thresh == 0u) return XG_OK;   // pure dropout with p = 0: nothing to do
    const GateFwdP a{pre_g, ldp, t, ldt, t_mod, y, ldy, drop};
    return gate_fwd_launch(st, a, a, 1, rows, R, s_div, s_mod, b_div, b_mod);
}
int xgk_gate_fwd2(hipStream_t st, float* pre_g0, float* pre_g1, int ldp, const float* t0, const float* t1, int ldt, int t_mod, float* y0,
                  float* y1, int ldy, int rows, int R, XgDrop drop0, XgDrop drop1, int s_div, int s_mod, int b_div, int b_mod) {
    const GateFwdP a{pre_g0, ldp, t0, ldt, t_mod, y0, ldy, drop0}, b{pre_g1, ldp, t1, ldt, t_mod, y1, ldy, drop1};
    return gate_fwd_launch(st, a, b, 2, rows, R, s_div, s_mod, b_div, b_mod);
}
static bool gate_bwd_v4(const GateBwdP& p, int R) {
    return R % 4 == 0 && p.lddy % 4 == 0 && p.ldg % 4 == 0 && p.ldt % 4 == 0 && (!p.dpre || p.lddp % 4 == 0) &&
           (!p.dt || p.lddt % 4 == 0) && ((uintptr_t)p.dy % 16 == 0) && ((uintptr_t)p.g % 16 == 0) && ((uintptr_t)p.t % 16 == 0) &&
           ((uintptr_t)p.dpre % 16 == 0) && ((uintptr_t)p.dt % 16 == 0);
}
static int gate_bwd_launch(hipStream_t st, const GateBwdP& a, const GateBwdP& b, int n, int rows, int R) {
    const bool v4 = gate_bwd_v4(a, R) && (n < 2 || gate_bwd_v4(b, R));
    if (v4) { dim3 g = grid1((int64_t)rows * R / 4); g.y = n; hipLaunchKernelGGL((gate_bwd_kernel<4>), g, dim3(TPB), 0, st, a, b, rows, R); }
    else { dim3 g = grid1((int64_t)rows * R); g.y = n; hipLaunchKernelGGL((gate_bwd_kernel<1>), g, dim3(TPB), 0, st, a, b, rows, R); }
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_gate_bwd(hipStream_t st, const float* dy, int lddy, const float* g, int ldg, const float* t, int ldt, int t_mod,
                 float* dpre, int lddp, float* dt, int lddt, bool dt_accumulate, int rows, int R, XgDrop drop) {
    const GateBwdP a{dy, lddy, g, ldg, t, ldt, t_mod, dpre, lddp, dt, lddt, dt_accumulate ? 1 : 0, drop};
    return gate_bwd_launch(st, a, a, 1, rows, R);
}
int xgk_gate_bwd2(hipStream_t st, const float* dy0, const float* dy1, int lddy, const float* g0, const float* g1, int ldg, const float* t0,
                  const float* t1, int ldt, float* dpre0, float* dpre1, int lddp, float* dt0, float* dt1, int lddt, int rows, int R,
                  XgDrop drop0, XgDrop drop1) {
    const GateBwdP a{dy0, lddy, g0, ldg, t0, ldt, 0, dpre0, lddp, dt0, lddt, 0, drop0}, b{dy1, lddy, g1, ldg, t1, ldt, 0, dpre1, lddp, dt1, lddt, 0, drop1};
    return gate_bwd_launch(st, a, b, 2, rows, R);
}
int xgk_relu_drop_fwd(hipStream_t st, float* x, int64_t n, XgDrop drop) {
    if (drop.thresh == 0u || n <= 0) return XG_OK;
    hipLaunchKernelGGL(relu_drop_fwd_kernel, grid1(n), dim3(TPB), 0, st, x, n, drop);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_relu_drop_bwd(hipStream_t st, float* dy, const float* y, int64_t n, XgDrop drop, const float* src) {
    if (n <= 0) return XG_OK;
    hipLaunchKernelGGL(relu_drop_bwd_kernel, grid1(n), dim3(TPB), 0, st, dy, src ? src : dy, y, n, drop);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_colsum(hipStream_t st, const float* X, int ld, int N, int Cn, float* out) {
    return colreduce<0>(st, X, ld, nullptr, 0, N, Cn, out);
}
int xgk_colsum3(hipStream_t st, const float* X, int ld, int N, int Cn, float* out, float* out2, float* out3) {
    return colreduce<0>(st, X, ld, nullptr, 0, N, Cn, out, out2, out3);
}
int xgk_colsum_prod(hipStream_t st, const float* X, int ldx, const float* Y, int ldy, int N, int Cn, float* out) {
    return colreduce<1>(st, X, ldx, Y, ldy, N, Cn, out);
}
// Train-mode BatchNorm forward of 16 columns per workgroup with the column strip held in registers: one read of Z, the exact
// two-pass statistics (mean, then the sum of squared deviations) in a fixed summation order, the running statistics' update and
// -- when X is given -- the normalised, ReLU'd, dropped and row-masked output, all in ONE launch.  The column reductions it
// replaces took six launches (2 fills, 2 reductions with atomics, 2 scalings) + bn_running + bn_apply: ~110 us of launch chain in
// front of the encoder's recurrence at N = 3328.  Four lanes cover a row's 64 bytes (whole sectors; a first form with 4 columns
// per workgroup read 16 bytes per row and took 47 us).
template <int NV>
__global__ void __launch_bounds__(1024) bn_train_fwd_kernel(const float* __restrict__ Z, int N, int R, float* __restrict__ mean,
                                                              float* __restrict__ var, float* rmean, float* rvar, float mom,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ rowmask, float* __restrict__ X, float eps,
                                                              XgDrop drop) {
    __shared__ float4 red[16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rl = threadIdx.x >> 2, cq = threadIdx.x & 3, c = blockIdx.x * 16 + 4 * cq;
    float4 x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int r = rl + 256 * i;
        x[i] = r < N ? *reinterpret_cast<const float4*>(Z + (size_t)r * R + c) : float4{0.f, 0.f, 0.f, 0.f};
    }
    auto colsum = [&](float4 v) -> float4 {          // over all threads of this column quad
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
            v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
        }
        __syncthreads();                             // (red is reused by the second call)
        if (lane < 4) red[wave][lane] = v;           // lane == cq here
        __syncthreads();
        float4 t = red[0][cq];
#pragma unroll
        for (int w = 1; w < 16; ++w) { const float4 u = red[w][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        return t;
    };
    float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) { s.x += x[i].x; s.y += x[i].y; s.z += x[i].z; s.w += x[i].w; }
    s = colsum(s);
    const float inv = 1.0f / (float)N;
    const float4 mu = {s.x * inv, s.y * inv, s.z * inv, s.w * inv};
    float4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (rl + 256 * i < N) {
            const float dx = x[i].x - mu.x, dy = x[i].y - mu.y, dz = x[i].z - mu.z, dw = x[i].w - mu.w;
            q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
        }
    }
    q = colsum(q);
    const float4 vb = {q.x * inv, q.y * inv, q.z * inv, q.w * inv};
    if (rl == 0) {
        *reinterpret_cast<float4*>(mean + c) = mu;
        *reinterpret_cast<float4*>(var + c) = vb;
        if (rmean) {                                 // the running statistics' update (bn_running_kernel's arithmetic)
            const float ub = (float)N / (float)max(N - 1, 1);
            const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, v4[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rmean[c + j] = (1.f - mom) * rmean[c + j] + mom * m4[j];
                rvar[c + j] = (1.f - mom) * rvar[c + j] + mom * (v4[j] * ub);
            }
        }
    }
    if (!X) return;
    // the output of the layer (bn_apply_kernel's arithmetic, element for element)
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    const float sd[4] = {sqrtf(vb.x + eps), sqrtf(vb.y + eps), sqrtf(vb.z + eps), sqrtf(vb.w + eps)};
    const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int r = rl + 256 * i;
        if (r < N) {
            const float rm = rowmask[r];
            const float z4[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
            float o[4];
            const size_t idx = (size_t)r * R + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (z4[j] - m4[j]) / sd[j];
                o[j] = fmaxf(xh * g4[j] + b4[j], 0.f) * xg_keep(drop, (uint32_t)(idx + j)) * rm;
            }
            *reinterpret_cast<float4*>(X + idx) = float4{o[0], o[1], o[2], o[3]};
        }
    }
}

// returns 1 when the shape is not this kernel's kind (the caller takes xgk_bn_stats + xgk_bn_apply)
int xgk_bn_train_fwd(hipStream_t st, const float* Z, int N, int R, float* mean, float* var, float* rmean, float* rvar, float momentum,
                     const float* gamma, const float* beta, const float* rowmask, float* X, float eps, XgDrop drop) {
    if (rmean && !rvar) return XG_EINVAL;
    const bool al = ((uintptr_t)Z % 16 == 0) && ((uintptr_t)mean % 16 == 0) && ((uintptr_t)var % 16 == 0) &&
                    (!X || (((uintptr_t)X % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0) && rowmask));
    if (R % 16 != 0 || N < 1 || N > 20 * 256 || !al) return 1;      // (20 x 16 bytes per thread: the register budget of 16 waves per CU)
    if (N <= 16 * 256) hipLaunchKernelGGL((bn_train_fwd_kernel<16>), dim3(R / 16), dim3(1024), 0, st, Z, N, R, mean, var, rmean, rvar, momentum,
                                          gamma, beta, rowmask, X, eps, drop);
    else hipLaunchKernelGGL((bn_train_fwd_kernel<20>), dim3(R / 16), dim3(1024), 0, st, Z, N, R, mean, var, rmean, rvar, momentum,
                            gamma, beta, rowmask, X, eps, drop);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

int xgk_bn_stats(hipStream_t st, const float* Z, int N, int R, float* mean, float* var, float* rmean, float* rvar, float momentum) {
    if (rmean && !rvar) return XG_EINVAL;
    {
        XgDrop nodrop{};
        const int rc = xgk_bn_train_fwd(st, Z, N, R, mean, var, rmean, rvar, momentum, nullptr, nullptr, nullptr, nullptr, 0.f, nodrop);
        if (rc != 1) return rc;
    }
    if (hipMemsetAsync(mean, 0, sizeof(float) * R, st) != hipSuccess) return XG_EHIP;
    if (hipMemsetAsync(var, 0, sizeof(float) * R, st) != hipSuccess) return XG_EHIP;
    XG_TRY(colreduce<0>(st, Z, R, nullptr, 0, N, R, mean));
    hipLaunchKernelGGL(scale_kernel, grid1(R), dim3(TPB), 0, st, mean, 1.0f / (float)N, R);
    XG_TRY(colreduce<2>(st, Z, R, mean, 0, N, R, var));
    hipLaunchKernelGGL(scale_kernel, grid1(R), dim3(TPB), 0, st, var, 1.0f / (float)N, R);
    XG_CHECK_LAUNCH();
    if (rmean) return xgk_bn_running(st, mean, var, rmean, rvar, N, R, momentum);
    return XG_OK;
}
int xgk_bn_running(hipStream_t st, const float* mean, const float* var, float* rmean, float* rvar, int N, int R,
                   float momentum) {
    hipLaunchKernelGGL(bn_running_kernel, grid1(R), dim3(TPB), 0, st, mean, var, rmean, rvar, N, R, momentum);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_bn_apply(hipStream_t st, const float* Z, const float* mean, const float* var, const float* gamma,
                 const float* beta, const float* rowmask, float* X, int N, int R, float eps, XgDrop drop) {
    const bool v4 = R % 4 == 0 && ((uintptr_t)Z % 16 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)mean % 16 == 0) &&
                    ((uintptr_t)var % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0);
    if (v4) hipLaunchKernelGGL((bn_apply_kernel<4>), grid1((int64_t)N * R / 4), dim3(TPB), 0, st, Z, mean, var, gamma, beta,
                               rowmask, X, N, R, eps, drop);
    else hipLaunchKernelGGL((bn_apply_kernel<1>), grid1((int64_t)N * R), dim3(TPB), 0, st, Z, mean, var, gamma, beta, rowmask, X,
                            N, R, eps, drop);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_bn_bwd_reduce(hipStream_t st, float* dX, const float* X, const float* Z, const float* mean, const float* var,
                      const float* rowmask, int N, int R, float eps, XgDrop drop, float* sum_dy, float* sum_dyxhat) {
    const int rpc = 64;
    dim3 grid(xg_cdiv(R, 64), xg_cdiv(N, rpc));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, st, dX, X, Z, mean, var, rowmask, N, R, eps, drop,
                       sum_dy, sum_dyxhat, rpc);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_bn_bwd_apply(hipStream_t st, float* dY, const float* Z, const float* mean, const float* var,
                     const float* gamma, const float* sum_dy, const float* sum_dyxhat, int N, int R, float eps,
                     bool train, float* g_beta, float* g_gamma) {
    if ((g_beta == nullptr) != (g_gamma == nullptr)) return XG_EINVAL;
    const bool v4 = R % 4 == 0 && ((uintptr_t)dY % 16 == 0) && ((uintptr_t)Z % 16 == 0) && ((uintptr_t)mean % 16 == 0) &&
                    ((uintptr_t)var % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)sum_dy % 16 == 0) &&
                    ((uintptr_t)sum_dyxhat % 16 == 0);
    if (v4) hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), grid1((int64_t)N * R / 4), dim3(TPB), 0, st, dY, Z, mean, var, gamma,
                               sum_dy, sum_dyxhat, N, R, eps, train ? 1 : 0, g_beta, g_gamma);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), grid1((int64_t)N * R), dim3(TPB), 0, st, dY, Z, mean, var, gamma, sum_dy,
                            sum_dyxhat, N, R, eps, train ? 1 : 0, g_beta, g_gamma);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_embed_gather(hipStream_t st, const float* table, int E, const int64_t* tok, int inner, int64_t s_inner,
                     int64_t s_outer, int n, int V, float* out, int ldo) {
    if (n <= 0) return XG_OK;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(n), dim3(128), 0, st, table, E, tok, inner, s_inner, s_outer, V, out, ldo);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_step_prep(hipStream_t st, const float* table, int E, const int64_t* tok, int V, float* xt, int B,
                  const float* state, float* state_copy, int64_t nstate) {
    if (((uintptr_t)state % 16) || ((uintptr_t)state_copy % 16)) return XG_EINVAL;
    const int nb = B + (int)xg_cdiv64(nstate, 4 * 128);
    hipLaunchKernelGGL(step_prep_kernel, dim3(nb), dim3(128), 0, st, table, E, tok, V, xt, B, state, state_copy, nstate);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_compact(hipStream_t st, CompactArgs& a) {
    if (a.n <= 0 || a.n > XG_COMPACT_MAX) return XG_EINVAL;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
        const CompactEntry& e = a.e[i];
        if (!e.dst || !e.src || e.dst_block <= 0 || e.total < 0 || ((uintptr_t)e.dst % 16) || ((uintptr_t)e.src % 16)) return XG_EINVAL;
        a.start[i] = blocks;
        blocks += (int)xg_cdiv64(e.total, CPB);
    }
    a.start[a.n] = blocks;
    if (blocks == 0) return XG_OK;
    hipLaunchKernelGGL(compact_kernel, dim3(blocks), dim3(256), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_embed_scatter_add(hipStream_t st, float* dtable, int E, const int64_t* tok, int inner, int64_t s_inner,
                          int64_t s_outer, int n, int V, const float* dX, int ldx) {
    if (n <= 0) return XG_OK;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(n), dim3(128), 0, st, dtable, E, tok, inner, s_inner, s_outer, V, dX, ldx);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_masked_mean(hipStream_t st, const float* V, const float* mask, float* out, int B, int K, int R) {
    hipLaunchKernelGGL(masked_mean_kernel, grid1((int64_t)B * R), dim3(TPB), 0, st, V, mask, out, B, K, R);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_axpy(hipStream_t st, float* y, const float* x, float a, int64_t n) {
    if (n <= 0) return XG_OK;
    hipLaunchKernelGGL(axpy_kernel, grid1(n), dim3(TPB), 0, st, y, x, a, n);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_fill(hipStream_t st, float* y, float v, int64_t n) {
    if (n <= 0) return XG_OK;
    hipLaunchKernelGGL(fill_kernel, grid1(n), dim3(TPB), 0, st, y, v, n);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_copy2d(hipStream_t st, float* dst, int ldd, const float* src, int lds, int rows, int cols, bool add) {
    if (rows <= 0 || cols <= 0) return XG_OK;
    hipLaunchKernelGGL(copy2d_kernel, grid1((int64_t)rows * cols), dim3(TPB), 0, st, dst, ldd, src, lds, rows, cols,
                       add ? 1 : 0);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
