// Temporal soft attention of LSTMCore_two_layer_gate (reference caption_src/sub_modules.py:677-680)
// with the loop-invariant projection q = v2a(V) hoisted out of the time loop.
//
//   e_k   = w_a . tanh(p + q_k)      (a2w bias cancels in the softmax)
//   alpha = softmax_k(e)             over ALL K frames, unmasked
//   af    = sum_k alpha_k V_k
//
// One workgroup per video: the K*A block of q (156 KB at K=26, A=1536) is streamed once per step
// with 16-B coalesced loads (one wave per frame row), scores are reduced with wave64 shuffles, the
// softmax over K <= 1024 frames lives in LDS, and the context is a coalesced pass over the video's
// (K,R) tile of V.  HBM/L2-bound: 4*(K*A + K*R + A + R + 2K) bytes per video per step.
#include "xg_common.h"
#include "xg_kernels.h"
#include <cstdlib>

namespace {

constexpr int ATPB = 512;  // 8 waves per video

__global__ void __launch_bounds__(ATPB) attn_fwd_kernel(const float* __restrict__ p, const float* __restrict__ vproj,
                                                          const float* __restrict__ V, const float* __restrict__ w,
                                                          float* __restrict__ alpha, float* __restrict__ af, int K,
                                                          int R, int A) {
    extern __shared__ float sm[];          // e[K]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ATPB / 64;
    const float* pb = p + (size_t)b * A;
    const float* qb = vproj + (size_t)b * K * A;
    for (int k = wave; k < K; k += nw) {
        const float* qk = qb + (size_t)k * A;
        float acc = 0.f;
        if ((A & 3) == 0) {
            for (int a = lane * 4; a < A; a += 256) {
                const float4 q4 = *reinterpret_cast<const float4*>(qk + a);
                const float4 p4 = *reinterpret_cast<const float4*>(pb + a);
                const float4 w4 = *reinterpret_cast<const float4*>(w + a);
                acc += w4.x * xg_tanh(p4.x + q4.x) + w4.y * xg_tanh(p4.y + q4.y) + w4.z * xg_tanh(p4.z + q4.z) +
                       w4.w * xg_tanh(p4.w + q4.w);
            }
        } else {
            for (int a = lane; a < A; a += 64) acc += w[a] * xg_tanh(pb[a] + qk[a]);
        }
        acc = wave_sum(acc);
        if (lane == 0) sm[k] = acc;
    }
    __syncthreads();
    // softmax over K (every thread redundantly scans the K scores: K is tens, not thousands)
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, sm[k]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(sm[k] - mx);
    const float inv = 1.0f / den;
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += ATPB) {
        const float al = expf(sm[k] - mx) * inv;
        sm[k] = al;
        if (alpha) alpha[(size_t)b * K + k] = al;
    }
    __syncthreads();
    const float* Vb = V + (size_t)b * K * R;
    for (int r = threadIdx.x; r < R; r += ATPB) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += sm[k] * Vb[(size_t)k * R + r];
        af[(size_t)b * R + r] = s;
    }
}

// de_k = alpha_k (dalpha_k - sum_j alpha_j dalpha_j), dalpha_k = daf . V_k ; dp_a = sum_k de_k w_a (1 - th^2)
__global__ void __launch_bounds__(ATPB) attn_bwd_kernel(const float* __restrict__ daf, int lddaf,
                                                          const float* __restrict__ p, const float* __restrict__ vproj,
                                                          const float* __restrict__ V, const float* __restrict__ w,
                                                          const float* __restrict__ alpha, float* __restrict__ de,
                                                          float* __restrict__ dp, int K, int R, int A) {
    extern __shared__ float sm[];          // dalpha[K] then de[K]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ATPB / 64;
    const float* Vb = V + (size_t)b * K * R;
    const float* dafb = daf + (size_t)b * lddaf;
    for (int k = wave; k < K; k += nw) {
        float acc = 0.f;
        for (int r = lane; r < R; r += 64) acc += dafb[r] * Vb[(size_t)k * R + r];
        acc = wave_sum(acc);
        if (lane == 0) sm[k] = acc;
    }
    __syncthreads();
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += alpha[(size_t)b * K + k] * sm[k];
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += ATPB) {
        const float d = alpha[(size_t)b * K + k] * (sm[k] - dot);
        sm[k] = d;
        de[(size_t)b * K + k] = d;
    }
    __syncthreads();
    const float* pb = p + (size_t)b * A;
    const float* qb = vproj + (size_t)b * K * A;
    for (int a = threadIdx.x; a < A; a += ATPB) {
        const float pa = pb[a];
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const float th = xg_tanh(pa + qb[(size_t)k * A + a]);
            s += sm[k] * (1.0f - th * th);
        }
        dp[(size_t)b * A + a] = s * w[a];
    }
}

// After the time loop (one pass over q instead of a read-modify-write of dq per step):
//   dq[b,k,a] = w_a sum_t de_t[b,k] (1 - th_t^2),  dw[a] += sum_{t,b,k} de_t[b,k] th_t,  th_t = tanh(p_t[b,a] + q[b,k,a])
// grid (ceil(A/256), B); thread = one a, loops k and t.
template <int TMAX>
__device__ __forceinline__ void attn_bwd_post_body(const float* __restrict__ P, const float* __restrict__ vproj,
                                                   const float* __restrict__ w, const float* __restrict__ DE,
                                                   float* __restrict__ dvproj, float* __restrict__ dw, int T,
                                                   int B, int K, int A, int bx) {
    extern __shared__ float sde[];         // DE[:, b, :]  (T*K)
    const int b = blockIdx.y;
    const int a = bx * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < T * K; i += 256) sde[i] = DE[((size_t)(i / K) * B + b) * K + (i % K)];
    __syncthreads();
    if (a >= A) return;
    const float wa = w[a];
    // this thread's column of p for every step lives in registers (TMAX > 0) so the inner loop is pure VALU
    float pr[TMAX > 0 ? TMAX : 1];
    if (TMAX > 0) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) pr[t] = t < T ? P[((size_t)t * B + b) * A + a] : 0.f;
    }
    float dwa = 0.f;
    for (int k = 0; k < K; ++k) {
        const float q = vproj[((size_t)b * K + k) * A + a];
        float acc = 0.f;
        if (TMAX > 0) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                const float d = t < T ? sde[t * K + k] : 0.f;
                const float th = xg_tanh(pr[t] + q);
                acc += d * (1.0f - th * th);
                dwa += d * th;
            }
        } else {
            for (int t = 0; t < T; ++t) {
                const float d = sde[t * K + k];
                if (d != 0.f) {
                    const float th = xg_tanh(P[((size_t)t * B + b) * A + a] + q);
                    acc += d * (1.0f - th * th);
                    dwa += d * th;
                }
            }
        }
        dvproj[((size_t)b * K + k) * A + a] = acc * wa;
    }
    atomicAdd(dw + a, dwa);
}
template <int TMAX>
__global__ void __launch_bounds__(256) attn_bwd_post_kernel(const float* __restrict__ P, const float* __restrict__ vproj,
                                                              const float* __restrict__ w, const float* __restrict__ DE,
                                                              float* __restrict__ dvproj, float* __restrict__ dw, int T,
                                                              int B, int K, int A) {
    attn_bwd_post_body<TMAX>(P, vproj, w, DE, dvproj, dw, T, B, K, A, (int)blockIdx.x);
}

// dV[b,k,r] (+)= sum_t ALPHA[t,b,k] * DAF[t,b,r]
__global__ void attn_dV_kernel(const float* __restrict__ ALPHA, const float* __restrict__ DAF, int lddaf,
                               int64_t tstride, float* __restrict__ dV, int T, int B, int K, int R, int acc) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * K * R) return;
    const int r = (int)(idx % R);
    const int k = (int)((idx / R) % K);
    const int b = (int)(idx / ((int64_t)R * K));
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += ALPHA[((size_t)t * B + b) * K + k] * DAF[(size_t)t * tstride + (size_t)b * lddaf + r];
    dV[idx] = acc ? dV[idx] + s : s;
}

// The same with the step axis in registers (T <= TMAX): thread = one r of one video, its T values of daf are read ONCE
// (the kernel above re-reads them for each of the K frames: 143 MB through L2 at config 2, 84 us on the path between the
// reverse-time loop and the encoder backward); alpha[:, b, :] sits in LDS.  grid (ceil(R / 256), B).
template <int TMAX>
__device__ __forceinline__ void attn_dV_reg_body(const float* __restrict__ ALPHA, const float* __restrict__ DAF,
                                                 int lddaf, int64_t tstride, float* __restrict__ dV, int T, int B,
                                                 int K, int R, int acc, int bx) {
    const int b = blockIdx.y;
    const int r = bx * 256 + threadIdx.x;
    float d[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) d[t] = (t < T && r < R) ? DAF[(size_t)t * tstride + (size_t)b * lddaf + r] : 0.f;
    if (r >= R) return;
    // alpha[t, b, k] is the same for the whole workgroup: wave-uniform addresses, i.e. scalar loads (eight consecutive frames
    // per step) and a scalar operand of the fma -- no LDS.  Eight frames at a time; when accumulating, the eight old values
    // are requested together (a load -> add -> store loop per frame cost a memory round trip each: 87 us for 26 frames)
    const float* al = ALPHA + (size_t)b * K;
    for (int k0 = 0; k0 < K; k0 += 8) {
        float s[8], old[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            old[j] = acc ? dV[((size_t)b * K + min(k0 + j, K - 1)) * R + r] : 0.f;
            s[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T) {
                const float* at = al + (size_t)t * B * K;
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += at[min(k0 + j, K - 1)] * d[t];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < K) dV[((size_t)b * K + k0 + j) * R + r] = old[j] + s[j];
    }
}
template <int TMAX>
__global__ void __launch_bounds__(256) attn_dV_reg_kernel(const float* __restrict__ ALPHA, const float* __restrict__ DAF,
                                                            int lddaf, int64_t tstride, float* __restrict__ dV, int T, int B,
                                                            int K, int R, int acc) {
    attn_dV_reg_body<TMAX>(ALPHA, DAF, lddaf, tstride, dV, T, B, K, R, acc, (int)blockIdx.x);
}
// Both passes that follow the reverse-time loop as ONE launch (round 4): dV = sum_t alpha_t dAF_t (latency-bound: 256 workgroups,
// 35 us alone) in the shadow of dq / dw (VALU-bound: 768 workgroups of tanh, 67 us): grid (nxa + nxr, B), the first nxa blocks of
// a row take the attention columns, the rest the context columns.  Same arithmetic as the two kernels above.
template <int TMAX>
__global__ void __launch_bounds__(256) attn_post_dV_kernel(const float* __restrict__ P, const float* __restrict__ vproj,
                                                             const float* __restrict__ w, const float* __restrict__ DE,
                                                             float* __restrict__ dvproj, float* __restrict__ dw,
                                                             const float* __restrict__ ALPHA, const float* __restrict__ DAF, int lddaf,
                                                             int64_t tstride, float* __restrict__ dV, int T, int B, int K, int A,
                                                             int R, int nxa) {
    if ((int)blockIdx.x < nxa) attn_bwd_post_body<TMAX>(P, vproj, w, DE, dvproj, dw, T, B, K, A, (int)blockIdx.x);
    else attn_dV_reg_body<TMAX>(ALPHA, DAF, lddaf, tstride, dV, T, B, K, R, 0, (int)blockIdx.x - nxa);
}


// ------------------------------------------------------------------------------------------------
// Fast paths (A % 4 == 0, A <= 2048, R % 4 == 0): one 1024-thread workgroup per video, every global
// load issued up front so the kernel pays ONE memory round trip: each wave keeps its slice of p and
// w_a in registers and streams whole q rows (lane = 16 B, wave = 1 KB per instruction), the V tile of
// the context pass is already in flight while the scores are computed.
constexpr int FT = 1024, FW = FT / 64;
#ifdef XG_ATTN_TRACE
__device__ long long attn_trace_buf[1024 * 8];
#define AT_STAMP(i) do { if (threadIdx.x == 0) attn_trace_buf[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define AT_STAMP(i) do {} while (0)
#endif

// HALF: the same kernel at <= 64 VGPRs (8 waves per SIMD), so that its 16 waves fit into the half of a CU a background GEMM
// workgroup leaves free (xg_gemm.hip: XGK_GEMM_BG) -- the 128-VGPR form needs an EMPTY CU.  Two waves share a frame, each
// taking half of the column groups (half the p / w / q registers); their partial scores meet in LDS.
template <int NI, int VMAX, bool HALF>      // NI: 256-column groups of A per lane; VMAX: V rows held per thread (K <= VMAX * nkp)
__global__ void __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(HALF ? 8 : 4, HALF ? 8 : 4)))
attn_fwd_fast(const float* __restrict__ p, const float* __restrict__ vproj, const float* __restrict__ V,
              const float* __restrict__ w, float* __restrict__ alpha, float* __restrict__ af, int K, int R, int A) {
    extern __shared__ float sm[];                     // e[K] (HALF: two partials) | part[nkp][R]
    XG_CHAIN_PRIO();
    AT_STAMP(0);
    float* se = sm;
    float* part = sm + (HALF ? 2 : 1) * ((K + 3) & ~3);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pb = p + (size_t)b * A;
    const float* qb = vproj + (size_t)b * K * A;
    const float* Vb = V + (size_t)b * K * R;
    // context pass mapping: thread -> (4 consecutive r, k-part); its V loads go out first
    const int r4n = R >> 2, nkp = FT / r4n > 0 ? min(FT / r4n, K) : 1;
    const int myr = (tid % r4n) << 2, mykp = tid / r4n;
    float4 vreg[VMAX];
#pragma unroll
    for (int j = 0; j < VMAX; ++j) {
        const int k = mykp + j * nkp;
        vreg[j] = (mykp < nkp && k < K) ? *reinterpret_cast<const float4*>(Vb + (size_t)k * R + myr) : make_float4(0, 0, 0, 0);
    }
    constexpr int NIW = HALF ? NI / 2 : NI;           // column groups per wave
    const int g0 = HALF ? (wave & 1) * NIW : 0;       // ... starting at
    float4 pr[NIW], wr[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int a = lane * 4 + 256 * (g0 + i);
        const bool ok = a < A;
        pr[i] = ok ? *reinterpret_cast<const float4*>(pb + a) : make_float4(0, 0, 0, 0);
        wr[i] = ok ? *reinterpret_cast<const float4*>(w + a) : make_float4(0, 0, 0, 0);
    }
    AT_STAMP(1);
    for (int k = HALF ? (wave >> 1) : wave; k < K; k += HALF ? FW / 2 : FW) {
        const float* qk = qb + (size_t)k * A;
        float4 q[NIW];
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int a = lane * 4 + 256 * (g0 + i);
            q[i] = a < A ? *reinterpret_cast<const float4*>(qk + a) : make_float4(0, 0, 0, 0);
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            acc += wr[i].x * xg_tanh(pr[i].x + q[i].x) + wr[i].y * xg_tanh(pr[i].y + q[i].y) +
                   wr[i].z * xg_tanh(pr[i].z + q[i].z) + wr[i].w * xg_tanh(pr[i].w + q[i].w);
        acc = wave_sum(acc);
        if (lane == 0) se[(HALF ? (wave & 1) * ((K + 3) & ~3) : 0) + k] = acc;
    }
    AT_STAMP(2);
    __syncthreads();
    AT_STAMP(3);
    float4 s4 = make_float4(0, 0, 0, 0);
    if (K <= 64) {
        // softmax over the K scores, once per wave with lane k holding e_k (not K serial expf per thread)
        const float e = lane < K ? (HALF ? se[lane] + se[((K + 3) & ~3) + lane] : se[lane]) : -INFINITY;
        const float mx = wave_max(e);
        const float ex = lane < K ? expf(e - mx) : 0.f;
        const float al_lane = ex * (1.0f / wave_sum(ex));
        if (alpha && wave == 0 && lane < K) alpha[(size_t)b * K + lane] = al_lane;
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const int k = mykp + j * nkp;
            const float al = __shfl(al_lane, k < K ? k : 0);
            if (mykp < nkp && k < K) { s4.x += al * vreg[j].x; s4.y += al * vreg[j].y; s4.z += al * vreg[j].z; s4.w += al * vreg[j].w; }
        }
    } else {
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, se[k]);
        float den = 0.f;
        for (int k = 0; k < K; ++k) den += expf(se[k] - mx);
        const float inv = 1.0f / den;
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const int k = mykp + j * nkp;
            if (mykp < nkp && k < K) {
                const float al = expf(se[k] - mx) * inv;
                s4.x += al * vreg[j].x; s4.y += al * vreg[j].y; s4.z += al * vreg[j].z; s4.w += al * vreg[j].w;
            }
        }
        if (alpha && tid < K) alpha[(size_t)b * K + tid] = expf(se[tid] - mx) * inv;
    }
    if (mykp < nkp) *reinterpret_cast<float4*>(part + (size_t)mykp * R + myr) = s4;
    AT_STAMP(4);
    __syncthreads();
    for (int r = tid; r < R; r += FT) {
        float s = 0.f;
        for (int j = 0; j < nkp; ++j) s += part[(size_t)j * R + r];
        af[(size_t)b * R + r] = s;
    }
    AT_STAMP(5);
}

// backward: de_k = alpha_k (dalpha_k - sum_j alpha_j dalpha_j), dalpha_k = daf . V_k ; dp_a = w_a sum_k de_k (1 - th^2)
template <int NQ>   // q values per thread per row: thread owns a = tid*2 (+1), rows looped; NQ = ceil(K / 1) held rows
__global__ void __launch_bounds__(FT) attn_bwd_fast(const float* __restrict__ daf, int lddaf, const float* __restrict__ p,
                                                     const float* __restrict__ vproj, const float* __restrict__ V,
                                                     const float* __restrict__ w, const float* __restrict__ alpha,
                                                     float* __restrict__ de, float* __restrict__ dp, int K, int R, int A) {
    extern __shared__ float sm[];                     // dalpha[K]
    XG_CHAIN_PRIO();
    AT_STAMP(0);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qb = vproj + (size_t)b * K * A;
    const float* Vb = V + (size_t)b * K * R;
    const float* dafb = daf + (size_t)b * lddaf;
    // first needed, first requested: the dalpha operands of this wave's rows (k = wave, wave + 16, ...; R <= 1024 here)
    constexpr int DR = 2, DC = 4;                     // rows per wave, float4 chunks per row held in registers
    float4 dv[DC], vv[DR][DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        const int r = lane * 4 + 256 * c;
        dv[c] = r < R ? *reinterpret_cast<const float4*>(dafb + r) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DR; ++j) {
            const int k = wave + j * FW;
            vv[j][c] = (r < R && k < K) ? *reinterpret_cast<const float4*>(Vb + (size_t)k * R + r) : make_float4(0, 0, 0, 0);
        }
    }
    const float al_lane = lane < K ? alpha[(size_t)b * K + lane] : 0.f;
    // thread -> two consecutive attention columns; all K rows of q for them go to registers up front
    const int a0 = tid * 2;
    const bool a_ok = a0 < A;
    float2 q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = (a_ok && k < K) ? *reinterpret_cast<const float2*>(qb + (size_t)k * A + a0) : make_float2(0, 0);
    const float2 pa = a_ok ? *reinterpret_cast<const float2*>(p + (size_t)b * A + a0) : make_float2(0, 0);
    const float2 wa = a_ok ? *reinterpret_cast<const float2*>(w + a0) : make_float2(0, 0);
    AT_STAMP(1);
#pragma unroll
    for (int j = 0; j < DR; ++j) {
        const int k = wave + j * FW;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c)
            acc += dv[c].x * vv[j][c].x + dv[c].y * vv[j][c].y + dv[c].z * vv[j][c].z + dv[c].w * vv[j][c].w;
        acc = wave_sum(acc);
        if (lane == 0 && k < K) sm[k] = acc;
    }
    for (int k = wave + DR * FW; k < K; k += FW) {    // K > 32: remaining rows the slow way
        float acc = 0.f;
        for (int r = lane * 4; r < R; r += 256) {
            const float4 d4 = *reinterpret_cast<const float4*>(dafb + r);
            const float4 v4 = *reinterpret_cast<const float4*>(Vb + (size_t)k * R + r);
            acc += d4.x * v4.x + d4.y * v4.y + d4.z * v4.z + d4.w * v4.w;
        }
        acc = wave_sum(acc);
        if (lane == 0) sm[k] = acc;
    }
    AT_STAMP(2);
    __syncthreads();
    AT_STAMP(3);
    // softmax backward once per wave, lane k holding row k:  de_k = alpha_k (dalpha_k - sum_j alpha_j dalpha_j)
    const float da = lane < K ? sm[lane] : 0.f;
    const float dot = wave_sum(al_lane * da);
    const float d_lane = al_lane * (da - dot);
    if (wave == 0 && lane < K) de[(size_t)b * K + lane] = d_lane;
    AT_STAMP(4);
    if (a_ok) {
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            if (k < K) {
                const float dk = xg_readlane(d_lane, k);
                const float tx = xg_tanh(pa.x + q[k].x), ty = xg_tanh(pa.y + q[k].y);
                sx += dk * (1.0f - tx * tx);
                sy += dk * (1.0f - ty * ty);
            }
        }
        *reinterpret_cast<float2*>(dp + (size_t)b * A + a0) = make_float2(sx * wa.x, sy * wa.y);
    }
    AT_STAMP(5);
}

// The same backward as TWO workgroups per video (512 threads each): every thread's work is dominated by its 2 x K tanh
// (one 1024-thread workgroup per video keeps 128 CUs busy for ~5 us of VALU each and leaves the other 128 idle), so the
// attention columns are halved between two CUs; both compute the K dalpha dot products (V is read twice: 53 KB per video)
// and the softmax backward redundantly, part 0 writes de.  K <= NQ <= 48, A / 2 even and <= 1024.
template <int NQ>
__global__ void __launch_bounds__(512, 4) attn_bwd_split(const float* __restrict__ daf, int lddaf, const float* __restrict__ p,
                                                      const float* __restrict__ vproj, const float* __restrict__ V,
                                                      const float* __restrict__ w, const float* __restrict__ alpha,
                                                      float* __restrict__ de, float* __restrict__ dp, int K, int R, int A) {
    extern __shared__ float sm[];                     // dalpha[K]
    XG_CHAIN_PRIO();
    constexpr int NWV = 8;
    const int b = blockIdx.x >> 1, part = blockIdx.x & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qb = vproj + (size_t)b * K * A;
    const float* Vb = V + (size_t)b * K * R;
    const float* dafb = daf + (size_t)b * lddaf;
    constexpr int DR = (NQ + NWV - 1) / NWV, DC = 4;  // rows per wave (K <= NQ), float4 chunks per row held in registers
    float4 dv[DC], vv[DR][DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        const int r = lane * 4 + 256 * c;
        dv[c] = r < R ? *reinterpret_cast<const float4*>(dafb + r) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DR; ++j) {
            const int k = wave + j * NWV;
            vv[j][c] = (r < R && k < K) ? *reinterpret_cast<const float4*>(Vb + (size_t)k * R + r) : make_float4(0, 0, 0, 0);
        }
    }
    const float al_lane = lane < K ? alpha[(size_t)b * K + lane] : 0.f;
    const int ah = A >> 1;                            // this workgroup's columns: [part * ah, part * ah + ah)
    const int a0 = part * ah + tid * 2;
    const bool a_ok = tid * 2 < ah;
    // 4 waves per SIMD (128 VGPRs): an 8-wave workgroup then fits into HALF a CU, beside a background GEMM workgroup
    // (xg_gemm.hip: XGK_GEMM_BG).  The first half of the q rows is requested up front with the dalpha operands, the second
    // half when those registers are free again (behind the dot products, its latency under the first half's tanh)
    constexpr int NQ1 = (NQ + 1) / 2;
    float2 q[NQ];
#pragma unroll
    for (int k = 0; k < NQ1; ++k) q[k] = (a_ok && k < K) ? *reinterpret_cast<const float2*>(qb + (size_t)k * A + a0) : make_float2(0, 0);
    const float2 pa = a_ok ? *reinterpret_cast<const float2*>(p + (size_t)b * A + a0) : make_float2(0, 0);
    const float2 wa = a_ok ? *reinterpret_cast<const float2*>(w + a0) : make_float2(0, 0);
#pragma unroll
    for (int j = 0; j < DR; ++j) {
        const int k = wave + j * NWV;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c)
            acc += dv[c].x * vv[j][c].x + dv[c].y * vv[j][c].y + dv[c].z * vv[j][c].z + dv[c].w * vv[j][c].w;
        acc = wave_sum(acc);
        if (lane == 0 && k < K) sm[k] = acc;
    }
#pragma unroll
    for (int k = NQ1; k < NQ; ++k) q[k] = (a_ok && k < K) ? *reinterpret_cast<const float2*>(qb + (size_t)k * A + a0) : make_float2(0, 0);
    __syncthreads();
    const float da = lane < K ? sm[lane] : 0.f;
    const float dot = wave_sum(al_lane * da);
    const float d_lane = al_lane * (da - dot);
    if (part == 0 && wave == 0 && lane < K) de[(size_t)b * K + lane] = d_lane;
    if (a_ok) {
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            if (k < K) {
                const float dk = xg_readlane(d_lane, k);
                const float tx = xg_tanh(pa.x + q[k].x), ty = xg_tanh(pa.y + q[k].y);
                sx += dk * (1.0f - tx * tx);
                sy += dk * (1.0f - ty * ty);
            }
        }
        *reinterpret_cast<float2*>(dp + (size_t)b * A + a0) = make_float2(sx * wa.x, sy * wa.y);
    }
}

// The two-workgroup backward for 33..48 frames at <= 128 VGPRs (the form above needs 221 there: two waves per SIMD, i.e. an
// all but empty CU, and waited 140 us per step beside the bf16 head products of the hidden-1024 configuration).  Same
// arithmetic; the operands arrive in stages instead of all up front: the V rows of the dalpha dots three at a time, the q rows
// in three thirds, each third requested while the previous one is consumed.  Also faster alone (median 21.6 vs 27.2 us per
// step in situ: four waves per SIMD hide the staged loads); hidden-1024 iteration 8.35 -> 8.20 ms.
template <int NQ>
__global__ void __launch_bounds__(512, 4) attn_bwd_split_lr(const float* __restrict__ daf, int lddaf, const float* __restrict__ p,
                                                            const float* __restrict__ vproj, const float* __restrict__ V,
                                                            const float* __restrict__ w, const float* __restrict__ alpha,
                                                            float* __restrict__ de, float* __restrict__ dp, int K, int R, int A) {
    extern __shared__ float sm[];                     // dalpha[K]
    XG_CHAIN_PRIO();
    constexpr int NWV = 8, DC = 4, DR = (NQ + NWV - 1) / NWV, DB = 3, NT = (NQ + 2) / 3;
    const int b = blockIdx.x >> 1, part = blockIdx.x & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qb = vproj + (size_t)b * K * A;
    const float* Vb = V + (size_t)b * K * R;
    const float* dafb = daf + (size_t)b * lddaf;
    const int ah = A >> 1;
    const int a0 = part * ah + tid * 2;
    const bool a_ok = tid * 2 < ah;
    float4 dv[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        const int r = lane * 4 + 256 * c;
        dv[c] = r < R ? *reinterpret_cast<const float4*>(dafb + r) : make_float4(0, 0, 0, 0);
    }
    float2 qa[NT], qc[NT];                            // two thirds in flight at most
#pragma unroll
    for (int k = 0; k < NT; ++k) qa[k] = (a_ok && k < K) ? *reinterpret_cast<const float2*>(qb + (size_t)k * A + a0) : make_float2(0, 0);
    const float al_lane = lane < K ? alpha[(size_t)b * K + lane] : 0.f;
    const float2 pa = a_ok ? *reinterpret_cast<const float2*>(p + (size_t)b * A + a0) : make_float2(0, 0);
    const float2 wa = a_ok ? *reinterpret_cast<const float2*>(w + a0) : make_float2(0, 0);
#pragma unroll
    for (int j0 = 0; j0 < DR; j0 += DB) {             // dalpha_k = daf . V_k, this wave's rows k = wave + 8 j
        float4 vv[DB][DC];
#pragma unroll
        for (int jj = 0; jj < DB; ++jj)
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const int r = lane * 4 + 256 * c, k = wave + (j0 + jj) * NWV;
                vv[jj][c] = (j0 + jj < DR && r < R && k < K) ? *reinterpret_cast<const float4*>(Vb + (size_t)k * R + r) : make_float4(0, 0, 0, 0);
            }
#pragma unroll
        for (int jj = 0; jj < DB; ++jj) {
            const int k = wave + (j0 + jj) * NWV;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < DC; ++c)
                acc += dv[c].x * vv[jj][c].x + dv[c].y * vv[jj][c].y + dv[c].z * vv[jj][c].z + dv[c].w * vv[jj][c].w;
            acc = wave_sum(acc);
            if (lane == 0 && j0 + jj < DR && k < K) sm[k] = acc;
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) qc[k] = (a_ok && NT + k < K) ? *reinterpret_cast<const float2*>(qb + (size_t)(NT + k) * A + a0) : make_float2(0, 0);
    __syncthreads();
    const float da = lane < K ? sm[lane] : 0.f;
    const float dot = wave_sum(al_lane * da);
    const float d_lane = al_lane * (da - dot);
    if (part == 0 && wave == 0 && lane < K) de[(size_t)b * K + lane] = d_lane;
    float sx = 0.f, sy = 0.f;
    auto third = [&](const float2 (&q)[NT], int k0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            if (k0 + k < K) {
                const float dk = xg_readlane(d_lane, k0 + k);
                const float tx = xg_tanh(pa.x + q[k].x), ty = xg_tanh(pa.y + q[k].y);
                sx += dk * (1.0f - tx * tx);
                sy += dk * (1.0f - ty * ty);
            }
        }
    };
    third(qa, 0);
#pragma unroll
    for (int k = 0; k < NT; ++k) qa[k] = (a_ok && 2 * NT + k < K) ? *reinterpret_cast<const float2*>(qb + (size_t)(2 * NT + k) * A + a0) : make_float2(0, 0);
    third(qc, NT);
    third(qa, 2 * NT);
    if (a_ok) *reinterpret_cast<float2*>(dp + (size_t)b * A + a0) = make_float2(sx * wa.x, sy * wa.y);
}


}  // namespace

int xgk_attn_fwd(hipStream_t st, const float* p, const float* vproj, const float* V, const float* w, float* alpha,
                 float* af, int B, int K, int R, int A, bool half_cu) {
    if (K > 8192) return XG_EINVAL;
    const bool al16 = ((uintptr_t)p % 16 == 0) && ((uintptr_t)vproj % 16 == 0) && ((uintptr_t)V % 16 == 0) && ((uintptr_t)w % 16 == 0);
    if (al16 && A % 4 == 0 && A <= 2048 && R % 4 == 0 && R <= 4 * FT && R >= 4) {
        const int r4n = R / 4;
        const int nkp = FT / r4n > 0 ? (FT / r4n < K ? FT / r4n : K) : 1;
        if (K <= 16 * nkp) {
            const int ni = (A + 255) / 256;
            // the half-CU form: <= 64 frames (one softmax per wave), <= 4 V rows per thread, column groups in pairs
            const bool half = half_cu && K <= 64 && K <= 4 * nkp;
            const size_t lds = (size_t)((half ? 2 : 1) * ((K + 3) & ~3) + (size_t)nkp * R) * sizeof(float);
            if (lds <= 60000) {
#define XG_ATTN_LAUNCH(NI_, VM_) hipLaunchKernelGGL((attn_fwd_fast<NI_, VM_, false>), dim3(B), dim3(FT), lds, st, p, vproj, V, w, alpha, af, K, R, A)
#define XG_ATTN_LAUNCH_H(NI_) hipLaunchKernelGGL((attn_fwd_fast<NI_, 4, true>), dim3(B), dim3(FT), lds, st, p, vproj, V, w, alpha, af, K, R, A)
                if (half) {
                    if (ni <= 2) XG_ATTN_LAUNCH_H(2); else if (ni <= 4) XG_ATTN_LAUNCH_H(4);
                    else if (ni <= 6) XG_ATTN_LAUNCH_H(6); else XG_ATTN_LAUNCH_H(8);
                } else if (K <= 8 * nkp) {
                    if (ni <= 2) XG_ATTN_LAUNCH(2, 8); else if (ni <= 4) XG_ATTN_LAUNCH(4, 8);
                    else if (ni <= 6) XG_ATTN_LAUNCH(6, 8); else XG_ATTN_LAUNCH(8, 8);
                } else {                              // e.g. hidden 1024 (4 row parts) x 40 frames
                    if (ni <= 2) XG_ATTN_LAUNCH(2, 16); else if (ni <= 4) XG_ATTN_LAUNCH(4, 16);
                    else if (ni <= 6) XG_ATTN_LAUNCH(6, 16); else XG_ATTN_LAUNCH(8, 16);
                }
#undef XG_ATTN_LAUNCH
#undef XG_ATTN_LAUNCH_H
                XG_CHECK_LAUNCH();
                return XG_OK;
            }
        }
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(ATPB), K * sizeof(float), st, p, vproj, V, w, alpha, af, K, R, A);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_bwd(hipStream_t st, const float* daf, int lddaf, const float* p, const float* vproj, const float* V,
                 const float* w, const float* alpha, float* de, float* dp, int B, int K, int R, int A) {
    if (K > 8192) return XG_EINVAL;
    const bool al16 = ((uintptr_t)p % 16 == 0) && ((uintptr_t)vproj % 16 == 0) && ((uintptr_t)V % 16 == 0) &&
                      ((uintptr_t)w % 16 == 0) && ((uintptr_t)daf % 16 == 0) && ((uintptr_t)dp % 16 == 0);
    // two 512-thread workgroups per video whenever the shapes allow: up to 32 frames they hold <= 128 VGPRs, i.e. one of
    // them fits into the half of a CU a background GEMM workgroup leaves free (xg_gemm.hip: XGK_GEMM_BG) -- the
    // one-workgroup form (1024 threads x 125 VGPRs) needs an EMPTY CU and waited for one for up to 240 us per step beside
    // dW_logit.  For 33-48 frames: the staged form attn_bwd_split_lr (the one-workgroup form would spill there).
    if (al16 && A % 4 == 0 && A <= 2048 && R % 4 == 0 && R <= 1024 && lddaf % 4 == 0 && K <= 48) {
        const size_t lds = (size_t)K * sizeof(float);
        if (K <= 16) hipLaunchKernelGGL((attn_bwd_split<16>), dim3(2 * B), dim3(512), lds, st, daf, lddaf, p, vproj, V, w, alpha, de, dp, K, R, A);
        else if (K <= 32) hipLaunchKernelGGL((attn_bwd_split<32>), dim3(2 * B), dim3(512), lds, st, daf, lddaf, p, vproj, V, w, alpha, de, dp, K, R, A);
        else hipLaunchKernelGGL((attn_bwd_split_lr<48>), dim3(2 * B), dim3(512), lds, st, daf, lddaf, p, vproj, V, w, alpha, de, dp, K, R, A);
        XG_CHECK_LAUNCH();
        return XG_OK;
    }
    if (al16 && A % 4 == 0 && A <= 2 * FT && R % 4 == 0 && R <= 1024 && lddaf % 4 == 0 && K <= 48) {
        const size_t lds = (size_t)K * sizeof(float);
        if (K <= 16) hipLaunchKernelGGL((attn_bwd_fast<16>), dim3(B), dim3(FT), lds, st, daf, lddaf, p, vproj, V, w, alpha, de, dp, K, R, A);
        else if (K <= 32) hipLaunchKernelGGL((attn_bwd_fast<32>), dim3(B), dim3(FT), lds, st, daf, lddaf, p, vproj, V, w, alpha, de, dp, K, R, A);
        else hipLaunchKernelGGL((attn_bwd_fast<48>), dim3(B), dim3(FT), lds, st, daf, lddaf, p, vproj, V, w, alpha, de, dp, K, R, A);
        XG_CHECK_LAUNCH();
        return XG_OK;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B), dim3(ATPB), K * sizeof(float), st, daf, lddaf, p, vproj, V, w, alpha,
                       de, dp, K, R, A);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_bwd_post(hipStream_t st, const float* P, const float* vproj, const float* w, const float* DE,
                      float* dvproj, float* dw, int T, int B, int K, int A) {
    if ((size_t)T * K * sizeof(float) > 60000) return XG_EINVAL;
    const dim3 grid(xg_cdiv(A, 256), B);
    const size_t lds = (size_t)T * K * sizeof(float);
    // (the step axis is unrolled to TMAX: 21 steps in a 32-step body wasted a third of the tanh work, 73 us)
    if (T <= 8) hipLaunchKernelGGL((attn_bwd_post_kernel<8>), grid, dim3(256), lds, st, P, vproj, w, DE, dvproj, dw, T, B, K, A);
    else if (T <= 24) hipLaunchKernelGGL((attn_bwd_post_kernel<24>), grid, dim3(256), lds, st, P, vproj, w, DE, dvproj, dw, T, B, K, A);
    else if (T <= 32) hipLaunchKernelGGL((attn_bwd_post_kernel<32>), grid, dim3(256), lds, st, P, vproj, w, DE, dvproj, dw, T, B, K, A);
    else hipLaunchKernelGGL((attn_bwd_post_kernel<0>), grid, dim3(256), lds, st, P, vproj, w, DE, dvproj, dw, T, B, K, A);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_post_dV(hipStream_t st, const float* P, const float* vproj, const float* w, const float* DE, float* dvproj, float* dw,
                     const float* ALPHA, const float* DAF, int lddaf, int64_t daf_tstride, float* dV, int T, int B, int K, int A, int R) {
    if (T > 32 || (size_t)T * K * sizeof(float) > 60000) {           // long sequences: the two separate passes
        XG_TRY(xgk_attn_dV(st, ALPHA, DAF, lddaf, daf_tstride, dV, T, B, K, R, false));
        return xgk_attn_bwd_post(st, P, vproj, w, DE, dvproj, dw, T, B, K, A);
    }
    const int nxa = xg_cdiv(A, 256), nxr = xg_cdiv(R, 256);
    const dim3 grid(nxa + nxr, B);
    const size_t lds = (size_t)T * K * sizeof(float);
#define XG_PDV(TM) hipLaunchKernelGGL((attn_post_dV_kernel<TM>), grid, dim3(256), lds, st, P, vproj, w, DE, dvproj, dw, ALPHA, DAF, lddaf, \
                                      daf_tstride, dV, T, B, K, A, R, nxa)
    if (T <= 8) XG_PDV(8); else if (T <= 24) XG_PDV(24); else XG_PDV(32);
#undef XG_PDV
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_dV(hipStream_t st, const float* ALPHA, const float* DAF, int lddaf, int64_t daf_tstride, float* dV,
                int T, int B, int K, int R, bool accumulate) {
    if (T <= 32) {
        const dim3 grid(xg_cdiv(R, 256), B);
        const size_t lds = 0;
        if (T <= 8) hipLaunchKernelGGL((attn_dV_reg_kernel<8>), grid, dim3(256), lds, st, ALPHA, DAF, lddaf, daf_tstride, dV, T, B, K, R, accumulate ? 1 : 0);
        else if (T <= 24) hipLaunchKernelGGL((attn_dV_reg_kernel<24>), grid, dim3(256), lds, st, ALPHA, DAF, lddaf, daf_tstride, dV, T, B, K, R, accumulate ? 1 : 0);
        else hipLaunchKernelGGL((attn_dV_reg_kernel<32>), grid, dim3(256), lds, st, ALPHA, DAF, lddaf, daf_tstride, dV, T, B, K, R, accumulate ? 1 : 0);
        XG_CHECK_LAUNCH();
        return XG_OK;
    }
    const int64_t n = (int64_t)B * K * R;
    hipLaunchKernelGGL(attn_dV_kernel, dim3((unsigned)xg_cdiv64(n, 256)), dim3(256), 0, st, ALPHA, DAF, lddaf,
                       daf_tstride, dV, T, B, K, R, accumulate ? 1 : 0);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
