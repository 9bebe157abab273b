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
__global__ void __launch_bounds__(256) attn_bwd_post_kernel(const float* __restrict__ P, const float* __restrict__ vproj,
                                                              const float* __restrict__ w, const float* __restrict__ DE,
                                                              float* __restrict__ dvproj, float* __restrict__ dw, int T,
                                                              int B, int K, int A) {
    extern __shared__ float sde[];         // DE[:, b, :]  (T*K)
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < T * K; i += 256) sde[i] = DE[((size_t)(i / K) * B + b) * K + (i % K)];
    __syncthreads();
    if (a >= A) return;
    const float wa = w[a];
    float dwa = 0.f;
    for (int k = 0; k < K; ++k) {
        const float q = vproj[((size_t)b * K + k) * A + a];
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const float d = sde[t * K + k];
            if (d != 0.f) {
                const float th = xg_tanh(P[((size_t)t * B + b) * A + a] + q);
                acc += d * (1.0f - th * th);
                dwa += d * th;
            }
        }
        dvproj[((size_t)b * K + k) * A + a] = acc * wa;
    }
    atomicAdd(dw + a, dwa);
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

}  // namespace

int xgk_attn_fwd(hipStream_t st, const float* p, const float* vproj, const float* V, const float* w, float* alpha,
                 float* af, int B, int K, int R, int A) {
    if (K > 8192) return XG_EINVAL;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B), dim3(ATPB), K * sizeof(float), st, p, vproj, V, w, alpha, af, K, R, A);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_bwd(hipStream_t st, const float* daf, int lddaf, const float* p, const float* vproj, const float* V,
                 const float* w, const float* alpha, float* de, float* dp, int B, int K, int R, int A) {
    if (K > 8192) return XG_EINVAL;
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B), dim3(ATPB), K * sizeof(float), st, daf, lddaf, p, vproj, V, w, alpha,
                       de, dp, K, R, A);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_bwd_post(hipStream_t st, const float* P, const float* vproj, const float* w, const float* DE,
                      float* dvproj, float* dw, int T, int B, int K, int A) {
    if ((size_t)T * K * sizeof(float) > 60000) return XG_EINVAL;
    hipLaunchKernelGGL(attn_bwd_post_kernel, dim3(xg_cdiv(A, 256), B), dim3(256), (size_t)T * K * sizeof(float), st, P,
                       vproj, w, DE, dvproj, dw, T, B, K, A);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_attn_dV(hipStream_t st, const float* ALPHA, const float* DAF, int lddaf, int64_t daf_tstride, float* dV,
                int T, int B, int K, int R, bool accumulate) {
    const int64_t n = (int64_t)B * K * R;
    hipLaunchKernelGGL(attn_dV_kernel, dim3((unsigned)xg_cdiv64(n, 256)), dim3(256), 0, st, ALPHA, DAF, lddaf,
                       daf_tstride, dV, T, B, K, R, accumulate ? 1 : 0);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
