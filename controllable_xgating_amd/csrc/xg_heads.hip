// Vocabulary / category heads, criteria and rollout token choice (gfx950).
// reference: caption_src/SAModel.py:109-110 (log_softmax of logit / classifer), :186-196 (token
// choice), :200-215 (rollout bookkeeping), :221-267 (criteria).
// Row kernels: one 1024-thread workgroup per row of V (= 20000) logits; the row is read with
// coalesced loads, max / sum-exp are reduced with wave64 shuffles + LDS across the 16 waves.
#include "xg_common.h"
#include "xg_kernels.h"
#include "xg_select.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int RT = 1024;

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    return r;
}

__device__ __forceinline__ int out_row(int i, int inner, int outer, int permute) {
    return permute ? (i % inner) * outer + i / inner : i;
}

__global__ void __launch_bounds__(RT) log_softmax_kernel(const float* __restrict__ in, int ldin, float* __restrict__ out,
                                                           int ldout, int V, int inner, int outer, int permute) {
    __shared__ float red[RT / 64];
    const int i = blockIdx.x;
    const float* x = in + (size_t)i * ldin;
    float* y = out + (size_t)out_row(i, inner, outer, permute) * ldout;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += RT) mx = fmaxf(mx, x[v]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += RT) s += expf(x[v] - mx);
    s = block_sum(s, red);
    const float lse = mx + logf(s);
    for (int v = threadIdx.x; v < V; v += RT) y[v] = x[v] - lse;
}

// Register-resident rows (V <= NPT * 1024): every element is read ONCE (all loads in flight before the first use) and
// written once; the three-pass kernels above re-read the row from L2 for every pass.
template <int NPT>
__global__ void __launch_bounds__(RT) log_softmax_reg_kernel(const float* in, int ldin, float* out, int ldout, int V,
                                                               int inner, int outer, int permute) {
    __shared__ float red[RT / 64];
    const int i = blockIdx.x;
    const float* x = in + (size_t)i * ldin;
    float* y = out + (size_t)out_row(i, inner, outer, permute) * ldout;
    float xv[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) { const int v = threadIdx.x + j * RT; xv[j] = v < V ? x[v] : -INFINITY; }
    float mx = xv[0];
#pragma unroll
    for (int j = 1; j < NPT; ++j) mx = fmaxf(mx, xv[j]);
    mx = block_max(mx, red);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) s += expf(xv[j] - mx);          // exp(-inf) = 0 for the padding
    s = block_sum(s, red);
    const float lse = mx + logf(s);
#pragma unroll
    for (int j = 0; j < NPT; ++j) { const int v = threadIdx.x + j * RT; if (v < V) y[v] = xv[j] - lse; }
}
template <int NPT>
__global__ void __launch_bounds__(RT) log_softmax_bwd_reg_kernel(const float* dlogp, const float* logp, int ldp, float* dlogits,
                                                                   int ldd, int V, int inner, int outer, int permute) {
    __shared__ float red[RT / 64];
    const int i = blockIdx.x;
    const size_t ro = (size_t)out_row(i, inner, outer, permute != 0) * ldp;
    const float* dy = dlogp + ro;
    const float* y = logp + (permute == 2 ? (size_t)i * ldp : ro);
    float dv[NPT], yv[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int v = threadIdx.x + j * RT;
        dv[j] = v < V ? dy[v] : 0.f;
        yv[j] = v < V ? y[v] : -INFINITY;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) s += dv[j];
    s = block_sum(s, red);
    float* dx = dlogits + (size_t)i * ldd;
#pragma unroll
    for (int j = 0; j < NPT; ++j) { const int v = threadIdx.x + j * RT; if (v < V) dx[v] = dv[j] - expf(yv[j]) * s; }
}

__global__ void __launch_bounds__(RT) log_softmax_bwd_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                               int ldp, float* __restrict__ dlogits, int ldd, int V,
                                                               int inner, int outer, int permute) {
    __shared__ float red[RT / 64];
    const int i = blockIdx.x;
    const size_t ro = (size_t)out_row(i, inner, outer, permute != 0) * ldp;
    const float* dy = dlogp + ro;
    const float* y = logp + (permute == 2 ? (size_t)i * ldp : ro);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += RT) s += dy[v];
    s = block_sum(s, red);
    float* dx = dlogits + (size_t)i * ldd;
    for (int v = threadIdx.x; v < V; v += RT) dx[v] = dy[v] - expf(y[v]) * s;
}

// ---- criteria on a materialised (B,T,V) log-prob tensor
__device__ __forceinline__ int64_t tgt_of(const int64_t* target, int b, int t, int T, int roll) {
    return roll ? target[(size_t)b * T + (t + 1 < T ? t + 1 : 0)] : target[(size_t)b * T + t];   // SAModel.py:228
}
__global__ void nll_fwd_kernel(const float* logp, const int64_t* target, const float* mask, const float* mask2,
                               int B, int T, int V, int roll, float* out2) {
    __shared__ float red[256 / 64];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a = 0.f, m = 0.f;
    if (i < B * T) {
        const int b = i / T, t = i % T;
        m = mask[i] * (mask2 ? mask2[i] : 1.f);
        const int64_t tg = tgt_of(target, b, t, T, roll);
        a = -logp[(size_t)i * V + tg] * m;
    }
    a = block_sum(a, red);
    m = block_sum(m, red);
    if (threadIdx.x == 0) { atomicAdd(out2, a); atomicAdd(out2 + 1, m); }
}
__global__ void nll_bwd_kernel(const int64_t* target, const float* mask, const float* mask2, int B, int T, int V,
                               int roll, const float* sums, float scale, const float* scale_dev, float* dlogp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i % T;
    const float m = mask[i] * (mask2 ? mask2[i] : 1.f);
    dlogp[(size_t)i * V + tgt_of(target, b, t, T, roll)] = -scale * (scale_dev ? scale_dev[0] : 1.f) * m / sums[1];
}

// ---- RewardCriterion (caption_src/SAModel.py:259-267): loss = -sum(slp * reward * mask) / sum(mask) with
// mask[:, 0] = 1, mask[:, t] = (seq[:, t-1] > 0).  n_dev (device int32, may be null) is the reference's early-exit width:
// columns t >= n do not exist there, so they carry no mask here -- the caller keeps full-width (m, L) tensors and never
// has to bring n to the host.  reward element (b, t) at reward[b * rs_b + t * rs_t] (rs_t = 0: one value per video,
// caption_src/myutils.py:75-76).
__device__ __forceinline__ float reward_mask(const int64_t* seq, int ld_seq, int b, int t, int n) {
    if (t >= n) return 0.f;
    return (t == 0 || seq[(size_t)b * ld_seq + t - 1] > 0) ? 1.f : 0.f;
}
__global__ void reward_fwd_kernel(const float* slp, int ld_slp, const int64_t* seq, int ld_seq, const float* reward, int rs_b,
                                  int rs_t, const int32_t* n_dev, int m, int L, float* out2) {
    __shared__ float red[256 / 64];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_dev ? min(max(*n_dev, 0), L) : L;
    float a = 0.f, k = 0.f;
    if (i < m * L) {
        const int b = i / L, t = i % L;
        k = reward_mask(seq, ld_seq, b, t, n);
        if (k != 0.f) a = -slp[(size_t)b * ld_slp + t] * reward[(size_t)b * rs_b + (size_t)t * rs_t];
    }
    a = block_sum(a, red);
    k = block_sum(k, red);
    if (threadIdx.x == 0) { atomicAdd(out2, a); atomicAdd(out2 + 1, k); }
}
__global__ void reward_bwd_kernel(const int64_t* seq, int ld_seq, const float* reward, int rs_b, int rs_t, const int32_t* n_dev,
                                  int m, int L, const float* sums, const float* scale_dev, float* dslp, int ld_d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m * L) return;
    const int n = n_dev ? min(max(*n_dev, 0), L) : L;
    const int b = i / L, t = i % L;
    const float k = reward_mask(seq, ld_seq, b, t, n);
    dslp[(size_t)b * ld_d + t] = k != 0.f ? -(scale_dev ? scale_dev[0] : 1.f) * reward[(size_t)b * rs_b + (size_t)t * rs_t] / sums[1] : 0.f;
}

// ---- fused cross-entropy on time-major logits rows i = t*B + b (rows row0 .. row0 + gridDim.x - 1 per launch)
__device__ __forceinline__ void xent_row_finish(float mx, float s, float* red, const float* x, int i, int b, int t, int T,
                                                const int64_t* seq, const float* mask, const float* mask2, int roll,
                                                float* lse_out, float* sums2) {
    const float gmx = block_max(mx, red);
    s = (mx == -INFINITY) ? 0.f : s * expf(mx - gmx);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float lse = gmx + logf(s);
        lse_out[i] = lse;
        const float m = mask[(size_t)b * T + t] * (mask2 ? mask2[(size_t)b * T + t] : 1.f);
        const int64_t tg = tgt_of(seq, b, t, T, roll);
        atomicAdd(sums2, -(x[tg] - lse) * m);
        atomicAdd(sums2 + 1, m);
    }
}
__global__ void __launch_bounds__(RT) xent_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* seq,
                                                        const float* mask, const float* mask2, int B, int T, int V,
                                                        int roll, float* lse_out, float* sums2, int row0) {
    __shared__ float red[RT / 64];
    const int i = row0 + blockIdx.x, t = i / B, b = i % B;
    const float* x = logits + (size_t)i * ld;
    // one pass over the row: online max / sum-exp per thread, merged across the workgroup
    // (batches of 8 loads are issued before any of them is consumed: the online update is a serial dependency and a
    //  load-use-load-use loop costs one memory round trip per element)
    float mx = -INFINITY, s = 0.f;
    for (int v0 = threadIdx.x; v0 < V; v0 += 8 * RT) {
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int v = v0 + j * RT; xv[j] = v < V ? x[v] : -INFINITY; }
        float bm = xv[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) bm = fmaxf(bm, xv[j]);
        if (bm > mx) { s *= expf(mx - bm); mx = bm; }        // bm > -inf here, mx may be -inf (s = 0 then)
        if (mx > -INFINITY) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s += expf(xv[j] - mx);   // exp(-inf) = 0 for the padding
        }
    }
    xent_row_finish(mx, s, red, x, i, b, t, T, seq, mask, mask2, roll, lse_out, sums2);
}
// Register-resident row (V <= NV4 * 4096, 16-byte aligned rows): every thread requests ALL its 16-byte pieces before it
// uses the first one -- one memory round trip per workgroup instead of one per batch (vocabulary 20000: 89 -> 5x us)
template <int NV4>
__global__ void __launch_bounds__(RT) xent_fwd_reg_kernel(const float* __restrict__ logits, int ld, const int64_t* seq,
                                                            const float* mask, const float* mask2, int B, int T, int V,
                                                            int roll, float* lse_out, float* sums2, int row0) {
    __shared__ float red[RT / 64];
    const int i = row0 + blockIdx.x, t = i / B, b = i % B;
    const float* x = logits + (size_t)i * ld;
    float4 xv[NV4];
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int v = (threadIdx.x + j * RT) * 4;
        if (v + 3 < V) xv[j] = *reinterpret_cast<const float4*>(x + v);
        else {
            xv[j].x = v < V ? x[v] : -INFINITY;         xv[j].y = v + 1 < V ? x[v + 1] : -INFINITY;
            xv[j].z = v + 2 < V ? x[v + 2] : -INFINITY; xv[j].w = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV4; ++j) mx = fmaxf(fmaxf(mx, fmaxf(xv[j].x, xv[j].y)), fmaxf(xv[j].z, xv[j].w));
    float s = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
        for (int j = 0; j < NV4; ++j) s += (expf(xv[j].x - mx) + expf(xv[j].y - mx)) + (expf(xv[j].z - mx) + expf(xv[j].w - mx));
    }
    xent_row_finish(mx, s, red, x, i, b, t, T, seq, mask, mask2, roll, lse_out, sums2);
}
// dlogits = coef * (softmax - onehot), coef = scale * mask / sum(mask), in place
template <bool BF>     // BF: also write a bf16 copy of the gradient (the bf16 products read it: gemm_mode 1)
__global__ void __launch_bounds__(RT) xent_bwd_kernel(float* __restrict__ logits, int ld, const int64_t* seq,
                                                        const float* mask, const float* mask2, int B, int T, int V,
                                                        int roll, const float* lse, const float* sums2,
                                                        const float* scale_dev, float scale, int row0, unsigned short* d16) {
    const int i = row0 + blockIdx.x, t = i / B, b = i % B;
    float* x = logits + (size_t)i * ld;
    const float m = mask[(size_t)b * T + t] * (mask2 ? mask2[(size_t)b * T + t] : 1.f);
    const float coef = (scale_dev ? scale_dev[0] : 1.f) * scale * m / sums2[1];
    const int64_t tg = tgt_of(seq, b, t, T, roll);
    const float l = lse[i];
    for (int v = threadIdx.x; v < V; v += RT) {
        const float pv = expf(x[v] - l);
        const float gv = coef * (pv - (v == tg ? 1.f : 0.f));
        x[v] = gv;
        if (BF) {                                // (round to nearest even)
            unsigned u = __float_as_uint(gv);
            u += 0x7FFFu + ((u >> 16) & 1u);
            d16[(size_t)i * ld + v] = (unsigned short)(u >> 16);
        }
    }
}

// Register-resident row of the same pass (16-byte aligned rows, V <= NV4 * 4096): all of a thread's 16-byte pieces are requested
// before the first is used and written back as 16-byte stores -- the dword loop above pays a memory round trip per trip of
// 1024 elements (vocabulary 20000: 20 dependent trips; 141 us beside the reverse-time loop, this form 5x us).  Same
// arithmetic per element: results are bit-identical.
template <int NV4, bool BF>
__global__ void __launch_bounds__(RT) xent_bwd_reg_kernel(float* __restrict__ logits, int ld, const int64_t* seq,
                                                            const float* mask, const float* mask2, int B, int T, int V,
                                                            int roll, const float* lse, const float* sums2,
                                                            const float* scale_dev, float scale, int row0, unsigned short* d16) {
    const int i = row0 + blockIdx.x, t = i / B, b = i % B;
    float* x = logits + (size_t)i * ld;
    float4 xv[NV4];
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int v = (threadIdx.x + j * RT) * 4;
        if (v + 3 < V) xv[j] = *reinterpret_cast<const float4*>(x + v);
        else {
            xv[j].x = v < V ? x[v] : 0.f;         xv[j].y = v + 1 < V ? x[v + 1] : 0.f;
            xv[j].z = v + 2 < V ? x[v + 2] : 0.f; xv[j].w = 0.f;
        }
    }
    const float m = mask[(size_t)b * T + t] * (mask2 ? mask2[(size_t)b * T + t] : 1.f);
    const float coef = (scale_dev ? scale_dev[0] : 1.f) * scale * m / sums2[1];
    const int tg = (int)tgt_of(seq, b, t, T, roll);
    const float l = lse[i];
    auto bf = [](float gv) -> unsigned short {           // (round to nearest even)
        unsigned u = __float_as_uint(gv);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    };
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int v = (threadIdx.x + j * RT) * 4;
        float4 gq;
        gq.x = coef * (expf(xv[j].x - l) - (v == tg ? 1.f : 0.f));
        gq.y = coef * (expf(xv[j].y - l) - (v + 1 == tg ? 1.f : 0.f));
        gq.z = coef * (expf(xv[j].z - l) - (v + 2 == tg ? 1.f : 0.f));
        gq.w = coef * (expf(xv[j].w - l) - (v + 3 == tg ? 1.f : 0.f));
        if (v + 3 < V) {
            *reinterpret_cast<float4*>(x + v) = gq;
            if (BF) {
                ushort4 h; h.x = bf(gq.x); h.y = bf(gq.y); h.z = bf(gq.z); h.w = bf(gq.w);
                *reinterpret_cast<ushort4*>(d16 + (size_t)i * ld + v) = h;
            }
        } else {
            const float ga[3] = {gq.x, gq.y, gq.z};
            for (int q = 0; q < 3 && v + q < V; ++q) {
                x[v + q] = ga[q];
                if (BF) d16[(size_t)i * ld + v + q] = bf(ga[q]);
            }
        }
    }
}

// ---- rollout token choice: one workgroup per video
__global__ void __launch_bounds__(RT) choose_kernel(const float* __restrict__ logp, int V, int mode, const float* uniforms,
                                                      const int64_t* forced, int64_t fstride, float temperature,
                                                      int64_t* tok, float* tok_logp) {
    __shared__ float red[RT / 64];
    __shared__ int redi[RT / 64];
    __shared__ double wave_tot[RT / 64];
    const int b = blockIdx.x;
    const float* x = logp + (size_t)b * V;
    if (mode == XG_ROLLOUT_REPLAY) {
        if (threadIdx.x == 0) {
            int64_t t = forced[(size_t)b * fstride];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            tok[b] = t;
            tok_logp[b] = x[t];
        }
        return;
    }
    if (mode == XG_ROLLOUT_GREEDY) {
        // argmax, ties -> lowest index (torch.max, SAModel.py:186)
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int v = threadIdx.x; v < V; v += RT) {
            const float f = x[v];
            if (f > best || (f == best && v < bi)) { best = f; bi = v; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { red[wave] = best; redi[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < RT / 64; ++i)
                if (red[i] > best || (red[i] == best && redi[i] < bi)) { best = red[i]; bi = redi[i]; }
            tok[b] = bi;
            tok_logp[b] = best;
        }
        return;
    }
    // SAMPLE: inverse CDF over w_v = exp(logp_v / temperature) (unnormalised, like torch.multinomial):
    // each thread owns a contiguous chunk; chunk sums -> block scan -> owner thread walks its chunk.
    const int per = (V + RT - 1) / RT;
    const int v0 = threadIdx.x * per, v1 = min(V, v0 + per);
    const float invt = 1.0f / temperature;
    float s = 0.f;
    for (int v = v0; v < v1; ++v) s += expf(x[v] * invt);
    // block-wide inclusive scan of the chunk sums in double, then the first chunk whose running sum passes the target
    __shared__ float target_s; __shared__ int owner; __shared__ float base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = (double)s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wave_tot[wave] = inc;
    if (threadIdx.x == 0) owner = RT - 1;
    __syncthreads();
    double off = 0.0, tot = 0.0;
#pragma unroll
    for (int i = 0; i < RT / 64; ++i) { const double wt = wave_tot[i]; if (i < wave) off += wt; tot += wt; }
    inc += off;
    const double target = (double)uniforms[b] * tot;
    if (inc > target) atomicMin(&owner, (int)threadIdx.x);
    __syncthreads();
    if ((int)threadIdx.x == owner) { base_s = (float)(inc - (double)s); target_s = (float)target; }
    __syncthreads();
    if ((int)threadIdx.x == owner) {
        float run = base_s; int pick = min(V, v1) - 1;
        if (pick < v0) pick = V - 1;
        for (int v = v0; v < v1; ++v) {
            run += expf(x[v] * invt);
            if (run > target_s) { pick = v; break; }
        }
        tok[b] = pick;
        tok_logp[b] = x[pick];
    }
}

// ------------------------------------------------------------------------------------------------
// One launch per rollout step (SAModel.py:182-215): token choice straight from the RAW logits of the previous step
// (log-probs are never materialised: logp[v] = logit[v] - lse), the unfinished / emit bookkeeping, the reference's
// n (first step at which every row is finished) and the embedding gather of the chosen token.  One workgroup per video.
// (RollStepArgs / roll_bookkeep: xg_select.h, shared with the SELECT prologue of the step kernel)

// STAGE: the row's V logits are parked in LDS by the first pass (V * 4 bytes <= 150 KB), so the log-sum-exp / chunk-sum /
// owner passes read LDS instead of going back to L2 three more times.
template <bool STAGE>
__global__ void __launch_bounds__(RT) rollout_step_kernel(RollStepArgs a) {
    XG_CHAIN_PRIO();
    extern __shared__ float xs[];
    __shared__ float red[RT / 64];
    __shared__ int redi[RT / 64];
    __shared__ double wave_tot[RT / 64];
    __shared__ int64_t s_tok;
    __shared__ float s_bcast[3];
    __shared__ int s_owner;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.t == 0) {                                       // <bos> = 0, mask of ones (:183-184, :212-213)
        if (tid == 0) { a.tok[b] = 0; a.unf[b] = 1.0f; s_tok = 0; }
    } else {
        const float* x = a.logits + (size_t)b * a.V;
        const bool second = b >= a.split;
        const int mode = (second && a.mode == XG_ROLLOUT_SAMPLE) ? XG_ROLLOUT_GREEDY : a.mode;
        int32_t* maxf = a.maxf + (second ? 1 : 0);
        // pass 1: row max (+ argmax, ties -> lowest index like torch.max) and sum exp for the log-sum-exp
        float best = -INFINITY; int bi = 0x7fffffff;
        // every load of the row is issued before any is consumed (24 per thread cover V <= 24576: ONE memory round trip
        // instead of three), then the (branchy) argmax update
        constexpr int NLD = 24;
        for (int v0 = tid; v0 < a.V; v0 += NLD * RT) {
            float fl[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) { const int v = v0 + j * RT; fl[j] = v < a.V ? x[v] : -INFINITY; }
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int v = v0 + j * RT;
                if (v < a.V) {
                    const float f = fl[j];
                    if (STAGE) xs[v] = f;
                    if (f > best || (f == best && v < bi)) { best = f; bi = v; }
                }
            }
        }
        const float* xr = STAGE ? xs : x;            // valid after the barrier of the reduction below
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) { red[wave] = best; redi[wave] = bi; }
        __syncthreads();
        best = red[0]; bi = redi[0];
        for (int i = 1; i < RT / 64; ++i)
            if (red[i] > best || (red[i] == best && redi[i] < bi)) { best = red[i]; bi = redi[i]; }
        const float mx = best;
        // log-sum-exp: its own pass, except for a temperature-1 draw whose chunk sums are the same exponentials
        const bool lse_from_scan = mode == XG_ROLLOUT_SAMPLE && a.temperature == 1.0f;
        float lse = 0.f;
        if (!lse_from_scan) {
            float se = 0.f;
            for (int v = tid; v < a.V; v += RT) se += __expf(xr[v] - mx);
            se = block_sum(se, red);
            lse = mx + logf(se);
        }
        int64_t tk;
        if (mode == XG_ROLLOUT_GREEDY) {
            tk = bi;
        } else if (mode == XG_ROLLOUT_REPLAY) {
            tk = a.forced[(size_t)b * a.fstride];
            tk = tk < 0 ? 0 : (tk >= a.V ? a.V - 1 : tk);
        } else {
            // inverse CDF over w_v = exp((logit_v - max) / temperature) (= exp(logp_v / temperature) up to a constant, :190-194)
            const int per = (a.V + RT - 1) / RT;
            const int v0 = tid * per, v1 = min(a.V, v0 + per);
            const float invt = 1.0f / a.temperature;
            float cs = 0.f;
            for (int v = v0; v < v1; ++v) cs += __expf((xr[v] - mx) * invt);     // (v_exp_f32; the re-walk below uses the same)
            // block-wide inclusive scan of the RT chunk sums in double (was a 1024-step serial walk by one thread: 50 us)
            double inc = (double)cs;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double up = __shfl_up(inc, o, 64);
                if (lane >= o) inc += up;
            }
            if (lane == 63) wave_tot[wave] = inc;
            if (tid == 0) s_owner = RT - 1;
            __syncthreads();
            double off = 0.0, tot = 0.0;
#pragma unroll
            for (int i = 0; i < RT / 64; ++i) { const double wt = wave_tot[i]; if (i < wave) off += wt; tot += wt; }
            inc += off;
            if (lse_from_scan) lse = mx + logf((float)tot);
            const double target = (double)a.uniforms[b] * tot;
            if (inc > target) atomicMin(&s_owner, tid);          // first chunk whose running sum passes the target
            __syncthreads();
            if (tid == s_owner) { s_bcast[0] = (float)(inc - (double)cs); s_bcast[1] = (float)target; }
            __syncthreads();
            if (tid == s_owner) {
                float run = s_bcast[0]; int pick = min(a.V, v1) - 1;
                if (pick < v0) pick = a.V - 1;
                for (int v = v0; v < v1; ++v) {
                    run += __expf((xr[v] - mx) * invt);
                    if (run > s_bcast[1]) { pick = v; break; }
                }
                s_tok = pick;
            }
            __syncthreads();
            tk = s_tok;
        }
        if (tid == 0) {
            roll_bookkeep(a, b, mode, tk, xr[tk] - lse, lse, maxf);
            s_tok = tk;
        }
    }
    __syncthreads();
    const int64_t tk = s_tok;
    for (int e = tid; e < a.E; e += RT) a.xt[(size_t)b * a.E + e] = a.table[(size_t)tk * a.E + e];
}
// ------------------------------------------------------------------------------------------------
// Rollout steps of at most 128 rows (round 3): the vocabulary product and the token choice as TWO light launches instead of a
// (B, V) product that writes 10 MB of logits and a one-workgroup-per-row pass that reads them back three times.
//   vocab_part_kernel   logits tile = h2' W_logit^T + b for ALL rows x 32 vocabulary columns per workgroup (4 waves, one 32 x 32
//                       MFMA tile each, the W slab shared), and in its epilogue the tile's per-row statistics: max, argmax (lowest
//                       column on ties), sum exp(x - max), sum exp((x - max) / temperature).  The logits themselves are stored only
//                       for rows that need them later (sampled rows: the draw re-reads ONE tile of its row, the SCST backward
//                       all of it; greedy rows of a paired rollout: never).
//   roll_select_kernel  one workgroup per row over the V / 32 tile statistics (10 KB instead of 80 KB): log-sum-exp, greedy
//                       argmax, inverse-CDF draw (block scan over the tile sums in double, then a walk through the one tile that
//                       holds the target), the bookkeeping and the embedding gather of rollout_step_kernel.
// Same arithmetic as rollout_step_kernel up to the grouping of the sums (by tile instead of by thread chunk).
typedef float v_f32x16 __attribute__((ext_vector_type(16)));
typedef float v_f32x4 __attribute__((ext_vector_type(4)));
constexpr int VT_LD = 36;                 // LDS row stride of a 32-deep slab (floats): conflict-free b128 fragment reads
struct VocabPartArgs {
    const float* H; int ldh;              // (B, R) rows
    const float* W; const float* bias;    // (V, R) row-major, (V)
    int B, R, V;
    float* logits; int wr_rows;           // rows [0, wr_rows) of the (B, V) logits are stored
    float* part;                          // (B, rs_pitch(ntiles), 4): max, sum exp(x - max), sum exp((x - max) / T), argmax column (int bits); slot order: xg_select.h
    float inv_t;
};
__global__ void __launch_bounds__(256) vocab_part_kernel(VocabPartArgs a) {
    XG_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float smem[2 * (128 + 32) * VT_LD];
    constexpr int STAGE = (128 + 32) * VT_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.x * 32, ntiles = gridDim.x;
    // per-thread load pieces of a slab: A 128 x 32 -> 4 float4 (row f >> 3, k (f & 7) * 4), W 32 x 32 -> 1 float4
    const float* ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        ap[i] = a.H + (size_t)min(f >> 3, a.B - 1) * a.ldh + ((f & 7) << 2);
    }
    const float* wp = a.W + (size_t)min(n0 + (tid >> 3), a.V - 1) * a.R + ((tid & 7) << 2);
    // global loads run TWO slabs ahead of their MFMAs (two register sets): a slab is only 16 MFMAs per wave, a third of a load's
    // round trip even with three workgroups per CU
    v_f32x4 ra[2][4], rw[2];
    auto ld = [&](int s, v_f32x4 (&xa)[4], v_f32x4& xw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const v_f32x4*>(ap[i] + s * 32);
        xw = *reinterpret_cast<const v_f32x4*>(wp + s * 32);
    };
    auto st = [&](int buf, const v_f32x4 (&xa)[4], const v_f32x4& xw) {
        float* As = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<v_f32x4*>(As + (f >> 3) * VT_LD + ((f & 7) << 2)) = xa[i];
        }
        *reinterpret_cast<v_f32x4*>(As + 128 * VT_LD + (tid >> 3) * VT_LD + ((tid & 7) << 2)) = xw;
    };
    v_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ns = a.R / 32;
    ld(0, ra[0], rw[0]);
    if (ns > 1) ld(1, ra[1], rw[1]);
    st(0, ra[0], rw[0]);
    __syncthreads();
    auto slab = [&](int s, v_f32x4 (&mine)[4], v_f32x4& minew, const v_f32x4 (&other)[4], const v_f32x4& otherw) {
        if (s + 2 < ns) ld(s + 2, mine, minew);          // (this set's slab s is in LDS)
        const float* As = smem + (s & 1) * STAGE;
        const float* Bs = As + 128 * VT_LD;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const v_f32x4 fa = *reinterpret_cast<const v_f32x4*>(As + (wave * 32 + l31) * VT_LD + kb * 8 + half * 4);
            const v_f32x4 fb = *reinterpret_cast<const v_f32x4*>(Bs + l31 * VT_LD + kb * 8 + half * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc, 0, 0, 0);
        }
        if (s + 1 < ns) st((s + 1) & 1, other, otherw);   // slab s + 1: requested two slabs ago
        __syncthreads();
    };
    for (int s = 0; s < ns; s += 2) {
        slab(s, ra[0], rw[0], ra[1], rw[1]);
        if (s + 1 < ns) slab(s + 1, ra[1], rw[1], ra[0], rw[0]);
    }
    // epilogue through LDS (the staging buffers are free: every wave is past the loop's last barrier): the tile as [128][33],
    // then thread (row, half-row) walks 16 columns -- a dozen LDS reads and two exchanges with its neighbour instead of 16 five-step
    // shuffle reductions per wave, and the stored rows leave as 16-byte pieces
    constexpr int TL = 33;
    float* tl = smem;
    {
        const int col = n0 + l31;
        const float bv = col < a.V ? a.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tl[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TL + l31] = acc[r] + bv;
    }
    __syncthreads();
    const int row = tid >> 1, h = tid & 1, c0 = n0 + h * 16;
    float x[16];
    float m = -INFINITY; int mc = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x[j] = c0 + j < a.V ? tl[row * TL + h * 16 + j] : -INFINITY;
        if (x[j] > m) { m = x[j]; mc = c0 + j; }          // (ascending columns: the first maximum stays)
    }
    {
        const float om = __shfl_xor(m, 1, 64); const int oc = __shfl_xor(mc, 1, 64);
        if (om > m || (om == m && oc < mc)) { m = om; mc = oc; }
    }
    float e1 = 0.f, et = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float d = x[j] - m;                         // (-inf for columns past V: exp -> 0)
        const float e = __expf(d);
        e1 += e;
        et += a.inv_t == 1.0f ? e : __expf(d * a.inv_t);
    }
    e1 += __shfl_xor(e1, 1, 64); et += __shfl_xor(et, 1, 64);
    if (row < a.B) {
        if (h == 0) {
            v_f32x4 pv = {m, e1, et, __int_as_float(mc)};
            *reinterpret_cast<v_f32x4*>(a.part + ((size_t)row * rs_pitch(ntiles) + rs_slot(blockIdx.x, rs_per(ntiles))) * 4) = pv;
        }
        if (row < a.wr_rows) {
            float* dst = a.logits + (size_t)row * a.V + c0;
            if ((a.V & 3) == 0 && c0 + 16 <= a.V) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v_f32x4 v = {x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
                    *reinterpret_cast<v_f32x4*>(dst + 4 * j) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) if (c0 + j < a.V) dst[j] = x[j];
            }
        }
    }
}

// sched_group_barrier pipeline of one block: M MFMAs with D LDS instructions and V global loads spread evenly between them
// (as xg_gemm.hip's w1_interleave: a lone wave per SIMD must not issue its memory instructions in a cluster)
template <int M, int D, int V>
__device__ __forceinline__ void vt_interleave() {
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
        vt_interleave<M - 1, D - d, V - v>();
    } else {
        constexpr int mm = M / MEM;
        __builtin_amdgcn_sched_group_barrier(0x008, mm, 0);
        if constexpr (V > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); vt_interleave<M - mm, D, V - 1>(); }
        else { __builtin_amdgcn_sched_group_barrier(0x080, 1, 0); vt_interleave<M - mm, D - 1, V>(); }
    }
}

// The same product over 80-column tiles: V = 20000 is 250 tiles -- ONE per CU, where the 32-column tiles are 625 on 256 CUs (three
// on some, two on others: 81 % of the matrix time of the busiest CU is useful).  16 x 16 x 4 MFMAs (a tile is 2 x 5 of them per wave:
// rows 32 w + 16 rb .., columns 16 g ..), one workgroup per CU = one wave per SIMD, so every LDS / global instruction sits between
// MFMAs (vt_interleave).  k order inside a 16-deep block: lane group g = lane / 16 owns k = 4 g .. 4 g + 3 (one ds_read_b128 per
// fragment, the same permutation for both operands).
constexpr int V16_TW = 80, V16_NG = 5;
#ifdef VP_TRACE
// in-kernel phase stamps of the vocabulary product (tools/r6/vp_trace.py): entry, first slab staged, K loop done, tile in LDS, end
__device__ long long vp_trace_buf[512 * 8];
#define VP_STAMP(i) do { if (threadIdx.x == 0) { vp_trace_buf[blockIdx.x * 8 + (i)] = wall_clock64(); if ((i) == 1 || (i) == 2) vp_trace_buf[blockIdx.x * 8 + 4 + (i)] = clock64(); } } while (0)
#else
#define VP_STAMP(i) do {} while (0)
#endif
// Round 6 (in-kernel stamps, tools/r6/vp_trace.py: K loop 23.6 of 29.1 us = 83 % of the matrix rate the clock allows, tile -> LDS 1.2,
// statistics + stores 2.7): (1) THREE LDS stages instead of two -- a slab is stored two slabs ahead of its products, so the fragments
// of the NEXT block (also across the slab boundary) are requested under the current block's MFMAs and no wave starts a slab with an
// exposed LDS round trip behind the barrier; (2) the epilogue stays in registers: a row's 80 logits sit in the 16 lanes of one DPP
// row (5 per lane), so max / argmax / the two exponential sums are DPP reductions inside the row (xg_select.h helpers) -- no
// 41 KB tile image in LDS, no barrier, and the stored rows leave as five 64-byte runs per row.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) vocab_part16_kernel(VocabPartArgs a) {
    XG_CHAIN_PRIO();
    VP_STAMP(0);
    // (16 rows beyond the 80 of the W slab: the threads whose third W piece lies past the slab store it there instead of under a branch --
    //  a branch in the middle of a block cuts the scheduling region, and the seven LDS stores then sit in a cluster with their
    //  s_waitcnt vmcnt in front of the block's MFMAs instead of between them: 39.6 cycles per MFMA instead of 32)
    // Row stride 40 floats: a fragment read (row l15, this lane group's four k: ds_read_b128) is served 16 lanes at a time -- rows
    // {0-3, 12-15} of lane group g with rows {4-11} of group g + 1 --, and at 36 (the 32 x 32 layout's stride) seven of those pairs
    // share banks; at 40 the 16 lanes cover the 64 banks exactly once.
    constexpr int V16_LD = 40;
    constexpr int STAGE = (128 + V16_TW + 16) * V16_LD;              // floats per LDS stage
    constexpr int NST = 3;
    __shared__ __attribute__((aligned(16))) float smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = (lane >> 4) << 2;
    const int n0 = blockIdx.x * V16_TW, ntiles = gridDim.x;
    const float* ap[4]; const float* wp[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        ap[i] = a.H + (size_t)min(f >> 3, a.B - 1) * a.ldh + ((f & 7) << 2);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f = min(tid + 256 * i, V16_TW * 8 - 1);            // (pieces past the 80 x 32 slab repeat its last one: never stored)
        wp[i] = a.W + (size_t)min(n0 + (f >> 3), a.V - 1) * a.R + ((f & 7) << 2);
    }
    // two register sets of global loads: slab x travels in set x & 1, requested at the top of slab x - 3 and stored into stage x % 3
    // during the second block of slab x - 2 (1.5 slabs = ~2 us in flight; a slab is 2560 MFMA cycles)
    v_f32x4 ra[2][4], rw[2][3];
    v_f32x4 acc[2][V16_NG];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int gi = 0; gi < V16_NG; ++gi) acc[rb][gi] = v_f32x4{0.f, 0.f, 0.f, 0.f};
    const int ns = a.R / 32;
    auto load_slab = [&](int s, v_f32x4 (&xa)[4], v_f32x4 (&xw)[3]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const v_f32x4*>(ap[i] + s * 32);
#pragma unroll
        for (int i = 0; i < 3; ++i) xw[i] = *reinterpret_cast<const v_f32x4*>(wp[i] + s * 32);
    };
    auto store_slab = [&](float* As, const v_f32x4 (&xa)[4], const v_f32x4 (&xw)[3]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<v_f32x4*>(As + (f >> 3) * V16_LD + ((f & 7) << 2)) = xa[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = tid + 256 * i;                          // (f >= 640: rows 208 .. 223 of the stage, never read)
            *reinterpret_cast<v_f32x4*>(As + (128 + (f >> 3)) * V16_LD + ((f & 7) << 2)) = xw[i];
        }
    };
    // fragments of one 16-deep block: A rows wave * 32 + rb * 16 + l15, W rows gi * 16 + l15, this lane group's 4 k
    v_f32x4 fa[2][2], fb[2][V16_NG];
    auto read_frags = [&](const float* As, int kb, v_f32x4 (&xa)[2], v_f32x4 (&xb)[V16_NG]) {
        const float* Bs = As + 128 * V16_LD;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) xa[rb] = *reinterpret_cast<const v_f32x4*>(As + (wave * 32 + rb * 16 + l15) * V16_LD + kb * 16 + g4);
#pragma unroll
        for (int gi = 0; gi < V16_NG; ++gi) xb[gi] = *reinterpret_cast<const v_f32x4*>(Bs + (gi * 16 + l15) * V16_LD + kb * 16 + g4);
    };
    // ---- prologue: slabs 0 and 1 in LDS, slab 2 requested, the first block's fragments requested
    load_slab(0, ra[0], rw[0]);
    if (ns > 1) load_slab(1, ra[1], rw[1]);
    store_slab(smem, ra[0], rw[0]);
    if (ns > 2) load_slab(2, ra[0], rw[0]);
    if (ns > 1) store_slab(smem + STAGE, ra[1], rw[1]);
    __syncthreads();
    read_frags(smem, 0, fa[0], fb[0]);
    VP_STAMP(1);
    int st_cur = 0;                                                 // stage of the slab being multiplied (s % 3)
    auto slab = [&](int s, auto set_tag, auto load_tag, auto store_tag, auto next_tag) {
        constexpr int SET = decltype(set_tag)::value;                // == s & 1: the register set slab s + 2 sits in, and slab s + 3 does not
        constexpr bool LOAD = decltype(load_tag)::value, STORE = decltype(store_tag)::value, NEXT = decltype(next_tag)::value;   // slabs s + 3 / s + 2 / s + 1 exist
        const int st_nx = st_cur == NST - 1 ? 0 : st_cur + 1, st_st = st_cur == 0 ? NST - 1 : st_cur - 1;      // (s + 1) % 3, (s + 2) % 3
        const float* As = smem + st_cur * STAGE;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {                            // two 16-deep blocks per slab: 40 MFMAs each
            if (kb == 0) {
                if (LOAD) load_slab(s + 3, ra[SET ^ 1], rw[SET ^ 1]);
                read_frags(As, 1, fa[1], fb[1]);
            } else {
                if (STORE) store_slab(smem + st_st * STAGE, ra[SET], rw[SET]);
                if (NEXT) read_frags(smem + st_nx * STAGE, 0, fa[0], fb[0]);     // (stored two slabs ago, visible since the last barrier)
            }
            // (the NEXT reads overwrite fa[0] / fb[0]: block 0's MFMAs are program-ordered in front of them; kb == 1 multiplies set 1)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int gi = 0; gi < V16_NG; ++gi)
                        acc[rb][gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kb][rb][kk], fb[kb][gi][kk], acc[rb][gi], 0, 0, 0);
            if (kb == 0) vt_interleave<40, 2 + V16_NG, LOAD ? 7 : 0>();
            else vt_interleave<40, (STORE ? 7 : 0) + (NEXT ? 2 + V16_NG : 0), 0>();
            __builtin_amdgcn_sched_barrier(0);
        }
        // workgroup barrier that orders LDS only: __syncthreads() also waits for every outstanding GLOBAL load (vmcnt(0)), i.e. for the
        // slab requested at the top of this very slab -- the two register sets exist so that it does not have to be back yet
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        st_cur = st_nx;
    };
    {
        using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
        using Y = std::true_type; using N = std::false_type;
        int s = 0;
        for (; s + 4 < ns; s += 2) { slab(s, T0{}, Y{}, Y{}, Y{}); slab(s + 1, T1{}, Y{}, Y{}, Y{}); }
        const int rem = ns - s;                                     // 1 .. 4, s even
        if (rem == 4) { slab(s, T0{}, Y{}, Y{}, Y{}); slab(s + 1, T1{}, N{}, Y{}, Y{}); slab(s + 2, T0{}, N{}, N{}, Y{}); slab(s + 3, T1{}, N{}, N{}, N{}); }
        else if (rem == 3) { slab(s, T0{}, N{}, Y{}, Y{}); slab(s + 1, T1{}, N{}, N{}, Y{}); slab(s + 2, T0{}, N{}, N{}, N{}); }
        else if (rem == 2) { slab(s, T0{}, N{}, N{}, Y{}); slab(s + 1, T1{}, N{}, N{}, N{}); }
        else slab(s, T0{}, N{}, N{}, N{});
    }
    VP_STAMP(2);
    VP_STAMP(3);
    // ---- epilogue in registers.  MFMA result layout: acc[rb][gi][r] = row wave * 32 + rb * 16 + g4 + r, column n0 + gi * 16 + l15:
    // a row's 80 columns are the five values of each of the 16 lanes of one DPP row
    float bv[V16_NG]; bool cok[V16_NG];
#pragma unroll
    for (int gi = 0; gi < V16_NG; ++gi) {
        const int col = n0 + gi * 16 + l15;
        cok[gi] = col < a.V;
        bv[gi] = cok[gi] ? a.bias[col] : 0.f;
    }
    const int sper = rs_per(ntiles), slot = rs_slot(blockIdx.x, sper);
    // phase 1, branch-free (the eight rows' reductions interleave): x = logit in place of the accumulator, row statistics
    float smx[8], se1[8], set_[8]; int smc[8];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY; int mc = 0x7fffffff;
#pragma unroll
            for (int gi = 0; gi < V16_NG; ++gi) {
                const float x = cok[gi] ? acc[rb][gi][r] + bv[gi] : -INFINITY;
                acc[rb][gi][r] = x;
                const bool up = x > m;                                // (ascending columns: the first maximum stays)
                mc = up ? n0 + gi * 16 + l15 : mc;
                m = up ? x : m;
            }
            const float mx = rs_row_max(m);
            mc = rs_row_min(m == mx ? mc : 0x7fffffff);
            float e1 = 0.f, et = 0.f;
#pragma unroll
            for (int gi = 0; gi < V16_NG; ++gi) {
                const float d = acc[rb][gi][r] - mx;                  // (-inf for columns past V: exp -> 0)
                const float e = __expf(d);
                e1 += e;
                et += __expf(d * a.inv_t);
            }
            smx[rb * 4 + r] = mx; smc[rb * 4 + r] = mc; se1[rb * 4 + r] = rs_row_sum(e1); set_[rb * 4 + r] = rs_row_sum(et);
        }
    }
    // phase 2: the stores
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 32 + rb * 16 + g4 + r, q = rb * 4 + r;
            if (row < a.B && l15 == 0) {
                v_f32x4 pv = {smx[q], se1[q], set_[q], __int_as_float(smc[q])};
                *reinterpret_cast<v_f32x4*>(a.part + ((size_t)row * (sper << 4) + slot) * 4) = pv;
            }
        }
    }
    if (wave * 32 < a.wr_rows) {                                    // (wave-uniform: the greedy half's waves store nothing)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 32 + rb * 16 + g4 + r;
                float* dst = a.logits + (size_t)row * a.V + n0 + l15;
#pragma unroll
                for (int gi = 0; gi < V16_NG; ++gi) if (row < a.wr_rows && cok[gi]) dst[gi * 16] = acc[rb][gi][r];
            }
        }
    }
    VP_STAMP(4);
}

constexpr int ST = 256;                   // threads of a selection workgroup
__global__ void __launch_bounds__(ST) roll_select_kernel(RollSelectArgs q) {
    XG_CHAIN_PRIO();
    const RollStepArgs& a = q.r;
    __shared__ float redf[ST / 64]; __shared__ int redi[ST / 64]; __shared__ double wave_tot[ST / 64];
    __shared__ double s_d[2]; __shared__ float s_f[2];
    __shared__ int s_owner; __shared__ int64_t s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool second = b >= a.split;
    const int mode = (second && a.mode == XG_ROLLOUT_SAMPLE) ? XG_ROLLOUT_GREEDY : a.mode;
    int32_t* maxf = a.maxf + (second ? 1 : 0);
    const int nt = q.ntiles, per = (nt + ST - 1) / ST;           // tiles per thread, contiguous: thread tid owns [tid per, ..)
    const int sper = rs_per(nt);
    const v_f32x4* pr = reinterpret_cast<const v_f32x4*>(q.part) + (size_t)b * (sper << 4);
    constexpr int PMAX = 4;                                     // V <= 32 * 4 * 256 (launcher)
    v_f32x4 pv[PMAX];
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int j = tid * per + i;
        const v_f32x4 z = {-INFINITY, 0.f, 0.f, 0.f};
        pv[i] = (i < per && j < nt) ? pr[rs_slot(j, sper)] : z;
        const int c = __float_as_int(pv[i][3]);
        if (i < per && j < nt && (pv[i][0] > best || (pv[i][0] == best && c < bi))) { best = pv[i][0]; bi = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { redf[wave] = best; redi[wave] = bi; }
    __syncthreads();
    best = redf[0]; bi = redi[0];
    for (int i = 1; i < ST / 64; ++i)
        if (redf[i] > best || (redf[i] == best && redi[i] < bi)) { best = redf[i]; bi = redi[i]; }
    const float mx = best;
    // this thread's share of sum exp(x - mx) and of the temperature-scaled weights, tile by tile
    const float invt = 1.0f / a.temperature;
    float w1[PMAX], wt[PMAX];
    double c1 = 0.0, ct = 0.0;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const bool on = i < per && tid * per + i < nt;
        w1[i] = on ? pv[i][1] * __expf(pv[i][0] - mx) : 0.f;
        wt[i] = on ? pv[i][2] * __expf((pv[i][0] - mx) * invt) : 0.f;
        c1 += (double)w1[i]; ct += (double)wt[i];
    }
    // block-wide inclusive scan of the temperature-scaled chunk sums (tile order = column order) + block sum of the plain ones
    double inc = ct, tot1 = c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot1 += __shfl_xor(tot1, o, 64);
    __syncthreads();
    if (lane == 63) wave_tot[wave] = inc;
    if (lane == 0) redf[wave] = (float)tot1;
    if (tid == 0) s_owner = ST - 1;
    __syncthreads();
    double off = 0.0, tot = 0.0; float se = 0.f;
#pragma unroll
    for (int i = 0; i < ST / 64; ++i) { const double wtot = wave_tot[i]; if (i < wave) off += wtot; tot += wtot; se += redf[i]; }
    inc += off;
    const float lse = mx + logf(se);
    int64_t tk; float xtk = mx;
    if (mode == XG_ROLLOUT_GREEDY) {
        tk = bi;
    } else if (mode == XG_ROLLOUT_REPLAY) {
        tk = a.forced[(size_t)b * a.fstride];
        tk = tk < 0 ? 0 : (tk >= a.V ? a.V - 1 : tk);
        xtk = a.logits[(size_t)b * a.V + tk];
    } else {
        // inverse CDF over w_v = exp((logit_v - max) / temperature) (:190-194): the first thread whose running sum passes the
        // target owns the draw, walks its tiles' sums to the tile that holds it, then that tile's 32 logits
        const double target = (double)a.uniforms[b] * tot;
        if (inc > target) atomicMin(&s_owner, tid);
        __syncthreads();
        if (tid == s_owner) {
            double run = inc - ct;
            int jt = min(nt, tid * per + per) - 1;               // (rounding: the last tile of the share if nothing passes)
            if (jt < tid * per) jt = nt - 1;
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                if (i < per && tid * per + i < nt) {
                    if (run + (double)wt[i] > target) { jt = tid * per + i; break; }
                    run += (double)wt[i];
                }
            }
            s_d[0] = run; s_d[1] = target; s_owner = jt;
        }
        __syncthreads();
        if (wave == 0) {     // the logits of that tile (<= 128: two per lane), a shuffle scan, the first column whose running sum passes
            const int jt = s_owner, v0 = jt * q.tw, nvalid = min(q.tw, a.V - v0);
            const float* xrow = a.logits + (size_t)b * a.V + v0;
            const bool on0 = lane < nvalid, on1 = lane + 64 < nvalid;
            const float x0 = on0 ? xrow[lane] : 0.f, x1 = on1 ? xrow[lane + 64] : 0.f;
            float e0 = on0 ? __expf((x0 - mx) * invt) : 0.f, e1s = on1 ? __expf((x1 - mx) * invt) : 0.f;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float u0 = __shfl_up(e0, o, 64), u1 = __shfl_up(e1s, o, 64);
                if (lane >= o) { e0 += u0; e1s += u1; }
            }
            const float base = (float)s_d[0], tf = (float)s_d[1];
            const float tot0 = __shfl(e0, 63, 64);
            const unsigned long long m0 = __ballot(on0 && base + e0 > tf);
            const unsigned long long m1 = __ballot(on1 && base + tot0 + e1s > tf);
            int pl;
            if (m0) pl = __ffsll((long long)m0) - 1;
            else if (m1) pl = 64 + __ffsll((long long)m1) - 1;
            else pl = nvalid - 1;
            const float xp = pl < 64 ? __shfl(x0, pl, 64) : __shfl(x1, pl - 64, 64);
            if (lane == 0) { s_tok = v0 + pl; s_f[0] = xp; }
        }
        __syncthreads();
        tk = s_tok; xtk = s_f[0];
    }
    if (tid == 0) {
        roll_bookkeep(a, b, mode, tk, xtk - lse, lse, maxf);
        s_tok = tk;
    }
    __syncthreads();
    const int64_t t2 = s_tok;
    for (int e = tid; e < a.E; e += ST) a.xt[(size_t)b * a.E + e] = a.table[(size_t)t2 * a.E + e];
}

__global__ void rollout_finalize_kernel(const int32_t* maxf, int32_t* n_steps, int Tm1, int nparts) {
    for (int i = 0; i < nparts; ++i) {
        const int m = maxf[i];                    // = max over the part's rows of their finishing step, or T
        n_steps[i] = m <= 0 ? Tm1 : (m - 1 < Tm1 ? m - 1 : Tm1);
    }
}
// dlogits row b = d * (onehot(tok) - softmax(logits)) from raw logits + lse, in place
__global__ void __launch_bounds__(RT) rollout_dlogits_lse_kernel(float* __restrict__ logits, const float* lse, const int64_t* tok,
                                                                   const float* dslp, int64_t dstride, int V, int B) {
    // grid (B, steps): step y's rows are logits[y*B + b], lse[y*B + b], tok[y*B + b], dslp[b*dstride + y]
    const int b = blockIdx.x, y = blockIdx.y;
    const size_t row = (size_t)y * B + b;
    const float d = dslp[(size_t)b * dstride + y], l = lse[row];
    const int64_t tk = tok[row];
    float* x = logits + row * V;
    if (d == 0.f) {                                   // finished rows (RewardCriterion's mask): nothing to exponentiate
        for (int v = threadIdx.x; v < V; v += RT) x[v] = 0.f;
        return;
    }
    for (int v = threadIdx.x; v < V; v += RT) x[v] = d * ((v == tk ? 1.f : 0.f) - expf(x[v] - l));
}

__global__ void ss_select_kernel(const int64_t* seq, int T, int t, int B, const float* u_sel, float ss_prob,
                                 const int64_t* sampled, int64_t* tok) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool take = sampled != nullptr && u_sel[b] < ss_prob;      // sample_mask = sample_prob < ss_prob (:91)
    tok[b] = take ? sampled[b] : seq[(size_t)b * T + t];
}

// SAModel.py:200-215: unfinished &= it>0 ; it *= unfinished ; append ; n = first t with none unfinished
__global__ void rollout_book_kernel(int t, int B, int Tm1, int replay, const int64_t* tok, const float* tok_logp,
                                    float* unfinished, int64_t* seq, float* seq_logp, int32_t* n_steps,
                                    int32_t* alive) {
    // single workgroup; alive[0] = 1 while the reference's loop would still be running
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int64_t it = tok[b];
        float u = (t == 1) ? (it > 0 ? 1.f : 0.f) : unfinished[b] * (it > 0 ? 1.f : 0.f);
        unfinished[b] = u;
        if (u > 0.f) atomicOr(&any, 1);
    }
    __syncthreads();
    const int was_alive = alive[0];
    const int now_alive = was_alive && (any || replay);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (now_alive) {
            seq[(size_t)b * Tm1 + (t - 1)] = replay ? tok[b] : (unfinished[b] > 0.f ? tok[b] : 0);
            seq_logp[(size_t)b * Tm1 + (t - 1)] = tok_logp[b];
        } else {
            seq[(size_t)b * Tm1 + (t - 1)] = 0;
            seq_logp[(size_t)b * Tm1 + (t - 1)] = 0.f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (was_alive && !now_alive) n_steps[0] = t - 1;      // reference breaks before appending (SAModel.py:206-207)
        alive[0] = now_alive;
    }
}

__global__ void __launch_bounds__(RT) rollout_dlogits_kernel(const float* __restrict__ logp, const int64_t* tok,
                                                               const float* dslp, int64_t dstride, float* dlogits, int V) {
    const int b = blockIdx.x;
    const float d = dslp[(size_t)b * dstride];
    const int64_t tk = tok[b];
    const float* y = logp + (size_t)b * V;
    float* dx = dlogits + (size_t)b * V;
    for (int v = threadIdx.x; v < V; v += RT) dx[v] = d * ((v == tk ? 1.f : 0.f) - expf(y[v]));
}

}  // namespace

int xgk_log_softmax(hipStream_t st, const float* in, int ldin, float* out, int ldout, int rows, int V, int inner,
                    int outer, bool permute) {
    if (rows <= 0) return XG_OK;
    const int npt = xg_cdiv(V, RT), pm = permute ? 1 : 0;
    if (npt <= 8) hipLaunchKernelGGL((log_softmax_reg_kernel<8>), dim3(rows), dim3(RT), 0, st, in, ldin, out, ldout, V, inner, outer, pm);
    else if (npt <= 16) hipLaunchKernelGGL((log_softmax_reg_kernel<16>), dim3(rows), dim3(RT), 0, st, in, ldin, out, ldout, V, inner, outer, pm);
    else if (npt <= 24) hipLaunchKernelGGL((log_softmax_reg_kernel<24>), dim3(rows), dim3(RT), 0, st, in, ldin, out, ldout, V, inner, outer, pm);
    else hipLaunchKernelGGL(log_softmax_kernel, dim3(rows), dim3(RT), 0, st, in, ldin, out, ldout, V, inner, outer, pm);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_log_softmax_bwd(hipStream_t st, const float* dlogp, const float* logp, int ldp, float* dlogits, int ldd,
                        int rows, int V, int inner, int outer, int permute) {
    if (rows <= 0) return XG_OK;
    const int npt = xg_cdiv(V, RT);
    if (npt <= 8) hipLaunchKernelGGL((log_softmax_bwd_reg_kernel<8>), dim3(rows), dim3(RT), 0, st, dlogp, logp, ldp, dlogits, ldd, V, inner, outer, permute);
    else if (npt <= 16) hipLaunchKernelGGL((log_softmax_bwd_reg_kernel<16>), dim3(rows), dim3(RT), 0, st, dlogp, logp, ldp, dlogits, ldd, V, inner, outer, permute);
    else if (npt <= 24) hipLaunchKernelGGL((log_softmax_bwd_reg_kernel<24>), dim3(rows), dim3(RT), 0, st, dlogp, logp, ldp, dlogits, ldd, V, inner, outer, permute);
    else hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(rows), dim3(RT), 0, st, dlogp, logp, ldp, dlogits, ldd, V, inner, outer, permute);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_nll_fwd(hipStream_t st, const float* logp, const int64_t* target, const float* mask, const float* mask2,
                int B, int T, int V, int roll, float* out2) {
    if (hipMemsetAsync(out2, 0, 2 * sizeof(float), st) != hipSuccess) return XG_EHIP;
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(xg_cdiv(B * T, 256)), dim3(256), 0, st, logp, target, mask, mask2, B, T, V,
                       roll, out2);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_nll_bwd(hipStream_t st, const int64_t* target, const float* mask, const float* mask2, int B, int T, int V,
                int roll, const float* sums, float scale, const float* scale_dev, float* dlogp) {
    if (hipMemsetAsync(dlogp, 0, sizeof(float) * (size_t)B * T * V, st) != hipSuccess) return XG_EHIP;
    hipLaunchKernelGGL(nll_bwd_kernel, dim3(xg_cdiv(B * T, 256)), dim3(256), 0, st, target, mask, mask2, B, T, V, roll,
                       sums, scale, scale_dev, dlogp);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_xent_fwd(hipStream_t st, const float* logits, int ld, const int64_t* seq, const float* mask,
                 const float* mask2, int B, int T, int V, int roll, float* lse, float* sums2, int row0, int nrows,
                 bool zero_sums) {
    if (zero_sums && hipMemsetAsync(sums2, 0, 2 * sizeof(float), st) != hipSuccess) return XG_EHIP;
    if (nrows < 0) nrows = B * T - row0;
    if (nrows <= 0) return XG_OK;
    const bool al = ((uintptr_t)logits % 16 == 0) && ld % 4 == 0;
#define XG_XENT_REG(NV4_) hipLaunchKernelGGL((xent_fwd_reg_kernel<NV4_>), dim3(nrows), dim3(RT), 0, st, logits, ld, seq, mask, mask2, B, T, V, roll, lse, sums2, row0)
    if (al && V <= 2 * 4096) XG_XENT_REG(2);
    else if (al && V <= 5 * 4096) XG_XENT_REG(5);
    else if (al && V <= 8 * 4096) XG_XENT_REG(8);
    else hipLaunchKernelGGL(xent_fwd_kernel, dim3(nrows), dim3(RT), 0, st, logits, ld, seq, mask, mask2, B, T, V, roll, lse, sums2, row0);
#undef XG_XENT_REG
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_xent_bwd(hipStream_t st, float* logits_inout, int ld, const int64_t* seq, const float* mask,
                 const float* mask2, int B, int T, int V, int roll, const float* lse, const float* sums2,
                 const float* scale_dev, float scale, int row0, int nrows, unsigned short* d16) {
    if (nrows < 0) nrows = B * T - row0;
    if (nrows <= 0) return XG_OK;
    const bool al = ((uintptr_t)logits_inout % 16 == 0) && ld % 4 == 0 && (!d16 || (uintptr_t)d16 % 8 == 0);
#define XG_XENT_BREG(NV4_) do { \
        if (d16) hipLaunchKernelGGL((xent_bwd_reg_kernel<NV4_, true>), dim3(nrows), dim3(RT), 0, st, logits_inout, ld, seq, mask, mask2, B, T, V, \
                                    roll, lse, sums2, scale_dev, scale, row0, d16); \
        else hipLaunchKernelGGL((xent_bwd_reg_kernel<NV4_, false>), dim3(nrows), dim3(RT), 0, st, logits_inout, ld, seq, mask, mask2, B, T, V, \
                                roll, lse, sums2, scale_dev, scale, row0, d16); } while (0)
    if (al && V > 1024 && V <= 2 * 4096) XG_XENT_BREG(2);
    else if (al && V > 1024 && V <= 5 * 4096) XG_XENT_BREG(5);
    else if (al && V > 1024 && V <= 8 * 4096) XG_XENT_BREG(8);
    else if (d16) hipLaunchKernelGGL(xent_bwd_kernel<true>, dim3(nrows), dim3(RT), 0, st, logits_inout, ld, seq, mask, mask2, B, T, V, roll,
                                     lse, sums2, scale_dev, scale, row0, d16);
    else hipLaunchKernelGGL(xent_bwd_kernel<false>, dim3(nrows), dim3(RT), 0, st, logits_inout, ld, seq, mask, mask2, B, T, V, roll,
                            lse, sums2, scale_dev, scale, row0, d16);
#undef XG_XENT_BREG
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_choose(hipStream_t st, const float* logp, int B, int V, int mode, const float* uniforms,
               const int64_t* forced, int64_t forced_stride, float temperature, int64_t* tok, float* tok_logp) {
    hipLaunchKernelGGL(choose_kernel, dim3(B), dim3(RT), 0, st, logp, V, mode, uniforms, forced, forced_stride,
                       temperature, tok, tok_logp);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_rollout_step(hipStream_t st, int B, const float* logits, const float* uniforms, const int64_t* forced,
                     int64_t fstride, const float* unf_prev, const float* table, int64_t* tok, float* tok_logp, float* unf,
                     float* lse, int64_t* seq, float* seq_logp, int32_t* maxf, float* xt, float temperature, int V, int E,
                     int t, int T, int mode, int split) {
    RollStepArgs a{logits, uniforms, forced, fstride, unf_prev, table, tok, tok_logp, unf, lse, seq, seq_logp, maxf, xt,
                   temperature, V, E, t, T, mode, split};
    const size_t lds = (size_t)V * sizeof(float);
    if (t >= 1 && lds <= 150 * 1024) {
        static std::atomic<unsigned> optin{0};        // > 64 KiB of dynamic LDS is opted into once per device
        XG_TRY(xg_lds_optin(optin, reinterpret_cast<const void*>(&rollout_step_kernel<true>), 150 * 1024));
        hipLaunchKernelGGL((rollout_step_kernel<true>), dim3(B), dim3(RT), lds, st, a);
    } else {
        hipLaunchKernelGGL((rollout_step_kernel<false>), dim3(B), dim3(RT), 0, st, a);
    }
    XG_CHECK_LAUNCH();
    return XG_OK;
}
// the vocabulary product of a rollout step with per-tile row statistics, then the token choice over them (see vocab_part_kernel).
// `part` = (B, ceil(V / 32), 4) floats of scratch.  Returns 1 when the shape is not this path's (caller takes the two old launches).
bool xgk_vocab_select_ok(int B, int R, int V, const float* H, int ldh, const float* W) {
    return B >= 1 && B <= 128 && R % 32 == 0 && R >= 32 && V >= 32 && V <= 32 * 4 * ST && ldh % 4 == 0 &&
           ((uintptr_t)H % 16) == 0 && ((uintptr_t)W % 16) == 0;
}
// tile width of the statistics: 80 columns (one 16 x 16 x 4 tile set per CU-sized workgroup) from a vocabulary of 4096 on, where the
// balance over the 256 CUs is what matters; 32 columns below (XG_VOCAB_TW=32 / 80 of the -DXG_DIAG build forces either)
int xgk_vocab_tile_width(int V) {
    static const int forced = xg_diag_env("XG_VOCAB_TW") ? atoi(xg_diag_env("XG_VOCAB_TW")) : 0;
    if (forced == 32 || forced == V16_TW) return forced;
    return V >= 4096 ? V16_TW : 32;
}
int xgk_vocab_part(hipStream_t st, int B, int R, int V, const float* H, int ldh, const float* W, const float* bias, float* logits,
                   int wr_rows, float* part, float temperature) {
    VocabPartArgs a{H, ldh, W, bias, B, R, V, logits, wr_rows, part, 1.0f / temperature};
    if (xgk_vocab_tile_width(V) == V16_TW) hipLaunchKernelGGL(vocab_part16_kernel, dim3(xg_cdiv(V, V16_TW)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(vocab_part_kernel, dim3(xg_cdiv(V, 32)), dim3(256), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
RollSelectArgs xgk_roll_select_args(const float* logits, const float* part, const float* uniforms, const int64_t* forced, int64_t fstride,
                                    const float* unf_prev, const float* table, int64_t* tok, float* tok_logp, float* unf, float* lse,
                                    int64_t* seq, float* seq_logp, int32_t* maxf, float* xt, float temperature, int V, int E, int t, int T,
                                    int mode, int split) {
    const int tw = xgk_vocab_tile_width(V);
    return RollSelectArgs{{logits, uniforms, forced, fstride, unf_prev, table, tok, tok_logp, unf, lse, seq, seq_logp, maxf, xt,
                           temperature, V, E, t, T, mode, split}, part, xg_cdiv(V, tw), tw};
}
int xgk_roll_select(hipStream_t st, int B, const float* logits, const float* part, const float* uniforms, const int64_t* forced,
                    int64_t fstride, const float* unf_prev, const float* table, int64_t* tok, float* tok_logp, float* unf,
                    float* lse, int64_t* seq, float* seq_logp, int32_t* maxf, float* xt, float temperature, int V, int E,
                    int t, int T, int mode, int split) {
    if (t < 1) return XG_EINVAL;
    const RollSelectArgs q = xgk_roll_select_args(logits, part, uniforms, forced, fstride, unf_prev, table, tok, tok_logp, unf, lse, seq,
                                                  seq_logp, maxf, xt, temperature, V, E, t, T, mode, split);
    hipLaunchKernelGGL(roll_select_kernel, dim3(B), dim3(ST), 0, st, q);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
#ifdef VP_TRACE
extern "C" int xg_debug_vp_trace(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vp_trace_buf), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
int xgk_rollout_finalize(hipStream_t st, const int32_t* maxf, int32_t* n_steps, int Tm1, int nparts) {
    hipLaunchKernelGGL(rollout_finalize_kernel, dim3(1), dim3(1), 0, st, maxf, n_steps, Tm1, nparts);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_rollout_dlogits_lse(hipStream_t st, float* logits, const float* lse, const int64_t* tok, const float* dslp,
                            int64_t dstride, int B, int V, int steps) {
    if (steps <= 0) return XG_OK;
    hipLaunchKernelGGL(rollout_dlogits_lse_kernel, dim3(B, steps), dim3(RT), 0, st, logits, lse, tok, dslp, dstride, V, B);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_ss_select(hipStream_t st, const int64_t* seq, int T, int t, int B, const float* u_sel, float ss_prob,
                  const int64_t* sampled, int64_t* tok) {
    hipLaunchKernelGGL(ss_select_kernel, dim3(xg_cdiv(B, 256)), dim3(256), 0, st, seq, T, t, B, u_sel, ss_prob, sampled, tok);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_rollout_book(hipStream_t st, int t, int B, int Tm1, int replay, const int64_t* tok, const float* tok_logp,
                     float* unfinished, int64_t* seq, float* seq_logp, int32_t* n_steps, int32_t* alive) {
    hipLaunchKernelGGL(rollout_book_kernel, dim3(1), dim3(256), 0, st, t, B, Tm1, replay, tok, tok_logp, unfinished, seq,
                       seq_logp, n_steps, alive);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
int xgk_rollout_dlogits(hipStream_t st, const float* logp, const int64_t* tok, const float* dslp, int64_t dstride,
                        float* dlogits, int B, int V) {
    hipLaunchKernelGGL(rollout_dlogits_kernel, dim3(B), dim3(RT), 0, st, logp, tok, dslp, dstride, dlogits, V);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

extern "C" int xg_nll_fwd(void* stream, const float* logp, const int64_t* target, const float* mask,
                          const float* mask2, int B, int T, int V, int roll, float* out) {
    if (!logp || !target || !mask || !out || B <= 0 || T <= 0 || V <= 0) return XG_EINVAL;
    return xgk_nll_fwd((hipStream_t)stream, logp, target, mask, mask2, B, T, V, roll, out);
}
extern "C" int xg_nll_bwd(void* stream, const int64_t* target, const float* mask, const float* mask2, int B, int T,
                          int V, int roll, const float* sums, float scale, const float* scale_dev, float* dlogp) {
    if (!target || !mask || !sums || !dlogp || B <= 0 || T <= 0 || V <= 0) return XG_EINVAL;
    return xgk_nll_bwd((hipStream_t)stream, target, mask, mask2, B, T, V, roll, sums, scale, scale_dev, dlogp);
}

extern "C" int xg_reward_fwd(void* stream, const float* slp, int ld_slp, const int64_t* seq, int ld_seq, const float* reward,
                             int rs_b, int rs_t, const int32_t* n_dev, int m, int L, float* out) {
    if (!slp || !seq || !reward || !out || m <= 0 || L <= 0 || ld_slp < L || ld_seq < L) return XG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, 2 * sizeof(float), st) != hipSuccess) return XG_EHIP;
    hipLaunchKernelGGL(reward_fwd_kernel, dim3(xg_cdiv(m * L, 256)), dim3(256), 0, st, slp, ld_slp, seq, ld_seq, reward, rs_b, rs_t,
                       n_dev, m, L, out);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
extern "C" int xg_reward_bwd(void* stream, const int64_t* seq, int ld_seq, const float* reward, int rs_b, int rs_t,
                             const int32_t* n_dev, int m, int L, const float* sums, const float* scale_dev, float* dslp, int ld_d) {
    if (!seq || !reward || !sums || !dslp || m <= 0 || L <= 0 || ld_seq < L || ld_d < L) return XG_EINVAL;
    hipLaunchKernelGGL(reward_bwd_kernel, dim3(xg_cdiv(m * L, 256)), dim3(256), 0, (hipStream_t)stream, seq, ld_seq, reward, rs_b,
                       rs_t, n_dev, m, L, sums, scale_dev, dslp, ld_d);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
