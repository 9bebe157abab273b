// clip_gradient + Adam in one pass (reference caption_src/myutils.py:79-85 elementwise clamp,
// caption_src/starttrain.py:76,137 torch.optim.Adam with default betas/eps).
// HBM-bound: 4 streams read (p,g,m,v), 3 written; one element per thread (wave-coalesced 256 B rows; 5.6 TB/s measured).
#include "xg_common.h"
#include "xg_kernels.h"

namespace {
template <bool ZERO>   // ZERO: the gradient is left at zero instead of clamped (the next iteration's optimizer.zero_grad())
__global__ void clip_adam_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, float lr, float b1, float b2, float eps, float wd, float bc1,
                                 float sqrt_bc2, float clip) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    if (ZERO) g[i] = 0.f;
    else if (clip > 0.f) g[i] = gi;                                       // clamp_ is in place on .grad
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}
// ---- the same update with its step-dependent scalars in DEVICE memory (hyper[0] = lr, [1] = 1 - beta1^t, [2] = sqrt(1 - beta2^t),
// [3] = t as int bits): a launch whose arguments never change, i.e. one that can sit in a HIP graph and be replayed
__global__ void adam_tick_kernel(float* hyper, float b1, float b2) {
    const int t = __float_as_int(hyper[3]) + 1;
    hyper[3] = __int_as_float(t);
    hyper[1] = (float)(1.0 - pow((double)b1, (double)t));
    hyper[2] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
template <bool ZERO>
__global__ void clip_adam_dev_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                     float* __restrict__ v, const float* __restrict__ hyper, float b1, float b2, float eps, float wd,
                                     float clip) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = hyper[0], bc1 = hyper[1], sqrt_bc2 = hyper[2];
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    if (ZERO) g[i] = 0.f;
    else if (clip > 0.f) g[i] = gi;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}
}  // namespace

extern "C" int xg_adam_tick(void* stream, float* hyper, float beta1, float beta2) {
    if (!hyper) return XG_EINVAL;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, beta1, beta2);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
extern "C" int xg_clip_adam_dev(void* stream, int64_t n, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                const float* hyper, float beta1, float beta2, float eps, float weight_decay, float clip, int zero_grad) {
    if (n < 0 || !hyper || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return XG_EINVAL;
    if (n == 0) return XG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (zero_grad) hipLaunchKernelGGL(clip_adam_dev_kernel<true>, dim3((unsigned)xg_cdiv64(n, 256)), dim3(256), 0, st, n, param, grad,
                                      exp_avg, exp_avg_sq, hyper, beta1, beta2, eps, weight_decay, clip);
    else hipLaunchKernelGGL(clip_adam_dev_kernel<false>, dim3((unsigned)xg_cdiv64(n, 256)), dim3(256), 0, st, n, param, grad, exp_avg,
                            exp_avg_sq, hyper, beta1, beta2, eps, weight_decay, clip);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

int xgk_clip_adam(hipStream_t st, int64_t n, float* p, float* g, float* m, float* v, float lr, float b1, float b2,
                  float eps, float wd, int step, float clip, bool zero_grad) {
    if (n <= 0) return XG_OK;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    if (zero_grad) hipLaunchKernelGGL(clip_adam_kernel<true>, dim3((unsigned)xg_cdiv64(n, 256)), dim3(256), 0, st, n, p, g, m, v,
                                      lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2), clip);
    else hipLaunchKernelGGL(clip_adam_kernel<false>, dim3((unsigned)xg_cdiv64(n, 256)), dim3(256), 0, st, n, p, g, m, v, lr, b1,
                            b2, eps, wd, (float)bc1, (float)sqrt(bc2), clip);
    XG_CHECK_LAUNCH();
    return XG_OK;
}

extern "C" int xg_clip_adam(void* stream, int64_t n, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int step, float clip) {
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return XG_EINVAL;
    return xgk_clip_adam((hipStream_t)stream, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                         step, clip, false);
}

extern "C" int xg_clip_adam_zero(void* stream, int64_t n, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, float clip) {
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return XG_EINVAL;
    return xgk_clip_adam((hipStream_t)stream, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                         step, clip, true);
}
