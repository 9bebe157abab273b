// Internal helpers shared by the HIP translation units of libxgate_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/xgate.h"

#define XG_CHECK_LAUNCH()                                  \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return XG_EHIP; \
    } while (0)

#define XG_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != XG_OK) return _rc; \
    } while (0)

// ---- dropout hash: must stay bit-identical to oracle/paramgen.py:hash_u32 (test infrastructure) ----
__host__ __device__ __forceinline__ uint32_t xg_fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t xg_hash(uint32_t seed, uint32_t site, uint32_t step, uint32_t idx) {
    uint32_t h = xg_fmix32(idx + 0x9E3779B9u * (step + 1u));
    return xg_fmix32(h ^ (seed ^ (site * 0x632BE5ABu)));
}

// Dropout descriptor passed by value to kernels.  scale == 1 and thresh == 0 when inactive.
struct XgDrop {
    uint32_t seed, site, step, thresh;
    float scale;
};
__host__ __forceinline__ XgDrop xg_make_drop(const XgRun* run, uint32_t site, uint32_t step) {
    XgDrop d;
    d.seed = run->seed;
    d.site = site;
    d.step = step;
    if (run->train && run->drop_p > 0.f) {
        double t = (double)run->drop_p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
        d.scale = 1.0f / (1.0f - run->drop_p);
    } else {
        d.thresh = 0u;
        d.scale = 1.0f;
    }
    return d;
}
// multiplier (0 or scale) for flat element index idx
__device__ __forceinline__ float xg_keep(const XgDrop& d, uint32_t idx) {
    if (d.thresh == 0u) return 1.0f;
    return xg_hash(d.seed, d.site, d.step, idx) >= d.thresh ? d.scale : 0.0f;
}

// dropout sites (oracle/xgate_oracle.py header)
enum { XG_SITE_EMB_RGB = 0, XG_SITE_EMB_OPFL = 1, XG_SITE_GATE_RGB = 2, XG_SITE_GATE_OPFL = 3, XG_SITE_FUSION = 4,
       XG_SITE_DGATE = 5, XG_SITE_L1 = 6, XG_SITE_L2 = 7, XG_SITE_CLS = 8 };

// v_rcp_f32 (1 ulp) instead of an IEEE division: the division expands to ~10 instructions and the recurrent / attention
// kernels are bound by exactly this arithmetic (26 x 1536 tanh per video per step).
__device__ __forceinline__ float xg_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float xg_sigmoid(float x) { return xg_rcp(1.0f + __expf(-x)); }
// tanh(x) = 1 - 2 / (1 + 2^(2 log2(e) x)): v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma -- five instructions (the |x| / copysign
// form it replaces took nine, at the same ~1e-7 ABSOLUTE error; the attention kernels are bound by exactly this arithmetic:
// 26 x 1536 tanh per video per step).  Saturates correctly through exp2 -> inf / 0.
__device__ __forceinline__ float xg_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return 1.0f - 2.0f * xg_rcp(1.0f + e);
}

// Wave64 reductions on the DPP path (4 cross-lane adds inside each row of 16 + one readlane per row) instead of six
// ds_bpermute round trips.  Every lane of the wave must be active (all call sites are wave-uniform); the result is
// wave-uniform.
template <int CTRL>
__device__ __forceinline__ float xg_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float xg_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum(float v) {
    v += xg_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += xg_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += xg_dpp<0x141>(v);     // row_half_mirror
    v += xg_dpp<0x140>(v);     // row_mirror
    return (xg_readlane(v, 0) + xg_readlane(v, 16)) + (xg_readlane(v, 32) + xg_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, xg_dpp<0xB1>(v));
    v = fmaxf(v, xg_dpp<0x4E>(v));
    v = fmaxf(v, xg_dpp<0x141>(v));
    v = fmaxf(v, xg_dpp<0x140>(v));
    return fmaxf(fmaxf(xg_readlane(v, 0), xg_readlane(v, 16)), fmaxf(xg_readlane(v, 32), xg_readlane(v, 48)));
}

// Kernels that use more than 64 KiB of dynamic LDS must opt in once per kernel AND per device (hipFuncSetAttribute is
// per device).  `done` is a per-kernel bit set, one bit per device; thread-safe; devices >= 32 opt in on every launch.
static inline int xg_lds_optin(std::atomic<unsigned>& done, const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return XG_EHIP;
    const unsigned bit = dev >= 0 && dev < 32 ? 1u << dev : 0u;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return XG_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return XG_EHIP;
    done.fetch_or(bit, std::memory_order_release);
    return XG_OK;
}

// Kernels of the recurrent launch chains raise their waves' issue priority: they share CUs with background GEMM
// workgroups (xg_gemm.hip: XGK_GEMM_BG) whose older waves would otherwise win the arbitration (a chain step beside
// dW_logit: 134 us at equal priority, 86 us raised; 50 us alone).  -DXG_NO_CHAIN_PRIO for the comparison.
#ifdef XG_NO_CHAIN_PRIO
#define XG_CHAIN_PRIO() ((void)0)
#else
#define XG_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

// Diagnosis switches are read from the environment ONLY in a -DXG_DIAG build (lib/libxgate_hip_diag.so, used
// by tests / tools); the product library has none: xg_diag_env() is a constant there, every `static const ... =
// xg_diag_env(...)` folds to its default and the library keeps no mutable global state (include/xgate.h).
#ifdef XG_DIAG
#include <cstdlib>
static inline const char* xg_diag_env(const char* name) { return getenv(name); }
#else
static inline constexpr const char* xg_diag_env(const char*) { return nullptr; }
#endif

// -DXG_NULL_LAUNCH (measurement build, tools/host8_enqueue.py): every kernel launch of the library becomes the launch of an empty
// kernel on the same stream -- same host-side call sequence, events and stream waits, no GPU time behind it.  What the host loop
// costs can then be told apart from what it WAITS for when several launcher processes share one GPU.  Results are garbage.
#ifdef XG_NULL_LAUNCH
static __global__ void xg_null_kernel(int) {}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, ...) do { xg_null_kernel<<<dim3(1), dim3(64), 0, (stream_)>>>(0); } while (0)
#endif

static inline int xg_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t xg_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
