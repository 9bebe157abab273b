// EXPERIMENT (diag library only, XG_BLASLT=<mask>, see xg_gemm.hip: xgk_gemm_cs): plain fp32 products through the vendor library
// (hipBLASLt), to measure what its kernels -- 11-30 % faster than xg_gemm.hip's on five of nine shapes when they run ALONE --
// are worth inside the training iteration.  Answer (round 5, tools/ubench/blaslt_iter.sh): nothing, 5.52-5.56 ms against
// 5.53-5.54 ms, whichever class of products is routed; the product library does not link hipBLASLt.
// Row-major C[M,N] = op(A) op(B) is the column-major product C^T = op(B)^T op(A)^T.
// Plans (operation + layouts + the heuristic's first algorithm) are cached per shape; no workspace is used (the heuristic is asked
// for algorithms that need none), so concurrent calls on different streams share nothing but the handle.
#include "xg_common.h"
#include "xg_kernels.h"
#include <hipblaslt/hipblaslt.h>
#include <mutex>
#include <unordered_map>
#include <cstring>

namespace {

struct Key {
    int ta, tb, M, N, K, lda, ldb, ldc, epi;
    bool operator==(const Key& o) const { return std::memcmp(this, &o, sizeof(Key)) == 0; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        const int* p = reinterpret_cast<const int*>(&k);
        size_t h = 1469598103934665603ull;
        for (size_t i = 0; i < sizeof(Key) / sizeof(int); ++i) { h ^= (size_t)(unsigned)p[i]; h *= 1099511628211ull; }
        return h;
    }
};
struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    bool ok = false;
};

std::mutex g_mu;
hipblasLtHandle_t g_handle = nullptr;
bool g_handle_failed = false;
std::unordered_map<Key, Plan, KeyHash> g_plans;

#define LT_TRY(x) do { if ((x) != HIPBLAS_STATUS_SUCCESS) return false; } while (0)

bool make_plan(const Key& k, Plan& p) {
    // column-major problem: m' = N, n' = M, k' = K; A' = our B, B' = our A
    const hipblasOperation_t opa = k.tb ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = k.ta ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    LT_TRY(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
    LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
    const uint32_t epi = (uint32_t)k.epi;
    LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    if (k.epi & HIPBLASLT_EPILOGUE_BIAS) {
        const int32_t bt = HIP_R_32F;
        LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    LT_TRY(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_32F, k.tb ? k.K : k.N, k.tb ? k.N : k.K, k.ldb));
    LT_TRY(hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_32F, k.ta ? k.M : k.K, k.ta ? k.K : k.M, k.lda));
    LT_TRY(hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, k.N, k.M, k.ldc));
    hipblasLtMatmulPreference_t pref = nullptr;
    LT_TRY(hipblasLtMatmulPreferenceCreate(&pref));
    const uint64_t wsb = 0;
    bool ok = hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) == HIPBLAS_STATUS_SUCCESS;
    hipblasLtMatmulHeuristicResult_t res[1];
    int n = 0;
    ok = ok && hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &n) == HIPBLAS_STATUS_SUCCESS && n >= 1;
    (void)hipblasLtMatmulPreferenceDestroy(pref);
    if (!ok || res[0].workspaceSize != 0) return false;      // (no workspace is passed to hipblasLtMatmul)
    p.algo = res[0].algo;
    p.ok = true;
    return true;
}

}  // namespace

// returns XG_OK when the product was enqueued, 1 when the library is unavailable / has no kernel for it (the caller's own kernel
// runs), an error code otherwise
int xgk_blaslt_gemm(hipStream_t st, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, bool relu, bool accumulate) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    Key k{};
    k.ta = transA; k.tb = transB; k.M = M; k.N = N; k.K = K; k.lda = lda; k.ldb = ldb; k.ldc = ldc;
    k.epi = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS) : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
    Plan* p = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (g_handle_failed) return 1;
        if (!g_handle && hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) { g_handle = nullptr; g_handle_failed = true; return 1; }
        auto it = g_plans.find(k);
        if (it == g_plans.end()) {
            Plan np;
            (void)make_plan(k, np);              // a failed plan is cached too (ok == false): the caller's kernel from then on
            it = g_plans.emplace(k, np).first;
        }
        p = &it->second;                         // (unordered_map: references stay valid across later insertions)
    }
    if (!p->ok) return 1;
    // the bias pointer is per call (the plan's descriptor is shared): set it under the lock, together with the enqueue
    const float alpha = 1.f, beta = accumulate ? 1.f : 0.f;
    std::lock_guard<std::mutex> lock(g_mu);
    if (bias && hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return 1;
    const hipblasStatus_t rc = hipblasLtMatmul(g_handle, p->desc, &alpha, B, p->la, A, p->lb, &beta, C, p->lc, C, p->lc, &p->algo, nullptr, 0, st);
    return rc == HIPBLAS_STATUS_SUCCESS ? XG_OK : XG_EHIP;
}
