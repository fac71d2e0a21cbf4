// Second vendor comparison row: hipBLASLt (BASELINE.md's C3 target reads "rocBLAS/hipBLASLt"; the reference's vendor row is
// cuBLAS through cublasGemmEx, kernels/hgemm/cublas/hgemm_cublas.cu:15-84 -- hipBLASLt is the ROCm library a maintainer
// would reach for next). Lives in libcln_amd_vendor.so beside the rocBLAS row; NOT a reference name, hence the cln_ prefix.
//   int cln_hgemm_hipblaslt_nn / _tn(a, b, c, M, N, K, stream)      fp16 in / out, fp32 accumulate
// One plan per (layout, M, N, K) is cached: descriptors, the heuristic's top algorithm and its workspace.
#include <hipblaslt/hipblaslt.h>
#include <mutex>
#include <vector>
#include "common.h"

namespace {

struct LtPlan {
  int layout, M, N, K;
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws_bytes = 0;
};

hipblasLtHandle_t g_lt = nullptr;
void* g_ws = nullptr;
size_t g_ws_bytes = 0;
std::vector<LtPlan> g_plans;
std::mutex g_mu;
constexpr size_t MAX_WS = 64u << 20;

// row-major C[M,N] = A[M,K] B  <=>  column-major C^T[N,M] = op(Bcm)[N,K] * Acm[K,M]
// NN: b row-major [K,N] = column-major [N,K], ld N, no transpose; TN: b storage [N,K] = column-major [K,N], ld K, transposed
LtPlan* find_or_make(int layout, int M, int N, int K) {
  for (LtPlan& p : g_plans)
    if (p.layout == layout && p.M == M && p.N == N && p.K == K) return &p;
  if (!g_lt && hipblasLtCreate(&g_lt) != HIPBLAS_STATUS_SUCCESS) return nullptr;
  LtPlan p;
  p.layout = layout, p.M = M, p.N = N, p.K = K;
  // every failure path below releases what this call created (nothing is cached on failure, so a retry must not leak again)
  auto fail = [&]() -> LtPlan* {
    if (p.la) hipblasLtMatrixLayoutDestroy(p.la);
    if (p.lb) hipblasLtMatrixLayoutDestroy(p.lb);
    if (p.lc) hipblasLtMatrixLayoutDestroy(p.lc);
    if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
    return nullptr;
  };
  if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return p.desc = nullptr, fail();
  const int32_t op_b = layout ? HIPBLAS_OP_T : HIPBLAS_OP_N, op_a = HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_b, sizeof(op_b));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_a, sizeof(op_a));
  bool ok = true;
  if (layout) ok &= hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16F, K, N, K) == HIPBLAS_STATUS_SUCCESS;
  else ok &= hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16F, N, K, N) == HIPBLAS_STATUS_SUCCESS;
  ok &= hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16F, K, M, K) == HIPBLAS_STATUS_SUCCESS;
  ok &= hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16F, N, M, N) == HIPBLAS_STATUS_SUCCESS;
  if (!ok) return fail();
  hipblasLtMatmulPreference_t pref;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return fail();
  const uint64_t max_ws = MAX_WS;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws));
  hipblasLtMatmulHeuristicResult_t res[1];
  int got = 0;
  const hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(g_lt, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &got);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (hs != HIPBLAS_STATUS_SUCCESS || got < 1 || res[0].state != HIPBLAS_STATUS_SUCCESS) return fail();
  p.algo = res[0].algo, p.ws_bytes = res[0].workspaceSize;
  if (p.ws_bytes > MAX_WS) return fail();
  if (!g_ws) {  // ONE workspace of the preference's maximum, allocated once: earlier plans never see theirs freed under a queued matmul
    if (hipMalloc(&g_ws, MAX_WS) != hipSuccess) return g_ws = nullptr, fail();
    g_ws_bytes = MAX_WS;
  }
  g_plans.push_back(p);
  return &g_plans.back();
}

int lt_gemm(int layout, const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  LtPlan* p = find_or_make(layout, M, N, K);
  if (!p) return CLN_ERR_VENDOR;
  const float alpha = 1.0f, beta = 0.0f;
  const hipblasStatus_t s = hipblasLtMatmul(g_lt, p->desc, &alpha, b, p->la, a, p->lb, &beta, c, p->lc, c, p->lc, &p->algo,
                                            g_ws, p->ws_bytes, (hipStream_t)stream);
  return s == HIPBLAS_STATUS_SUCCESS ? CLN_OK : CLN_ERR_VENDOR;
}

}  // namespace

// b row-major [K,N]
CLN_API int cln_hgemm_hipblaslt_nn(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  return lt_gemm(0, a, b, c, M, N, K, stream);
}
// b storage [N,K] (column-major [K,N])
CLN_API int cln_hgemm_hipblaslt_tn(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  return lt_gemm(1, a, b, c, M, N, K, stream);
}
