// Vendor-library comparison row: rocBLAS behind the reference's cuBLAS entry points
// (reference kernels/hgemm/cublas/hgemm_cublas.cu:13-84 handle + gemmEx calls, :222-261 bindings).
// Built into its own shared object (libcln_amd_vendor.so) so the hand-written kernel library has
// no vendor dependency; the handle is a process global created/destroyed by the driver exactly as
// in the reference (hgemm.py:111-112, :184-185).
#include <rocblas/rocblas.h>
#include "common.h"

static rocblas_handle g_handle = nullptr;

CLN_API int init_cublas_handle() {
  if (g_handle) return CLN_OK;
  return rocblas_create_handle(&g_handle) == rocblas_status_success ? CLN_OK : CLN_ERR_VENDOR;
}
CLN_API int destroy_cublas_handle() {
  if (!g_handle) return CLN_OK;
  rocblas_status s = rocblas_destroy_handle(g_handle);
  g_handle = nullptr;
  return s == rocblas_status_success ? CLN_OK : CLN_ERR_VENDOR;
}

static int gemm_ex(rocblas_operation opB, const void* a, const void* b, void* c, int M, int N, int K, int ldb,
                   void* stream) {
  if (!g_handle) return CLN_ERR_VENDOR;
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (rocblas_set_stream(g_handle, (hipStream_t)stream) != rocblas_status_success) return CLN_ERR_VENDOR;
  const float alpha = 1.0f, beta = 0.0f;
  // row-major C[M,N] = A[M,K] B  <=>  column-major C^T[N,M] = op(B)[N,K] * A^T[K,M]
  rocblas_status s = rocblas_gemm_ex(g_handle, opB, rocblas_operation_none, N, M, K, &alpha, b, rocblas_datatype_f16_r,
                                     ldb, a, rocblas_datatype_f16_r, K, &beta, c, rocblas_datatype_f16_r, N, c,
                                     rocblas_datatype_f16_r, N, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0,
                                     0);
  return s == rocblas_status_success ? CLN_OK : CLN_ERR_VENDOR;
}

// b row-major [K,N]
CLN_API int hgemm_cublas_tensor_op_nn(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  return gemm_ex(rocblas_operation_none, a, b, c, M, N, K, N, stream);
}
// b storage [N,K] (column-major [K,N])
CLN_API int hgemm_cublas_tensor_op_tn(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  return gemm_ex(rocblas_operation_transpose, a, b, c, M, N, K, K, stream);
}

// ---- SGEMM vendor rows (reference kernels/sgemm/sgemm_cublas.cu:15-120). The reference creates and destroys a
// cuBLAS handle inside every call; here a lazily created process-global handle serves both (and the HGEMM rows).
// sgemm_cublas_tf32 asks cuBLAS for TF32 tensor-op math; gfx950 has no TF32, so both names run the exact-f32
// rocBLAS SGEMM (which itself uses the f32 matrix instruction).
static int sgemm_vendor(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  if (!g_handle && init_cublas_handle() != CLN_OK) return CLN_ERR_VENDOR;
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (rocblas_set_stream(g_handle, (hipStream_t)stream) != rocblas_status_success) return CLN_ERR_VENDOR;
  const float alpha = 1.0f, beta = 0.0f;
  rocblas_status s = rocblas_sgemm(g_handle, rocblas_operation_none, rocblas_operation_none, N, M, K, &alpha,
                                   (const float*)b, N, (const float*)a, K, &beta, (float*)c, N);
  return s == rocblas_status_success ? CLN_OK : CLN_ERR_VENDOR;
}
CLN_API int sgemm_cublas(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  return sgemm_vendor(a, b, c, M, N, K, stream);
}
CLN_API int sgemm_cublas_tf32(const void* a, const void* b, void* c, int M, int N, int K, void* stream) {
  return sgemm_vendor(a, b, c, M, N, K, stream);
}
