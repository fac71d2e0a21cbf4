// Vendor YARDSTICKS for the bandwidth rows of bench.py (round 6): what the ROCm stack itself reaches on the same bytes, on the same
// box, in the same run -- so that "the cold-HBM stream time of a 33 MB tensor is not ours to cut" has a number beside it.
//   cln_yardstick_copy          hipMemcpyDtoDAsync (the runtime's blit kernel): 1 read + 1 write per byte, the floor of every 1R + 1W row kernel
//   cln_yardstick_reduce_f32    rocprim::reduce (plus<float>) over fp32
//   cln_yardstick_reduce_f16    rocprim::reduce over fp16 through a transform iterator (fp32 accumulation): the block_all_reduce_sum_f16*_f32 rungs
// Lives in libcln_amd_vendor.so (comparison rows only; optional at build). Not part of the reference surface and not in include/cln_amd.h.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstring>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "common.h"

CLN_API int cln_yardstick_copy(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src) return CLN_ERR_BAD_ARG;
  return hipMemcpyDtoDAsync((hipDeviceptr_t)dst, (hipDeviceptr_t)const_cast<void*>(src), bytes, (hipStream_t)stream) == hipSuccess ? CLN_OK : ((void)hipGetLastError(), CLN_ERR_LAUNCH);
}

// tmp == NULL: *tmp_bytes receives the scratch size rocprim wants for n elements (no launch)
CLN_API int cln_yardstick_reduce_f32(const void* x, void* y, long long n, void* tmp, size_t* tmp_bytes, void* stream) {
  if (!tmp_bytes || n <= 0) return CLN_ERR_BAD_ARG;
  size_t need = *tmp_bytes;
  if (!tmp) need = 0;
  const hipError_t e = rocprim::reduce(tmp, need, reinterpret_cast<const float*>(x), reinterpret_cast<float*>(y), 0.0f, (size_t)n, rocprim::plus<float>(), (hipStream_t)stream);
  if (!tmp) *tmp_bytes = need;
  return e == hipSuccess ? CLN_OK : ((void)hipGetLastError(), CLN_ERR_VENDOR);
}

struct HalfToFloat {
  __host__ __device__ float operator()(const __half& h) const { return __half2float(h); }
};
CLN_API int cln_yardstick_reduce_f16(const void* x, void* y, long long n, void* tmp, size_t* tmp_bytes, void* stream) {
  if (!tmp_bytes || n <= 0) return CLN_ERR_BAD_ARG;
  size_t need = *tmp_bytes;
  if (!tmp) need = 0;
  auto it = rocprim::make_transform_iterator(reinterpret_cast<const __half*>(x), HalfToFloat());
  const hipError_t e = rocprim::reduce(tmp, need, it, reinterpret_cast<float*>(y), 0.0f, (size_t)n, rocprim::plus<float>(), (hipStream_t)stream);
  if (!tmp) *tmp_bytes = need;
  return e == hipSuccess ? CLN_OK : ((void)hipGetLastError(), CLN_ERR_VENDOR);
}
