// Row normalisations with scalar gain/bias: layer-norm (8 rungs) and rms-norm (9 rungs).
//
// Replaces reference kernels/layer-norm/layer_norm.cu:53-414 (kernels) / :732-814 (bindings) and
// kernels/rms-norm/rms_norm.cu:54-370 / :457-813.
// Arithmetic follows the reference KERNELS (these are their drop-ins):
//   layer-norm: mean = sum(x)/K; rstd = rsqrt(sum((x-mean)^2) / (K + 1e-5)); y = (x-mean)*rstd*g + b
//               (population variance, eps added to K -- layer_norm.cu:69, :103)
//   rms-norm:   rstd = rsqrt(sum(x^2)/K + 1e-5); y = x*rstd*g            (rms_norm.cu:54-70)
// Statistics are always fp32 here (the reference's *_f16 rungs keep them in fp16); both agree with
// the script's torch oracles (layer_norm.py:25-29, rms_norm.py:26-31) within fp16 tolerance.
// One workgroup per row, row register-resident: 1 read + 1 write of HBM per element.
#include "rowwise.cuh"

using namespace rowwise;

namespace {

// Row statistics: lanes reduce their own elements, wave64 DPP/permlane all-reduce, lane 0 of every wave parks
// its partial in LDS, and after ONE barrier every thread folds the <= 16 wave partials itself (same order
// everywhere -> identical result, no broadcast hop). Each scratch array is written once per kernel, so no
// protecting barrier is needed; layer-norm keeps the reference kernels' two-pass mean / variance
// (layer_norm.cu:53-110) and therefore uses two scratch arrays and two barriers.
__device__ __forceinline__ float row_sum(float v, float* scratch, int rpw) {
  v = wave_sum(v);
  const int nw = rpw > 1 ? 1 : (int)(blockDim.x >> 6);  // wave-per-row groups (rowwise.cuh rows_per_wg): the row is one wave
  if (nw > 1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    v = scratch[0];
    for (int i = 1; i < nw; ++i) v += scratch[i];
  }
  return v;
}

template <typename T, int VEC, int MAXV, bool FULL>
__global__ void layer_norm_kernel(const T* __restrict__ x, T* __restrict__ y, float g, float b, int K, int stream_nt, int rpw) {
  __shared__ float scratch[2][16];
  const RowPos rp = row_pos(rpw);
  const size_t off = rp.row * K;
  RowRegs<T, VEC, MAXV> r;
  r.template load<FULL>(x + off, K, 0.f, rp.tid, rp.tpr);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += r.x[i][e];
  const float mean = row_sum(s, scratch[0], rpw) / (float)K;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int col = (i * rp.tpr + rp.tid) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = (FULL || col < K) ? (r.x[i][e] - mean) : 0.f;
      r.x[i][e] = d;
      v += d * d;
    }
  }
  const float a = rsqrtf(row_sum(v, scratch[1], rpw) / ((float)K + 1e-5f)) * g;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.x[i][e] = fmaf(r.x[i][e], a, b);
  r.template store<FULL>(y + off, K, stream_nt, rp.tid, rp.tpr);
}

template <typename T, int VEC, int MAXV, bool FULL>
__global__ void rms_norm_kernel(const T* __restrict__ x, T* __restrict__ y, float g, int K, int stream_nt, int rpw) {
  __shared__ float scratch[16];
  const RowPos rp = row_pos(rpw);
  const size_t off = rp.row * K;
  RowRegs<T, VEC, MAXV> r;
  r.template load<FULL>(x + off, K, 0.f, rp.tid, rp.tpr);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) v += r.x[i][e] * r.x[i][e];
  const float a = rsqrtf(row_sum(v, scratch, rpw) / (float)K + 1e-5f) * g;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.x[i][e] = r.x[i][e] * a;
  r.template store<FULL>(y + off, K, stream_nt, rp.tid, rp.tpr);
}

template <typename T, int VEC>
int launch_ln(const void* x, void* y, float g, float b, int N, int K, hipStream_t st) {
  if (!x || !y || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned(x, sizeof(T) * VEC) || !cln_aligned(y, sizeof(T) * VEC)) return CLN_ERR_BAD_ARG;
  if (K % VEC) return CLN_ERR_UNSUPPORTED;
  const int nt = row_threads(K, VEC), vpt = vecs_per_thread(K, VEC, nt), rpw = rows_per_wg(nt, N);
#define CALL(MV, FL)                                                                                          \
  CLN_LAUNCH((layer_norm_kernel<T, VEC, MV, FL>), dim3(N / rpw), dim3(nt * rpw), 0, st, (const T*)x, (T*)y, g, b, K, cln_stream_nt(2LL * N * K * (long long)sizeof(T)), rpw)
  ROWWISE_DISPATCH_MAXV_FULL(vpt, K, nt, VEC, CALL);
#undef CALL
  return cln_check_launch();
}
template <typename T, int VEC>
int launch_rms(const void* x, void* y, float g, int N, int K, hipStream_t st) {
  if (!x || !y || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned(x, sizeof(T) * VEC) || !cln_aligned(y, sizeof(T) * VEC)) return CLN_ERR_BAD_ARG;
  if (K % VEC) return CLN_ERR_UNSUPPORTED;
  const int nt = row_threads(K, VEC), vpt = vecs_per_thread(K, VEC, nt), rpw = rows_per_wg(nt, N);
#define CALL(MV, FL) \
  CLN_LAUNCH((rms_norm_kernel<T, VEC, MV, FL>), dim3(N / rpw), dim3(nt * rpw), 0, st, (const T*)x, (T*)y, g, K, cln_stream_nt(2LL * N * K * (long long)sizeof(T)), rpw)
  ROWWISE_DISPATCH_MAXV_FULL(vpt, K, nt, VEC, CALL);
#undef CALL
  return cln_check_launch();
}

}  // namespace

// (x, y, g, b, N rows, K cols, stream) -- reference `void layer_norm_*(Tensor x, Tensor y, float g, float b)`
#define CLN_LN(name, T, VEC)                                                                       \
  CLN_API int name(const void* x, void* y, float g, float b, int N, int K, void* stream) {         \
    return launch_ln<T, VEC>(x, y, g, b, N, K, (hipStream_t)stream);                               \
  }
CLN_LN(layer_norm_f32, float, 1)
CLN_LN(layer_norm_f32x4, float, 4)
CLN_LN(layer_norm_f16_f16, half_t, 1)
CLN_LN(layer_norm_f16x2_f16, half_t, 2)
CLN_LN(layer_norm_f16x8_f16, half_t, 8)
CLN_LN(layer_norm_f16x8_pack_f16, half_t, 8)
CLN_LN(layer_norm_f16x8_pack_f32, half_t, 8)
CLN_LN(layer_norm_f16_f32, half_t, 1)

// (x, y, g, N rows, K cols, stream) -- reference `void rms_norm_*(Tensor x, Tensor y, float g)`
#define CLN_RMS(name, T, VEC)                                                              \
  CLN_API int name(const void* x, void* y, float g, int N, int K, void* stream) {          \
    return launch_rms<T, VEC>(x, y, g, N, K, (hipStream_t)stream);                         \
  }
CLN_RMS(rms_norm_f32, float, 1)
CLN_RMS(rms_norm_f32x4, float, 4)
CLN_RMS(rms_norm_f16_f16, half_t, 1)
CLN_RMS(rms_norm_f16x2_f16, half_t, 2)
CLN_RMS(rms_norm_f16x8_f16, half_t, 8)
CLN_RMS(rms_norm_f16x8_f32, half_t, 8)
CLN_RMS(rms_norm_f16x8_pack_f16, half_t, 8)
CLN_RMS(rms_norm_f16x8_pack_f32, half_t, 8)
CLN_RMS(rms_norm_f16_f32, half_t, 1)
