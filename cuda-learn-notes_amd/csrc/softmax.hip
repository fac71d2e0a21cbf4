// Softmax, 11 rungs. Replaces reference kernels/softmax/softmax.cu:102-390 (kernels) and
// :776-885 (bindings).
//   softmax_f32 / softmax_f32x4         : ONE distribution over the whole 1-D tensor, no max
//                                         subtraction (softmax.cu:102-148). The reference sums
//                                         block partials with atomicAdd + __threadfence and divides
//                                         in the same launch (racy by construction); here it is two
//                                         stream-ordered launches: sum(exp) -> *total, then divide.
//   *_per_token                          : one row per workgroup; "safe" subtracts the row max,
//                                         "online" merges (m, d) pairs (softmax.cu:22-41, :314-390).
// Row kernels keep the row in registers: 1 HBM read + 1 HBM write per element.
#include "rowwise.cuh"

using namespace rowwise;

namespace {

enum Mode { UNSAFE = 0, SAFE = 1, ONLINE = 2 };

template <int VEC>
__global__ __launch_bounds__(256) void exp_sum_kernel(const float* __restrict__ x, float* __restrict__ total,
                                                      long long n) {
  __shared__ float scratch[16];
  float s = 0.f;
  const long long nvec = n / VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const Pack<float, VEC> p = *reinterpret_cast<const Pack<float, VEC>*>(x + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += __expf(p.v[e]);
  }
  if (blockIdx.x == 0)
    for (long long i = nvec * VEC + threadIdx.x; i < n; i += blockDim.x) s += __expf(x[i]);
  s = block_sum<256>(s, scratch);
  if (threadIdx.x == 0) atomicAdd(total, s);
}
template <int VEC>
__global__ __launch_bounds__(256) void exp_div_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ total, long long n) {
  const float inv = 1.0f / *total;
  const long long nvec = n / VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    Pack<float, VEC> p = *reinterpret_cast<const Pack<float, VEC>*>(x + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) p.v[e] = __expf(p.v[e]) * inv;
    *reinterpret_cast<Pack<float, VEC>*>(y + i * VEC) = p;
  }
  if (blockIdx.x == 0)
    for (long long i = nvec * VEC + threadIdx.x; i < n; i += blockDim.x) y[i] = __expf(x[i]) * inv;
}

// n <= 64 Ki elements (the reference script's N = 128 * 128, softmax.py:66): ONE workgroup of 1024 lanes does both passes -- sum(exp), block
// reduce, divide -- in one launch; x is re-read from L1 / L2 for the second pass. A launch costs more than this kernel runs: one launch instead
// of two (profiles/r04_scripts_vs_torch_before.log: 12.9 us against torch's 6.5 us with the two-launch form). *total still receives the sum.
constexpr long long SOFTMAX_ONE_BLOCK_MAX = 65536;
template <int VEC>
__global__ __launch_bounds__(1024) void softmax_one_block_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ total,
                                                                 long long n) {
  __shared__ float scratch[16];
  const long long nvec = n / VEC;
  float s = 0.f;
  for (long long i = threadIdx.x; i < nvec; i += 1024) {
    const Pack<float, VEC> p = *reinterpret_cast<const Pack<float, VEC>*>(x + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += __expf(p.v[e]);
  }
  for (long long i = nvec * VEC + threadIdx.x; i < n; i += 1024) s += __expf(x[i]);
  s = block_sum<1024>(s, scratch);
  if (threadIdx.x == 0) *total = s;
  const float inv = 1.0f / s;
  for (long long i = threadIdx.x; i < nvec; i += 1024) {
    Pack<float, VEC> p = *reinterpret_cast<const Pack<float, VEC>*>(x + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) p.v[e] = __expf(p.v[e]) * inv;
    *reinterpret_cast<Pack<float, VEC>*>(y + i * VEC) = p;
  }
  for (long long i = nvec * VEC + threadIdx.x; i < n; i += 1024) y[i] = __expf(x[i]) * inv;
}

template <int VEC>
int launch_global(const void* x, void* y, void* total, long long n, hipStream_t st) {
  if (!x || !y || !total || n <= 0) return CLN_ERR_BAD_ARG;
  if (n <= SOFTMAX_ONE_BLOCK_MAX) {  // (in place is fine: a lane re-reads only the elements it then overwrites itself)
    CLN_LAUNCH((softmax_one_block_kernel<VEC>), dim3(1), dim3(1024), 0, st, (const float*)x, (float*)y, (float*)total, n);
    return cln_check_launch();
  }
  const int grid = cln_stream_grid(n / VEC + 1, 256, 8LL * n);
  CLN_LAUNCH((exp_sum_kernel<VEC>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)total, n);
  if (cln_check_launch() != CLN_OK) return CLN_ERR_LAUNCH;  // a failed first pass would leave total = 0 -> inf
  CLN_LAUNCH((exp_div_kernel<VEC>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y,
                     (const float*)total, n);
  return cln_check_launch();
}

// row reductions: a row of one wave (wave-per-row groups, rpw > 1, or a 64-thread workgroup) stays inside the wave; larger rows go through LDS
__device__ __forceinline__ float row_sum_g(float v, float* scratch, int rpw) { return rpw > 1 ? wave_sum(v) : block_sum_rt(v, scratch); }
__device__ __forceinline__ float row_max_g(float v, float* scratch, int rpw) { return rpw > 1 ? wave_max(v) : block_max_rt(v, scratch); }

template <typename T, int VEC, int MAXV, int MODE, bool FULL>
__global__ void softmax_row_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int stream_nt, int rpw) {
  __shared__ float scratch[16];
  const RowPos rp = row_pos(rpw);
  const size_t off = rp.row * H;
  RowRegs<T, VEC, MAXV> r;
  r.template load<FULL>(x + off, H, -INFINITY, rp.tid, rp.tpr);
  float m = 0.f, d = 0.f;
  if constexpr (MODE == UNSAFE) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        r.x[i][e] = __expf(r.x[i][e]);
        d += r.x[i][e];
      }
    d = row_sum_g(d, scratch, rpw);
  } else if constexpr (MODE == SAFE) {
    m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) m = fmaxf(m, r.x[i][e]);
    m = row_max_g(m, scratch, rpw);
    // fp16 rows: 2^(x log2e - m log2e) as ONE fma + v_exp_f32 instead of sub, mul, exp -- x carries 11 bits and y is rounded to 11, the fma's
    // 2^-24 |x log2e| is far below both; the fp32 rungs keep the reference's expf(x - max) (softmax.cu:219-235) operation for operation
    [[maybe_unused]] const float ml2 = m * 1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if constexpr (sizeof(T) == 2) r.x[i][e] = __builtin_amdgcn_exp2f(fmaf(r.x[i][e], 1.4426950408889634f, -ml2));
        else r.x[i][e] = __expf(r.x[i][e] - m);
        d += r.x[i][e];
      }
    d = row_sum_g(d, scratch, rpw);
  } else {
    // online normaliser: per-thread (m, d), then one max-reduction and one rescaled sum-reduction
    m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float v = r.x[i][e];
        const float mn = fmaxf(m, v);
        d = d * __expf(m - mn) + __expf(v - mn);  // m == mn == -inf only for all-padding lanes
        m = mn;
      }
    if (m == -INFINITY) d = 0.f;
    const float mg = row_max_g(m, scratch, rpw);
    d = row_sum_g((m == -INFINITY) ? 0.f : d * __expf(m - mg), scratch, rpw);
    m = mg;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) r.x[i][e] = __expf(r.x[i][e] - m);
  }
  const float inv = 1.0f / d;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.x[i][e] *= inv;
  r.template store<FULL>(y + off, H, stream_nt, rp.tid, rp.tpr);
}

template <typename T, int VEC, int MODE>
int launch_rows(const void* x, void* y, int S, int H, hipStream_t st) {
  if (!x || !y || S <= 0 || H <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned(x, sizeof(T) * VEC) || !cln_aligned(y, sizeof(T) * VEC)) return CLN_ERR_BAD_ARG;
  if (H % VEC) return CLN_ERR_UNSUPPORTED;
  const int nt = row_threads(H, VEC), vpt = vecs_per_thread(H, VEC, nt), rpw = rows_per_wg(nt, S);
#define CALL(MV, FL) \
  CLN_LAUNCH((softmax_row_kernel<T, VEC, MV, MODE, FL>), dim3(S / rpw), dim3(nt * rpw), 0, st, (const T*)x, (T*)y, H, cln_stream_nt(2LL * S * H * (long long)sizeof(T)), rpw)
  ROWWISE_DISPATCH_MAXV_FULL(vpt, H, nt, VEC, CALL);
#undef CALL
  return cln_check_launch();
}

}  // namespace

// (x, y, total_workspace[1] (zeroed by the caller), n, stream) -- reference softmax_f32[x4](x, y)
CLN_API int softmax_f32(const void* x, void* y, void* total, long long n, void* stream) {
  return launch_global<1>(x, y, total, n, (hipStream_t)stream);
}
CLN_API int softmax_f32x4(const void* x, void* y, void* total, long long n, void* stream) {
  if (!cln_aligned16(x) || !cln_aligned16(y)) return CLN_ERR_BAD_ARG;
  return launch_global<4>(x, y, total, n, (hipStream_t)stream);
}

// (x, y, S rows, H cols, stream)
#define CLN_SM(name, T, VEC, MODE)                                                  \
  CLN_API int name(const void* x, void* y, int S, int H, void* stream) {            \
    return launch_rows<T, VEC, MODE>(x, y, S, H, (hipStream_t)stream);              \
  }
CLN_SM(softmax_f32_per_token, float, 1, UNSAFE)
CLN_SM(softmax_f32x4_per_token, float, 4, UNSAFE)
CLN_SM(safe_softmax_f32_per_token, float, 1, SAFE)
CLN_SM(safe_softmax_f32x4_per_token, float, 4, SAFE)
CLN_SM(safe_softmax_f16_f32_per_token, half_t, 1, SAFE)
CLN_SM(safe_softmax_f16x2_f32_per_token, half_t, 2, SAFE)
CLN_SM(safe_softmax_f16x8_pack_f32_per_token, half_t, 8, SAFE)
CLN_SM(online_safe_softmax_f32_per_token, float, 1, ONLINE)
CLN_SM(online_safe_softmax_f32x4_pack_per_token, float, 4, ONLINE)
