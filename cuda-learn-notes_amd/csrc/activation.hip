// Activation family: relu / sigmoid / gelu / swish / elu / hardswish / hardshrink, six rungs
// each (f32, f32x4, f16, f16x2, f16x8, f16x8_pack) = 42 exported functions `void f(Tensor x, Tensor y)`.
// Reference: kernels/relu/relu.cu:20-150, sigmoid/sigmoid.cu:19-180, gelu/gelu.cu:19-230, swish/swish.cu:18-160,
// elu/elu.cu:19-170, hardswish/hardswish.cu:12-190, hardshrink/hardshrink.cu:18-180 (same binding macro in each).
//
// gfx950 design: ONE streaming template (HBM-bound, 1R + 1W per element) parameterised by the op and by the
// per-lane access the rung name states (4 B, 16 B; 2 B, 4 B, 8 x 2 B, 16 B); capped grid-stride grid.
// Arithmetic is fp32 for every rung (the reference's f16 rungs use half intrinsics: hexp, __hdiv ...): the
// result is rounded to fp16 once, which is at least as close to the script's torch column as the reference's own
// half arithmetic. Constants follow the reference kernels: sigmoid / gelu clamp their argument to
// +-88.3762626647949 (sigmoid.cu:19-20, gelu.cu:19-20), gelu is the tanh approximation (gelu.cu:46-48), elu alpha
// = 1 (elu.cu:19), hardswish thresholds +-3 (hardswish.cu:12-13), hardshrink lambda = 0.5 (hardshrink.cu:19).
#include "common.h"

namespace {

struct Relu { static __device__ __forceinline__ float f(float x) { return fmaxf(0.f, x); } };
// 1 / d as ONE v_rcp_f32 (1 ulp) instead of the IEEE division sequence hipcc emits for `/` without fast-math (v_div_scale x2, v_rcp, 4-5 fma,
// v_div_fmas, v_div_fixup: ~11 VALU per element -- at 8 halves per lane the f16x8 rungs of hardswish / sigmoid / swish became VALU-bound:
// hardswish f16x8_pack 4096x2048 6.9 us vs relu 4.9 us and torch 4.8 us, profiles/r04_scripts_vs_torch_before.log). The reference builds with
// --use_fast_math, i.e. the same approximation; all four stay within 9 % of the parity tolerance (2e-6 rel + 1e-6 abs against fp64).
static __device__ __forceinline__ float rcp1(float d) { return __builtin_amdgcn_rcpf(d); }
struct Sigmoid {
  static __device__ __forceinline__ float f(float x) {
    x = fminf(fmaxf(x, -88.3762626647949f), 88.3762626647949f);
    return rcp1(1.0f + __expf(-x));
  }
};
struct Gelu {
  // tanh form (reference gelu.cu GELU_OPS, torch.nn.GELU("tanh")): 0.5 x (1 + tanh u) = x / (1 + e^(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3) --
  // one exponential and one reciprocal instead of tanhf's ~25-instruction expansion (f16x8_pack 4096^2: 15.5 us, VALU-bound, with tanhf)
  static __device__ __forceinline__ float f(float x) {
    x = fminf(fmaxf(x, -88.3762626647949f), 88.3762626647949f);
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x * rcp1(1.0f + __expf(-2.0f * u));
  }
};
struct Swish { static __device__ __forceinline__ float f(float x) { return x * rcp1(1.0f + __expf(-x)); } };
struct Elu { static __device__ __forceinline__ float f(float x) { return x > 0.f ? x : (__expf(x) - 1.f); } };
struct HardSwish {
  // the reference's piecewise form (hardswish.cu:37-45: x >= 3 -> x, x <= -3 -> 0, else x (x + 3) / 6) as two selects over the middle branch.
  // Round 4 shipped the branch-free x * med3(x + 3, 0, 6) / 6 (4 VALU): it returned NaN for x = -inf (-inf * 0), +inf instead of x above
  // FLT_MAX / 6 and was 1 ulp off x for x >= 3 -- the selects cost 3 more VALU per element and keep the reference's edges.
  // Round 5: one select instead of two -- t = med3(x + 3, 0, 6) * (1/6) is EXACTLY 1.0f for x >= 3 (6 * fp32(1/6) = 1.0000000298 rounds to 1), so x * t
  // returns x itself there (also above FLT_MAX / 6: the clamp comes before the product), and only x <= -3 needs the select (t = 0: -inf * 0 would be NaN).
  // 6 VALU per element instead of 7; the middle branch is x * ((x + 3) / 6), within an fp32 rounding of the reference's x (x + 3) / 6.
  static __device__ __forceinline__ float f(float x) {
    const float t = __builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * (1.0f / 6.0f);
    return x <= -3.f ? 0.f : x * t;
  }
};
struct HardShrink {
  static __device__ __forceinline__ float f(float x) { return (x > 0.5f || x < -0.5f) ? x : 0.f; }
};

__device__ __forceinline__ float ld(float v) { return v; }
__device__ __forceinline__ float ld(half_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T st(float v);
template <> __device__ __forceinline__ float st<float>(float v) { return v; }
template <> __device__ __forceinline__ half_t st<half_t>(float v) { return (half_t)v; }

// VEC elements per lane; one VEC*sizeof(T)-byte access when CHUNK == VEC, else VEC / CHUNK accesses of CHUNK elements
// (f16x8 = four half2 accesses in the reference, f32x4 = one float4).
// Walk of the NARROW rungs (2- / 4-byte accesses; round 6, as elementwise.hip): block-contiguous, no loop -- workgroup b owns the 256 K consecutive packs
// from 256 K b on, every lane issues its K loads, then its K stores, K = 16: 64 bytes in flight per lane. (In rounds 1-5 these rungs walked the capped
// grid-stride loop with ONE pack in flight per lane: relu_f16 16.7 us, relu_f16x8 -- four half2 accesses -- 18.3 us at [4096,4096]; now 13.7 / 12.8.)
template <typename Op, typename T, int VEC, int CHUNK, int K>
__global__ __launch_bounds__(256) void unary_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, int stream_nt) {
  typedef T chunk_t __attribute__((ext_vector_type(CHUNK)));
  constexpr int NC = VEC / CHUNK;
  const long long nvec = n / VEC;
  const long long base = (long long)blockIdx.x * (256 * K) + threadIdx.x;
  if (base + (K - 1) * 256 < nvec) {  // every pack of this lane exists (all workgroups but the last)
    if constexpr (VEC == 1) {
      T v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = x[base + k * 256];
#pragma unroll
      for (int k = 0; k < K; ++k) cln_store_stream(y + base + k * 256, st<T>(Op::f(ld(v[k]))), stream_nt);
    } else {
      chunk_t v[K][NC];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) v[k][c] = *reinterpret_cast<const chunk_t*>(x + (base + k * 256) * VEC + c * CHUNK);
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
          for (int e = 0; e < CHUNK; ++e) v[k][c][e] = st<T>(Op::f(ld(v[k][c][e])));
          cln_store_stream(reinterpret_cast<chunk_t*>(y + (base + k * 256) * VEC + c * CHUNK), v[k][c], stream_nt);
        }
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const long long i = base + k * 256;
      if (i < nvec) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) y[i * VEC + e] = st<T>(Op::f(ld(x[i * VEC + e])));
      }
    }
  }
  if (VEC > 1 && blockIdx.x == 0) {  // ragged tail (the reference requires N % VEC == 0)
    for (long long i = nvec * VEC + threadIdx.x; i < n; i += 256) y[i] = st<T>(Op::f(ld(x[i])));
  }
}

// The 16-byte rungs keep the capped grid-stride walk of rounds 1-5 (cln_stream_grid: 256 CUs x 32 workgroups, one trip per thread above 512 MB of traffic): on
// tensors that sit in L2 / the Infinity Cache -- the reference scripts' shapes, one buffer set re-used -- it is 1-10 % AHEAD of the block-contiguous form
// (profiles/r06_scripts_vs_torch.log of the two forms: [4096,2048] f16 4.7-6.4 us against 4.9-6.9), on rotating HBM sets 1-3 % behind it: the scripts decide.
template <typename Op, typename T, int VEC>
__global__ __launch_bounds__(256) void unary_stride_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, int stream_nt) {
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const long long nvec = n / VEC;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    vec_t v = *reinterpret_cast<const vec_t*>(x + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = st<T>(Op::f(ld(v[e])));
    cln_store_stream(reinterpret_cast<vec_t*>(y + i * VEC), v, stream_nt);
  }
  if (blockIdx.x == 0) {  // ragged tail (the reference requires N % VEC == 0)
    for (long long i = nvec * VEC + threadIdx.x; i < n; i += 256) y[i] = st<T>(Op::f(ld(x[i])));
  }
}

template <typename Op, typename T, int VEC, int CHUNK>
int launch_unary(const void* x, void* y, long long n, hipStream_t st_) {
  if (!x || !y || n < 0) return CLN_ERR_BAD_ARG;
  if (n == 0) return CLN_OK;
  if (sizeof(T) * CHUNK >= 16 && (!cln_aligned16(x) || !cln_aligned16(y))) return CLN_ERR_BAD_ARG;
  // An "unpacked" rung (f16x8: eight halves per lane moved as four half2 accesses) runs the kernel of its ACCESS width: access c of a lane is the
  // lane's pack in the c-th 256-pack row of the workgroup's block, so every access instruction of a wave covers 256 contiguous bytes. (Rounds 1-5 gave a lane eight CONSECUTIVE halves, as the reference kernel does: four instructions that each touch
  // 4 of every 16 bytes -- relu_f16x8 18.3 us against 14.2 for relu_f16x2 at [4096,4096], tools/rung_survey.py.)
  // Narrow rungs (2- / 4-byte accesses): 64 bytes of loads in flight per lane = 16 packs per lane (the survey's relu_f16x2 ran 14.4 us with 4 packs
  // per lane and 12.9 with 16). The 16-byte rungs: unary_stride_kernel above.
  constexpr int AB = (int)sizeof(T) * CHUNK;                      // bytes per access
  constexpr int KB = AB >= 16 ? 4 : (64 / AB > 16 ? 16 : 64 / AB);
  const long long nvec = n / CHUNK, traffic = 2LL * n * (long long)sizeof(T);
  if constexpr (AB >= 16) {
    const int grid = cln_stream_grid(n / VEC + 1, 256, traffic);
    CLN_LAUNCH((unary_stride_kernel<Op, T, VEC>), dim3(grid), dim3(256), 0, st_, (const T*)x, (T*)y, n, cln_stream_nt(traffic));
  } else if (nvec < 1024 * KB) {
    CLN_LAUNCH((unary_kernel<Op, T, CHUNK, CHUNK, 1>), dim3((unsigned)((nvec + 255) / 256 + (nvec == 0))), dim3(256), 0, st_, (const T*)x, (T*)y, n, cln_stream_nt(traffic));
  } else {
    CLN_LAUNCH((unary_kernel<Op, T, CHUNK, CHUNK, KB>), dim3((unsigned)((nvec + 256 * KB - 1) / (256 * KB))), dim3(256), 0, st_, (const T*)x, (T*)y, n, cln_stream_nt(traffic));
  }
  return cln_check_launch();
}

}  // namespace

// (x, y, n_elements, stream) -- reference `void <op>_<rung>(torch::Tensor x, torch::Tensor y)`
#define CLN_UN(op, Op)                                                                                             \
  CLN_API int op##_f32(const void* x, void* y, long long n, void* s) { return launch_unary<Op, float, 1, 1>(x, y, n, (hipStream_t)s); }        \
  CLN_API int op##_f32x4(const void* x, void* y, long long n, void* s) { return launch_unary<Op, float, 4, 4>(x, y, n, (hipStream_t)s); }      \
  CLN_API int op##_f16(const void* x, void* y, long long n, void* s) { return launch_unary<Op, half_t, 1, 1>(x, y, n, (hipStream_t)s); }       \
  CLN_API int op##_f16x2(const void* x, void* y, long long n, void* s) { return launch_unary<Op, half_t, 2, 2>(x, y, n, (hipStream_t)s); }     \
  CLN_API int op##_f16x8(const void* x, void* y, long long n, void* s) { return launch_unary<Op, half_t, 8, 2>(x, y, n, (hipStream_t)s); }     \
  CLN_API int op##_f16x8_pack(const void* x, void* y, long long n, void* s) { return launch_unary<Op, half_t, 8, 8>(x, y, n, (hipStream_t)s); }
CLN_UN(relu, Relu)
CLN_UN(sigmoid, Sigmoid)
CLN_UN(gelu, Gelu)
CLN_UN(swish, Swish)
CLN_UN(elu, Elu)
CLN_UN(hardswish, HardSwish)
CLN_UN(hardshrink, HardShrink)
