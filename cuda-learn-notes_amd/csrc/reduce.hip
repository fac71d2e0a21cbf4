// block_all_reduce_sum: scalar sum of a whole tensor, 20 rungs over dtype / access width / pack
// accumulator. Replaces reference kernels/reduce/block_all_reduce.cu:42-687 (kernels) and
// :734-813 (bindings). HBM-bound: sizeof(T) bytes read per element.
//
// gfx950 design: capped grid-stride streaming (256 CUs x 16 workgroups), per-lane access width
// fixed by the rung name, wave64 xor-butterfly -> one LDS hop -> ONE atomic per workgroup
// (the reference's structure, re-derived for 64 lanes: NUM_WARPS = NT/64).
// "acc" in the rung name is the precision of the IN-PACK sum, as in the reference
// (e.g. f16x8_pack_f16 adds the 8 halves of a pack in fp16, block_all_reduce.cu:252-262);
// everything across packs / lanes / waves is fp32 (int32 for i8), result y is fp32 / int32.
#include "common.h"
#include "stream_scratch.h"

namespace {

struct F32 {};   struct F16 {};   struct BF16 {};   struct E4M3 {};   struct E5M2 {};   struct I8 {};

__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_rn(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// In-pack sum policies ---------------------------------------------------------------------------
template <typename IN, typename ACC, int VEC>
struct PackSum;

template <int VEC>
struct PackSum<F32, F32, VEC> {
  using elem = float;
  using out = float;
  static __device__ __forceinline__ float sum(const float* p) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += p[i];
    return s;
  }
};
template <int VEC>
struct PackSum<F16, F32, VEC> {
  using elem = half_t;
  using out = float;
  static __device__ __forceinline__ float sum(const half_t* p) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += (float)p[i];
    return s;
  }
};
template <int VEC>
struct PackSum<F16, F16, VEC> {
  using elem = half_t;
  using out = float;
  static __device__ __forceinline__ float sum(const half_t* p) {
    half_t s = p[0];
#pragma unroll
    for (int i = 1; i < VEC; ++i) s = s + p[i];  // fp16 adds (v_add_f16)
    return (float)s;
  }
};
template <int VEC>
struct PackSum<BF16, F32, VEC> {
  using elem = unsigned short;
  using out = float;
  static __device__ __forceinline__ float sum(const unsigned short* p) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += bf16_to_f32(p[i]);
    return s;
  }
};
template <int VEC>
struct PackSum<BF16, BF16, VEC> {
  using elem = unsigned short;
  using out = float;
  static __device__ __forceinline__ float sum(const unsigned short* p) {
    float s = bf16_to_f32(p[0]);
#pragma unroll
    for (int i = 1; i < VEC; ++i) s = bf16_to_f32(f32_to_bf16_rn(s + bf16_to_f32(p[i])));  // bf16-rounded adds
    return s;
  }
};
// fp8 -> fp16 accumulate (reference fp8 rungs convert to half and add in half, block_all_reduce.cu:497-607).
// gfx950 decodes OCP e4m3fn / e5m2 in hardware (v_cvt_f32_fp8 / v_cvt_f32_bf8).
template <int VEC>
struct PackSum<E4M3, F16, VEC> {
  using elem = unsigned char;
  using out = float;
  static __device__ __forceinline__ float sum(const unsigned char* p) {
    half_t s = (half_t)0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s = s + (half_t)__builtin_amdgcn_cvt_f32_fp8((int)p[i], 0);
    return (float)s;
  }
};
template <int VEC>
struct PackSum<E5M2, F16, VEC> {
  using elem = unsigned char;
  using out = float;
  static __device__ __forceinline__ float sum(const unsigned char* p) {
    half_t s = (half_t)0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s = s + (half_t)__builtin_amdgcn_cvt_f32_bf8((int)p[i], 0);
    return (float)s;
  }
};
template <int VEC>
struct PackSum<I8, I8, VEC> {
  using elem = signed char;
  using out = int;
  static __device__ __forceinline__ int sum(const signed char* p) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += (int)p[i];
    return s;
  }
};

template <typename E, int VEC>
struct alignas(sizeof(E) * VEC) Pack {
  E v[VEC];
};

// One workgroup of 1024 threads per CU at most (256 x 16 waves cover the chip's wave slots): the final
// device-scope atomics on the single result word serialise at ~12 ns each (MI355X_MICROARCH "fanin"), so
// the grid is capped at the CU count -- 4096 small workgroups spent 50 us in that tail alone.
template <typename PS, int VEC>
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const typename PS::elem* __restrict__ a,
                                                          typename PS::out* __restrict__ y, long long n, ClnScratch* sc) {
  using E = typename PS::elem;
  using O = typename PS::out;
  __shared__ O scratch[16];
  O s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const long long nvec = n / VEC;
  const long long stride = (long long)gridDim.x * 1024;
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  // 8 independent loads in flight per lane (round 4; 4 before): at the reference scripts' own sizes a lane owns 8-16 packs in all
  // (4096^2 f16x8: 8), so the kernel is a few round trips to HBM long and each batch of loads that has to wait for the previous one is
  // ~1 us of a 6-8 us launch. Sixteen in flight was built and measured in round 5 (f32 +2-3 %, f16 -2 %, the fp8 x16
  // instances spill 10 registers): not shipped, removed in round 6 (profiles/r05_reduce_grid_probe.log, last rows)
  for (; i + 7 * stride < nvec; i += 8 * stride) {
    Pack<E, VEC> p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = *reinterpret_cast<const Pack<E, VEC>*>(a + (i + u * stride) * VEC);
    s0 += PS::sum(p[0].v), s1 += PS::sum(p[1].v), s2 += PS::sum(p[2].v), s3 += PS::sum(p[3].v);
    s0 += PS::sum(p[4].v), s1 += PS::sum(p[5].v), s2 += PS::sum(p[6].v), s3 += PS::sum(p[7].v);
  }
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    const Pack<E, VEC> p0 = *reinterpret_cast<const Pack<E, VEC>*>(a + i * VEC);
    const Pack<E, VEC> p1 = *reinterpret_cast<const Pack<E, VEC>*>(a + (i + stride) * VEC);
    const Pack<E, VEC> p2 = *reinterpret_cast<const Pack<E, VEC>*>(a + (i + 2 * stride) * VEC);
    const Pack<E, VEC> p3 = *reinterpret_cast<const Pack<E, VEC>*>(a + (i + 3 * stride) * VEC);
    s0 += PS::sum(p0.v);
    s1 += PS::sum(p1.v);
    s2 += PS::sum(p2.v);
    s3 += PS::sum(p3.v);
  }
  for (; i < nvec; i += stride) {
    const Pack<E, VEC> p = *reinterpret_cast<const Pack<E, VEC>*>(a + i * VEC);
    s0 += PS::sum(p.v);
  }
  O s = (s0 + s1) + (s2 + s3);
  if (blockIdx.x == 0) {  // ragged tail, element-wise
    for (long long t = nvec * VEC + threadIdx.x; t < n; t += 1024) {
      E one[VEC] = {};
      one[0] = a[t];
      // sum of a pack whose other slots are zero == decode of the single element
      s += PS::sum(one);
    }
  }
  // wave64 butterfly
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) scratch[w] = s;
  __syncthreads();
  if (w == 0) {
    O t = (lane < 16) ? scratch[lane] : (O)0;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
    if (sc) cln_scratch_finish<O>(sc, y, t, gridDim.x, lane);  // the block that completes the launch moves the total into y and re-zeroes the scratch (stream_scratch.h)
    else if (lane == 0) atomicAdd(y, t);                       // no scratch slot (stream capture): y was zeroed on the stream by the launcher
  }
}

template <typename IN, typename ACC, int VEC>
int launch_reduce(const void* a, void* y, long long n, hipStream_t st) {
  using PS = PackSum<IN, ACC, VEC>;
  using E = typename PS::elem;
  if (!a || !y || n < 0) return CLN_ERR_BAD_ARG;
  if (n == 0) return hipMemsetAsync(y, 0, sizeof(typename PS::out), st) == hipSuccess ? CLN_OK : ((void)hipGetLastError(), CLN_ERR_LAUNCH);
  if (!cln_aligned(a, sizeof(E) * VEC >= 16 ? 16 : sizeof(E) * VEC)) return CLN_ERR_BAD_ARG;
  long long g = (n / VEC + 1023) / 1024;
  // 256 workgroups at most. 512 / 1024 / 2048 workgroups of 1024 threads measured 6-55 % SLOWER at every size (f16 4096^2 9.2 -> 10.7 / 11.9 / 14.3 us,
  // f32 8192^2 48.1 -> 50.9 / 52.5 / 54.2): the completion tickets grow with the grid (profiles/r05_reduce_grid_probe.log)
  constexpr int cap = 256;
  const int grid = (int)(g < 1 ? 1 : (g > cap ? cap : g));
  ClnScratch* sc = cln_stream_scratch(st);
  if (!sc && hipMemsetAsync(y, 0, sizeof(typename PS::out), st) != hipSuccess) return (void)hipGetLastError(), CLN_ERR_LAUNCH;
  CLN_LAUNCH((reduce_sum_kernel<PS, VEC>), dim3(grid), dim3(1024), 0, st, (const E*)a, (typename PS::out*)y, n, sc);
  return cln_check_launch();
}

}  // namespace

// (a, y, n_elements, stream): y is a 1-element fp32 (int32 for i8) buffer that the launch OVERWRITES with the sum (round 5; before: the
// caller had to zero it, as the reference binding does with torch::zeros on cuda:0, block_all_reduce.cu:737-738 -- a zeroed y still works).
#define CLN_RED(name, IN, ACC, VEC)                                            \
  CLN_API int name(const void* a, void* y, long long n, void* stream) {        \
    return launch_reduce<IN, ACC, VEC>(a, y, n, (hipStream_t)stream);          \
  }
CLN_RED(block_all_reduce_sum_f32_f32, F32, F32, 1)
CLN_RED(block_all_reduce_sum_f32x4_f32, F32, F32, 4)
CLN_RED(block_all_reduce_sum_f16_f16, F16, F16, 1)
CLN_RED(block_all_reduce_sum_f16_f32, F16, F32, 1)
CLN_RED(block_all_reduce_sum_f16x2_f16, F16, F16, 2)
CLN_RED(block_all_reduce_sum_f16x2_f32, F16, F32, 2)
CLN_RED(block_all_reduce_sum_f16x8_pack_f16, F16, F16, 8)
CLN_RED(block_all_reduce_sum_f16x8_pack_f32, F16, F32, 8)
CLN_RED(block_all_reduce_sum_bf16_bf16, BF16, BF16, 1)
CLN_RED(block_all_reduce_sum_bf16_f32, BF16, F32, 1)
CLN_RED(block_all_reduce_sum_bf16x2_bf16, BF16, BF16, 2)
CLN_RED(block_all_reduce_sum_bf16x2_f32, BF16, F32, 2)
CLN_RED(block_all_reduce_sum_bf16x8_pack_bf16, BF16, BF16, 8)
CLN_RED(block_all_reduce_sum_bf16x8_pack_f32, BF16, F32, 8)
CLN_RED(block_all_reduce_sum_fp8_e4m3_f16, E4M3, F16, 1)
CLN_RED(block_all_reduce_sum_fp8_e4m3x16_pack_f16, E4M3, F16, 16)
CLN_RED(block_all_reduce_sum_fp8_e5m2_f16, E5M2, F16, 1)
CLN_RED(block_all_reduce_sum_fp8_e5m2x16_pack_f16, E5M2, F16, 16)
CLN_RED(block_all_reduce_sum_i8_i32, I8, I8, 1)
CLN_RED(block_all_reduce_sum_i8x16_pack_i32, I8, I8, 16)
