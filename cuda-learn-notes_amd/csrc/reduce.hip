// block_all_reduce_sum: scalar sum of a whole tensor, 20 rungs over dtype / access width / pack
// accumulator. Replaces reference kernels/reduce/block_all_reduce.cu:42-687 (kernels) and
// :734-813 (bindings). HBM-bound: sizeof(T) bytes read per element.
//
// gfx950 design: block-contiguous streaming on up to 1024 workgroups of 256 threads (the walk note above the kernel), per-lane access width
// fixed by the rung name, wave64 xor-butterfly -> one LDS hop -> ONE atomic + ticket per workgroup
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

// Walk (round 6): BLOCK-CONTIGUOUS chunks on small workgroups. Up to 1024 workgroups of 256 threads; a workgroup takes chunk c = blockIdx, blockIdx +
// grid, ... of 256 K consecutive packs, every lane issues its K loads (lane + 256 k of the chunk) and then sums them: K = 4 for 16-byte packs, 8 below.
// Rounds 1-5 ran ONE 1024-thread workgroup per CU over a strided walk (pack i + u * grid * 1024, 8 loads in flight). With rocprim::reduce as the
// yardstick beside it (bench.py `yardstick`) that form was 2-6 % behind; the stream alone, partials stored without any atomic
// (tools/ubench/stream_forms.hip, profiles/r06_stream_forms_ubench.log): f32 [4096,4096] strided 12.8 us, this walk 11.9; [8192,8192] 45.9 -> 44.6;
// f16 7.5 -> 7.0 and 24.1 -> 22.4 us. The same ubench with 1024-thread workgroups over contiguous chunks gains 0-2 % only: it is the small workgroup
// (a wave that retires is replaced without waiting for fifteen others) at least as much as the address pattern.
// Finish: one RETURNING atomic + one ticket per workgroup on 32 sets of <= 32 workgroups (stream_scratch.h; the single-word fan-in serialises at
// ~12 ns per atomic, which is why round 5 measured 512-2048 workgroups on EIGHT sets 6-55 % slower: profiles/r05_reduce_grid_probe.log).
constexpr int RED_NT = 256, RED_MAX_WG = 1024, RED_SETS = 32;
template <typename PS, int VEC>
__global__ __launch_bounds__(RED_NT) void reduce_sum_kernel(const typename PS::elem* __restrict__ a, typename PS::out* __restrict__ y, long long n, ClnScratch* sc) {
  using E = typename PS::elem;
  using O = typename PS::out;
  using P = Pack<E, VEC>;
  constexpr int K = sizeof(P) >= 16 ? 4 : 8;
  __shared__ O scratch[RED_NT / 64];
  O s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const long long nvec = n / VEC, chunk = (long long)RED_NT * K, nfull = nvec / chunk;
  const P* ap = reinterpret_cast<const P*>(a);
  for (long long c = blockIdx.x; c < nfull; c += gridDim.x) {
    const P* p0 = ap + c * chunk + threadIdx.x;
    P p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = p0[k * RED_NT];
#pragma unroll
    for (int k = 0; k < K; k += 4) s0 += PS::sum(p[k].v), s1 += PS::sum(p[k + 1].v), s2 += PS::sum(p[k + 2].v), s3 += PS::sum(p[k + 3].v);
  }
  if (blockIdx.x == (unsigned)(nfull % gridDim.x)) {  // the packs past the last whole chunk: the workgroup whose turn it would be
    for (long long i = nfull * chunk + threadIdx.x; i < nvec; i += RED_NT) s0 += PS::sum(ap[i].v);
  }
  O s = (s0 + s1) + (s2 + s3);
  if (blockIdx.x == 0) {  // ragged tail, element-wise
    for (long long t = nvec * VEC + threadIdx.x; t < n; t += RED_NT) {
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
    O t = (lane < RED_NT / 64) ? scratch[lane] : (O)0;
#pragma unroll
    for (int m = RED_NT / 128; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
    if (sc) cln_scratch_finish<O, RED_SETS>(sc, y, t, gridDim.x, lane);  // the block that completes the launch moves the total into y and re-zeroes the scratch (stream_scratch.h)
    else if (lane == 0) atomicAdd(y, t);                                 // no scratch slot (stream capture): y was zeroed on the stream by the launcher
  }
}

template <typename IN, typename ACC, int VEC>
int launch_reduce(const void* a, void* y, long long n, hipStream_t st) {
  using PS = PackSum<IN, ACC, VEC>;
  using E = typename PS::elem;
  if (!a || !y || n < 0) return CLN_ERR_BAD_ARG;
  if (n == 0) return hipMemsetAsync(y, 0, sizeof(typename PS::out), st) == hipSuccess ? CLN_OK : ((void)hipGetLastError(), CLN_ERR_LAUNCH);
  if (!cln_aligned(a, sizeof(E) * VEC >= 16 ? 16 : sizeof(E) * VEC)) return CLN_ERR_BAD_ARG;
  constexpr int K = sizeof(E) * VEC >= 16 ? 4 : 8;
  const long long chunks = (n / VEC + (long long)RED_NT * K - 1) / ((long long)RED_NT * K);
  const int grid = (int)(chunks < 1 ? 1 : (chunks > RED_MAX_WG ? RED_MAX_WG : chunks));
  ClnScratch* sc = cln_stream_scratch(st);
  if (!sc && hipMemsetAsync(y, 0, sizeof(typename PS::out), st) != hipSuccess) return (void)hipGetLastError(), CLN_ERR_LAUNCH;
  CLN_LAUNCH((reduce_sum_kernel<PS, VEC>), dim3(grid), dim3(RED_NT), 0, st, (const E*)a, (typename PS::out*)y, n, sc);
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
