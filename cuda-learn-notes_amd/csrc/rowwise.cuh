// Row-wise streaming helpers shared by softmax / layer-norm / rms-norm: one workgroup per row,
// the row lives in registers between the reduction pass and the write pass (1 HBM read + 1 HBM
// write per element), per-lane access width fixed by the rung name (VEC * sizeof(T) bytes).
#pragma once
#include "common.h"

namespace rowwise {

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
  T v[VEC];
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(half_t x) { return (float)x; }
template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ half_t from_f32<half_t>(float x) { return (half_t)x; }

// Threads per row: eight packs per lane (8 independent 16-byte loads in flight per lane; a 4096-element fp16 row is
// ONE wave -- no LDS hop, no barrier in the reductions -- and an fp32 row two), rounded to whole waves, capped at 1024.
// Measured at 4096 x 4096 (round 2, hipGraph-timed): 8 instead of 4 packs per lane = softmax f16x8 4.88 -> 5.47 TB/s,
// layer-norm f16x8 5.35 -> 5.60, the f32x4 rows 6.3 -> 6.5; 2 and 1 packs per lane are slower than 4.
inline int row_threads(int K, int VEC) {
  const int nvec = K / VEC;
  int nt = (((nvec + 7) / 8 + 63) / 64) * 64;
  if (nt > 1024) nt = 1024;
  if (nt < 64) nt = 64;
  return nt;
}
inline int vecs_per_thread(int K, int VEC, int nt) { return (K / VEC + nt - 1) / nt; }

// Register-resident row: MAXV packs of VEC elements per thread, as fp32.
template <typename T, int VEC, int MAXV>
struct RowRegs {
  float x[MAXV][VEC];
  __device__ __forceinline__ void load(const T* row, int K, float fill) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int col = (i * blockDim.x + threadIdx.x) * VEC;
      if (col < K) {
        const Pack<T, VEC> p = *reinterpret_cast<const Pack<T, VEC>*>(row + col);
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[i][e] = to_f32(p.v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[i][e] = fill;
      }
    }
  }
  __device__ __forceinline__ void store(T* row, int K, int nt = 0) const {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int col = (i * blockDim.x + threadIdx.x) * VEC;
      if (col < K) {
        Pack<T, VEC> p;
#pragma unroll
        for (int e = 0; e < VEC; ++e) p.v[e] = from_f32<T>(x[i][e]);
        cln_store_stream(reinterpret_cast<Pack<T, VEC>*>(row + col), p, nt);
      }
    }
  }
};

// Dispatch a functor on MAXV in {1,2,4,8}; returns false if the row does not fit.
#define ROWWISE_DISPATCH_MAXV(vpt, CALL) \
  do {                                   \
    if ((vpt) <= 1) { CALL(1); }         \
    else if ((vpt) <= 2) { CALL(2); }    \
    else if ((vpt) <= 4) { CALL(4); }    \
    else if ((vpt) <= 8) { CALL(8); }    \
    else return CLN_ERR_UNSUPPORTED;     \
  } while (0)

}  // namespace rowwise
