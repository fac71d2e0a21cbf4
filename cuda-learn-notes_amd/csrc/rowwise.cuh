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
constexpr int ROWS_PER_WG_DEFAULT = 1;  // measured: 1 / 2 / 4 rows per workgroup within 1 % of each other at 4096^2 ... 8192^2, 8 slower (profiles/r05_bw_rows_per_wg_probe.log)

// Rows per workgroup (round 5). A row that fits ONE wave (row_threads == 64: up to 4096 fp16 / 2048 fp32 elements at 8 packs per lane)
// needs no LDS and no barrier, so several such rows can share a workgroup: 4096 one-wave workgroups become 1024 of four waves. Built to test whether
// the workgroup count is the 4-5 us fixed cost of the 10-15 us launches at 4096^2: it is NOT -- f16 softmax / layer-norm / rms-norm
// 14.3 / 14.1 / 14.1 us at 1 row per workgroup, 14.6 / 13.8 / 13.9 at 4, 16.9 / 16.5 / 16.2 at 8 -- so the default stays one row per workgroup;
// the kernels keep `rpw` as a runtime argument (1 from every C-ABI entry point; the round-5 environment knob that set it is gone).
inline int rows_per_wg(int nt, int rows, int want = ROWS_PER_WG_DEFAULT) {
  if (nt != 64) return 1;
  int r = want;
  while (r > 1 && rows % r) r >>= 1;  // whole workgroups only: no row guard in the kernels
  return r < 1 ? 1 : r;
}
// position of this thread in its row group: TPR threads per row, row = first row of the workgroup + group index
struct RowPos {
  int tid, tpr;
  size_t row;
};
__device__ __forceinline__ RowPos row_pos(int rpw) {
  RowPos p;
  if (rpw == 1) {
    p.tid = threadIdx.x, p.tpr = blockDim.x, p.row = blockIdx.x;
  } else {  // wave-per-row groups
    p.tid = threadIdx.x & 63, p.tpr = 64, p.row = (size_t)blockIdx.x * rpw + (threadIdx.x >> 6);
  }
  return p;
}

// Register-resident row: MAXV packs of VEC elements per thread, as fp32.
template <typename T, int VEC, int MAXV>
struct RowRegs {
  float x[MAXV][VEC];
  // FULL: the row is exactly MAXV * tpr packs long (e.g. 4096 / 8192 halves at 8 packs per lane) -- no bounds test, no fill: the guarded form costs a
  // v_cndmask / v_mov per element (layer_norm f16x8, 64 elements per lane: 604 VALU instructions, 160 of them selects and moves), and these kernels
  // are not far enough from VALU-bound for that to be free (8192^2 f16: rms-norm 409 VALU -> 5.5 TB/s, layer-norm 604 -> 5.25, softmax 733 -> 5.05)
  template <bool FULL = false>
  __device__ __forceinline__ void load(const T* row, int K, float fill, int tid = threadIdx.x, int tpr = blockDim.x) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int col = (i * tpr + tid) * VEC;
      if (FULL || col < K) {
        const Pack<T, VEC> p = *reinterpret_cast<const Pack<T, VEC>*>(row + col);
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[i][e] = to_f32(p.v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[i][e] = fill;
      }
    }
  }
  template <bool FULL = false>
  __device__ __forceinline__ void store(T* row, int K, int nt = 0, int tid = threadIdx.x, int tpr = blockDim.x) const {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int col = (i * tpr + tid) * VEC;
      if (FULL || col < K) {
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
// the same, CALL(MAXV, FULL): FULL when the row fills every pack of every lane exactly (K == MAXV * nt * VEC)
#define ROWWISE_DISPATCH_MAXV_FULL(vpt, K, nt, VEC, CALL)                      \
  do {                                                                         \
    const int mv_ = (vpt) <= 1 ? 1 : (vpt) <= 2 ? 2 : (vpt) <= 4 ? 4 : 8;      \
    const bool full_ = (long long)(K) == (long long)mv_ * (nt) * (VEC);        \
    if ((vpt) > 8) return CLN_ERR_UNSUPPORTED;                                 \
    if (full_) {                                                               \
      if (mv_ == 1) { CALL(1, true); } else if (mv_ == 2) { CALL(2, true); } else if (mv_ == 4) { CALL(4, true); } else { CALL(8, true); }     \
    } else {                                                                   \
      if (mv_ == 1) { CALL(1, false); } else if (mv_ == 2) { CALL(2, false); } else if (mv_ == 4) { CALL(4, false); } else { CALL(8, false); } \
    }                                                                          \
  } while (0)

}  // namespace rowwise
