// VALU ("CUDA-core") HGEMM teaching rungs for gfx950.
//
// Replaces reference kernels/hgemm/naive/hgemm.cu:23-766 (naive, sliced-k, 8x8 thread tile with
// f16x4/f16x8 packs, bank-conflict-free transposed smem, double buffer) and
// kernels/hgemm/naive/hgemm_async.cu:29-727 (BK=16/32, 16x8 thread tile, cp.async).
// CDNA4 design: the packed-fp16 instruction that fits a thread-tile outer product is
// v_dot2_f32_f16 (2 MACs per lane, fp32 accumulate) rather than the reference's __hfma2 fp16
// accumulators, so both operands are staged in LDS as k-PAIRS (half2 along k): A is transposed
// on the way in, B rows k/k+1 are interleaved on the way in. "async" rungs issue the next tile's
// global loads before the math and write them to LDS after it (global_load is asynchronous until
// its s_waitcnt -- the CDNA analogue of cp.async for a non-lane-linear LDS image).
#pragma once
#include "common.h"

namespace hgemm {

__global__ __launch_bounds__(256) void hgemm_naive_f16_kernel(const half_t* __restrict__ A,
                                                              const half_t* __restrict__ B,
                                                              half_t* __restrict__ C, int M, int N, int K) {
  const int n = blockIdx.x * 16 + (threadIdx.x & 15);
  const int m = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= M || n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf((float)A[(size_t)m * K + k], (float)B[(size_t)k * N + n], acc);
  C[(size_t)m * N + n] = (half_t)acc;
}

// 32x32 C tile, BK = 32, one thread per C element, both operands through LDS
// (reference hgemm_sliced_k_f16_kernel hgemm.cu:44-93).
__global__ __launch_bounds__(1024) void hgemm_sliced_k_f16_kernel(const half_t* __restrict__ A,
                                                                  const half_t* __restrict__ B,
                                                                  half_t* __restrict__ C, int M, int N, int K) {
  __shared__ half_t As[32][32 + 2];
  __shared__ half_t Bs[32][32 + 2];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m = blockIdx.y * 32 + ty, n = blockIdx.x * 32 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    As[ty][tx] = (m < M && k0 + tx < K) ? A[(size_t)m * K + k0 + tx] : (half_t)0;
    Bs[ty][tx] = (k0 + ty < K && n < N) ? B[(size_t)(k0 + ty) * N + n] : (half_t)0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf((float)As[ty][k], (float)Bs[k][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) C[(size_t)m * N + n] = (half_t)acc;
}

// 128x128 block tile, (TM x 8) outputs per thread, v_dot2_f32_f16 on k-pairs.
//   TM = 8 : 256 threads as 16(m) x 16(n)      (reference t_8x8 rungs)
//   TM = 16: 128 threads as  8(m) x 16(n)      (reference t_16x8 rungs)
template <int BK, int TM, bool DBUF, bool ASYNC>
__global__ __launch_bounds__(128 * 8 / TM * 2) void hgemm_valu_tile_kernel(const half_t* __restrict__ A,
                                                                             const half_t* __restrict__ B,
                                                                             half_t* __restrict__ C, int M, int N,
                                                                             int K) {
  constexpr int BM = 128, BN = 128, KP = BK / 2;
  constexpr int NT = (BM / TM) * (BN / 8);
  constexpr int NBUF = DBUF ? 2 : 1;
  constexpr int A_UNITS = BM * BK / 8;   // 16-byte chunks of the A tile
  constexpr int B_UNITS = KP * (BN / 8); // (k-pair, 8-column group) units of the B tile
  constexpr int A_PER = (A_UNITS + NT - 1) / NT, B_PER = (B_UNITS + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) h2 As2[NBUF][KP][BM];
  __shared__ __attribute__((aligned(16))) h2 Bs2[NBUF][KP][BN];

  const int tid = threadIdx.x;
  const int tx = tid % (BN / 8), ty = tid / (BN / 8);
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  h8 ra[A_PER], rb0[B_PER], rb1[B_PER];
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int unit = tid + u * NT;
      if (A_UNITS % NT == 0 || unit < A_UNITS) {
        const int m = unit % BM, ch = unit / BM;
        ra[u] = *reinterpret_cast<const h8*>(A + (size_t)(m0 + m) * K + k0 + ch * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int unit = tid + u * NT;
      if (B_UNITS % NT == 0 || unit < B_UNITS) {
        const int ng = unit % (BN / 8), p = unit / (BN / 8);
        const half_t* src = B + (size_t)(k0 + 2 * p) * N + n0 + ng * 8;
        rb0[u] = *reinterpret_cast<const h8*>(src);
        rb1[u] = *reinterpret_cast<const h8*>(src + N);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int unit = tid + u * NT;
      if (A_UNITS % NT == 0 || unit < A_UNITS) {
        const int m = unit % BM, ch = unit / BM;
#pragma unroll
        for (int j = 0; j < 4; ++j) As2[buf][ch * 4 + j][m] = h2{ra[u][2 * j], ra[u][2 * j + 1]};
      }
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int unit = tid + u * NT;
      if (B_UNITS % NT == 0 || unit < B_UNITS) {
        const int ng = unit % (BN / 8), p = unit / (BN / 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs2[buf][p][ng * 8 + j] = h2{rb0[u][j], rb1[u][j]};
      }
    }
  };

  float acc[TM][8];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;

  // One k-pair slice: TM + 8 LDS reads feed TM x 8 dot products.
  auto slice = [&](int buf, int p) {
    h2 a2[TM], b2[8];
#pragma unroll
    for (int i = 0; i < TM; ++i) a2[i] = As2[buf][p][ty * TM + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) b2[j] = Bs2[buf][p][tx * 8 + j];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_fdot2(a2[i], b2[j], acc[i][j], false);
  };
  auto compute = [&](int buf) {
    if constexpr (KP >= 16 && ASYNC) {
      // a REAL loop of two-slice steps: fully unrolled, hipcc hoists the LDS reads of all 16 slices above the first dot
      // product while the issue-early global loads are also in flight (TM = 16: 96 ds_read_b128 up front, 96 spilled
      // registers, 260 B of scratch). Two slices per trip keep one slice of reads ahead of the math.
#pragma unroll 1
      for (int p = 0; p < KP; p += 2) {
        slice(buf, p);
        slice(buf, p + 1);
      }
    } else {
#pragma unroll
      for (int p = 0; p < KP; ++p) slice(buf, p);
    }
  };

  const int nt = K / BK;
  if constexpr (!DBUF) {
    for (int t = 0; t < nt; ++t) {
      gload(t * BK);
      __syncthreads();
      lstore(0);
      __syncthreads();
      compute(0);
    }
  } else {
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      const bool more = (t + 1) < nt;
      if constexpr (ASYNC) {
        if (more) gload((t + 1) * BK);  // in flight during the math
        compute(cur);
        if (more) lstore(cur ^ 1);
      } else {
        if (more) {
          gload((t + 1) * BK);
          lstore(cur ^ 1);  // waits for the loads before the math starts
        }
        compute(cur);
      }
      __syncthreads();
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[i][j];
    *reinterpret_cast<h8*>(C + (size_t)(m0 + ty * TM + i) * N + n0 + tx * 8) = o;
  }
}

inline int launch_valu_naive(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t s) {
  CLN_LAUNCH(hgemm_naive_f16_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, s, (const half_t*)a,
                     (const half_t*)b, (half_t*)c, M, N, K);
  return cln_check_launch();
}
inline int launch_valu_sliced_k(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t s) {
  CLN_LAUNCH(hgemm_sliced_k_f16_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(1024), 0, s,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K);
  return cln_check_launch();
}
template <int BK, int TM, bool DBUF, bool ASYNC>
int launch_valu_tile(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t s) {
  if (M % 128 || N % 128 || K % BK) return CLN_ERR_UNSUPPORTED;
  constexpr int NT = (128 / TM) * 16;
  CLN_LAUNCH((hgemm_valu_tile_kernel<BK, TM, DBUF, ASYNC>), dim3(N / 128, M / 128), dim3(NT), 0, s,
                     (const half_t*)a, (const half_t*)b, (half_t*)c, M, N, K);
  return cln_check_launch();
}

}  // namespace hgemm
