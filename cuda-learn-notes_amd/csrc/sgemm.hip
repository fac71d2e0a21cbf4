// SGEMM: C[M,N] = A[M,K] * B[K,N], all fp32 row-major. 15 functions here + the two vendor
// rows in hgemm_vendor.hip. Reference: kernels/sgemm/sgemm.cu:21-480 (naive, sliced-k, 8x8 thread tile with
// f32x4 / bank-conflict-free / double-buffer rungs), sgemm_async.cu (k16 tiles, cp.async), and
// sgemm_wmma_tf32_stage.cu:71, :263, :575-700 (TF32 WMMA multi-stage).
//
// gfx950 design:
//  * there is NO TF32 on CDNA4, but there IS an exact-f32 matrix instruction, v_mfma_f32_32x32x2_f32, at the f32
//    vector rate (157 TF peak): the reference's TF32 rungs map onto it and return EXACT fp32 products
//    instead of 10-bit-mantissa ones. Round 6: sgemm_dma.cuh -- tiles fed by LDS-DMA through a 3-slot ring, ONE barrier
//    per 16-deep stage placed in the stage's middle (in the shadow of an MFMA), no address arithmetic in the loop;
//    tile 64x128 / 128x128 / 256x128 by a per-shape cost (sgemm_plan below). 4096^3: 149 TF = 0.95 of the peak
//    (rounds 2-5, register-staged 128x128x16 with a barrier per stage: 130), rocBLAS sgemm 140, hipBLASLt 150.
//    `stages` 2 and 3 run the same ring (read / landed / in flight is what keeps the barrier off the stage boundary);
//    `swizzle` = XCD-aware block order. Every tile form adds the k products in the same order: the result does not
//    depend on the tile the planner picks (except where K is split in two for grids of <= 128 tiles: sgemm_plan).
//  * the CUDA-core ladder becomes a VALU thread-tile kernel: 128 x (16*TN) block, 8 x TN outputs per lane,
//    v_fma_f32 from a transposed A image, optional double buffer and issue-early/write-late ("async").
#include <string.h>

#include "common.h"
#include "sgemm_dma.cuh"

namespace {

int check3(const void* a, const void* b, const void* c, int M, int N, int K) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(a) || !cln_aligned16(b) || !cln_aligned16(c)) return CLN_ERR_BAD_ARG;
  return CLN_OK;
}

__global__ __launch_bounds__(256) void sgemm_naive_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K) {
  const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= M || n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], B[(size_t)k * N + n], acc);
  C[(size_t)m * N + n] = acc;
}

__global__ __launch_bounds__(1024) void sgemm_sliced_k_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ C, int M, int N, int K) {
  __shared__ float As[32][33], Bs[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m = blockIdx.y * 32 + ty, n = blockIdx.x * 32 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    As[ty][tx] = (m < M && k0 + tx < K) ? A[(size_t)m * K + k0 + tx] : 0.f;
    Bs[ty][tx] = (k0 + ty < K && n < N) ? B[(size_t)(k0 + ty) * N + n] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf(As[ty][k], Bs[k][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) C[(size_t)m * N + n] = acc;
}

// 128 x (16*TN) block tile, 256 threads as 16(m) x 16(n), 8 x TN outputs per thread.
template <int BK, int TN, bool DBUF, bool ASYNC>
__global__ __launch_bounds__(256) void sgemm_valu_tile_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 16 * TN, NBUF = DBUF ? 2 : 1;
  constexpr int A_U = BM * BK / 4, B_U = BK * BN / 4;  // float4 units
  constexpr int A_PER = (A_U + 255) / 256, B_PER = (B_U + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[NBUF][BK][BM + 4];  // transposed: [k][m]
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK][BN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f4 ra[A_PER], rb[B_PER];
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int unit = tid + u * 256;
      if (A_U % 256 == 0 || unit < A_U) {
        const int m = unit / (BK / 4), kc = unit % (BK / 4);
        ra[u] = *reinterpret_cast<const f4*>(A + (size_t)(m0 + m) * K + k0 + kc * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int unit = tid + u * 256;
      if (B_U % 256 == 0 || unit < B_U) {
        const int k = unit / (BN / 4), nc = unit % (BN / 4);
        rb[u] = *reinterpret_cast<const f4*>(B + (size_t)(k0 + k) * N + n0 + nc * 4);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int unit = tid + u * 256;
      if (A_U % 256 == 0 || unit < A_U) {
        const int m = unit / (BK / 4), kc = unit % (BK / 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) As[buf][kc * 4 + j][m] = ra[u][j];
      }
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int unit = tid + u * 256;
      if (B_U % 256 == 0 || unit < B_U) {
        const int k = unit / (BN / 4), nc = unit % (BN / 4);
        *reinterpret_cast<f4*>(&Bs[buf][k][nc * 4]) = rb[u];
      }
    }
  };
  float acc[8][TN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;
  auto compute = [&](int buf) {
#pragma unroll(TN >= 16 ? 2 : 4)
    for (int k = 0; k < BK; ++k) {
      float a[8], b[TN];
      *reinterpret_cast<f4*>(a) = *reinterpret_cast<const f4*>(&As[buf][k][ty * 8]);
      *reinterpret_cast<f4*>(a + 4) = *reinterpret_cast<const f4*>(&As[buf][k][ty * 8 + 4]);
#pragma unroll
      for (int j = 0; j < TN; j += 4) *reinterpret_cast<f4*>(b + j) = *reinterpret_cast<const f4*>(&Bs[buf][k][tx * TN + j]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
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
        if (more) gload((t + 1) * BK);
        compute(cur);
        if (more) lstore(cur ^ 1);
      } else {
        if (more) {
          gload((t + 1) * BK);
          lstore(cur ^ 1);
        }
        compute(cur);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < TN; j += 4)
      *reinterpret_cast<f4*>(C + (size_t)(m0 + ty * 8 + i) * N + n0 + tx * TN + j) =
          f4{acc[i][j], acc[i][j + 1], acc[i][j + 2], acc[i][j + 3]};
}

template <int BK, int TN, bool DBUF, bool ASYNC>
int launch_valu(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t st) {
  int rc = check3(a, b, c, M, N, K);
  if (rc) return rc;
  if (M % 128 || N % (16 * TN) || K % BK) return CLN_ERR_UNSUPPORTED;
  CLN_LAUNCH((sgemm_valu_tile_kernel<BK, TN, DBUF, ASYNC>), dim3(N / (16 * TN), M / 128), dim3(256), 0, st,
             (const float*)a, (const float*)b, (float*)c, M, N, K);
  return cln_check_launch();
}
// ---- exact-f32 matrix-core rungs: tile choice. Cost of a tile form = (most tiles any CU gets) x (tile area) / (efficiency of the form on a full
// chip, measured on the reference's sweep 4096..16384 x K 2048..8192, profiles/r06_sgemm_dma_sweep.log): the wide tile moves the fewest L2 bytes
// and wins by 1-2 % once the output is >= 8192^2; 128x128 staggers its epilogues (four tiles per CU, three resident) and wins below; 64x128 only
// pays where the larger tiles leave CUs idle (3072^3: 576 tiles of 128x128 are 2.25 per CU).
struct SgemmPlan {
  int bm, bn;
  bool ksplit;  // few tiles, long K: two workgroups per tile, each half of K (sgemm_dma.cuh KSPLIT)
};
SgemmPlan sgemm_plan(int M, int N, int K = 0) {
  const double area = (double)M * N;
  const bool big = area >= 8192.0 * 8192.0, small = area < 4096.0 * 4096.0;
  const long long t128 = (M % 128 || N % 128) ? 0 : (long long)(M / 128) * (N / 128);
  struct Cand {
    int bm, bn;
    double eff;
  } cands[3] = {{128, 128, t128 <= 256 ? 0.94 : big ? 0.975 : 1.005},  // (one 4-wave tile per CU leaves every SIMD a single wave)
                {256, 128, big ? 1.0 : 0.995},
                {64, 128, small ? 0.97 : 0.94}};
  SgemmPlan best{64, 128, false};
  double best_cost = 1e300;
  for (const Cand& c : cands) {
    if (M % c.bm || N % c.bn) continue;
    const long long tiles = (long long)(M / c.bm) * (N / c.bn);
    const double cost = (double)((tiles + 255) / 256) * c.bm * c.bn / c.eff;
    if (cost < best_cost) {
      best_cost = cost;
      best = {c.bm, c.bn, false};
    }
  }
  // at most 128 tiles of 64x128 leave half of the CUs idle (1024^3: 66 TF against the vendors' 119): from K = 512 on the K range is split in two
  // (1024^3 -> 256 workgroups; the two partial products meet in C by one fp32 atomic add each onto zeros -- commutative, so still bit-repeatable,
  // but NOT the k order of the unsplit forms)
  if (best.bm == 64 && (long long)(M / 64) * (N / 128) <= 128 && K >= 512) best.ksplit = true;
  return best;
}
int launch_mfma(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle, hipStream_t st) {
  (void)stages;
  int rc = check3(a, b, c, M, N, K);
  if (rc) return rc;
  if (M % 64 || N % 128 || K % 16) return CLN_ERR_UNSUPPORTED;
  const SgemmPlan p = sgemm_plan(M, N, K);
  if (p.ksplit) return sgemm_dma::launch<2, 2, 1, 2, 16, 3, true>(a, b, c, M, N, K, swizzle, st);
  if (p.bm == 256) return sgemm_dma::launch<2, 2, 4, 2, 16, 3>(a, b, c, M, N, K, swizzle, st);
  if (p.bm == 128) return sgemm_dma::launch<2, 2, 2, 2, 16, 3>(a, b, c, M, N, K, swizzle, st);
  return sgemm_dma::launch<2, 2, 1, 2, 16, 3>(a, b, c, M, N, K, swizzle, st);
}

}  // namespace

// cln_describe for the two matrix-core names (describe.hip): the tile form sgemm_plan picks for the shape, no launch
int cln_sgemm_describe(const char* name, int M, int N, int K, int stages, char* buf, int len) {
  (void)stages;
  if (strcmp(name, "sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages") != 0 && strcmp(name, "sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages_dsmem") != 0)
    return CLN_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || M % 64 || N % 128 || K % 16) return CLN_ERR_UNSUPPORTED;
  const SgemmPlan p = sgemm_plan(M, N, K);
  const int n = snprintf(buf, (size_t)len, "sgemm_dma<%dx%dx16,4 waves,%dx%d wave tiles,3-slot LDS-DMA ring,mid-stage barrier,v_mfma_f32_32x32x2_f32>%s [stages ignored: one ring]",
                         p.bm, p.bn, p.bm == 256 ? 128 : p.bm / 2, 64, p.ksplit ? " x 2 halves of K (memset + one fp32 atomic add per element and half)" : "");
  return n < len ? n : len - 1;
}

#define CLN_S3(name, expr)                                                                         \
  CLN_API int name(const void* a, const void* b, void* c, int M, int N, int K, void* stream_) {   \
    hipStream_t stream = (hipStream_t)stream_;                                                     \
    return (expr);                                                                                 \
  }
static int sg_naive(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t st) {
  int rc = check3(a, b, c, M, N, K);
  if (rc) return rc;
  CLN_LAUNCH(sgemm_naive_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, st, (const float*)a,
             (const float*)b, (float*)c, M, N, K);
  return cln_check_launch();
}
static int sg_sliced(const void* a, const void* b, void* c, int M, int N, int K, hipStream_t st) {
  int rc = check3(a, b, c, M, N, K);
  if (rc) return rc;
  CLN_LAUNCH(sgemm_sliced_k_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(1024), 0, st, (const float*)a,
             (const float*)b, (float*)c, M, N, K);
  return cln_check_launch();
}
CLN_S3(sgemm_naive_f32, sg_naive(a, b, c, M, N, K, stream))
CLN_S3(sgemm_sliced_k_f32, sg_sliced(a, b, c, M, N, K, stream))
CLN_S3(sgemm_t_8x8_sliced_k_f32x4, (launch_valu<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x8_sliced_k_f32x4_bcf, (launch_valu<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x8_sliced_k_f32x4_bcf_offset, (launch_valu<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf, (launch_valu<8, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf_offset, (launch_valu<8, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x4_sliced_k16_f32x4_bcf_dbuf, (launch_valu<16, 4, true, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x4_sliced_k16_f32x4_bcf_dbuf_async, (launch_valu<16, 4, true, true>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x8_sliced_k16_f32x4_bcf_dbuf, (launch_valu<16, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x8_sliced_k16_f32x4_bcf_dbuf_async, (launch_valu<16, 8, true, true>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x16_sliced_k16_f32x4_bcf_dbuf, (launch_valu<16, 16, true, false>(a, b, c, M, N, K, stream)))
CLN_S3(sgemm_t_8x16_sliced_k16_f32x4_bcf_dbuf_async, (launch_valu<16, 16, true, true>(a, b, c, M, N, K, stream)))

#define CLN_S6(name)                                                                                          \
  CLN_API int name(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle,       \
                   int swizzle_stride, void* stream) {                                                         \
    (void)swizzle_stride;                                                                                      \
    return launch_mfma(a, b, c, M, N, K, stages, swizzle, (hipStream_t)stream);                                \
  }
CLN_S6(sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages)
CLN_S6(sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages_dsmem)
