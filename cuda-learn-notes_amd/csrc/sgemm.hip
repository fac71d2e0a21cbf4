// SGEMM: C[M,N] = A[M,K] * B[K,N], all fp32 row-major. 15 functions here + the two vendor
// rows in hgemm_vendor.hip. Reference: kernels/sgemm/sgemm.cu:21-480 (naive, sliced-k, 8x8 thread tile with
// f32x4 / bank-conflict-free / double-buffer rungs), sgemm_async.cu (k16 tiles, cp.async), and
// sgemm_wmma_tf32_stage.cu:71, :263, :575-700 (TF32 WMMA multi-stage).
//
// gfx950 design:
//  * there is NO TF32 on CDNA4, but there IS an exact-f32 matrix instruction, v_mfma_f32_32x32x2_f32, at the f32
//    vector rate (157 TF peak): the reference's TF32 rungs map onto it and return EXACT fp32 products (bitwise an
//    fmaf chain) instead of 10-bit-mantissa ones. 128x128x16 tile, 4 waves (2x2), each 64x64 = 2x2 MFMA tiles;
//    A staged into a padded m-major LDS image (stride 17 floats: conflict-free ds_read_b32 of 32 rows), B into a
//    k-major image; `stages` LDS buffers with issue-early / write-late register staging; XCD-aware block order
//    for `swizzle`.
//  * the CUDA-core ladder becomes a VALU thread-tile kernel: 128 x (16*TN) block, 8 x TN outputs per lane,
//    v_fma_f32 from a transposed A image, optional double buffer and issue-early/write-late ("async").
#include "common.h"

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

// ---- exact-f32 matrix-core kernel ------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: A operand lane l = A[i = l&31][k = l>>5], B operand lane l = B[k = l>>5][j = l&31];
// result reg r: C[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
template <int STAGES>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K, int tiles_n,
                                                         int swizzle) {
  constexpr int BM = 128, BN = 128, BK = 16, AS = BK + 1, BS = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[STAGES][BM][AS];  // m-major, padded: bank = (17 m + k) % 32
  __shared__ __attribute__((aligned(16))) float Bs[STAGES][BK][BS];  // k-major
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  if (swizzle) {  // bijective XCD remap (block b runs on XCD b % 8): contiguous runs of tiles per XCD
    const int nblk = gridDim.x, xcd = bid & 7, local = bid >> 3, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f4 ra[2], rb[2];  // 128x16 floats = 512 float4 / 256 threads
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int unit = tid + u * 256;
      ra[u] = *reinterpret_cast<const f4*>(A + (size_t)(m0 + (unit >> 2)) * K + k0 + (unit & 3) * 4);
      rb[u] = *reinterpret_cast<const f4*>(B + (size_t)(k0 + (unit >> 5)) * N + n0 + (unit & 31) * 4);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int unit = tid + u * 256;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[buf][unit >> 2][(unit & 3) * 4 + j] = ra[u][j];
      *reinterpret_cast<f4*>(&Bs[buf][unit >> 5][(unit & 31) * 4]) = rb[u];
    }
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = K / BK;
  const int l31 = lane & 31, kh = lane >> 5;
  // prologue: STAGES-1 tiles resident, one more in registers
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nt) {
      gload(s * BK);
      lstore(s);
    }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t % STAGES;
    const bool more = (t + STAGES - 1) < nt;
    if (more) gload((t + STAGES - 1) * BK);  // lands during the MFMAs
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[buf][wm * 64 + i * 32 + l31][kk * 2 + kh];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[buf][kk * 2 + kh][wn * 64 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore((t + STAGES - 1) % STAGES);  // the buffer read in iteration t-1 (all waves passed its barrier)
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        C[(size_t)row * N + n0 + wn * 64 + j * 32 + l31] = acc[i][j][r];
      }
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
int launch_mfma(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle, hipStream_t st) {
  int rc = check3(a, b, c, M, N, K);
  if (rc) return rc;
  if (M % 128 || N % 128 || K % 16) return CLN_ERR_UNSUPPORTED;
  const int tiles_n = N / 128, grid = (M / 128) * tiles_n;
  if (stages >= 3) {
    CLN_LAUNCH((sgemm_mfma_kernel<3>), dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)c, M,
               N, K, tiles_n, swizzle ? 1 : 0);
  } else {
    CLN_LAUNCH((sgemm_mfma_kernel<2>), dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)c, M,
               N, K, tiles_n, swizzle ? 1 : 0);
  }
  return cln_check_launch();
}

}  // namespace

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
