// dot-product (5), sgemv (3), hgemv (3), mat-transpose (13).
//   reference kernels/dot-product/dot_product.cu:35-276, kernels/sgemv/sgemv.cu:29-190,
//   kernels/hgemv/hgemv.cu:34-196, kernels/mat-transpose/mat_transpose.cu:29-360.
// gfx950 design (all HBM/latency bound, nothing for the matrix cores):
//  * dot_prod: the block_all_reduce structure with two input streams: up to 1024 workgroups of 256 threads over block-contiguous chunks,
//    2-4 independent pack pairs in flight per lane, fp32 accumulate, DPP wave reduce, one atomic + ticket per workgroup.
//  * gemv: y[m] = sum_k a[m,k] x[k]; G lanes per row (G = 16 / 32 / 64 by K), 64/G rows per wave, the rung's
//    access width per lane, fp32 accumulate (the reference's hgemv adds in half), reduction inside the G-lane
//    group on the VALU (DPP row ops; v_permlane16_swap / 32_swap across rows), 4 waves per workgroup.
//  * transpose: y[n,m] = x[m,n], bit-exact. The reference's 13 rungs differ in which side is coalesced, 4-wide
//    packing, 1-D vs 2-D grids, diagonal block order and a padded / unpadded shared tile; here:
//      *_col2row*  -> read-coalesced streaming kernel (scattered 4-byte writes), 1 or 4 elements per lane
//      *_row2col*  -> write-coalesced streaming kernel (scattered reads)
//      f32x4_*2d   -> (round 6) 4 x 4 register blocks, 8 x 8 lanes per 32 x 32 block: 16-byte accesses and full lines on both sides, no LDS
//      diagonal2d  -> write-coalesced kernel with the reference's diagonal block order
//      *_shared_*  -> 64x64 tile through LDS, 16-byte accesses on BOTH sides; `bcf` pads the tile rows (+1)
#include "common.h"
#include "stream_scratch.h"

namespace {

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pk {
  T v[VEC];
};
__device__ __forceinline__ float tof(float x) { return x; }
__device__ __forceinline__ float tof(half_t x) { return (float)x; }

// ---- dot product ---------------------------------------------------------------------------------
// Round 6: the walk of block_all_reduce_sum (reduce.hip) -- up to 1024 workgroups of 256 threads over consecutive chunks of 256 K pack pairs, K = 2 for
// 16-byte packs (4 loads in flight per lane, two streams), 4 below; one returning atomic + ticket per workgroup on 32 sets.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dot_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                  float* __restrict__ y, long long n, ClnScratch* sc) {
  using P = Pk<T, VEC>;
  constexpr int K = sizeof(P) >= 16 ? 2 : 4;
  __shared__ float scratch[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long long nvec = n / VEC, chunk = 256LL * K, nfull = nvec / chunk;
  const P* ap = reinterpret_cast<const P*>(a);
  const P* bp = reinterpret_cast<const P*>(b);
  auto fma_pack = [&](const P& pa, const P& pb, float& acc) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc = fmaf(tof(pa.v[e]), tof(pb.v[e]), acc);
  };
  for (long long c = blockIdx.x; c < nfull; c += gridDim.x) {
    const long long base = c * chunk + threadIdx.x;
    P pa[K], pb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) pa[k] = ap[base + k * 256];
#pragma unroll
    for (int k = 0; k < K; ++k) pb[k] = bp[base + k * 256];
    fma_pack(pa[0], pb[0], s0);
    fma_pack(pa[1], pb[1], s1);
    if constexpr (K == 4) fma_pack(pa[2], pb[2], s2), fma_pack(pa[3], pb[3], s3);
  }
  if (blockIdx.x == (unsigned)(nfull % gridDim.x))  // the packs past the last whole chunk
    for (long long i = nfull * chunk + threadIdx.x; i < nvec; i += 256) fma_pack(ap[i], bp[i], s0);
  float s = (s0 + s1) + (s2 + s3);
  if (blockIdx.x == 0)
    for (long long t = nvec * VEC + threadIdx.x; t < n; t += 256) s = fmaf(tof(a[t]), tof(b[t]), s);
  s = wave_sum(s);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) scratch[w] = s;
  __syncthreads();
  if (w == 0) {
    float t = 0.f;
    if (lane == 0) t = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    if (sc) cln_scratch_finish<float, 32>(sc, y, t, gridDim.x, lane);  // the block that completes the launch moves the total into y (stream_scratch.h): y need not be zeroed
    else if (lane == 0) atomicAdd(y, t);
  }
}
template <typename T, int VEC>
int launch_dot(const void* a, const void* b, void* y, long long n, hipStream_t st) {
  if (!a || !b || !y || n < 0) return CLN_ERR_BAD_ARG;
  if (n == 0) return hipMemsetAsync(y, 0, sizeof(float), st) == hipSuccess ? CLN_OK : ((void)hipGetLastError(), CLN_ERR_LAUNCH);
  if (sizeof(T) * VEC >= 16 && (!cln_aligned16(a) || !cln_aligned16(b))) return CLN_ERR_BAD_ARG;
  constexpr int K = sizeof(T) * VEC >= 16 ? 2 : 4;
  const long long chunks = (n / VEC + 256LL * K - 1) / (256LL * K);
  const int grid = (int)(chunks < 1 ? 1 : (chunks > 1024 ? 1024 : chunks));
  ClnScratch* sc = cln_stream_scratch(st);
  if (!sc && hipMemsetAsync(y, 0, sizeof(float), st) != hipSuccess) return (void)hipGetLastError(), CLN_ERR_LAUNCH;
  CLN_LAUNCH((dot_kernel<T, VEC>), dim3(grid), dim3(256), 0, st, (const T*)a, (const T*)b, (float*)y, n, sc);
  return cln_check_launch();
}

// ---- gemv ----------------------------------------------------------------------------------------
// G lanes cooperate on one row; a wave holds 64/G rows. Reduction inside the G-lane group.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += cln_dpp<0xB1>(v);
  v += cln_dpp<0x4E>(v);
  v += cln_dpp<0x141>(v);
  v += cln_dpp<0x140>(v);  // 16-lane row total in every lane
  if constexpr (G >= 32) {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
  }
  if constexpr (G >= 64) {
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
  }
  return v;
}
template <typename T, int VEC, int G>
__global__ __launch_bounds__(256) void gemv_kernel(const T* __restrict__ a, const T* __restrict__ x,
                                                   T* __restrict__ y, int M, int K) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane % G, sub = lane / G;
  const int m = (blockIdx.x * 4 + wave) * RPW + sub;
  float s = 0.f;
  if (m < M) {
    const T* row = a + (size_t)m * K;
    // U row pieces requested before the first product (round 6: one piece in flight per lane left hgemv_k128_f16x4 at 0.87x torch.mv on
    // [65536,1024]); the products are added in the same k order as the one-piece loop: same bits
    constexpr int U = 8;
    int k = g * VEC;
    for (; k + (U - 1) * G * VEC < K; k += U * G * VEC) {
      Pk<T, VEC> pa[U], px[U];
#pragma unroll
      for (int u = 0; u < U; ++u) pa[u] = *reinterpret_cast<const Pk<T, VEC>*>(row + k + u * G * VEC);
#pragma unroll
      for (int u = 0; u < U; ++u) px[u] = *reinterpret_cast<const Pk<T, VEC>*>(x + k + u * G * VEC);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s = fmaf(tof(pa[u].v[e]), tof(px[u].v[e]), s);
    }
    for (; k < K; k += G * VEC) {
      const Pk<T, VEC> pa = *reinterpret_cast<const Pk<T, VEC>*>(row + k);
      const Pk<T, VEC> px = *reinterpret_cast<const Pk<T, VEC>*>(x + k);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s = fmaf(tof(pa.v[e]), tof(px.v[e]), s);
    }
  }
  s = group_sum<G>(s);
  if (m < M && g == 0) y[m] = (T)s;
}
// Long rows, the two f16 rungs (hgemv_k128_f16x4: U = 4 pieces of 4 halves; hgemv_k32_f16: U = 8 pieces of one half -- 26.6 -> 22.2 us on [65536,1024]): a wave owns R consecutive rows with all 64 lanes on each; one x piece feeds R rows (x is read R times less often), a load
// instruction covers 64 * VEC contiguous elements of ONE row, R * U row pieces are in flight per lane. Same-box A/B against the 32-lanes-per-row kernel and
// torch.mv (scratch harness, rotating sets): [65536,1024] 26.4 -> 22.3 us (torch.mv 24.8), [16384,4096] 25.1 (24.8), [8192,8192] 25.9 at R = 2 (26.2);
// R = 4 from 16384 rows on, 2 below (8192 rows leave R = 4 two waves per SIMD: 31.6 us). The f32x4 rung measured level or slower in this form and keeps 32 lanes per row.
template <typename T, int VEC, int R, int U>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const T* __restrict__ a, const T* __restrict__ x, T* __restrict__ y, int M, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = (blockIdx.x * 4 + wave) * R;
  if (m0 >= M) return;
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = 0.f;
  const T* row[R];
#pragma unroll
  for (int r = 0; r < R; ++r) row[r] = a + (size_t)(m0 + r < M ? m0 + r : M - 1) * K;
  int k = lane * VEC;
  for (; k + (U - 1) * 64 * VEC < K; k += U * 64 * VEC) {
    Pk<T, VEC> pa[R][U], px[U];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) pa[r][u] = *reinterpret_cast<const Pk<T, VEC>*>(row[r] + k + u * 64 * VEC);
#pragma unroll
    for (int u = 0; u < U; ++u) px[u] = *reinterpret_cast<const Pk<T, VEC>*>(x + k + u * 64 * VEC);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[r] = fmaf(tof(pa[r][u].v[e]), tof(px[u].v[e]), s[r]);
  }
  for (; k < K; k += 64 * VEC) {
    const Pk<T, VEC> px = *reinterpret_cast<const Pk<T, VEC>*>(x + k);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const Pk<T, VEC> pa = *reinterpret_cast<const Pk<T, VEC>*>(row[r] + k);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[r] = fmaf(tof(pa.v[e]), tof(px.v[e]), s[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float t = group_sum<64>(s[r]);
    if (lane == 0 && m0 + r < M) y[m0 + r] = (T)t;
  }
}
template <typename T, int VEC, int R, int U>
int launch_gemv_rows(const void* a, const void* x, void* y, int M, int K, hipStream_t st) {
  if (!a || !x || !y || M <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  constexpr int RPB = 4 * R;
  CLN_LAUNCH((gemv_rows_kernel<T, VEC, R, U>), dim3((M + RPB - 1) / RPB), dim3(256), 0, st, (const T*)a, (const T*)x, (T*)y, M, K);
  return cln_check_launch();
}
template <typename T, int VEC, int G>
int launch_gemv(const void* a, const void* x, void* y, int M, int K, hipStream_t st) {
  if (!a || !x || !y || M <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (K % VEC) return CLN_ERR_UNSUPPORTED;
  constexpr int RPB = 4 * (64 / G);
  CLN_LAUNCH((gemv_kernel<T, VEC, G>), dim3((M + RPB - 1) / RPB), dim3(256), 0, st, (const T*)a, (const T*)x,
             (T*)y, M, K);
  return cln_check_launch();
}

// ---- transpose -----------------------------------------------------------------------------------
// x: [row, col] -> y: [col, row]
template <int VEC>
__global__ __launch_bounds__(256) void tr_read_coalesced(const float* __restrict__ x, float* __restrict__ y, int row,
                                                         int col) {
  const long long total = (long long)row * col / VEC, stride = (long long)gridDim.x * 256;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    const long long e = t * VEC;
    const int r = (int)(e / col), c = (int)(e - (long long)r * col);
    const Pk<float, VEC> p = *reinterpret_cast<const Pk<float, VEC>*>(x + e);
#pragma unroll
    for (int k = 0; k < VEC; ++k) y[(size_t)(c + k) * row + r] = p.v[k];
  }
}
template <int VEC, bool DIAG>
__global__ __launch_bounds__(256) void tr_write_coalesced(const float* __restrict__ x, float* __restrict__ y, int row,
                                                          int col) {
  const long long total = (long long)row * col / VEC, stride = (long long)gridDim.x * 256;
  for (long long t0 = (long long)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += stride) {
    long long t = t0;
    if constexpr (DIAG) {  // walk the output in a diagonal block order (reference mat_transpose.cu:81-90)
      const long long nb = total / 256;
      if (nb > 1 && t < nb * 256) {
        const long long b = t / 256, side = (long long)sqrtf((float)nb);
        if (side * side == nb) t = ((b % side) * side + (b / side + b % side) % side) * 256 + (t % 256);
      }
    }
    const long long e = t * VEC;  // output element index: y[c, r..r+VEC)
    const int c = (int)(e / row), r = (int)(e - (long long)c * row);
    Pk<float, VEC> p;
#pragma unroll
    for (int k = 0; k < VEC; ++k) p.v[k] = x[(size_t)(r + k) * col + c];
    *reinterpret_cast<Pk<float, VEC>*>(y + e) = p;
  }
}
// 64 x 64 tile through LDS, float4 on both sides. PAD = 1 makes the transposed (column) reads conflict-free.
template <int PAD>
__global__ __launch_bounds__(256) void tr_lds_tile(const float* __restrict__ x, float* __restrict__ y, int row,
                                                   int col) {
  __shared__ float tile[64][64 + PAD];
  const int tiles_c = col / 64;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 16 + (t >> 4), c4 = (t & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(tr * 64 + r) * col + tc * 64 + c4);
    tile[r][c4] = v.x, tile[r][c4 + 1] = v.y, tile[r][c4 + 2] = v.z, tile[r][c4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = it * 16 + (t >> 4), r4 = (t & 15) * 4;  // output row = input column c, 4 consecutive input rows
    float4 v = {tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]};
    *reinterpret_cast<float4*>(y + (size_t)(tc * 64 + c) * row + tr * 64 + r4) = v;
  }
}

// No LDS, no barrier (round 6): a wave owns a 32 x 32 block as 8 x 8 lanes of 4 x 4 REGISTER blocks. Lane (a, b) reads rows 4a .. 4a+3 at columns
// 4b .. 4b+3 (four 16-byte loads: the 8 lanes of one a cover one 128-byte line per row), transposes by register renaming and writes rows 4b .. 4b+3 of y at
// columns 4a .. 4a+3 (the 8 lanes of one b cover one 128-byte line per output row): full lines on both sides. Against the LDS tile on the same box
// (tools/ubench/transpose_forms.hip, profiles/r06_transpose_forms_ubench.log): [8192,8192] 124.8 -> 108.0 us (4.30 -> 4.97 TB/s; hipMemcpyDtoD of the same
// bytes 5.33), [4096,4096] 30.2 -> 29.2, [2048,2048] 7.5 -> 8.1 (the LDS tile stays ahead below ~32 MB). The two f32x4 *2d rungs run it when both extents
// divide by 32; the `shared` rungs keep the LDS tile their name states.
__global__ __launch_bounds__(256) void tr_reg4x4(const float* __restrict__ x, float* __restrict__ y, int row, int col) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane >> 3, b = lane & 7;
  const int blocks_c = col / 32;
  const long long w = (long long)blockIdx.x * 4 + wave;  // one 32 x 32 block per wave
  const int br = (int)(w / blocks_c), bc = (int)(w - (long long)br * blocks_c);
  if (br * 32 >= row) return;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(x + (size_t)(br * 32 + 4 * a + i) * col + bc * 32 + 4 * b);
  float* yo = y + (size_t)(bc * 32 + 4 * b) * row + br * 32 + 4 * a;
  *reinterpret_cast<float4*>(yo) = float4{v[0].x, v[1].x, v[2].x, v[3].x};
  *reinterpret_cast<float4*>(yo + (size_t)row) = float4{v[0].y, v[1].y, v[2].y, v[3].y};
  *reinterpret_cast<float4*>(yo + 2 * (size_t)row) = float4{v[0].z, v[1].z, v[2].z, v[3].z};
  *reinterpret_cast<float4*>(yo + 3 * (size_t)row) = float4{v[0].w, v[1].w, v[2].w, v[3].w};
}

enum TrKind { TR_READ1, TR_READ4, TR_WRITE1, TR_WRITE4, TR_DIAG, TR_LDS, TR_LDS_BCF, TR_READ4_2D, TR_WRITE4_2D };
int launch_tr(int kind, const void* x, void* y, int row, int col, hipStream_t st) {
  if (!x || !y || row <= 0 || col <= 0) return CLN_ERR_BAD_ARG;
  const long long n = (long long)row * col;
  const float* xp = (const float*)x;
  float* yp = (float*)y;
  if ((kind == TR_READ4_2D || kind == TR_WRITE4_2D)) {  // the f32x4 *2d rungs: the register-block kernel where the shape allows, else the 1-D rung of the same name
    if (row % 32 == 0 && col % 32 == 0 && cln_aligned16(x) && cln_aligned16(y)) {
      const long long waves = (long long)(row / 32) * (col / 32);
      CLN_LAUNCH(tr_reg4x4, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const float*)x, (float*)y, row, col);
      return cln_check_launch();
    }
    kind = kind == TR_READ4_2D ? TR_READ4 : TR_WRITE4;
  }
  const bool v4 = (kind == TR_READ4 || kind == TR_WRITE4);
  if (v4 && ((kind == TR_READ4 ? col : row) % 4 || !cln_aligned16(x) || !cln_aligned16(y))) return CLN_ERR_UNSUPPORTED;
  if ((kind == TR_LDS || kind == TR_LDS_BCF) && (row % 64 || col % 64 || !cln_aligned16(x) || !cln_aligned16(y)))
    return CLN_ERR_UNSUPPORTED;
  const int grid = cln_stream_grid(n / (v4 ? 4 : 1), 256);
  switch (kind) {
    case TR_READ1: CLN_LAUNCH((tr_read_coalesced<1>), dim3(grid), dim3(256), 0, st, xp, yp, row, col); break;
    case TR_READ4: CLN_LAUNCH((tr_read_coalesced<4>), dim3(grid), dim3(256), 0, st, xp, yp, row, col); break;
    case TR_WRITE1: CLN_LAUNCH((tr_write_coalesced<1, false>), dim3(grid), dim3(256), 0, st, xp, yp, row, col); break;
    case TR_WRITE4: CLN_LAUNCH((tr_write_coalesced<4, false>), dim3(grid), dim3(256), 0, st, xp, yp, row, col); break;
    case TR_DIAG: CLN_LAUNCH((tr_write_coalesced<1, true>), dim3((int)((n + 255) / 256)), dim3(256), 0, st, xp, yp, row, col); break;
    case TR_LDS: CLN_LAUNCH((tr_lds_tile<0>), dim3((row / 64) * (col / 64)), dim3(256), 0, st, xp, yp, row, col); break;
    case TR_LDS_BCF: CLN_LAUNCH((tr_lds_tile<1>), dim3((row / 64) * (col / 64)), dim3(256), 0, st, xp, yp, row, col); break;
    default: return CLN_ERR_BAD_ARG;
  }
  return cln_check_launch();
}

}  // namespace

// (a, b, y fp32[1] -- overwritten with the dot product, need not be zeroed (round 5) --, n, stream) -- reference `torch::Tensor dot_prod_*(Tensor a, Tensor b)`
#define CLN_DOT(name, T, VEC)                                                              \
  CLN_API int name(const void* a, const void* b, void* y, long long n, void* stream) {     \
    return launch_dot<T, VEC>(a, b, y, n, (hipStream_t)stream);                            \
  }
CLN_DOT(dot_prod_f32_f32, float, 1)
CLN_DOT(dot_prod_f32x4_f32, float, 4)
CLN_DOT(dot_prod_f16_f32, half_t, 1)
CLN_DOT(dot_prod_f16x2_f32, half_t, 2)
CLN_DOT(dot_prod_f16x8_pack_f32, half_t, 8)

// (a [M,K], x [K], y [M], M, K, stream) -- reference `void sgemv_*(Tensor a, Tensor x, Tensor y)`; K constraints
// as in the reference bindings (sgemv.cu:138-190: K % 32, K % 128, K == 16).
// The one-element rungs on long rows (K a multiple of 64, >= 512) give the whole wave to ONE row: a load instruction then covers one contiguous run of 64
// elements instead of two 32-element runs of two rows ([65536,1024]: hgemv_k32_f16 28.1 -> 26.6 us, sgemv_k32_f32 44.6 -> 42.7; the x4 rungs lose --
// 26.4 -> 31.0 / 42.9 -> 48.2 -- with half the pieces of a row in flight per lane, and keep 32 lanes per row).
#define CLN_GEMV(name, T, VEC, G, COND)                                                              \
  CLN_API int name(const void* a, const void* x, void* y, int M, int K, void* stream) {              \
    if (!(COND)) return CLN_ERR_UNSUPPORTED;                                                          \
    if (G == 32 && VEC == 1 && sizeof(T) == 2 && K % 64 == 0 && K >= 512 && M >= 4096)                                                  \
      return M >= 16384 ? launch_gemv_rows<T, VEC, 4, 8>(a, x, y, M, K, (hipStream_t)stream)                                            \
                        : launch_gemv_rows<T, VEC, 2, 8>(a, x, y, M, K, (hipStream_t)stream);                                           \
    if (G == 32 && VEC == 1 && K % 64 == 0 && K >= 512) return launch_gemv<T, VEC, 64>(a, x, y, M, K, (hipStream_t)stream); \
    if (G == 32 && VEC > 1 && sizeof(T) == 2 && K % (64 * VEC) == 0 && K >= 4 * 64 * VEC && M >= 4096)                                   \
      return M >= 16384 ? launch_gemv_rows<T, VEC, 4, 4>(a, x, y, M, K, (hipStream_t)stream)                                            \
                        : launch_gemv_rows<T, VEC, 2, 4>(a, x, y, M, K, (hipStream_t)stream);                                           \
    return launch_gemv<T, VEC, G>(a, x, y, M, K, (hipStream_t)stream);                                \
  }
CLN_GEMV(sgemv_k32_f32, float, 1, 32, K % 32 == 0)
CLN_GEMV(sgemv_k128_f32x4, float, 4, 32, K % 128 == 0)
CLN_GEMV(sgemv_k16_f32, float, 1, 16, K == 16)
CLN_GEMV(hgemv_k32_f16, half_t, 1, 32, K % 32 == 0)
CLN_GEMV(hgemv_k128_f16x4, half_t, 4, 32, K % 128 == 0)
CLN_GEMV(hgemv_k16_f16, half_t, 1, 16, K == 16)

// (x [row,col], y [col,row], row, col, stream) -- reference `void mat_transpose_*(Tensor x, Tensor y)`
#define CLN_TR(name, KIND)                                                          \
  CLN_API int name(const void* x, void* y, int row, int col, void* stream) {        \
    return launch_tr(KIND, x, y, row, col, (hipStream_t)stream);                    \
  }
CLN_TR(mat_transpose_f32_col2row, TR_READ1)
CLN_TR(mat_transpose_f32x4_col2row, TR_READ4)
CLN_TR(mat_transpose_f32_row2col, TR_WRITE1)
CLN_TR(mat_transpose_f32x4_row2col, TR_WRITE4)
CLN_TR(mat_transpose_f32_col2row2d, TR_READ1)
CLN_TR(mat_transpose_f32x4_col2row2d, TR_READ4_2D)
CLN_TR(mat_transpose_f32_row2col2d, TR_WRITE1)
CLN_TR(mat_transpose_f32x4_row2col2d, TR_WRITE4_2D)
CLN_TR(mat_transpose_f32_diagonal2d, TR_DIAG)
CLN_TR(mat_transpose_f32x4_shared_col2row2d, TR_LDS)
CLN_TR(mat_transpose_f32x4_shared_row2col2d, TR_LDS)
CLN_TR(mat_transpose_f32x4_shared_bcf_col2row2d, TR_LDS_BCF)
CLN_TR(mat_transpose_f32x4_shared_bcf_row2col2d, TR_LDS_BCF)
