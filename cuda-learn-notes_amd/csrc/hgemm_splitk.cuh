// Split-K for the one-wave-per-SIMD HGEMM (hgemm_w4.cuh): problems whose M x N gives far fewer tiles than the chip has CUs
// but whose K is long (1024 x 1024 x 16384: 16 tiles of 256 x 256; 128 x 8192 x 8192: 32 tiles of 128 x 256).
//
// The reference has no such kernel (its sweep is M = N = K, hgemm.py:277-281); without it these shapes fall to the 64 x 64 ring
// (a workgroup on every CU, but 0.44 of the big kernel's rate per CU and every operand byte fetched 4x as often):
// profiles/r04_hgemm_rect_probe_before.log -- 0.59-0.9 of rocBLAS, which splits K itself.
//
//   launch 1: grid (tiles, S). Workgroup (t, s) runs the UNCHANGED hgemm_w4 main loop over K columns [s K/S, (s+1) K/S) of tile t
//             (EPI 5: leading dimension != loop extent) and writes its 256 fp32 accumulators per lane as they lie in the registers:
//             workspace [S][tiles][4 waves][FM x FN fragments][64 lanes] x 16 B -- every store instruction 1 KiB contiguous, no LDS pass.
//   launch 2: hgemm_splitk_reduce -- one workgroup per (tile, wave, 16-row fragment row): sums the S partials in fp32 (ascending s),
//             rounds ONCE to fp16, transposes the 16 x (BN/2) strip through LDS and writes whole 16-byte row segments of C.
// Numerics: fp32 accumulation throughout, one rounding -- the same as the single-pass kernel up to the fp32 summation order.
#pragma once
#include "hgemm_w4.cuh"

namespace hgemm {

template <int BM, int BN>
__global__ __launch_bounds__(256) void hgemm_splitk_reduce_kernel(const float* __restrict__ ws, half_t* __restrict__ Cmat, int N, int tiles_n,
                                                                  int tiles, int S) {
  constexpr int FM = BM / 32, FN = BN / 32, WTM = BM / 2, WTN = BN / 2;
  constexpr int RS = FN * 32 + 16;  // LDS row stride in bytes (16 rows of WTN halves + pad)
  __shared__ __attribute__((aligned(16))) char strip[16 * RS];
  const int i = blockIdx.x % FM, tw = blockIdx.x / FM, w = tw & 3, tile = tw >> 2;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t split_stride = (size_t)tiles * (BM * BN);
  for (int j = wv; j < FN; j += 4) {
    const float* p = ws + (size_t)tile * (BM * BN) + w * (WTM * WTN) + ((i * FN + j) * 64 + lane) * 4;
    f4 s = *reinterpret_cast<const f4*>(p);
    int ks = 1;
    for (; ks + 3 < S; ks += 4) {  // four loads in flight; summed in ascending split order
      const f4 v0 = *reinterpret_cast<const f4*>(p + (size_t)ks * split_stride);
      const f4 v1 = *reinterpret_cast<const f4*>(p + (size_t)(ks + 1) * split_stride);
      const f4 v2 = *reinterpret_cast<const f4*>(p + (size_t)(ks + 2) * split_stride);
      const f4 v3 = *reinterpret_cast<const f4*>(p + (size_t)(ks + 3) * split_stride);
      s = s + v0;
      s = s + v1;
      s = s + v2;
      s = s + v3;
    }
    for (; ks < S; ++ks) s = s + *reinterpret_cast<const f4*>(p + (size_t)ks * split_stride);
    // lane l of fragment (i, j) holds row l & 15, columns 16 j + 4 (l >> 4) ... + 3 (the layout store_wide_tile_via_lds reads)
    const h4 o = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
    *reinterpret_cast<h4*>(strip + (lane & 15) * RS + (j * 16 + 4 * (lane >> 4)) * 2) = o;
  }
  __syncthreads();
  constexpr int LPR = FN * 2;  // 16-byte pieces per row
  if (threadIdx.x < 16 * LPR) {
    const int r = threadIdx.x / LPR, c = threadIdx.x - r * LPR;
    const u4 v = *reinterpret_cast<const u4*>(strip + r * RS + c * 16);
    const size_t row = (size_t)tm * BM + (w >> 1) * WTM + i * 16 + r;
    *reinterpret_cast<u4*>(Cmat + row * N + (size_t)tn * BN + (w & 1) * WTN + c * 8) = v;
  }
}

// K / S must be a K the kernel's peeled structure covers (w4_k_ok); the workspace holds S * M * N floats
inline bool w4_splitk_ok(int K, int S) { return S >= 2 && K % (64 * S) == 0 && w4_k_ok(K / S) && K / 64 < (1 << 22); }

template <int LAYOUT, int VAR, int BM, int BN>
int launch_w4_splitk(const void* a, const void* b, void* c, float* ws, int M, int N, int K, int S, hipStream_t stream) {
  using C = W4Cfg<BM, BN, LAYOUT>;
  if (M % BM || N % BN || !w4_splitk_ok(K, S) || ws == nullptr) return CLN_ERR_UNSUPPORTED;
  const int tiles_m = M / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n, Kl = K / S;
  if (S > 65535) return CLN_ERR_UNSUPPORTED;
  const int sw = (K / 64) << 8;  // no block swizzle (few tiles), plain stores; bits 8..: the leading dimension in units of 64
  if ((Kl / 64) & 1) {
    static cln_lds_attr lds_attr_odd;
    if (cln_ensure_lds(lds_attr_odd, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, 5, VAR, 0, BM, BN, true>), C::LDS_BYTES) != CLN_OK)
      return CLN_ERR_LAUNCH;
    CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, 5, VAR, 0, BM, BN, true>), dim3(tiles, S), dim3(256), C::LDS_BYTES, stream, (const half_t*)a, (const half_t*)b,
               reinterpret_cast<half_t*>(ws), M, N, Kl, tiles_m, tiles_n, sw, tiles_n);
  } else {
    static cln_lds_attr lds_attr;
    if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, 5, VAR, 0, BM, BN>), C::LDS_BYTES) != CLN_OK)
      return CLN_ERR_LAUNCH;
    CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, 5, VAR, 0, BM, BN>), dim3(tiles, S), dim3(256), C::LDS_BYTES, stream, (const half_t*)a, (const half_t*)b,
               reinterpret_cast<half_t*>(ws), M, N, Kl, tiles_m, tiles_n, sw, tiles_n);
  }
  int rc = cln_check_launch();
  if (rc != CLN_OK) return rc;
  CLN_LAUNCH((hgemm_splitk_reduce_kernel<BM, BN>), dim3(tiles * 4 * (BM / 32)), dim3(256), 0, stream, (const float*)ws, (half_t*)c, N, tiles_n, tiles, S);
  return cln_check_launch();
}

// ONE launch (round 5): the same grid with EPI 6 -- the last-arriving workgroup of a tile sums the partials and stores C
// (hgemm_w4.cuh). `ws` = W4_TICKET_FLOATS zeroed tickets + S * M * N floats. The reduction of a tile runs on ONE CU instead of 4 * BM / 32
// workgroups of the reduce kernel, so it pays while S is small (the reduce launch costs ~5 us; a CU sums ~1 MiB of partials in about that time):
// the planner (hgemm.hip splitk_fused_max_s) takes this form at 2 splits and the two-launch form above: measurements there.
inline size_t w4_splitk_ws_bytes(int M, int N, int S) { return (size_t)W4_TICKET_FLOATS * 4 + (size_t)S * M * N * sizeof(float); }
template <int LAYOUT, int VAR, int BM, int BN>
int launch_w4_splitk_fused(const void* a, const void* b, void* c, float* ws, int M, int N, int K, int S, hipStream_t stream) {
  using C = W4Cfg<BM, BN, LAYOUT>;
  if (M % BM || N % BN || !w4_splitk_ok(K, S) || ws == nullptr) return CLN_ERR_UNSUPPORTED;
  const int tiles_m = M / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n, Kl = K / S;
  if (S > 65535 || tiles > W4_TICKET_FLOATS) return CLN_ERR_UNSUPPORTED;
  const int sw = (K / 64) << 8;
  if ((Kl / 64) & 1) {
    static cln_lds_attr lds_attr_odd;
    if (cln_ensure_lds(lds_attr_odd, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, 6, VAR, 0, BM, BN, true>), C::LDS_BYTES) != CLN_OK)
      return CLN_ERR_LAUNCH;
    CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, 6, VAR, 0, BM, BN, true>), dim3(tiles, S), dim3(256), C::LDS_BYTES, stream, (const half_t*)a, (const half_t*)b,
               (half_t*)c, M, N, Kl, tiles_m, tiles_n, sw, tiles_n, ws);
  } else {
    static cln_lds_attr lds_attr;
    if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_w4_kernel<LAYOUT, 6, VAR, 0, BM, BN>), C::LDS_BYTES) != CLN_OK)
      return CLN_ERR_LAUNCH;
    CLN_LAUNCH((hgemm_w4_kernel<LAYOUT, 6, VAR, 0, BM, BN>), dim3(tiles, S), dim3(256), C::LDS_BYTES, stream, (const half_t*)a, (const half_t*)b,
               (half_t*)c, M, N, Kl, tiles_m, tiles_n, sw, tiles_n, ws);
  }
  return cln_check_launch();
}

// Tail split (round 4): a tile count just past a whole number of rounds of 256 (4352^3: 289 tiles of 256 x 256, 5888^3: 529, 7168^3: 784) leaves the
// last round almost empty. The output is cut along M: rows [0, m_split) -- whole rounds of tiles -- run the single-pass kernel, the remaining
// tile rows run split-K so that they, too, spread over the chip. Both are the launchers above / in hgemm_w4.cuh on sub-matrices (row offsets
// only: leading dimensions unchanged), back to back on the caller's stream.
template <int LAYOUT, int EPI, int VAR, int BM, int BN, bool FUSED = false>
int launch_w4_tail_split(const void* a, const void* b, void* c, float* ws, int M, int N, int K, int m_split, int S, int swizzle, int swizzle_stride,
                         hipStream_t stream) {
  if (m_split <= 0 || m_split >= M || m_split % BM || (M - m_split) % BM) return CLN_ERR_UNSUPPORTED;
  int rc = launch_w4<LAYOUT, EPI, VAR, 0, BM, BN>(a, b, c, m_split, N, K, swizzle, swizzle_stride, stream);
  if (rc != CLN_OK) return rc;
  const half_t* a2 = reinterpret_cast<const half_t*>(a) + (size_t)m_split * K;
  half_t* c2 = reinterpret_cast<half_t*>(c) + (size_t)m_split * N;
  if constexpr (FUSED) return launch_w4_splitk_fused<LAYOUT, VAR, BM, BN>(a2, b, c2, ws, M - m_split, N, K, S, stream);  // `ws` = tickets + partials
  else return launch_w4_splitk<LAYOUT, VAR, BM, BN>(a2, b, c2, ws, M - m_split, N, K, S, stream);
}

}  // namespace hgemm
