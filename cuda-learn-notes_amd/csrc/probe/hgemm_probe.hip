// Tuning / ablation hook of the HGEMM kernels -- TEST-ONLY library (libcln_amd_probe.so), never part of the product
// libcln_amd.so: explicit (tile, BK, stages) ring instantiations, the ping-pong variants and their no-store / ablation
// forms (some ablations produce garbage by design; they exist to price one phase of the kernel).
#include "hgemm_dispatch.h"
#include "hgemm_mfma.cuh"
#include "hgemm_w4.cuh"
#include "hgemm_w4s.cuh"
#include "hgemm_splitk.cuh"

using namespace hgemm;

namespace {
int check_args(const void* a, const void* b, const void* c, int M, int N, int K) {
  if (!a || !b || !c) return CLN_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(a) || !cln_aligned16(b) || !cln_aligned16(c)) return CLN_ERR_BAD_ARG;
  return CLN_OK;
}
using C1S_128_NN = Cfg<128, 128, 32, 2, 2, 1, NN>;
}  // namespace

// layout: 0 NN, 1 TN. kind: 0 ring (tile,bk,stages), 1 single-stage 128x128x32, 2 naive.
CLN_API int cln_hgemm_variant(int kind, int layout, int tile, int bk, int stages, const void* a, const void* b,
                              void* c, int M, int N, int K, int swizzle, int swizzle_stride, void* stream_) {
  int rc = check_args(a, b, c, M, N, K);
  if (rc != CLN_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (kind == 0) {
    return layout == TN ? ring_exact_tn(tile, bk, stages, a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : ring_exact_nn(tile, bk, stages, a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 1) {
    return layout == TN ? launch_1stage<Cfg<128, 128, 32, 2, 2, 1, TN>>(a, b, c, M, N, K, stream)
                        : launch_1stage<C1S_128_NN>(a, b, c, M, N, K, stream);
  }
  if (kind == 3) {  // ping-pong 256x256x64
    return layout == TN ? launch_pp<TN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp<NN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 4) return launch_pp<NN, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);  // no-store probe
  if (kind == 5) {  // ping-pong + LDS-staged epilogue; `stages` selects 8 or 4 slots per K tile
    if (stages == 4)
      return layout == TN ? launch_pp<TN, 2, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                          : launch_pp<NN, 2, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    return layout == TN ? launch_pp<TN, 2, 8>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp<NN, 2, 8>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 8) {  // 4-slot ping-pong, split DMA, LDS epilogue (stages==1: no-store probe)
    if (stages == 1) return launch_pp<NN, 1, 4, 0, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    return layout == TN ? launch_pp<TN, 2, 4, 0, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp<NN, 2, 4, 0, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 9) {  // k-half ping-pong (BK=32 sub-tiles, 4-deep ring); stages==1: no-store probe
    if (stages == 1) return launch_pp32<NN, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    return layout == TN ? launch_pp32<TN, 2>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp32<NN, 2>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 10) {  // ping-pong on mfma_32x32x16; stages==1: no-store probe; stages>=16: ablation bits = stages-16
    if (stages == 1) return launch_m32<NN, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    if (stages >= 16) {
      switch (stages - 16) {
        case 1: return launch_m32<NN, 1, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
        case 2: return launch_m32<NN, 1, 2>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
        case 3: return launch_m32<NN, 1, 3>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
        case 7: return launch_m32<NN, 1, 7>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
        case 8: return launch_m32<NN, 2, 8>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
        default: return CLN_ERR_BAD_ARG;
      }
    }
    return layout == TN ? launch_m32<TN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_m32<NN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 11) {  // 192-row ping-pong (4 slots, un-split DMA, LDS epilogue); stages == 1: the 256-row kernel in the same form
    if (stages == 1) return layout == TN ? launch_pp<TN, 2, 4, 0, 0, 256>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                                         : launch_pp<NN, 2, 4, 0, 0, 256>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    return layout == TN ? launch_pp<TN, 2, 4, 0, 0, 192>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp<NN, 2, 4, 0, 0, 192>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 12) {  // 2-slot ping-pong (64 MFMAs per compute slot); stages == 2: 192-row form
    if (stages == 2) return layout == TN ? launch_pp<TN, 2, 2, 0, 0, 192>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                                         : launch_pp<NN, 2, 2, 0, 0, 192>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    return layout == TN ? launch_pp<TN, 2, 2, 0, 0, 256>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp<NN, 2, 2, 0, 0, 256>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 13) {  // 4-slot ping-pong with the DMA issued inside the first compute slot; stages == 2: 192-row form
    if (stages == 2) return layout == TN ? launch_pp<TN, 2, 4, 0, 2, 192>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                                         : launch_pp<NN, 2, 4, 0, 2, 192>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    return layout == TN ? launch_pp<TN, 2, 4, 0, 2, 256>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                        : launch_pp<NN, 2, 4, 0, 2, 256>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  }
  if (kind == 14) {  // one wave per SIMD, 128x128 wave tiles; stages = schedule variant (+16: boustrophedon MFMA order), 100+ = no-store probes
#define W4_CASE(V)                                                                                         \
  case V: return layout == TN ? launch_w4<TN, 2, V>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)    \
                              : launch_w4<NN, 2, V>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    switch (stages) {
      W4_CASE(0) W4_CASE(1) W4_CASE(3) W4_CASE(4) W4_CASE(9) W4_CASE(10) W4_CASE(13)
      W4_CASE(20) W4_CASE(25) W4_CASE(26) W4_CASE(27) W4_CASE(28)
      case 104: return layout == TN ? launch_w4<TN, 1, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)
                                    : launch_w4<NN, 1, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 203: return layout == TN ? launch_w4<TN, 3, 26>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)   // non-temporal C stores
                                    : launch_w4<NN, 3, 26>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 204: return layout == TN ? launch_w4<TN, 4, 26>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)   // write-through C stores
                                    : launch_w4<NN, 4, 26>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 111: return launch_w4<NN, 1, 4, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);  // no-store + ablations
      case 112: return launch_w4<NN, 1, 4, 2>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 113: return launch_w4<NN, 1, 4, 3>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 114: return launch_w4<NN, 1, 4, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 118: return launch_w4<NN, 1, 4, 8>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);   // every DMA piece re-reads one KiB (issue cost without traffic)
      case 117: return launch_w4<NN, 1, 4, 7>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      default: return CLN_ERR_BAD_ARG;
    }
#undef W4_CASE
  }
  if (kind == 15) {  // one wave per SIMD on 192-row / 192-column tiles; tile = 0: 192x256, 1: 256x192, 2: 192x192, 3: 128x256, 4: 256x128
#define W4_SHAPE(BM, BN)                                                                                              \
  return layout == TN ? launch_w4<TN, 2, 26, 0, BM, BN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)           \
                      : launch_w4<NN, 2, 26, 0, BM, BN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    if (tile == 0) { W4_SHAPE(192, 256) }
    if (tile == 1) { W4_SHAPE(256, 192) }
    if (tile == 2) { W4_SHAPE(192, 192) }
    if (tile == 3) { W4_SHAPE(128, 256) }
    if (tile == 4) { W4_SHAPE(256, 128) }
    if (tile == 5) { W4_SHAPE(160, 160) }
#undef W4_SHAPE
    return CLN_ERR_BAD_ARG;
  }
  if (kind == 16) {  // one wave per SIMD over a `stages`-deep ring of 32-deep K slots (hgemm_w4s.cuh); S = 2 exists here only
#define W4S_CASE(SS)                                                                                     \
  case SS: return layout == TN ? launch_w4s<TN, SS>(a, b, c, M, N, K, swizzle, swizzle_stride, stream)   \
                               : launch_w4s<NN, SS>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
    switch (stages) {
      W4S_CASE(2) W4S_CASE(3) W4S_CASE(4) W4S_CASE(5)
      default: return CLN_ERR_BAD_ARG;
    }
#undef W4S_CASE
  }
  if (kind == 17) {  // split-K over the one-wave-per-SIMD kernel (hgemm_splitk.cuh): `tile` = shape, `stages` = number of splits
    static float* ws = nullptr;  // probe-only workspace: 512 MiB, allocated once
    constexpr size_t WS_BYTES = 512u << 20;
    if (!ws && hipMalloc(&ws, WS_BYTES) != hipSuccess) return ws = nullptr, CLN_ERR_LAUNCH;
    if ((size_t)stages * M * N * 4 > WS_BYTES) return CLN_ERR_UNSUPPORTED;
#define SK_SHAPE(TT, BMM, BNN)                                                                                              \
  if (tile == TT) return layout == TN ? launch_w4_splitk<TN, 26, BMM, BNN>(a, b, c, ws, M, N, K, stages, stream)           \
                                      : launch_w4_splitk<NN, 26, BMM, BNN>(a, b, c, ws, M, N, K, stages, stream);
    SK_SHAPE(0, 256, 256) SK_SHAPE(1, 128, 256) SK_SHAPE(2, 256, 128) SK_SHAPE(3, 192, 256) SK_SHAPE(4, 192, 192) SK_SHAPE(5, 160, 160)
#undef SK_SHAPE
    return CLN_ERR_BAD_ARG;
  }
  if (kind == 20) {  // (round 6) kind 17 as ONE launch: the last-arriving workgroup of a tile reduces (EPI 6) at ANY number of splits -- the product takes this form
    // at 2 splits only (csrc/hgemm.hip splitk_fused_max_s); tools/hg_splitk_fused_probe.py times 17 against 20 on the planner's (tile, splits)
    static float* ws = nullptr;  // probe-only workspace: tickets + 512 MiB of partials, allocated and zeroed once (the tickets reset themselves)
    constexpr size_t WS_BYTES = 512u << 20;
    if (!ws) {
      if (hipMalloc(&ws, WS_BYTES + W4_TICKET_FLOATS * 4) != hipSuccess) return ws = nullptr, CLN_ERR_LAUNCH;
      if (hipMemset(ws, 0, W4_TICKET_FLOATS * 4) != hipSuccess) return CLN_ERR_LAUNCH;
    }
    if ((size_t)stages * M * N * 4 > WS_BYTES) return CLN_ERR_UNSUPPORTED;
#define SKF_SHAPE(TT, BMM, BNN)                                                                                                  \
  if (tile == TT) return layout == TN ? launch_w4_splitk_fused<TN, 26, BMM, BNN>(a, b, c, ws, M, N, K, stages, stream)           \
                                      : launch_w4_splitk_fused<NN, 26, BMM, BNN>(a, b, c, ws, M, N, K, stages, stream);
    SKF_SHAPE(0, 256, 256) SKF_SHAPE(1, 128, 256) SKF_SHAPE(3, 192, 256) SKF_SHAPE(4, 192, 192) SKF_SHAPE(5, 160, 160)
#undef SKF_SHAPE
    return CLN_ERR_BAD_ARG;
  }
  if (kind == 18) {  // tail split: `tile` = tile rows (of 256) given to split-K, `stages` = number of splits
    static float* ws = nullptr;
    constexpr size_t WS_BYTES = 512u << 20;
    if (!ws && hipMalloc(&ws, WS_BYTES) != hipSuccess) return ws = nullptr, CLN_ERR_LAUNCH;
    if ((size_t)stages * tile * 256 * N * 4 > WS_BYTES) return CLN_ERR_UNSUPPORTED;
    const int m_split = M - tile * 256;
    return layout == TN ? launch_w4_tail_split<TN, 3, 26, 256, 256>(a, b, c, ws, M, N, K, m_split, stages, swizzle, swizzle_stride, stream)
                        : launch_w4_tail_split<NN, 3, 26, 256, 256>(a, b, c, ws, M, N, K, m_split, stages, swizzle, swizzle_stride, stream);
  }
  // 19 = the persistent tile walk of the one-wave-per-SIMD kernel (hgemm_w4.cuh EPI 7, round 5): one workgroup per CU walks its tiles, the next tile's
  // prologue requested before the C store. Measured +-1 % against one workgroup per tile (profiles/r05_hgemm_persist_probe.log): not in the product.
  if (kind == 19) return layout == TN ? launch_w4_persist<TN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream) : launch_w4_persist<NN>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
  if (kind == 6) return launch_pp<NN, 1, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);  // 4-slot no-store probe
  if (kind == 7) {  // ablations of the 4-slot no-store probe; `stages` = ABL bits (results are garbage by design)
    switch (stages) {
      case 1: return launch_pp<NN, 1, 4, 1>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 2: return launch_pp<NN, 1, 4, 2>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 3: return launch_pp<NN, 1, 4, 3>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 7: return launch_pp<NN, 1, 4, 7>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 8: return launch_pp<NN, 1, 4, 8>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      case 4: return launch_pp<NN, 1, 4, 4>(a, b, c, M, N, K, swizzle, swizzle_stride, stream);
      default: return CLN_ERR_BAD_ARG;
    }
  }
  if (kind == 2) return layout == TN ? launch_naive<TN>(a, b, c, M, N, K, stream) : launch_naive<NN>(a, b, c, M, N, K, stream);
  return CLN_ERR_BAD_ARG;
}
