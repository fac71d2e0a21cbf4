// C-ABI entry points of the HGEMM library: one symbol per function exported by the reference's
// pybind module (kernels/hgemm/pybind/hgemm.cc:58-107), same names, same argument meaning.
//
//   G3:  int name(a, b, c, M, N, K, stream)
//   G6:  int name(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)
//
// a: [M,K] row-major fp16; b: [K,N] row-major fp16 (NN) or storage [N,K] (TN, reference
// as_col_major kernels/hgemm/tools/utils.py:135-140); c: [M,N] row-major fp16.
// Which names are distinct gfx950 kernels and which are aliases is recorded in
// cuda-learn-notes_amd/manifest.py (generated table in DESIGN.md).
#include "hgemm_dispatch.h"
#include "hgemm_mfma.cuh"
#include "hgemm_valu.cuh"

using namespace hgemm;

namespace {

int check_args(const void* a, const void* b, const void* c, int M, int N, int K) {
  if (!a || !b || !c) return CLN_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(a) || !cln_aligned16(b) || !cln_aligned16(c)) return CLN_ERR_BAD_ARG;
  return CLN_OK;
}

// "best" policy shared by the reference's top rungs (warp4x4x2 family), from the size sweep in
// profiles/r01_hgemm_midsize_probe.log: 256x256 ping-pong tiles once they cover about half of the 256 CUs
// (3072^3: 144 tiles -> 969 TF vs 783 with 128x128); below that, 64x128 tiles while 128x128 tiles would leave
// CUs with fewer than two workgroups (2048^3: 712 vs 647 TF; 1024^3: 182 vs 155), else 128x128.
int best_tile(int M, int N) {
  if (M % 256 == 0 && N % 256 == 0 && (M / 256) * (N / 256) >= 120) return T256;
  if (M % 64 == 0 && N % 128 == 0 && (long long)((M + 127) / 128) * (N / 128) < 512) return T64x128;
  return T128;
}

// Top rungs (reference warp4x4x2 family): for 256x256-tileable problems the `stages` knob selects a
// distinct pipeline structure: 2 -> quadrant ping-pong over a 2 x 64-deep ring with split DMA,
// 4 -> k-half ping-pong over a 4 x 32-deep ring, 3/5 -> plain multi-stage ring.
template <int LAYOUT>
int best_dispatch(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle, int stride,
                  hipStream_t st) {
  const int tile = best_tile(M, N);
  if (tile == T256) {
    if ((stages == 2 || stages < 2 || stages > 5) && K % 64 == 0)
      return launch_pp<LAYOUT, 2, 4, 0, 1>(a, b, c, M, N, K, swizzle, stride, st);
    if (stages == 4 && K % 32 == 0) return launch_pp32<LAYOUT, 2>(a, b, c, M, N, K, swizzle, stride, st);
  }
  return LAYOUT == TN ? ring_dispatch_tn(tile, a, b, c, M, N, K, stages, swizzle, stride, st)
                      : ring_dispatch_nn(tile, a, b, c, M, N, K, stages, swizzle, stride, st);
}

using C1S_128_NN = Cfg<128, 128, 32, 2, 2, 1, NN>;
using C1S_64x128_NN = Cfg<64, 128, 32, 1, 2, 1, NN>;

}  // namespace

#define CLN_G3(name, expr)                                                                        \
  CLN_API int name(const void* a, const void* b, void* c, int M, int N, int K, void* stream_) {  \
    int rc = check_args(a, b, c, M, N, K);                                                        \
    if (rc != CLN_OK) return rc;                                                                  \
    hipStream_t stream = (hipStream_t)stream_;                                                    \
    return (expr);                                                                                \
  }
#define CLN_G6(name, expr)                                                                        \
  CLN_API int name(const void* a, const void* b, void* c, int M, int N, int K, int stages,       \
                   int swizzle, int swizzle_stride, void* stream_) {                              \
    int rc = check_args(a, b, c, M, N, K);                                                        \
    if (rc != CLN_OK) return rc;                                                                  \
    hipStream_t stream = (hipStream_t)stream_;                                                    \
    return (expr);                                                                                \
  }

// ---- VALU rungs (reference kernels/hgemm/naive/hgemm.cu:784-998, hgemm_async.cu:734-908) -------
CLN_G3(hgemm_naive_f16, launch_valu_naive(a, b, c, M, N, K, stream))
CLN_G3(hgemm_sliced_k_f16, launch_valu_sliced_k(a, b, c, M, N, K, stream))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4_pack, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4_bcf, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x4_pack_bcf, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x8_pack_bcf, (launch_valu_tile<8, 8, false, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf, (launch_valu_tile<8, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf, (launch_valu_tile<16, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async,
       (launch_valu_tile<16, 8, true, true>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf, (launch_valu_tile<32, 8, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async,
       (launch_valu_tile<32, 8, true, true>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf, (launch_valu_tile<32, 16, true, false>(a, b, c, M, N, K, stream)))
CLN_G3(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async,
       (launch_valu_tile<32, 16, true, true>(a, b, c, M, N, K, stream)))

// ---- matrix-core rungs, no `stages` argument ----------------------------------------------------
// reference kernels/hgemm/wmma/hgemm_wmma.cu:594-758, kernels/hgemm/mma/basic/hgemm_mma.cu:270-336
CLN_G3(hgemm_wmma_m16n16k16_naive, launch_naive<NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_mma_m16n8k16_naive, launch_naive<NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_wmma_m16n16k16_mma4x2, launch_1stage<C1S_64x128_NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_wmma_m16n16k16_mma4x2_warp2x4, launch_1stage<C1S_128_NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_mma_m16n8k16_mma2x4_warp4x4, launch_1stage<C1S_128_NN>(a, b, c, M, N, K, stream))
CLN_G3(hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async,
       ring_exact_nn(T128, (K % 64 == 0) ? 64 : 32, 2, a, b, c, M, N, K, 0, 1, stream))
CLN_G3(hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async,
       ring_exact_nn(T128, 32, 2, a, b, c, M, N, K, 0, 1, stream))

// ---- multi-stage rings --------------------------------------------------------------------------
// reference kernels/hgemm/wmma/hgemm_wmma_stage.cu:1001-1464
CLN_G6(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem,
       ring_dispatch_nn(T256x128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem,
       ring_dispatch_nn(T256, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
// reference kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2124-2717, mma/swizzle/hgemm_mma_stage_swizzle.cu:757
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem,
       ring_dispatch_nn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle,
       best_dispatch<NN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
// TN family: reference hgemm_mma_stage_tn.cu:517, hgemm_mma_stage_tn_swizzle_x4.cu:860,
// cutlass/hgemm_mma_stage_tn_cute.cu:521 (128x256 tile)
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn,
       ring_dispatch_tn(T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4,
       best_dispatch<TN>(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream))
CLN_G6(hgemm_mma_stages_block_swizzle_tn_cute,
       ring_dispatch_tn((N % 256 == 0) ? T128x256 : T128, a, b, c, M, N, K, stages, swizzle, swizzle_stride,
                        stream))

// ---- tuning / test hooks (not part of the reference surface) -----------------------------------
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
