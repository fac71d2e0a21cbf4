// Head dims above 256 ("fine-grained tiling" rungs of the reference:
// kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qk.cu:72, flash_attn_mma_tiling_qkv.cu:70 --
// config C5 is D = 512). The reference streams Q, K and V through O(1) shared memory in 16-wide d slices; here
// the head dim is split across the waves of a query-row group instead (each wave keeps its 256 columns of Q and of
// O^T in registers and the partial S^T tiles are summed through LDS):
//   512        pairs of waves, two 4-wave groups one phase apart, K/V double-buffered   (flash_attn_dsplit.cuh)
//   640, 768, 1024  FOUR waves, one per SIMD, each with all 64 rows of a quarter of d, softmax once per row by the row's owner wave
//              (flash_attn_dw4.cuh, round 5); before: quads of waves x 2 row groups, K / V through two-slot rings of 16-key tiles (flash_attn_dring.cuh,
//              round 3, now probe-only); its predecessor probe/flash_attn_dwide.cuh -- one 32-key K tile and one V tile, no prefetch, D = 640 padded
//              to 768 -- lives on in the probe library)
//   320, 384   (round 6: flash_attn_pair2.cuh with DREAL, planned in flash_attn.hip; the d-split form of rounds 2-5 is probe-only)
#pragma once
#include "flash_attn_dsplit.cuh"
#include "flash_attn_dw4.cuh"

namespace fa {
inline int launch_fa2_large_d(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D,
                              hipStream_t s, bool one_stage = false) {
  if (one_stage) {  // `stages = 1`: the same kernels with every tile request waited for where it is issued (no load under compute); d-split: one burst per tile
    constexpr int O1 = fa2::OPT_DEFAULT | fa2::OPT_1STAGE;
    switch (D) {
      case 640: return fa2::launch_dw4<640, fa2::DW4_CARRY | fa2::DW4_UNROLL2 | fa2::DW4_1STAGE>(q, k, v, o, B, H, N, s);
      case 768: return fa2::launch_dw4<768, fa2::DW4_DEFAULT | fa2::DW4_UNROLL2 | fa2::DW4_1STAGE>(q, k, v, o, B, H, N, s);
      case 1024: return fa2::launch_dw4<1024, fa2::DW4_DEFAULT | fa2::DW4_UNROLL2 | fa2::DW4_1STAGE>(q, k, v, o, B, H, N, s);
      default: return CLN_ERR_UNSUPPORTED;
    }
  }
  switch (D) {
    // (D = 320 / 384 ran the d-split kernel on the D = 512 LDS geometry in rounds 2-5 -- 745-817 / 793-876 TF at [1,16,4096,D], profiles/r02_fa_native_320_384.log;
    // since round 6 they run fa2_fwd_pair2<DREAL> on the same geometry (flash_attn_pair2.cuh: +12-16 %, profiles/r06_pair2_d320_d384.log); the d-split form is probe
    // variant 210 / 220 of kind 8)
    // (D = 512, config C5, is planned onto the 16x16x32 pair kernel of flash_attn_m16.cuh since round 3; this 32x32x16 form --
    // 990-1010 TF at [1,32,4096,512] -- stays as the geometry of 320 / 384 and as probe variant 210)
    // D = 640 / 768 / 1024 (flash_attn_dring.cuh): [1,16,4096,D] 604 -> 649 / 684 -> 703 / 691 -> 780 TF, [1,8,8192,1024] 657 -> 804
    // over the d-wide kernel (profiles/r03_fa_dring_probe.log); the two row groups run one phase apart at D = 1024 only
    // (lock-step measured 1-4 % faster at 640 / 768)
    // two K and two V fragments in flight at 640 / 768 (+5 % / +1.5 % over one; deeper: no gain; D = 1024 has no registers for it:
    // profiles/r03_fa_dring_prefetch_probe.log)
    // round 5: the one-wave-per-SIMD kernel (flash_attn_dw4.cuh) -- [1,16,4096,D] 684 -> 765 / 751 -> 818 / 793 -> 912 TF, [1,8,8192,1024] 803 -> 928 on one box
    // (profiles/r05_fa_dw4_probe.log); the 8-wave ring kernel above stays in the probe library (variants 1000..1216)
    case 640: return fa2::launch_dw4<640, fa2::DW4_CARRY | fa2::DW4_UNROLL2>(q, k, v, o, B, H, N, s);
    case 768: return fa2::launch_dw4<768, fa2::DW4_DEFAULT | fa2::DW4_UNROLL2>(q, k, v, o, B, H, N, s);
    case 1024: return fa2::launch_dw4<1024, fa2::DW4_DEFAULT | fa2::DW4_UNROLL2>(q, k, v, o, B, H, N, s);
    default: return CLN_ERR_UNSUPPORTED;
  }
}
}  // namespace fa
