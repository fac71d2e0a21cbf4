// Compile unit of the sum-checked optimistic-softmax attention kernels (flash_attn_m16x.cuh). It is built with
// -fno-slp-vectorize (see _build.py EXTRA_FLAGS): hipcc's SLP pass pairs the per-score f32 row-sum adds of neighbouring
// steps into v_pk_add_f32, which drags the exponentials of a whole phase behind its last MFMA and is slower than two
// plain adds beside MFMAs (MI355X_MICROARCH.md, per-instruction constants). Linked into the product library
// (and, as a dependency of nothing else, not into the probe library, which has its own unit probe/flash_attn_m16x_probe.hip).
#include "flash_attn_m16x.cuh"
#include "flash_attn_m16x_api.h"

namespace fa2 {

// The dispatched forms (fa2_plan, flash_attn.hip). 32 rows per wave (256-row workgroups, 128-key tiles): NDEF = 4 (half of the
// exponentials under the PV MFMAs), phase-A priority, split prologue -- the best of profiles/r03_fa_m16x_probe.log at both head
// dims. 64 rows per wave (D = 64, 512-row workgroups, 64-key tiles; long sequences): NDEF = 1 of the 4 key blocks.
// `one_stage` (the names' stages = 1): the same kernels with each tile requested in one burst and waited for where it is requested
// (M16X_ONE_STAGE), bit-identical output.
// `f32_scale` (the *_acc_f32 names, V as [B,H,N,D] only): the same kernels with Q as loaded and the scores scaled in fp32 (M16X_FSCALE).
int m16x_run(int D, int rows_per_wave, bool vt, bool one_stage, bool f32_scale, const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t s) {
  constexpr int OX = M16X_PRIO | M16X_SPLIT_PROLOGUE;
  constexpr int O1 = OX | M16X_ONE_STAGE | (M16X_ONE_POS << M16X_ONE_POS_SHIFT);
  if (f32_scale) {
    if (vt) return CLN_ERR_UNSUPPORTED;
    constexpr int F = OX | M16X_FSCALE, F1 = O1 | M16X_FSCALE;
    // 32 rows per wave: the row sums come from the matrix pipe as well (M16X_MFMA_SUM: l is then the sum of the SAME fp16 P values the numerator
    // uses) -- same speed as the fp32 scale alone, max-abs-error on amplified-key inputs 1.0e-3 instead of 1.6-2.0e-3 (profiles/r04_fa_fscale_probe.log);
    // the 64-row form has no registers for the extra accumulators (623 vs 1013 TF) and keeps the VALU row sums
    constexpr int FM = F | M16X_MFMA_SUM, FM1 = F1 | M16X_MFMA_SUM;
    if (D == 64 && rows_per_wave == 32) return one_stage ? launch_m16x<64, 32, 128, 8, 4, FM1>(q, k, v, o, B, H, N, s) : launch_m16x<64, 32, 128, 8, 4, FM>(q, k, v, o, B, H, N, s);
    if (D == 128 && rows_per_wave == 32) return one_stage ? launch_m16x<128, 32, 128, 4, 4, FM1>(q, k, v, o, B, H, N, s) : launch_m16x<128, 32, 128, 4, 4, FM>(q, k, v, o, B, H, N, s);
    if (D == 64 && rows_per_wave == 64) return one_stage ? launch_m16x<64, 64, 64, 4, 1, F1>(q, k, v, o, B, H, N, s) : launch_m16x<64, 64, 64, 4, 1, F>(q, k, v, o, B, H, N, s);
    return CLN_ERR_UNSUPPORTED;
  }
  if (one_stage) {
    if (!vt) {
      if (D == 64 && rows_per_wave == 32) return launch_m16x<64, 32, 128, 8, 4, O1>(q, k, v, o, B, H, N, s);
      if (D == 128 && rows_per_wave == 32) return launch_m16x<128, 32, 128, 4, 4, O1>(q, k, v, o, B, H, N, s);
      if (D == 64 && rows_per_wave == 64) return launch_m16x<64, 64, 64, 4, 1, O1>(q, k, v, o, B, H, N, s);
    } else {
      if (D == 64 && rows_per_wave == 32) return launch_m16x<64, 32, 128, 8, 4, O1, true>(q, k, v, o, B, H, N, s);
      if (D == 128 && rows_per_wave == 32) return launch_m16x<128, 32, 128, 4, 4, O1, true>(q, k, v, o, B, H, N, s);
      if (D == 64 && rows_per_wave == 64) return launch_m16x<64, 64, 64, 4, 1, O1, true>(q, k, v, o, B, H, N, s);
    }
    return CLN_ERR_UNSUPPORTED;
  }
  if (!vt) {
    if (D == 64 && rows_per_wave == 32) return launch_m16x<64, 32, 128, 8, 4, OX>(q, k, v, o, B, H, N, s);
    if (D == 128 && rows_per_wave == 32) return launch_m16x<128, 32, 128, 4, 4, OX>(q, k, v, o, B, H, N, s);
    if (D == 64 && rows_per_wave == 64) return launch_m16x<64, 64, 64, 4, 1, OX>(q, k, v, o, B, H, N, s);
  } else {  // V given as [B,H,D,N] (the three *_swizzle_qkv names): the same kernels with the V^T tile image and plain 8-byte fragment reads
    if (D == 64 && rows_per_wave == 32) return launch_m16x<64, 32, 128, 8, 4, OX, true>(q, k, v, o, B, H, N, s);
    if (D == 128 && rows_per_wave == 32) return launch_m16x<128, 32, 128, 4, 4, OX, true>(q, k, v, o, B, H, N, s);
    if (D == 64 && rows_per_wave == 64) return launch_m16x<64, 64, 64, 4, 1, OX, true>(q, k, v, o, B, H, N, s);
  }
  return CLN_ERR_UNSUPPORTED;
}

}  // namespace fa2
