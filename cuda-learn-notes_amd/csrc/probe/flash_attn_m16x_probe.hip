// TEST-ONLY compile unit: the probe variants of the sum-checked optimistic-softmax attention kernels
// (flash_attn_m16x.cuh; NDEF / priority / prologue forms measured in profiles/r03_fa_m16x_probe.log). Built, like
// flash_attn_m16x.hip, with
// -fno-slp-vectorize (see _build.py EXTRA_FLAGS): hipcc's SLP pass pairs the per-score f32 row-sum adds of neighbouring
// steps into v_pk_add_f32, which drags the exponentials of a whole phase behind its last MFMA and is slower than two
// plain adds beside MFMAs (MI355X_MICROARCH.md, per-instruction constants). Linked into the probe library only.
#include "flash_attn_m16x.cuh"
#include "flash_attn_m16s.cuh"
#include "flash_attn_m32x.cuh"
#include "flash_attn_m16x_api.h"

namespace fa2 {

// code = 16 * (NDEF - 1) + OX for the 32-rows-per-wave forms; 96 + ...: fragment prefetch depth 4 instead of 8 (D = 64);
// 120 + 16 * (NDEF - 1) + OX: 64 rows per wave (D = 64, 64-key tiles)
int m16x_probe_run(int D, int code, const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t s) {
#define MX(DD, CODE, RPWW, BCC, PDD, NDEFF, OXX) \
  if (D == DD && code == CODE) return launch_m16x<DD, RPWW, BCC, PDD, NDEFF, OXX>(q, k, v, o, B, H, N, s);
  MX(64, 5, 32, 128, 8, 1, 5) MX(64, 21, 32, 128, 8, 2, 5) MX(64, 33, 32, 128, 8, 3, 1) MX(64, 37, 32, 128, 8, 3, 5) MX(64, 38, 32, 128, 8, 3, 6)
  MX(64, 48, 32, 128, 8, 4, 0) MX(64, 49, 32, 128, 8, 4, 1) MX(64, 53, 32, 128, 8, 4, 5) MX(64, 54, 32, 128, 8, 4, 6) MX(64, 52, 32, 128, 8, 4, 4)
  MX(64, 69, 32, 128, 8, 5, 5) MX(64, 85, 32, 128, 8, 6, 5)
  MX(64, 101, 32, 128, 4, 4, 5)
  MX(64, 62, 32, 128, 8, 4, 21) MX(64, 134, 64, 64, 4, 1, 21)  // 21 = the shipped options + non-temporal O stores
  // ablations of the shipped D = 64 kernel (OX 5 + 32 K reads, + 64 V reads, + 128 exponentials, + 256 LDS-DMA; garbage results)
  MX(64, 70, 32, 128, 8, 4, 37) MX(64, 71, 32, 128, 8, 4, 69) MX(64, 72, 32, 128, 8, 4, 101) MX(64, 73, 32, 128, 8, 4, 133) MX(64, 74, 32, 128, 8, 4, 261)
  MX(64, 75, 32, 128, 8, 4, 357) MX(64, 76, 32, 128, 8, 4, 485)
  MX(64, 80, 32, 128, 8, 4, 1509) MX(64, 81, 32, 128, 8, 4, 1381)  // 876 / 875 without the two barriers per tile (485 / 357 + 1024)
  MX(64, 82, 32, 128, 8, 4, 2053) MX(64, 83, 32, 128, 8, 4, 2565) MX(128, 82, 32, 128, 4, 4, 2053)  // snake order over the query blocks (5 + 2048; 83: + paired QK)
  MX(64, 84, 32, 128, 8, 4, 4101) MX(64, 86, 32, 128, 8, 4, 6149) MX(128, 84, 32, 128, 4, 4, 4101)  // one softmax item behind each MFMA (5 + 4096; 86: + snake)
  MX(64, 87, 32, 128, 8, 4, 8197) MX(64, 88, 32, 128, 8, 4, 8196) MX(128, 87, 32, 128, 4, 4, 8197)  // late phase-A exponentials (5 + 8192; 88: without the priority flips)
  MX(64, 89, 32, 128, 8, 4, 16389)  // deferred blocks checked by their own row sums in phase B (5 + 16384); the D = 128 and the 64-row forms spill with it
  MX(64, 77, 32, 128, 8, 4, 517) MX(64, 78, 32, 128, 8, 2, 517) MX(64, 79, 32, 128, 8, 6, 517)  // 517 = the shipped options + QK^T steps of two key blocks interleaved
  MX(64, 60, 32, 128, 8, 4, 12) MX(64, 61, 32, 128, 8, 4, 13) MX(64, 133, 64, 64, 4, 1, 12)  // static priority for the second group (12), on top of the phase-A flips (13)
  MX(64, 120, 64, 64, 4, 1, 0) MX(64, 125, 64, 64, 4, 1, 5) MX(64, 141, 64, 64, 4, 2, 5)
  MX(128, 5, 32, 128, 4, 1, 5) MX(128, 21, 32, 128, 4, 2, 5) MX(128, 49, 32, 128, 4, 4, 1) MX(128, 53, 32, 128, 4, 4, 5) MX(128, 54, 32, 128, 4, 4, 6)
  MX(128, 85, 32, 128, 4, 6, 5)
  MX(128, 62, 32, 128, 4, 4, 21)
  MX(128, 60, 32, 128, 4, 4, 12) MX(128, 61, 32, 128, 4, 4, 13)
  // 170 + pos: the `stages = 1` form (M16X_ONE_STAGE = 32768 on top of the shipped options 5), burst + wait at position pos
  MX(64, 170, 32, 128, 8, 4, 32773) MX(64, 171, 32, 128, 8, 4, 98309) MX(64, 172, 32, 128, 8, 4, 163845) MX(64, 173, 32, 128, 8, 4, 229381)
  MX(128, 170, 32, 128, 4, 4, 32773) MX(128, 171, 32, 128, 4, 4, 98309) MX(128, 172, 32, 128, 4, 4, 163845) MX(128, 173, 32, 128, 4, 4, 229381)
  MX(64, 180, 64, 64, 4, 1, 32773) MX(64, 181, 64, 64, 4, 1, 98309) MX(64, 182, 64, 64, 4, 1, 163845) MX(64, 183, 64, 64, 4, 1, 229381)
  // 190 / 191: scores scaled in fp32 (M16X_FSCALE = 262144 on top of the shipped options 5): Q unscaled, one v_fma_f32 per score; 191 = its single-stage form
  MX(64, 190, 32, 128, 8, 4, 262149) MX(128, 190, 32, 128, 4, 4, 262149) MX(64, 192, 64, 64, 4, 1, 262149)
  // 197 / 198 / 199: the fp32-scaled form with 5 / 6 / 3 of the 8 key blocks exponentiated in phase B (the shipped acc_f32 form: 4)
  MX(64, 197, 32, 128, 8, 5, 262149) MX(64, 198, 32, 128, 8, 6, 262149) MX(64, 199, 32, 128, 8, 3, 262149)
  MX(128, 197, 32, 128, 4, 5, 262149) MX(128, 198, 32, 128, 4, 6, 262149) MX(128, 199, 32, 128, 4, 3, 262149)
  // 194..: row sums on the matrix pipe (M16X_MFMA_SUM = 524288 on top of the shipped options 5); 195: + fp32-scaled scores
  MX(64, 194, 32, 128, 8, 4, 524293) MX(128, 194, 32, 128, 4, 4, 524293) MX(64, 196, 64, 64, 4, 1, 524293) MX(64, 195, 32, 128, 8, 4, 786437) MX(128, 195, 32, 128, 4, 4, 786437)
  // 186 / 187: partial row sums by v_dot2_f32_f16 (M16X_DOT2_SUM = 1048576 on top of the shipped options 5); 187 = the 64-rows-per-wave form
  // 188 / 189: the shipped kernels with per-wave time stamps (M16X_STAMP = 2097152 on top of the options 5; cln_probe_set_stamps first)
  MX(64, 188, 32, 128, 8, 4, 2097157) MX(128, 188, 32, 128, 4, 4, 2097157) MX(64, 189, 64, 64, 4, 1, 2097157)
  MX(64, 186, 32, 128, 8, 4, 1048581) MX(128, 186, 32, 128, 4, 4, 1048581) MX(64, 187, 64, 64, 4, 1, 1048581)
#undef MX
  // 150 + id: the one-wave-per-SIMD form (flash_attn_m16s.cuh: 4 waves x 64 rows), <D, BC, PD, NDEF>
#define MS(DD, CODE, BCC, PDD, NDEFF, FINEE) \
  if (D == DD && code == CODE) return launch_m16s<DD, BCC, PDD, NDEFF, FINEE>(q, k, v, o, B, H, N, s);
  // 150 / 151: the VALU slice of a step behind ALL its MFMAs (NDEF = 1 / 2); 153 / 155: one MFMA, then its share of the slice
  // (fragment prefetch depth 4 / 8). (NDEF = 1 with the fine interleave returned wrong results on the GPU: hipcc is free to
  // copy an S^T register right behind the asm MFMA that writes it, and its hazard pass cannot see that MFMA -- the form is
  // correct only where the allocator happens not to; profiles/r03_fa_m16s_one_wave_per_simd_probe.log.)
  MS(64, 150, 64, 4, 1, false) MS(64, 151, 64, 4, 2, false) MS(64, 153, 64, 4, 2, true) MS(64, 155, 64, 8, 2, true)
  // (round 5: the same form instantiated at D = 128 -- <128, 64, 4, 2, false / true>, <128, 64, 8, 2, true> -- spills 4-8 registers, returns wrong results (its
  // hazard distances were only ever arranged for D = 64's two k-steps) and, timed as it is, runs 19-27 % BEHIND the 8-wave kernel: profiles/r05_fa_m16s_d128_probe.log,
  // r05_fa_m16s_d128_probe_timing.log. Halving the fragment reads does not pay for losing the partner wave under whose MFMAs the softmax hides. Not kept instantiated.)
#undef MS
  // 160 + id: the sum-checked two-group kernel on v_mfma_f32_32x32x16_f16 (flash_attn_m32x.cuh, D = 64, 128-key tiles): <BC, PD, OX>
  if (D == 64 && code == 160) return launch_m32x<128, 4, 1>(q, k, v, o, B, H, N, s);
  if (D == 64 && code == 161) return launch_m32x<128, 4, 0>(q, k, v, o, B, H, N, s);
  if (D == 64 && code == 162) return launch_m32x<128, 8, 1>(q, k, v, o, B, H, N, s);
  return CLN_ERR_UNSUPPORTED;
}

}  // namespace fa2

// device buffer (10 x 8 bytes per wave of the launch) the M16X_STAMP forms write into; NULL switches the stores off
CLN_API int cln_probe_set_stamps(void* buf) {
  fa2::g_m16x_stamps = reinterpret_cast<unsigned long long*>(buf);
  return CLN_OK;
}
