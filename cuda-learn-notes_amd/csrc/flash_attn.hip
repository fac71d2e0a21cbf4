// C-ABI entry points of the FlashAttention-2 forward library: one symbol per function exported by the
// reference's pybind module (kernels/flash-attn/pybind/flash_attn.cc:182-215).
//
//   int name(q, k, v, o, B, H, N, D, stages, stream)
//
// q,k,v,o: fp16 [B,H,N,D] contiguous; the *_swizzle_qkv variants of share_kv / share_qkv /
// tiling_qk take v TRANSPOSED, [B,H,D,N] (reference flash_attn_mma.py:377-378, :542-565).
// stages (reference kStage template parameter, flash_attn_mma_share_qkv.cu:843-884):
//   1 = a tile is requested, waited for, then used: the stage-2 kernel of the shape with each tile's requests issued in one burst and
//       waited for right there (no request of a wave in flight while it computes); bit-identical to stages = 2,
//   2 = the next K/V tile is prefetched under the MFMA phases of the current one.
// Every name goes through ONE planner (fa2_plan) that picks the gfx950 kernel for (family, shape, stages); the same
// plan is what cln_describe() prints, so the name -> kernel map in manifest.py is checked against the code
// (tests/test_describe.py).
#include "flash_attn_large_d.cuh"
#include "flash_attn_splitkv.cuh"
#include "flash_attn_v2.cuh"
#include "flash_attn_m16.cuh"
#include "flash_attn_pair2.cuh"
#include "flash_attn_m16x_api.h"
#include <string.h>

namespace {

enum FaFamily { FAM_SPLIT_KV = 0, FAM_SPLIT_Q = 1 };
enum FaKind { K_NONE = 0, K_SPLITKV, K_V2, K_DSPLIT, K_DW4, K_M16X64R, K_M16 };

struct FaPlan {
  int rc = CLN_OK;     // CLN_ERR_* when the shape is not supported
  int kind = K_NONE;
  int d_inst = 0;      // head dim of the instantiation (> D: padded form)
  int nw = 0;          // waves per workgroup
  int bc = 0;          // keys per KV tile
  bool one_stage = false;       // stages = 1: the stage-2 kernel of the shape with every tile fetch waited for where it is issued
  bool f32_scale = false;       // the *_acc_f32 names at D <= 128: scores scaled in fp32 (Q as loaded) instead of the fp16 pre-scaled Q
};

FaPlan fa2_plan(int family, bool vt, int B, int H, int N, int D, int stages, int max_d, bool acc32 = false) {
  FaPlan p;
  // The reference's *_acc_f32 names accumulate both GEMMs in fp32 (flash_attn_mma_share_qkv_F32F16F16F32.cu:66) where the plain names accumulate
  // in fp16: the precision rung of the ladder. Here every kernel accumulates in fp32; what the D <= 128 kernels round is Q * log2(e)/sqrt(d),
  // once, to fp16 (2^-11 relative per score term: 4e-3 on O under keys amplified 4-6x, 5e-4 on N(0,1) inputs). The *_acc_f32 names run the
  // same kernels with the scores scaled in fp32 instead (one v_fma_f32 per score: 2e-3 / 2e-4), 4-10 % slower (profiles/r04_fa_fscale_probe.log).
  p.f32_scale = acc32 && !vt && family == FAM_SPLIT_Q && D <= 128;
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0) return p.rc = CLN_ERR_BAD_ARG, p;
  if ((long long)B * H * (long long)(N / 32 + 1) > 0x7fffffffLL) return p.rc = CLN_ERR_UNSUPPORTED, p;  // grid size (x)
  if (D > max_d) return p.rc = CLN_ERR_UNSUPPORTED, p;  // "headdim not support!"
  const long long bh = (long long)B * H;
  if (family == FAM_SPLIT_KV) {
    // the split-KV rung: its own kernel (flash_attn_splitkv.cuh); stages = 1 loads a tile and uses it, stages >= 2 keeps the next
    // tile's K fragments in flight in a second set of registers (reference kStage of flash_attn_mma_split_kv.cu)
    if ((D != 32 && D != 64 && D != 96 && D != 128) || N % 32 != 0) return p.rc = CLN_ERR_UNSUPPORTED, p;
    if (bh > 65535) return p.rc = CLN_ERR_UNSUPPORTED, p;  // this kernel carries B*H in gridDim.y
    p.kind = K_SPLITKV, p.d_inst = D, p.nw = 4, p.bc = 128, p.one_stage = stages == 1;
    return p;
  }
  const bool small_d = D == 32 || D == 64 || D == 96 || D == 128 || D == 256;
  // ---- stages = 1 (reference kStage = 1, flash_attn_mma_share_qkv.cu:711-762: a tile is requested, waited for, then used): at EVERY head
  // dim the stage-2 kernel of the shape in its single-stage form -- each tile requested in one burst and waited for where it is requested,
  // no request of a wave in flight while it computes; same LDS image, same arithmetic, bit-identical output (`one_stage`). Rounds 1-3 ran
  // a separate 4-wave load-then-compute kernel (probe/flash_attn.cuh) for D <= 256: 0.24-0.44x of stages = 2 (profiles/r03_fa_stage1_vs_stage2.log);
  // that kernel now lives in the probe library only.
  p.one_stage = stages == 1;
  if (small_d) {
    if (D == 64 && N % 512 == 0) {  // (both V layouts)
      // >= 512 query rows per CU, in (nearly) whole rounds of 256 workgroups: 64 query rows per wave -- every K / V fragment
      // feeds four 16x16x32 MFMAs (flash_attn_m16x.cuh with RPW = 64, 64-key tiles; round 2 ran the 32x32x16 form of
      // probe/flash_attn_dsplit2.cuh here): [1,48,8192,64] 1063 -> 1138 TF, [2,32,4096,64] 1032 -> 1094, [1,16,16384,64] 1076 -> 1152
      // (profiles/r03_fa_m16x_64rows_probe.log)
      const long long wgs = bh * (N / 512), rounds = (wgs + 255) / 256;
      if (wgs >= 256 && wgs * 100 >= rounds * 256 * 88) return p.kind = K_M16X64R, p.d_inst = 64, p.nw = 8, p.bc = 64, p;
    }
    // 256-row workgroups of the two-group kernels against 128-row (4-wave) workgroups of the v2 kernel, by the number w of 256-row
    // blocks (profiles/r04_fa_small_grid_probe.log; rounds 1-3 switched at w >= 192 for every head dim and fell to 2-wave workgroups below 128):
    //  D = 128: the two-group kernel at every w (w = 32 ... 128: +4-12 % over 4-wave v2, 2x over the 2-wave form the old rule picked)
    //  D = 64:  v2 with 4 waves while its 2w workgroups each get a CU of their own (w <= 128: 1.2x the two-group kernel),
    //           the two-group kernel above (w = 160: +14-19 % over v2)
    //  D = 256: (non-transposed V only) the two-group kernel at every w (w = 32 ... 160: 1.7-3.7x the 4-wave v2 kernel, which holds the
    //           whole register file and one wave per SIMD)
    const long long w256 = N % 256 == 0 ? bh * (N / 256) : 0;
    const long long w_min = D == 64 ? 129 : 1;
    // (D = 64 with N = 256: one 256-row block per head and two 128-key tiles -- the two-group pipeline never reaches its steady state;
    //  4-wave v2 leads by 5-19 % at every B H measured, profiles/r04_fa_short_n_probe.log)
    if ((!vt || D == 64 || D == 128) && w256 >= w_min && !(D == 64 && N == 256)) {
      //  D = 64 / 128: ping-pong kernel on 16x16x32 MFMAs (the energy-cheaper matrix shape, +3.5-5 % at D = 64 and
      //                +5.5-6.5 % at D = 128 over the 32x32x16 form of flash_attn_dsplit.cuh, profiles/r02_fa_m16_probe.log)
      //                with the sum-checked optimistic softmax, phase-A priority and the split prologue of round 3
      //                (flash_attn_m16x.cuh: +1-2.5 % over flash_attn_m16.cuh, profiles/r03_fa_m16x_probe.log). (The one-wave-per-SIMD kernel probe/flash_attn_w4.cuh measured parity at
      //                best and lives in the probe library only.)
      //  D = 256: two-group ping-pong kernel, 8 waves x 32 rows
      if (D == 64) return p.kind = K_M16, p.d_inst = 64, p.nw = 8, p.bc = 128, p;
      if (D == 128) return p.kind = K_M16, p.d_inst = 128, p.nw = 8, p.bc = 128, p;
      // D = 256: the same 16x16x32 layout, one wave per 32 rows holding the whole d, scores scaled in fp32 (max-abs-error
      // identical to the 32x32x16 kernel it replaces): [4,8,2048,256] 1057 -> 1125 TF, [2,32,4096,256] 1132 -> 1183
      // (profiles/r02_fa_m16_d256_probe.log, variant 544)
      if (D == 256) return p.kind = K_M16, p.d_inst = 256, p.nw = 8, p.bc = 32, p;
    }
    if (D == 256) {  // needs the whole register file (one wave per SIMD): 4 waves x 32 rows only
      if (N % 128 != 0) return p.rc = CLN_ERR_UNSUPPORTED, p;
      return p.kind = K_V2, p.d_inst = 256, p.nw = 4, p.bc = 64, p;
    }
    // v2 kernel, 32 query rows per wave: 4 waves whenever N allows (2-wave workgroups stage every K / V tile for half the rows:
    // 0.5-0.75x at every grid size measured); 8 waves (D = 32 / 96 only -- at D = 64 / 128 those shapes run the two-group kernel)
    // once the 8-wave workgroups alone cover more than half the CUs at D = 96 (w > 128: +0-6 % over 4 waves; at w <= 128 every
    // 4-wave workgroup has a CU of its own and 4 waves win by 1.15x), past a full round of them at D = 32 (4 waves lead by 2-6 % up to w = 256)
    int nw = 0;
    for (int cand : {8, 4, 2}) {
      if (N % (cand * 32) != 0) continue;
      if (cand == 8 && (D == 64 || D == 128 || w256 <= (D == 32 ? 256 : 128))) continue;
      nw = cand;
      break;
    }
    if (nw == 0) return p.rc = CLN_ERR_UNSUPPORTED, p;
    return p.kind = K_V2, p.d_inst = D, p.nw = nw, p.bc = 64, p;
  }
  if (vt) return p.rc = CLN_ERR_UNSUPPORTED, p;
  // ---- head dims above 256 ("fine-grained tiling" rungs, flash_attn_large_d.cuh). The reference's tiling kernels template
  // on kStage 1 / 2 (flash_attn_mma_tiling_qkv.cu:63, :189-223: with kStage = 1 a tile is loaded, waited for, then used).
  // stages = 1 here: the SAME d-split / ring kernels with every tile fetch waited for where it is issued, so no load runs
  // under compute (`one_stage`); stages = 2: the pipelines. (Round 3's first form ran the 4-wave kernel of probe/flash_attn.cuh
  // with the output head dim sliced and S recomputed per slice: 94 TF at D = 768 / 1024, profiles/r03_fa_stage1_vs_stage2.log.)
  switch (D) {
    case 512:  // config C5: the d-split PAIR kernel on 16x16x32 MFMAs, scores scaled in fp32 (flash_attn_m16.cuh, round 3: +2.7 %
      // over the 32x32x16 form at identical max-abs-error once its MFMA destinations were kept off the operand registers)
      if (N % 128 != 0) return p.rc = CLN_ERR_UNSUPPORTED, p;
      return p.kind = K_M16, p.d_inst = 512, p.nw = 8, p.bc = 32, p;
    case 320: case 384:  // round 6: the pair2 kernel on the D = 512 LDS geometry, every loop over the real head dim (rounds 2-5: the d-split kernel on that geometry)
      if (N % 128 != 0) return p.rc = CLN_ERR_UNSUPPORTED, p;
      return p.kind = K_M16, p.d_inst = 512, p.nw = 8, p.bc = 32, p;
    case 640: case 768: case 1024:  // round 5: one wave per SIMD, 64 rows per workgroup (flash_attn_dw4.cuh)
      if (N % 64 != 0) return p.rc = CLN_ERR_UNSUPPORTED, p;
      return p.kind = K_DW4, p.d_inst = D, p.nw = 4, p.bc = 16, p;
    default: return p.rc = CLN_ERR_UNSUPPORTED, p;
  }
}

template <bool VT>
int fa2_run(const FaPlan& p, const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D,
            hipStream_t s) {
  switch (p.kind) {
    case K_SPLITKV:
      if constexpr (!VT) {
        switch (D) {
          case 32: return p.one_stage ? fa2::launch_splitkv<32, false>(q, k, v, o, B, H, N, s) : fa2::launch_splitkv<32, true>(q, k, v, o, B, H, N, s);
          case 64: return p.one_stage ? fa2::launch_splitkv<64, false>(q, k, v, o, B, H, N, s) : fa2::launch_splitkv<64, true>(q, k, v, o, B, H, N, s);
          case 96: return p.one_stage ? fa2::launch_splitkv<96, false>(q, k, v, o, B, H, N, s) : fa2::launch_splitkv<96, true>(q, k, v, o, B, H, N, s);
          case 128: return p.one_stage ? fa2::launch_splitkv<128, false>(q, k, v, o, B, H, N, s) : fa2::launch_splitkv<128, true>(q, k, v, o, B, H, N, s);
        }
      }
      return CLN_ERR_UNSUPPORTED;
    case K_V2:
#define FA_V2_NW(DD, OPTT, HAS8)                                                                       \
    if (p.nw == 8) {                                                                                   \
      if constexpr (HAS8) return fa2::launch_v2<DD, 8, VT, OPTT>(q, k, v, o, B, H, N, s);             \
      else return CLN_ERR_UNSUPPORTED; /* the plan never names it: these shapes run fa2_fwd_m16x */   \
    }                                                                                                  \
    if (p.nw == 4) return fa2::launch_v2<DD, 4, VT, OPTT>(q, k, v, o, B, H, N, s);                     \
    return fa2::launch_v2<DD, 2, VT, OPTT>(q, k, v, o, B, H, N, s);
#define FA_V2(DD, OPTT, HAS8)                                                      \
  case DD:                                                                         \
    if constexpr (!VT) {                                                           \
      if (p.f32_scale && p.one_stage) { FA_V2_NW(DD, ((OPTT) & ~fa2::OPT_PRE) | fa2::OPT_1STAGE, HAS8) } \
      if (p.f32_scale) { FA_V2_NW(DD, (OPTT) & ~fa2::OPT_PRE, HAS8) }              \
    }                                                                              \
    if (p.one_stage) { FA_V2_NW(DD, (OPTT) | fa2::OPT_1STAGE, HAS8) }              \
    FA_V2_NW(DD, OPTT, HAS8)
      switch (D) {
        FA_V2(32, 13 | fa2::OPT_PRE, true)
        FA_V2(64, 13 | fa2::OPT_PRE, false)
        FA_V2(96, 15 | fa2::OPT_PRE, true)
        FA_V2(128, 15 | fa2::OPT_PRE, false)
        case 256: return p.one_stage ? fa2::launch_v2<256, 4, VT, 15 | fa2::OPT_1STAGE>(q, k, v, o, B, H, N, s)
                                     : fa2::launch_v2<256, 4, VT, 15>(q, k, v, o, B, H, N, s);
      }
#undef FA_V2
#undef FA_V2_NW
      return CLN_ERR_UNSUPPORTED;
    case K_M16X64R:
      return fa2::m16x_run(64, 64, VT, p.one_stage, p.f32_scale, q, k, v, o, B, H, N, s);
    case K_M16:
      if (D == 64 || D == 128) return fa2::m16x_run(D, 32, VT, p.one_stage, p.f32_scale, q, k, v, o, B, H, N, s);  // 128-key tiles; own compile unit
      if constexpr (!VT) {
        constexpr int ONE = 262144;  // flash_attn_m16.cuh: the tile requested in one burst at the top of phase A and waited for there
        if (D == 256) return p.one_stage ? fa2::launch_m16_pair<2, false, false, ONE>(q, k, v, o, B, H, N, s)
                                         : fa2::launch_m16_pair<2, false, false>(q, k, v, o, B, H, N, s);
        // D = 512 (config C5): pairs of waves split the ROWS for QK^T and the softmax (done once per row), d for PV (flash_attn_pair2.cuh, round 6)
        if (D == 512) return p.one_stage ? fa2::launch_pair2<4, 2, fa2::PAIR2_HOIST | fa2::PAIR2_ONE_STAGE>(q, k, v, o, B, H, N, s)
                                         : fa2::launch_pair2<4, 2, fa2::PAIR2_HOIST>(q, k, v, o, B, H, N, s);
        if (D == 384) return p.one_stage ? fa2::launch_pair2<4, 2, fa2::PAIR2_HOIST | fa2::PAIR2_ONE_STAGE, 384>(q, k, v, o, B, H, N, s)
                                         : fa2::launch_pair2<4, 2, fa2::PAIR2_HOIST, 384>(q, k, v, o, B, H, N, s);
        if (D == 320) return p.one_stage ? fa2::launch_pair2<4, 2, fa2::PAIR2_HOIST | fa2::PAIR2_ONE_STAGE, 320>(q, k, v, o, B, H, N, s)
                                         : fa2::launch_pair2<4, 2, fa2::PAIR2_HOIST, 320>(q, k, v, o, B, H, N, s);
      }
      return CLN_ERR_UNSUPPORTED;
    case K_DSPLIT:  // (no plan names it since round 6: every form of the 32x32x16 d-split kernel -- D = 64 / 128 / 256 / 512 and 320 / 384 on the 512 geometry -- is a probe variant of kind 8)
      return CLN_ERR_UNSUPPORTED;
    case K_DW4:
      if constexpr (!VT) return fa::launch_fa2_large_d(q, k, v, o, B, H, N, D, s, p.one_stage);
      return CLN_ERR_UNSUPPORTED;
    default: return CLN_ERR_UNSUPPORTED;
  }
}

template <bool VT>
int fa2_dispatch(int family, const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D,
                 int stages, int max_d, bool acc32, hipStream_t s) {
  if (!q || !k || !v || !o) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(q) || !cln_aligned16(k) || !cln_aligned16(v) || !cln_aligned16(o)) return CLN_ERR_BAD_ARG;
  const FaPlan p = fa2_plan(family, VT, B, H, N, D, stages, max_d, acc32);
  if (p.rc != CLN_OK) return p.rc;
  return fa2_run<VT>(p, q, k, v, o, B, H, N, D, s);
}

int fa2_describe(int family, bool vt, int B, int H, int N, int D, int stages, int max_d, bool acc32, char* buf, int len) {
  const FaPlan p = fa2_plan(family, vt, B, H, N, D, stages, max_d, acc32);
  const char* qs = p.f32_scale ? "fp32-scaled scores" : "pre-scaled Q";
  if (p.rc != CLN_OK) return p.rc;
  const char* st = p.one_stage ? " [single stage: every tile fetch waited for where it is issued]" : "";
  const char* vts = vt ? ",V^T" : "";
  switch (p.kind) {
    case K_SPLITKV:
      return snprintf(buf, len, "fa2_fwd_splitkv<D=%d,%s> 4 waves share 32 rows, 128-key tiles split over the waves, "
                                "cross-wave max via LDS", D, p.one_stage ? "load-then-compute" : "next K fragments prefetched into registers");
    case K_V2:
      return snprintf(buf, len, "fa2_fwd_v2<D=%d,NW=%d,BC=64,%s%s%s%s> %d waves x 32 rows%s", D, p.nw, p.one_stage ? "load-then-compute" : "prefetch",
                      D <= 128 ? "," : "", D <= 128 ? qs : "", vts, p.nw, st);
    case K_M16X64R:
      return snprintf(buf, len, "fa2_fwd_m16x64r<D=64,BC=64,16x16x32 MFMA,%s,sum-checked softmax%s> 8 waves x 64 rows, two groups one "
                                "phase apart, K/V fragments shared by 4 query blocks%s", qs, vts, st);
    case K_M16:
      if (D <= 128)
        return snprintf(buf, len, "fa2_fwd_m16x<D=%d,BC=%d,16x16x32 MFMA,%s,sum-checked softmax%s> 8 waves x 32 rows, two groups "
                                  "one phase apart%s", D, p.bc, qs, vts, st);
      if (D == 320 || D == 384)
        return snprintf(buf, len, "fa2_fwd_pair2<D=%d,BC=32,16x16x32 MFMA,LDS geometry of D=512,pairs of waves: rows split for QK^T and the softmax, d for PV> 8 waves, 128 rows, two groups one phase apart%s", D, st);
      if (D == 512)
        return snprintf(buf, len, "fa2_fwd_pair2<D=512,BC=32,16x16x32 MFMA,pairs of waves: rows split for QK^T and the softmax, d for PV> 8 waves, 128 rows, two groups one phase apart%s", st);
      return snprintf(buf, len, "fa2_fwd_m16<D=%d,BC=%d,16x16x32 MFMA> 8 waves x 32 rows, two groups one phase apart%s", D, p.bc, st);
    case K_DSPLIT:
      if (p.d_inst != D)
        return snprintf(buf, len, "fa2_fwd_dsplit<D=%d,NSP=2,BC=32,LDS geometry of D=%d> 8 waves, pairs split the real d evenly%s", D, p.d_inst, st);
      return snprintf(buf, len, "fa2_fwd_dsplit<D=%d,NSP=%d,BC=%d> 8 waves, two groups one phase apart%s", D,
                      D == 512 ? 2 : 1, p.bc, st);
    case K_DW4:
      return snprintf(buf, len, "fa2_fwd_dw4<D=%d,BC=16,2-slot K/V rings,O^T in AGPRs> 4 waves (one per SIMD) split d (%d columns each), 64 rows, "
                                "softmax once per row by its owner wave%s", D, D / 4, st);
    default: return CLN_ERR_UNSUPPORTED;
  }
}

struct FaName {
  const char* name;
  int family;
  bool vt;
  int max_d;
};

}  // namespace

// max head dim per function follows the reference driver table (flash_attn_mma.py:436-506)
#define CLN_FA_LIST(X)                                                              \
  X(flash_attn_mma_stages_split_kv, FAM_SPLIT_KV, false, 128)                      \
  X(flash_attn_mma_stages_split_q, FAM_SPLIT_Q, false, 128)                        \
  X(flash_attn_mma_stages_split_q_shared_kv, FAM_SPLIT_Q, false, 256)              \
  X(flash_attn_mma_stages_split_q_shared_qkv, FAM_SPLIT_Q, false, 256)             \
  X(flash_attn_mma_stages_split_q_tiling_qk, FAM_SPLIT_Q, false, 1024)             \
  X(flash_attn_mma_stages_split_q_tiling_qkv, FAM_SPLIT_Q, false, 1024)            \
  X(flash_attn_mma_stages_split_q_shared_kv_acc_f32, FAM_SPLIT_Q, false, 256)      \
  X(flash_attn_mma_stages_split_q_shared_qkv_acc_f32, FAM_SPLIT_Q, false, 256)     \
  X(flash_attn_mma_stages_split_q_tiling_qk_acc_f32, FAM_SPLIT_Q, false, 1024)     \
  X(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32, FAM_SPLIT_Q, false, 1024)    \
  X(flash_attn_mma_stages_split_q_shared_kv_swizzle_q, FAM_SPLIT_Q, false, 256)    \
  X(flash_attn_mma_stages_split_q_shared_kv_swizzle_qk, FAM_SPLIT_Q, false, 256)   \
  X(flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv, FAM_SPLIT_Q, true, 256)   \
  X(flash_attn_mma_stages_split_q_shared_qkv_swizzle_q, FAM_SPLIT_Q, false, 256)   \
  X(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qk, FAM_SPLIT_Q, false, 256)  \
  X(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv, FAM_SPLIT_Q, true, 256)  \
  X(flash_attn_mma_stages_split_q_tiling_qk_swizzle_q, FAM_SPLIT_Q, false, 1024)   \
  X(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qk, FAM_SPLIT_Q, false, 1024)  \
  X(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv, FAM_SPLIT_Q, true, 256)   \
  X(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_q, FAM_SPLIT_Q, false, 1024)  \
  X(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qk, FAM_SPLIT_Q, false, 1024) \
  X(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qkv, FAM_SPLIT_Q, false, 1024) \
  X(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_q, FAM_SPLIT_Q, false, 1024)   \
  X(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qk, FAM_SPLIT_Q, false, 1024)  \
  X(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qkv, FAM_SPLIT_Q, false, 1024) \
  /* BUILD_FLASH_ATTN_MMA_OTHERS set (flash_attn.cc:161-180) -- always built here */        \
  X(flash_attn_mma_stages_split_q_shared_qkv_Os2g, FAM_SPLIT_Q, false, 256)                 \
  X(flash_attn_mma_stages_split_q_shared_kv_acc_f32_rr, FAM_SPLIT_Q, false, 256)            \
  X(flash_attn_mma_stages_split_q_shared_qkv_acc_f32_rr, FAM_SPLIT_Q, false, 256)

#define CLN_FA(name, FAM, VT, MAXD)                                                                       \
  CLN_API int name(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D,     \
                   int stages, void* stream) {                                                            \
    return fa2_dispatch<VT>(FAM, q, k, v, o, B, H, N, D, stages, MAXD, strstr(#name, "_acc_f32") != nullptr, (hipStream_t)stream); \
  }
CLN_FA_LIST(CLN_FA)

// describe hook of this library group (see cln_describe in describe.hip): returns the length written, or a
// CLN_ERR_* code; CLN_ERR_BAD_ARG when `name` is not a flash-attn name.
int cln_fa_describe(const char* name, int B, int H, int N, int D, int stages, char* buf, int len) {
#define CLN_FA_ROW(n, FAM, VT, MAXD) {#n, FAM, VT, MAXD},
  static const FaName table[] = {CLN_FA_LIST(CLN_FA_ROW)};
  for (const FaName& e : table)
    if (strcmp(e.name, name) == 0) return fa2_describe(e.family, e.vt, B, H, N, D, stages, e.max_d, strstr(e.name, "_acc_f32") != nullptr, buf, len);
  return CLN_ERR_BAD_ARG;
}
