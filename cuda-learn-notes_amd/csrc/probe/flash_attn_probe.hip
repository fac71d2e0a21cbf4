// Tuning / ablation hook for the v2 FlashAttention kernel (not part of the reference surface).
//   int cln_fa2_variant(D, nw, vt, opt, abl, q, k, v, o, B, H, N, stream)
#include "flash_attn.cuh"
#include "flash_attn_v3.cuh"
#include "flash_attn_bigd.cuh"
#include "flash_attn_dsplit.cuh"
#include "flash_attn_pipe.cuh"
#include "flash_attn_dwide.cuh"
#include "flash_attn_v4.cuh"
#include "flash_attn_rb.cuh"
#include "flash_attn_w4.cuh"
#include "flash_attn_dsplit2.cuh"
#include "flash_attn_m16.cuh"
#include "flash_attn_m16x_api.h"
#include "flash_attn_dring.cuh"
#include "flash_attn_dw4.cuh"
#include "flash_attn_dw4b.cuh"
#include <type_traits>

#define V3(DD, NWW, OPTT) \
  if (D == DD && nw == NWW && opt == OPTT && abl == 100 && !vt) \
    return fa2::launch_v3<DD, NWW, false, OPTT>(q, k, v, o, B, H, N, (hipStream_t)stream);
#define V2(DD, NWW, OPTT, ABLL) \
  if (D == DD && nw == NWW && opt == OPTT && abl == ABLL && !vt) \
    return fa2::launch_v2<DD, NWW, false, OPTT, ABLL>(q, k, v, o, B, H, N, (hipStream_t)stream);

CLN_API int cln_fa2_variant(int D, int nw, int vt, int opt, int abl, const void* q, const void* k, const void* v,
                            void* o, int B, int H, int N, void* stream) {
  // register-blocked kernel (flash_attn_rb.cuh): abl 400.. = option sets, 410.. = ablations
  {
    using namespace fa2;
    constexpr int B0 = RB_PIN | RB_DEFER | RB_XCD | RB_ASMMAX;
#define RB(DD, ABLN, BCC, OPTT, ABLL) \
  if (D == DD && abl == ABLN) return launch_rb<DD, BCC, OPTT, ABLL>(q, k, v, o, B, H, N, (hipStream_t)stream);
    RB(64, 400, 64, B0 | RB_PRE, 0) RB(64, 401, 64, B0 | RB_PRE | RB_ASMQK, 0) RB(64, 402, 64, B0, 0)
    RB(64, 403, 64, (B0 | RB_PRE) & ~RB_PIN, 0) RB(64, 404, 32, B0 | RB_PRE, 0) RB(64, 405, 32, B0 | RB_PRE | RB_ASMQK, 0)
    RB(64, 406, 64, B0 | RB_PRE | RB_ASMQK | RB_PD2, 0) RB(64, 407, 64, (B0 | RB_PRE | RB_ASMQK) & ~RB_DEFER, 0)
    RB(64, 408, 64, B0 | RB_ASMQK, 0) RB(64, 409, 64, (B0 | RB_PRE) & ~RB_DEFER, 0)
    RB(64, 410, 64, B0 | RB_PRE | RB_ASMQK, 1) RB(64, 411, 64, B0 | RB_PRE | RB_ASMQK, 2) RB(64, 412, 64, B0 | RB_PRE | RB_ASMQK, 3)
    RB(64, 420, 64, B0 | RB_PRE | RB_HALF, 0) RB(64, 421, 64, B0 | RB_PRE | RB_HALF | RB_ASMQK, 0)
    RB(64, 422, 64, B0 | RB_PRE | RB_HALF | RB_ASMQK | RB_PD2, 0) RB(128, 420, 32, B0 | RB_HALF, 0) RB(128, 421, 32, B0 | RB_HALF | RB_ASMQK, 0)
    RB(128, 400, 32, B0, 0) RB(128, 401, 32, B0 | RB_ASMQK, 0) RB(128, 405, 32, B0 & ~RB_PIN, 0)
    RB(128, 408, 32, B0 | RB_ASMQK | RB_PD2, 0) RB(128, 409, 32, B0 & ~RB_DEFER, 0) RB(128, 407, 32, (B0 | RB_ASMQK) & ~RB_DEFER, 0)
    RB(128, 410, 32, B0 | RB_ASMQK, 1) RB(128, 411, 32, B0 | RB_ASMQK, 2)
#undef RB
  }
  // ping-pong kernel with 64 rows per wave (flash_attn_dsplit2.cuh): abl 700.. = fragment prefetch depth
  if (D == 64 && abl == 700) return fa2::launch_dsplit2<4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 701) return fa2::launch_dsplit2<2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 702) return fa2::launch_dsplit2<8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 703) return fa2::launch_dsplit2<1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 710.. = the key-split form (256 rows per workgroup, the two groups walk one half of the KV tiles each)
  if (D == 64 && abl == 710) return fa2::launch_dsplit2<4, true>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 711) return fa2::launch_dsplit2<2, true>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // one-wave-per-SIMD kernel with a hand-placed stream (flash_attn_w4.cuh): abl 600 + schedule variant, 610.. ablations
#define FW4(DD, ABLN, VARR, ABLL) \
  if (D == DD && abl == ABLN) return fa2::launch_fa_w4<DD, VARR, ABLL>(q, k, v, o, B, H, N, (hipStream_t)stream);
  FW4(64, 600, 0, 0) FW4(64, 601, 1, 0) FW4(64, 602, 2, 0) FW4(128, 600, 0, 0) FW4(128, 601, 1, 0) FW4(128, 602, 2, 0)
  FW4(64, 632, 8, 32) FW4(64, 633, 0, 32 + 31) FW4(128, 632, 0, 32)
  FW4(64, 608, 8, 0) FW4(64, 609, 9, 0) FW4(128, 608, 8, 0) FW4(128, 609, 9, 0)
  FW4(64, 610, 0, 1) FW4(64, 611, 0, 2) FW4(64, 612, 0, 4) FW4(64, 613, 0, 8) FW4(64, 614, 0, 16) FW4(64, 615, 0, 31)
  FW4(128, 610, 0, 1) FW4(128, 611, 0, 2) FW4(128, 612, 0, 4) FW4(128, 613, 0, 8) FW4(128, 614, 0, 16) FW4(128, 615, 0, 31)
#undef FW4
  // ping-pong kernel with the VALU diet (OPT_PRE): abl 500.. 
  if (D == 64 && abl == 500) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 501) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 502) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 503) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 504) return fa2::launch_dsplit<64, 1, 4, 12 | fa2::OPT_STAGGER | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);  // no deferral
  if (D == 64 && abl == 505) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 506) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_SOLO>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 507) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_PRE | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 508) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_PD8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 509) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_PD16>(q, k, v, o, B, H, N, (hipStream_t)stream);
#define PPA(ABLN, ABLL) \
  if (D == 64 && abl == ABLN) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE, ABLL>(q, k, v, o, B, H, N, (hipStream_t)stream);
  PPA(520, 1) PPA(521, 2) PPA(522, 64) PPA(523, 8) PPA(524, 16) PPA(525, 24) PPA(526, 2 | 64) PPA(527, 1 | 2 | 64)  // energy ablations of the shipped C4 kernel
  PPA(528, 128)  // life stamps of every wave
  // 540.. = the ping-pong kernel on 16x16x32 MFMAs (flash_attn_m16.cuh): fragment prefetch depth 4 / 2 / 8;
  // 545.. = 64 query rows per wave (512-row workgroups, 64-key tiles); D = 128: 540 / 542 (32 rows, 64-key tiles)
  if (D == 64 && abl == 540) return fa2::launch_m16<64, 32, 128, 4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 541) return fa2::launch_m16<64, 32, 128, 2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 542) return fa2::launch_m16<64, 32, 128, 8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 545) return fa2::launch_m16<64, 64, 64, 4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 546) return fa2::launch_m16<64, 64, 64, 8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 540) return fa2::launch_m16<128, 32, 64, 4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 542) return fa2::launch_m16<128, 32, 64, 8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 543) return fa2::launch_m16<128, 32, 128, 4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 540) return fa2::launch_m16_pair<2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 541) return fa2::launch_m16_pair<1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 540) return fa2::launch_m16_pair<2, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 541) return fa2::launch_m16_pair<1, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 544 = scores scaled in fp32 (Q not pre-scaled)
  if (D == 256 && abl == 544) return fa2::launch_m16_pair<2, false, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 544) return fa2::launch_m16_pair<2, true, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 550 + DBG: bisecting the wrong results of 544 (1 = pad before the partial stores, 2 = drain the exchange reads, 4 = scale in a
  // VGPR, 8 = a second barrier behind the exchange reads)
  if (D == 512 && abl == 551) return fa2::launch_m16_pair<2, true, false, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 552) return fa2::launch_m16_pair<2, true, false, 2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 554) return fa2::launch_m16_pair<2, true, false, 4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 566) return fa2::launch_m16_pair<2, true, false, 16>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 614) return fa2::launch_m16_pair<2, true, false, 64>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 678) return fa2::launch_m16_pair<2, true, false, 128>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 806) return fa2::launch_m16_pair<2, true, false, 256>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 1062) return fa2::launch_m16_pair<2, true, false, 512>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 1574) return fa2::launch_m16_pair<2, true, false, 1024>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 2598) return fa2::launch_m16_pair<2, true, false, 2048>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 4646) return fa2::launch_m16_pair<2, true, false, 4096>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 1100) return fa2::launch_m16_pair<4, true, false, 0>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 1101) return fa2::launch_m16_pair<1, true, false, 0>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 8742) return fa2::launch_m16_pair<2, true, false, 8192>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 8743) return fa2::launch_m16_pair<2, true, false, 8192 + 2048>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 8744) return fa2::launch_m16_pair<2, true, false, 8192 + 4096>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 8745) return fa2::launch_m16_pair<1, true, false, 8192>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 8742) return fa2::launch_m16_pair<2, false, false, 8192>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 16928) return fa2::launch_m16_pair<2, true, false, 16384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 16929) return fa2::launch_m16_pair<2, true, false, 16384 + 2048>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 16930) return fa2::launch_m16_pair<2, true, false, 16384 + 4096>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 16931) return fa2::launch_m16_pair<1, true, false, 16384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 16932) return fa2::launch_m16_pair<2, true, false, 16384 + 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 16928) return fa2::launch_m16_pair<2, false, false, 16384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 5440 = the ROUND-2 code of 544 / of the D = 256 production kernel (no cln_mfma_keep: MFMA destinations on operand registers)
  if (D == 512 && abl == 5440) return fa2::launch_m16_pair<2, true, false, 32768>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 5440) return fa2::launch_m16_pair<2, false, false, 32768>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 545) return fa2::launch_m16_pair<2, true, false, 65536>(q, k, v, o, B, H, N, (hipStream_t)stream);   // phase-B priority
  if (D == 256 && abl == 545) return fa2::launch_m16_pair<2, false, false, 65536>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 1010..1014: ablations of the D = 1024 ring kernel for the LDS counters (garbage results): no K fragment reads, no V reads, one of the
  // four partial-S reads, no partial-S writes, all four
  if (D == 1024 && abl == 1010) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_K, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 1011) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_V, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 1012) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_XR, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 1013) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_XW, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 1014) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_K | fa2::OPT_ABL_V | fa2::OPT_ABL_XR | fa2::OPT_ABL_XW, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 1020 + 10 * KPF + VPF (two digits each < 10): fragment prefetch depths of the ring kernel, production stagger / priority per head dim
  // 1200 + ...: the same with the DMA pieces of a tile request spread over the phase (OPT_SPREAD)
#define DR_SP(DD, ST, PR, KP, VP) \
  if (D == DD && abl == 1200 + 10 * KP + VP) return fa2::launch_dring<DD, fa2::OPT_DEFAULT | fa2::OPT_SPREAD, ST, PR, KP, VP>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DR_SP(640, false, 0, 2, 2) DR_SP(640, false, 0, 1, 1) DR_SP(768, false, 0, 2, 2) DR_SP(768, false, 0, 1, 1) DR_SP(1024, true, 1, 1, 1) DR_SP(1024, true, 1, 2, 1)
#undef DR_SP
#define DR_PF(DD, ST, PR, KP, VP) \
  if (D == DD && abl == 1100 + 10 * KP + VP) return fa2::launch_dring<DD, fa2::OPT_DEFAULT, ST, PR, KP, VP>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DR_PF(640, false, 0, 1, 1) DR_PF(640, false, 0, 3, 3) DR_PF(640, false, 0, 5, 5) DR_PF(640, false, 0, 5, 3)
  DR_PF(768, false, 0, 1, 1) DR_PF(768, false, 0, 3, 3) DR_PF(768, false, 0, 4, 4) DR_PF(768, false, 0, 6, 4)
  DR_PF(1024, true, 1, 2, 1)  // (deeper forms spill at D = 1024: 4-19 registers, -42 ... -49 %)
#undef DR_PF
  if (D == 1024 && abl == 1015) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_DMA, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);  // no K / V DMA after the prologue
  if (D == 1024 && abl == 1016) return fa2::launch_dring<1024, fa2::OPT_DEFAULT | fa2::OPT_ABL_DMA | fa2::OPT_ABL_K | fa2::OPT_ABL_V | fa2::OPT_ABL_XR | fa2::OPT_ABL_XW, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);  // MFMAs, softmax and barriers only
  if (D == 1024 && abl == 1002) return fa2::launch_dring<1024, fa2::OPT_DEFAULT, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 768 && abl == 1002) return fa2::launch_dring<768, fa2::OPT_DEFAULT, true, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 768 && abl == 1003) return fa2::launch_dring<768, fa2::OPT_DEFAULT, false, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 558) return fa2::launch_m16_pair<2, true, false, 8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 90 = the 4-wave load-then-compute kernel of rounds 1-3 (flash_attn.cuh, PREFETCH = false): what `stages = 1` ran for D <= 256 until
  // round 4 made it the single-stage form of the stage-2 kernels (0.24-0.44x of stages = 2, profiles/r03_fa_stage1_vs_stage2.log)
  if (abl == 90 && !vt && N % 128 == 0) {
    if (D == 64) return fa::launch_fa2<64, 64, 64, false, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
    if (D == 128) return fa::launch_fa2<128, 128, 64, false, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
    if (D == 256) return fa::launch_fa2<256, 256, 64, false, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  }
  // 262144-form of the pair kernel (stages = 1, one burst per tile): 545 at D = 256 / 512; 546 = round 3's wait after every piece (D = 512)
  if (D == 256 && abl == 545) return fa2::launch_m16_pair<2, false, false, 262144>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 545) return fa2::launch_m16_pair<2, true, false, 262144>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 547) return fa2::launch_m16_pair<2, false, false, 262144 | 524288>(q, k, v, o, B, H, N, (hipStream_t)stream);  // burst at the top of phase B
  if (D == 512 && abl == 547) return fa2::launch_m16_pair<2, true, false, 262144 | 524288>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 546) return fa2::launch_m16_pair<2, true, false, 131072>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 800.. = the sum-checked optimistic softmax form (flash_attn_m16x.cuh, its own compile unit): abl = 800 + code,
  //         code = 16 * (NDEF - 1) + OX (OX: 1 = phase-A priority, 4 = split prologue); 860.. = prefetch depth 4; 880.. = 64 rows per wave
  if (abl >= 800 && abl < 1000 && D <= 128) return fa2::m16x_probe_run(D, abl - 800, q, k, v, o, B, H, N, (hipStream_t)stream);
  // 1300 + opt = the one-wave-per-SIMD kernel for head dims 640 / 768 / 1024 (flash_attn_dw4.cuh, round 5); opt bits: 1 = `stages = 1` form, 2 = running
  // maximum raised on every growth (drives the AGPR rescale path on every tile), 4 = no K / V DMA after the prologue, 8 = no softmax (4, 8: garbage results),
  // 16 = last MFMA group of a phase carried across the barrier, 32 = M0-walking tile requests, 64 = softmax in four sections (112 = production);
  // 1500 + 10 * KPF + VPF = K / V fragments in flight (on the production options)
#define DW4(DD, OPTV) if (D == DD && abl == 1300 + OPTV) return fa2::launch_dw4<DD, OPTV>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DW4(640, 0) DW4(768, 0) DW4(1024, 0) DW4(640, 1) DW4(768, 1) DW4(1024, 1) DW4(640, 2) DW4(768, 2) DW4(1024, 2)
  DW4(1024, 4) DW4(1024, 116) DW4(768, 4) DW4(768, 116)
  DW4(640, 16) DW4(640, 17) DW4(768, 16) DW4(1024, 16) DW4(640, 32) DW4(768, 32) DW4(1024, 32) DW4(640, 64) DW4(768, 64) DW4(1024, 64)
  DW4(640, 48) DW4(768, 48) DW4(1024, 48) DW4(640, 112) DW4(768, 112) DW4(1024, 112) DW4(640, 113) DW4(768, 113) DW4(1024, 113)
  DW4(640, 114) DW4(768, 114) DW4(1024, 114) DW4(640, 96) DW4(768, 96) DW4(1024, 96)
  // + 128 = two tiles per loop iteration (compile-time ring-slot parity)
  DW4(640, 144) DW4(640, 145) DW4(640, 240) DW4(768, 240) DW4(768, 241) DW4(1024, 240) DW4(1024, 241) DW4(768, 176) DW4(1024, 176) DW4(640, 128) DW4(768, 128) DW4(1024, 128)
  // + 256 / 512 = wave w idles 16 w / 32 w clocks after every loop barrier (DW4_SKEW4 / DW4_SKEW8) on the production options
  DW4(640, 400) DW4(640, 656) DW4(768, 496) DW4(768, 752) DW4(1024, 496) DW4(1024, 752)
  // + 1024 = requests issued, never waited for (garbage results); + 4 = not issued at all -- on the production options
  DW4(640, 1168) DW4(768, 1264) DW4(1024, 1264) DW4(640, 148) DW4(768, 244) DW4(1024, 244)
  // + 1024 = one loop copy per wave index, the LDS-DMA request of a group behind MFMA w of the group (DW4_STAG)
  // D = 512 (config C5) on the same kernel with 128 rows per workgroup (GeoDW4<512, 2>)
  DW4(512, 0) DW4(512, 1) DW4(512, 2) DW4(512, 16) DW4(512, 48) DW4(512, 112) DW4(512, 113) DW4(512, 114) DW4(512, 4) DW4(512, 116)
#undef DW4
  // 1600 + opt = the one-barrier-per-tile form for D = 640 / 768 (flash_attn_dw4b.cuh); opt bits 1 / 2 / 4 / 8 as above
#define DW4B(DD, OPTV) if (D == DD && abl == 1600 + OPTV) return fa2::launch_dw4b<DD, OPTV>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DW4B(640, 0) DW4B(768, 0) DW4B(640, 1) DW4B(768, 1) DW4B(640, 2) DW4B(768, 2) DW4B(640, 4) DW4B(768, 4)
#undef DW4B
#define DW4BP(DD, KP, VP) if (D == DD && abl == 1700 + 10 * KP + VP) return fa2::launch_dw4b<DD, 0, KP, VP>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DW4BP(640, 3, 3) DW4BP(768, 3, 3) DW4BP(640, 1, 1) DW4BP(768, 1, 1) DW4BP(768, 4, 2)
#undef DW4BP
#define DW4P(DD, KP, VP) if (D == DD && abl == 1500 + 10 * KP + VP) return fa2::launch_dw4<DD, fa2::DW4_DEFAULT, KP, VP>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DW4P(1024, 1, 1) DW4P(1024, 3, 3) DW4P(1024, 4, 2) DW4P(768, 1, 1) DW4P(768, 3, 3) DW4P(640, 3, 3) DW4P(640, 1, 1) DW4P(512, 1, 1) DW4P(512, 4, 4) DW4P(512, 4, 2)
#undef DW4P
  // 2100 + 10 * KPF + VPF: the same sweep on the two-tiles-per-iteration form (the production options of each head dim)
#define DW4PU(DD, OO, KP, VP) if (D == DD && abl == 2100 + 10 * KP + VP) return fa2::launch_dw4<DD, OO, KP, VP>(q, k, v, o, B, H, N, (hipStream_t)stream);
  DW4PU(640, 144, 3, 3) DW4PU(640, 144, 4, 4) DW4PU(640, 144, 3, 2) DW4PU(768, 240, 3, 3) DW4PU(768, 240, 4, 4) DW4PU(768, 240, 3, 2) DW4PU(768, 240, 2, 3)
#undef DW4PU
  // 1000 = the ring kernel for head dims 640 / 768 / 1024 (flash_attn_dring.cuh), row groups one phase apart; 1001 = lock-step
  if (D == 640 && abl == 1000) return fa2::launch_dring<640, fa2::OPT_DEFAULT>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 768 && abl == 1000) return fa2::launch_dring<768, fa2::OPT_DEFAULT>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 1000) return fa2::launch_dring<1024, fa2::OPT_DEFAULT>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 640 && abl == 1001) return fa2::launch_dring<640, fa2::OPT_DEFAULT, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 768 && abl == 1001) return fa2::launch_dring<768, fa2::OPT_DEFAULT, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 1001) return fa2::launch_dring<1024, fa2::OPT_DEFAULT, false>(q, k, v, o, B, H, N, (hipStream_t)stream);
  // 530.. = row sums on the matrix pipe (OPT_SUMM)
  if (D == 64 && abl == 530) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_SUMM>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 531) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_SUMM>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 530) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_PRE | fa2::OPT_SUMM>(q, k, v, o, B, H, N, (hipStream_t)stream);
#undef PPA
  if (D == 64 && abl == 510) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_PRE | fa2::OPT_PD16>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 511) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_STAGGER | fa2::OPT_PRE | fa2::OPT_PD8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 508) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_PRE | fa2::OPT_PD8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 509) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_PRE | fa2::OPT_PD16>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 505) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE | fa2::OPT_PRE | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 500) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 501) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE | fa2::OPT_PRE | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 504) return fa2::launch_dsplit<128, 1, 2, 14 | fa2::OPT_KPRE | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 500) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 504) return fa2::launch_dsplit<256, 1, 1, 14 | fa2::OPT_KPRE | fa2::OPT_PRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  V2(64, 8, 16397, 0) V2(64, 4, 16397, 0) V2(64, 2, 16397, 0) V2(128, 8, 16399, 0) V2(128, 4, 16399, 0) V2(32, 8, 16397, 0) V2(96, 4, 16399, 0)
  // small-grid sweep (tools/fa_small_grid_probe.py): every wave count of the product's v2 instantiations
  V2(128, 2, 16399, 0) V2(32, 4, 16397, 0) V2(32, 2, 16397, 0) V2(96, 8, 16399, 0) V2(96, 2, 16399, 0) V2(256, 4, 15, 0)
  V2(64, 8, 16396, 0) V2(128, 8, 16398, 0)
  V2(64, 8, 13, 0) V2(64, 4, 13, 0) V2(64, 4, 77, 0) V2(128, 8, 15, 0) V2(128, 4, 15, 0) V2(128, 4, 79, 0)
  V2(64, 8, 13, 1) V2(64, 8, 13, 2) V2(64, 8, 13, 7) V2(128, 8, 15, 1) V2(128, 8, 15, 7)
  V2(64, 8, 525, 0) V2(128, 8, 527, 0) V2(64, 8, 524, 0)
  if (D == 512 && abl == 200) return fa2::launch_bigd<512, 512, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 201) return fa2::launch_bigd<512, 256, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 203) return fa2::launch_bigd<512, 256, 15 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 204) return fa2::launch_bigd<512, 256, 15 | fa2::OPT_KPRE | fa2::OPT_VPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 205) return fa2::launch_bigd<512, 256, 15 | fa2::OPT_KPRE | fa2::OPT_VPRE | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 206) return fa2::launch_bigd<512, 256, 15 | fa2::OPT_VPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 210) return fa2::launch_dsplit<512, 2, 1, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 220) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 242) return fa2::launch_dsplit<512, 2, 1, 15, 32>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 243) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_KPRE, 32>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 223) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_SOLO>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 223) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE | fa2::OPT_SOLO>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 210) return fa2::launch_dsplit<128, 1, 1, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 220) return fa2::launch_dsplit<128, 1, 1, 15 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 230) return fa2::launch_dsplit<128, 1, 2, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 231) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 230) return fa2::launch_dsplit<64, 1, 2, 13>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 231) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 232) return fa2::launch_dsplit<64, 1, 4, 13>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 233) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 240) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_KPRE, 32>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 240) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE, 32>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 240) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE, 32>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 250) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 251) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_STAGGER | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 252) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 253) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 250) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 251) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_STAGGER | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 250) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 251) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_STAGGER | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 262) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 268) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 276) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 16>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 284) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 24>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 286) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 26>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 261) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 292) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER, 32>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 270) return fa2::launch_pipe<64, 2, 13>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 271) return fa2::launch_pipe<64, 4, 13>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 270) return fa2::launch_pipe<128, 2, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 270) return fa2::launch_pipe<256, 1, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 280) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 281) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 282) return fa2::launch_dsplit<64, 1, 2, 13 | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 280) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 281) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE | fa2::OPT_STAGGER | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 280) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 280) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_ONES>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 768 && abl == 210) return fa2::launch_dwide<768, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 210) return fa2::launch_dwide<1024, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 364) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE, 64>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 364) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE, 64>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 365) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE, 65>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 365) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE, 65>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 366) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE, 66>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 366) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE, 66>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 302) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE, 2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 302) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE, 2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 301) return fa2::launch_dsplit<128, 1, 2, 15 | fa2::OPT_KPRE, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 301) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 384 && abl == 210) return fa2::launch_dsplit<512, 2, 1, 15, 0, 384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 320 && abl == 210) return fa2::launch_dsplit<512, 2, 1, 15, 0, 320>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 384 && abl == 220) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_KPRE, 0, 384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 320 && abl == 220) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_KPRE, 0, 320>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 384 && abl == 222) return fa2::launch_dsplit<512, 2, 1, fa2::OPT_DEFAULT | fa2::OPT_PD8, 0, 384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 320 && abl == 222) return fa2::launch_dsplit<512, 2, 1, fa2::OPT_DEFAULT | fa2::OPT_PD8, 0, 320>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 384 && abl == 223) return fa2::launch_dsplit<512, 2, 1, fa2::OPT_DEFAULT | fa2::OPT_PD16, 0, 384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 320 && abl == 223) return fa2::launch_dsplit<512, 2, 1, fa2::OPT_DEFAULT | fa2::OPT_PD16, 0, 320>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 384 && abl == 221) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_PD8, 0, 384>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 320 && abl == 221) return fa2::launch_dsplit<512, 2, 1, 15 | fa2::OPT_PD8, 0, 320>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 640 && abl == 210) return fa2::launch_dwide<768, 15, true>(q, k, v, o, B, H, N, (hipStream_t)stream, D);
  if (D == 512 && abl == 211) return fa2::launch_dsplit<512, 2, 1, 15, 1>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 212) return fa2::launch_dsplit<512, 2, 1, 15, 2>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 214) return fa2::launch_dsplit<512, 2, 1, 15, 4>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 218) return fa2::launch_dsplit<512, 2, 1, 15, 8>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 226) return fa2::launch_dsplit<512, 2, 1, 15, 16>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 234) return fa2::launch_dsplit<512, 2, 1, 15, 24>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 241) return fa2::launch_dsplit<512, 2, 1, 15, 31>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 210) return fa2::launch_dsplit<256, 1, 1, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 220) return fa2::launch_dsplit<256, 1, 1, 15 | fa2::OPT_KPRE>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 512 && abl == 202) return fa2::launch_bigd<512, 128, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 384 && abl == 201) return fa2::launch_bigd<384, 128, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 1024 && abl == 201) return fa2::launch_bigd<1024, 256, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 768 && abl == 201) return fa2::launch_bigd<768, 256, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 256 && abl == 200) return fa2::launch_bigd<256, 256, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  V2(64, 8, 77, 0) V2(128, 8, 79, 0) V2(64, 8, 4109, 0) V2(64, 8, 8205, 0) V2(128, 8, 4111, 0) V2(128, 8, 8207, 0) V2(64, 8, 1037, 0) V2(64, 8, 3085, 0) V2(64, 8, 2061, 0) V2(128, 8, 1039, 0) V2(128, 8, 3087, 0) V2(128, 8, 2063, 0)
  if (D == 64 && abl == 300 && opt == 13) return fa2::launch_v4<64, false, 13>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 64 && abl == 300 && opt == 15) return fa2::launch_v4<64, false, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 300 && opt == 13) return fa2::launch_v4<128, false, 13>(q, k, v, o, B, H, N, (hipStream_t)stream);
  if (D == 128 && abl == 300 && opt == 15) return fa2::launch_v4<128, false, 15>(q, k, v, o, B, H, N, (hipStream_t)stream);
  V3(64, 8, 16397) V3(64, 4, 16397) V3(64, 8, 13) V3(64, 8, 269) V3(64, 8, 15) V3(64, 4, 13) V3(128, 8, 15) V3(128, 8, 13) V3(128, 8, 271) V3(128, 4, 15)
  return CLN_ERR_UNSUPPORTED;
}
