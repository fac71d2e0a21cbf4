// FlashAttention-2 forward, REGISTER-BLOCKED form for head dims 64 and 128 (BASELINE config C4 = [4,8,2048,64] and its
// D = 128 sibling). Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66 ("shared-QKV": Q lives
// in registers, K and V share one staging region).
//
// Why another kernel (measured on the ping-pong kernel of flash_attn_dsplit.cuh, profiles/r01_*):
//   * every 32x32x16 MFMA consumed one fresh 1-KiB K or V fragment from LDS (a wave owned 32 query rows): with the
//     fragment reads ablated the D = 128 kernel ran 34 % faster -- LDS read queueing, not the matrix pipe, set the pace;
//   * two waves per SIMD never overlapped their VALU (softmax) and matrix phases: time per SIMD stayed MFMA + VALU.
// This kernel changes the wave shape instead of the phase clock:
//   * a workgroup = 4 waves, ONE wave per SIMD with the whole 512-register file; a wave owns 64 query rows = two
//     32-row groups, and every K / V fragment it reads feeds TWO MFMAs (one per row group): half the LDS bytes per flop;
//   * MFMA and softmax VALU work of DIFFERENT tiles are interleaved inside one instruction stream (an in-order wave
//     hides ~5 single-issue instructions behind each 32-cycle MFMA): region R1 = QK^T of tile j+1 (accumulators
//     double-buffered) with the exponentials / conversions / row sums of tile j written between its MFMAs step by
//     step; region R2 = P V of tile j with the row-max chain of tile j+1. The interleave is explicit in the source
//     (one softmax slice per fragment step, pinned with sched_barrier) -- hipcc's own placement does not get there
//     (flash_attn_pipe.cuh, round 1);
//   * RB_PRE: Q is pre-multiplied by log2(e)/sqrt(d) once and the S^T accumulators START at -m (the running row
//     max, a per-lane constant because a lane owns one query row), so P = exp2(acc) with no per-element fma; the
//     running max is the deferred one (rescale only when a row grew by more than 2^8), and a rescale also shifts the
//     pending accumulators;
//   * K/V tiles of 64 keys in a 3-slot LDS ring filled by LDS-DMA two tiles ahead (same lane-linear images and
//     source-side XOR swizzles as the ping-pong kernel), ONE workgroup barrier per tile.
#pragma once
#include "flash_attn_bigd.cuh"
#include <type_traits>

namespace fa2 {

enum : int { RB_PRE = 1, RB_PIN = 2, RB_DEFER = 4, RB_XCD = 8, RB_ASMMAX = 16, RB_PD2 = 32, RB_ASMQK = 64, RB_HALF = 128 };
constexpr int RB_OPT_D64 = RB_PRE | RB_PIN | RB_DEFER | RB_XCD | RB_ASMMAX;
constexpr int RB_OPT_D128 = RB_PIN | RB_DEFER | RB_XCD | RB_ASMMAX;
constexpr int RB_BC_D64 = 64, RB_BC_D128 = 32;
// flipped to true once the kernel beat the ping-pong kernel on the GPU (profiles/r02_fa_rb_probe.log)
constexpr bool RB_PRODUCTION_D64 = false;
constexpr bool RB_PRODUCTION_D128 = false;

template <int D, int BC_>
struct GeoRB {
  static constexpr int BC = BC_, NW = 4, BR = 256, NT = 256, NSTG = 3;
  static constexpr int NEL = BC;  // scores per lane per tile: 2 row groups x BC / 2
  static constexpr int KB = BC / 32, NK = D / 16, NDB = D / 32, NST = BC / 16;
  static constexpr int ROW = D * 2;            // bytes per K / V row
  static constexpr int TILE = BC * ROW;        // one K or V tile
  static constexpr int STAGE = 2 * TILE;       // K + V
  static constexpr int RING = NSTG * STAGE;
  static constexpr int OS = D * 2 + 16;
  static constexpr int EPI = NW * 64 * OS;
  static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;
  static constexpr int PPW = TILE / 1024 / NW;  // DMA pieces per wave per operand tile
  static constexpr int RPP = 1024 / ROW;        // rows per 1-KiB DMA piece
  static constexpr int CPR = ROW / 16;          // 16-byte chunks per row
  static_assert((D == 64 || D == 128) && (BC == 32 || BC == 64), "register-blocked kernel: head dims 64 and 128");
  // chunk-index swizzles of the lane-linear LDS images (flash_attn_dsplit.cuh GeoSplit)
  static __device__ __forceinline__ int swz_k(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
  static __device__ __forceinline__ int swz_v(int row) { return CPR >= 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ float rb_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int D, int BC, int OPT, int ABL = 0>
__global__ __launch_bounds__(256, 1) void fa2_fwd_rb_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                            const half_t* __restrict__ V, half_t* __restrict__ O,
                                                            int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoRB<D, BC>;
  constexpr bool PRE = (OPT & RB_PRE) != 0, PIN = (OPT & RB_PIN) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  int head_i, qbi;
  {
    const int bid = blockIdx.x;
    if ((OPT & RB_XCD) && (n_heads & 7) == 0) {  // heads pinned to XCDs: a head's K/V stays in one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qbi = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qbi = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qbi * G::BR + wave * 64;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // ---- LDS-DMA: every wave fills the 1-KiB pieces i*4 + wave of the K tile and of the V tile. A piece is RPP rows;
  // lane l carries 16-byte chunk c = l % CPR of row (piece*RPP + l / CPR) to the lane-linear LDS position, reading it
  // from the source chunk (c ^ swizzle(row))  (flash_attn_dsplit.cuh).
  const char* src_k = reinterpret_cast<const char*>(K + head);
  const char* src_v = reinterpret_cast<const char*>(V + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = wave * G::RPP + lr;
  const unsigned src_lane_k = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_k(rlow)) << 4);
  const unsigned src_lane_v = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_v(rlow)) << 4);
  auto dma_piece = [&](int n, int jt, int slot) {  // n < PPW: K piece n; else V piece n - PPW
    if constexpr ((ABL & 1) != 0) return;
    const int op = n >= G::PPW, i = op ? n - G::PPW : n;
    const int piece = i * 4 + wave;
    const unsigned voff = op ? src_lane_v : (src_lane_k ^ (unsigned)(((i * 4 * G::RPP) & 15) << 4));
    const char* s = (op ? src_v : src_k) + (size_t)jt * G::TILE + piece * 1024;
    hgemm::glds16_asm(s, voff, lds0 + slot * G::STAGE + op * G::TILE + piece * 1024);
  };

  // ---- Q fragments (B operand of S^T = K Q^T): lane (row l31 of group g) holds d = 16*ks + 8*hi .. +7
  h8 qf[2][G::NK];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const half_t* qp = Q + head + (size_t)(q_row0 + g * 32 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < G::NK; ++ks) qf[g][ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }
  const int T = N / G::BC;
  __builtin_assume(T > 1);  // N % 256 == 0 (launcher)
#pragma unroll
  for (int n = 0; n < 2 * G::PPW; ++n) dma_piece(n, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: Q and tile 0 are in
  if constexpr (PRE) {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ks = 0; ks < G::NK; ++ks) qf[g][ks] = qf[g][ks] * sc;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ks = 0; ks < G::NK; ++ks) asm volatile("" : "+v"(qf[g][ks]));  // keep the fragments out of the KV loop
#pragma unroll
  for (int n = 0; n < 2 * G::PPW; ++n) dma_piece(n, T > 1 ? 1 : 0, 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // fragment offsets (lane constants; see flash_attn_dsplit.cuh for the swizzle algebra)
  const int kbase = l31 * G::ROW + ((hi ^ G::swz_k(l31)) << 4);
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = v_row * G::ROW + ((G::swz_v(v_row) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) + ((i16 & 1) << 3);

  f16v ot[2][G::NDB];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int b = 0; b < G::NDB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[g][b][r] = 0.f;
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  // With more than 256 registers in play hipcc keeps the results of its MFMA builtins in the ACCUMULATOR half of the
  // register file, which the VALU cannot read. Two ways to get the scores to the softmax code:
  //  * default: every score is moved out exactly ONCE (v_accvgpr_read fused with the row-max chain in R2) into `sv`, and
  //    the exponentials of the next R1 read `sv` (hipcc by itself emits two to three reads per score);
  //  * RB_ASMQK: the QK^T MFMAs are written in inline asm in their VGPR form (results in the VALU half, Q fragments
  //    in the accumulator half), no move at all; the score sets of tile j and j+1 then alternate between two VGPR
  //    arrays (tile loop unrolled by two). hipcc neither schedules nor pads an asm MFMA (cdna guide 5.7): the
  //    statements carry their own s_nop, and the first VALU read of a finished set sits behind rb_fence().
  constexpr bool ASMQK = (OPT & RB_ASMQK) != 0;
  f16v s_acc[2][2][G::KB];  // [set][row group][key block]; only set 0 is used without RB_ASMQK
  float sv[2][BC / 2];      // !RB_ASMQK: scores of the tile being exponentiated; sv[g][h] = s_acc[0][g][h / 16][h % 16]
  f16v minit[2];            // RB_PRE: the accumulators of a tile start at -m (all 16 registers of a lane equal)
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[g][r] = 0.f;
  // opaque to hipcc and pinned in the half the MFMAs read it from: otherwise the splat of -m is re-materialised
  // (16 moves per row group) in front of every tile
  auto pin_minit = [&]() {
    if constexpr (PRE && ASMQK) asm volatile("" : "+v"(minit[0]), "+v"(minit[1]));
    else if constexpr (PRE) asm volatile("" : "+a"(minit[0]), "+a"(minit[1]));
  };
  pin_minit();
  h8 pf[2][G::NST];

  constexpr int NQK = G::NK * G::KB;   // K fragments per tile (two MFMAs each)
  constexpr int NPV = G::NST * G::NDB;  // V fragments per tile (two MFMAs each)
  constexpr int PD = (OPT & RB_PD2) ? 2 : 4;  // fragments in flight ahead of their MFMAs

  auto k_frag = [&](int kb_j, int t) {  // keys (t % KB)*32 + l31, k-step t / KB
    const int ks = t / G::KB;
    return *reinterpret_cast<const h8*>(smem + (kb_j ^ ((ks & 7) << 5)) + (ks >> 3) * 256 + (t % G::KB) * 32 * G::ROW);
  };
  auto v_frag = [&](int vb_j, int idx) {  // rows 16*st + v_row and + 8, d block b
    const int st = idx / G::NDB, b = idx % G::NDB;
    const char* vp = smem + (vb_j ^ ((b & 3) << 6)) + (16 * st) * G::ROW + (b >> 2) * 256;
    return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
  };
  auto max2 = [&](float m, float a, float b) {
    if constexpr ((OPT & RB_ASMMAX) != 0) return rb_max3(m, a, b);
    else return fmaxf(fmaxf(m, a), b);
  };
  // one K-fragment step of S^T = K Q^T into score set `set`: two MFMAs (one per row group) on the same fragment;
  // k-step 0 starts the chain at -m (RB_PRE) or 0
  auto qk_step = [&](int set, const h8& kf, int t) {
    const int ks = t / G::KB, kb = t % G::KB;
    if constexpr (ASMQK) {
      // "s_nop 1": hipcc may marshal an operand with a VALU move right in front of the statement (VALU write -> MFMA
      // read needs 2 wait states it cannot see). "=&v": the result tuple must not overlap any source tuple.
      if (ks == 0) {
        if constexpr (PRE)
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %5\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %6"
                       : "=&v"(s_acc[set][0][kb]), "=&v"(s_acc[set][1][kb])
                       : "v"(kf), "a"(qf[0][0]), "a"(qf[1][0]), "v"(minit[0]), "v"(minit[1]));
        else
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, 0"
                       : "=&v"(s_acc[set][0][kb]), "=&v"(s_acc[set][1][kb])
                       : "v"(kf), "a"(qf[0][0]), "a"(qf[1][0]));
      } else {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1"
                     : "+v"(s_acc[set][0][kb]), "+v"(s_acc[set][1][kb])
                     : "v"(kf), "a"(qf[0][ks]), "a"(qf[1][ks]));
      }
    } else {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (ks == 0) {
          if constexpr (PRE) s_acc[set][g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g][0], minit[g], 0, 0, 0);
          else s_acc[set][g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g][0], f16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
        } else {
          s_acc[set][g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g][ks], s_acc[set][g][kb], 0, 0, 0);
        }
      }
    }
  };
  // the same, ONE row group at a time (RB_HALF: every MFMA gets its own share of the softmax work behind it -- two
  // MFMAs back to back leave the second one waiting ~28 cycles for the matrix pipe with nothing issued meanwhile)
  auto qk_one = [&](int set, const h8& kf, int t, int g) {
    const int ks = t / G::KB, kb = t % G::KB;
    if constexpr (ASMQK) {
      if (ks == 0) {
        if constexpr (PRE)
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s_acc[set][g][kb]) : "v"(kf), "a"(qf[g][0]), "v"(minit[g]));
        else
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s_acc[set][g][kb]) : "v"(kf), "a"(qf[g][0]));
      } else {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s_acc[set][g][kb]) : "v"(kf), "a"(qf[g][ks]));
      }
    } else {
      if (ks == 0) {
        if constexpr (PRE) s_acc[set][g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g][0], minit[g], 0, 0, 0);
        else s_acc[set][g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g][0], f16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
      } else {
        s_acc[set][g][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[g][ks], s_acc[set][g][kb], 0, 0, 0);
      }
    }
  };
  // RB_ASMQK: nothing may read score set `set` before the last asm MFMA that wrote it has retired (XDL write -> VALU
  // read: up to 19 wait states for a 16-pass MFMA; hipcc does not see the MFMA). The operands tie every later read of
  // the set to this statement.
  auto rb_fence = [&](int set) {
    if constexpr (ASMQK) {
      if constexpr (G::KB == 2)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s_acc[set][0][0]), "+v"(s_acc[set][0][1]), "+v"(s_acc[set][1][0]), "+v"(s_acc[set][1][1]));
      else
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s_acc[set][0][0]), "+v"(s_acc[set][1][0]));
    }
  };
  // exponentials / row sums / fp16 conversion of elements [g0, g0 + cnt) of the pending tile's scores
  // (element g: row group g / (BC/2), h = g % (BC/2); P^T fragment u = h / 8 = k-step of 16 keys, slot e = h % 8)
  float psum[2] = {0.f, 0.f};
  auto p_slice = [&](int set, int g0, int cnt) {
#pragma unroll
    for (int g = g0; g < g0 + cnt; g += 2) {
      const int grp = g / (BC / 2), h = g % (BC / 2), u = h / 8, e = h % 8;
      float x0, x1;
      if constexpr (ASMQK) x0 = s_acc[set][grp][h / 16][h % 16], x1 = s_acc[set][grp][h / 16][h % 16 + 1];
      else x0 = sv[grp][h], x1 = sv[grp][h + 1];
      if constexpr (!PRE) {
        x0 = fmaf(x0, scale_log2e, -m_run[grp]);
        x1 = fmaf(x1, scale_log2e, -m_run[grp]);
      }
      const float a0 = (ABL & 2) ? x0 : __builtin_amdgcn_exp2f(x0);
      const float a1 = (ABL & 2) ? x1 : __builtin_amdgcn_exp2f(x1);
      psum[grp] += a0 + a1;
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      pf[grp][u][e] = a[0], pf[grp][u][e + 1] = a[1];
    }
  };
  // fold elements [g0, g0 + cnt) of the finished score set into the row max (and, without RB_ASMQK, move them to the
  // VALU side)
  float mx[2];
  auto out_slice = [&](int set, int g0, int cnt) {
#pragma unroll
    for (int g = g0; g < g0 + cnt; g += 2) {
      const int grp = g / (BC / 2), h = g % (BC / 2);
      float x0 = s_acc[set][grp][h / 16][h % 16], x1 = s_acc[set][grp][h / 16][h % 16 + 1];
      if constexpr (!ASMQK) {
        asm volatile("" : "+v"(x0), "+v"(x1));  // one move per score, into the VALU half of the register file
        sv[grp][h] = x0, sv[grp][h + 1] = x1;
      }
      if (h == 0) mx[grp] = fmaxf(x0, x1);
      else mx[grp] = max2(mx[grp], x0, x1);
    }
  };
  // decide / rescale once a tile's row maxima are known; FIRST: tile 0 (adopt the max unconditionally)
  auto decide = [&](int set, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    float d[2];
    bool grow = false;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[g]), __float_as_uint(mx[g]), false, false);
      const float m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      d[g] = PRE ? m : fmaf(m, scale_log2e, -m_run[g]);  // growth of the row max over the running one (log2 domain)
      grow |= d[g] > ((OPT & RB_DEFER) ? 8.0f : 0.0f);
    }
    if (FIRST || __builtin_expect(__builtin_amdgcn_ballot_w64(grow) != 0, 0)) {
      asm volatile("; rescale" ::: "memory");  // a real (cold) branch: never if-converted into the tile loop
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float delta = FIRST ? d[g] : fmaxf(d[g], 0.f);
        m_run[g] += delta;  // tile 0: m_run was 0
        if constexpr (!FIRST) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run[g] *= alpha;
#pragma unroll
          for (int b = 0; b < G::NDB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[g][b][r] *= alpha;
        }
        if constexpr (PRE) {  // the pending scores were accumulated from the old -m
#pragma unroll
          for (int h = 0; h < BC / 2; ++h) {
            if constexpr (ASMQK) s_acc[set][g][h / 16][h % 16] -= delta;
            else sv[g][h] -= delta;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) minit[g][r] = -m_run[g];
        }
      }
      pin_minit();
    }
  };

  // ---- prologue: S(0), its row max, the first running max
#pragma unroll
  for (int t = 0; t < NQK; ++t) qk_step(0, k_frag(kbase, t), t);
  rb_fence(0);
  out_slice(0, 0, G::NEL);
  decide(0, std::true_type{});
  hgemm::wait_vmcnt<0>();  // tile 1 landed (this wave's pieces); the barrier makes everyone's visible
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int slot_j = 0;  // ring slot of tile j (tile j+1: slot_j + 1, tile j+2: slot_j + 2, mod 3)
  // one KV tile; CUR = score set of tile j (exponentiated here), the other set receives tile j+1
  auto tile = [&](int j, auto cur_tag) __attribute__((always_inline)) {
    constexpr int CUR = ASMQK ? decltype(cur_tag)::value : 0, NXT = ASMQK ? 1 - CUR : 0;
    const int s1 = slot_j == 2 ? 0 : slot_j + 1, s2 = slot_j == 0 ? 2 : slot_j - 1;
    const int jd = j + 2 < T ? j + 2 : T - 1;  // past the end: refill a dead slot with the last tile (branch-free)
    const int kb_j = kbase + s1 * G::STAGE, vb_j = vbase + slot_j * G::STAGE + G::TILE;
    // ================= R1: S(j+1) = K_{j+1} Q^T   ||   P(j) = exp2(S(j) - m), row sums, fp16 conversion
    psum[0] = 0.f, psum[1] = 0.f;
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(kb_j, i);
      constexpr int EPS = G::NEL / NQK;          // softmax elements per fragment step
      constexpr int DSTEP = NQK / (2 * G::PPW);  // one DMA piece every DSTEP steps
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        if constexpr ((OPT & RB_HALF) != 0) {
          qk_one(NXT, kf[t % PD], t, 0);
          p_slice(CUR, t * EPS, EPS / 2);
          if (t % DSTEP == 0) dma_piece(t / DSTEP, jd, s2);
          __builtin_amdgcn_sched_barrier(0);
          qk_one(NXT, kf[t % PD], t, 1);
          __builtin_amdgcn_sched_barrier(0);  // the fragment register is re-filled only after both MFMAs were issued
          if (t + PD < NQK) kf[t % PD] = k_frag(kb_j, t + PD);
          p_slice(CUR, t * EPS + EPS / 2, EPS / 2);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          qk_step(NXT, kf[t % PD], t);
          if (t + PD < NQK) kf[t % PD] = k_frag(kb_j, t + PD);
          if (t % DSTEP == 0) dma_piece(t / DSTEP, jd, s2);
          p_slice(CUR, t * EPS, EPS);
          if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    l_run[0] += psum[0], l_run[1] += psum[1];
    __builtin_amdgcn_sched_barrier(0);
    // ================= R2: O^T += V_j^T P(j)^T   ||   row max of S(j+1) (moved to the VALU side without RB_ASMQK)
    {
      h8 vf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) vf[i] = v_frag(vb_j, i);
      constexpr int MPS = G::NEL / NPV;  // scores per fragment step
      constexpr int FENCE_AT = ASMQK ? 1 : 0;  // the asm MFMAs of R1 retire under the first PV step
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        const int st = i / G::NDB, b = i % G::NDB;
        if constexpr ((OPT & RB_HALF) != 0) {
          ot[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i % PD], pf[0][st], ot[0][b], 0, 0, 0);
          if (ASMQK && i == 0) rb_fence(NXT);
          if (i >= FENCE_AT) out_slice(NXT, (i - FENCE_AT) * MPS, MPS / 2);
          __builtin_amdgcn_sched_barrier(0);
          ot[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i % PD], pf[1][st], ot[1][b], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (i + PD < NPV) vf[i % PD] = v_frag(vb_j, i + PD);
          if (i >= FENCE_AT) out_slice(NXT, (i - FENCE_AT) * MPS + MPS / 2, MPS / 2);
          if (FENCE_AT && i == NPV - 1) out_slice(NXT, (NPV - FENCE_AT) * MPS, FENCE_AT * MPS);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          ot[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i % PD], pf[0][st], ot[0][b], 0, 0, 0);
          ot[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i % PD], pf[1][st], ot[1][b], 0, 0, 0);
          if (i + PD < NPV) vf[i % PD] = v_frag(vb_j, i + PD);
          if (ASMQK && i == 0) rb_fence(NXT);
          if (i >= FENCE_AT) out_slice(NXT, (i - FENCE_AT) * MPS, MPS);
          if (FENCE_AT && i == NPV - 1) out_slice(NXT, (NPV - FENCE_AT) * MPS, FENCE_AT * MPS);
          if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    decide(NXT, std::false_type{});
    // own DMA pieces of tile j+2 landed; everyone behind the barrier is done with tile j's slot
    hgemm::wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot_j = s1;
  };
  for (int j = 0; j < T; j += 2) {  // T is even (N % 256 == 0): the two score sets alternate statically
    tile(j, std::integral_constant<int, 0>{});
    tile(j + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows), 16-byte row segments out
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;  // not carried across the KV loop
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float l_tot;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[g]), __float_as_uint(l_run[g]), false, false);
      l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
    char* ob = smem + (wave * 2 + g) * (32 * G::OS);
#pragma unroll
    for (int b = 0; b < G::NDB; ++b) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[g][b][rq * 4 + e] * inv);
        *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int LPR = D / 8;  // 16-byte segments per row
    half_t* og = O + head + (size_t)(q_row0 + g * 32) * D;
#pragma unroll 4
    for (int it = 0; it < (32 * LPR) / 64; ++it) {
      const int idx = it * 64 + lane_e;
      const int row = idx / LPR, c = idx % LPR;
      *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    }
  }
}

template <int D, int BC, int OPT, int ABL = 0>
int launch_rb(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoRB<D, BC>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_rb_kernel<D, BC, OPT, ABL>), G::LDS_BYTES) != CLN_OK)
    return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_rb_kernel<D, BC, OPT, ABL>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
