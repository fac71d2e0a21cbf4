// FlashAttention-2 forward for head dims 640 / 768 / 1024 (round 5): ONE wave per SIMD. Four waves split the head dim of one
// 64-row query block; each wave has the whole 512-entry register file of its SIMD. Reference rungs: the fine-grained tiling
// kernels, whose head-dim switch goes up to d = 1024 (kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:70, :852-870;
// flash_attn_mma.py:436-506).
//
// Why a third kernel for these head dims (flash_attn_dring.cuh sits at 0.27-0.31 of the fp16 MFMA peak). The ring kernel
// runs 8 waves = 2 row groups x 4 d-parts with 256 registers each. Per 16-key step and CU that is
//     128 KiB of K / V fragment reads (both row groups read every fragment),  64 KiB of partial-S reads (every wave of a row group
//     reads all four partials and repeats the same softmax), 16 KiB of partial-S writes and 64 KiB of LDS-DMA writes
// against 1024 matrix clocks: the LDS array, the DMA path and the matrix pipe are all near saturation and their times ADD
// (profiles/r03_fa_dring_lds_counters.log). The review's suggestion -- the reference's packed-fp16 O between tiles
// (flash_attn_mma_tiling_qkv.cu:756) so that a workgroup owns 128 rows -- does not carry over: gfx950 has no fp16-accumulating MFMA, so
// packed O means 16 unpack + 8 pack VALU instructions (>= 96 issue clocks) around every 32-clock 32x32x16 MFMA, and Q (fp16) + O (fp16) of
// 128 rows x 1024 columns is the whole 512 KiB register file. What CAN be halved at 64 rows per CU is everything else:
//   * ONE wave owns all 64 rows of its quarter of d -> every K / V fragment is read from LDS ONCE per step (64 KiB instead of 128);
//     O^T (64 x D/4 fp32 = 256 registers at D = 1024) lives in the AGPR half through inline-asm MFMAs with a tied accumulator
//     (the hgemm_w4.cuh idiom), Q (128 registers) and the working set in the VGPR half;
//   * the softmax is done ONCE per row: wave w owns rows 16w..16w+15 of the block, reads the four partials of those rows only
//     (reduce-scatter: 16 KiB instead of 64), exponentiates 4 scores per lane instead of 8 x 4-fold redundancy, and publishes P as
//     fp16 (2 KiB) plus one rescale factor per row; every wave reads the 64 x 16 P block in the B-operand layout of the PV MFMA;
//   * the single instruction stream is software-pipelined: QK^T of tile j+1 is issued under the softmax of tile j (phase A), the
//     partial-S writes of tile j+1 and the next K request under the PV MFMAs of tile j (phase B); two workgroup barriers per step.
//   * the running maximum is only raised when a row's tile maximum exceeds it by > 8 (log2 domain; P <= 256 in fp16, sums in fp32):
//     on the accumulators' AGPRs a rescale costs 3 instructions per register, so it has to be rare. The first tile never rescales
//     (O and l are zero). tests/test_gpu_flash_attn.py drives the rescale path with keys whose scale grows along the sequence.
// LDS: K and V two-slot rings of 16-key tiles exactly as in the ring kernel (lane-linear LDS-DMA images, source-side XOR swizzles:
// K chunk ^= row & 15, V chunk ^= (row & 3) << 2), then [4 waves][64 rows][16 keys] fp32 partials (16 KiB, chunk ^= (row >> 1) & 3:
// conflict-free for the 8-lane groups of ds_write_b128 and the 16-lane groups of ds_read_b128, which use the SAME lane -> (row, chunk)
// map here), P as [4 key chunks][64 rows] x 8 B (2 KiB), alpha / l as 64 floats. 146.25 KiB at D = 1024.
#pragma once
#include "flash_attn_dsplit.cuh"

namespace fa2 {

// RM = 64-row blocks per workgroup: 1 at D = 640 / 768 / 1024 (O^T of 64 rows x D/4 columns fills the AGPR half), 2 at D = 512 (config C5: 128 rows,
// half the K / V bytes per flop -- the same register budget: 128 rows x 128 columns).
template <int D, int RM = (D == 512 ? 2 : 1)>
struct GeoDW4 {
  static_assert(((D == 640 || D == 768 || D == 1024) && RM == 1) || (D == 512 && RM == 2), "head dims 512 (128 rows) / 640 / 768 / 1024 (64 rows)");
  static constexpr int NSP = 4, DH = D / 4, BC = 16, NW = 4, BR = 64 * RM, NT = 256;
  static constexpr int ROW = D * 2, TILE = BC * ROW, NP = TILE / 1024, PPW = NP / NW;  // 1-KiB DMA pieces per tile: 20 / 24 / 32 -> 5 / 6 / 8 per wave
  static_assert(NP % NW == 0, "every wave carries the same number of pieces");
  static constexpr int RING = 4 * TILE;           // K slot 0, K slot 1, V slot 0, V slot 1
  static constexpr int SX = RING;                 // partial S^T
  static constexpr int PX = SX + NW * 4096 * RM;  // P (fp16)
  static constexpr int AX = PX + 2048 * RM;       // rescale factors / row sums
  static constexpr int MAIN = AX + 256 * RM;
  static constexpr int OS = DH * 2 + 16, EPI = NW * 32 * OS;  // epilogue staging: 32 rows per wave and pass
  static constexpr int LDS_BYTES = MAIN > EPI ? MAIN : EPI;
  static constexpr int NKS = DH / 32, NDB = DH / 32, CPP = DH / 8;  // k-steps, output blocks, 16-byte chunks per part
  static_assert(LDS_BYTES <= 160 * 1024 && (ROW / 16) % 16 == 0, "LDS / swizzle range");
};

// DW4_CARRY: the last MFMA group of a phase (operands already in registers) is issued AFTER the barrier, under the LDS round trips the next phase
// starts with. DW4_M0WALK: the pieces of a tile request walk M0 (2 instructions per piece instead of 4). DW4_SPREAD: the softmax in four sections
// behind four MFMA groups instead of two. (DW4_ABL_*: probe ablations, garbage results.)
// DW4_ABL_NOWAIT (probe ablation, garbage results): the requests are issued but their landing is never waited for -- separates the cost of ISSUING the
// LDS-DMA from the cost of WAITING for it (a two-slot ring gives a request one step, ~1 us, to land).
// DW4_UNROLL2: two tiles per loop iteration, so that the ring-slot parity of every LDS address is a compile-time constant (fragment addresses become
// register + immediate: no per-fragment address arithmetic) and the hazard pads in front of MFMA groups whose operands come straight from LDS go away --
// the kernel is ISSUE-bound: one wave per SIMD issues ~1 instruction per 4-5 clocks and a 16-key step carried 213 instructions for 36 MFMAs at D = 768.
// DW4_SKEW4 / DW4_SKEW8 (probe, negative result): after every barrier of the loop wave w idles w * 4 (8) wait states = 16 (32) clocks, so that the four
// waves' LDS-DMA requests -- issued at the same point of the same instruction stream -- reach the CU's one address unit a piece-time apart instead of
// together. Bit-identical; -2 % (D = 1024) ... -4 % (D = 768): the idling costs what it was meant to save, request collisions are NOT what the 45 % of
// wave cycles spent waiting to issue (profiles/r05_pmc_fa_d1024_dw4.json) are made of (profiles/r05_fa_dw4_skew_probe.log). A zero-idle form -- four copies
// of the loop, one per wave index, each placing its request behind MFMA w of a group -- was written too: hipcc then no longer keeps O^T in one AGPR
// block (80 / 240 spilled registers at D = 768 / 1024), not measured.
enum : int { DW4_1STAGE = 1, DW4_NO_DEFER = 2, DW4_ABL_DMA = 4, DW4_ABL_SOFTMAX = 8, DW4_CARRY = 16, DW4_M0WALK = 32, DW4_SPREAD = 64, DW4_UNROLL2 = 128,
              DW4_SKEW4 = 256, DW4_SKEW8 = 512, DW4_ABL_NOWAIT = 1024, DW4_DEFAULT = DW4_CARRY | DW4_M0WALK | DW4_SPREAD };

template <int D, int OPT = 0, int KPF = 2, int VPF = 2>
__global__ __launch_bounds__(256, 1) void fa2_fwd_dw4_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                             const half_t* __restrict__ V, half_t* __restrict__ O, int N,
                                                             int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoDW4<D>;
  constexpr int NKS = G::NKS, NDB = G::NDB, PPW = G::PPW;
  constexpr int RM = G::BR / 64, RB16 = 4 * RM, RB32 = 2 * RM;  // 16-row blocks (S^T partials), 32-row blocks (O^T) of the workgroup's rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = d-part, = owner of rows 16*wave .. +15 in the softmax
  const int l31 = lane & 31, hi = lane >> 5, i16 = lane & 15, g4 = lane >> 4;
  const int part = wave;

  int head_i, qb_i;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {  // heads pinned to XCDs: the workgroups of a head share one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb_i = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb_i = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb_i * G::BR;
  const unsigned lds0 = hgemm::lds_addr_of(smem);
  const char* Kh = reinterpret_cast<const char*>(K + head);
  const char* Vh = reinterpret_cast<const char*>(V + head);

  // ---- LDS-DMA: piece p = i * 4 + wave of an operand tile is the lane-linear KiB p of its image: byte o = p * 1024 + lane * 16 =
  // chunk c of row r, fetched from source chunk c ^ swizzle(r) of the same row (flash_attn_dring.cuh)
  unsigned k_voff[PPW], v_voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int o = (i * G::NW + wave) * 1024 + lane * 16;
    const int r = o / G::ROW, c = (o % G::ROW) >> 4;
    k_voff[i] = (unsigned)(r * G::ROW + ((c ^ (r & 15)) << 4));
    v_voff[i] = (unsigned)(r * G::ROW + ((c ^ ((r & 3) << 2)) << 4));
  }
  const int T = N / G::BC;
  __builtin_assume(T > 0);
  auto clampt = [&](int t) __attribute__((always_inline)) { return t < T ? t : T - 1; };  // past the end: refill a dead slot (uniform counts)
  auto piece = [&](bool is_v, int t, int slot, int i) __attribute__((always_inline)) {
    if constexpr ((OPT & DW4_ABL_DMA) != 0) return;
    const char* src = (is_v ? Vh : Kh) + (size_t)clampt(t) * G::TILE;
    const unsigned dst = lds0 + (is_v ? 2 * G::TILE : 0) + slot * G::TILE + (unsigned)(i * G::NW + wave) * 1024u;
    // M0 is ours for the whole kernel (no other instruction of it reads M0): no save / restore around the request
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(is_v ? v_voff[i] : k_voff[i]), "s"(src), "s"(dst) : "memory", "m0");
  };
  // DW4_M0WALK: piece 0 of a tile request sets M0, every piece leaves it 4 KiB further (this wave's next destination): the pieces of ONE
  // request must be issued in order 0, 1, ... and nothing else may touch M0 between them (nothing else in this kernel uses M0)
  auto piece_w = [&](bool is_v, int t, int slot, int i) __attribute__((always_inline)) {  // tile t into ring slot `slot` (= t & 1)
    if constexpr ((OPT & DW4_ABL_DMA) != 0) return;
    if constexpr ((OPT & DW4_M0WALK) == 0) {
      piece(is_v, t, slot, i);
    } else {
      const char* src = (is_v ? Vh : Kh) + (size_t)clampt(t) * G::TILE;
      const unsigned voff = is_v ? v_voff[i] : k_voff[i];
      if (i == 0) {
        const unsigned dst = lds0 + (is_v ? 2 * G::TILE : 0) + slot * G::TILE + (unsigned)wave * 1024u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0", "scc");
      } else {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" ::"v"(voff), "s"(src) : "memory", "m0", "scc");
      }
    }
  };
  auto req_tile = [&](bool is_v, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece_w(is_v, t, slot, i);
  };

  // ---- Q fragments (B operand of S^T = K Q^T on 16x16x32): query 16*rb + i16, d = part*DH + 32*ks + 8*g4 .. +7
  h8 qf[RB16][NKS];
#pragma unroll
  for (int rb = 0; rb < RB16; ++rb) {
    const half_t* qp = Q + head + (size_t)(q_row0 + rb * 16 + i16) * D + part * G::DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[rb][ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  // O^T accumulators [32-row block][32-wide d block]: tied to AGPR tuples by the inline-asm MFMAs below
  f16v ot[RB32][NDB];
#pragma unroll
  for (int rb = 0; rb < RB32; ++rb)
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[rb][b][r] = 0.f;
#pragma unroll
  for (int rb = 0; rb < RB32; ++rb)
#pragma unroll
    for (int b = 0; b < NDB; ++b) asm volatile("" : "+a"(ot[rb][b]));  // zero-fill done HERE (inline-asm MFMAs are invisible to the hazard pass)
  asm volatile("s_nop 7");
  float m_run[RM], l_run[RM];  // owner lanes: rows 16*RM*wave + 16*o + i16 (replicated over g4 for m, partial over g4 for l)
#pragma unroll
  for (int o = 0; o < RM; ++o) m_run[o] = -1.0e30f, l_run[o] = 0.f;

  req_tile(false, 0, 0);
  req_tile(true, 0, 0);
  req_tile(false, 1, 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: Q and the first tiles
#pragma unroll
  for (int rb = 0; rb < RB16; ++rb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[rb][ks]));  // keep the Q loads out of the KV loop
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- fragment addresses (as flash_attn_dring.cuh). K fragment ks (A operand, 16 keys x 32 d): row i16, logical chunk part*CPP + 4*ks + g4.
  // V^T fragment b (A operand of the 32x32x16 PV step, 32 d x 16 keys): two transposing reads, rows v_row and v_row + 8: lane half `hi`
  // holds keys 4*hi .. +3 and 8 + 4*hi .. +3; P is published in the same key order.
  constexpr bool POW2 = G::CPP % 16 == 0;
  const int v_row = 4 * hi + (i16 >> 2);
  const int v_w = ((lane >> 4) & 1) * 2 + ((i16 & 3) >> 1);
  auto k_off_of = [&](int ks) { return (unsigned)(i16 * G::ROW + (((part * G::CPP + 4 * ks + g4) ^ i16) << 4)); };
  auto v_off_of = [&](int b) {
    return (unsigned)(2 * G::TILE + v_row * G::ROW + (((part * G::CPP + 4 * b + v_w) ^ ((v_row & 3) << 2)) << 4) + ((i16 & 1) << 3));
  };
  // DW4_UNROLL2 at POW2: FOUR bases per operand (the XOR with (i & 3) << 6 done once), fragment i = base[i & 3] + (i >> 2) * 256 + slot * TILE with the
  // last two terms in the instruction's immediate offset: no address arithmetic in the loop
  constexpr bool U2 = (OPT & DW4_UNROLL2) != 0;
  constexpr int NKB = POW2 ? (U2 ? 4 : 1) : NKS, NVB = POW2 ? (U2 ? 4 : 1) : NDB;
  unsigned koff[NKB], voffs[NVB];
#pragma unroll
  for (int ks = 0; ks < NKB; ++ks) koff[ks] = k_off_of(ks);  // (POW2: k_off_of(q) == k_off_of(0) ^ q << 6 for q < 4)
#pragma unroll
  for (int b = 0; b < NVB; ++b) voffs[b] = v_off_of(b);
  if constexpr (POW2) {
#pragma unroll
    for (int i = 0; i < NKB; ++i) asm volatile("" : "+v"(koff[i]), "+v"(voffs[i]));  // opaque: hipcc would otherwise hoist one register per fragment
  }
  auto k_addr = [&](int ks) __attribute__((always_inline)) -> unsigned {
    if constexpr (POW2 && U2) return koff[ks & 3] + (unsigned)((ks >> 2) * 256);
    else if constexpr (POW2) return (koff[0] ^ (unsigned)((ks & 3) << 6)) + (unsigned)((ks >> 2) * 256);
    else return koff[ks];
  };
  auto v_addr = [&](int b) __attribute__((always_inline)) -> unsigned {
    if constexpr (POW2 && U2) return voffs[b & 3] + (unsigned)((b >> 2) * 256);
    else if constexpr (POW2) return (voffs[0] ^ (unsigned)((b & 3) << 6)) + (unsigned)((b >> 2) * 256);
    else return voffs[b];
  };
  // partial-S image: wave w, row R (0..63), 16 keys fp32 = 64-byte rows, 16-byte chunk g4 ^ ((R >> 1) & 3); writer lane (i16, g4) of
  // row block rb holds keys 4*g4..+3 of row 16*rb + i16, the owner wave reads rows 16*wave + i16 in the SAME lane layout
  const int sx_lane = i16 * 64 + ((g4 ^ ((i16 >> 1) & 3)) << 4);
  char* sx_w = smem + G::SX + wave * (4096 * RM) + sx_lane;            // + rb * 1024
  const char* sx_r = smem + G::SX + wave * (1024 * RM) + sx_lane;      // + o * 1024 + p * 4096 * RM
  char* px_w = smem + G::PX + g4 * (512 * RM) + (wave * 16 * RM + i16) * 8;  // + o * 128
  char* ax_w = smem + G::AX + (wave * 16 * RM + i16) * 4;                // + o * 64
  const char* px_r = smem + G::PX + hi * (512 * RM) + l31 * 8;         // + rb * 256 (+ 1024 * RM: second key chunk)
  const char* ax_r = smem + G::AX + l31 * 4;                           // + rb * 128

  f4 s[RB16];
  constexpr bool CARRY = (OPT & DW4_CARRY) != 0;
  h8 kf_carry;  // DW4_CARRY: the K fragment of the last k-step, read before the barrier that ends phase A, multiplied after it
  // S^T partial of tile t (in K slot t & 1) over this wave's quarter of d: 4 row blocks x NKS k-steps, KPF fragments in flight.
  // `hook(ks)` runs after the four MFMAs of k-step ks (DMA pieces, softmax sections). With DW4_CARRY the last k-step is left to qk_finish().
  auto qk_group = [&](int ks, const h8& kfr, bool pad = true) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < RB16; ++rb) {
      // inline asm with the accumulator tied to ONE VGPR tuple (early-clobber on the first k-step): left to the builtin, hipcc parks the
      // four partial tiles in AGPRs beside O^T and copies them out after every MFMA (s_nop 7 + 4 v_accvgpr_read per MFMA)
      // (s_nop 1 in front of a group: a VGPR written by a VALU instruction needs two wait states before an MFMA reads it as A / B, and hipcc's
      // hazard pass does not see into inline asm -- the fragments normally come straight from LDS reads, but register copies it inserts are VALU)
      // (DW4_UNROLL2 drops the pad where the K fragment comes straight from its ds_read_b128 -- Q never changes --: only the carried group keeps it)
      if (ks == 0 && rb == 0 && pad) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s[rb]) : "v"(kfr), "v"(qf[rb][0]));
      else if (ks == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s[rb]) : "v"(kfr), "v"(qf[rb][0]));
      else if (rb == 0 && pad) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s[rb]) : "v"(kfr), "v"(qf[rb][ks]));
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s[rb]) : "v"(kfr), "v"(qf[rb][ks]));
    }
  };
  auto qk_tile = [&](int slot, auto&& hook) __attribute__((always_inline)) {
    const char* kb = smem + slot * G::TILE;
    constexpr int KD = KPF < NKS ? KPF : NKS;
    h8 kf[KD];
#pragma unroll
    for (int i = 0; i < KD; ++i) kf[i] = *reinterpret_cast<const h8*>(kb + k_addr(i));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (CARRY && ks == NKS - 1) {
        kf_carry = kf[ks % KD];
      } else {
        qk_group(ks, kf[ks % KD], !U2);
        if (ks + KD < NKS) kf[ks % KD] = *reinterpret_cast<const h8*>(kb + k_addr(ks + KD));
        __builtin_amdgcn_sched_barrier(0);  // the hook's VALU work goes BEHIND the four MFMAs (into their shadow), not in front of them
      }
      hook(ks);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto qk_finish = [&]() __attribute__((always_inline)) {
    if constexpr (CARRY) {
      asm volatile("s_nop 1" : "+v"(kf_carry));  // (a register copy of the carried fragment, if hipcc made one, is VALU: see the P fragments below)
      qk_group(NKS - 1, kf_carry, false);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (the inline-asm MFMAs are invisible to hipcc's hazard pass: the partials are stored a barrier and >= 2 PV MFMAs after the last QK^T
  // MFMA was issued -- far beyond the 11 wait states an 8-pass MFMA result needs before an LDS store may read it; the pad covers the prologue)
  auto write_partials = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 7" ::: "memory");
#pragma unroll
    for (int rb = 0; rb < RB16; ++rb) *reinterpret_cast<f4*>(sx_w + rb * 1024) = s[rb];
  };
#define DW4_BARRIER()              \
  do {                             \
    __builtin_amdgcn_s_barrier();  \
    asm volatile("" ::: "memory"); \
  } while (0)
  auto skew = [&]() __attribute__((always_inline)) {
    if constexpr ((OPT & (DW4_SKEW4 | DW4_SKEW8)) != 0) {
      constexpr int WS = (OPT & DW4_SKEW8) != 0 ? 7 : 3;  // s_nop n = n + 1 wait states of 4 clocks
      if (wave >= 1) asm volatile("s_nop %0" ::"n"(WS));
      if (wave >= 2) asm volatile("s_nop %0" ::"n"(WS));
      if (wave >= 3) asm volatile("s_nop %0" ::"n"(WS));
    }
  };

  // ---- prologue: S(0), K(2) requested into the slot S(0) has just left
  qk_tile(0, [&](int) {});
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragment reads of K slot 0 are done
  DW4_BARRIER();
  qk_finish();
  req_tile(false, 2, 0);
  if constexpr ((OPT & DW4_1STAGE) != 0) hgemm::wait_vmcnt<0>();
  write_partials();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  DW4_BARRIER();

  // DW4_CARRY: the PV MFMAs of the last d block of a tile (V fragment and both P fragments already in registers) are issued after the barrier
  // that ends phase B, at the head of the next phase A. Zero fragments before the first tile: the carried MFMAs then add nothing.
  h8 vf_carry, pf_carry[RB32];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    vf_carry[e] = (half_t)0.f;
#pragma unroll
    for (int rb = 0; rb < RB32; ++rb) pf_carry[rb][e] = (half_t)0.f;
  }
  auto pv_finish = [&]() __attribute__((always_inline)) {
    if constexpr (CARRY) {
      // (s_nop 1: see qk_group -- before the first tile the carried fragments are zeros written by v_mov right here; without the pad the first
      // MFMA read stale registers: NaN at D = 640 / 768, profiles/r05_fa_dw4_probe.log)
      asm volatile("" : "+v"(vf_carry));
#pragma unroll
      for (int rb = 0; rb < RB32; ++rb) asm volatile("" : "+v"(pf_carry[rb]));
      asm volatile("s_nop 1" ::: "memory");  // every carried fragment is in its final registers, two wait states before the first MFMA reads them
#pragma unroll
      for (int rb = 0; rb < RB32; ++rb) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[rb][NDB - 1]) : "v"(vf_carry), "v"(pf_carry[rb]));
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // one tile: `par` = j & 1 (a literal in the DW4_UNROLL2 form: every slot offset below is then a compile-time constant)
  auto step = [&](const int j, const int par) __attribute__((always_inline)) {
    // ================= phase A: softmax of tile j by the row owners + S^T partial of tile j+1; V(j+1) requested
    {
      f4 ap[RM][4];  // the four d-parts' partials of this wave's 16 * RM rows (summed in part order by every owner: the order is fixed)
#pragma unroll
      for (int o = 0; o < RM; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) ap[o][p] = *reinterpret_cast<const f4*>(sx_r + o * 1024 + p * (4096 * RM));
      pv_finish();  // (tile j-1's last d block: runs under the LDS round trip of the reads above and of the first K fragments)
      f4 a[RM];
      float p4[RM][4];
      float alpha[RM], mx[RM];
      constexpr bool SPREAD = (OPT & DW4_SPREAD) != 0;
      // softmax sections: behind the MFMA groups of k-steps 0, 1 (two sections) or 0 .. 3 (DW4_SPREAD: at most ~10 VALU per 64-clock group and row block)
      auto sec_max_lane = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < RM; ++o) {
          a[o] = (ap[o][0] + ap[o][1]) + (ap[o][2] + ap[o][3]);
          mx[o] = fmaxf(fmaxf(a[o][0], a[o][1]), fmaxf(a[o][2], a[o][3]));  // row maximum over the 16 keys of the tile: 4 in the lane ...
        }
      };
      auto sec_max_row = [&]() __attribute__((always_inline)) {  // ... then the four g4 lanes of the row
#pragma unroll
        for (int o = 0; o < RM; ++o) {
          const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx[o]), __float_as_uint(mx[o]), false, false);
          mx[o] = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
          const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[o]), __float_as_uint(mx[o]), false, false);
          mx[o] = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
          const float mxs = mx[o] * scale_log2e;
          bool grow;
          if constexpr ((OPT & DW4_NO_DEFER) != 0) grow = mxs > m_run[o];
          else grow = (mxs - m_run[o]) > 8.0f;
          const float m_new = grow ? mxs : m_run[o];
          alpha[o] = grow ? __builtin_amdgcn_exp2f(m_run[o] - m_new) : 1.f;
          m_run[o] = m_new;
          l_run[o] *= alpha[o];
          if (j == 0) alpha[o] = 1.f;  // O and l are still zero: nothing to rescale
        }
      };
      auto sec_exp = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < RM; ++o) {
          const float nm = -m_run[o];
#pragma unroll
          for (int e = 0; e < 4; ++e) p4[o][e] = __builtin_amdgcn_exp2f(fmaf(a[o][e], scale_log2e, nm));
        }
      };
      auto sec_publish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < RM; ++o) {
          l_run[o] += (p4[o][0] + p4[o][1]) + (p4[o][2] + p4[o][3]);
          const h2 lo = __builtin_convertvector(f2{p4[o][0], p4[o][1]}, h2), hh = __builtin_convertvector(f2{p4[o][2], p4[o][3]}, h2);
          *reinterpret_cast<h4*>(px_w + o * 128) = h4{lo[0], lo[1], hh[0], hh[1]};
          *reinterpret_cast<float*>(ax_w + o * 64) = alpha[o];
        }
      };
      qk_tile(par ^ 1, [&](int ks) __attribute__((always_inline)) {
        if constexpr ((OPT & DW4_1STAGE) == 0) {
          if (ks < PPW) piece_w(true, j + 1, par ^ 1, ks);
        }
        if constexpr ((OPT & DW4_ABL_SOFTMAX) != 0) return;
        if constexpr (SPREAD) {
          if (ks == 0) sec_max_lane();
          else if (ks == 1) sec_max_row();
          else if (ks == 2) sec_exp();
          else if (ks == 3) sec_publish();
        } else {
          if (ks == 0) sec_max_lane(), sec_max_row();
          else if (ks == 1) sec_exp(), sec_publish();
        }
      });
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): P / alpha are in LDS, the K fragment reads are done
    if constexpr ((OPT & (DW4_1STAGE | DW4_ABL_DMA | DW4_ABL_NOWAIT)) == 0) hgemm::wait_vmcnt<2 * PPW>();  // V(j) has landed
    DW4_BARRIER();
    skew();

    // ================= phase B: O^T += V^T P^T of tile j; the partial S^T of tile j+1 published, K(j+3) requested
    {
      const char* vb = smem + par * G::TILE;
      constexpr int VD = VPF < NDB ? VPF : NDB;
      auto rd_v = [&](int b) __attribute__((always_inline)) -> h8 {
        const char* vp = vb + v_addr(b);
        return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
      };
      float al[RB32];
      h8 pf[RB32];
#pragma unroll
      for (int rb = 0; rb < RB32; ++rb) {
        al[rb] = *reinterpret_cast<const float*>(ax_r + rb * 128);
        pf[rb] = h8_cat(*reinterpret_cast<const h4*>(px_r + rb * 256), *reinterpret_cast<const h4*>(px_r + rb * 256 + 1024 * RM));
      }
      h8 vf[VD];
#pragma unroll
      for (int i = 0; i < VD; ++i) vf[i] = rd_v(i);
      // The P fragments are assembled from two 8-byte reads each: hipcc does that with v_mov (VALU), and a VGPR written by a VALU instruction needs two
      // wait states before an MFMA reads it as A / B -- invisible to the hazard pass behind inline asm. Pin EVERY fragment here (the copies cannot
      // sink below an asm statement that takes them in / out), pad once, and no MFMA group of the phase needs a pad of its own. Round 5 found this the
      // hard way: a pad in front of the first MFMA of a group only, and the copy of the SECOND row block's fragment scheduled between the two MFMAs ->
      // one 32 x 32 tile of O wrong (profiles/r05_fa_dw4_unroll2_debug.log); tests/test_no_spills.py now scans the code object for the pattern.
#pragma unroll
      for (int rb = 0; rb < RB32; ++rb) asm volatile("" : "+v"(pf[rb]));
      asm volatile("s_nop 1" ::: "memory");
      qk_finish();  // (tile j+1's last k-step: runs under the LDS round trip of the reads above)
      if constexpr ((OPT & DW4_1STAGE) != 0) {  // `stages = 1`: both tile requests of the step in ONE burst, waited for right here --
        req_tile(false, j + 3, par ^ 1);        // no request of the wave is in flight while it computes (the V slot of tile j+1 has been
        req_tile(true, j + 1, par ^ 1);         // free since the barrier before last: same LDS images, same arithmetic, bit-identical)
        hgemm::wait_vmcnt<0>();
      }
      bool any_scaled = false;
#pragma unroll
      for (int rb = 0; rb < RB32; ++rb) any_scaled = any_scaled || al[rb] != 1.f;
      if (__builtin_amdgcn_ballot_w64(any_scaled) != 0) {  // rare (see the header): a row's maximum grew by more than 2^8
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");       // the last PV MFMAs have written their AGPRs
#pragma unroll
        for (int rb = 0; rb < RB32; ++rb)
#pragma unroll
          for (int b = 0; b < NDB; ++b) {
            asm volatile("" : "+a"(ot[rb][b]));  // re-defined AFTER the pad (asm volatile statements keep their order)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[rb][b][r] *= al[rb];
            asm volatile("" : "+a"(ot[rb][b]));
            __builtin_amdgcn_sched_barrier(0);  // one 16-register tile at a time: unfenced, hipcc reads all 256 AGPRs first and spills Q
          }
        asm volatile("s_nop 7" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NDB; ++b) {
        if (CARRY && b == NDB - 1) {
          vf_carry = vf[b % VD];
#pragma unroll
          for (int rb = 0; rb < RB32; ++rb) pf_carry[rb] = pf[rb];
        } else {
          if constexpr (!U2) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[0][b]) : "v"(vf[b % VD]), "v"(pf[0]));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[0][b]) : "v"(vf[b % VD]), "v"(pf[0]));
#pragma unroll
          for (int rb = 1; rb < RB32; ++rb) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[rb][b]) : "v"(vf[b % VD]), "v"(pf[rb]));
          if (b + VD < NDB) vf[b % VD] = rd_v(b + VD);
        }
        if (b == 0) write_partials();
        if constexpr ((OPT & DW4_1STAGE) == 0) {
          if (b < PPW) piece_w(false, j + 3, par ^ 1, b);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the partials are in LDS, the V fragment reads are done
    if constexpr ((OPT & (DW4_1STAGE | DW4_ABL_DMA | DW4_ABL_NOWAIT)) == 0) hgemm::wait_vmcnt<2 * PPW>();  // K(j+2) has landed
    DW4_BARRIER();
    skew();
  };
  if constexpr (U2) {  // T = N / 16 is a multiple of 4 (N % 64 == 0, launcher)
    for (int j = 0; j < T; j += 2) {
      step(j, 0);
      step(j + 1, 1);
    }
  } else {
    for (int j = 0; j < T; ++j) step(j, j & 1);
  }
  pv_finish();  // the last tile's last d block
  hgemm::wait_vmcnt<0>();  // the dead refills of the last tiles: nothing may land in the staging area below
  // ---- epilogue: row sums to LDS, O = O^T / l staged through LDS in two passes of 32 rows per wave
#pragma unroll
  for (int o = 0; o < RM; ++o) {
    float l_tot = l_run[o];
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
    l_tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
    l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    *reinterpret_cast<float*>(ax_w + o * 64) = l_tot;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  DW4_BARRIER();
  float inv[RB32];
#pragma unroll
  for (int rb = 0; rb < RB32; ++rb) inv[rb] = 1.0f / *reinterpret_cast<const float*>(ax_r + rb * 128);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  DW4_BARRIER();  // every wave has its row sums: the staging area may overwrite the exchange images
#undef DW4_BARRIER
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");  // last MFMA results -> v_accvgpr_read (hgemm_w4.cuh)
#pragma unroll
  for (int rb = 0; rb < RB32; ++rb)
#pragma unroll
    for (int b = 0; b < NDB; ++b) asm volatile("" : "+a"(ot[rb][b]));
  char* ob = smem + wave * (32 * G::OS);
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
  constexpr int LPR = G::DH / 8;  // 16-byte segments per row of this wave's column block
#pragma unroll
  for (int rb = 0; rb < RB32; ++rb) {
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[rb][b][rq * 4 + e] * inv[rb]);
        *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    half_t* og = O + head + (size_t)(q_row0 + rb * 32) * D + part * G::DH;
    for (int idx = lane_e; idx < 32 * LPR; idx += 64) {
      const int row = idx / LPR, c = idx % LPR;
      *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // wave-private staging rows: the reads of this pass precede the writes of the next
  }
}

template <int D, int OPT = 0, int KPF = 2, int VPF = 2>
int launch_dw4(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoDW4<D>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_dw4_kernel<D, OPT, KPF, VPF>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_dw4_kernel<D, OPT, KPF, VPF>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream, (const half_t*)q,
             (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
