// FlashAttention-2 forward for head dims 640 / 768, ONE workgroup barrier per 16-key tile (round 5). Same decomposition as flash_attn_dw4.cuh -- four
// waves, one per SIMD, each with all 64 rows of a quarter of d, O^T in AGPRs, softmax once per row by the row's owner wave -- but the two
// exchanges of a tile (partial S^T -> owners, P -> everybody) no longer cost a barrier each: both exchange images are DOUBLE-BUFFERED (which fits
// the 160 KiB only at these head dims: rings 80 / 96 KiB + 36.5 KiB; at D = 1024 the rings alone are 128 KiB) and the pipeline is skewed by one more
// tile, so that every step between two barriers carries, for three different tiles,
//     O^T += V^T P^T of tile j-1   |   softmax of tile j (owners)   |   S^T partial of tile j+2
// and everything a step READS was written before the barrier that opened it:
//     step j reads   partials S(j) from SX[j & 1], P(j-1) / alpha(j-1) from PX / AX[(j-1) & 1], V(j-1) from V slot (j-1) & 1, K(j+2) from K slot j & 1
//     step j writes  partials S(j+1) (computed in step j-1, still in registers) to SX[(j+1) & 1], P(j) / alpha(j) to PX / AX[j & 1],
//                    and requests K(j+3) into K slot (j+1) & 1 and V(j) into V slot j & 1 (both slots were last read in step j-1);
//                    the requests are waited for before the closing barrier.
// Against the two-barrier kernel: half the barriers, and the LDS round trips of the two exchanges sit beside a whole step's MFMAs instead of at the
// head of a phase each. [1,16,4096,D]: profiles/r05_fa_dw4b_probe.log. The running maximum, the rescale path and the arithmetic are those of
// flash_attn_dw4.cuh: results are bit-identical to it.
#pragma once
#include "flash_attn_dw4.cuh"

namespace fa2 {

template <int D>
struct GeoDW4B {
  static_assert(D == 640 || D == 768, "head dims 640 / 768 (the double-buffered exchange images do not fit beside the D = 1024 rings)");
  static constexpr int NSP = 4, DH = D / 4, BC = 16, NW = 4, BR = 64, NT = 256;
  static constexpr int ROW = D * 2, TILE = BC * ROW, NP = TILE / 1024, PPW = NP / NW;
  static_assert(NP % NW == 0, "every wave carries the same number of pieces");
  static constexpr int RING = 4 * TILE;            // K slot 0, K slot 1, V slot 0, V slot 1
  static constexpr int SXB = NW * 4096;            // one partial-S^T image
  static constexpr int SX = RING;                  // two of them
  static constexpr int PX = SX + 2 * SXB;          // two P images (2 KiB each)
  static constexpr int AX = PX + 2 * 2048;         // two alpha vectors (256 B each)
  static constexpr int MAIN = AX + 2 * 256;
  static constexpr int OS = DH * 2 + 16, EPI = NW * 32 * OS;
  static constexpr int LDS_BYTES = MAIN > EPI ? MAIN : EPI;
  static constexpr int NKS = DH / 32, NDB = DH / 32, CPP = DH / 8;
  static_assert(LDS_BYTES <= 160 * 1024 && (ROW / 16) % 16 == 0, "LDS / swizzle range");
};

template <int D, int OPT = 0, int KPF = 2, int VPF = 2>
__global__ __launch_bounds__(256, 1) void fa2_fwd_dw4b_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                              const half_t* __restrict__ V, half_t* __restrict__ O, int N,
                                                              int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoDW4B<D>;
  constexpr int NKS = G::NKS, NDB = G::NDB, PPW = G::PPW;
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

  // ---- LDS-DMA (flash_attn_dw4.cuh): piece i of a tile request = KiB (i * 4 + wave) of the lane-linear image, source chunk swizzled
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
  // piece i of tile t into ring slot `slot` (0 / 1) of the K (is_v = false) or V ring; M0 walks: piece 0 sets it, every piece leaves it 4 KiB further
  auto piece_w = [&](bool is_v, int t, int slot, int i) __attribute__((always_inline)) {
    if constexpr ((OPT & DW4_ABL_DMA) != 0) return;
    const char* src = (is_v ? Vh : Kh) + (size_t)clampt(t) * G::TILE;
    const unsigned voff = is_v ? v_voff[i] : k_voff[i];
    if (i == 0) {
      const unsigned dst = lds0 + (is_v ? 2 * G::TILE : 0) + slot * G::TILE + (unsigned)wave * 1024u;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0", "scc");
    } else {
      asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" ::"v"(voff), "s"(src) : "memory", "m0", "scc");
    }
  };
  auto req_tile = [&](bool is_v, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece_w(is_v, t, slot, i);
  };

  // ---- Q fragments (B operand of S^T = K Q^T on 16x16x32): query 16*rb + i16, d = part*DH + 32*ks + 8*g4 .. +7
  h8 qf[4][NKS];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    const half_t* qp = Q + head + (size_t)(q_row0 + rb * 16 + i16) * D + part * G::DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[rb][ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  f16v ot[2][NDB];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[rb][b][r] = 0.f;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int b = 0; b < NDB; ++b) asm volatile("" : "+a"(ot[rb][b]));  // zero-fill done HERE (inline-asm MFMAs are invisible to the hazard pass)
  asm volatile("s_nop 7");
  float m_run = -1.0e30f, l_run = 0.f;

  req_tile(false, 0, 0);
  req_tile(true, 0, 0);
  req_tile(false, 1, 1);
  req_tile(true, 0, 1);  // filler for V slot 1: tile "-1" is multiplied by P = 0 in step 0 and must hold finite values
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: Q and the first tiles
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[rb][ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- fragment addresses (flash_attn_dw4.cuh)
  const int v_row = 4 * hi + (i16 >> 2);
  const int v_w = ((lane >> 4) & 1) * 2 + ((i16 & 3) >> 1);
  unsigned koff[NKS], voffs[NDB];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) koff[ks] = (unsigned)(i16 * G::ROW + (((part * G::CPP + 4 * ks + g4) ^ i16) << 4));
#pragma unroll
  for (int b = 0; b < NDB; ++b)
    voffs[b] = (unsigned)(2 * G::TILE + v_row * G::ROW + (((part * G::CPP + 4 * b + v_w) ^ ((v_row & 3) << 2)) << 4) + ((i16 & 1) << 3));
  const int sx_lane = i16 * 64 + ((g4 ^ ((i16 >> 1) & 3)) << 4);
  char* sx_w = smem + G::SX + wave * 4096 + sx_lane;          // + buf * SXB + rb * 1024
  const char* sx_r = smem + G::SX + wave * 1024 + sx_lane;    // + buf * SXB + p * 4096
  char* px_w = smem + G::PX + g4 * 512 + (wave * 16 + i16) * 8;   // + buf * 2048
  char* ax_w = smem + G::AX + (wave * 16 + i16) * 4;              // + buf * 256
  const char* px_r = smem + G::PX + hi * 512 + l31 * 8;           // + buf * 2048 + rb * 256 (+ 1024: second key chunk)
  const char* ax_r = smem + G::AX + l31 * 4;                      // + buf * 256 + rb * 128

  f4 s[4];
  auto qk_group = [&](int ks, const h8& kfr) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {  // (s_nop 1 / tied accumulators: see flash_attn_dw4.cuh)
      if (ks == 0 && rb == 0) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s[rb]) : "v"(kfr), "v"(qf[rb][0]));
      else if (ks == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s[rb]) : "v"(kfr), "v"(qf[rb][0]));
      else if (rb == 0) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s[rb]) : "v"(kfr), "v"(qf[rb][ks]));
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s[rb]) : "v"(kfr), "v"(qf[rb][ks]));
    }
  };
  // S^T partial over this wave's quarter of d from K slot `slot`; hook(ks) runs behind the four MFMAs of k-step ks
  auto qk_tile = [&](int slot, auto&& hook) __attribute__((always_inline)) {
    const char* kb = smem + slot * G::TILE;
    constexpr int KD = KPF < NKS ? KPF : NKS;
    h8 kf[KD];
#pragma unroll
    for (int i = 0; i < KD; ++i) kf[i] = *reinterpret_cast<const h8*>(kb + koff[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qk_group(ks, kf[ks % KD]);
      if (ks + KD < NKS) kf[ks % KD] = *reinterpret_cast<const h8*>(kb + koff[ks + KD]);
      __builtin_amdgcn_sched_barrier(0);
      hook(ks);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto write_partials = [&](int buf) __attribute__((always_inline)) {
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the S^T MFMAs (inline asm) were issued before the last barrier: results are in the registers
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) *reinterpret_cast<f4*>(sx_w + buf * G::SXB + rb * 1024) = s[rb];
  };
#define DW4B_BARRIER()             \
  do {                             \
    __builtin_amdgcn_s_barrier();  \
    asm volatile("" ::: "memory"); \
  } while (0)

  // ---- prologue: S(0) published, S(1) in registers, K(2) landed, P(-1) = 0 / alpha(-1) = 1
  qk_tile(0, [&](int) {});
  __builtin_amdgcn_s_waitcnt(0xC07F);
  write_partials(0);
  *reinterpret_cast<h4*>(px_w + 2048) = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  *reinterpret_cast<float*>(ax_w + 256) = 1.f;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  DW4B_BARRIER();  // every wave is past its reads of K slot 0
  req_tile(false, 2, 0);
  qk_tile(1, [&](int) {});
  __builtin_amdgcn_s_waitcnt(0xC07F);
  hgemm::wait_vmcnt<0>();
  DW4B_BARRIER();

  for (int j = 0; j <= T; ++j) {  // step j: PV of tile j-1, softmax of tile j, S^T partial of tile j+2 (the last step's softmax / S^T are never used)
    const int cur = j & 1, prv = cur ^ 1;
    write_partials(prv);  // S(j+1): (j+1) & 1
    f4 ap[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ap[p] = *reinterpret_cast<const f4*>(sx_r + cur * G::SXB + p * 4096);
    float al[2];
    h8 pf[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      al[rb] = *reinterpret_cast<const float*>(ax_r + prv * 256 + rb * 128);
      pf[rb] = h8_cat(*reinterpret_cast<const h4*>(px_r + prv * 2048 + rb * 256), *reinterpret_cast<const h4*>(px_r + prv * 2048 + rb * 256 + 1024));
    }
    const char* vb = smem + prv * G::TILE;  // V slot (j-1) & 1 (voffs carry the 2 * TILE of the V ring)
    constexpr int VD = VPF < NDB ? VPF : NDB;
    auto rd_v = [&](int b) __attribute__((always_inline)) -> h8 {
      const char* vp = vb + voffs[b];
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
    };
    h8 vf[VD];
#pragma unroll
    for (int i = 0; i < VD; ++i) vf[i] = rd_v(i);
    asm volatile("" : "+v"(pf[0]));  // (P fragments assembled by v_mov: pinned and padded before any MFMA reads them, flash_attn_dw4.cuh)
    asm volatile("" : "+v"(pf[1]));
    asm volatile("s_nop 1" ::: "memory");
    if constexpr ((OPT & DW4_1STAGE) != 0) {  // `stages = 1`: the step's two tile requests in ONE burst, waited for right here
      req_tile(false, j + 3, prv);
      req_tile(true, j, cur);
      hgemm::wait_vmcnt<0>();
    }
    if (__builtin_amdgcn_ballot_w64(al[0] != 1.f || al[1] != 1.f) != 0) {  // rare: a row's maximum grew by more than 2^8 (flash_attn_dw4.cuh)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int b = 0; b < NDB; ++b) {
          asm volatile("" : "+a"(ot[rb][b]));
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[rb][b][r] *= al[rb];
          asm volatile("" : "+a"(ot[rb][b]));
          __builtin_amdgcn_sched_barrier(0);
        }
      asm volatile("s_nop 7" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    // softmax of tile j in sections behind the PV MFMA groups (owners: rows 16*wave + i16, keys 4*g4 .. +3)
    f4 a;
    float p4[4];
    float alpha = 1.f, mx = 0.f;
    auto softmax_section = [&](int sec) __attribute__((always_inline)) {
      if constexpr ((OPT & DW4_ABL_SOFTMAX) != 0) return;
      if (sec == 0) {
        a = (ap[0] + ap[1]) + (ap[2] + ap[3]);
        mx = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
      } else if (sec == 1) {
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
        const float mxs = mx * scale_log2e;
        bool grow;
        if constexpr ((OPT & DW4_NO_DEFER) != 0) grow = mxs > m_run;
        else grow = (mxs - m_run) > 8.0f;
        grow = grow && j < T;  // the last step's tile does not exist: it must not touch the running maximum / sum
        const float m_new = grow ? mxs : m_run;
        alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;
        m_run = m_new;
        l_run *= alpha;
        if (j == 0) alpha = 1.f;  // O and l are still zero: nothing to rescale
      } else if (sec == 2) {
        const float nm = -m_run;
#pragma unroll
        for (int e = 0; e < 4; ++e) p4[e] = __builtin_amdgcn_exp2f(fmaf(a[e], scale_log2e, nm));
      } else if (sec == 3) {
        if (j < T) l_run += (p4[0] + p4[1]) + (p4[2] + p4[3]);  // (the last step's tile does not exist)
        const h2 lo = __builtin_convertvector(f2{p4[0], p4[1]}, h2), hh = __builtin_convertvector(f2{p4[2], p4[3]}, h2);
        *reinterpret_cast<h4*>(px_w + cur * 2048) = h4{lo[0], lo[1], hh[0], hh[1]};
        *reinterpret_cast<float*>(ax_w + cur * 256) = alpha;
      }
    };
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[0][b]) : "v"(vf[b % VD]), "v"(pf[0]));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[1][b]) : "v"(vf[b % VD]), "v"(pf[1]));
      if (b + VD < NDB) vf[b % VD] = rd_v(b + VD);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((OPT & DW4_1STAGE) == 0) {
        if (b < PPW) piece_w(false, j + 3, prv, b);  // K(j+3) -> K slot (j+1) & 1
      }
      if (b < 4) softmax_section(b);
      __builtin_amdgcn_sched_barrier(0);
    }
    qk_tile(cur, [&](int ks) __attribute__((always_inline)) {  // K(j+2) in K slot j & 1
      if constexpr ((OPT & DW4_1STAGE) == 0) {
        if (ks < PPW) piece_w(true, j, cur, ks);  // V(j) -> V slot j & 1
      }
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this step's LDS writes are done, its fragment reads have returned
    hgemm::wait_vmcnt<0>();              // K(j+3) and V(j) have landed
    DW4B_BARRIER();
  }
#undef DW4B_BARRIER
  hgemm::wait_vmcnt<0>();  // the dead refills of the last tiles: nothing may land in the staging area below
  // ---- epilogue: row sums to LDS, O = O^T / l staged through LDS in two passes of 32 rows per wave
#pragma unroll
  for (int o = 0; o < 1; ++o) {
    float l_tot = l_run;
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
    l_tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
    l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    *reinterpret_cast<float*>(ax_w + o * 64) = l_tot;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float inv[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) inv[rb] = 1.0f / *reinterpret_cast<const float*>(ax_r + rb * 128);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");  // every wave has its row sums: the staging area may overwrite the exchange images
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");  // last MFMA results -> v_accvgpr_read (hgemm_w4.cuh)
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int b = 0; b < NDB; ++b) asm volatile("" : "+a"(ot[rb][b]));
  char* ob = smem + wave * (32 * G::OS);
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
  constexpr int LPR = G::DH / 8;  // 16-byte segments per row of this wave's column block
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
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
int launch_dw4b(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoDW4B<D>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_dw4b_kernel<D, OPT, KPF, VPF>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_dw4b_kernel<D, OPT, KPF, VPF>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream, (const half_t*)q,
             (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
