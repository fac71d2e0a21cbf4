// FlashAttention-2 forward, head dim 512 (config C5): the d-split PAIR kernel of flash_attn_m16.cuh with the softmax done ONCE per row
// (round 6). Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:70, :529-647 (QK^T and PV tiled over d).
//
// What changes against fa2_fwd_m16_pair_kernel<PAIR = true>, and why. There a pair of waves owns 32 query rows and splits d for BOTH
// products: each wave forms a PARTIAL S^T over its 256 columns of d for all 32 rows, the two partials (4 KiB of fp32 per wave) cross
// through LDS, and then BOTH waves run the same softmax on the same 32 x 32 scores (bit-identical by construction, and redundant: every
// exponential of the head is computed twice). On gfx950 plain VALU time and matrix time of a SIMD add,
// so the duplicate softmax is paid in full: per tile and SIMD 2048 matrix clocks + ~900 VALU clocks, of which half are the duplicate.
// Here the pair splits the ROWS for S^T = K Q^T and the softmax, and d only for O^T += V^T P^T:
//   * wave `part` (0 / 1) of a pair owns query rows 16 part .. 16 part + 15 of the pair's 32: its Q fragments cover the FULL d (16 k-steps:
//     the same 64 registers as 32 rows x 256 columns), its 32 QK^T MFMAs per tile give the complete scores of its rows -- no partial, no
//     fp32 exchange; the K fragment reads double (every wave walks whole K rows; LDS has the room: 256 B/clk, MI355X_MICROARCH.md LDS);
//   * the softmax runs on 8 scores per lane instead of 16 (8 exponentials, 4 conversions, one row-max reduction), m and l live in the
//     owner wave only;
//   * what crosses LDS is P as fp16 (1 KiB per wave, lane-linear: the owner's registers ARE the partner's B operand) and one rescale
//     factor per row (the deferred running max moves only when a row grows by more than 2^8: the factor is 1.0 and a ballot skips the
//     multiply); both waves read both halves back, so no register select depends on `part`;
//   * O^T += V^T P^T is unchanged: each wave accumulates its 256 output columns for all 32 rows (128 registers), every V^T fragment
//     feeds two MFMAs; 1 / l crosses once, in the epilogue.
// Same LDS image, LDS-DMA ring, two-group phase offset and two workgroup barriers per 32-key tile as the pair kernel; scores scaled in
// fp32 (no fp16 rounding of Q * log2(e) / sqrt(d)). The arithmetic per row is the pair kernel's (same products, same key order, fp32 sums in
// the same order within a lane), but the QK^T accumulation order over d differs (one chain of 16 k-steps instead of two chains of 8 added in
// fp32): results agree to fp32 rounding, not bit for bit.
#pragma once
#include <type_traits>

#include "flash_attn_m16.cuh"

namespace fa2 {

struct GeoPair2 {
  static constexpr int D = 512, DH = 256, BC = 32, NW = 8, BR = 128, NT = 512;
  static constexpr int ROW = D * 2, TILE = BC * ROW, STAGE = 2 * TILE, RING = 2 * STAGE;
  // exchange: pair rg (4 per workgroup) owns 4 KiB: P of part 0 (1 KiB, lane-linear h8) | P of part 1 | rescale factors [2][16] fp32
  static constexpr int SX = RING, SXP = 4096, PX_A = 2048;
  static constexpr int OS = DH * 2 + 16, EPI = NW * 32 * OS;
  static constexpr int LX = SX + 4 * SXP;  // 1 / l of the epilogue: [4 pairs][2][16] fp32, beyond the O staging rows (EPI < LX)
  static constexpr int LDS_BYTES = LX + 4 * 128;
  static constexpr int RPP = 1024 / ROW, CPR = ROW / 16, PPW = TILE / 1024 / 4;
  static constexpr int NKB = BC / 16, NKS = D / 32, NDB = DH / 16, NQK = NKB * NKS;
  static_assert(EPI <= LX && LDS_BYTES <= 160 * 1024, "LDS");
  static __device__ __forceinline__ int swz_k(int row) { return row & 15; }
  static __device__ __forceinline__ int swz_v(int row) { return (row & 15) << 1; }
};

enum : int {
  PAIR2_ONE_STAGE = 1,  // the `stages = 1` form: a wave requests all its pieces of tile j + 1 in one burst at the top of phase A and waits for them there
  // fragment addresses WITHOUT VALU work inside the loop: the ring laid out [K slot 0 | K slot 1 | V slot 0 | V slot 1] (a slot is 32 KiB apart from its twin, so slot,
  // key block and the 256-byte step all fit the 16-bit offset field of ds_read), two tiles per loop iteration (the slot is a compile-time constant), and the lane's
  // swizzled bases (4 for K: base ^ (ks & 3) << 6; 16 for V^T: base ^ db << 5) computed once, in front of the loop, and pinned in registers. 18 XORs and the slot
  // arithmetic leave every tile -- plain VALU time is never hidden under MFMAs on gfx950
  PAIR2_HOIST = 2,
};

// DREAL != 0: the tensors have DREAL < 512 columns (320 / 384). The LDS geometry stays that of D = 512 (1024-byte rows; columns [p DREAL/2, (p + 1) DREAL/2) of a
// row at the start of its 512-byte half p, the rest of a half never read -- the image of flash_attn_dsplit.cuh's PAD form); every loop runs over the REAL head dim:
// DREAL / 32 k-steps for QK^T (10 / 12), DREAL / 32 output blocks per wave for PV (10 / 12): no MFMA multiplies padding.
template <int PDK = 4, int PDV = 2, int OPT = 0, int DREAL = 0>
__global__ __launch_bounds__(512, 2) void fa2_fwd_pair2_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                               const half_t* __restrict__ V, half_t* __restrict__ O,
                                                               int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoPair2;
  constexpr bool PAD = DREAL != 0;
  static_assert(!PAD || (DREAL % 64 == 0 && DREAL > 256 && DREAL < 512), "head dims 320 / 384 ride on the D = 512 geometry");
  constexpr int D = PAD ? DREAL : G::D, DH = D / 2;         // columns per row in memory; columns of a wave's PV half
  constexpr int NKS = D / 32, KH = DH / 32, NDB = DH / 16, NQK = G::NKB * NKS;  // k-steps of QK^T (KH of them per 512-byte half of a row), output blocks per wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 1024: fragment addresses XOR bits 5 .. 8 into (LDS address of smem + offset)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
  const int grp = wave >> 2, widx = wave & 3;
  const int part = widx & 1, rg = grp * 2 + (widx >> 1);

  int head_i, qb_i;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {  // heads pinned to XCDs: a head's K / V stays in one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb_i = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb_i = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb_i * G::BR + rg * 32;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // ---- LDS-DMA (the pair kernel's image): a piece = one 1-KiB row; group 0 fills the K tile, group 1 the V tile; the swizzle is applied to the SOURCE chunk
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = widx * G::RPP + lr;
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ (grp == 0 ? G::swz_k(rlow) : G::swz_v(rlow))) << 4);
  auto dma_piece = [&](int jt, int slot, int i) __attribute__((always_inline)) {
    const int piece = i * 4 + widx;
    unsigned voff = src_lane ^ (unsigned)((grp == 0 ? G::swz_k(4 * i * G::RPP) : G::swz_v(4 * i * G::RPP)) << 4);
    const char* s;
    if constexpr (PAD) {  // voff >> 4 = the logical chunk X this lane's LDS position holds: half p = X >> 5, chunk cc = X & 31 of it; real if cc < DH / 8 (else never read)
      const unsigned X = voff >> 4, pp = X >> 5, cc = X & 31;
      voff = cc < (unsigned)(DH / 8) ? (pp * (unsigned)(DH / 8) + cc) << 4 : 0u;
      s = src_h + (size_t)jt * (G::BC * D * 2) + piece * (D * 2);
    } else {
      s = src_h + (size_t)jt * G::TILE + piece * 1024;
    }
    if constexpr ((OPT & PAIR2_HOIST) != 0) hgemm::glds16_asm(s, voff, lds0 + grp * G::STAGE + slot * G::TILE + piece * 1024);
    else hgemm::glds16_asm(s, voff, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
  };

  // ---- Q fragments: this wave's 16 rows over the whole head dim (B operand of S^T = K Q^T: query i16, d = 32 ks + 8 g4 .. + 7), as loaded
  h8 qf[NKS];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + part * 16 + i16) * D + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  f4 ot[NDB][2];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) ot[b][qb] = f4{0.f, 0.f, 0.f, 0.f};
  float m_run = 0.f, l_run = 0.f;

  const int T = N / G::BC;
  __builtin_assume(T > 0);
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: also retires the Q loads
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // LDS byte addresses, the symbol's address folded in once (common.h lds_ld). K fragment (kb, ks): row 16 kb + i16, logical chunk 4 ks + g4;
  // V^T fragment (db) of this wave's d half: rows 4 g4 + (i16 >> 2) and + 16, logical chunk 32 part + 2 db + ((i16 & 3) >> 1), 8-byte half i16 & 1
  const unsigned kbase = lds0 + i16 * G::ROW + ((g4 ^ G::swz_k(i16)) << 4);
  const int v_row = 4 * g4 + (i16 >> 2);
  const unsigned vbase = lds0 + v_row * G::ROW + (((((i16 & 3) >> 1)) ^ G::swz_v(v_row)) << 4) + ((i16 & 1) << 3) + part * 512;
  const unsigned px = lds0 + G::SX + rg * G::SXP;
  const unsigned p_w = px + part * 1024 + lane * 16, p_r = px + lane * 16;               // + 1024: the other half of the pair's rows
  const unsigned a_w = px + G::PX_A + part * 64 + i16 * 4, a_r = px + G::PX_A + i16 * 4;  // + 64

  constexpr bool HOIST = (OPT & PAIR2_HOIST) != 0;
  unsigned kx[4], vx[NDB];  // HOIST: the swizzled bases, pinned (an opaque asm: hipcc would otherwise re-derive them inside the loop to save registers)
  if constexpr (HOIST) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      kx[c] = kbase ^ (unsigned)(c << 6);
      asm volatile("" : "+v"(kx[c]));
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      vx[db] = (vbase + G::STAGE) ^ (unsigned)(db << 5);  // V slots behind the two K slots
      asm volatile("" : "+v"(vx[db]));
    }
  }

  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  auto tile = [&](int j, auto slot_c) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_c)::value;  // HOIST: the ring slot of tile j as a compile-time constant (-1: taken from j)
    const int jn = j + 1 < T ? j + 1 : T - 1;
    const int nslot = HOIST ? 1 - SLOT : (j + 1) & 1;  // ring slot of tile j + 1
    const unsigned kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    auto k_frag = [&](int t) __attribute__((always_inline)) {  // t = 2 ks + kb: the two key blocks alternate, a dependent MFMA sits two behind
      const int kb = t & 1;
      const int ks = (t >> 1) % KH + 8 * ((t >> 1) / KH);  // position in the LDS row, in 64-byte steps: KH real steps at the start of each 512-byte half
      if constexpr (HOIST) return lds_ld<h8>(kx[ks & 3] + (unsigned)(SLOT * G::TILE + (ks >> 2) * 256 + kb * 16 * G::ROW));
      else return lds_ld<h8>((kb_j ^ (unsigned)((ks & 3) << 6)) + (ks >> 2) * 256 + kb * 16 * G::ROW);
    };
    auto v_frag = [&](int db) __attribute__((always_inline)) {
      if constexpr (HOIST) {
        const unsigned vp = vx[db] + (unsigned)(SLOT * G::TILE);
        return h8_cat(lds_read_tr16_at(vp), lds_read_tr16_at(vp + 16 * G::ROW));
      } else {
        const unsigned vp = vb_j ^ (unsigned)(db << 5);
        return h8_cat(lds_read_tr16_at(vp), lds_read_tr16_at(vp + 16 * G::ROW));
      }
    };
    // ================= phase A: S^T of this wave's 16 rows (complete), softmax, P and the rescale factors published
    if constexpr ((OPT & PAIR2_ONE_STAGE) != 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G::PPW; ++i) dma_piece(jn, nslot, i);
      hgemm::wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
    f4 s[2];
    {
      h8 kf[PDK];
#pragma unroll
      for (int i = 0; i < PDK; ++i) kf[i] = k_frag(i);
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        const int ks = t >> 1, kb = t & 1;
        if (ks == 0) s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PDK], qf[0], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PDK], qf[ks], s[kb], 0, 0, 0);
        cln_mfma_keep(s[kb], kf[t % PDK], qf[ks]);  // destination disjoint from the operands (common.h)
        if (t + PDK < NQK) kf[t % PDK] = k_frag(t + PDK);
        if constexpr ((OPT & PAIR2_ONE_STAGE) == 0) {  // piece i behind step (i + 1) NQK / PPW - 1: spread evenly over the phase (NQK = 32 / 24 / 20)
#pragma unroll
          for (int i = 0; i < G::PPW; ++i)
            if (t == ((i + 1) * NQK) / G::PPW - 1) dma_piece(jn, nslot, i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kb][r] = fmaf(s[kb][r], scale_log2e, -m_run);  // log2 domain, relative to the running reference
    float alpha = 1.0f;
    {
      float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
      mx = fmaxf(fmaxf(mx, s[0][3]), s[1][0]);
      mx = fmaxf(fmaxf(mx, s[1][1]), s[1][2]);
      mx = fmaxf(mx, s[1][3]);
      const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
      const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      const float d = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));  // the row's maximum: the same value in its four lanes
      const bool first = j == 0;  // tile 0 adopts its true maximum
      if (first || __builtin_amdgcn_ballot_w64(d > 8.0f) != 0) {
        const float delta = first ? d : fmaxf(d, 0.f);
        alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);  // (tile 0: l and O^T are zero, nothing to rescale)
        m_run += delta;
        l_run *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kb][r] -= delta;
      }
    }
    {
      h8 pf;  // k-slot 8 g4 + e <-> key 16 (e >> 2) + 4 g4 + (e & 3): this lane's registers are the B operand of the 32-key PV step as they are
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int kb = e >> 2, r = e & 3;
        const float a0 = __builtin_amdgcn_exp2f(s[kb][r]);
        const float a1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
        psum += a0 + a1;
        const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
        pf[e] = a[0], pf[e + 1] = a[1];
      }
      l_run += psum;
      lds_st<h8>(p_w, pf);
      lds_st<float>(a_w, alpha);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): P and the factors are in LDS before the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B: O^T[this wave's 256 columns] += V^T P^T for the pair's 32 rows
    h8 pf[2];
    float al[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      pf[qb] = lds_ld<h8>(p_r + qb * 1024);
      al[qb] = lds_ld<float>(a_r + qb * 64);
    }
    h8 vf[PDV];
#pragma unroll
    for (int i = 0; i < PDV; ++i) vf[i] = v_frag(i);
    __builtin_amdgcn_sched_barrier(0);
    if (__builtin_amdgcn_ballot_w64(al[0] != 1.0f || al[1] != 1.0f) != 0) {  // cold: a row of the pair moved its reference
#pragma unroll
      for (int b = 0; b < NDB; ++b)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[b][qb][r] *= al[qb];
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        ot[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[db % PDV], pf[qb], ot[db][qb], 0, 0, 0);
        cln_mfma_keep(ot[db][qb], vf[db % PDV], pf[qb]);
      }
      if (db + PDV < NDB) vf[db % PDV] = v_frag(db + PDV);
      __builtin_amdgcn_sched_barrier(0);
    }
    hgemm::wait_vmcnt<0>();  // own DMA pieces of tile j + 1 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  if constexpr (HOIST) {  // N is a multiple of 128: an even number of 32-key tiles
    for (int j = 0; j < T; j += 2) {
      tile(j, std::integral_constant<int, 0>{});
      tile(j + 1, std::integral_constant<int, 1>{});
    }
  } else {
    for (int j = 0; j < T; ++j) tile(j, std::integral_constant<int, -1>{});
  }
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: 1 / l of the owner's rows crosses LDS once; then this wave's 256 output columns of the pair's 32 rows, staged through LDS
  const int lane_e = cln_fresh_lane(), i16_e = lane_e & 15, g4_e = lane_e >> 4;
  {
    float l_tot = l_run;
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
    l_tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
    l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    lds_st<float>(lds0 + G::LX + rg * 128 + part * 64 + i16_e * 4, 1.0f / l_tot);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  constexpr int OS = DH * 2 + 16;
  char* ob = smem + wave * (32 * OS);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float inv = lds_ld<float>(lds0 + G::LX + rg * 128 + qb * 64 + i16_e * 4);
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][qb][e] * inv);
      *reinterpret_cast<h4*>(ob + (qb * 16 + i16_e) * OS + (b * 16 + g4_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = DH / 8;
  half_t* og = O + head + (size_t)q_row0 * D + part * DH;
#pragma unroll 4
  for (int it = 0; it < (32 * LPR) / 64; ++it) {
    const int idx = it * 64 + lane_e;
    const int row = idx / LPR, c = idx % LPR;
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * OS + c * 16);
  }
}

template <int PDK = 4, int PDV = 2, int OPT = 0, int DREAL = 0>
int launch_pair2(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoPair2;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_pair2_kernel<PDK, PDV, OPT, DREAL>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)(DREAL ? DREAL : G::D));
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_pair2_kernel<PDK, PDV, OPT, DREAL>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
