// FlashAttention-2 forward, head dim 64: the sum-checked two-group kernel of flash_attn_m16x.cuh rebuilt on v_mfma_f32_32x32x16_f16
// (round 3, PROBE ONLY: instantiated in flash_attn_m16x_probe.hip, not in the product library).
// Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66.
//
// Why. tools/ubench/shadow16.hip (profiles/r03_mfma_shadow_ubench.log): the D = 64 kernel's unit of work -- the matrix work of two
// 16x16x32 MFMAs with its share of the softmax (4 v_exp_f32, 4 v_add_f32, 2 v_cvt_pk_f16_f32) -- costs 36-37 ns per SIMD on the 16x16x32
// instruction and 31-33 ns on ONE 32x32x16 MFMA of the same flops: the 16x16x32 form issues every ~20 cycles instead of 16 and leaves a
// shorter shadow. Everything else is flash_attn_m16x.cuh: 8 waves x 32 query rows, two groups of four waves one phase apart, K / V tiles of
// 128 keys fetched by LDS-DMA into a two-slot ring in the SAME swizzled images (GeoM16<64, 32, 128>), Q pre-scaled, S^T accumulators started
// at -m, exponentials without a maximum first, the row sums as the overflow check, cold path with the true maxima.
//
// Register-level dataflow on 32x32x16 (the lane conventions of flash_attn_v2.cuh): lane l = 32 hi + l31 owns query row l31.
//   S^T block kb (32 keys x 32 queries) = sum over 4 k-steps of  K[32 kb + l31][16 ks + 8 hi ..+7] (A)  x  Q[l31][16 ks + 8 hi ..+7] (B);
//     accumulator register r of the lane holds key 32 kb + 8 (r >> 2) + 4 hi + (r & 3).
//   P^T k-step u of block kb = registers 8u .. 8u + 7 converted in order: k-slot 8 hi + e is key 32 kb + 16 u + 8 (e >> 2) + 4 hi + (e & 3),
//     and the V^T fragment (A: row d = 32 b + l31) is two transposing reads of 4 keys each, 8 key rows apart -- the P registers' key order.
//   The two key blocks of a pair are interleaved k-step by k-step, so an MFMA never follows the one that wrote its accumulator.
#pragma once
#include "flash_attn_m16x.cuh"

namespace fa2 {

template <int BC_, int PD = 4, int OX = 0>
__global__ __launch_bounds__(512, 2) void fa2_fwd_m32x_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                              const half_t* __restrict__ V, half_t* __restrict__ O,
                                                              int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoM16<64, 32, BC_>;
  constexpr int D = 64, NKB = G::BC / 32, NKS = D / 16, NDB = D / 32, NQK = NKB * NKS, NPV = NKB * 2 * NDB;
  constexpr int NDEF = NKB / 2, NOPT = NKB - NDEF;  // the first half of the key blocks is exponentiated in phase A, the second under the PV MFMAs
  constexpr int NPAIR = 8;                          // register pairs of one key block
  static_assert(NKB % 2 == 0 && NKB >= 4, "key blocks are processed in interleaved pairs");
  constexpr int A_STEPS = NQK - NOPT * NKS;         // QK^T steps before the first optimistic block is complete (pairs: NOPT blocks = NOPT * NKS steps)
  constexpr int A_RATE = (NOPT * NPAIR + (NQK - A_STEPS) - 1) / (NQK - A_STEPS);  // items per step in the second part of phase A
  constexpr int B_STEPS = NOPT * 2 * NDB;           // PV steps that use optimistic blocks only
  constexpr int B_RATE = (NDEF * NPAIR + B_STEPS - 1) / B_STEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2, widx = wave & 3;

  int head_i, qb_i;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {  // heads pinned to XCDs: a head's K/V stays in one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb_i = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb_i = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb_i * G::BR + wave * G::RPW;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // ---- LDS-DMA of the K (group 0) / V (group 1) tile images: exactly flash_attn_m16x.cuh's
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR;
  const int sw_src = grp == 0 ? G::swz_k(widx * G::RPP + lr) : G::swz_v(widx * G::RPP + lr);
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ sw_src) << 4);
  auto dma_piece = [&](int jt, int slot, int i) __attribute__((always_inline)) {
    const int piece = i * 4 + widx;
    hgemm::glds16_asm(src_h + (size_t)jt * G::TILE + piece * 1024, src_lane, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
  };

  const int T = N / G::BC;
  __builtin_assume(T > 0);
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  h8 qf[NKS];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }
  f16v ot[NDB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = 0.f, l_run = 0.f;
  f16v minit;
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  asm volatile("" : "+v"(minit));
  auto scale_q = [&]() __attribute__((always_inline)) {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = qf[ks] * sc;
      asm volatile("" : "+v"(qf[ks]));
    }
  };
  // group 0 needs K tile 0 (its own pieces) and its Q rows; group 1's V pieces and Q rows are not needed before the second barrier
  if (grp == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    scale_q();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    scale_q();
  }

  const int kbase = l31 * G::ROW + ((hi ^ G::swz_k(l31)) << 4);
  const int i16 = lane & 15, dh = (lane >> 4) & 1;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = v_row * G::ROW + ((((dh << 1) | ((i16 & 3) >> 1)) ^ G::swz_v(v_row)) << 4) + ((i16 & 1) << 3);

  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  for (int j = 0; j < T; ++j) {
    const int jn = j + 1 < T ? j + 1 : T - 1;
    const int kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    // QK^T step t: pair g = t / (2 NKS) works on blocks 2g, 2g + 1, alternating: kb = 2g + (t & 1), ks = (t % (2 NKS)) >> 1
    auto kb_of = [](int t) { return 2 * (t / (2 * NKS)) + (t & 1); };
    auto ks_of = [](int t) { return (t % (2 * NKS)) >> 1; };
    auto k_frag = [&](int t) __attribute__((always_inline)) {
      return *reinterpret_cast<const h8*>(smem + (kb_j ^ (ks_of(t) << 5)) + kb_of(t) * 32 * G::ROW);
    };
    // PV step idx: kb = idx / (2 NDB), u = (idx / NDB) & 1, b = idx % NDB
    auto v_frag = [&](int idx) __attribute__((always_inline)) {
      const int kb = idx / (2 * NDB), u = (idx / NDB) & 1, b = idx % NDB;
      const char* vp = smem + (vb_j ^ (b << 6)) + (32 * kb + 16 * u) * G::ROW;
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
    };
    f16v s[NKB];
    h8 pf[NKB][2];
    float psum = 0.f;
    // item i of key block kb: registers 2i, 2i + 1 -> k-slots 2 (i & 3), + 1 of P^T step i >> 2
    auto exp_item = [&](int kb, int i, float& acc) __attribute__((always_inline)) {
      const float a0 = __builtin_amdgcn_exp2f(s[kb][2 * i]);
      const float a1 = __builtin_amdgcn_exp2f(s[kb][2 * i + 1]);
      acc += a0 + a1;
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      asm volatile("" ::"v"(a), "v"(acc));  // keeps the item in the step it was written in (flash_attn_m16x.cuh)
      const int u = i >> 2, e = 2 * (i & 3);
      pf[kb][u][e] = a[0], pf[kb][u][e + 1] = a[1];
    };

    // ================= phase A: S^T = K Q^T; the optimistic blocks are exponentiated behind the MFMAs of the later pairs
    if constexpr ((OX & M16X_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
      constexpr int DSTEP = NQK / G::PPW;
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        const int kb = kb_of(t), ks = ks_of(t);
        if (ks == 0) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[0], minit, 0, 0, 0);  // chain starts at -m
        else s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[ks], s[kb], 0, 0, 0);
        cln_mfma_keep(s[kb], kf[t % PD], qf[ks]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
        if ((t % DSTEP) == DSTEP - 1) dma_piece(jn, (j + 1) & 1, t / DSTEP);
        if (t >= A_STEPS) {
#pragma unroll
          for (int it = (t - A_STEPS) * A_RATE; it < (t - A_STEPS + 1) * A_RATE && it < NOPT * NPAIR; ++it) exp_item(it / NPAIR, it % NPAIR, psum);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      // ---- the check: partial sums of the optimistic blocks, raw scores of the deferred ones
      bool bad = !(psum <= 32768.0f);
      float mx = s[NOPT][0];
#pragma unroll
      for (int kb = NOPT; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
      bad |= mx > 14.0f;
      const bool first = j == 0;  // tile 0 has no reference yet: it adopts its true maximum
      if (first || __builtin_amdgcn_ballot_w64(bad) != 0) {
        // ---- cold path: true row maximum (the two lanes of a query exchange theirs), standard rescale, the optimistic blocks again
        float mxa = s[0][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mxa = fmaxf(mxa, s[kb][r]);
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxa), __float_as_uint(mxa), false, false);
        const float d = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));  // relative to the running reference
        const float delta = first ? d : fmaxf(d, 0.f);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
        l_run *= alpha;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) minit[r] = -m_run;
        asm volatile("" : "+v"(minit));
#pragma unroll
        for (int b = 0; b < NDB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[b][r] *= alpha;
        psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NOPT; ++kb)
#pragma unroll
          for (int it = 0; it < NPAIR; ++it) exp_item(kb, it, psum);
      }
      l_run += psum;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B: O^T += V^T P^T; the deferred key blocks are exponentiated under its first MFMAs
    if constexpr ((OX & M16X_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
    float psum_d = 0.f;
    h8 vf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < NPV; ++idx) {
      const int kb = idx / (2 * NDB), u = (idx / NDB) & 1, b = idx % NDB;
      ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx % PD], pf[kb][u], ot[b], 0, 0, 0);
      cln_mfma_keep(ot[b], vf[idx % PD], pf[kb][u]);
      if (idx + PD < NPV) vf[idx % PD] = v_frag(idx + PD);
#pragma unroll
      for (int it = idx * B_RATE; it < (idx + 1) * B_RATE && it < NDEF * NPAIR; ++it) exp_item(NOPT + it / NPAIR, it % NPAIR, psum_d);
      __builtin_amdgcn_sched_barrier(0);
    }
    l_run += psum_d;
    hgemm::wait_vmcnt<0>();  // own DMA pieces of tile j+1 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows). Lane (query l31) holds d = 32 b + 8 rq + 4 hi .. + 3.
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
  char* ob = smem + wave * (G::RPW * G::OS);
  float l_tot;
  {
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
  }
  const float inv = 1.0f / l_tot;
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
    }
  constexpr int LPR = D / 8;
  half_t* og = O + head + (size_t)q_row0 * D;
#pragma unroll 4
  for (int it = 0; it < (G::RPW * LPR) / 64; ++it) {
    const int idx = it * 64 + lane_e;
    const int row = idx / LPR, c = idx % LPR;
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
  }
}

template <int BC_, int PD = 4, int OX = 0>
int launch_m32x(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoM16<64, 32, BC_>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_m32x_kernel<BC_, PD, OX>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf(64.0f);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_m32x_kernel<BC_, PD, OX>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream, (const half_t*)q, (const half_t*)k,
             (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
