// FlashAttention-2 forward, head dim 64, two-group ping-pong kernel with 64 query rows per wave.
// Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66 (split-Q, Q in registers, K/V through LDS).
//
// Why (profiles/r02_fa_energy_ablation.log): in the shipped D = 64 kernel (flash_attn_dsplit.cuh, 8 waves x 32 rows) every
// 32x32x16 MFMA consumes one fresh 1-KiB K or V fragment from LDS, and that fragment traffic is the kernel's largest
// single cost at the package power cap (without it: 891 -> 1260 TF). Here a wave owns TWO 32-row groups, so every
// fragment feeds two MFMAs (half the LDS bytes per flop) while the two 4-wave groups still run one phase apart (a
// SIMD always has one wave in a matrix phase beside its partner's softmax) -- which the one-wave-per-SIMD kernels
// (flash_attn_rb.cuh, flash_attn_w4.cuh) had to give up. The price is 512 query rows per workgroup: it needs >= 512 rows
// per CU to fill the chip, so it serves long sequences / many heads ([1,48,8192,64]: 1536 rows per CU), not config C4
// (256 rows per CU).
//   * 8 waves, wave w owns rows w*64 .. +63; KV tiles of 64 keys (so the two groups' scores fit the 256-register budget);
//   * phase A: S^T(g) = K Q(g)^T for both row groups on each K fragment (16 MFMAs), then row max / rescale decision and
//     the first half of the exponentials; phase B: O^T(g) += V^T P(g)^T (16 MFMAs) with the second half of the
//     exponentials between its halves (the split softmax of the 32-row kernel);
//   * Q pre-multiplied by log2(e)/sqrt(d), accumulators started at -m through the MFMA C operand (OPT_PRE), deferred
//     running max; K tiles fetched by group 0, V tiles by group 1 (LDS-DMA, 2 pieces per wave and tile), double-buffered.
// KVS = true (round 2, for shapes with only 256 rows per CU such as config C4): the workgroup owns 256 query rows and the
// two groups split the KEYS instead -- group 0 walks the first half of the KV tiles, group 1 the second half, each through
// its own double-buffered K+V ring (4 DMA pieces per wave and tile), still one phase apart. Wave (g, w) and wave (1-g, w)
// hold partial (O^T, m, l) of the same 64 rows; they are merged once at the end through LDS (each wave exports the 32-row
// group it does not finalise, imports its partner's half of the one it does): O = O_0 2^(m_0-m) + O_1 2^(m_1-m).
#pragma once
#include "flash_attn_dsplit.cuh"

namespace fa2 {

struct GeoSplit2 {
  static constexpr int D = 64, BC = 64, BCB = 2, NW = 8, BR = 64 * NW, NT = 512;
  static constexpr int ROW = D * 2, TILE = BC * ROW, STAGE = 2 * TILE, RING = 2 * STAGE;
  static constexpr int OS = D * 2 + 16, EPI = NW * 64 * OS;
  static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;
  // KVS: one ring per group; exchange area = per wave 34 floats per lane (32 O^T values, m, l of the exported row group)
  static constexpr int XCH_WAVE = 34 * 64 * 4, XCH = NW * XCH_WAVE;
  static constexpr int LDS_BYTES_KVS = 2 * RING > XCH ? 2 * RING : XCH;
  static constexpr int PPW = TILE / 1024 / 4;  // DMA pieces per wave per tile (4 waves fill one operand)
  static constexpr int RPP = 1024 / ROW, CPR = ROW / 16;
  static constexpr int NK = D / 16, NDB = D / 32, NQK = BCB * NK, NPV = 2 * BCB * NDB;
  // 128-byte rows: two rows span the 64 banks (flash_attn_dsplit.cuh GeoSplit)
  static __device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }
  static __device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 1) << 2; }
};

template <int PD = 4, bool KVS = false>
__global__ __launch_bounds__(512, 2) void fa2_fwd_dsplit2_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                                 const half_t* __restrict__ V, half_t* __restrict__ O,
                                                                 int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoSplit2;
  constexpr int D = G::D, BCB = G::BCB, NK = G::NK, NDB = G::NDB, NQK = G::NQK, NPV = G::NPV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2, widx = wave & 3;

  int head_i, qb;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {  // heads pinned to XCDs: a head's K/V stays in one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = KVS ? qb * (G::BR / 2) + widx * 64 : qb * G::BR + wave * 64;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // ---- LDS-DMA: wave widx of group 0 fills the 1-KiB pieces i*4 + widx of the K tile, group 1 those of the V tile
  // (lane-linear image, swizzle applied to the source chunk; flash_attn_dsplit.cuh)
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = widx * G::RPP + lr;
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (grp == 0 ? (unsigned)((lc ^ G::swz_k(rlow)) << 4) : (unsigned)((lc ^ G::swz_v(rlow)) << 4));
  const unsigned kmask = grp == 0 ? 0xFFu : 0u;
  // KVS: every wave fetches K pieces (i = 0, 1) and V pieces (i = 2, 3) of its own group's tile
  const char* src_k = reinterpret_cast<const char*>(K + head);
  const char* src_v = reinterpret_cast<const char*>(V + head);
  const unsigned src_lane_k = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_k(rlow)) << 4);
  const unsigned src_lane_v = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_v(rlow)) << 4);
  constexpr int PPW = KVS ? 2 * G::PPW : G::PPW;
  const int gring = KVS ? grp * G::RING : 0;  // byte offset of this group's ring
  auto dma_piece = [&](int jt, int slot, int i) __attribute__((always_inline)) {
    // K rows piece*RPP + lr: (row >> 1) & 7 = ((i*4*RPP >> 1) & 7) ^ ... -- the piece part of the row only touches
    // bits >= 3 of (row >> 1) when RPP = 8 (i*32 rows): swz_k(row) = swz_k(rlow) for every i, no per-piece term
    if constexpr (KVS) {
      const int op = i >> 1, piece = (i & 1) * 4 + widx;
      const char* s = (op ? src_v : src_k) + (size_t)jt * G::TILE + piece * 1024;
      hgemm::glds16_asm(s, op ? src_lane_v : src_lane_k, lds0 + gring + slot * G::STAGE + op * G::TILE + piece * 1024);
    } else {
      const int piece = i * 4 + widx;
      const unsigned voff = src_lane ^ ((unsigned)((((i * 4 * G::RPP) >> 1) & 7) << 4) & kmask);
      const char* s = src_h + (size_t)jt * G::TILE + piece * 1024;
      hgemm::glds16_asm(s, voff, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
    }
  };

  // ---- Q fragments of both row groups, pre-scaled
  h8 qf[2][NK];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const half_t* qp = Q + head + (size_t)(q_row0 + g * 32 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qf[g][ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }
  f16v ot[2][NDB];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[g][b][r] = 0.f;
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  f16v minit[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[g][r] = 0.f;
    asm volatile("" : "+v"(minit[g]));
  }

  const int T = KVS ? N / G::BC / 2 : N / G::BC;  // tiles this group walks
  const int jt0 = KVS ? grp * T : 0;              // its first tile
  __builtin_assume(T > 0);
#pragma unroll
  for (int i = 0; i < PPW; ++i) dma_piece(jt0, 0, i);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: also retires the Q loads
  {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) qf[g][ks] = qf[g][ks] * sc;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+v"(qf[g][ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int kbase = l31 * G::ROW + ((hi ^ G::swz_k(l31)) << 4);
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = v_row * G::ROW + ((G::swz_v(v_row) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) + ((i16 & 1) << 3);

  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  for (int j = 0; j < T; ++j) {
    const int jn = j + 1 < T ? j + 1 : T - 1;  // past the end: refill a dead slot with the last tile (branch-free)
    const int kb_j = kbase + gring + (j & 1) * G::STAGE, vb_j = vbase + gring + (j & 1) * G::STAGE + G::TILE;
    auto k_frag = [&](int t) __attribute__((always_inline)) {  // keys (t % BCB)*32 + l31, k-step t / BCB
      const int ks = t / BCB;
      return *reinterpret_cast<const h8*>(smem + (kb_j ^ ((ks & 7) << 5)) + (t % BCB) * 32 * G::ROW);
    };
    auto v_frag = [&](int idx) __attribute__((always_inline)) {  // idx = st * NDB + b: rows 16*st + v_row and + 8, block b
      const int st = idx / NDB, b = idx % NDB;
      const char* vp = smem + (vb_j ^ ((b & 3) << 6)) + (16 * st) * G::ROW;
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
    };
    // ================= phase A: S^T(g) = K Q(g)^T; this group's operand of tile j+1 is fetched meanwhile
    f16v s[2][BCB];
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (t < BCB) s[g][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[g][0], minit[g], 0, 0, 0);  // chain starts at -m
          else s[g][t % BCB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[g][t / BCB], s[g][t % BCB], 0, 0, 0);
          cln_mfma_keep(s[g][t % BCB], kf[t % PD], qf[g][t / BCB]);  // destination disjoint from the operands (common.h)
        }
        if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
        if ((t % (NQK / PPW)) == NQK / PPW - 1) dma_piece(jt0 + jn, (j + 1) & 1, t / (NQK / PPW));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    h8 pf[2][2 * BCB];
    auto row_max_and_rescale = [&]() __attribute__((always_inline)) {
      float d[2];
      bool grow = false;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float mx = s[g][0][0];
#pragma unroll
        for (int kb2 = 0; kb2 < BCB; ++kb2)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[g][kb2][r]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        d[g] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // the scores are relative to the running max
        grow |= d[g] > 8.0f;                                             // deferred: rescale past 2^8 only
      }
      const bool first = j == 0;  // tile 0 adopts its max unconditionally (the accumulators started at 0)
      if (first || __builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const float delta = first ? d[g] : fmaxf(d[g], 0.f);
          const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
          m_run[g] += delta;
          l_run[g] *= alpha;
#pragma unroll
          for (int kb2 = 0; kb2 < BCB; ++kb2)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[g][kb2][r] -= delta;
#pragma unroll
          for (int r = 0; r < 16; ++r) minit[g][r] = -m_run[g];
          asm volatile("" : "+v"(minit[g]));
#pragma unroll
          for (int b = 0; b < NDB; ++b)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {  // serialise the register round trips of the rescale
              float t0 = ot[g][b][r], t1 = ot[g][b][r + 1], t2 = ot[g][b][r + 2], t3 = ot[g][b][r + 3];
              asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
              ot[g][b][r] = t0 * alpha, ot[g][b][r + 1] = t1 * alpha, ot[g][b][r + 2] = t2 * alpha, ot[g][b][r + 3] = t3 * alpha;
            }
        }
      }
    };
    auto p_half = [&](int h) __attribute__((always_inline)) {  // fragments u = h*BCB .. (h+1)*BCB - 1 of both groups
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float psum = 0.f;
#pragma unroll
        for (int u = h * BCB; u < (h + 1) * BCB; ++u)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const int kb2 = u >> 1, r = (u & 1) * 8 + e;
            const float a0 = __builtin_amdgcn_exp2f(s[g][kb2][r]);
            const float a1 = __builtin_amdgcn_exp2f(s[g][kb2][r + 1]);
            psum += a0 + a1;
            const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
            pf[g][u][e] = a[0], pf[g][u][e + 1] = a[1];
          }
        l_run[g] += psum;
      }
    };
    row_max_and_rescale();
    p_half(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B: O^T(g) += V^T P(g)^T, second half of the exponentials between its halves
    h8 vf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    __builtin_amdgcn_sched_barrier(0);
    auto pv_range = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
      for (int idx = i0; idx < i1; ++idx) {
        const int st = idx / NDB, b = idx % NDB;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          ot[g][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx % PD], pf[g][st], ot[g][b], 0, 0, 0);
          cln_mfma_keep(ot[g][b], vf[idx % PD], pf[g][st]);
        }
        if (idx + PD < NPV) vf[idx % PD] = v_frag(idx + PD);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    pv_range(0, NPV / 2);  // P fragments of the first half are ready since phase A
    p_half(1);             // VALU under those MFMAs
    __builtin_amdgcn_sched_barrier(0);
    pv_range(NPV / 2, NPV);
    hgemm::wait_vmcnt<0>();  // own DMA pieces of tile j+1 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (grp == 0) {  // group 1's last phase B: keep the barrier count equal and the ring intact until it is done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows)
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
  auto finish_group = [&](auto gc, f16v (&og_t)[NDB], float l_part, char* ob, int row0) __attribute__((always_inline)) {
    float l_tot;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_part), __float_as_uint(l_part), false, false);
      l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(og_t[b][rq * 4 + e] * inv);
        *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int LPR = D / 8;
    half_t* og = O + head + (size_t)row0 * D;
#pragma unroll 4
    for (int it = 0; it < (32 * LPR) / 64; ++it) {
      const int idx = it * 64 + lane_e;
      const int row = idx / LPR, c = idx % LPR;
      *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    }
  };
  if constexpr (!KVS) {
#pragma unroll
    for (int g = 0; g < 2; ++g) finish_group(std::integral_constant<int, 0>{}, ot[g], l_run[g], smem + (wave * 2 + g) * (32 * G::OS), q_row0 + g * 32);
  } else {
    // merge the two key halves: wave (grp, widx) finalises row group `grp`, exports the other one to its partner
    auto merge = [&](auto goc) __attribute__((always_inline)) {
      constexpr int GO = decltype(goc)::value, GX = 1 - GO;
      float* xw = reinterpret_cast<float*>(smem + wave * G::XCH_WAVE) + lane_e;
#pragma unroll
      for (int b = 0; b < NDB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) xw[(b * 16 + r) * 64] = ot[GX][b][r];
      xw[32 * 64] = m_run[GX];
      xw[33 * 64] = l_run[GX];
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const float* xr = reinterpret_cast<const float*>(smem + (wave ^ 4) * G::XCH_WAVE) + lane_e;
      const float m_p = xr[32 * 64], l_p = xr[33 * 64];
      const float m = fmaxf(m_run[GO], m_p);
      const float sa = __builtin_amdgcn_exp2f(m_run[GO] - m), sb = __builtin_amdgcn_exp2f(m_p - m);
#pragma unroll
      for (int b = 0; b < NDB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[GO][b][r] = ot[GO][b][r] * sa + xr[(b * 16 + r) * 64] * sb;
      const float l_part = l_run[GO] * sa + l_p * sb;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();  // every wave has read its partner's export: the area becomes the staging buffer
      asm volatile("" ::: "memory");
      finish_group(goc, ot[GO], l_part, smem + wave * (32 * G::OS), q_row0 + GO * 32);
    };
    if (grp == 0) merge(std::integral_constant<int, 0>{});
    else merge(std::integral_constant<int, 1>{});
  }
}

template <int PD = 4, bool KVS = false>
int launch_dsplit2(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoSplit2;
  constexpr int BRW = KVS ? G::BR / 2 : G::BR;  // query rows per workgroup
  constexpr int LDS = KVS ? G::LDS_BYTES_KVS : G::LDS_BYTES;
  if (N % BRW != 0 || (KVS && N % (2 * G::BC) != 0)) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_dsplit2_kernel<PD, KVS>), LDS) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)G::D);
  const int n_qblk = N / BRW;
  CLN_LAUNCH((fa2_fwd_dsplit2_kernel<PD, KVS>), dim3(n_qblk * B * H), dim3(G::NT), LDS, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
