// FlashAttention-2 forward, head dims 64 / 128, two-group ping-pong kernel on v_mfma_f32_16x16x32_f16.
// Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66 (split-Q, Q in registers, K/V through LDS).
//
// Why (profiles/r02_fa_clock_power.log, r02_fa_rowsum_on_mfma_probe.log): attention sits at the
// package power cap and matrix energy is the scarce resource; the 32x32x16 shape of the other attention kernels measured
// 12 % less energy-efficient per flop than 16x16x32 in the HGEMM probe. This kernel is the shipped D = 64 ping-pong
// kernel (flash_attn_dsplit.cuh: 8 waves x 32 query rows, 128-key tiles, split softmax, pre-scaled Q, accumulators
// started at -m, deferred running max) re-laid-out for the 16x16x32 shape:
//   * S^T block (16 keys x 16 queries) = K[16 keys x 32 d] Q^T[32 d x 16 queries]: lane (query = lane & 15, g = lane >> 4)
//     holds keys 4g .. 4g+3 of the block -- a query row's scores live in FOUR lanes (two cross-row swaps for the row max
//     and the final row sum instead of one);
//   * P^T operand of k-step u (32 keys) = the lane's registers of key blocks 2u and 2u+1: k-slot 8g+j <-> key
//     32u + 4g + j (j < 4), 32u + 16 + 4g + (j-4) -- the V^T fragments are read in the SAME permuted key order (two
//     transposing reads 16 rows apart), so no data moves between the two matrix products;
//   * K image: 128-byte rows, chunk ^ ((row >> 1) & 7); V image: chunk ^ (((row >> 1) & 3) << 1) (the 16 rows x 32 bytes
//     of one transposing read cover all 64 banks twice); 256-byte rows (D = 128): chunk ^ (row & 15) / ((row & 7) << 1).
// Template: D = 64 | 128; RPW = query rows per wave, 32 (256-row workgroups) or 64 (512-row workgroups: every K / V
// fragment then feeds four MFMAs -- the long-sequence form, cf. probe/flash_attn_dsplit2.cuh); BC = keys per tile.
#pragma once
#include "flash_attn_dsplit.cuh"

namespace fa2 {

template <int D_, int RPW_, int BC_>
struct GeoM16 {
  static_assert((D_ == 64 || D_ == 128) && (RPW_ == 32 || RPW_ == 64) && BC_ % 64 == 0, "supported forms");
  static constexpr int D = D_, RPW = RPW_, BC = BC_, NW = 8, BR = RPW * NW, NT = 512;
  static constexpr int ROW = D * 2, TILE = BC * ROW, STAGE = 2 * TILE, RING = 2 * STAGE;
  static constexpr int OS = D * 2 + 16, EPI = NW * RPW * OS;
  static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;
  static constexpr int PPW = TILE / 1024 / 4;  // DMA pieces per wave per tile (4 waves fill one operand)
  static constexpr int RPP = 1024 / ROW, CPR = ROW / 16;
  static constexpr int NKB = BC / 16, NKS = D / 32, NQB = RPW / 16, NU = BC / 32, NDB = D / 16;
  static constexpr int NQK = NKB * NKS, NPV = NU * NDB;  // fragment steps per phase (each feeds NQB MFMAs)
  static __device__ __forceinline__ int swz_k(int row) { return ROW == 128 ? (row >> 1) & 7 : row & 15; }
  static __device__ __forceinline__ int swz_v(int row) { return ROW == 128 ? ((row >> 1) & 3) << 1 : (row & 7) << 1; }
};

template <int D_, int RPW_, int BC_, int PD = 4>
__global__ __launch_bounds__(512, 2) void fa2_fwd_m16_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                             const half_t* __restrict__ V, half_t* __restrict__ O,
                                                             int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoM16<D_, RPW_, BC_>;
  constexpr int D = G::D, NKB = G::NKB, NKS = G::NKS, NQB = G::NQB, NU = G::NU, NDB = G::NDB, NQK = G::NQK, NPV = G::NPV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
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

  // ---- LDS-DMA: wave widx of group 0 fills the 1-KiB pieces i*4 + widx of the K tile, group 1 those of the V tile
  // (lane-linear image, the swizzle applied to the SOURCE chunk). A piece is RPP rows: row = (i*4 + widx)*RPP + lr; both
  // swizzles only look at row bits below 4*RPP (128-byte rows: bits 1..3 of 32 rows; 256-byte rows: bits 0..3 of 16),
  // so they are functions of widx*RPP + lr: no per-piece term.
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR;
  const int sw_src = grp == 0 ? G::swz_k(widx * G::RPP + lr) : G::swz_v(widx * G::RPP + lr);
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ sw_src) << 4);
  auto dma_piece = [&](int jt, int slot, int i) __attribute__((always_inline)) {
    const int piece = i * 4 + widx;
    const char* s = src_h + (size_t)jt * G::TILE + piece * 1024;
    hgemm::glds16_asm(s, src_lane, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
  };

  // ---- Q fragments (B operand of S^T = K Q^T): query 16*qb + i16, d = 32*ks + 8*g4 .. +7, pre-scaled
  h8 qf[NQB][NKS];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const half_t* qp = Q + head + (size_t)(q_row0 + qb * 16 + i16) * D + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  f4 ot[NDB][NQB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) ot[b][qb] = f4{0.f, 0.f, 0.f, 0.f};
  float m_run[NQB], l_run[NQB];
  f4 minit[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    m_run[qb] = 0.f, l_run[qb] = 0.f;
    minit[qb] = f4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(minit[qb]));
  }

  const int T = N / G::BC;
  __builtin_assume(T > 0);
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: also retires the Q loads
  {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = qf[qb][ks] * sc;
  }
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // K fragment (kb, ks): row 16*kb + i16, logical chunk 4*ks + g4 -> (lane constant) ^ (ks << 6), + kb * 2048
  const int kbase = i16 * G::ROW + ((g4 ^ G::swz_k(i16)) << 4);
  // V^T fragment (u, db): rows 32*u + 4*g4 + (i16 >> 2) and + 16, logical chunk 2*db + ((i16 & 3) >> 1), 8-byte half i16 & 1
  const int v_row = 4 * g4 + (i16 >> 2);
  const int vbase = v_row * G::ROW + (((((i16 & 3) >> 1)) ^ G::swz_v(v_row)) << 4) + ((i16 & 1) << 3);

  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  for (int j = 0; j < T; ++j) {
    const int jn = j + 1 < T ? j + 1 : T - 1;  // past the end: refill a dead slot with the last tile (branch-free)
    const int kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    auto k_frag = [&](int t) __attribute__((always_inline)) {  // t = kb * NKS + ks
      const int kb = t / NKS, ks = t % NKS;
      return *reinterpret_cast<const h8*>(smem + (kb_j ^ (ks << 6)) + kb * 16 * G::ROW);
    };
    auto v_frag = [&](int idx) __attribute__((always_inline)) {  // idx = u * NDB + db
      const int u = idx / NDB, db = idx % NDB;
      const char* vp = smem + (vb_j ^ (db << 5)) + (32 * u) * G::ROW;
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 16 * G::ROW));
    };
    // ================= phase A: S^T = K Q^T; this group's operand of tile j+1 is fetched meanwhile
    f4 s[NKB][NQB];
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
      constexpr int DSTEP = NQK / G::PPW;
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        const int kb = t / NKS, ks = t % NKS;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          if (ks == 0) s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PD], qf[qb][0], minit[qb], 0, 0, 0);  // chain starts at -m
          else s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PD], qf[qb][ks], s[kb][qb], 0, 0, 0);
          cln_mfma_keep(s[kb][qb], kf[t % PD], qf[qb][ks]);  // destination disjoint from the operands (common.h)
        }
        if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
        if ((t % DSTEP) == DSTEP - 1) dma_piece(jn, (j + 1) & 1, t / DSTEP);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    h8 pf[NU][NQB];
    auto row_max_and_rescale = [&]() __attribute__((always_inline)) {
      float d[NQB];
      bool grow = false;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        float mx = s[0][qb][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        d[qb] = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));  // the scores are relative to the running max
        grow |= d[qb] > 8.0f;                                               // deferred: rescale past 2^8 only
      }
      const bool first = j == 0;  // tile 0 adopts its max unconditionally (the accumulators started at 0)
      if (first || __builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          const float delta = first ? d[qb] : fmaxf(d[qb], 0.f);
          const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
          l_run[qb] *= alpha;
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][qb][r] -= delta;
#pragma unroll
          for (int r = 0; r < 4; ++r) minit[qb][r] = -m_run[qb];
          asm volatile("" : "+v"(minit[qb]));
#pragma unroll
          for (int b = 0; b < NDB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[b][qb][r] *= alpha;
        }
      }
    };
    auto p_half = [&](int h) __attribute__((always_inline)) {  // k-steps u = h*NU/2 .. (h+1)*NU/2 - 1 of both query blocks
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        float psum = 0.f;
#pragma unroll
        for (int u = h * (NU / 2); u < (h + 1) * (NU / 2); ++u)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const int kb = 2 * u + (e >> 2), r = e & 3;
            const float a0 = __builtin_amdgcn_exp2f(s[kb][qb][r]);
            const float a1 = __builtin_amdgcn_exp2f(s[kb][qb][r + 1]);
            psum += a0 + a1;
            const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
            pf[u][qb][e] = a[0], pf[u][qb][e + 1] = a[1];
          }
        l_run[qb] += psum;
      }
    };
    row_max_and_rescale();
    p_half(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B: O^T += V^T P^T, second half of the exponentials between its halves
    h8 vf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    __builtin_amdgcn_sched_barrier(0);
    auto pv_range = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
      for (int idx = i0; idx < i1; ++idx) {
        const int u = idx / NDB, b = idx % NDB;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          ot[b][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[idx % PD], pf[u][qb], ot[b][qb], 0, 0, 0);
          cln_mfma_keep(ot[b][qb], vf[idx % PD], pf[u][qb]);
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

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows). Lane (query 16*qb + i16) holds d = 16*b + 4*g4 .. +3.
  const int lane_e = cln_fresh_lane(), i16_e = lane_e & 15, g4_e = lane_e >> 4;
  char* ob = smem + wave * (G::RPW * G::OS);
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float l_tot = l_run[qb];
    {
      const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
      l_tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
      const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
      l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    }
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][qb][e] * inv);
      *reinterpret_cast<h4*>(ob + (qb * 16 + i16_e) * G::OS + (b * 16 + g4_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
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

template <int D_, int RPW_, int BC_, int PD = 4>
int launch_m16(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoM16<D_, RPW_, BC_>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_m16_kernel<D_, RPW_, BC_, PD>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)G::D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_m16_kernel<D_, RPW_, BC_, PD>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Head dim 512 (config C5) on 16x16x32 MFMAs: the d-split PAIR form of flash_attn_dsplit.cuh (NSP = 2) in the lane layout
// above. A pair of waves (wave, wave ^ 1) owns 32 query rows, each holds one 256-column half of d: Q 64 + O^T 128
// registers, a partial S^T over its half, the partials swapped through LDS (4 KiB each way), both run the same softmax on
// the (commutative, hence bit-identical) sum and each accumulates its own 256 output columns. 128 query rows per
// workgroup, 32-key tiles, 1024-byte rows: every row starts at bank 0, so the K image is chunk ^ (row & 15) and the V
// image chunk ^ ((row & 15) << 1) (16 rows x 32 bytes of one transposing read = all 64 banks twice); a DMA piece is ONE
// row (row = 4*i + widx), so both swizzles carry a per-piece term. Q is pre-scaled; the wave with part = 0 starts its
// partial at -m through the MFMA C operand, so the exchanged sum is already relative to the running max.
// PROBE ONLY (variants 540 / 541 of kind 8, tested, not dispatched): +3.2 % at config C5 (1025 vs 994 TF), but rounding
// Q * log2(e)/sqrt(512) to fp16 raises the max-abs-error from 1.0e-4 to 3.1e-4 at C5 (3.6e-4 -> 9.7e-4 at [2,3,256,512]);
// a form that scales the exchanged sum in fp32 instead measured +0.8 % only and was dropped (profiles/r02_fa_m16_pair_probe.log).
// PAIR = false: the same kernel for head dim 256 -- ONE wave per 32 query rows holds the whole d (still Q 64 + O^T 128
// registers), no exchange, 256 query rows per workgroup, 512-byte rows (a DMA piece is two rows: row = 8*i + 2*widx + lr).
template <bool PAIR>
struct GeoM16Pair {
  static constexpr int D = PAIR ? 512 : 256, DH = 256, BC = 32, NW = 8, BR = PAIR ? 128 : 256, NT = 512;
  static constexpr int ROW = D * 2, TILE = BC * ROW, STAGE = 2 * TILE, RING = 2 * STAGE, SX = PAIR ? NW * 4096 : 0;
  static constexpr int RPP = 1024 / ROW, CPR = ROW / 16;  // rows per 1-KiB DMA piece, 16-byte chunks per row
  static constexpr int OS = DH * 2 + 16, EPI = NW * 32 * OS;
  static constexpr int LDS_BYTES = RING + SX > EPI ? RING + SX : EPI;
  static constexpr int PPW = TILE / 1024 / 4;
  static constexpr int NKB = BC / 16, NKS = DH / 32, NQB = 2, NDB = DH / 16, NQK = NKB * NKS, NPV = NDB;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static __device__ __forceinline__ int swz_k(int row) { return row & 15; }
  static __device__ __forceinline__ int swz_v(int row) { return (row & 15) << 1; }
};

// PRE = false: Q stays as loaded; the (summed) scores are scaled and made relative to the running max in fp32, one v_fma per
// score (16 per lane and tile) -- the accuracy of the 32x32x16 kernels at these head dims.
template <int PD = 2, bool PAIR = true, bool PRE = true, int DBG = 0>
__global__ __launch_bounds__(512, 2) void fa2_fwd_m16_pair_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                                  const half_t* __restrict__ V, half_t* __restrict__ O,
                                                                  int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoM16Pair<PAIR>;
  constexpr int D = G::D, DH = G::DH, NKB = G::NKB, NKS = G::NKS, NQB = G::NQB, NDB = G::NDB, NQK = G::NQK, NPV = G::NPV;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 1024: the fragment addresses below XOR bits 5 .. 8 into (LDS address of smem + offset)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
  const int grp = wave >> 2, widx = wave & 3;
  const int part = PAIR ? widx & 1 : 0, rg = PAIR ? grp * 2 + (widx >> 1) : wave;

  int head_i, qb_i;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {
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

  // ---- LDS-DMA: a piece = RPP rows (1 KiB): row = (4*i + widx)*RPP + lr, lane = lr*CPR + chunk position; source chunk =
  // position ^ swizzle(row), and swizzle(row) = swizzle(widx*RPP + lr) ^ swizzle(4*i*RPP) (disjoint bits)
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = widx * G::RPP + lr;
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ (grp == 0 ? G::swz_k(rlow) : G::swz_v(rlow))) << 4);
  auto dma_piece = [&](int jt, int slot, int i) __attribute__((always_inline)) {
    const int piece = i * 4 + widx;
    const unsigned voff = src_lane ^ (unsigned)((grp == 0 ? G::swz_k(4 * i * G::RPP) : G::swz_v(4 * i * G::RPP)) << 4);
    const char* s = src_h + (size_t)jt * G::TILE + piece * 1024;
    hgemm::glds16_asm(s, voff, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
  };

  // ---- Q fragments of this wave's half of d, pre-scaled
  h8 qf[NQB][NKS];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const half_t* qp = Q + head + (size_t)(q_row0 + qb * 16 + i16) * D + part * DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  f4 ot[NDB][NQB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) ot[b][qb] = f4{0.f, 0.f, 0.f, 0.f};
  float dbg_stale = 0.f;
  float m_run[NQB], l_run[NQB];
  f4 minit[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    m_run[qb] = 0.f, l_run[qb] = 0.f;
    minit[qb] = f4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(minit[qb]));
  }

  const int T = N / G::BC;
  __builtin_assume(T > 0);
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if constexpr (PRE) {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = qf[qb][ks] * sc;
  }
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // K fragment (kb, ks): row 16*kb + i16, logical chunk part*32 + 4*ks + g4; V^T fragment (db): rows 4*g4 + (i16 >> 2)
  // and + 16, logical chunk part*32 + 2*db + ((i16 & 3) >> 1), 8-byte half i16 & 1
  // (LDS byte addresses, the symbol's address folded in once: common.h lds_ld -- no per-access v_add_u32 of the symbol inside the loop)
  const unsigned kbase = lds0 + i16 * G::ROW + ((g4 ^ G::swz_k(i16)) << 4) + part * 512;
  const int v_row = 4 * g4 + (i16 >> 2);
  const unsigned vbase = lds0 + v_row * G::ROW + (((((i16 & 3) >> 1)) ^ G::swz_v(v_row)) << 4) + ((i16 & 1) << 3) + part * 512;
  const unsigned sx_mine = lds0 + G::RING + wave * 4096 + lane * 16;
  const unsigned sx_peer = lds0 + G::RING + (wave ^ 1) * 4096 + lane * 16;
  const bool lead = PRE && part == 0;  // this wave's partial starts at -m (PAIR = false: every wave)

  if (grp == 1) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  for (int j = 0; j < T; ++j) {
    const int jn = j + 1 < T ? j + 1 : T - 1;
    const unsigned kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    auto k_frag = [&](int t) __attribute__((always_inline)) {  // t = kb * NKS + ks
      const int kb = t / NKS, ks = t % NKS;
      return lds_ld<h8>((kb_j ^ (unsigned)(ks << 6)) + kb * 16 * G::ROW);
    };
    auto v_frag = [&](int db) __attribute__((always_inline)) {
      const unsigned vp = vb_j ^ (unsigned)(db << 5);
      return h8_cat(lds_read_tr16_at(vp), lds_read_tr16_at(vp + 16 * G::ROW));
    };
    // ================= phase A: partial S^T over this wave's half of d
    // DBG 262144 = the `stages = 1` form in ONE burst (round 4): the wave requests all its pieces of tile j + 1 here and waits for them
    // here; no request of the wave is in flight while it computes (DBG 131072, round 3: the wait after EVERY piece, 0.27x)
    if constexpr ((DBG & 262144) != 0 && (DBG & 524288) == 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G::PPW; ++i) dma_piece(jn, (j + 1) & 1, i);
      hgemm::wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((DBG & 65536) != 0) __builtin_amdgcn_s_setprio(0);
    f4 s[NKB][NQB];
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
      constexpr int DSTEP = NQK / G::PPW;
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        const int kb = t / NKS, ks = t % NKS;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          if (ks == 0) {
            if constexpr (!PRE && (DBG & 16384) != 0) {
              // zero-C chain start with an EARLY-CLOBBER destination: it cannot be given the registers of its A / B operand
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s[kb][qb]) : "v"(kf[t % PD]), "v"(qf[qb][0]));
            } else {
              const f4 c0 = lead ? minit[qb] : f4{0.f, 0.f, 0.f, 0.f};
              s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PD], qf[qb][0], c0, 0, 0, 0);
            }
          } else {
            s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PD], qf[qb][ks], s[kb][qb], 0, 0, 0);
          }
          if constexpr ((DBG & 32768) == 0) cln_mfma_keep(s[kb][qb], kf[t % PD], qf[qb][ks]);  // destination disjoint from the operands (common.h; DBG 32768: the round-2 code)
        }
        // keep the K fragment alive past its MFMAs: hipcc may otherwise give the LAST MFMA that reads it the fragment's
        // registers as its destination (no early-clobber on v_mfma_f32_16x16x32_f16) -- see mfma_overlap_scan.py
        if constexpr ((DBG & 8192) != 0) asm volatile("" ::"v"(kf[t % PD]));
        if constexpr ((DBG & 1024) != 0) asm volatile("s_nop 15" ::: "memory");
        if constexpr ((DBG & 2048) != 0) asm volatile("s_sleep 1" ::: "memory");
        if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
        if ((DBG & 262144) == 0 && (t % DSTEP) == DSTEP - 1) {
          dma_piece(jn, (j + 1) & 1, t / DSTEP);
          // DBG 131072 = the `stages = 1` form: every tile fetch is waited for where it is issued, no load runs under compute
          if constexpr ((DBG & 131072) != 0) hgemm::wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr ((DBG & 1) != 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    if constexpr (PAIR)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) lds_st<f4>(sx_mine + (kb * NQB + qb) * 1024, s[kb][qb]);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the partial is in LDS before the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B: S = own + partner's partial, softmax, O^T[half] += V[:, half]^T P^T
    if constexpr ((DBG & 262144) != 0 && (DBG & 524288) != 0) {  // probe: the stages = 1 burst at the top of phase B instead
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G::PPW; ++i) dma_piece(jn, (j + 1) & 1, i);
      hgemm::wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((DBG & 65536) != 0) __builtin_amdgcn_s_setprio(1);  // the VALU-carrying phase wins the issue arbitration
    f4 pp_first[2];
    if constexpr (PAIR)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        const f4 pp = lds_ld<f4>(sx_peer + (kb * NQB + qb) * 1024);
        if constexpr ((DBG & 512) != 0) {
          if (kb == 0 && qb == 0) pp_first[0] = pp;
          if (kb == 1 && qb == 0) pp_first[1] = pp;
        }
        s[kb][qb] += pp;
      }
    if constexpr ((DBG & 2) != 0) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((DBG & 64) != 0) asm volatile("s_sleep 4" ::: "memory");
    if constexpr ((DBG & 16) != 0)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(s[kb][qb][r]));  // per-element opacity: no SLP pairing into v_pk_fma_f32
    if constexpr ((DBG & 8) != 0) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    float sc_v = scale_log2e;
    if constexpr ((DBG & 4) != 0) asm volatile("" : "+v"(sc_v));  // the scale in a VGPR (no SGPR-pair operand of a packed f32 op)
    if constexpr (!PRE && (DBG & (128 | 256)) != 0) {
      // bisect: the packed fma written by hand with the scale in an SGPR PAIR; 128: the high lane reads the LOW dword of the
      // pair (op_sel_hi 0), 256: the high lane reads the HIGH dword (what hipcc emits)
      const unsigned sb = __builtin_amdgcn_readfirstlane(__float_as_uint(scale_log2e));
      const unsigned long long sp = ((unsigned long long)sb << 32) | sb;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            f2 a = f2{s[kb][qb][r], s[kb][qb][r + 1]}, o2;
            const f2 c = f2{m_run[qb], m_run[qb]};
            if constexpr ((DBG & 128) != 0)
              asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(o2) : "v"(a), "s"(sp), "v"(c));
            else
              asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(o2) : "v"(a), "s"(sp), "v"(c));
            s[kb][qb][r] = o2[0], s[kb][qb][r + 1] = o2[1];
          }
    } else if constexpr (!PRE)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kb][qb][r] = fmaf(s[kb][qb][r], sc_v, -m_run[qb]);  // log2 domain, relative
    h8 vf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    __builtin_amdgcn_sched_barrier(0);
    {
      float d[NQB];
      bool grow = false;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        float mx = s[0][qb][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        d[qb] = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
        grow |= d[qb] > 8.0f;
      }
      const bool first = j == 0;
      if (first || __builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          const float delta = first ? d[qb] : fmaxf(d[qb], 0.f);
          const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
          l_run[qb] *= alpha;
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][qb][r] -= delta;
#pragma unroll
          for (int r = 0; r < 4; ++r) minit[qb][r] = -m_run[qb];
          asm volatile("" : "+v"(minit[qb]));
#pragma unroll
          for (int b = 0; b < NDB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[b][qb][r] *= alpha;
        }
      }
    }
    if constexpr (PAIR && (DBG & 512) != 0) {  // bisect: was the partner's partial complete when it was read?
      asm volatile("s_sleep 8" ::: "memory");
      const f4 a = *reinterpret_cast<const volatile f4*>(smem + (sx_peer - lds0)), b = *reinterpret_cast<const volatile f4*>(smem + (sx_peer - lds0) + 2048);
      bool diff = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) diff |= (a[r] != pp_first[0][r]) | (b[r] != pp_first[1][r]);
      if (__builtin_amdgcn_ballot_w64(diff) != 0) dbg_stale += 1.0f;
    }
    h8 pf[NQB];  // the one 32-key step: k-slot 8*g4 + e <-> key 16*(e >> 2) + 4*g4 + (e & 3)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int kb = e >> 2, r = e & 3;
        const float a0 = __builtin_amdgcn_exp2f(s[kb][qb][r]);
        const float a1 = __builtin_amdgcn_exp2f(s[kb][qb][r + 1]);
        psum += a0 + a1;
        const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
        pf[qb][e] = a[0], pf[qb][e + 1] = a[1];
      }
      l_run[qb] += psum;
    }
#pragma unroll
    for (int db = 0; db < NPV; ++db) {
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        ot[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[db % PD], pf[qb], ot[db][qb], 0, 0, 0);
        if constexpr ((DBG & 32768) == 0) cln_mfma_keep(ot[db][qb], vf[db % PD], pf[qb]);
      }
      if constexpr ((DBG & 4096) != 0) asm volatile("s_sleep 1" ::: "memory");
      if (db + PD < NPV) vf[db % PD] = v_frag(db + PD);
      __builtin_amdgcn_sched_barrier(0);
    }
    hgemm::wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: this wave's 256 output columns of its 32 rows, staged through LDS (wave-private rows)
  const int lane_e = cln_fresh_lane(), i16_e = lane_e & 15, g4_e = lane_e >> 4;
  char* ob = smem + wave * (32 * G::OS);
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float l_tot = l_run[qb];
    {
      const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
      l_tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
      const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
      l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    }
    if constexpr ((DBG & 512) != 0) l_tot = dbg_stale > 0.f ? l_tot * 1e-3f : l_tot;  // a detected stale read blows the row up by 1000
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][qb][e] * inv);
      *reinterpret_cast<h4*>(ob + (qb * 16 + i16_e) * G::OS + (b * 16 + g4_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = DH / 8;
  half_t* og = O + head + (size_t)q_row0 * D + part * DH;
#pragma unroll 4
  for (int it = 0; it < (32 * LPR) / 64; ++it) {
    const int idx = it * 64 + lane_e;
    const int row = idx / LPR, c = idx % LPR;
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
  }
}

template <int PD = 2, bool PAIR = true, bool PRE = true, int DBG = 0>
int launch_m16_pair(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoM16Pair<PAIR>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_m16_pair_kernel<PD, PAIR, PRE, DBG>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)G::D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_m16_pair_kernel<PD, PAIR, PRE, DBG>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
