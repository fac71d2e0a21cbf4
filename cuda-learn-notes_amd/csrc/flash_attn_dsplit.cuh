// FlashAttention-2 forward for head dim 512 (BASELINE config C5 = [1,32,4096,512]), "d-split ping-pong" form.
// Reference rungs: kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:70, :732 (and tiling_qk.cu:72): the
// reference streams Q, K and V in 16-wide d slices through O(1) shared memory. probe/flash_attn_bigd.cuh keeps Q and a
// 256-wide O^T slice in the register file of ONE wave per SIMD and pays for it with (a) S recomputed per output
// slice (1.5x the MFMA work), (b) every LDS-DMA issue stall and the whole softmax exposed (nobody else on the SIMD).
// This kernel splits the head dim across a PAIR of waves instead:
//   * 8 waves = 4 pairs, a pair owns 32 query rows; wave `part` of a pair holds Q[:, part*256 .. +256) (64
//     registers) and O^T[part*256 .. +256, :] (128 registers) -> two waves per SIMD, 256 registers each;
//   * per 32-key tile each wave computes a PARTIAL S^T over its half of d (16 MFMAs), the partners swap partials
//     through LDS (4 KiB each way) and both run the same softmax on the sum (identical bits: fp32 add commutes),
//     then each does P.V for its own 256 output columns (16 MFMAs): no recomputation, MFMA work = algorithmic;
//   * the two 4-wave groups (pairs {0,1} / {2,3}; waves w and w+4 share a SIMD) run ONE PHASE APART, ping-pong
//     style: while a group is in its softmax + PV phase the other is in QK^T, so each SIMD always has matrix work
//     from one wave beside the VALU / LDS-DMA work of the other. Two workgroup barriers per tile serve as the phase
//     clock, the partial-S hand-off and the K/V ring protection at once;
//   * K tiles are fetched by group 0 and V tiles by group 1 (LDS-DMA, 8 x 1 KiB per wave per tile, interleaved with
//     the QK^T MFMAs), double-buffered, same source-side XOR swizzles as the big-D kernel.
#pragma once
#include "flash_attn_v2.cuh"
#include "hgemm_mfma.cuh"  // glds16_asm, lds_addr_of, wait_vmcnt

namespace fa2 {

// NSP = waves sharing one 32-row query group: 2 at D = 512 (each holds half of d), 1 at D = 256 / 128 (no split, no
// exchange: the same two-group phase structure with 8 x 32 = 256 query rows per workgroup). BCB = 32-key blocks per
// KV tile (2 at D = 128: 16 + 16 MFMAs per wave and tile like the larger head dims).
template <int D, int NSP, int BCB = 1>
struct GeoSplit {
  static constexpr int BC = 32 * BCB, NW = 8, BR = 32 * NW / NSP, NT = 512, DH = D / NSP;
  static constexpr int ROW = D * 2;            // bytes per K / V row
  static constexpr int TILE = BC * ROW;        // one K or V tile
  static constexpr int STAGE = 2 * TILE;       // K + V
  static constexpr int RING = 2 * STAGE;
  static constexpr int SX = NSP == 2 ? NW * 4096 : 0;  // partial-S exchange: 4 KiB per wave
  static constexpr int OS = DH * 2 + 16;
  static constexpr int EPI = NW * 32 * OS;
  static constexpr int LDS_BYTES = RING + SX > EPI ? RING + SX : EPI;
  static constexpr int PPW = TILE / 1024 / 4;  // DMA pieces per wave per tile (4 waves fill one operand)
  static constexpr int RPP = 1024 / ROW;       // rows per 1-KiB DMA piece
  static constexpr int CPR = ROW / 16;         // 16-byte chunks per row
  static_assert((D == 512 && NSP == 2 && BCB == 1) || (D == 256 && NSP == 1 && BCB == 1) || (D == 128 && NSP == 1) ||
                    (D == 64 && NSP == 1 && BCB >= 2),
                "d-split kernel: DH = 256, 128 with 32- or 64-key tiles, 64 with 64- or 128-key tiles");
  // XOR swizzles of the 16-byte chunk index (LDS images are lane-linear, so the swizzle is applied to the DMA source
  // address and to the fragment read). Rows of >= 256 bytes: K chunk ^= row & 15, V chunk ^= (row & 3) << 2
  // (probe/flash_attn_bigd.cuh). 128-byte rows (D = 64): two rows span the 64 banks, so K uses (row >> 1) & 7 and V moves
  // the 64-byte block by (row >> 1) & 1.
  static __device__ __forceinline__ int swz_k(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
  static __device__ __forceinline__ int swz_v(int row) { return CPR >= 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// DREAL != 0 (D = 512 only): the tensors have DREAL < D columns (320 / 384). The LDS geometry stays that of D = 512
// (1024-byte rows, one 512-byte half per wave of a pair), but the pair splits the REAL head dim evenly: wave `part` owns
// columns [part * DREAL/2, (part + 1) * DREAL/2), which the DMA places at the start of its LDS half (the rest of the half
// is never read), and every loop runs over DREAL/2 columns -- 12 / 10 k-steps and 6 / 5 output blocks per wave instead
// of 16 / 8: no MFMA multiplies padding (round 1 padded Q with zeros: 25 % / 37 % of the MFMAs were wasted).
template <int D, int NSP, int BCB, int OPT, int ABL = 0, int DREAL = 0>
__global__ __launch_bounds__(512, 1) void fa2_fwd_dsplit_kernel(const half_t* __restrict__ Q,
                                                                const half_t* __restrict__ K,
                                                                const half_t* __restrict__ V, half_t* __restrict__ O,
                                                                int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoSplit<D, NSP, BCB>;
  constexpr bool PAD = DREAL != 0;
  static_assert(!PAD || (D == 512 && NSP == 2 && DREAL % 64 == 0 && DREAL > 256 && DREAL < 512),
                "head dims 320 / 384 ride on the D = 512 geometry");
  constexpr int DR = PAD ? DREAL : D;            // columns per row in memory
  constexpr int DHR = PAD ? DREAL / 2 : G::DH;   // columns this wave works on
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ABL & 128 (probe only): s_memrealtime (100 MHz, chip-wide) at kernel entry, after the prologue, at the start of
  // tiles 1 2 3 4 8 16, after the KV loop and after the O stores -- every wave writes them over the head of its O rows
  unsigned long long life[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr ((ABL & 128) != 0) life[0] = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2, widx = wave & 3;
  const int part = NSP == 2 ? (widx & 1) : 0, rg = NSP == 2 ? grp * 2 + (widx >> 1) : wave;

  int head_i, qb;
  {
    const int bid = blockIdx.x;
    if ((OPT & OPT_XCD) && (n_heads & 7) == 0) {
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * DR;
  const int q_row0 = qb * G::BR + rg * 32;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // ---- LDS-DMA: wave widx of group 0 fills the 1-KiB pieces i*4 + widx of the K tile, group 1 the same pieces of
  // the V tile. A piece is RPP rows; lane l carries 16-byte chunk c = l % CPR of row (piece*RPP + l / CPR) to the
  // lane-linear LDS position, reading it from the source chunk (c ^ swizzle(row)):
  //   K: row & 15 = (i*4*RPP & 15) + widx*RPP + l/CPR (disjoint bits),   V: (row & 3) << 2 = ((widx*RPP + l/CPR) & 3) << 2.
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = widx * G::RPP + lr;
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (grp == 0 ? (unsigned)((lc ^ G::swz_k(rlow)) << 4) : (unsigned)((lc ^ G::swz_v(rlow)) << 4));
  const unsigned kmask = grp == 0 ? 0xFFu : 0u;  // the i-dependent part of the swizzle applies to K rows only
  auto dma_piece = [&](int jt, int slot, int i) {
    const int piece = i * 4 + widx;
    unsigned voff = src_lane ^ ((unsigned)(((i * 4 * G::RPP) & 15) << 4) & kmask);
    const char* s;
    if constexpr (PAD) {  // a piece is one 1024-byte LDS row (RPP = 1); voff >> 4 = the logical chunk X this lane's
      // LDS position holds: half p = X >> 5, chunk cc = X & 31 of that half; real if cc < DHR/8 (else never read)
      const unsigned X = voff >> 4, pp = X >> 5, cc = X & 31;
      voff = cc < (unsigned)(DHR / 8) ? (pp * (unsigned)(DHR / 8) + cc) << 4 : 0u;
      s = src_h + (size_t)jt * (G::BC * DR * 2) + piece * (DR * 2);
    } else {
      s = src_h + (size_t)jt * G::TILE + piece * 1024;
    }
    hgemm::glds16_asm(s, voff, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
  };

  // ---- Q fragments: this wave's half of the head dim
  h8 qf[DHR / 16];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * DR + part * DHR + hi * 8;
#pragma unroll
    for (int ks = 0; ks < DHR / 16; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }
  // OPT_PRE (the VALU diet; the D = 64 kernel is VALU-bound: ~10 VALU instructions per MFMA, 49 VALU cycles against
  // 32 matrix cycles, profiles/r01_pmc_fa_v2_variants.json): Q is multiplied by log2(e)/sqrt(d) ONCE here and the S^T
  // accumulators of every tile START at -m through the C operand of their first MFMA (m = the deferred running row
  // max, a per-lane constant because a lane owns one query row) -- so P = exp2(acc): no per-element v_fma and no
  // accumulator zeroing (2 of the ~5 VALU instructions per score). A rescale (rare) also shifts the pending scores.
  constexpr bool PRE = (OPT & OPT_PRE) != 0;
  static_assert(!PRE || NSP == 1, "OPT_PRE: one wave per row group");
  f16v ot[DHR / 32];
#pragma unroll
  for (int b = 0; b < DHR / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = PRE ? 0.f : -1.0e30f, l_run = 0.f;
  // OPT_SUMM: the row sums ride on the matrix pipe -- one more MFMA per 16-key step with an all-ones A operand:
  // lacc[m][n] = sum_k P^T[k][n] for every m, i.e. each lane's 16 registers all hold the sum of ITS query row (no
  // per-score v_add, no cross-half exchange at the end). +25 % MFMAs at D = 64 for -42 VALU instructions per tile.
  constexpr bool SUMM = (OPT & OPT_SUMM) != 0;
  static_assert(!SUMM || (PRE && NSP == 1), "OPT_SUMM: on the pre-scaled single-wave-per-row-group kernel");
  f16v lacc;
  h8 ones;
  if constexpr (SUMM) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.0f;
    asm volatile("" : "+v"(ones));
  }
  f16v minit;  // OPT_PRE: -m in all 16 registers (opaque to hipcc: otherwise the splat is re-materialised per tile)
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  if constexpr (PRE) asm volatile("" : "+v"(minit));

  const int T = N / G::BC;
  __builtin_assume(T > 0);  // the launcher rejects N < BR: no zero-trip path (its phi copies cost registers)
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: also retires the Q loads in its bookkeeping
  if constexpr (PRE) {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int ks = 0; ks < DHR / 16; ++ks) qf[ks] = qf[ks] * sc;
  }
  // ... and pin the fragments here: hipcc otherwise sinks the Q loads below the barrier and into the first KV iteration
#pragma unroll
  for (int ks = 0; ks < DHR / 16; ++ks) asm volatile("" : "+v"(qf[ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // Fragment offsets (see probe/flash_attn_bigd.cuh for the swizzles); this wave reads d columns part*256 .. +256 of K
  // (k-steps part*16 ..) and of V (output blocks part*8 ..): + part*512 bytes in both images. The swizzle only
  // touches bits 5..7 (K) / 6..7 (V) of the byte offset, so fragment i is (one lane constant) ^ (i << 5 | 6) plus a
  // compile-time immediate: two address registers instead of twelve -- the register file is full by design.
  const int kbase = l31 * G::ROW + ((hi ^ G::swz_k(l31)) << 4) + part * 512;
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = v_row * G::ROW + ((G::swz_v(v_row) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) +
                    ((i16 & 1) << 3) + part * 512;

  char* sx_mine = smem + G::RING + wave * 4096 + lane * 16;
  const char* sx_peer = smem + G::RING + (wave ^ 1) * 4096 + lane * 16;

  if constexpr ((OPT & OPT_SOLO) != 0) {  // static priority for the younger half (loses every arbitration otherwise)
    if (grp == 1) __builtin_amdgcn_s_setprio(1);
  }
  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ABL & 32: s_memtime at the phase boundaries of tile 16
  auto mark = [&](int j, int i) {
    if constexpr ((ABL & 32) != 0)
      if (j == 16) stamp[i] = __builtin_amdgcn_s_memtime();
  };
  if constexpr ((ABL & 128) != 0) life[1] = __builtin_amdgcn_s_memrealtime();
  for (int j = 0; j < T; ++j) {
    mark(j, 0);
    if constexpr ((ABL & 128) != 0) {
      if (j == 1) life[2] = __builtin_amdgcn_s_memrealtime();
      if (j == 2) life[3] = __builtin_amdgcn_s_memrealtime();
      if (j == 3) life[4] = __builtin_amdgcn_s_memrealtime();
      if (j == 4) life[5] = __builtin_amdgcn_s_memrealtime();
      if (j == 8) life[6] = __builtin_amdgcn_s_memrealtime();
      if (j == T - 1) life[7] = __builtin_amdgcn_s_memrealtime();
    }
    const char* kb = smem + (j & 1) * G::STAGE;
    const char* vb = kb + G::TILE;
    // ================= phase A: partial S^T = K[:, half] Q[:, half]^T; fetch this group's operand of tile j+1
    const int jn = j + 1 < T ? j + 1 : T - 1;  // past the end: refill a dead slot with the last tile (branch-free)
    const int kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    auto k_frag = [&](int t) {  // keys (t % BCB)*32 + l31, k-step t / BCB
      const int ks = t / BCB;
      if constexpr ((ABL & 64) != 0) return qf[ks];  // ablation: no K fragment reads
      return *reinterpret_cast<const h8*>(smem + (kb_j ^ ((ks & 7) << 5)) + (ks >> 3) * 256 + (t % BCB) * 32 * G::ROW);
    };
    auto v_frag = [&](int idx) {  // idx = st * (DH/32) + b: rows 16*st + v_row and + 8 (same swizzle), block b
      const int st = idx / (DHR / 32), b = idx % (DHR / 32);
      if constexpr ((ABL & 64) != 0) return qf[idx % (DHR / 16)];  // ablation: no V fragment reads
      const char* vp = smem + (vb_j ^ ((b & 3) << 6)) + (16 * st) * G::ROW + (b >> 2) * 256;
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
    };
    constexpr int NK = DHR / 16, NQK = BCB * NK, NPV = 2 * BCB * (DHR / 32), NQK0 = NQK < NPV ? NQK : NPV;
    constexpr int DSTEP = NQK / G::PPW >= 1 ? NQK / G::PPW : 1;  // QK^T MFMAs per DMA piece (12 / 10 MFMAs carry 8 pieces at 384 / 320)
    static_assert(NQK >= G::PPW, "every DMA piece of the next tile needs an MFMA to ride on");
    // fragments in flight ahead of the MFMA that consumes them (the phase stamps of round 1 show the MFMA loops running
    // at 2-3x their matrix time: every MFMA waits for an LDS fragment read issued only a few MFMAs earlier)
    constexpr int PD0 = (OPT & OPT_PD16) ? 16 : (OPT & OPT_PD8) ? 8 : (OPT & OPT_KPRE) ? 4 : 1;
    constexpr int PD = PD0 < NQK0 ? PD0 : NQK0;
    f16v s[BCB];
    if constexpr (!PRE) {
#pragma unroll
      for (int kb2 = 0; kb2 < BCB; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb2][r] = 0.f;
    }
    // ABL 512 = the `stages = 1` form in ONE burst (round 4): the wave requests all its pieces of tile j + 1 here, at the top of phase A,
    // and waits for them here -- no request of the wave in flight while it computes (ABL 256, round 3: a wait after every piece, 0.37x)
    if constexpr ((ABL & 512) != 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G::PPW; ++i) dma_piece(jn, (j + 1) & 1, i);
      hgemm::wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      // MFMA t works on k-step t / BCB of key block t % BCB (consecutive MFMAs alternate accumulators at BCB = 2)
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        if (ABL & 16) s[t % BCB][t / BCB] += (float)kf[t % PD][0];
        else if (PRE && t < BCB) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[0], minit, 0, 0, 0);  // chain starts at -m
        else s[t % BCB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[t / BCB], s[t % BCB], 0, 0, 0);
        cln_mfma_keep(s[t % BCB], kf[t % PD], qf[t / BCB]);  // destination disjoint from the operands (common.h)
        if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
        if (!(ABL & 1) && !(ABL & 512) && (t % DSTEP) == DSTEP - 1 && t / DSTEP < G::PPW) {
          dma_piece(jn, (j + 1) & 1, t / DSTEP);
          // ABL 256 = the `stages = 1` form: every tile fetch is waited for where it is issued, no load runs under compute
          if constexpr ((ABL & 256) != 0) hgemm::wait_vmcnt<0>();
        }
        if (PD > 1 || (t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    mark(j, 1);
    // ---- softmax pieces. The P^T fragment of k-step u (16 keys) comes from accumulator registers
    // s[u / 2][(u & 1) * 8 .. + 8]; the 2*BCB fragments are produced in two halves.
    // SPLIT (OPT_STAGGER, NSP = 1 only): row max, rescale decision and the FIRST half of the exponentials run in
    // phase A behind the QK^T MFMAs, the second half in phase B between the two halves of the PV MFMAs -- phase A is
    // otherwise matrix-only and short, and the partner group idles at the barrier for the length of the softmax.
    constexpr bool SPLIT = NSP == 1 && (OPT & OPT_STAGGER) != 0;
    // OPT_ONES: the softmax sections run at raised priority. On one SIMD the arbiter otherwise lets the OLDER wave's
    // ready-but-blocked MFMA hold the VALU port, and the partner's VALU work does not overlap with the matrix pipe at
    // all (tools/ubench/overlap2.hip: MFMA wave + softmax-mix wave 380 us = the sum; with the VALU wave at
    // s_setprio 3: 262 us, MFMA hidden).
    auto valu_prio = [&](bool on) {
      if constexpr ((OPT & OPT_ONES) != 0) {
        if (on) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
      }
    };
    h8 pf[2 * BCB];
    auto row_max_and_rescale = [&]() {
      float mx = s[0][0];
#pragma unroll
      for (int kb2 = 0; kb2 < BCB; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb2][r]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      // growth of the row max over the running one, in the log2 domain (OPT_PRE: the scores are already relative)
      const float d = PRE ? mx : mx * scale_log2e - m_run;
      bool grow;
      if constexpr ((OPT & OPT_DEFER) != 0) grow = d > 8.0f;
      else grow = d > 0.f;
      const bool first = PRE && j == 0;  // tile 0 adopts its max unconditionally (the accumulators started at 0)
      if (first || __builtin_amdgcn_ballot_w64(grow) != 0) {
        // OPT_PRE: m_run += delta (delta is exact: the scores are relative). Without it m_run starts at the -1e30
        // sentinel, where "m_run + (mxs - m_run)" would cancel catastrophically: take the max directly.
        const float delta = first ? d : fmaxf(d, 0.f);
        const float m_new = PRE ? m_run + delta : fmaxf(m_run, mx * scale_log2e);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
        if constexpr (SUMM) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
        }
        if constexpr (PRE) {  // the pending scores were accumulated from the old -m
#pragma unroll
          for (int kb2 = 0; kb2 < BCB; ++kb2)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb2][r] -= delta;
#pragma unroll
          for (int r = 0; r < 16; ++r) minit[r] = -m_run;
          asm volatile("" : "+v"(minit));
        }
#pragma unroll
        for (int b = 0; b < DHR / 32; ++b)
#pragma unroll
          for (int r = 0; r < 16; r += 4) {  // serialise the register round trips of the rescale
            float t0 = ot[b][r], t1 = ot[b][r + 1], t2 = ot[b][r + 2], t3 = ot[b][r + 3];
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
            ot[b][r] = t0 * alpha, ot[b][r + 1] = t1 * alpha, ot[b][r + 2] = t2 * alpha, ot[b][r + 3] = t3 * alpha;
          }
      }
    };
    auto p_half = [&](int h) {  // fragments u = h*BCB .. (h+1)*BCB - 1
      const float nm = -m_run;
      float psum = 0.f;
#pragma unroll
      for (int u = h * BCB; u < (h + 1) * BCB; ++u)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const int kb2 = u >> 1, r = (u & 1) * 8 + e;
          const float x0 = PRE ? s[kb2][r] : fmaf(s[kb2][r], scale_log2e, nm);
          const float x1 = PRE ? s[kb2][r + 1] : fmaf(s[kb2][r + 1], scale_log2e, nm);
          const float a0 = (ABL & 2) ? s[kb2][r] : __builtin_amdgcn_exp2f(x0);
          const float a1 = (ABL & 2) ? s[kb2][r + 1] : __builtin_amdgcn_exp2f(x1);
          if constexpr (!SUMM) psum += a0 + a1;
          const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
          pf[u][e] = a[0], pf[u][e + 1] = a[1];
        }
      if constexpr (!SUMM) l_run += psum;
    };
    if constexpr (SPLIT) {
      valu_prio(true);
      row_max_and_rescale();
      p_half(0);
      valu_prio(false);
    }
    if constexpr (NSP == 2 && !(ABL & 4))
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f4*>(sx_mine + q * 1024) = f4{s[0][4 * q], s[0][4 * q + 1], s[0][4 * q + 2], s[0][4 * q + 3]};
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the partial is in LDS before the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    mark(j, 2);
    // ================= phase B: S = own + partner's partial, softmax, O^T[half] += V[:, half]^T P^T
    if constexpr (NSP == 2 && !(ABL & 4))
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 p = *reinterpret_cast<const f4*>(sx_peer + q * 1024);
      s[0][4 * q] += p[0], s[0][4 * q + 1] += p[1], s[0][4 * q + 2] += p[2], s[0][4 * q + 3] += p[3];
    }
    h8 vf[PD];  // first V fragments fly under the softmax
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    if (PD > 1) __builtin_amdgcn_sched_barrier(0);
    mark(j, 3);
    if constexpr (!SPLIT) {
      valu_prio(true);
      row_max_and_rescale();
      p_half(0);
      p_half(1);
      valu_prio(false);
    }
    mark(j, 4);
    auto pv_range = [&](int i0, int i1) {
      if (!(ABL & 8)) {
#pragma unroll
        for (int idx = i0; idx < i1; ++idx) {
          const int st = idx / (DHR / 32), b = idx % (DHR / 32);
          ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx % PD], pf[st], ot[b], 0, 0, 0);
          cln_mfma_keep(ot[b], vf[idx % PD], pf[st]);
          if constexpr (SUMM)
            if (b == DHR / 32 - 1) lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[st], lacc, 0, 0, 0);
          if (idx + PD < NPV) vf[idx % PD] = v_frag(idx + PD);
          if (PD > 1 || (idx & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if constexpr (SPLIT) {
      pv_range(0, NPV / 2);  // fragments of the first half are ready since phase A
      valu_prio(true);
      p_half(1);             // VALU under those MFMAs
      valu_prio(false);
      __builtin_amdgcn_sched_barrier(0);
      pv_range(NPV / 2, NPV);
    } else {
      pv_range(0, NPV);
    }
    mark(j, 5);
    // own DMA pieces of tile j+1 landed; everyone behind this barrier is done with what the next phase overwrites
    hgemm::wait_vmcnt<0>();
    mark(j, 6);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    mark(j, 7);
  }
  if (grp == 0) {  // group 1's last phase B: keep the barrier count equal and the ring intact until it is done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  if constexpr ((ABL & 128) != 0) life[8] = __builtin_amdgcn_s_memrealtime();
  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows)
  float l_tot;
  if constexpr (SUMM) {
    l_tot = lacc[0];
  } else {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  char* ob = smem + wave * (32 * G::OS);
  // lane-derived epilogue addresses are formed HERE from a recomputed lane id: formed at kernel entry they were carried
  // across the KV loop in a full register file, i.e. spilled (15 dwords at D = 512, 11 at D = 256)
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
#pragma unroll
  for (int b = 0; b < DHR / 32; ++b) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = DHR / 8;
  half_t* og = O + head + (size_t)q_row0 * DR + part * DHR;
#pragma unroll 4
  for (int it = 0; it < (32 * LPR) / 64; ++it) {
    const int idx = it * 64 + lane_e;
    const int row = idx / LPR, c = idx % LPR;
    const u4 v = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    *reinterpret_cast<u4*>(og + (size_t)row * DR + c * 8) = v;
  }
  if constexpr ((ABL & 128) != 0) {
    life[9] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the O stores have been acknowledged
    life[10] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hwid));
    life[11] = hwid;
    if (lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(og);
#pragma unroll
      for (int i = 0; i < 12; ++i) dbg[i] = life[i];
    }
  }
  if constexpr ((ABL & 32) != 0) {  // probe only: block 0 overwrites the head of O with its 8 x 8 time stamps
    if (blockIdx.x == 0 && lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(O);
#pragma unroll
      for (int i = 0; i < 8; ++i) dbg[wave * 8 + i] = stamp[i];
    }
  }
}

template <int D, int NSP, int BCB, int OPT, int ABL = 0, int DREAL = 0>
int launch_dsplit(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoSplit<D, NSP, BCB>;
  constexpr int dreal = DREAL ? DREAL : D;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_dsplit_kernel<D, NSP, BCB, OPT, ABL, DREAL>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dreal);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_dsplit_kernel<D, NSP, BCB, OPT, ABL, DREAL>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
